#!/bin/bash
# Safe wrapper for gpurun: own timeout, stdin closed (a stray `head`/`cat` cannot hang the box).
# usage: tools/grun.sh SECONDS 'command'
T=$1; shift
exec /usr/local/graft/bin/gpurun --timeout $((T + 60)) -- "timeout $T bash -c '$*' < /dev/null"
