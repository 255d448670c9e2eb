#!/bin/bash
# GPU box: kbench with the low-rank first hidden layer on / off (one box, same clocks).  usage: tools/ab.sh [kbench args]
echo "== low-rank layer 1 (default)"; python tools/kbench.py "$@" 2>&1 | grep -v "^/opt"

echo "== DS_NO_LOWRANK=1";           DS_NO_LOWRANK=1 python tools/kbench.py "$@" 2>&1 | grep -v "^/opt"
