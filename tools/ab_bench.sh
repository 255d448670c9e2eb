#!/bin/bash
# A/B of one environment switch on the bench step: tools/ab_bench.sh VAR valueA valueB [bench args]
# ("default" = unset).  Prints ms_per_step and the per-kernel breakdown of each side; JSON lines under gpurun_out/.
VAR=$1; A=$2; B=$3; shift 3
mkdir -p gpurun_out
for v in "$A" "$B"; do
  if [ "$v" = default ]; then unset $VAR; else export $VAR=$v; fi
  python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-mcmc --no-int8 "$@" 2>gpurun_out/ab_${VAR}_$v.err | tail -1 > gpurun_out/ab_${VAR}_$v.json
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_${VAR}_$v.json"))
k=d["kernel_ms_per_step"]
print("$VAR=$v", "ms_per_step %.2f" % d["ms_per_step"], "sum %.2f" % sum(k.values()), {a: round(b, 2) for a, b in k.items()})
PY
done
