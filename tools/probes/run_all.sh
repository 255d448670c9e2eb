#!/bin/bash
# builds (if needed) and runs the MFMA probes; output -> gpurun_out/mfma44_probe.txt
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for p in mfma44_probe mfma44_rate mfma44_loop; do
  [ -x tools/probes/$p ] || hipcc -O3 --offload-arch=gfx950 tools/probes/$p.hip -o tools/probes/$p
done
{ echo "== mfma44_probe"; ./tools/probes/mfma44_probe | grep -v "^layout"; echo "== mfma44_rate"; ./tools/probes/mfma44_rate; echo "== mfma44_loop"; ./tools/probes/mfma44_loop; echo "== clock"; python tools/gpu_probe.py clock; } > gpurun_out/mfma44_probe.txt 2>&1 < /dev/null
tail -30 gpurun_out/mfma44_probe.txt
