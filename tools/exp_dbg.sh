#!/bin/bash
# Round-3 timing experiments behind DESIGN.md section 4b (GPU box, repo root; output kept as profiles/r03_layer_experiments.txt).
# Needs a library built with `make -C deepsolid_amd/csrc clean && make -C deepsolid_amd/csrc EXP=1`: in the shipped build the
# switches below are compiled out (tests/test_gpu_parity.py::test_debug_switches_cannot_change_results).  Rebuild WITHOUT EXP
# afterwards.  NOTE: DS_LG_DBG bits 1 / 2 switch parts of the
# arithmetic off -- the energies of those runs are WRONG by construction; only the kernel times are read.
echo "## per-electron hidden GEMM k_jet_gemm<double,4,5,2>, 4096 walkers, ms per step of the two hidden layers (DS_LG_DBG: 1 = no epilogue, 2 = accumulators start at zero instead of S)"
for d in 0 1 2 3; do echo "DS_LG_DBG=$d $(DS_LG_DBG=$d python tools/kbench.py --batch 4096 --steps 3 --check 0 2>&1 | grep -E 'single_hidden' | sed 's/.*single_first/single_first/' | cut -c1-48)"; done
echo "## both layer paths, 1024 walkers (tools/layer_check.py): stage agreement and per-kernel ms"
python tools/layer_check.py --systems bcc_li --batch 8 --time-batch 1024 2>&1 | grep -v "^/opt" | cut -c1-330
echo "## phase stamps of one workgroup of the per-electron hidden GEMM (tools/gemm_timeline.py; shader-clock cycles)"
python tools/gemm_timeline.py 2>&1 | grep -v "^/opt"
echo "## phase stamps of one wave of the electron-group kernel (DS_LAYER_GROUPS=1, tools/layer_timeline.py)"
DS_LAYER_GROUPS=1 python tools/layer_timeline.py 2>&1 | grep -v "^/opt"
