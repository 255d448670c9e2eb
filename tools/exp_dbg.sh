#!/bin/bash
# Timing experiments on the hidden GEMM behind DESIGN.md section 4 (GPU box, repo root).
# Needs a library built with `make -C deepsolid_amd/csrc clean && make -C deepsolid_amd/csrc EXP=1`: in the shipped build the
# switches below are compiled out (tests/test_gpu_parity.py::test_debug_switches_cannot_change_results).  Rebuild WITHOUT EXP
# afterwards.  NOTE: DS_DBG bits 1 / 2 switch parts of the
# arithmetic off -- the energies of those runs are WRONG by construction; only the kernel times are read.
echo "## per-electron hidden GEMM k_jet_gemm<double,4,5,2>, 4096 walkers, ms per step of the two hidden layers (DS_DBG: 1 = no epilogue, 2 = accumulators start at zero instead of S)"
for d in 0 1 2 3; do echo "DS_DBG=$d $(DS_DBG=$d python tools/kbench.py --batch 4096 --steps 3 --check 0 2>&1 | grep -E 'single_hidden' | sed 's/.*single_first/single_first/' | cut -c1-48)"; done
echo "## phase stamps of one workgroup of the hidden GEMM (tools/gemm_timeline.py; shader-clock cycles)"
python tools/gemm_timeline.py 2>&1 | grep -v "^/opt"
