#!/bin/bash
# Timing experiments on the low-rank layer kernel k_layer1_lr (GPU box; needs a library built with `make EXP=1`, loaded through
# DEEPSOLID_HIP_LIB).  DS_DBG bits << 8: 1 no epilogue, 2 phase 1 cut to 4 k-steps, 4 phase 2 cut to 4 k-steps, 8 no S1 loads.
# The energies of these runs are WRONG by construction; only single_hidden (LR layer + dense layer 2) is read.
for d in 0 256 512 1024 2048 768 1792 3840; do
  echo "DS_DBG=$d $(DS_DBG=$d python tools/kbench.py --batch 4096 --steps 3 --check 0 2>&1 | grep -E 'single_hidden' | sed 's/.*single_first/single_first/' | cut -c1-60)"
done
