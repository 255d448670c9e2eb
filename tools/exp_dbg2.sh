#!/bin/bash
# Second part of the round-3 timing experiments (electron-group kernels; needs a `make EXP=1` library, see tools/exp_dbg.sh).
# The energies of the runs with DS_LG_DBG bits other than 8 / 16 / 32 are WRONG by construction; only kernel times are read.
run() { echo "$1: $(env $2 DS_LAYER_GROUPS=1 python tools/kbench.py --batch 1024 --steps 3 --check 0 2>&1 | grep -E 'single_hidden' | sed 's/.*single_first/single_first/' | cut -c1-48)"; }
echo "## electron-group kernels (DS_LAYER_GROUPS=1), 1024 walkers, ms per step: layer 0 (single_first), the two hidden layers (single_hidden)"
run "as shipped                                   " "DS_LG_DBG=0"
run "no epilogue arithmetic (1)                   " "DS_LG_DBG=1"
run "no spin sums in LDS (64)                     " "DS_LG_DBG=64"
run "no row sums (128)                            " "DS_LG_DBG=128"
run "no G stores (256)                            " "DS_LG_DBG=256"
run "no per-electron Y / SSP stores (512)         " "DS_LG_DBG=512"
run "none of the four (960)                       " "DS_LG_DBG=960"
run "k-loop 8 instead of 80 k-steps (4)           " "DS_LG_DBG=4"
run "start skew by hardware wave slot, 5 x 8128 cyc " "DS_LG_DBG=5242888"
run "start skew by workgroup hash, 0..15 x 8128 cyc" "DS_LG_DBG=1048592"
run "24 KB more LDS (88 KB): 1 workgroup per CU    " "DS_LG_PAD_LDS=24576 DS_LG_DBG=0"
run "four-deep operand ring                       " "DS_LG_RING=4 DS_LG_DBG=0"
run "hidden layers gather the pair rows too       " "DS_LG_GATHER=1 DS_LG_DBG=0"
