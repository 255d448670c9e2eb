"""Writes tests/golden/f32_budget.npz: per fixture walker, the float64 forward-Laplacian oracle E_kin at the float32-rounded walker
and what the same restatement loses when it runs in float32 on the CPU (tests/common.py::compute_float32_budget_walker) -- the
per-walker error budget of the float32 GPU tests.  The numbers are oracle outputs on fixture inputs (data, not code); the tests
recompute any entry that is missing.  Usage: python tools/f32_budget_fixture.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import common                      # noqa: E402

CASES = {'lih': 2, 'bcc_li': 2, 'diamond': 4}

if __name__ == '__main__':
    out = {}
    for name, nb in CASES.items():
        pairs = [common.compute_float32_budget_walker(name, b) for b in range(nb)]
        out[f'{name}_ref'] = np.asarray([p[0] for p in pairs])
        out[f'{name}_loss'] = np.asarray([p[1] for p in pairs])
        print(name, out[f'{name}_ref'], out[f'{name}_loss'])
    np.savez(os.path.join(common.GOLDEN, 'f32_budget.npz'), **out)
