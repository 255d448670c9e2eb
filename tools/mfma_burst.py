#!/usr/bin/env python3
"""fp64 MFMA probe at different burst lengths (is the 51.5 TF/s ceiling a sustained-clock effect?)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepsolid_amd import _lib
lib = _lib.load()
scratch = torch.zeros(16, dtype=torch.float64, device='cuda')
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for iters in (200, 1000, 5000, 25000, 125000):
    res = []
    for rep in range(3):
        torch.cuda.synchronize()
        import time; time.sleep(0.2)                      # let the clocks relax between bursts
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fl = lib.ds_mfma_f64_peak(iters, 2, 8, C.c_void_p(scratch.data_ptr()), st)
        e1.record(); torch.cuda.synchronize()
        res.append(fl / (e0.elapsed_time(e1) * 1e-3) / 1e12)
    print(iters, 'iters:', ' '.join(f'{r:.1f}' for r in res), 'TF/s  (%.2f ms)' % (fl / max(res) / 1e9))
