#!/usr/bin/env python3
"""GPU box probe: device properties + fp64 MFMA issue-rate micro-benchmark (ds_mfma_f64_peak)."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from deepsolid_amd import _lib


def mfma_f64_peak(blocks_per_cu=2, n_acc=8, iters=100000):
    lib = _lib.load()
    scratch = torch.zeros(16, dtype=torch.float64, device='cuda')
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    lib.ds_mfma_f64_peak(1000, blocks_per_cu, n_acc, C.c_void_p(scratch.data_ptr()), st)
    torch.cuda.synchronize()
    best = 0.0
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        flops = lib.ds_mfma_f64_peak(iters, blocks_per_cu, n_acc, C.c_void_p(scratch.data_ptr()), st)
        e1.record()
        torch.cuda.synchronize()
        best = max(best, flops / (e0.elapsed_time(e1) * 1e-3) / 1e12)
    return best


if __name__ == '__main__':
    p = torch.cuda.get_device_properties(0)
    sweep = {f'{w}wave_per_simd_{a}acc': round(mfma_f64_peak(w, a, 400000 // (w * a)), 2)
             for w in (1, 2, 4) for a in (1, 4, 8, 16)}
    info = dict(name=p.name, cus=p.multi_processor_count, mem_gb=round(p.total_memory / 2 ** 30, 1),
                clock_mhz=getattr(p, 'clock_rate', 0) / 1000, mfma_f64_16x16x4_tflops=sweep)
    print(json.dumps(info))
