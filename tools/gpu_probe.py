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


def mfma_clock(blocks_per_cu=2, n_acc=8, iters=200000):
    """Core clock during a sustained fp64 MFMA loop: shader-clock cycles / constant-rate ticks (one wave's view),
    and the MFMA pipe occupancy it implies (16-pass v_mfma_f64_16x16x4 = 64 cycles each)."""
    lib = _lib.load()
    scratch = torch.zeros(16, dtype=torch.float64, device='cuda')
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = {}
    for tag in ('cold', 'warm'):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        flops = lib.ds_mfma_f64_peak(iters, blocks_per_cu, n_acc, C.c_void_p(scratch.data_ptr()), st)
        e1.record()
        torch.cuda.synchronize()
        cyc, ticks = float(scratch[1]), float(scratch[2])
        ms = e0.elapsed_time(e1)
        out[tag] = dict(shader_cycles=cyc, realtime_ticks=ticks, kernel_ms=round(ms, 3),
                        tflops=round(flops / (ms * 1e-3) / 1e12, 2),
                        ticks_per_us=round(ticks / (ms * 1e3), 2), cycles_per_us=round(cyc / (ms * 1e3), 1),
                        mfma_per_simd=iters * n_acc * blocks_per_cu,
                        pipe_busy_frac_if_64clk=round(iters * n_acc * blocks_per_cu * 64 / cyc, 3) if cyc else None)
    return out


if __name__ == '__main__':
    p = torch.cuda.get_device_properties(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'clock':
        print(json.dumps(dict(name=p.name, mfma_f64_clock=mfma_clock())))
        sys.exit(0)
    sweep = {f'{w}wave_per_simd_{a}acc': round(mfma_f64_peak(w, a, 400000 // (w * a)), 2)
             for w in (1, 2, 4) for a in (1, 4, 8, 16)}
    info = dict(name=p.name, cus=p.multi_processor_count, mem_gb=round(p.total_memory / 2 ** 30, 1),
                clock_mhz=getattr(p, 'clock_rate', 0) / 1000, mfma_f64_16x16x4_tflops=sweep)
    print(json.dumps(info))
