"""Development harness (GPU) of the two-workgroup int8 layer kernel: tools/probes/i8s_probe.hip.

    python tools/i8s_probe.py [--walkers 4096] [--reps 5] [--lib tools/probes/libi8sprobe.so]

Random layer-like jets (per-column magnitudes spread over 2^-20 .. 2^4, zero padding slots 74..79), random weights and shared
term; runs ds_i8.h's 8-wave kernel and ds_i8s.h's split kernel on the same input, compares the outputs (they share the products'
arithmetic: z agrees to the last bit, the Laplacian slot to an ulp) and times both with HIP events.  Prints one JSON line."""
import argparse
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--walkers', type=int, default=4096)
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--lib', default=os.path.join(ROOT, 'tools', 'probes', 'libi8sprobe.so'))
    ap.add_argument('--skip-old', action='store_true')
    args = ap.parse_args()
    lib = C.CDLL(args.lib)
    vp = C.c_void_p
    lib.i8s_prep_w.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp]
    for f in (lib.i8s_old, lib.i8s_new):
        f.argtypes = [vp, C.c_size_t, vp, vp, vp, C.c_int, vp, C.c_int, C.c_int, vp]
    lib.i8s_wp_bytes.restype = C.c_long
    dev = torch.device('cuda', 0)
    N, K, NOUT, P, D = 24, 320, 256, 80, 74
    B = args.walkers
    nt = B * N
    g = torch.Generator(device=dev).manual_seed(1)
    distinct = min(B, 64) * N
    X = torch.randn(distinct, K, P, generator=g, device=dev, dtype=torch.float64)
    X *= torch.exp2(torch.randint(-20, 5, (distinct, 1, P), generator=g, device=dev).double())
    X[:, :, D:] = 0
    X = X.repeat((nt + distinct - 1) // distinct, 1, 1)[:nt].contiguous()
    W = torch.randn(K, NOUT, generator=g, device=dev, dtype=torch.float64) / 18.0
    Sb = torch.randn(B, NOUT, P, generator=g, device=dev, dtype=torch.float64) * 0.3
    Sb[:, :, D:] = 0
    WP = torch.zeros(lib.i8s_wp_bytes(), dtype=torch.uint8, device=dev)
    SW = torch.zeros(NOUT, dtype=torch.float64, device=dev)
    st = vp(torch.cuda.current_stream().cuda_stream)
    lib.i8s_prep_w(vp(W.data_ptr()), K, NOUT, vp(WP.data_ptr()), vp(SW.data_ptr()), st)
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    out = {}
    res = {}
    for name, fn in (('old', lib.i8s_old), ('new', lib.i8s_new)):
        if name == 'old' and args.skip_old:
            continue
        G = torch.full((nt, K, P), float('nan'), dtype=torch.float64, device=dev)
        run = lambda: fn(vp(X.data_ptr()), K * P, vp(WP.data_ptr()), vp(SW.data_ptr()), vp(Sb.data_ptr()), N, vp(G.data_ptr()), nt, ncu, st)
        run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        out[name + '_ms'] = e0.elapsed_time(e1) / args.reps
        res[name] = G[:, :NOUT, :]
    if 'old' in res:
        a, b = res['old'], res['new']
        out['nan_new'] = int(torch.isnan(b).sum())
        d = (a - b).abs()
        out['max_abs_diff'] = float(d.max())
        lap = d[:, :, 1].max()
        d[:, :, 1] = 0
        out['max_abs_diff_without_laplacian_slot'] = float(d.max())
        out['max_abs_diff_laplacian_slot'] = float(lap)
        out['max_abs_old'] = float(a.abs().max())
        bad = (a != b)
        bad[:, :, 1] = False
        if bool(bad.any()):
            # where do the kernels disagree? (tile, feature row, slot) histograms -- development aid
            idx = bad.nonzero()
            out['bad_count'] = int(idx.shape[0])
            out['bad_tiles'] = idx[:, 0].unique().tolist()[:40]
            out['bad_n_tiles'] = int(idx[:, 0].unique().numel())
            out['bad_rows'] = idx[:, 1].unique().tolist()[:70]
            out['bad_slots'] = idx[:, 2].unique().tolist()
    else:
        out['nan_new'] = int(torch.isnan(res['new']).sum())
    print(json.dumps(out))


if __name__ == '__main__':
    main()
