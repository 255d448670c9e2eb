"""Round-5 gated probe (GPU): the dense hidden-layer contraction on the int8 matrix pipe (tools/probes/i8split_probe.hip) against
the shipped float64 kernel k_jet_gemm<double,4,5,2>, on REAL layer jets of the benchmark cell.

  python tools/i8probe.py [--walkers 4096] [--reps 5]

Takes the layer-2 input tiles (stage 'g2': 256 one-electron rows + 64 pair-mean rows, 80 jet slots) of 8 synthetic bcc-Li walkers
from the library, tiles them to 24 x 4096 electron tiles, and reports
  * ms per launch of the slicing kernel and of the int8 contraction (HIP events on the launch stream),
  * the shipped dense layer's ms per 4096-walker launch in the same process (the library's own event pairs),
  * the error of the int8 result against a float64 reference on the distinct tiles.
Prints one JSON object (committed as profiles/r05_i8split_probe.json).
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--walkers', type=int, default=4096)
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--distinct', type=int, default=8, help='walkers whose jets are taken from the library')
    args = ap.parse_args()
    lib = C.CDLL(os.path.join(ROOT, 'tools', 'probes', 'libi8probe.so'))
    for f in ('i8p_xp_bytes', 'i8p_xs_bytes', 'i8p_wp_bytes'):
        getattr(lib, f).restype = C.c_int64
    lib.i8p_xp_bytes.argtypes = [C.c_int64, C.c_int]
    lib.i8p_xs_bytes.argtypes = [C.c_int64, C.c_int]
    lib.i8p_wp_bytes.argtypes = [C.c_int, C.c_int]
    vp = C.c_void_p
    lib.i8p_prep_w.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp]
    lib.i8p_slice.argtypes = [vp, C.c_int64, C.c_int64, C.c_int, vp, vp, vp]
    lib.i8p_gemm.argtypes = [vp, vp, vp, vp, vp, C.c_int64, C.c_int, C.c_int, vp]
    lib.i8p_ref.argtypes = [vp, C.c_int64, C.c_int64, vp, C.c_int, C.c_int, vp, vp, vp]

    from deepsolid_amd import network, systems, hamiltonian
    dev = torch.device('cuda', 0)
    cell, klist = systems.build('bcc_li')
    net_kw = dict(systems.DETNET_DEFAULTS)
    net = network.make_solid_fermi_net(klist=klist, simulation_cell=cell, method_name='eval_logdet', **net_kw)
    params = net.init(0)
    sysd = net.apply.system
    N = sum(cell.nelec)
    P = (3 * N + 2 + 15) // 16 * 16
    h1, h2 = net_kw['hidden_dims'][1]
    A = np.asarray(cell.original_cell.atom_coords()).reshape(-1, 3).shape[0]
    ldk = max(max(a + 2 * b for a, b in [(4 * A, 4)] + [tuple(h) for h in net_kw['hidden_dims'][:-1]]), net_kw['hidden_dims'][-1][0])
    K, Nout = h1 + 2 * h2, net_kw['hidden_dims'][2][0]
    nd = args.distinct
    x = torch.as_tensor(systems.synthetic_walkers(cell, nd, seed=1234), device=dev)
    g2 = sysd.debug_stage(params, x, 'g2', nd * N * ldk * P).reshape(nd * N, ldk, P)[:, :K, :].contiguous()      # (tiles, 320, 80)
    wfull = torch.as_tensor(params['single'][2]['w']).to(dev, torch.float64)                                       # (832, 256)
    W = torch.cat([wfull[:h1], wfull[-2 * h2:]], 0).contiguous()                                                     # per-electron rows
    assert W.shape == (K, Nout)
    nt_d = nd * N
    ntiles = args.walkers * N
    reps_t = (ntiles + nt_d - 1) // nt_d
    X = g2.repeat(reps_t, 1, 1)[:ntiles].contiguous()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ptr = lambda t: C.c_void_p(t.data_ptr())
    XP = torch.empty(lib.i8p_xp_bytes(ntiles, K), dtype=torch.uint8, device=dev)
    XS = torch.empty(lib.i8p_xs_bytes(ntiles, K) // 8, dtype=torch.float64, device=dev)
    WP = torch.empty(lib.i8p_wp_bytes(K, Nout), dtype=torch.uint8, device=dev)
    SW = torch.empty(Nout, dtype=torch.float64, device=dev)
    Z = torch.empty(ntiles, Nout, P, dtype=torch.float64, device=dev)
    assert lib.i8p_prep_w(ptr(W), K, Nout, ptr(WP), ptr(SW), st) == 0

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return ts

    t_slice = timed(lambda: lib.i8p_slice(ptr(X), ntiles, K * P, K, ptr(XP), ptr(XS), st), args.reps)
    t_gemm = timed(lambda: lib.i8p_gemm(ptr(XP), ptr(XS), ptr(WP), ptr(SW), ptr(Z), ntiles, K, Nout, st), args.reps)
    # float64 reference on the distinct tiles
    Zr = torch.empty(nt_d, Nout, P, dtype=torch.float64, device=dev)
    Za = torch.empty_like(Zr)
    assert lib.i8p_ref(ptr(X), nt_d, K * P, ptr(W), K, Nout, ptr(Zr), ptr(Za), st) == 0
    torch.cuda.synchronize()
    Zi = Z[:nt_d]
    d = (Zi - Zr).abs()
    colmax = Zr.abs().amax(dim=1, keepdim=True).clamp_min(1e-300)          # per (tile, slot): max over the features
    res = {
        'workload': f'bcc-Li 24 e-, layer-2 input jets of {nd} walkers tiled to {args.walkers} walkers: {ntiles} tiles x (K = {K}) x {Nout} features x {P} slots',
        'planes': 6, 'products': 21, 'fraction_bits': 47,
        'slice_ms': t_slice, 'gemm_ms': t_gemm,
        'slice_ms_median': float(np.median(t_slice)), 'gemm_ms_median': float(np.median(t_gemm)),
        'err_max_abs': float(d.max()),
        'err_rel_to_max': float(d.max() / Zr.abs().max()),
        'err_rel_to_column_max': float((d / colmax).max()),
        'err_rel_to_sum_abs': float((d / Za.clamp_min(1e-300)).max()),
        'last_tile_matches_first_copy': bool(torch.equal(Z[(reps_t - 1) * nt_d:ntiles], Z[:ntiles - (reps_t - 1) * nt_d])),
    }
    # the shipped float64 dense layer in the same process
    xb = torch.as_tensor(systems.synthetic_walkers(cell, args.walkers, seed=1234), device=dev)
    le = hamiltonian.local_energy_seperate(net.apply, cell)
    le(params, xb)
    torch.cuda.synchronize()
    sysd.profile(True)
    for _ in range(2):
        le(params, xb)
    torch.cuda.synchronize()
    prof = sysd.profile_read()
    sysd.profile(False)
    res['f64_kernel_ms_per_launch'] = {k: v[0] / max(v[1], 1) for k, v in prof.items() if v[1]}
    res['f64_dense_layer_ms'] = prof['single_hidden'][0] / prof['single_hidden'][1]
    res['speedup_gemm_only'] = res['f64_dense_layer_ms'] / res['gemm_ms_median']
    res['speedup_with_standalone_slicing'] = res['f64_dense_layer_ms'] / (res['gemm_ms_median'] + res['slice_ms_median'])
    print(json.dumps(res))


if __name__ == '__main__':
    main()
