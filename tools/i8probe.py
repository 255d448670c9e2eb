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
    ap.add_argument('--lmodes', default='', help='the same for the fused slicing + contraction kernel')
    ap.add_argument('--modes', default='', help='timing-experiment modes of the contraction kernel to run as well (comma list)')
    ap.add_argument('--rate', action='store_true', help='only the bare v_mfma_i32_16x16x64_i8 issue-rate measurement')
    ap.add_argument('--distinct', type=int, default=8, help='walkers whose jets are taken from the library')
    args = ap.parse_args()
    lib = C.CDLL(os.path.join(ROOT, 'tools', 'probes', 'libi8probe.so'))
    for f in ('i8p_xp_bytes', 'i8p_wp_bytes'):
        getattr(lib, f).restype = C.c_int64
    lib.i8p_xp_bytes.argtypes = [C.c_int64, C.c_int]
    lib.i8p_wp_bytes.argtypes = [C.c_int, C.c_int]
    vp = C.c_void_p
    lib.i8p_prep_w.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp]
    lib.i8p_slice.argtypes = [vp, C.c_int64, C.c_int64, C.c_int, vp, vp]
    lib.i8p_gemm.argtypes = [vp, vp, vp, vp, C.c_int64, C.c_int, C.c_int, C.c_int, vp]
    lib.i8p_layer.argtypes = [vp, C.c_int64, vp, vp, vp, C.c_int64, C.c_int, C.c_int, C.c_int, vp, vp]
    lib.i8p_ref.argtypes = [vp, C.c_int64, C.c_int64, vp, C.c_int, C.c_int, vp, vp, vp]

    lib.i8p_rate.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp]
    if args.rate:
        dev = torch.device('cuda', 0)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        out = {}
        for threads in (256, 512):
            for nacc in (4, 8, 16):
                clk = torch.zeros(2, dtype=torch.int64, device=dev)
                sink = torch.zeros(4, dtype=torch.int32, device=dev)
                iters = 20000
                run = lambda: lib.i8p_rate(256, threads, iters, nacc, C.c_void_p(clk.data_ptr()), C.c_void_p(sink.data_ptr()), st)
                run()
                torch.cuda.synchronize()
                clk.zero_()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                run()
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1)
                c = clk.cpu().numpy()
                n_mfma = 256 * (threads // 64) * iters * nacc
                ghz = c[0] / c[1] * 0.1
                out[f'{threads}thr_nacc{nacc}'] = dict(ms=ms, tops=n_mfma * 32768 / ms * 1e-9, ghz=float(ghz),
                                                       cycles_per_mfma_per_simd=float(ms * 1e-3 * ghz * 1e9 / (n_mfma / 1024)))
        lib.i8p_burst_rate.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp]
        bout = {}
        for threads in (256, 512):
            for order in (0, 1, 2):
                clk = torch.zeros(2, dtype=torch.int64, device=dev)
                sink = torch.zeros(4, dtype=torch.int32, device=dev)
                iters = 4000
                run = lambda: lib.i8p_burst_rate(256, threads, iters, order, C.c_void_p(clk.data_ptr()), C.c_void_p(sink.data_ptr()), st)
                run()
                torch.cuda.synchronize()
                clk.zero_()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                run()
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1)
                c = clk.cpu().numpy()
                n_mfma = 256 * (threads // 64) * iters * 21 * (2 if order == 2 else 1)
                ghz = c[0] / c[1] * 0.1
                bout[f'{threads}thr_order{order}'] = dict(ms=ms, ghz=float(ghz), cycles_per_mfma_per_simd=float(ms * 1e-3 * ghz * 1e9 / (n_mfma / 1024)))
        print(json.dumps({'i8_mfma_rate': out, 'burst_rate': bout}))
        return
    os.environ.pop('DS_I8', None)         # the library's own dense layer as the float64 kernel (its default again since round 6)
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

    t_slice = timed(lambda: lib.i8p_slice(ptr(X), ntiles, K * P, K, ptr(XP), st), args.reps)
    modes = {}
    for m in [int(v) for v in args.modes.split(',') if v]:
        modes[m] = float(np.median(timed(lambda: lib.i8p_gemm(ptr(XP), ptr(WP), ptr(SW), ptr(Z), ntiles, K, Nout, m, st), 3)))
    t_gemm = timed(lambda: lib.i8p_gemm(ptr(XP), ptr(WP), ptr(SW), ptr(Z), ntiles, K, Nout, 0, st), args.reps)
    Z2 = torch.empty_like(Z)
    lmodes = {}
    clk = torch.zeros(16, dtype=torch.int64, device=dev)
    phases = {}
    for m in [int(v) for v in args.lmodes.split(',') if v]:
        clk.zero_()
        lmodes[m] = float(np.median(timed(lambda: lib.i8p_layer(ptr(X), K * P, ptr(WP), ptr(SW), ptr(Z2), ntiles, K, Nout, m, ptr(clk), st), 3)))
        if m & 32:
            c = clk.cpu().numpy().astype(float)
            for w, o in (('wave0', 0), ('wave5', 8)):
                tot = c[o]
                phases[f'mode{m}_{w}'] = dict(ghz=c[o] / c[o + 1] * 0.1, barrier=c[o + 2] / tot, slicing=c[o + 3] / tot, bursts=c[o + 4] / tot,
                                              epilogue=c[o + 5] / tot, cycles_per_tile=tot / 4 / 256 / (ntiles / 256))
    t_layer = timed(lambda: lib.i8p_layer(ptr(X), K * P, ptr(WP), ptr(SW), ptr(Z2), ntiles, K, Nout, 0, ptr(clk), st), args.reps)
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
        'slice_ms': t_slice, 'gemm_ms': t_gemm, 'experiment_modes_ms': modes,
        'slice_ms_median': float(np.median(t_slice)), 'gemm_ms_median': float(np.median(t_gemm)),
        'layer_ms': t_layer, 'layer_ms_median': float(np.median(t_layer)), 'layer_experiment_modes_ms': lmodes, 'layer_phase_shares': phases,
        'layer_equals_gemm_on_sliced_planes': bool(torch.equal(Z2, Z)),
        'layer_err_rel_to_column_max': float(((Z2[:nt_d] - Zr).abs() / colmax).max()),
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
    res['speedup_layer_kernel_slicing_included'] = res['f64_dense_layer_ms'] / res['layer_ms_median']
    res['speedup_gemm_only'] = res['f64_dense_layer_ms'] / res['gemm_ms_median']
    res['speedup_with_standalone_slicing'] = res['f64_dense_layer_ms'] / (res['gemm_ms_median'] + res['slice_ms_median'])
    print(json.dumps(res))


if __name__ == '__main__':
    main()
