#!/usr/bin/env python3
"""GPU box: the electron-group layer kernels (csrc/ds_layer.h, default path) against the per-electron k_jet_gemm path
(DS_LAYER_GROUPS=0) -- every layer output G_l, the kinetic energy, and the per-kernel times of both.

    python tools/layer_check.py [--systems lih bcc_li ...] [--batch 64] [--time-batch 1024]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from deepsolid_amd import hamiltonian, network, systems
from deepsolid_amd.device import DeviceSystem
from deepsolid_amd.ewaldsum import EwaldTables

ap = argparse.ArgumentParser()
ap.add_argument('--systems', nargs='*', default=['lih', 'bcc_li'])
ap.add_argument('--batch', type=int, default=16)
ap.add_argument('--time-batch', type=int, default=0)
ap.add_argument('--dtype', default='f64')
ap.add_argument('--hidden', default=None, help='e.g. 64,16;128,32;64,16  (hidden_dims override)')
args = ap.parse_args()
dtype = torch.float64 if args.dtype == 'f64' else torch.float32


def make(cell, klist, net_kw, groups):
    os.environ['DS_LAYER_GROUPS'] = '1' if groups else '0'
    return DeviceSystem(cell, klist, net_kw, EwaldTables(cell), dtype)


from oracle.testing import CASES

for name in args.systems:
    case = CASES.get(name, dict(system=name))             # a fixture case of oracle/testing.py (system + network options) or a bare system
    cell, klist = systems.build(case['system'], twist=case.get('twist', (0, 0, 0)), **case.get('system_kw', {}))
    if case.get('sym_type'):
        from deepsolid_amd import supercell
        supercell.set_symmetry_lat(cell, case['sym_type'])
    net_kw = dict(systems.DETNET_DEFAULTS)
    net_kw.update(case.get('net_kw', {}))
    if args.hidden:
        net_kw['hidden_dims'] = tuple(tuple(int(v) for v in h.split(',')) for h in args.hidden.split(';'))
    net = network.make_solid_fermi_net(klist=klist, simulation_cell=cell, method_name='eval_logdet', dtype=dtype, **net_kw)
    params = net.init(0)
    N = sum(cell.nelec)
    B = args.batch
    x = torch.as_tensor(systems.synthetic_walkers(cell, B), dtype=dtype, device='cuda')
    new, old = make(cell, klist, net_kw, True), make(cell, klist, net_kw, False)
    D = 3 * N + 2
    P = (D + 15) // 16 * 16
    nl = len(net_kw['hidden_dims'])
    A = np.asarray(cell.original_cell.atom_coords()).reshape(-1, 3).shape[0]
    nch = 2 if cell.nelec[1] else 1
    h1 = [4 * A] + [h[0] for h in net_kw['hidden_dims']]
    h2 = [4] + [h[1] for h in net_kw['hidden_dims']]
    ldk = max(max(h1[l] + nch * h2[l] for l in range(nl)), h1[-1])
    print(f'== {name}: N={N} P={P} ldk={ldk} B={B} {args.dtype}', flush=True)
    for l in range(1, min(nl, 3) + 1):
        gn = new.debug_stage(params, x, f'g{l}', B * N * ldk * P).cpu().numpy().reshape(B, N, ldk, P)[:, :, :h1[l]]
        go = old.debug_stage(params, x, f'g{l}', B * N * ldk * P).cpu().numpy().reshape(B, N, ldk, P)[:, :, :h1[l]]
        err = np.abs(gn - go)
        scale = max(1.0, np.abs(go).max())
        worst = np.unravel_index(np.argmax(err), err.shape)
        print(f'  g{l}: max |new - old| = {err.max():.3e} (rel {err.max() / scale:.2e}) at (walker, electron, row, slot) = {worst};'
              f' value slot {err[..., 0].max():.2e}, lap slot {err[..., 1].max():.2e}, grad slots {err[..., 2:D].max():.2e},'
              f' padding {np.abs(gn[..., D:]).max():.1e}; nan {np.isnan(gn).sum()}', flush=True)
    ke_n = new.local_energy(params, x)[0]
    ke_o = old.local_energy(params, x)[0]
    dk = (ke_n - ke_o).abs().max().item()
    print(f'  E_kin: max |new - old| = {dk:.3e}   (|E_kin| ~ {ke_o.abs().max().item():.3f}; walker 0: {(ke_n[0] - ke_o[0]).abs().max().item():.3e})', flush=True)
    if args.time_batch:
        xb = torch.as_tensor(systems.synthetic_walkers(cell, args.time_batch), dtype=dtype, device='cuda')
        for tag, sd in (('groups', new), ('per-electron', old)):
            sd.local_energy(params, xb)
            torch.cuda.synchronize()
            sd.profile(True)
            t0 = time.perf_counter()
            for _ in range(3):
                sd.local_energy(params, xb)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 3
            prof = sd.profile_read()
            sd.profile(False)
            print(f'  [{tag}] {dt * 1e3:.2f} ms per {args.time_batch} walkers = {args.time_batch / dt:.0f} evals/s: ' +
                  '  '.join(f'{k}={v[0] / 3:.2f}' for k, v in prof.items() if v[1]), flush=True)
