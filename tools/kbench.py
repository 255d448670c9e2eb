#!/usr/bin/env python3
"""Kernel iteration harness (GPU box): per-kernel HIP-event times of the local-energy chain on
bcc-Li + a correctness spot-check against the forward-Laplacian oracle."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from deepsolid_amd import network, systems, hamiltonian

ap = argparse.ArgumentParser()
ap.add_argument('--system', default='bcc_li')
ap.add_argument('--batch', type=int, default=1024)
ap.add_argument('--steps', type=int, default=3)
ap.add_argument('--check', type=int, default=2)
ap.add_argument('--dtype', default='f64')
ap.add_argument('--noprof', action='store_true')
ap.add_argument('--S', default='0', help="supercell multiplier passed to the system builder: an integer, or 'a,b,c' for diag(a, b, c) (0: its default)")
args = ap.parse_args()
dtype = torch.float64 if args.dtype == 'f64' else torch.float32
_S = [int(v) for v in str(args.S).split(',')]
_S = (np.diag(_S) if len(_S) == 3 else _S[0])
cell, klist = systems.build(args.system, **({'S': _S} if str(args.S) != '0' else {}))
net_kw = dict(systems.DETNET_DEFAULTS)
net = network.make_solid_fermi_net(klist=klist, simulation_cell=cell, method_name='eval_logdet', dtype=dtype, **net_kw)
params = net.init(0)
x_np = systems.synthetic_walkers(cell, args.batch)
x = torch.as_tensor(x_np, dtype=dtype, device='cuda')
el = hamiltonian.local_energy_seperate(net.apply, cell)
ke, ew = el(params, x)
torch.cuda.synchronize()
sysd = net.apply.system
sysd.profile(not args.noprof)
t0 = time.perf_counter()
for _ in range(args.steps):
    ke, ew = el(params, x)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.steps
prof = sysd.profile_read()
sysd.profile(False)
print(f'{args.system}{" S=" + str(args.S) if str(args.S) != "0" else ""} N={sum(cell.nelec)} B={args.batch} {args.dtype}: {dt * 1e3:.2f} ms/step = {args.batch / dt:.0f} evals/s')
print('  ' + '  '.join(f'{k}={v[0] / args.steps:.2f}' for k, v in prof.items() if v[1]))
if args.check:
    from oracle import forward_laplacian as ofl, network as onet, ewaldsum as oew
    p = onet.params_to_torch({k: [{kk: vv.cpu().double().numpy() for kk, vv in d.items()} for d in v] for k, v in params.items()})
    errs = []
    for b in range(args.check):
        ref = complex(ofl.stages(p, torch.as_tensor(x_np[b]), klist, cell, net_kw)['ke'])
        errs.append(abs(complex(ke[b].cpu()) - ref) / max(1.0, abs(ref)))
    print(f'  max rel |dKE| vs oracle over {args.check} walkers: {max(errs):.3e}')
# value chain: log psi and one mcmc_step (20 moves)
from deepsolid_amd import qmc
slog = network.make_solid_fermi_net(klist=klist, simulation_cell=cell, method_name='eval_slogdet', dtype=dtype, **net_kw)
slog.apply(params, x); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    lp = slog.apply(params, x)
torch.cuda.synchronize()
print(f'  logpsi (value chain): {(time.perf_counter() - t0) / args.steps * 1e3:.2f} ms per batch')
step = qmc.make_mcmc_step(slog.apply, args.batch, cell.a, steps=20)
xx, pm = step(params, x, 1, 0.02); torch.cuda.synchronize()
t0 = time.perf_counter()
xx, pm = step(params, x, 2, 0.02)
torch.cuda.synchronize()
print(f'  mcmc_step (20 moves): {(time.perf_counter() - t0) * 1e3:.1f} ms, pmove = {float(pm):.3f}')
