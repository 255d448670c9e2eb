#!/usr/bin/env python3
"""GPU box: time of one log-psi forward (value chain) and a 20-move mcmc_step: python tools/fwd_bench.py [system] [batch] [dtype]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepsolid_amd import network, qmc, systems
name = sys.argv[1] if len(sys.argv) > 1 else 'bcc_li'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dtype = torch.float32 if len(sys.argv) > 3 and sys.argv[3] == 'f32' else torch.float64
cell, klist = systems.build(name)
kw = dict(systems.DETNET_DEFAULTS)
slog = network.make_solid_fermi_net(klist=klist, simulation_cell=cell, method_name='eval_slogdet', dtype=dtype, **kw)
params = slog.init(0)
x = torch.as_tensor(systems.synthetic_walkers(cell, B), dtype=dtype, device='cuda')
for _ in range(3):
    lp = slog.apply(params, x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    lp = slog.apply(params, x)
e1.record(); torch.cuda.synchronize()
fwd = e0.elapsed_time(e1) / 20
step = qmc.make_mcmc_step(slog.apply, B, cell.a, steps=20)
xx, pm = step(params, x, 1, 0.02); torch.cuda.synchronize()
e0.record()
for r in range(3):
    xx, pm = step(params, xx, 2 + r, 0.02)
e1.record(); torch.cuda.synchronize()
lp0 = lp[0] if isinstance(lp, tuple) else lp
print(f'{name} B={B} {dtype}: DS_VAL_NB={os.environ.get("DS_VAL_NB", "auto")} sum(lp) {float(lp0.double().sum()).hex()} forward {fwd:.3f} ms  mcmc_step {e0.elapsed_time(e1) / 3:.2f} ms  pmove {float(pm):.3f}  lp[0] {float(lp[0][0] if isinstance(lp, tuple) else lp[0]):.12f}')
