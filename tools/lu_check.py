#!/usr/bin/env python3
"""GPU box: log det by the one-lane-per-row register LU (k_det_lu_wave, default for 16 < n <= 64) against the Gauss-Jordan
inverse kernel (DS_NO_LU_WAVE=1): log|psi|, phase, and the time of a forward with each.
    python tools/lu_check.py [system] [batch] [f64|f32]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepsolid_amd import network, systems

name = sys.argv[1] if len(sys.argv) > 1 else 'graphene'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dtype = torch.float32 if (len(sys.argv) > 3 and sys.argv[3] == 'f32') else torch.float64
cell, klist = systems.build(name)
net_kw = dict(systems.DETNET_DEFAULTS)
x = torch.as_tensor(systems.synthetic_walkers(cell, B), dtype=dtype, device='cuda')
from deepsolid_amd.device import DeviceSystem
from deepsolid_amd.ewaldsum import EwaldTables
net = network.make_solid_fermi_net(klist=klist, simulation_cell=cell, method_name='eval_phase_and_slogdet', dtype=dtype, **net_kw)
params = net.init(0)
out = {}
for tag, env in (('gauss-jordan', '1'), ('lu-wave', None)):
    if env: os.environ['DS_NO_LU_WAVE'] = env
    else: os.environ.pop('DS_NO_LU_WAVE', None)
    sd = DeviceSystem(cell, klist, net_kw, EwaldTables(cell), dtype)        # (a new library handle: the switch is read at create)
    la, ph = sd.logpsi(params, x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): sd.logpsi(params, x)
    torch.cuda.synchronize()
    out[tag] = (ph, la, (time.perf_counter() - t0) / 10)
    print(f'[{tag}] forward {out[tag][2] * 1e3:.3f} ms per {B} walkers', flush=True)
dl = (out['lu-wave'][1] - out['gauss-jordan'][1]).abs().max().item()
pa, pb = out['lu-wave'][0], out['gauss-jordan'][0]
dp = (pa - pb).abs().max().item()
print(f'{name} {dtype}: max |d log|psi|| = {dl:.3e} (|log psi| ~ {out["gauss-jordan"][1].abs().max().item():.1f}), max |d phase| = {dp:.3e}, nan {int(torch.isnan(out["lu-wave"][1]).sum())}')
