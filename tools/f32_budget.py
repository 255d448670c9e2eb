#!/usr/bin/env python3
"""Float32 error budget per fixture walker: relative E_kin error of (a) the oracle's forward-Laplacian restatement run in float32
on the CPU and (b) -- on a GPU box -- the HIP float32 chain, both against the float64 oracle at the float32-rounded walker.
    python tools/f32_budget.py diamond [--gpu]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
import torch
from common import load_case
from oracle import forward_laplacian as ofl, network as onet

name = sys.argv[1]
gpu = '--gpu' in sys.argv
fx, cell, klist, net_kw, params = load_case(name)
p64 = onet.params_to_torch(params)
p32 = onet.params_to_torch(params, dtype=torch.float32)
x32 = torch.as_tensor(fx['x'], dtype=torch.float32)
ke_hip = None
if gpu:
    from deepsolid_amd import hamiltonian, network
    dp = {k: [{kk: torch.as_tensor(vv, dtype=torch.float32, device='cuda') for kk, vv in d.items()} for d in v] for k, v in params.items()}
    net = network.make_solid_fermi_net(klist=klist, simulation_cell=cell, method_name='eval_logdet', dtype=torch.float32, **net_kw)
    ke_hip = hamiltonian.local_energy_seperate(net.apply, cell)(dp, x32.cuda())[0].cpu().numpy()
for b in range(x32.shape[0]):
    ref = complex(ofl.stages(p64, x32[b].double(), klist, cell, net_kw)['ke'])
    sc = max(1.0, abs(ref))
    with onet.working_dtype(torch.float32):
        e_fl = abs(complex(ofl.stages(p32, x32[b], klist, cell, net_kw)['ke']) - ref) / sc
    line = f'{name} walker {b}: |E_kin| = {abs(ref):9.3f}   oracle-f32 rel. error {e_fl:.2e}'
    if ke_hip is not None:
        line += f'   HIP-f32 rel. error {abs(complex(ke_hip[b]) - ref) / sc:.2e}'
    print(line, flush=True)
