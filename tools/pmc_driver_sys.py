#!/usr/bin/env python3
"""Workload for PMC passes on another system: three local-energy evaluations.  PMC_SYSTEM (default diamond), PMC_DTYPE (f32 | f64),
PMC_WALKERS (default 1024)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from deepsolid_amd import hamiltonian, network, systems

name = os.environ.get('PMC_SYSTEM', 'diamond')
dtype = torch.float32 if os.environ.get('PMC_DTYPE', 'f32') == 'f32' else torch.float64
B = int(os.environ.get('PMC_WALKERS', 1024))
cell, klist = systems.build(name)
net = network.make_solid_fermi_net(klist=klist, simulation_cell=cell, method_name='eval_logdet', dtype=dtype, **systems.DETNET_DEFAULTS)
params = net.init(0)
x = torch.as_tensor(systems.synthetic_walkers(cell, B), dtype=dtype, device='cuda')
for _ in range(3):
    ke, ew = hamiltonian.local_energy_seperate(net.apply, cell)(params, x)
    torch.cuda.synchronize()
print('system', name, 'walkers', B)
