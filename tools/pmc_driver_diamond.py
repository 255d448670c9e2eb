#!/usr/bin/env python3
"""Workload for counter passes on the 96-electron float32 chain: one local-energy evaluation of 96 diamond walkers."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from deepsolid_amd import hamiltonian, network, systems

cell, klist = systems.build('diamond')
net = network.make_solid_fermi_net(klist=klist, simulation_cell=cell, method_name='eval_logdet', dtype=torch.float32, **systems.DETNET_DEFAULTS)
params = net.init(0)
x = torch.as_tensor(systems.synthetic_walkers(cell, 96), dtype=torch.float32, device='cuda')
for _ in range(2):
    ke, ew = hamiltonian.local_energy_seperate(net.apply, cell)(params, x)
torch.cuda.synchronize()
