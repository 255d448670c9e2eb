#!/usr/bin/env python3
"""Workload for tools/pmc_kernel.sh on another BASELINE system: two local-energy evaluations.
    python tools/pmc_driver_system.py SYSTEM BATCH [f64|f32]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from deepsolid_amd import hamiltonian, network, systems

name, B = sys.argv[1], int(sys.argv[2])
dtype = torch.float32 if len(sys.argv) > 3 and sys.argv[3] == 'f32' else torch.float64
cell, klist = systems.build(name)
net = network.make_solid_fermi_net(klist=klist, simulation_cell=cell, method_name='eval_logdet', dtype=dtype, **systems.DETNET_DEFAULTS)
params = net.init(0)
x = torch.as_tensor(systems.synthetic_walkers(cell, B), dtype=dtype, device='cuda')
for _ in range(2):
    ke, ew = hamiltonian.local_energy_seperate(net.apply, cell)(params, x)
    torch.cuda.synchronize()
