#!/usr/bin/env python3
"""Workload for profiling the value chain: 5 log-psi evaluations of 4096 bcc-Li walkers."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from deepsolid_amd import network, systems

cell, klist = systems.build('bcc_li')
net = network.make_solid_fermi_net(klist=klist, simulation_cell=cell, method_name='eval_slogdet', **systems.DETNET_DEFAULTS)
params = net.init(0)
x = torch.as_tensor(systems.synthetic_walkers(cell, 4096), device='cuda')
for _ in range(5):
    lp = net.apply(params, x)
torch.cuda.synchronize()
