#!/usr/bin/env python3
"""GPU box: phase stamps of one wave of the hidden-layer k_jet_gemm (DS_DBG=32): start, operands ready, products done, end."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
os.environ.setdefault('DS_DBG', '32')
from deepsolid_amd import hamiltonian, network, systems
cell, klist = systems.build('bcc_li')
net = network.make_solid_fermi_net(klist=klist, simulation_cell=cell, method_name='eval_logdet', **systems.DETNET_DEFAULTS)
params = net.init(0)
x = torch.as_tensor(systems.synthetic_walkers(cell, 1024), device='cuda')
el = hamiltonian.local_energy_seperate(net.apply, cell)
el(params, x); torch.cuda.synchronize()
sysd = net.apply.system
for rep in range(3):
    sysd.profile(True, only='single_hidden')
    el(params, x); torch.cuda.synchronize()
    buf = (C.c_uint64 * 8)()
    sysd.lib.ds_debug_timeline(sysd.handle, buf, 8)
    t = np.array(buf[:4], dtype=np.int64)
    print('start->operands', t[1] - t[0], ' products', t[2] - t[1], ' epilogue', t[3] - t[2], ' total', t[3] - t[0], ' clock', sysd.profile_clock()[2])
