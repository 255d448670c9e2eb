#!/usr/bin/env python3
"""GPU box: phase timeline of one wave of k_layer_group (csrc/ds_layer.h) from in-kernel shader-clock stamps.
Run with DS_LG_DBG=32 (the stamps drain the memory counters, so the timed wave itself runs a little slower)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

os.environ.setdefault('DS_LG_DBG', '32')
from deepsolid_amd import hamiltonian, network, systems

cell, klist = systems.build('bcc_li')
net = network.make_solid_fermi_net(klist=klist, simulation_cell=cell, method_name='eval_logdet', **systems.DETNET_DEFAULTS)
params = net.init(0)
x = torch.as_tensor(systems.synthetic_walkers(cell, 1024), device='cuda')
el = hamiltonian.local_energy_seperate(net.apply, cell)
el(params, x)
torch.cuda.synchronize()
sysd = net.apply.system
sysd.profile(True, only='single_hidden')
el(params, x)
torch.cuda.synchronize()
n = 600
buf = (C.c_uint64 * n)()
sysd.lib.ds_debug_timeline(sysd.handle, buf, n)
t = np.array(buf[:], dtype=np.int64)
t = t[t > 0]
print('stamps:', len(t), ' (the last hidden-layer launch overwrites the earlier ones)')
d = np.diff(t)
per = 7                                           # stamps per pass: start, ready, products done, 4 electrons
names = ['start->ready', 'products', 'e0', 'e1', 'e2', 'e3', 'next pass']
npass = len(t) // per
for p in range(min(npass, 16)):
    seg = d[p * per:(p + 1) * per]
    print(f'pass {p:2d}: ' + '  '.join(f'{nm}={int(v):6d}' for nm, v in zip(names, seg)))
cyc, ticks, ghz = sysd.profile_clock()
print('clock probe:', cyc, ticks, ghz)
