#!/usr/bin/env python3
"""GPU box: per-phase shader-clock totals of one workgroup of the trace kernel (k_det_trace_mfma_split for diamond float32, the default;
k_det_trace_mfma for `python tools/trace_timeline.py 1024 bcc_li f64` or `512 graphene f64`; DS_DBG=32):
fragment setup, products, barrier after them, pair sums, barrier after them, per-tile trace reduction, whole kernel -- per wave."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
os.environ['DS_DBG'] = '32'
from deepsolid_amd import hamiltonian, network, systems
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
name = sys.argv[2] if len(sys.argv) > 2 else 'diamond'
dtype = torch.float64 if (len(sys.argv) > 3 and sys.argv[3] == 'f64') else torch.float32
cell, klist = systems.build(name)
net = network.make_solid_fermi_net(klist=klist, simulation_cell=cell, method_name='eval_logdet', dtype=dtype, **systems.DETNET_DEFAULTS)
params = net.init(0)
x = torch.as_tensor(systems.synthetic_walkers(cell, B), dtype=dtype, device='cuda')
el = hamiltonian.local_energy_seperate(net.apply, cell)
el(params, x); torch.cuda.synchronize()
sysd = net.apply.system
names = ['setup', 'products', 'barrier1', 'pairs', 'barrier2', 'trace-red', 'total']
for rep in range(2):
    el(params, x); torch.cuda.synchronize()
    buf = (C.c_uint64 * 32)()
    sysd.lib.ds_debug_timeline(sysd.handle, buf, 32)
    t = np.array(buf[:32], dtype=np.int64).reshape(4, 8)
    for w in range(4):          # (the first four waves)
        print(f'wave {w}: ' + '  '.join(f'{n} {t[w, i]:>8d}' for i, n in enumerate(names)))
