#!/usr/bin/env python3
"""Workload for the PMC passes: one calibration copy of known size + three local-energy evaluations of 4096 bcc-Li walkers
(the bench line's batch = one launch of every kernel of the chain per layer and evaluation; the first evaluation also warms
the clocks up -- tools/pmc_summarize.py reports the MEDIAN duration of a kernel's launches)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from deepsolid_amd import _lib, hamiltonian, network, systems

lib = _lib.load()
n = 1 << 28                                            # 2 GiB read + 2 GiB written: far beyond the 256 MiB MALL
src = torch.ones(n, dtype=torch.float64, device='cuda')
dst = torch.empty_like(src)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
lib.ds_calib_copy(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), n, st)
torch.cuda.synchronize()
cell, klist = systems.build('bcc_li')
net = network.make_solid_fermi_net(klist=klist, simulation_cell=cell, method_name='eval_logdet', **systems.DETNET_DEFAULTS)
params = net.init(0)
B = int(os.environ.get('PMC_WALKERS', 4096))
x = torch.as_tensor(systems.synthetic_walkers(cell, B), device='cuda')
for _ in range(3):
    ke, ew = hamiltonian.local_energy_seperate(net.apply, cell)(params, x)
    torch.cuda.synchronize()
print('calib_bytes_each_way', 8 * n, 'walkers', B)
