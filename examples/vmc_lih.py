#!/usr/bin/env python3
"""End-to-end example on one MI355X: variational Monte Carlo of the 4-electron LiH cell with the reference's default
network -- a few Adam iterations (train.py / process.py Adam branch), then an energy evaluation of the result
(process.py `optimizer='none'`).  Synthetic start (random parameters, uniform walkers): the numbers are not physics,
the point is the call sequence.  usage: python examples/vmc_lih.py [iterations] [batch]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from deepsolid_amd import inference, network, systems

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
cell, klist = systems.build('lih')
kw = dict(systems.DETNET_DEFAULTS)
logdet = network.make_solid_fermi_net(klist=klist, simulation_cell=cell, method_name='eval_logdet', **kw)
slogdet = network.make_solid_fermi_net(klist=klist, simulation_cell=cell, method_name='eval_slogdet', **kw)
params = logdet.init(0)
data = torch.as_tensor(systems.synthetic_walkers(cell, batch), device='cuda')
data, params, state, width, rows = inference.run_training(slogdet, logdet, params, data, cell, iterations=iters, burn_in=20,
                                                          learning_rate=2e-3, move_width=0.1)
print('training:   E[0] = %.4f  ->  E[%d] = %.4f Ha   (pmove %.2f)' % (rows[0]['energy'], iters - 1, rows[-1]['energy'], rows[-1]['pmove']))
data, width, rows = inference.run_inference(slogdet, logdet, params, data, cell, iterations=10, burn_in=10, move_width=width)
print('evaluation: E = %.4f +- %.4f Ha over 10 x %d walkers' % (sum(r['energy'] for r in rows) / len(rows),
                                                                (sum(r['variance'] for r in rows) / len(rows) / (10 * batch)) ** 0.5, batch))
