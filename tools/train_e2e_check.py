#!/usr/bin/env python3
"""End-to-end check of the training loop on the benchmark cell (4096 bcc-Li walkers: fused pair stream, int8 value layers, low-rank
layer; with DS_I8=1 also the opt-in int8 energy layer): a few Adam iterations through inference.run_training, with the defaults,
with the int8 energy layer switched on, and with the float64 / layer-by-layer sides forced; the energies of the two runs must agree to round-off of the
sampler's decisions (same seeds; a decision can flip only on a ~1e-12 tie).    python tools/train_e2e_check.py [iterations]"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

WORKER = r'''
import sys, json, torch
sys.path.insert(0, %r)
from deepsolid_amd import network, systems, inference
cell, klist = systems.build('bcc_li')
kw = dict(systems.DETNET_DEFAULTS)
slog = network.make_solid_fermi_net(klist=klist, simulation_cell=cell, method_name='eval_slogdet', **kw)
ldet = network.make_solid_fermi_net(klist=klist, simulation_cell=cell, method_name='eval_logdet', **kw)
params = slog.init(0)
x = torch.as_tensor(systems.synthetic_walkers(cell, 4096, seed=3), device='cuda')
data, params, opt, width, rows = inference.run_training(slog, ldet, params, x, cell, iterations=%d, key=7, burn_in=3, mcmc_steps=10,
                                                        learning_rate=1e-3)
print(json.dumps([{k: (float(v) if isinstance(v, (int, float)) else str(v)) for k, v in r.items()} for r in rows]))
'''


def run(env_extra, iters):
    env = dict(os.environ, **env_extra)
    out = subprocess.run([sys.executable, '-c', WORKER % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), iters)], env=env,
                         capture_output=True, text=True, timeout=1200)
    if out.returncode:
        print(out.stderr[-2000:])
        raise SystemExit(1)
    import json
    return json.loads(out.stdout.strip().splitlines()[-1])


if __name__ == '__main__':
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    a = run({}, iters)
    c = run({'DS_I8': '1'}, iters)
    b = run({'DS_NO_I8_VAL': '1', 'DS_NO_PAIR_FUSE': '1', 'DS_NO_LOWRANK': '1'}, iters)
    for ra, rc, rb in zip(a, c, b):
        print('default  ', ra)
        print('DS_I8=1  ', rc)
        print('reference', rb)
