"""Development probe: parameter gradient of the HIP reverse sweep vs torch autograd over the oracle,
leaf by leaf.  usage: python tools/grad_check.py [case ...] [--batch B]"""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))

from common import load_case, oracle_net           # noqa: E402
from deepsolid_amd import systems                   # noqa: E402
from deepsolid_amd.device import DeviceSystem       # noqa: E402
from oracle import train as otrain                  # noqa: E402


def leaves(tree, prefix=''):
    if isinstance(tree, dict):
        for k in sorted(tree):
            yield from leaves(tree[k], prefix + '/' + str(k))
    elif isinstance(tree, (list, tuple)):
        for i, v in enumerate(tree):
            yield from leaves(v, prefix + '/' + str(i))
    else:
        yield prefix, tree


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    B = 7
    if '--batch' in sys.argv:
        B = int(sys.argv[sys.argv.index('--batch') + 1])
    for name in args or ['lih']:
        fx, cell, klist, net_kw, params = load_case(name)
        sysd = DeviceSystem.for_network(cell, klist, net_kw, torch.float64)
        x = systems.synthetic_walkers(cell, B, seed=77)
        rng = np.random.default_rng(5)
        cot = rng.normal(size=(B, 2))
        dp = {k: [{kk: torch.as_tensor(vv, dtype=torch.float64, device='cuda') for kk, vv in d.items()} for d in v]
              for k, v in params.items()}
        xd = torch.as_tensor(x, device='cuda')
        t0 = time.time()
        flat, la, ph = sysd.logpsi_vjp(dp, xd, torch.as_tensor(cot, device='cuda'))
        torch.cuda.synchronize()
        t1 = time.time()
        got = sysd.unpack_grad(flat, dp)
        net = oracle_net(cell, klist, net_kw, 'eval_logdet')
        cc = torch.complex(torch.as_tensor(cot[:, 0]), torch.as_tensor(cot[:, 1]))
        ref = otrain.logpsi_vjp(net.apply, params, torch.as_tensor(x), cc)
        from oracle.network import params_to_torch
        pt = params_to_torch(params)
        la_ref = torch.stack([net.apply(pt, torch.as_tensor(xx)).real for xx in x])
        print(f'== {name}  B={B}  hip {t1 - t0:.3f}s   logabs err {float((la.cpu() - la_ref).abs().max()):.2e}')
        worst = 0.0
        for (pa, g), (_, r) in zip(leaves(got), leaves(ref)):
            g = g.cpu().numpy(); r = r.detach().numpy()
            err = np.abs(g - r).max() / max(1e-30, np.abs(r).max())
            worst = max(worst, err)
            print(f'  {pa:24s} shape {str(r.shape):14s} |ref| {np.abs(r).max():.3e}  rel err {err:.2e}')
        print('  worst', worst)


if __name__ == '__main__':
    main()
