"""Times the parameter-gradient pass (ds_logpsi_vjp) next to the value pass and the local energy.
usage: python tools/grad_bench.py [system] [batch] [vjp]"""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from deepsolid_amd import systems                                   # noqa: E402
from deepsolid_amd.device import DeviceSystem                       # noqa: E402
from deepsolid_amd.network import init_solid_fermi_net_params       # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'bcc_li'
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    cell, klist = systems.build(name)
    net_kw = dict(systems.DETNET_DEFAULTS)
    sysd = DeviceSystem.for_network(cell, klist, net_kw, torch.float64)
    from oracle.testing import make_test_params
    params = make_test_params(0, cell.original_cell.atom_coords(), cell.nelec, net_kw)
    dp = {k: [{kk: torch.as_tensor(np.asarray(vv), dtype=torch.float64, device='cuda') for kk, vv in d.items()} for d in v]
          for k, v in params.items()}
    x = torch.as_tensor(systems.synthetic_walkers(cell, B), device='cuda')
    cot = torch.randn(B, 2, dtype=torch.float64, device='cuda') / B

    def timed(f, n=5):
        f(); torch.cuda.synchronize()
        t = time.time()
        for _ in range(n):
            f()
        torch.cuda.synchronize()
        return (time.time() - t) / n * 1e3
    if len(sys.argv) > 3 and sys.argv[3] == 'vjp':        # profiling: only the gradient pass
        print(f'{name} B={B}: logpsi_vjp {timed(lambda: sysd.logpsi_vjp(dp, x, cot)):.2f} ms')
        return
    t_val = timed(lambda: sysd.logpsi(dp, x))
    t_vjp = timed(lambda: sysd.logpsi_vjp(dp, x, cot))
    t_el = timed(lambda: sysd.local_energy(dp, x), 2)
    print(f'{name} B={B}: logpsi {t_val:.2f} ms   logpsi_vjp {t_vjp:.2f} ms   local_energy {t_el:.2f} ms')


if __name__ == '__main__':
    main()
