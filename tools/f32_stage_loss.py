#!/usr/bin/env python3
"""GPU box: where a float32 run of the chain loses its digits.  Every stage buffer of the forward-Laplacian chain (ds_debug_stage) in
float32 against the same stage in float64 at the float32-rounded walkers: max |f32 - f64| / max |f64| per walker and stage.
    python tools/f32_stage_loss.py diamond"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
import torch
from common import load_case
from deepsolid_amd.device import DeviceSystem
from deepsolid_amd.ewaldsum import EwaldTables

name = sys.argv[1] if len(sys.argv) > 1 else 'diamond'
fx, cell, klist, net_kw, params = load_case(name)
nb = len(fx['ke_ref'])
x32 = torch.as_tensor(fx['x'][:nb], dtype=torch.float32)
tab = EwaldTables(cell)
out = {}
for dt in (torch.float64, torch.float32):
    sysd = DeviceSystem(cell, klist, net_kw, tab, dt)
    dp = {k: [{kk: torch.as_tensor(vv, dtype=dt, device='cuda') for kk, vv in d.items()} for d in v] for k, v in params.items()}
    x = x32.to(dt).cuda()
    st = {}
    for stage in ('g0', 'g1', 'g2', 'g3', 'h2_0', 'h2_1', 'h2_2', 'q', 'mout', 'minv', 'dets', 'tr'):
        try:
            v = sysd.debug_stage(dp, x, stage, 1 << 30).double().cpu().numpy()
        except RuntimeError as e:
            print('stage', stage, 'failed:', e)
            continue
        st[stage] = v.reshape(nb, -1)
    ke = torch.view_as_complex(sysd.local_energy(dp, x)[0].double()).cpu().numpy()
    st['ke'] = np.stack([ke.real, ke.imag], -1)
    out[dt] = st
print(f'{name}: max |f32 - f64| / max |f64| per walker')
for stage, ref in out[torch.float64].items():
    got = out[torch.float32][stage]
    errs = [np.abs(got[b] - ref[b]).max() / max(np.abs(ref[b]).max(), 1e-300) for b in range(nb)]
    print(f'  {stage:6s} ' + '  '.join(f'{e:.2e}' for e in errs))
# dets: [.., log|det|, arg, trY2 re, im] per determinant: the absolute errors of log|det| say more than a relative maximum
d64, d32 = out[torch.float64]['dets'], out[torch.float32]['dets']
print('  log|det| abs err ' + '  '.join(f'{np.abs(d32[b].reshape(-1, 4)[:, 0] - d64[b].reshape(-1, 4)[:, 0]).max():.2e}' for b in range(nb)))
print('  trY2 rel err     ' + '  '.join(f'{np.abs(d32[b].reshape(-1, 4)[:, 2:] - d64[b].reshape(-1, 4)[:, 2:]).max() / np.abs(d64[b].reshape(-1, 4)[:, 2:]).max():.2e}' for b in range(nb)))
# How much of the inverse's error is the float32 INVERSION and how much the float32 INPUT: invert the float32 chain's value matrices
# (slot 0 of slot tile 0 of MOUT, spin-up channel) in float64 on the host and compare with the float64 chain's inverse.
n = cell.nelec[0]; K = int(net_kw['determinants']); P = (3 * sum(cell.nelec) + 2 + 15) // 16 * 16
def mats(mout_row):
    m = mout_row[:K * n * n * 2 * P].reshape(K, P // 16, n, n, 2, 16)[:, 0, :, :, :, 0]
    return m[..., 0] + 1j * m[..., 1]
e_inv32, e_in = [], []
for b in range(nb):
    m32, m64 = mats(out[torch.float32]['mout'][b]), mats(out[torch.float64]['mout'][b])
    i64 = out[torch.float64]['minv'][b][:K * n * n * 2].reshape(K, n, n, 2); i64 = i64[..., 0] + 1j * i64[..., 1]
    i32 = out[torch.float32]['minv'][b][:K * n * n * 2].reshape(K, n, n, 2); i32 = i32[..., 0] + 1j * i32[..., 1]
    host = np.linalg.inv(m32)                      # float64 inverse of the float32 chain's matrices
    e_in.append(np.abs(host - i64).max() / np.abs(i64).max())
    e_inv32.append(np.abs(i32 - host).max() / np.abs(i64).max())
    if b == 0:
        print('  cond(M) of the 8 spin-up matrices, walker 0:', ' '.join(f'{np.linalg.cond(m64[k]):.1e}' for k in range(K)))
print('  inverse: error of the float32 INPUT (float64 inverse of the float32 matrices vs float64 chain)  ' + '  '.join(f'{e:.2e}' for e in e_in))
print('  inverse: error of the float32 INVERSION (device float32 inverse vs float64 inverse of the same matrices)  ' + '  '.join(f'{e:.2e}' for e in e_inv32))
