"""Round 5: the int8-split hidden layer against the float64 kernel on a FULL bench batch (4096 synthetic walkers, bcc-Li):
distribution of |E_L(int8 path) - E_L(float64 path)| per walker.  Two systems in one process (DS_I8 is read at creation)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepsolid_amd import network, systems       # noqa: E402
from deepsolid_amd.device import DeviceSystem    # noqa: E402
from deepsolid_amd.ewaldsum import EwaldTables   # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cell, klist = systems.build('bcc_li')
net_kw = dict(systems.DETNET_DEFAULTS)
net = network.make_solid_fermi_net(klist=klist, simulation_cell=cell, method_name='eval_logdet', **net_kw)
params = net.init(0)
x = torch.as_tensor(systems.synthetic_walkers(cell, B, seed=1234), device='cuda')
out = {}
for flag in (None, '1'):
    if flag:
        os.environ.pop('DS_I8', None)
    else:
        os.environ['DS_I8'] = '1'
    sysd = DeviceSystem(cell, klist, net_kw, EwaldTables(cell), torch.float64)
    out[flag] = torch.view_as_complex(sysd.local_energy(params, x)[0]).cpu().numpy()
    assert (sysd.int8_layers() > 0) == (flag is None)
d = np.abs(out[None] - out['1'])
ok = np.isfinite(d)
q = np.quantile(d[ok], [0.5, 0.9, 0.99, 0.999, 1.0])
print(json.dumps({'walkers': B, 'finite': int(ok.sum()), 'abs_diff_ha_quantiles_50_90_99_999_max': q.tolist(),
                  'abs_E_kin_median': float(np.median(np.abs(out['1'][ok]))), 'abs_E_kin_max': float(np.abs(out['1'][ok]).max()),
                  'rel_diff_max': float((d[ok] / np.maximum(1.0, np.abs(out['1'][ok]))).max())}))
