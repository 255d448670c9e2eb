"""Walker-chunk sweep: ds_local_energy processes `ws_bytes / per_walker` walkers per pass of the kernel chain, so the
chunk is set by the workspace the CALLER hands over (ds_workspace_bytes asks for 1024 walkers).  This driver times the
bcc-Li bench step (B = 4096, f64) with workspaces sized for 512 / 1024 / 2048 walkers and checks that the energies are
bit-identical to the default.          usage (GPU box): python tools/chunk_sweep.py
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from deepsolid_amd import network, systems  # noqa: E402


def main():
    B = 4096
    cell, klist = systems.build('bcc_li')
    net = network.make_solid_fermi_net(klist=klist, simulation_cell=cell, method_name='eval_logdet', **systems.DETNET_DEFAULTS)
    params = net.init(0)
    dev = torch.device('cuda:0')
    x = torch.as_tensor(systems.synthetic_walkers(cell, B), device=dev)
    sysd = net.apply.system
    per_1024 = int(sysd.lib.ds_workspace_bytes(sysd.handle, 1024))
    base = None
    for chunk in [int(c) for c in sys.argv[1:]] or (1024, 512, 2048, 1024):
        sysd._ws = None
        torch.cuda.empty_cache()
        nbytes = per_1024 * chunk // 1024 + 4096
        if chunk > 1024:
            sysd._ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)       # workspace() keeps a larger buffer
            out = lambda: sysd.local_energy(params, x)
        else:
            out = lambda: sysd.local_energy(params, x, ws_bytes=nbytes)
        ke, ew = out()[:2]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            out()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        e = torch.cat([ke.reshape(-1), ew.reshape(-1)]).cpu()
        if base is None:
            base = e
        print(json.dumps({'chunk': chunk, 'ms_per_step': ms, 'evals_per_s': B / ms * 1e3, 'ws_gb': nbytes / 2 ** 30,
                          'bit_identical_to_1024': bool(torch.equal(e, base))}), flush=True)


if __name__ == '__main__':
    main()
