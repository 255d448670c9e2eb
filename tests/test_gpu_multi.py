"""Multi-GPU tests: run only where the box has >= 2 (>= 8) MI355X -- RCCL with N > 1 ranks over xGMI.

One process per GPU under `torch.distributed.run` (the driver's launch line).  Walkers shard across the ranks with no
data-path collective (process.py:72-77,96); the only exchange is the packed all-reduce of the batch statistics
(train.py:78-80) and, when training, of the packed gradient (train.py:176-177).  These tests assert exactly that:
  * every rank reports in (`ranks_seen == N`) and ranks hold different walkers,
  * the reduced energy equals the single-rank energy of the concatenated batch,
  * the all-reduced packed gradient equals the mean of the per-rank gradients,
  * `bench.py --gpus N` prints the strong-scaling line (ONE global batch split, BASELINE configs 4 / 5) beside the weak one.
On a one-GPU box they are skipped; the same plumbing runs on CPU with gloo in tests/test_host_cpu.py."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NGPU = torch.cuda.device_count() if torch.cuda.is_available() else 0


def _torchrun(n, script_args, port, timeout=1800, extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    env.update(extra_env or {})
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port)] + script_args
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)


@pytest.mark.parametrize('n', [2, 8])
def test_bench_strong_and_weak_lines_over_rccl(n):
    if NGPU < n:
        pytest.skip(f'needs {n} GPUs, this box has {NGPU}')
    r = _torchrun(n, [os.path.join(ROOT, 'bench.py'), '--gpus', str(n), '--system', 'bcc_li', '--batch', '4096', '--scaling', 'strong',
                      '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-mcmc'], 29541 + n)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(line) == 1
    d = json.loads(line[0])
    assert d['n_gpus'] == n and d['ranks_seen'] == n and d['scaling'] == 'strong'
    assert d['config']['global_batch'] == 4096 and d['config']['batch_per_gpu'] == 4096 // n      # process.py:72-77
    assert d['value'] > 0 and d['other_scaling']['scaling'] == 'weak' and d['other_scaling']['batch_per_gpu'] == 4096
    assert d['other_scaling']['value'] > d['value'] * 0.9          # N full batches are not slower per walker than one split batch


WORKER = r'''
import os, sys, json, torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
import numpy as np
from deepsolid_amd import network, systems, train, constants
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
shared = os.environ.get('DS_TEST_SHARED_GPU') == '1'                 # both ranks on the ONE visible GPU, gloo between them
local = 0 if shared else int(os.environ['LOCAL_RANK'])
torch.cuda.set_device(local)
dev = torch.device('cuda', local)
if shared:
    dist.init_process_group('gloo')
else:
    dist.init_process_group('nccl', device_id=dev)
def all_gather(t):                                                # (gloo gathers host tensors; RCCL device tensors)
    src = t.cpu() if shared else t
    out = [torch.zeros_like(src) for _ in range(world)]
    dist.all_gather(out, src)
    return [o.to(dev) for o in out]
cell, klist = systems.build('lih')
net_kw = dict(systems.DETNET_DEFAULTS)
net = network.make_solid_fermi_net(klist=klist, simulation_cell=cell, method_name='eval_logdet', **net_kw)
params = net.init(0)                                              # identical on every rank
B = 96                                                            # per rank
x_all = torch.as_tensor(systems.synthetic_walkers(cell, B * world, seed=5), device=dev)
x = x_all[rank * B:(rank + 1) * B].contiguous()                   # contiguous walker slice per GPU (process.py:96)
loss_fn = train.make_loss(net.apply, None, cell, clip_local_energy=5.0)
loss, aux = loss_fn(params, x)                                    # statistics all-reduced over RCCL
(_, _), flat = loss_fn.value_and_grad_packed(params, x)
flat_mean = constants.pmean_if_pmap(flat.clone())                 # ONE all-reduce of the packed gradient
gathered = all_gather(flat)
seen = torch.zeros(world, dtype=torch.float64, device=dev); seen[rank] = 1.0
dist.all_reduce(seen)
first = all_gather(x[0, :3].contiguous())
if rank == 0:
    # single-rank reference on the concatenated batch (no process-group reduction: a fresh evaluation of every walker)
    from deepsolid_amd import hamiltonian
    ke, ew = hamiltonian.local_energy_seperate(net.apply, cell)(params, x_all)
    e_all = (ke + ew)
    out = dict(ranks_seen=int(seen.sum().item()), loss=float(loss), loss_ref=float(e_all.real.mean()),
               imag=float(aux.imaginary), imag_ref=float(e_all.imag.mean()),
               var=float(aux.variance),
               var_ref=float(np.mean([(e_all[r * B:(r + 1) * B].abs() ** 2).mean().item() - abs(e_all[r * B:(r + 1) * B].real.mean().item()) ** 2
                                      for r in range(world)])),          # train.py:79: per-device variance, then pmean
               grad_err=float((flat_mean - torch.stack(gathered).mean(0)).abs().max()),
               grad_norm=float(flat_mean.abs().max()),
               distinct=len({tuple(np.round(f.cpu().numpy(), 12)) for f in first}))
    print('RESULT ' + json.dumps(out))
dist.destroy_process_group()
'''


@pytest.mark.parametrize('n', [2, 8])
def test_sharded_energy_and_gradient_over_rccl(n, tmp_path):
    if NGPU < n:
        pytest.skip(f'needs {n} GPUs, this box has {NGPU}')
    script = tmp_path / 'worker.py'
    script.write_text(WORKER % dict(root=ROOT))
    r = _torchrun(n, [str(script)], 29561 + n)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith('RESULT ')][0][7:])
    _check_sharded(res, n)


def _check_sharded(res, n):
    assert res['ranks_seen'] == n and res['distinct'] == n                 # every rank ran, on its own walkers
    assert abs(res['loss'] - res['loss_ref']) < 1e-12 * max(1.0, abs(res['loss_ref']))
    assert abs(res['imag'] - res['imag_ref']) < 1e-12
    assert abs(res['var'] - res['var_ref']) < 1e-10 * max(1.0, abs(res['var_ref']))
    assert res['grad_err'] <= 1e-14 * max(1.0, res['grad_norm'])            # all-reduced packed gradient == mean of the per-rank ones


def test_sharded_energy_and_gradient_two_ranks_on_one_gpu(tmp_path):
    """The N > 1 code with REAL kernels on the hardware a one-GPU box has: two ranks share device 0 (gloo between them), hold
    different walker slices, and the all-reduced energy / imaginary part / per-device-mean variance / packed gradient must equal
    the single-rank evaluation of the concatenated batch (train.py:76-80, process.py:72-77,96, constants.py:33-45)."""
    if NGPU < 1:
        pytest.skip('needs a GPU')
    script = tmp_path / 'worker.py'
    script.write_text(WORKER % dict(root=ROOT))
    r = _torchrun(2, [str(script)], 29571, extra_env={'DS_TEST_SHARED_GPU': '1'})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith('RESULT ')][0][7:])
    _check_sharded(res, 2)


def test_bench_two_ranks_on_one_gpu():
    """bench.py's N = 2 path (strong split of one global batch, max-over-ranks timing, ranks_seen) with both ranks on the one GPU."""
    if NGPU < 1:
        pytest.skip('needs a GPU')
    r = _torchrun(2, [os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--system', 'lih', '--batch', '512', '--scaling', 'strong',
                      '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-mcmc', '--backend', 'gloo', '--shared-gpu'], 29573)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][0])
    assert d['n_gpus'] == 2 and d['ranks_seen'] == 2 and d['scaling'] == 'strong'
    assert d['config']['global_batch'] == 512 and d['config']['batch_per_gpu'] == 256
    assert d['value'] > 0 and d['other_scaling']['scaling'] == 'weak'
