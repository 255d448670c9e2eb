"""CPU tests of the host logic: Ewald table setup against the reference-executed vectors,
minimal-image dispatch, Madelung known answers, and the 2-rank reduction over gloo."""
import os
import subprocess
import sys

import numpy as np
import pytest

from deepsolid_amd import distance, systems
from deepsolid_amd.cell import Cell
from deepsolid_amd.ewaldsum import EwaldTables

from common import load_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('name', ['h2', 'lih', 'lih_2x1x1', 'bcc_li', 'graphene', 'diamond'])
def test_ewald_tables_match_reference(name):
    fx, cell, _, _, _ = load_case(name)
    t = EwaldTables(cell)
    assert t.gpoints.shape[0] == int(fx['ewald_ng'])
    assert abs(t.alpha - float(fx['ewald_alpha'])) < 1e-13
    assert t.dist_mode == int(fx['dist_mode'])
    assert abs(t.gweight.sum() - float(fx['ewald_gweight_sum'])) < 1e-12
    assert abs(np.linalg.norm(t.gpoints, axis=1).sum() - float(fx['ewald_gnorm_sum'])) < 1e-7
    assert abs(t.ion_ion - float(fx['ewald_ion_ion'])) < 1e-10
    assert abs(t.ii_const - float(fx['ewald_ii_const'])) < 1e-10
    if 'ewald_gpoints' in fx:
        np.testing.assert_allclose(t.gpoints, fx['ewald_gpoints'], atol=1e-12)     # same order as the reference
        np.testing.assert_allclose(t.gweight, fx['ewald_gweight'], rtol=1e-12)


def test_minimal_image_dispatch_reproduces_reference_quirk():
    """distance.py:49-53 has no abs(): bcc (negative dots) -> 'orthogonal', fcc/hex -> 'general'."""
    assert distance.minimal_image_mode(np.diag([4.0, 100, 100])) == distance.DIAGONAL
    assert distance.minimal_image_mode(systems.bcc_li().a) == distance.ORTHOGONAL
    assert distance.minimal_image_mode(systems.lih_rocksalt().a) == distance.GENERAL
    assert distance.minimal_image_mode(systems.graphene().a) == distance.GENERAL


@pytest.mark.parametrize('structure,madelung', [('nacl', 1.747564594633), ('cscl', 1.762674773071)])
def test_madelung_known_answers(structure, madelung):
    """ion_ion + ii_const of a +-1 point-charge lattice = -M / r_nn (textbook Madelung constants).
    Independent known answer for the Ewald machinery (the reference checks against PySCF instead,
    hamiltonian.py:170)."""
    a = 5.0

    class Ions(Cell):
        def atom_charges(self):
            return np.asarray([1.0, -1.0])
    if structure == 'nacl':
        cell = Ions((np.ones((3, 3)) - np.eye(3)) * a / 2, [('H', [0, 0, 0]), ('H', [a / 2, a / 2, a / 2])], nelec=(1, 1))
        rnn = a / 2
    else:
        cell = Ions(np.eye(3) * a, [('H', [0, 0, 0]), ('H', [a / 2, a / 2, a / 2])], nelec=(1, 1))
        rnn = a * np.sqrt(3) / 2
    t = EwaldTables(cell)
    assert abs((t.ion_ion + t.ii_const) - (-madelung / rnn)) < 1e-9


def test_two_rank_reduction_over_gloo(tmp_path):
    """constants.pmean_packed / pmean_if_pmap with world_size 2 (the N>1 path of bench.py and
    train.total_energy); gloo on CPU stands in for RCCL."""
    script = tmp_path / 'worker.py'
    script.write_text('''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from deepsolid_amd import constants
dist.init_process_group('gloo')
r = dist.get_rank()
a, b, c = constants.pmean_packed(torch.tensor(1.0 + r), torch.tensor(10.0 * r), torch.tensor(-2.0))
assert abs(float(a) - 1.5) < 1e-15 and abs(float(b) - 5.0) < 1e-15 and float(c) == -2.0
z = constants.pmean_if_pmap(torch.tensor(complex(r, 2 * r), dtype=torch.complex128))
assert abs(complex(z) - complex(0.5, 1.0)) < 1e-15
s = constants.psum_if_pmap(torch.tensor(float(r + 1)))
assert float(s) == 3.0
# walkers shard with no overlap: rank-dependent seeds as in bench.py
from deepsolid_amd import systems
cell, _ = systems.build('lih')
x = systems.synthetic_walkers(cell, 4, seed=1234 + r)
g = [torch.zeros(4, 12, dtype=torch.float64) for _ in range(2)]
dist.all_gather(g, torch.as_tensor(x))
assert not torch.equal(g[0], g[1])
dist.destroy_process_group()
print('rank', r, 'ok')
''' % ROOT)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1')
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
                          '--master-addr', '127.0.0.1', '--master-port', '29533', str(script)],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count('ok') == 2


def test_poscar_reader_and_init_electrons(tmp_path):
    """reference config/poscar/bcc_li.vasp content (data fixture) -> 2-atom conventional bcc cell."""
    from deepsolid_amd import init_guess, supercell
    poscar = tmp_path / 'bcc_li.vasp'
    poscar.write_text("Li2\n1.0\n 3.4268178940 0.0 0.0\n 0.0 3.4268178940 0.0\n 0.0 0.0 3.4268178940\n Li\n 2\nCartesian\n"
                      " 0.0 0.0 0.0\n 1.713408947 1.713408947 1.713408947\n")
    cell = init_guess.read_poscar(str(poscar))
    a0 = 3.4268178940 / 0.52917721067
    np.testing.assert_allclose(cell.a, np.eye(3) * a0, atol=1e-9)
    np.testing.assert_allclose(cell.atom_coords()[1], np.full(3, 1.713408947 / 0.52917721067), atol=1e-9)
    assert cell.nelec == (3, 3)
    sim = supercell.get_supercell(cell, np.eye(3))
    x = init_guess.init_electrons(7, sim, sim.a, sim.nelec, batch_size=16, init_width=0.8)
    assert x.shape == (16, 18)
    frac = x.reshape(16, 6, 3) @ np.linalg.inv(sim.a)
    assert frac.min() >= 0 and frac.max() < 1
    # spin-polarised request: one spin flipped somewhere
    x2 = init_guess.init_electrons(7, sim, sim.a, (4, 2), batch_size=4)
    assert x2.shape == (4, 18)


@pytest.mark.parametrize('clip_type,clip', [('real', 5.0), ('real', 0.3), ('complex', 0.5), ('complex', 0.0)])
def test_clip_difference_matches_oracle(clip_type, clip):
    """train.py:105-129 (host logic of the energy gradient; single process, pmean = identity)."""
    import torch
    from deepsolid_amd import train as dtrain
    from oracle import train as otrain
    g = torch.Generator().manual_seed(4)
    diff = torch.complex(torch.randn(64, generator=g, dtype=torch.float64), torch.randn(64, generator=g, dtype=torch.float64))
    diff[3] *= 40.0
    a, b = dtrain.clip_difference(diff, clip, clip_type), otrain.clip_difference(diff, clip, clip_type)
    assert float((a - b).abs().max()) < 1e-13
    if clip > 0 and clip < 1:
        assert float((a - diff).abs().max()) > 0.1       # clipping active
    with pytest.raises(ValueError, match='Unrecognized clip type'):
        dtrain.clip_difference(diff, 1.0, 'nope')


def test_adam_matches_torch_optim():
    import torch
    from deepsolid_amd import train as dtrain
    g = torch.Generator().manual_seed(1)
    params = {'a': [{'w': torch.randn(5, 3, generator=g, dtype=torch.float64)}], 'b': torch.randn(4, generator=g, dtype=torch.float64)}
    ref = [params['a'][0]['w'].clone().requires_grad_(True), params['b'].clone().requires_grad_(True)]
    opt = torch.optim.Adam(ref, lr=1e-2)
    init, update = dtrain.adam(1e-2)
    state = init(params)
    for t in range(5):
        grads = {'a': [{'w': torch.randn(5, 3, generator=g, dtype=torch.float64)}], 'b': torch.randn(4, generator=g, dtype=torch.float64)}
        ref[0].grad, ref[1].grad = grads['a'][0]['w'].clone(), grads['b'].clone()
        opt.step()
        state, params = update(t, grads, params, state)
    assert float((params['a'][0]['w'] - ref[0].detach()).abs().max()) < 1e-12
    assert float((params['b'] - ref[1].detach()).abs().max()) < 1e-12


def test_training_step_averages_packed_gradient_over_two_ranks(tmp_path):
    """train.make_training_step (train.py:147-184, search_direction pmean :176-177) with world_size 2 over
    gloo: the per-rank packed gradient is averaged in one all-reduce, then unpacked; both ranks take the same
    optimiser step.  The loss is a stub (no GPU here); the reduction / unpack / update plumbing is the product's."""
    script = tmp_path / 'worker.py'
    script.write_text('''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from deepsolid_amd import train, constants
dist.init_process_group('gloo')
r = dist.get_rank()
calls = []
_orig = dist.all_reduce
def counting(t, *a, **k):
    calls.append(t.numel())
    return _orig(t, *a, **k)
dist.all_reduce = counting

class StubSystem:
    def unpack_grad(self, flat, params):
        return {'w': flat[:6].reshape(2, 3), 'b': flat[6:8]}

class StubLoss:
    system = StubSystem()
    def value_and_grad_packed(self, params, data):
        flat = torch.arange(8, dtype=torch.float64) * (1.0 + r)          # rank-dependent gradient
        return (torch.tensor(-1.0 - r), None), flat

params = {'w': torch.zeros(2, 3, dtype=torch.float64), 'b': torch.zeros(2, dtype=torch.float64)}
init, update = train.adam(0.1)
state = init(params)
mcmc = lambda p, d, key, w: (d, torch.tensor(0.5))
step = train.make_training_step(mcmc, StubLoss(), update)
data, params, state, loss, aux, pmove, g = step(0, torch.zeros(4, 12), params, state, 0, 0.1)
assert calls == [8], calls                                                # ONE message for the whole gradient
assert torch.allclose(g['w'].reshape(-1), torch.arange(6, dtype=torch.float64) * 1.5)
assert torch.allclose(g['b'], torch.tensor([9.0, 10.5], dtype=torch.float64))
both = [torch.zeros(2, 3, dtype=torch.float64) for _ in range(2)]
dist.all_gather(both, params['w'])
assert torch.equal(both[0], both[1])                                      # replicas stay in lock-step
dist.destroy_process_group()
print('rank', r, 'ok')
''' % ROOT)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1')
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
                          '--master-addr', '127.0.0.1', '--master-port', '29537', str(script)],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count('ok') == 2


def test_training_step_rejects_non_finite_steps():
    """process.py:303-318: with check_nan a step whose gradient / loss / local energies are not finite leaves the walkers,
    the parameters and the optimiser state exactly as they were (Adam updates in place: the decision must come BEFORE
    the update) and returns loss = aux = None; the next finite step proceeds normally.  Stub loss, the product's step."""
    import torch
    from deepsolid_amd import train

    class StubSystem:
        def unpack_grad(self, flat, params):
            return {'w': flat[:6].reshape(2, 3), 'b': flat[6:8]}

    class StubLoss:
        system = StubSystem()
        mode = 'ok'

        def value_and_grad_packed(self, params, data):
            flat = torch.arange(1, 9, dtype=torch.float64)
            loss = torch.tensor(-1.0, dtype=torch.float64)
            bad = torch.tensor(0.0, dtype=torch.float64)
            if self.mode == 'nan_grad':
                flat = flat.clone(); flat[3] = float('nan')
            if self.mode == 'inf_loss':
                loss = torch.tensor(float('inf'), dtype=torch.float64)
            if self.mode == 'bad_walker':
                bad = torch.tensor(2.0, dtype=torch.float64)
            return (loss, train.AuxiliaryLossData(variance=torch.tensor(0.1), local_energy=None, imaginary=torch.tensor(0.0),
                                                  kinetic=None, ewald=None, n_nonfinite=bad)), flat

    loss_fn = StubLoss()
    params = {'w': torch.ones(2, 3, dtype=torch.float64), 'b': torch.ones(2, dtype=torch.float64)}
    init, update = train.adam(0.1)
    state = init(params)
    mcmc = lambda p, d, key, w: (d + 1.0, torch.tensor(0.5))
    step = train.make_training_step(mcmc, loss_fn, update, check_nan=True)
    x0 = torch.zeros(4, 12, dtype=torch.float64)
    data, params, state, loss, aux, pmove, g = step(0, x0, params, state, 0, 0.1)
    assert loss is not None and torch.equal(data, x0 + 1.0) and state['count'] == 1
    snap = {k: v.clone() for k, v in params.items()}
    m_snap = [m.clone() for m in state['m']]
    for mode in ('nan_grad', 'inf_loss', 'bad_walker'):
        loss_fn.mode = mode
        d2, params, state, loss, aux, pmove, g = step(1, data, params, state, 0, 0.1)
        assert loss is None and aux is None and g is None and float(pmove) == 0.5
        assert d2 is data and state['count'] == 1                                   # walkers and optimiser untouched
        assert all(torch.equal(params[k], snap[k]) for k in snap) and all(torch.equal(a, b) for a, b in zip(state['m'], m_snap))
    loss_fn.mode = 'ok'
    d3, params, state, loss, aux, pmove, g = step(2, data, params, state, 0, 0.1)
    assert loss is not None and state['count'] == 2 and not torch.equal(params['w'], snap['w'])
    # without check_nan (the reference's default) the step is applied whatever it contains
    loss_fn.mode = 'nan_grad'
    plain = train.make_training_step(mcmc, loss_fn, update)
    _, params, state, loss, *_ = plain(3, data, params, state, 0, 0.1)
    assert loss is not None and torch.isnan(params['w']).any()


def test_checkpoint_reads_reference_layout_and_round_trips(tmp_path):
    """A file written the way reference checkpoint.py:94-124 writes it (np.savez of t / data / pickled
    params tree with a leading device axis / opt_state / mcmc_width) is read back; ours has the same layout."""
    import torch
    from deepsolid_amd import checkpoint
    from oracle.testing import make_test_params
    cell, _ = systems.build('lih')
    params = make_test_params(3, cell.original_cell.atom_coords(), cell.nelec, dict(systems.DETNET_DEFAULTS))
    ndev, bdev = 2, 5
    rep = lambda a: np.broadcast_to(np.asarray(a), (ndev,) + np.asarray(a).shape).copy()
    ref_params = {k: [{kk: rep(vv) for kk, vv in d.items()} for d in v] for k, v in params.items()}
    data = np.random.default_rng(0).normal(size=(ndev, bdev, 12))
    fname = tmp_path / 'qmcjax_ckpt_000041.npz'
    with open(fname, 'wb') as f:                         # the reference's writer, verbatim call shape
        np.savez(f, t=41, data=data, params=ref_params, opt_state=None, mcmc_width=rep(0.03))
    (tmp_path / 'qmcjax_ckpt_000099.npz').write_bytes(b'')                 # empty / corrupt files are skipped
    (tmp_path / 'qmcjax_ckpt_000098.npz').write_bytes(b'corrupt')
    assert checkpoint.find_last_checkpoint(str(tmp_path)) == str(fname)
    with pytest.raises(ValueError, match='Incorrect number of devices'):
        checkpoint.restore(str(fname), n_devices=1)
    with pytest.raises(ValueError, match='Wrong batch size'):
        checkpoint.restore(str(fname), batch_size=11, n_devices=2)
    t, d, p, opt, width = checkpoint.restore(str(fname), batch_size=10, n_devices=2)
    assert t == 42 and opt is None and d.shape == (2, 5, 12)
    x, p1, w1 = checkpoint.to_single_device(d, p, width)
    assert x.shape == (10, 12) and w1 == 0.03
    np.testing.assert_array_equal(p1['single'][1]['w'], params['single'][1]['w'])
    np.testing.assert_array_equal(p1['envelope'][0]['sigma'], params['envelope'][0]['sigma'])
    # our writer -> same layout (device axis of 1), torch tensors accepted
    tp = {k: [{kk: torch.as_tensor(vv) for kk, vv in dd.items()} for dd in v] for k, v in params.items()}
    out = checkpoint.save(str(tmp_path), 7, torch.as_tensor(x), tp, None, torch.tensor(0.05))
    assert os.path.basename(out) == 'qmcjax_ckpt_000007.npz'
    t2, d2, p2, _, w2 = checkpoint.restore(out, batch_size=10)
    assert t2 == 8 and d2.shape == (1, 10, 12) and p2['orbital'][0]['w'].shape == (1,) + params['orbital'][0]['w'].shape
    _, p3, w3 = checkpoint.to_single_device(d2, p2, w2)
    np.testing.assert_array_equal(p3['double'][0]['b'], params['double'][0]['b'])
    assert abs(w3 - 0.05) < 1e-7
    # the training driver passes the width as a python float: it must still get the device axis -- the reference restores it
    # with jax.pmap(lambda x: x)(jnp.asarray(mcmc_width_ckpt)) (process.py:252, constants.py:29), which fails on a 0-d array
    for wv in (0.07, np.float64(0.07), torch.tensor(0.07, dtype=torch.float64), np.asarray([0.07])):
        out = checkpoint.save(str(tmp_path), 9, torch.as_tensor(x), tp, None, wv)
        with np.load(out, allow_pickle=True) as ck:
            assert ck['mcmc_width'].shape == (1,) and abs(float(ck['mcmc_width'][0]) - 0.07) < 1e-12


def test_adam_state_survives_a_checkpoint_round_trip(tmp_path):
    """process.py:381 saves opt_state; a run resumed from the file must continue bit-identically to an
    uninterrupted one (moments and the optimiser's own step count restored, no second burn-in)."""
    import torch
    from deepsolid_amd import checkpoint, train
    g = torch.Generator().manual_seed(5)
    mk = lambda: {'single': [{'w': torch.linspace(-1, 1, 12, dtype=torch.float64).reshape(3, 4).clone(),
                              'b': torch.zeros(4, dtype=torch.float64)}]}
    grads = [{'single': [{'w': torch.randn(3, 4, generator=g, dtype=torch.float64),
                          'b': torch.randn(4, generator=g, dtype=torch.float64)}]} for _ in range(6)]
    init, update = train.adam(1e-2)
    pa = mk(); sa = init(pa)
    for t in range(6):
        sa, pa = update(t, grads[t], pa, sa)
    pb = mk(); sb = init(pb)
    for t in range(3):
        sb, pb = update(t, grads[t], pb, sb)
    f = checkpoint.save(str(tmp_path), 2, torch.zeros(4, 6), pb, sb, 0.02)
    t0, d, p, opt, w = checkpoint.restore(f, batch_size=4)
    _, p1, _ = checkpoint.to_single_device(d, p, w)
    opt1 = checkpoint.opt_state_to_single_device(opt)
    assert t0 == 3 and opt1['count'] == 3
    pc = {'single': [{k: torch.as_tensor(v) for k, v in p1['single'][0].items()}]}
    for t in range(3, 6):
        opt1, pc = update(t, grads[t], pc, opt1)
    assert torch.equal(pc['single'][0]['w'], pa['single'][0]['w']) and torch.equal(pc['single'][0]['b'], pa['single'][0]['b'])


@pytest.mark.parametrize('scaling,per_gpu', [('weak', 4096), ('strong', 2048)])
def test_bench_launcher_spawns_one_rank_per_gpu(scaling, per_gpu):
    """`python bench.py --gpus 2` with no launcher environment re-executes itself under torch.distributed.run;
    both ranks report in through the all-reduce (gloo here, RCCL on the GPU node) and rank 0 prints n_gpus == 2."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT')}
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-run', '--scaling', scaling],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    line = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(line) == 1
    d = json.loads(line[0])
    assert d['n_gpus'] == 2 and d['ranks_seen'] == 2 and d['scaling'] == scaling
    assert d['config']['batch_per_gpu'] == per_gpu and d['config']['global_batch'] == (4096 if scaling == 'strong' else 8192)
    assert d['max_rank_seconds'] == pytest.approx(2e-3)            # the MAX over ranks, not rank 0's own time


def test_klist_from_a_saved_hf_run(tmp_path):
    """hf.SCF.init_scf (hf.py:84-104) from plain arrays: restricted (occupations 0/2) and unrestricted mo_occ, and the
    .npz loader; the synthetic make_klist is the same construction with uniform fillings."""
    from deepsolid_amd import supercell
    cell, klist = systems.build('bcc_li', twist=(0.3, 0.0, 0.15))
    kpts = supercell.get_supercell_kpts(cell) + np.dot(np.linalg.inv(cell.a), [0.3, 0.0, 0.15]) * 2 * np.pi
    nk, nmo = kpts.shape[0], 5
    occ_r = np.zeros((nk, nmo)); occ_r[:, 0] = 2.0; occ_r[:4, 1] = 2.0            # 8 + 4 doubly occupied orbitals: (12, 12)
    up, dn = supercell.klist_from_scf(kpts, occ_r)
    assert up.shape == (12, 3) and dn.shape == (12, 3)
    np.testing.assert_allclose(up[:2], np.tile(kpts[0], (2, 1)))
    np.testing.assert_allclose(up, dn)
    occ_u = np.zeros((2, nk, nmo)); occ_u[0, :, 0] = 1.0; occ_u[0, :4, 1] = 1.0; occ_u[1, :, 0] = 1.0
    up, dn = supercell.klist_from_scf(kpts, occ_u)
    assert up.shape == (12, 3) and dn.shape == (8, 3)
    np.savez(tmp_path / 'hf.npz', kpts=kpts, mo_occ=occ_r)
    for a, b in zip(supercell.load_hf_klist(tmp_path / 'hf.npz'), supercell.klist_from_scf(kpts, occ_r)):
        np.testing.assert_array_equal(a, b)
    # the twist is a fraction of the SIMULATION cell's reciprocal lattice (hf.py:61): one electron moved by a supercell
    # vector picks up exp(2 pi i twist)
    for j, t in enumerate((0.3, 0.0, 0.15)):
        ph = np.exp(1j * klist[0] @ cell.a[j])
        np.testing.assert_allclose(ph, np.exp(2j * np.pi * t) * np.ones(len(ph)), atol=1e-12)


def test_device_plan_follows_the_reference_residual_pattern():
    """deepsolid_amd/device.py::device_plan = the C ABI's ds_device_widths (host code, runs without a GPU): the widths the kernels
    run for the reference's hidden_dims (zero-padded weights, exact) and an EXPLICIT residual flag per layer and stream, set exactly
    where the reference adds a residual (network.py:525-528: in == out of the reference's widths) -- also at layer 0 of the
    one-electron stream and of the pair stream (round 6: any input width; the residual is added behind the layer,
    csrc/ds_kernels.h::k_layer_res_add / k_pair_res_add)."""
    import itertools
    from deepsolid_amd.device import device_plan, device_widths
    assert device_widths(((256, 32),) * 3, 4) == ((256, 32),) * 3
    assert device_widths(((100, 20),) * 3, 4) == ((128, 32),) * 3
    dev, res = device_plan(((100, 10), (120, 12), (120, 12)), 4)
    assert dev == ((128, 16), (128, 16), (128, 16)) and res == ((False, False), (False, False), (True, True))
    dev, res = device_plan(((40, 10), (56, 12), (56, 12)), 8)          # 40 -> 56 pads to 64 -> 64 and still has no residual
    assert dev == ((64, 16),) * 3 and res == ((False, False), (False, False), (True, True))
    dev, res = device_plan(((64, 20), (64, 24), (64, 24)), 4)          # (refused until round 5: two different 32-wide pair widths)
    assert dev == ((64, 32),) * 3 and res == ((False, False), (True, False), (True, True))
    dev, res = device_plan(((64, 16), (64, 16)), 64)                   # the input width itself is 64: a residual at layer 0
    assert dev == ((64, 16), (64, 16)) and res == ((True, False), (True, True))
    assert device_plan(((64, 16), (64, 16)), 62)[1][0] == (False, False)        # 62 input rows pad to 64, but 62 != 64: none
    dev, res = device_plan(((8, 16), (8, 16)), 8)                      # layer 0 as wide as its 8 input features: residual at layer 0, device width 64
    assert dev == ((64, 16), (64, 16)) and res == ((True, False), (True, True))
    dev, res = device_plan(((64, 4), (64, 4), (64, 4)), 4, 4)          # first pair layer as wide as the 4 pair features ('nu'): residual there too
    assert dev == ((64, 16),) * 3 and res == ((False, True), (True, True), (True, True))
    assert device_plan(((64, 7), (64, 7)), 7, 7, n_double=2)[1] == ((False, True), (True, True))      # 'tri': 7 pair features
    dev, res = device_plan(((256, 32), (256, 32), (256, 24)), 4, n_double=2)    # the last pair width is unused
    assert dev[:2] == ((256, 32),) * 2 and dev[2][0] == 256 and res[2] == (True, False)
    for bad, n_in, n_in2 in ((((64, 40), (64, 40)), 4, 4),              # pair widths beyond 32
                             (((2048, 16), (64, 16)), 4, 4),            # one-electron widths beyond 1024
                             (((64, 0), (64, 4)), 4, 4)):               # a pair width of zero
        with pytest.raises(ValueError):
            device_plan(bad, n_in, n_in2)
    singles, pairs = (40, 64, 100, 128, 130), (8, 16, 20, 32)
    for n_in, n_in2 in ((4, 4), (64, 4), (128, 7), (40, 7)):
        for dims in itertools.product(itertools.product(singles, pairs), repeat=3):
            dev, res = device_plan(dims, n_in, n_in2)                     # (no refusal inside this grid)
            for l, ((a, b), (pa, pb), (r1, r2)) in enumerate(zip(dims, dev, res)):
                assert pa % 64 == 0 and pa >= a and pa - a < 64 and pb in (16, 32) and pb >= b
                assert r1 == (a == (dims[l - 1][0] if l else n_in)) and r2 == (b == (dims[l - 1][1] if l else n_in2))
                if l and r1:
                    assert pa == dev[l - 1][0]
                if l and r2:
                    assert pb == dev[l - 1][1]
