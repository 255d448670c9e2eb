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
