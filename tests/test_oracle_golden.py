"""Pins the CPU oracle (and the PySCF-free system builder) against vectors
produced by executing the reference's own code (tools/make_golden.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import distance as odist
from oracle import ewaldsum as oewald
from oracle import hamiltonian as oham
from oracle import network as onet
from oracle import qmc as oqmc
from oracle.testing import CASES

from common import GOLDEN, load_case, oracle_net, tt

ALL = list(CASES)
SMALL = [c for c in ALL if c not in ('graphene', 'diamond')]


@pytest.mark.parametrize('name', ALL)
def test_system_builder_matches_reference_supercell(name):
    """deepsolid_amd.supercell vs reference supercell.py run on the same primitive cell."""
    fx, cell, klist, _, _ = load_case(name)
    prim = cell.original_cell
    np.testing.assert_allclose(prim.a, fx['prim_a'], atol=1e-14)
    np.testing.assert_allclose(cell.a, fx['sim_a'], atol=1e-13)
    np.testing.assert_allclose(cell.atom_coords(), fx['sim_atoms'], atol=1e-12)
    np.testing.assert_array_equal(cell.atom_charges(), fx['sim_charges'])
    assert tuple(cell.nelec) == tuple(fx['nelec'])
    for mine, ref in ((prim.AV, fx['prim_AV']), (prim.BV, fx['prim_BV']),
                      (cell.AV, fx['sim_AV']), (cell.BV, fx['sim_BV'])):
        np.testing.assert_allclose(mine, ref, atol=1e-13)
    np.testing.assert_allclose(klist[0], fx['klist_up'], atol=1e-13)
    np.testing.assert_allclose(klist[1], fx['klist_dn'], atol=1e-13)


@pytest.mark.parametrize('name', ALL)
def test_forward_matches_reference(name):
    fx, cell, klist, net_kw, params = load_case(name)
    p = onet.params_to_torch(params)
    net = oracle_net(cell, klist, net_kw, 'eval_phase_and_slogdet')
    mats = oracle_net(cell, klist, net_kw, 'eval_mats')
    atoms = tt(cell.original_cell.atom_coords())
    for b in range(fx['x'].shape[0]):
        x = tt(fx['x'][b])
        ph, ls = net.apply(p, x)
        assert abs(float(ls) - fx['logabs'][b]) < 1e-10
        assert abs(complex(ph) - fx['phase'][b]) < 1e-9
        for s, m in enumerate(mats.apply(p, x)):
            ref = fx[f'orbitals_{s}'][b]
            np.testing.assert_allclose(m.numpy(), ref, rtol=1e-10, atol=1e-12 * np.abs(ref).max())
        ft = onet.construct_periodic_input_features(x, atoms, cell, net_kw['distance_type'])
        for got, key in zip(ft, ('feat_ae', 'feat_ee', 'feat_r_ae', 'feat_r_ee')):
            np.testing.assert_allclose(got.numpy(), fx[key][b], atol=1e-12)


@pytest.mark.parametrize('name', ['h2', 'lih', 'lih_2x1x1', 'bcc_li', 'graphene', 'diamond'])
def test_ewald_matches_reference(name):
    fx, cell, _, _, _ = load_case(name)
    ew = oewald.EwaldSum(cell)
    assert ew.gpoints_np.shape[0] == int(fx['ewald_ng'])
    assert abs(ew.alpha - float(fx['ewald_alpha'])) < 1e-13
    assert {'diagonal': 0, 'orthogonal': 1, 'general': 2}[ew.dist.mode] == int(fx['dist_mode'])
    assert abs(ew.gweight_np.sum() - float(fx['ewald_gweight_sum'])) < 1e-12
    assert abs(ew.ion_ion - float(fx['ewald_ion_ion'])) < 1e-10
    assert abs(ew.ii_const - float(fx['ewald_ii_const'])) < 1e-10
    if 'ewald_gpoints' in fx:
        np.testing.assert_allclose(ew.gpoints_np, fx['ewald_gpoints'], atol=1e-12)
        np.testing.assert_allclose(ew.gweight_np, fx['ewald_gweight'], rtol=1e-12)
    for b in range(fx['x'].shape[0]):
        got = [float(v) for v in ew.energy(tt(fx['x'][b]))]
        np.testing.assert_allclose(got, fx['ewald'][b], atol=2e-10)


@pytest.mark.parametrize('name', ['h2', 'lih', 'bcc_li', 'graphene'])
def test_batched_wrap_matches_reference(name):
    fx, cell, _, _, _ = load_case(name)
    wx, wrap = odist.enforce_pbc(cell.a, tt(fx['x']))
    np.testing.assert_allclose(wx.numpy(), fx['pbc_x'], atol=1e-11)
    np.testing.assert_array_equal(wrap.numpy(), fx['pbc_wrap'])


@pytest.mark.parametrize('name', ['h2', 'lih', 'lih_twist', 'bcc_li'])
def test_metropolis_matches_reference(name):
    """One mh_update and a 3-step mcmc_step, replaying the noise the reference consumed."""
    fx, cell, klist, net_kw, params = load_case(name)
    p = onet.params_to_torch(params)
    net = oracle_net(cell, klist, net_kw, 'eval_slogdet')
    f = lambda pp, xs: torch.stack([net.apply(pp, x) for x in xs])
    xn, lpn, nacc = oqmc.mh_update(p, f, tt(fx['mh_x1']), tt(fx['mh_lp1']), 0.0, cell.a,
                                   stddev=float(fx['mh_width']), normal=tt(fx['mh_normal']),
                                   uniform=tt(fx['mh_uniform']))
    np.testing.assert_allclose(xn.numpy(), fx['mh_x_new'], atol=1e-11)
    np.testing.assert_allclose(lpn.numpy(), fx['mh_lp_new'], atol=1e-9)
    assert float(nacc) == float(fx['mh_num_accepts'])
    step = oqmc.make_mcmc_step(f, fx['mcmc_x0'].shape[0], cell.a, steps=int(fx['mcmc_steps']))
    xo, pmove = step(p, tt(fx['mcmc_x0']), (tt(fx['mcmc_normals']), tt(fx['mcmc_uniforms'])),
                     float(fx['mcmc_width']))
    np.testing.assert_allclose(xo.numpy(), fx['mcmc_x_out'], atol=1e-11)
    assert abs(float(pmove) - float(fx['mcmc_pmove'])) < 1e-15
    if 'mha_x_new' in fx:      # asymmetric proposal (qmc.py:197-215), atoms = primitive-cell nuclei
        xa, lpa, nacca = oqmc.mh_update(p, f, tt(fx['mh_x1']), tt(fx['mh_lp1']), 0.0, cell.a, stddev=float(fx['mh_width']),
                                        normal=tt(fx['mha_normal']), uniform=tt(fx['mha_uniform']),
                                        atoms=cell.original_cell.atom_coords())
        np.testing.assert_allclose(xa.numpy(), fx['mha_x_new'], atol=1e-11)
        np.testing.assert_allclose(lpa.numpy(), fx['mha_lp_new'], atol=1e-9)
        assert float(nacca) == float(fx['mha_num_accepts'])


@pytest.mark.parametrize('name', ['h2', 'lih', 'lih_twist', 'bcc_li'])
def test_kinetic_vs_finite_differences_of_reference_forward(name):
    """`for`-mode restatement (hamiltonian.py:45-70) vs 4th-order FD of the
    reference-executed eval_logdet."""
    fx, cell, klist, net_kw, params = load_case(name)
    p = onet.params_to_torch(params)
    net = oracle_net(cell, klist, net_kw, 'eval_logdet')
    ke = oham.local_kinetic_energy_real_imag(net.apply)
    mode = 'for' if sum(cell.nelec) <= 4 else 'hessian'
    if mode == 'hessian':
        ke = oham.local_kinetic_energy_real_imag_hessian(net.apply)
    for b in range(len(fx['ke_fd'])):
        got = complex(sum(ke(p, tt(fx['x'][b]))))
        assert abs(got - fx['ke_fd'][b]) < float(fx['ke_fd_tol']) * max(1.0, abs(got))


@pytest.mark.parametrize('name', ['h2', 'lih', 'lih_twist'])
def test_laplacian_modes_agree(name):
    fx, cell, klist, net_kw, params = load_case(name)
    p = onet.params_to_torch(params)
    net = oracle_net(cell, klist, net_kw, 'eval_logdet')
    x = tt(fx['x'][1])
    n3 = x.shape[0]
    part = 3 if n3 % 3 == 0 else 2
    vals = {m: oham.local_energy_seperate(net.apply, cell, mode=m, partition_number=part)(p, x)
            for m in ('for', 'hessian', 'dim_batch', 'partition')}
    ref = complex(vals['for'][0])
    for m, (ke, ew) in vals.items():
        assert abs(complex(ke) - ref) < 1e-10 * max(1.0, abs(ref)), m
    with pytest.raises(ValueError):
        oham.local_energy_seperate(net.apply, cell, mode='nope')


@pytest.mark.parametrize('name', ['lih', 'lih_twist', 'bcc_li', 'lih_fulldet', 'lih_tri'])
def test_oracle_parameter_gradient_vs_reference_finite_differences(name):
    """torch autograd over the restated forward (oracle/train.py:logpsi_vjp) against the directional
    derivative of the REFERENCE-executed forward along oracle.testing.make_test_direction."""
    from oracle import train as otrain
    from oracle.testing import make_test_direction
    fx, cell, klist, net_kw, params = load_case(name)
    net = oracle_net(cell, klist, net_kw, 'eval_logdet')
    v = make_test_direction(int(fx['gradfd_seed']), params)

    def dot(t, d):
        if isinstance(t, dict):
            return sum(dot(t[k], d[k]) for k in t)
        if isinstance(t, (list, tuple)):
            return sum(dot(a, c) for a, c in zip(t, d))
        return float((t * torch.as_tensor(d)).sum())
    nb = 1 if name == 'bcc_li' else len(fx['gradfd_dlogabs'])
    for b in range(nb):
        for cot, key in ((1.0 + 0j, 'gradfd_dlogabs'), (1j, 'gradfd_darg')):
            g = otrain.logpsi_vjp(net.apply, params, tt(fx['x'][b:b + 1]), torch.tensor([cot]))
            assert abs(dot(g, v) - float(fx[key][b])) < 2e-7 * max(1.0, abs(float(fx[key][b])))


# ----------------------------------------------------------------------------------------------
# The reference's OWN hamiltonian.py / train.py, executed over its own network.py under the
# torch-backed `jax` stand-in (tools/jax_torch_standin.py, tools/make_golden.py): ke_ref / grad_ref.
KE_CASES = [c for c in ALL if CASES[c].get('ke_walkers')]
GRAD_CASES = [c for c in ALL if CASES[c].get('grad_walkers')]
CPU_GRAD_CASES = ('h2', 'lih', 'lih_fulldet', 'lih_fullenv', 'bcc_li', 'li_polarized')
KE_TOL_HA = 1e-9          # the kinetic-energy pin of the oracle against the reference (Hartree)


@pytest.mark.parametrize('name', KE_CASES)
def test_kinetic_matches_reference_hamiltonian(name):
    """oracle kinetic energy vs hamiltonian.py:45-70 run verbatim (|dE| <= 1e-9 Ha, relative above 1 Ha)."""
    fx, cell, klist, net_kw, params = load_case(name)
    p = onet.params_to_torch(params)
    net = oracle_net(cell, klist, net_kw, 'eval_logdet')
    n = sum(cell.nelec)
    ke_ref = fx['ke_ref']
    assert len(ke_ref) == CASES[name]['ke_walkers']
    default_net = not (net_kw.get('full_det', False) or net_kw.get('bias_orbitals', False)) and net_kw.get('envelope_type', 'isotropic') == 'isotropic'
    if n <= 8:        # autodiff restatement: the reference's default schedule on one walker, `hessian` mode on the others
        ke = oham.local_kinetic_energy_real_imag(net.apply)
        got = [complex(sum(ke(p, tt(fx['x'][0]))))]
        ke = oham.local_kinetic_energy_real_imag_hessian(net.apply)
        got += [complex(sum(ke(p, tt(fx['x'][b])))) for b in range(1, len(ke_ref) if not default_net else 2)]
    elif n <= 24:
        ke = oham.local_kinetic_energy_real_imag_hessian(net.apply)
        got = [complex(sum(ke(p, tt(fx['x'][b])))) for b in range(1 if default_net else 2)]
    else:
        got = []
    # every walker also through the forward-Laplacian restatement (the algorithm of the HIP chain)
    from oracle import forward_laplacian as ofl
    assert default_net or len(got) >= min(2, len(ke_ref))     # (the forward-Laplacian oracle covers the default options)
    for b in range(len(ke_ref) if default_net else 0):
        v = complex(ofl.stages(p, tt(fx['x'][b]), klist, cell, net_kw)['ke'])
        assert abs(v - ke_ref[b]) < KE_TOL_HA * max(1.0, abs(ke_ref[b])), (name, b, v, ke_ref[b])
    for b, v in enumerate(got):
        assert abs(v - ke_ref[b]) < KE_TOL_HA * max(1.0, abs(ke_ref[b])), (name, b, v, ke_ref[b])
    if 'ew_ref' in fx:                                                        # the Ewald term of the same call
        np.testing.assert_allclose(fx['ew_ref'], fx['ewald'][:len(fx['ew_ref'])].sum(-1), atol=1e-12)


@pytest.mark.parametrize('name', [c for c in KE_CASES if len(CASES[c].get('ke_modes', ())) > 1])
def test_reference_laplacian_modes_agree(name):
    """hamiltonian.py `hessian` :104-124, `dim_batch` :73-101, `partition` :127-159 return the `for` number."""
    fx = load_case(name)[0]
    for mode in CASES[name]['ke_modes'][1:]:
        v = fx['ke_ref_' + mode][0]
        assert abs(v - fx['ke_ref'][0]) < 1e-10 * max(1.0, abs(v)), mode


@pytest.mark.parametrize('name', ['lih', 'bcc_li', 'graphene'])
def test_forward_laplacian_equals_autodiff_hessian(name):
    """The forward-Laplacian restatement (the HIP chain's algorithm) against the autodiff `hessian`-mode
    restatement of hamiltonian.py:104-124 — two different algorithms, so a shared mistake cannot hide."""
    from oracle import forward_laplacian as ofl
    fx, cell, klist, net_kw, params = load_case(name)
    p = onet.params_to_torch(params)
    net = oracle_net(cell, klist, net_kw, 'eval_logdet')
    ke = oham.local_kinetic_energy_real_imag_hessian(net.apply)
    for b in range(1 if name == 'graphene' else 2):
        a = complex(sum(ke(p, tt(fx['x'][b]))))
        f = complex(ofl.stages(p, tt(fx['x'][b]), klist, cell, net_kw)['ke'])
        assert abs(a - f) < 1e-10 * max(1.0, abs(a)), (a, f)


def _leaves(tree, path=()):
    if isinstance(tree, dict):
        for k in sorted(tree):
            yield from _leaves(tree[k], path + (k,))
    elif isinstance(tree, (list, tuple)):
        for i, v in enumerate(tree):
            yield from _leaves(v, path + (i,))
    else:
        yield '/'.join(str(q) for q in path), tree


def check_gradient_against_reference(fx, params, grad, sfx='', rtol=1e-8):
    """grad (tree of torch/numpy leaves) vs the grad_ref_* records: per-leaf norm, per-leaf dot with the
    seeded direction, and the small leaves element by element."""
    from oracle.testing import make_test_direction
    v = make_test_direction(int(fx['grad_ref_seed']), params)
    names = [str(s) for s in fx['grad_ref_names']]
    gl, vl = dict(_leaves(grad)), dict(_leaves(v))
    assert sorted(gl) == sorted(names)
    scale = float(np.max(fx['grad_ref_norm' + sfx]))
    for i, nme in enumerate(names):
        g = np.asarray(gl[nme].detach().cpu() if hasattr(gl[nme], 'detach') else gl[nme], dtype=np.float64)
        assert abs(np.linalg.norm(g) - fx['grad_ref_norm' + sfx][i]) < rtol * scale, nme
        assert abs(float((g * vl[nme]).sum()) - fx['grad_ref_dot' + sfx][i]) < rtol * scale * max(1.0, np.linalg.norm(vl[nme])), nme
        key = 'grad_ref_leaf' + sfx + ':' + nme
        if key in fx:
            np.testing.assert_allclose(g, fx[key], atol=rtol * scale, err_msg=nme)


@pytest.mark.parametrize('name', GRAD_CASES)
def test_energy_gradient_matches_reference_train(name):
    """oracle value_and_grad vs the reference's train.make_loss (:37-142) differentiated as process.py:204 does."""
    from oracle import train as otrain
    fx, cell, klist, net_kw, params = load_case(name)
    if name not in CPU_GRAD_CASES:
        pytest.skip('this fixture is checked against the HIP path in tests/test_gpu_grad.py; the CPU autodiff oracle is '
                    'pinned on a subset to keep the CPU suite at a few minutes')
    net = oracle_net(cell, klist, net_kw, 'eval_logdet')
    nb = int(fx['grad_ref_walkers'])
    for clip_type in CASES[name].get('grad_clip_types', ('real',)):
        sfx = '' if clip_type == 'real' else '_' + clip_type
        loss_fn = otrain.make_loss(net.apply, cell, mode='hessian', clip_local_energy=5.0, clip_type=clip_type)
        (loss, aux), g = loss_fn.value_and_grad(params, tt(fx['x'][:nb]))
        assert abs(float(loss) - float(fx['grad_ref_loss' + sfx])) < 1e-9
        assert abs(float(aux.variance) - float(fx['grad_ref_variance' + sfx])) < 1e-8 * max(1.0, float(aux.variance))
        assert abs(float(aux.imaginary) - float(fx['grad_ref_imag' + sfx])) < 1e-9
        check_gradient_against_reference(fx, params, g, sfx)


@pytest.mark.parametrize('name', ['h2', 'lih', 'lih_twist', 'bcc_li'])
def test_other_samplers_match_reference(name):
    """The reference's own mh_one_electron_update (qmc.py:227-287, N moves through make_mcmc_step) and importance_update
    (qmc.py:83-124, with the value and gradient of its own network) replayed on the noise they consumed."""
    from torch.func import grad as tgrad
    fx, cell, klist, net_kw, params = load_case(name)
    if 'mh1_x_out' not in fx:
        pytest.skip('fixture predates the sampler records')
    p = onet.params_to_torch(params)
    net = oracle_net(cell, klist, net_kw, 'eval_slogdet')
    f = lambda pp, xs: torch.stack([net.apply(pp, x) for x in xs])
    x = tt(fx['mcmc_x0'])
    lp = 2.0 * f(p, x)
    nacc = 0.0
    n = x.shape[1] // 3
    for i in range(n):                                     # nsteps = N * steps, electron i % N (qmc.py:355-356)
        x, lp, nacc = oqmc.mh_one_electron_update(p, f, x, lp, nacc, cell.a, stddev=float(fx['mh1_width']), i=i,
                                                  normal=tt(fx['mh1_normals'][i]), uniform=tt(fx['mh1_uniforms'][i]))
    np.testing.assert_allclose(x.numpy(), fx['mh1_x_out'], atol=1e-11)
    assert abs(float(nacc) / (n * x.shape[0]) - float(fx['mh1_pmove'])) < 1e-15
    fg = lambda pp, xs: (f(pp, xs), torch.stack([tgrad(lambda y: net.apply(pp, y))(xx) for xx in xs]))
    x0 = tt(fx['mcmc_x0'])
    np.testing.assert_allclose((2.0 * f(p, x0)).numpy(), fx['imp_lp1'], atol=1e-9)
    xi, lpi, na = oqmc.importance_update(p, fg, x0, tt(fx['imp_lp1']), 0.0, cell.a, stddev=float(fx['imp_width']),
                                         normal=tt(fx['imp_normal']), uniform=tt(fx['imp_uniform']))
    np.testing.assert_allclose(xi.numpy(), fx['imp_x_new'], atol=1e-10)
    np.testing.assert_allclose(lpi.numpy(), fx['imp_lp_new'], atol=1e-8)
    assert float(na) == float(fx['imp_num_accepts'])


@pytest.mark.parametrize('name', ['lih', 'bcc_li'])
def test_float32_reference_fixture_is_consistent(name):
    """tests/golden/f32_reference.npz (tools/make_f32_reference.py: the reference's own hamiltonian.py in float32 and, at the same
    float32-rounded walkers, in float64): the float64 leg must agree with the oracle at the rounded walker to 1e-9 (the same bar
    as every ke_ref fixture), the rounded walkers must be the fixture's, and the float32 leg must be a float32-sized distance away
    (between 1e-8 and 1e-2 relative: neither a float64 run in disguise nor garbage)."""
    from oracle import forward_laplacian as ofl
    from oracle import network as onet
    fx, cell, klist, net_kw, params = load_case(name)
    fxr = np.load(os.path.join(GOLDEN, 'f32_reference.npz'))
    nb = len(fxr[name + '_ke_f32'])
    np.testing.assert_array_equal(fxr[name + '_x32'], fx['x'][:nb].astype(np.float32))
    assert fxr[name + '_ke_f32'].dtype == np.complex64
    p = onet.params_to_torch(params)
    for b in range(nb):
        r64 = complex(fxr[name + '_ke_f64_at_x32'][b])
        o = complex(ofl.stages(p, torch.as_tensor(fxr[name + '_x32'][b].astype(np.float64)), klist, cell, net_kw)['ke'])
        assert abs(o - r64) < 1e-9 * max(1.0, abs(r64)), (b, o, r64)
        rel = abs(complex(fxr[name + '_ke_f32'][b]) - r64) / max(1.0, abs(r64))
        assert 1e-8 < rel < 1e-2, (b, rel)
