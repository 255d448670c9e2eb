"""GPU parity of the parameter gradient (reference train.py:91-142):
`ds_logpsi_vjp` through the C ABI against (1) torch autograd over the CPU oracle and (2) finite
differences of the REFERENCE-executed forward stored in tests/golden (gradfd_*).

Tolerances (float64): 1e-9 relative to the largest entry of each parameter leaf against the oracle;
against the reference finite differences the fixtures' own truncation error (2e-7 absolute)."""
import numpy as np
import pytest
import torch

from oracle import train as otrain
from oracle.testing import make_test_direction

from common import load_case, oracle_net
from deepsolid_amd import systems

pytestmark = pytest.mark.gpu


def dev_params(params, dtype=torch.float64):
    return {k: [{kk: torch.as_tensor(np.asarray(vv), dtype=dtype, device='cuda') for kk, vv in d.items()} for d in v]
            for k, v in params.items()}


def system_for(cell, klist, net_kw, dtype=torch.float64):
    from deepsolid_amd.device import DeviceSystem
    return DeviceSystem.for_network(cell, klist, net_kw, dtype)


def leaves(tree):
    if isinstance(tree, dict):
        for k in sorted(tree):
            yield from leaves(tree[k])
    elif isinstance(tree, (list, tuple)):
        for v in tree:
            yield from leaves(v)
    else:
        yield tree


def tree_dot(tree, direction):
    return sum(float((a.double().cpu() * torch.as_tensor(np.asarray(d))).sum()) for a, d in zip(leaves(tree), leaves(direction)))


def assert_tree_close(got, ref, rtol):
    for g, r in zip(leaves(got), leaves(ref)):
        g = g.double().cpu().numpy()
        r = r.detach().numpy() if isinstance(r, torch.Tensor) else np.asarray(r)
        assert g.shape == r.shape
        scale = max(np.abs(r).max(), 1e-12)
        assert np.abs(g - r).max() <= rtol * scale, (g.shape, np.abs(g - r).max(), scale)


@pytest.mark.parametrize('name,batch', [('h2', 5), ('lih', 7), ('lih_twist', 4), ('lih_2x1x1', 3), ('bcc_li', 5),
                                        ('lih_fulldet', 4), ('lih_tri', 4), ('lih_bias', 4), ('lih_diagenv', 4), ('lih_fullenv', 4), ('lih_lastlayer', 4),
                                        ('lih_fn_defaults', 3), ('bcc_li_fulldet', 2), ('graphene', 2)])
def test_vjp_vs_oracle_autograd(name, batch):
    fx, cell, klist, net_kw, params = load_case(name)
    sysd = system_for(cell, klist, net_kw)
    x = systems.synthetic_walkers(cell, batch, seed=77)
    cot = np.random.default_rng(5).normal(size=(batch, 2))
    dp = dev_params(params)
    flat, la, ph = sysd.logpsi_vjp(dp, torch.as_tensor(x, device='cuda'), torch.as_tensor(cot, device='cuda'))
    got = sysd.unpack_grad(flat, dp)
    net = oracle_net(cell, klist, net_kw, 'eval_logdet')
    cc = torch.complex(torch.as_tensor(cot[:, 0]), torch.as_tensor(cot[:, 1]))
    ref = otrain.logpsi_vjp(net.apply, params, torch.as_tensor(x), cc)
    assert_tree_close(got, ref, 1e-9)
    from oracle.network import params_to_torch
    pt = params_to_torch(params)
    lp = torch.stack([net.apply(pt, torch.as_tensor(xx)) for xx in x])
    assert float((la.cpu() - lp.real).abs().max()) < 1e-10
    assert float((torch.angle(ph.cpu() * torch.exp(-1j * lp.imag))).abs().max()) < 1e-10


@pytest.mark.parametrize('name', ['lih', 'lih_twist', 'bcc_li', 'lih_fulldet', 'lih_tri'])
def test_vjp_vs_reference_finite_differences(name):
    """d/dh log psi(theta + h v) at h = 0 from the reference's own forward (tools/make_golden.py)."""
    fx, cell, klist, net_kw, params = load_case(name)
    sysd = system_for(cell, klist, net_kw)
    v = make_test_direction(int(fx['gradfd_seed']), params)
    dp = dev_params(params)
    n = len(fx['gradfd_dlogabs'])
    x = torch.as_tensor(fx['x'][:n], device='cuda')
    for b in range(n):
        for cot, key in (((1.0, 0.0), 'gradfd_dlogabs'), ((0.0, 1.0), 'gradfd_darg')):
            c = torch.zeros(n, 2, dtype=torch.float64, device='cuda')
            c[b] = torch.tensor(cot, dtype=torch.float64)
            flat, _, _ = sysd.logpsi_vjp(dp, x, c)
            got = tree_dot(sysd.unpack_grad(flat, dp), v)
            assert abs(got - float(fx[key][b])) < 2e-7 * max(1.0, abs(float(fx[key][b]))), (b, key, got, fx[key][b])


def test_vjp_groups_chunks_and_linearity():
    """Batches that span several 80-walker groups, a ragged tail, several passes (small workspace)."""
    fx, cell, klist, net_kw, params = load_case('lih')
    sysd = system_for(cell, klist, net_kw)
    B = 203
    x = torch.as_tensor(systems.synthetic_walkers(cell, B, seed=3), device='cuda')
    cot = torch.as_tensor(np.random.default_rng(9).normal(size=(B, 2)), device='cuda')
    dp = dev_params(params)
    full, la, _ = sysd.logpsi_vjp(dp, x, cot)
    full = full.clone()
    # the same call again: bit-identical (fixed reduction order, no atomics)
    again, _, _ = sysd.logpsi_vjp(dp, x, cot)
    assert torch.equal(full, again)
    # one group per pass
    one_group = int(sysd.lib.ds_vjp_workspace_bytes(sysd.handle, 1))
    chunked, la2, _ = sysd.logpsi_vjp(dp, x, cot, max_bytes=one_group)
    scale = float(full.abs().max())
    assert float((chunked - full).abs().max()) < 1e-12 * scale
    assert torch.equal(la, la2)
    # linearity in the cotangent / additivity over walkers
    parts = torch.zeros_like(full)
    for lo, hi in ((0, 1), (1, 80), (80, 81), (81, 203)):
        g, _, _ = sysd.logpsi_vjp(dp, x[lo:hi], cot[lo:hi])
        parts += g
    assert float((parts - full).abs().max()) < 1e-12 * scale
    # against the oracle on a subset that crosses a group boundary
    sel = slice(70, 90)
    g, _, _ = sysd.logpsi_vjp(dp, x[sel], cot[sel])
    net = oracle_net(cell, klist, net_kw, 'eval_logdet')
    ref = otrain.logpsi_vjp(net.apply, params, x[sel].cpu(), torch.view_as_complex(cot[sel].cpu().contiguous()))
    assert_tree_close(sysd.unpack_grad(g, dp), ref, 1e-9)


def test_vjp_zero_cotangent_and_empty_batch():
    fx, cell, klist, net_kw, params = load_case('lih')
    sysd = system_for(cell, klist, net_kw)
    dp = dev_params(params)
    x = torch.as_tensor(fx['x'], device='cuda')
    g, _, _ = sysd.logpsi_vjp(dp, x, torch.zeros(x.shape[0], 2, dtype=torch.float64, device='cuda'))
    assert float(g.abs().max()) == 0.0
    g, la, ph = sysd.logpsi_vjp(dp, x[:0], torch.zeros(0, 2, dtype=torch.float64, device='cuda'))
    assert g.shape == (sysd.param_count,) and float(g.abs().max()) == 0.0 and la.shape == (0,)


def test_vjp_float32():
    fx, cell, klist, net_kw, params = load_case('lih')
    s64, s32 = system_for(cell, klist, net_kw), system_for(cell, klist, net_kw, torch.float32)
    x = systems.synthetic_walkers(cell, 90, seed=8)
    cot = np.random.default_rng(2).normal(size=(90, 2))
    p64, p32 = dev_params(params), dev_params(params, torch.float32)
    g64, _, _ = s64.logpsi_vjp(p64, torch.as_tensor(x, device='cuda'), torch.as_tensor(cot, device='cuda'))
    g32, _, _ = s32.logpsi_vjp(p32, torch.as_tensor(x, dtype=torch.float32, device='cuda'),
                               torch.as_tensor(cot, dtype=torch.float32, device='cuda'))
    t64, t32 = s64.unpack_grad(g64, p64), s32.unpack_grad(g32, p32)
    for a, b in zip(leaves(t64), leaves(t32)):
        assert float((a - b.double()).abs().max()) < 2e-3 * max(1.0, float(a.abs().max()))


@pytest.mark.parametrize('clip_type,clip', [('real', 5.0), ('real', 0.5), ('complex', 0.7), ('real', 0.0)])
def test_energy_gradient_vs_oracle(clip_type, clip):
    """total_energy.value_and_grad == jax.value_and_grad(total_energy) through the custom JVP (train.py:91-142)."""
    from deepsolid_amd import network as dnet, train as dtrain
    fx, cell, klist, net_kw, params = load_case('lih')
    net = dnet.make_solid_fermi_net(klist=klist, simulation_cell=cell, method_name='eval_logdet', **net_kw)
    loss_fn = dtrain.make_loss(net.apply, None, cell, clip_local_energy=clip, clip_type=clip_type)
    dp = dev_params(params)
    x = torch.as_tensor(fx['x'], device='cuda')
    (loss, aux), grads = loss_fn.value_and_grad(dp, x)
    onet_ = oracle_net(cell, klist, net_kw, 'eval_logdet')
    oloss = otrain.make_loss(onet_.apply, cell, mode='hessian', clip_local_energy=clip, clip_type=clip_type)
    (l_ref, aux_ref), g_ref = oloss.value_and_grad(params, torch.as_tensor(fx['x']))
    assert abs(float(loss) - float(l_ref)) < 1e-8
    assert float((aux.local_energy.cpu() - aux_ref.local_energy).abs().max()) < 1e-8
    assert_tree_close(grads, g_ref, 1e-7)


@pytest.mark.parametrize('hidden_dims,use_last', [(((40, 10), (56, 12), (56, 12)), False), (((100, 20), (100, 20)), True),
                                                  (((64, 20), (64, 24), (64, 24)), False)])
def test_hidden_widths_without_kernel_instances(hidden_dims, use_last):
    """Widths the kernels have no instance for (one-electron widths that are not multiples of 64, pair widths other than 16 / 32)
    run with zero-padded weights (the library's own plan: ds_device_widths, deepsolid_amd/device.py::device_plan): exact, because a
    padded feature is tanh(0) = 0 in every layer.  The residual connections follow the REFERENCE widths (network.py:525-528) through
    explicit flags: 40 -> 56 has none although both pad to 64, 56 -> 56 has one; 10 -> 12 has none (both 16), 12 -> 12 has one;
    20 -> 24 has none although both run the 32-wide pair kernels (refused until round 5).  Loss, local energies and the energy
    gradient -- mapped back to the reference's parameter shapes -- against the oracle."""
    from deepsolid_amd import network as dnet, train as dtrain
    from oracle.testing import make_test_params
    cell, klist = systems.build('lih')
    net_kw = dict(systems.DETNET_DEFAULTS, hidden_dims=hidden_dims, use_last_layer=use_last)
    params = make_test_params(17, cell.original_cell.atom_coords(), cell.nelec, net_kw)
    net = dnet.make_solid_fermi_net(klist=klist, simulation_cell=cell, method_name='eval_logdet', **net_kw)
    loss_fn = dtrain.make_loss(net.apply, None, cell, clip_local_energy=5.0, clip_type='real')
    dp = dev_params(params)
    xn = systems.synthetic_walkers(cell, 5, seed=21)
    (loss, aux), grads = loss_fn.value_and_grad(dp, torch.as_tensor(xn, device='cuda'))
    onet_ = oracle_net(cell, klist, net_kw, 'eval_logdet')
    oloss = otrain.make_loss(onet_.apply, cell, mode='hessian', clip_local_energy=5.0, clip_type='real')
    (l_ref, aux_ref), g_ref = oloss.value_and_grad(params, torch.as_tensor(xn))
    assert abs(float(loss) - float(l_ref)) < 1e-8
    assert float((aux.local_energy.cpu() - aux_ref.local_energy).abs().max()) < 1e-8
    assert_tree_close(grads, g_ref, 1e-7)


@pytest.mark.parametrize('system,hidden_dims', [('lih', ((8, 16), (64, 16), (64, 16))),        # two atoms, 'nu': 8 input features = hidden_single[0]
                                                ('bcc_li', ((4, 16), (4, 16), (64, 16))),      # one atom: 4; layers 0 AND 1 residual at width 4
                                                ('lih', ((64, 4), (64, 4), (64, 16))),         # pair stream: 4 pair features = hidden_double[0] (and [1])
                                                ('lih', ((8, 4), (64, 16)))])                  # both streams at once, two layers
def test_first_layer_as_wide_as_its_input_features(system, hidden_dims):
    """hidden_single[0] == nf x atoms (or hidden_double[0] == nf): the reference adds a residual at the FIRST layer (network.py:525-528).  Refused until round 6
    (its K = features + pair-mean rows is 64 k + 4 / + 8, and a width like 8 pads to 64 device features of which only 8 have a
    residual); now layer 0 runs without its residual and csrc/ds_kernels.h::k_layer_res_add (k_pair_res_add) finishes it.  Loss, local energies and
    the energy gradient -- energy chain, value chain and reverse sweep -- against the oracle."""
    from deepsolid_amd import network as dnet, train as dtrain
    from oracle.testing import make_test_params
    cell, klist = systems.build(system, **({'nelec': (3, 2)} if system == 'bcc_li' else {}))
    net_kw = dict(systems.DETNET_DEFAULTS, hidden_dims=hidden_dims)
    params = make_test_params(29, cell.original_cell.atom_coords(), cell.nelec, net_kw)
    net = dnet.make_solid_fermi_net(klist=klist, simulation_cell=cell, method_name='eval_logdet', **net_kw)
    assert any(net.apply.system.residuals[0])
    loss_fn = dtrain.make_loss(net.apply, None, cell, clip_local_energy=5.0, clip_type='real')
    dp = dev_params(params)
    xn = systems.synthetic_walkers(cell, 5, seed=23)
    (loss, aux), grads = loss_fn.value_and_grad(dp, torch.as_tensor(xn, device='cuda'))
    onet_ = oracle_net(cell, klist, net_kw, 'eval_logdet')
    oloss = otrain.make_loss(onet_.apply, cell, mode='hessian', clip_local_energy=5.0, clip_type='real')
    (l_ref, aux_ref), g_ref = oloss.value_and_grad(params, torch.as_tensor(xn))
    assert abs(float(loss) - float(l_ref)) < 1e-8
    assert float((aux.local_energy.cpu() - aux_ref.local_energy).abs().max()) < 1e-8
    assert_tree_close(grads, g_ref, 1e-7)


def test_forty_determinants():
    """More than 32 determinants (the reference has no bound; the library's is 64 since round 4): loss, local energies and energy
    gradient of a 40-determinant LiH wave function against the oracle."""
    from deepsolid_amd import network as dnet, train as dtrain
    from oracle.testing import make_test_params
    cell, klist = systems.build('lih')
    net_kw = dict(systems.DETNET_DEFAULTS, determinants=40)
    params = make_test_params(23, cell.original_cell.atom_coords(), cell.nelec, net_kw)
    net = dnet.make_solid_fermi_net(klist=klist, simulation_cell=cell, method_name='eval_logdet', **net_kw)
    loss_fn = dtrain.make_loss(net.apply, None, cell, clip_local_energy=5.0, clip_type='real')
    xn = systems.synthetic_walkers(cell, 4, seed=22)
    (loss, aux), grads = loss_fn.value_and_grad(dev_params(params), torch.as_tensor(xn, device='cuda'))
    oloss = otrain.make_loss(oracle_net(cell, klist, net_kw, 'eval_logdet').apply, cell, mode='hessian', clip_local_energy=5.0, clip_type='real')
    (l_ref, aux_ref), g_ref = oloss.value_and_grad(params, torch.as_tensor(xn))
    assert abs(float(loss) - float(l_ref)) < 1e-8
    assert float((aux.local_energy.cpu() - aux_ref.local_energy).abs().max()) < 1e-8
    assert_tree_close(grads, g_ref, 1e-7)


def test_training_step_runs_and_lowers_the_energy_estimate():
    """train.make_training_step (train.py:147-184) with Adam: a few steps on LiH from a fixed seed."""
    from deepsolid_amd import network as dnet, qmc, train as dtrain
    fx, cell, klist, net_kw, params = load_case('lih')
    net = dnet.make_solid_fermi_net(klist=klist, simulation_cell=cell, method_name='eval_logdet', **net_kw)
    slog = dnet.make_solid_fermi_net(klist=klist, simulation_cell=cell, method_name='eval_slogdet', **net_kw)
    B = 512
    loss_fn = dtrain.make_loss(net.apply, None, cell, clip_local_energy=5.0, clip_type='real')
    mcmc = qmc.make_mcmc_step(slog.apply, B, cell.a, steps=5)
    init, update = dtrain.adam(2e-3)
    dp = dev_params(params)
    state = init(dp)
    step = dtrain.make_training_step(mcmc, loss_fn, update)
    data = torch.as_tensor(systems.synthetic_walkers(cell, B, seed=1), device='cuda')
    gen = torch.Generator(device='cuda'); gen.manual_seed(0)
    before = [p.clone() for p in leaves(dp)]
    losses = []
    for t in range(12):
        data, dp, state, loss, aux, pmove, g = step(t, data, dp, state, gen, 0.1)
        assert np.isfinite(float(loss)) and 0.0 < float(pmove) <= 1.0
        losses.append(float(loss))
    assert any(not torch.equal(a, b) for a, b in zip(before, leaves(dp)))
    assert np.mean(losses[-3:]) < np.mean(losses[:3])


def test_training_step_rejects_a_step_with_a_nan_walker():
    """process.py:303-318 (cfg.debug.check_nan) through the real kernels: one walker with a NaN coordinate makes its local energy
    NaN, `ds_energy_stats` counts it (n_nonfinite, all-reduced), and the step is DISCARDED before Adam touches anything --
    parameters, optimiser moments / step count and the walkers come back bit-identical, loss = aux = None.  The next step on
    finite walkers is applied.  Without check_nan the same step poisons the parameters (the reason it is on in run_training)."""
    from deepsolid_amd import network as dnet, qmc, train as dtrain
    fx, cell, klist, net_kw, params = load_case('lih')
    net = dnet.make_solid_fermi_net(klist=klist, simulation_cell=cell, method_name='eval_logdet', **net_kw)
    slog = dnet.make_solid_fermi_net(klist=klist, simulation_cell=cell, method_name='eval_slogdet', **net_kw)
    B = 256
    loss_fn = dtrain.make_loss(net.apply, None, cell, clip_local_energy=5.0, clip_type='real')
    real_mcmc = qmc.make_mcmc_step(slog.apply, B, cell.a, steps=3)
    poison = {'on': True}

    def mcmc(p, d, key, w):
        d2, pm = real_mcmc(p, d, key, w)
        if poison['on']:
            d2 = d2.clone()
            d2[7, 4] = float('nan')                       # one coordinate of one walker
        return d2, pm

    init, update = dtrain.adam(2e-3)
    dp = dev_params(params)
    state = init(dp)
    data = torch.as_tensor(systems.synthetic_walkers(cell, B, seed=2), device='cuda')
    # the statistics kernel sees the bad walker
    bad = data.clone(); bad[7, 4] = float('nan')
    _, aux_bad = loss_fn(dp, bad)
    assert float(aux_bad.n_nonfinite) == 1.0
    step = dtrain.make_training_step(mcmc, loss_fn, update, check_nan=True)
    before = [p.clone() for p in leaves(dp)]
    d1, dp, state, loss, aux, pmove, g = step(0, data, dp, state, 11, 0.1)
    assert loss is None and aux is None and g is None and d1 is data
    assert state['count'] == 0 and all(float(m.abs().max()) == 0.0 for m in state['m'])
    assert all(torch.equal(a, b) for a, b in zip(before, leaves(dp)))
    poison['on'] = False
    d2, dp, state, loss, aux, pmove, g = step(1, data, dp, state, 12, 0.1)
    assert loss is not None and np.isfinite(float(loss)) and state['count'] == 1 and float(aux.n_nonfinite) == 0.0
    assert any(not torch.equal(a, b) for a, b in zip(before, leaves(dp)))
    # the reference's default (no check): the poisoned step is applied and the parameters are lost
    poison['on'] = True
    dq = dev_params(params)
    sq = init(dq)
    plain = dtrain.make_training_step(mcmc, loss_fn, update)
    _, dq, sq, loss, *_ = plain(0, data, dq, sq, 11, 0.1)
    assert not all(bool(torch.isfinite(p).all()) for p in leaves(dq))


def test_training_loop_writes_stats_and_reference_checkpoints(tmp_path):
    """inference.run_training (process.py:204-383, Adam branch): CSV rows in the reference schema, checkpoints that
    `checkpoint.restore` (reference layout) reads back, parameters updated in place."""
    from deepsolid_amd import checkpoint, inference, network as dnet
    fx, cell, klist, net_kw, params = load_case('lih')
    logdet = dnet.make_solid_fermi_net(klist=klist, simulation_cell=cell, method_name='eval_logdet', **net_kw)
    slog = dnet.make_solid_fermi_net(klist=klist, simulation_cell=cell, method_name='eval_slogdet', **net_kw)
    dp = dev_params(params)
    w0 = dp['single'][0]['w'].clone()
    data = torch.as_tensor(systems.synthetic_walkers(cell, 256, seed=4), device='cuda')
    data, dp2, state, width, rows = inference.run_training(slog, logdet, dp, data, cell, iterations=6, key=3, burn_in=5,
                                                           mcmc_steps=4, learning_rate=1e-3, save_path=str(tmp_path),
                                                           save_every=3)
    assert len(rows) == 6 and all(np.isfinite(r['energy']) for r in rows)
    assert not torch.equal(w0, dp['single'][0]['w'])
    lines = open(tmp_path / 'train_stats.csv').read().strip().splitlines()
    assert lines[0] == 'step,energy,variance,pmove,imaginary,kinetic,ewald' and len(lines) == 7
    last = checkpoint.find_last_checkpoint(str(tmp_path))
    assert last.endswith('qmcjax_ckpt_000005.npz')
    t, d, p, opt, w = checkpoint.restore(last, batch_size=256)
    x, p1, w1 = checkpoint.to_single_device(d, p, w)
    assert t == 6 and x.shape == (256, 12)
    np.testing.assert_array_equal(p1['single'][0]['w'], dp['single'][0]['w'].cpu().numpy())
    # resume (process.py:120-123,381): Adam moments and step count come back, no second burn-in, last step always saved
    opt1 = checkpoint.opt_state_to_single_device(opt)
    assert opt1['count'] == 6 and len(opt1['m']) == len(state['m'])
    np.testing.assert_array_equal(np.asarray(opt1['m'][0]), state['m'][0].cpu().numpy())
    data2, dp3, state2, _, rows2 = inference.run_training(slog, logdet, dp, torch.as_tensor(x, device='cuda'), cell, iterations=2, key=3,
                                                          burn_in=5, mcmc_steps=4, learning_rate=1e-3, save_path=str(tmp_path),
                                                          save_every=100, t_init=t, opt_state=opt1)
    assert [r['step'] for r in rows2] == [6, 7] and state2['count'] == 8
    assert checkpoint.find_last_checkpoint(str(tmp_path)).endswith('qmcjax_ckpt_000007.npz')


REF_GRAD_CASES = [c for c in __import__('oracle.testing', fromlist=['CASES']).CASES
                  if __import__('oracle.testing', fromlist=['CASES']).CASES[c].get('grad_walkers')]


@pytest.mark.parametrize('name', REF_GRAD_CASES)
def test_energy_gradient_vs_reference_train(name):
    """`total_energy.value_and_grad` (HIP local energy + clip + ds_logpsi_vjp) against the gradient the REFERENCE's own
    train.make_loss / jax.value_and_grad produced (tools/make_golden.py, torch-backed jax stand-in): loss, variance,
    every leaf's norm, its projection on a seeded direction, and the small leaves element-wise (1e-8 of the largest norm)."""
    from deepsolid_amd import network as dnet, train
    from oracle.testing import CASES
    from test_oracle_golden import check_gradient_against_reference
    fx, cell, klist, net_kw, params = load_case(name)
    net = dnet.make_solid_fermi_net(klist=klist, simulation_cell=cell, method_name='eval_logdet', **net_kw)
    dp = dev_params(params)
    nb = int(fx['grad_ref_walkers'])
    x = torch.as_tensor(fx['x'][:nb], device='cuda')
    for clip_type in CASES[name].get('grad_clip_types', ('real',)):
        sfx = '' if clip_type == 'real' else '_' + clip_type
        loss_fn = train.make_loss(net.apply, None, cell, clip_local_energy=5.0, clip_type=clip_type)
        (loss, aux), g = loss_fn.value_and_grad(dp, x)
        assert abs(float(loss) - float(fx['grad_ref_loss' + sfx])) < 1e-9 * max(1.0, abs(float(loss)))
        assert abs(float(aux.variance) - float(fx['grad_ref_variance' + sfx])) < 1e-8 * max(1.0, float(aux.variance))
        assert abs(float(aux.imaginary) - float(fx['grad_ref_imag' + sfx])) < 1e-9 * max(1.0, abs(float(loss)))
        assert float(aux.n_nonfinite) == 0.0
        check_gradient_against_reference(fx, params, g, sfx)
