"""GPU tests of the callers around the kernel chain: Metropolis update (qmc.py), total_energy
primal (train.py), wavefunction properties of the reference's test/test_network.py at the full
BASELINE batch, determinism and ragged / empty batches."""
import numpy as np
import pytest
import torch

from oracle import network as onet
from oracle import train as otrain

from common import load_case, oracle_net, tt
from test_gpu_parity import dev_params

pytestmark = pytest.mark.gpu


def nets(cell, klist, net_kw, *methods):
    from deepsolid_amd import network
    return [network.make_solid_fermi_net(klist=klist, simulation_cell=cell, method_name=m, **net_kw) for m in methods]


@pytest.mark.parametrize('name', ['h2', 'lih', 'lih_twist', 'bcc_li'])
def test_metropolis_vs_reference_vectors(name):
    """mh_update and a 3-step mcmc_step replaying the noise the reference consumed."""
    from deepsolid_amd import qmc
    fx, cell, klist, net_kw, params = load_case(name)
    dp = dev_params(params)
    (slog,) = nets(cell, klist, net_kw, 'eval_slogdet')
    cu = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64, device='cuda')
    nacc = torch.zeros(1, dtype=torch.float64, device='cuda')
    xn, _, lpn, nacc = qmc.mh_update(dp, slog.apply, cu(fx['mh_x1']), None, cu(fx['mh_lp1']), nacc, cell.a,
                                     stddev=float(fx['mh_width']), normal=cu(fx['mh_normal']), uniform=cu(fx['mh_uniform']))
    np.testing.assert_allclose(xn.cpu().numpy(), fx['mh_x_new'], atol=1e-10)
    np.testing.assert_allclose(lpn.cpu().numpy(), fx['mh_lp_new'], atol=1e-8)
    assert float(nacc.item()) == float(fx['mh_num_accepts'])
    step = qmc.make_mcmc_step(slog.apply, fx['mcmc_x0'].shape[0], cell.a, steps=int(fx['mcmc_steps']))
    xo, pmove = step(dp, cu(fx['mcmc_x0']), (cu(fx['mcmc_normals']), cu(fx['mcmc_uniforms'])), float(fx['mcmc_width']))
    np.testing.assert_allclose(xo.cpu().numpy(), fx['mcmc_x_out'], atol=1e-10)
    assert abs(float(pmove) - float(fx['mcmc_pmove'])) < 1e-15
    # generator-driven path: reproducible, stays in the cell, plausible acceptance
    x1, p1 = step(dp, cu(fx['mcmc_x0']), 7, 0.02)
    x2, p2 = step(dp, cu(fx['mcmc_x0']), 7, 0.02)
    assert torch.equal(x1, x2) and float(p1) == float(p2) and 0.0 <= float(p1) <= 1.0
    frac = (x1.reshape(x1.shape[0], -1, 3) @ torch.linalg.inv(cu(cell.a)))
    assert frac.min() > -1e-9 and frac.max() < 1 + 1e-9
    with pytest.raises(ValueError):
        qmc.make_mcmc_step(slog.apply, 4, cell.a, importance_sampling=lambda *a: 0, one_electron_moves=True)
    if 'mha_x_new' in fx:      # asymmetric proposal (qmc.py:197-215) replaying the reference's noise
        nacc = torch.zeros(1, dtype=torch.float64, device='cuda')
        xa, _, lpa, nacc = qmc.mh_update(dp, slog.apply, cu(fx['mh_x1']), None, cu(fx['mh_lp1']), nacc, cell.a,
                                         stddev=float(fx['mh_width']), atoms=cell.original_cell.atom_coords(),
                                         normal=cu(fx['mha_normal']), uniform=cu(fx['mha_uniform']))
        np.testing.assert_allclose(xa.cpu().numpy(), fx['mha_x_new'], atol=1e-10)
        np.testing.assert_allclose(lpa.cpu().numpy(), fx['mha_lp_new'], atol=1e-8)
        assert float(nacc.item()) == float(fx['mha_num_accepts'])
        # ... and through make_mcmc_step with a generator
        astep = qmc.make_mcmc_step(slog.apply, fx['mcmc_x0'].shape[0], cell.a, steps=3, atoms=cell.original_cell.atom_coords())
        xs, pm = astep(dp, cu(fx['mcmc_x0']), 11, 0.05)
        assert xs.shape == fx['mcmc_x0'].shape and 0.0 <= float(pm) <= 1.0
    with pytest.raises(NotImplementedError):          # qmc.py:275
        qmc.mh_one_electron_update(dp, slog.apply, cu(fx['mh_x1']), None, cu(fx['mh_lp1']), nacc, cell.a,
                                   atoms=cu(np.zeros((1, 3))))


@pytest.mark.parametrize('name', ['lih', 'bcc_li'])
def test_total_energy_vs_oracle(name):
    from deepsolid_amd import train
    fx, cell, klist, net_kw, params = load_case(name)
    dp = dev_params(params)
    (net,) = nets(cell, klist, net_kw, 'eval_logdet')
    x = torch.as_tensor(fx['x'], device='cuda')
    loss, aux = train.make_loss(net.apply, None, cell)(dp, x)
    o_loss, o_aux = otrain.make_loss(oracle_net(cell, klist, net_kw, 'eval_logdet').apply, cell,
                                     mode='hessian')(onet.params_to_torch(params), tt(fx['x']))
    assert abs(float(loss) - float(o_loss)) < 1e-8
    assert abs(float(aux.imaginary) - float(o_aux.imaginary)) < 1e-8
    assert abs(float(aux.variance) - float(o_aux.variance)) < 1e-7 * max(1.0, abs(float(o_aux.variance)))
    np.testing.assert_allclose(aux.local_energy.cpu().numpy(), o_aux.local_energy.numpy(), atol=1e-8)


def test_wavefunction_properties_at_full_batch():
    """The three checks of reference test/test_network.py:65-122 on bcc-Li at B = 4096, plus the
    Laplacian's invariances (E_L is unchanged by the same moves) and bitwise determinism."""
    from deepsolid_amd import hamiltonian, systems
    fx, cell, klist, net_kw, params = load_case('bcc_li_twist')
    dp = dev_params(params)
    ps, ld = nets(cell, klist, net_kw, 'eval_phase_and_slogdet', 'eval_logdet')
    B, N = 4096, sum(cell.nelec)
    x = torch.as_tensor(systems.synthetic_walkers(cell, B, seed=99), device='cuda')
    p1, s1 = ps.apply(dp, x)
    # periodic BC: all electrons translated by a primitive lattice vector (test_network.py:65-83)
    trans = torch.as_tensor(cell.original_cell.a[2], device='cuda')
    kp = sum(np.sum(k, axis=0) for k in klist)
    p2, s2 = ps.apply(dp, x + trans.repeat(N))
    assert (s1 - s2).abs().max() < 1e-9
    assert (p1 * np.exp(1j * np.dot(kp, cell.original_cell.a[2])) - p2).abs().max() < 1e-8
    # twisted BC: one electron translated by a supercell vector (test_network.py:86-106)
    shift = torch.zeros(3 * N, dtype=torch.float64, device='cuda')
    shift[:3] = torch.as_tensor(cell.a[1], device='cuda')
    p3, s3 = ps.apply(dp, x + shift)
    tw = np.exp(1j * np.dot(klist[0][0], cell.a[1]))
    assert (s1 - s3).abs().max() < 1e-9
    assert (p3 / p1 - tw).abs().max() < 1e-8
    # antisymmetry: swap electrons 0 and 1 (same spin) (test_network.py:109-122)
    xs = torch.cat([x[:, 3:6], x[:, :3], x[:, 6:]], dim=1)
    p4, s4 = ps.apply(dp, xs)
    assert (s1 - s4).abs().max() < 1e-9
    assert (p1 + p4).abs().max() < 1e-8
    el = hamiltonian.local_energy_seperate(ld.apply, cell)
    k1, e1 = el(dp, x)
    k2, e2 = el(dp, x)
    assert torch.equal(torch.view_as_real(k1), torch.view_as_real(k2)) and torch.equal(e1, e2)
    k4, e4 = el(dp, xs)
    scale = k1.abs().clamp(min=1.0)
    assert ((k1 - k4).abs() / scale).max() < 1e-8 and (e1 - e4).abs().max() < 1e-8
    k5, e5 = el(dp, x + trans.repeat(N))
    assert ((k1 - k5).abs() / scale).max() < 1e-8 and (e1 - e5).abs().max() < 1e-8
    assert torch.isfinite(torch.view_as_real(k1)).all()


def test_ragged_and_empty_batches():
    """B = 1, B not a multiple of the workspace chunk, and an empty batch."""
    from deepsolid_amd import hamiltonian
    fx, cell, klist, net_kw, params = load_case('lih')
    dp = dev_params(params)
    (net,) = nets(cell, klist, net_kw, 'eval_logdet')
    sysd = net.apply.system
    x = torch.as_tensor(np.tile(fx['x'], (3, 1))[:17], device='cuda')
    ke_all, ew_all, _, _ = sysd.local_energy(dp, x)
    per = int(sysd.lib.ds_workspace_bytes(sysd.handle, 1))
    ke_c, ew_c, _, _ = sysd.local_energy(dp, x, ws_bytes=5 * per - 256 * 4)       # chunks of 4 walkers: 4+4+4+4+1
    assert torch.equal(ke_all, ke_c) and torch.equal(ew_all, ew_c)
    k1, e1 = hamiltonian.local_energy_seperate(net.apply, cell)(dp, x[5])
    assert torch.equal(torch.view_as_real(k1), ke_all[5]) and torch.equal(e1, ew_all[5])
    ke0, ew0, _, _ = sysd.local_energy(dp, x[:0])
    assert ke0.shape == (0, 2) and ew0.shape == (0,)
    with pytest.raises(ValueError):
        sysd.local_energy(dp, x[:, :-3])
    with pytest.raises(RuntimeError):
        sysd.local_energy(dp, x.cpu())


@pytest.mark.parametrize('name', ['lih', 'bcc_li'])
def test_gradient_one_electron_and_importance_moves_vs_oracle(name):
    """ds_logpsi_grad vs autodiff, and the two 'untested' samplers of qmc.py (one-electron moves :227,
    importance sampling :83) against the oracle's restatement on identical noise."""
    from torch.func import grad as tgrad
    from deepsolid_amd import qmc
    from oracle import qmc as oqmc
    fx, cell, klist, net_kw, params = load_case(name)
    dp = dev_params(params)
    p_cpu = onet.params_to_torch(params)
    (slog,) = nets(cell, klist, net_kw, 'eval_slogdet')
    o_slog = oracle_net(cell, klist, net_kw, 'eval_slogdet')
    cu = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64, device='cuda')
    x1 = fx['pbc_x'][:3]
    B, n3 = x1.shape
    la, g = slog.apply.value_and_grad(dp, cu(x1))
    g_ref = torch.stack([tgrad(lambda y: o_slog.apply(p_cpu, y))(tt(x1[b])) for b in range(B)])
    np.testing.assert_allclose(la.cpu().numpy(), fx['logabs'][:3], atol=1e-9)
    assert (g.cpu() - g_ref).abs().max() < 1e-8 * max(1.0, float(g_ref.abs().max()))
    rng = np.random.default_rng(5)
    f_o = lambda p, xs: torch.stack([o_slog.apply(p, x) for x in xs])
    fg_o = lambda p, xs: (f_o(p, xs), torch.stack([tgrad(lambda y: o_slog.apply(p, y))(x) for x in xs]))
    lp1 = 2.0 * f_o(p_cpu, tt(x1))
    # one-electron move of electron 1
    nz, un = rng.standard_normal((B, 3)), rng.uniform(size=B)
    xo, lpo, na = oqmc.mh_one_electron_update(p_cpu, f_o, tt(x1), lp1, 0.0, cell.a, stddev=0.3, i=n3 // 3 + 1, normal=tt(nz), uniform=tt(un))
    nacc = torch.zeros(1, dtype=torch.float64, device='cuda')
    xg, _, lpg, nacc = qmc.mh_one_electron_update(dp, slog.apply, cu(x1), None, cu(lp1), nacc, cell.a, stddev=0.3, i=n3 // 3 + 1,
                                                  normal=cu(nz), uniform=cu(un))
    np.testing.assert_allclose(xg.cpu().numpy(), xo.numpy(), atol=1e-10)
    np.testing.assert_allclose(lpg.cpu().numpy(), lpo.numpy(), atol=1e-8)
    assert float(nacc) == float(na)
    # importance-sampled move
    nz, un = rng.standard_normal((B, n3)), rng.uniform(size=B)
    xo, lpo, na = oqmc.importance_update(p_cpu, fg_o, tt(x1), lp1, 0.0, cell.a, stddev=0.2, normal=tt(nz), uniform=tt(un))
    nacc = torch.zeros(1, dtype=torch.float64, device='cuda')
    xg, _, lpg, nacc = qmc.importance_update(dp, slog.apply.value_and_grad, cu(x1), None, cu(lp1), nacc, cell.a, stddev=0.2,
                                             normal=cu(nz), uniform=cu(un))
    np.testing.assert_allclose(xg.cpu().numpy(), xo.numpy(), atol=1e-9)
    np.testing.assert_allclose(lpg.cpu().numpy(), lpo.numpy(), atol=1e-7)
    assert float(nacc) == float(na)
    # the factories wire them up like qmc.py:319-333
    s1 = qmc.make_mcmc_step(slog.apply, B, cell.a, steps=1, one_electron_moves=True)
    xa, pa = s1(dp, cu(x1), 3, 0.05)
    assert xa.shape == (B, n3) and 0.0 <= float(pa) <= 1.0
    s2 = qmc.make_mcmc_step(slog.apply, B, cell.a, steps=2, importance_sampling=slog.apply)
    xb, pb = s2(dp, cu(x1), 3, 0.05)
    assert xb.shape == (B, n3) and 0.0 <= float(pb) <= 1.0


@pytest.mark.parametrize('name', ['h2', 'lih', 'lih_twist', 'bcc_li'])
def test_other_samplers_vs_reference_vectors(name):
    """One-electron moves (qmc.py:227-287; N moves through make_mcmc_step) and the drift-biased importance move
    (qmc.py:83-124) replaying the noise the REFERENCE's own functions consumed (tools/make_golden.py: mh1_*, imp_*)."""
    from deepsolid_amd import qmc
    fx, cell, klist, net_kw, params = load_case(name)
    dp = dev_params(params)
    (slog,) = nets(cell, klist, net_kw, 'eval_slogdet')
    cu = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64, device='cuda')
    B = fx['mcmc_x0'].shape[0]
    s1 = qmc.make_mcmc_step(slog.apply, B, cell.a, steps=1, one_electron_moves=True)
    xo, pm = s1(dp, cu(fx['mcmc_x0']), (cu(fx['mh1_normals']), cu(fx['mh1_uniforms'])), float(fx['mh1_width']))
    np.testing.assert_allclose(xo.cpu().numpy(), fx['mh1_x_out'], atol=1e-10)
    assert abs(float(pm) - float(fx['mh1_pmove'])) < 1e-15
    nacc = torch.zeros(1, dtype=torch.float64, device='cuda')
    xi, _, lpi, nacc = qmc.importance_update(dp, slog.apply.value_and_grad, cu(fx['mcmc_x0']), None, cu(fx['imp_lp1']), nacc, cell.a,
                                             stddev=float(fx['imp_width']), normal=cu(fx['imp_normal']), uniform=cu(fx['imp_uniform']))
    np.testing.assert_allclose(xi.cpu().numpy(), fx['imp_x_new'], atol=1e-9)
    np.testing.assert_allclose(lpi.cpu().numpy(), fx['imp_lp_new'], atol=1e-7)
    assert float(nacc) == float(fx['imp_num_accepts'])


def test_inference_loop_writes_reference_csv(tmp_path):
    """optimizer='none' loop of process.py:289-374 on LiH: CSV schema, width adaptation, finite energies."""
    from deepsolid_amd import inference, init_guess
    fx, cell, klist, net_kw, params = load_case('lih')
    dp = dev_params(params)
    slog, ld = nets(cell, klist, net_kw, 'eval_slogdet', 'eval_logdet')
    x0 = torch.as_tensor(init_guess.init_electrons(3, cell, cell.a, cell.nelec, 64, init_width=0.8), device='cuda')
    data, width, rows = inference.run_inference(slog, ld, dp, x0, cell, iterations=5, key=11, move_width=0.3, mcmc_steps=4,
                                                burn_in=3, adapt_frequency=2, save_path=str(tmp_path))
    assert len(rows) == 5 and data.shape == x0.shape
    assert all(np.isfinite(r['energy']) and 0 <= r['pmove'] <= 1 for r in rows)
    text = (tmp_path / 'train_stats.csv').read_text().splitlines()
    assert text[0] == 'step,energy,variance,pmove,imaginary,kinetic,ewald' and len(text) == 6
    assert width != 0.3                      # adapted at t = 2 and t = 4
    data2, width2, rows2 = inference.run_inference(slog, ld, dp, x0, cell, iterations=5, key=11, move_width=0.3, mcmc_steps=4,
                                                   burn_in=3, adapt_frequency=2)
    assert torch.equal(data, data2) and rows[-1]['energy'] == rows2[-1]['energy']      # reproducible from the seed


def test_fused_mcmc_step_philox():
    """`ds_mcmc_step` (qmc.py:335-362 as one C-ABI call, in-kernel Philox): pure function of the int key, stateful with
    a torch.Generator, walkers stay in the cell, moves have the requested width, and the acceptance rate agrees with
    the per-move Python loop over ds_mh_propose / ds_mh_accept driven by torch's generator (two independent noise
    sources, 4096 x 20 decisions: |dp| < 0.02)."""
    from deepsolid_amd import qmc, systems
    fx, cell, klist, net_kw, params = load_case('bcc_li')
    dp = dev_params(params)
    (slog,) = nets(cell, klist, net_kw, 'eval_slogdet')
    B, width, steps = 4096, 0.05, 20
    x0 = torch.as_tensor(systems.synthetic_walkers(cell, B, seed=5), device='cuda')
    step = qmc.make_mcmc_step(slog.apply, B, cell.a, steps=steps)
    xa, pa = step(dp, x0, 123, width)
    xb, pb = step(dp, x0, 123, width)
    xc, pc = step(dp, x0, 124, width)
    assert torch.equal(xa, xb) and float(pa) == float(pb)          # same key -> same chain
    assert not torch.equal(xa, xc)
    assert 0.05 < float(pa) < 0.999 and abs(float(pa) - float(pc)) < 0.02
    gen = torch.Generator(device='cuda').manual_seed(9)
    xg1, _ = step(dp, x0, gen, width)
    xg2, _ = step(dp, x0, gen, width)
    assert not torch.equal(xg1, xg2)                              # a Generator key advances
    ainv = torch.linalg.inv(torch.as_tensor(cell.a, device='cuda'))
    frac = xa.reshape(B, -1, 3) @ ainv
    assert frac.min() > -1e-9 and frac.max() < 1 + 1e-9
    # the per-move loop with torch noise (the generic path: atoms=None but forced through mh_update)
    nacc = torch.zeros(1, dtype=torch.float64, device='cuda')
    x, lp = x0.clone(), 2.0 * slog.apply(dp, x0)
    g2 = torch.Generator(device='cuda').manual_seed(77)
    for _ in range(steps):
        x, _, lp, nacc = qmc.mh_update(dp, slog.apply, x, g2, lp, nacc, cell.a, stddev=width)
    p_loop = float(nacc.item()) / (steps * B)
    assert abs(p_loop - float(pa)) < 0.02, (p_loop, float(pa))
    # one move from identical walkers: the accepted displacements are N(0, width^2) per coordinate (minimum image)
    one = qmc.make_mcmc_step(slog.apply, B, cell.a, steps=1)
    x1, p1 = one(dp, x0, 5, 0.01)
    d = (x1 - x0).reshape(B, -1, 3) @ ainv
    d = (d - torch.round(d)) @ torch.as_tensor(cell.a, device='cuda')
    moved = d.abs().sum(dim=(1, 2)) > 0
    assert abs(float(moved.double().mean()) - float(p1)) < 1e-12
    dm = d[moved].reshape(-1)
    assert abs(float(dm.mean())) < 2e-4 and abs(float(dm.std()) / 0.01 - 1) < 0.03
    # explicit noise through the fused call == explicit noise through the per-move kernels
    nz = torch.randn(3, 64, x0.shape[1], dtype=torch.float64, device='cuda', generator=g2)
    un = torch.rand(3, 64, dtype=torch.float64, device='cuda', generator=g2)
    s3 = qmc.make_mcmc_step(slog.apply, 64, cell.a, steps=3)
    xf, pf = s3(dp, x0[:64], (nz, un), 0.04)
    x, lp, nacc = x0[:64].clone(), 2.0 * slog.apply(dp, x0[:64]), torch.zeros(1, dtype=torch.float64, device='cuda')
    for i in range(3):
        x, _, lp, nacc = qmc.mh_update(dp, slog.apply, x, None, lp, nacc, cell.a, stddev=0.04, normal=nz[i], uniform=un[i])
    assert torch.equal(xf, x) and float(pf) == float(nacc.item()) / (3 * 64)


def test_energy_stats_and_nonfinite_count():
    """ds_energy_stats: the packed vector behind total_energy (train.py:74-82) and the non-finite count."""
    from deepsolid_amd import train
    fx, cell, klist, net_kw, params = load_case('lih')
    dp = dev_params(params)
    (net,) = nets(cell, klist, net_kw, 'eval_logdet')
    sysd = net.apply.system
    g = torch.Generator(device='cuda').manual_seed(1)
    ke = torch.randn(1000, 2, dtype=torch.float64, device='cuda', generator=g)
    ew = torch.randn(1000, dtype=torch.float64, device='cuda', generator=g)
    st = sysd.energy_stats(ke, ew).cpu().numpy()
    e = (ke[:, 0] + ew).cpu().numpy() + 1j * ke[:, 1].cpu().numpy()
    np.testing.assert_allclose(st, [e.real.sum(), e.imag.sum(), (np.abs(e) ** 2).sum(), 1000, 0, ke[:, 0].sum().item(),
                                    ke[:, 1].sum().item(), ew.sum().item()], rtol=1e-12, atol=1e-10)
    assert np.array_equal(st, sysd.energy_stats(ke, ew).cpu().numpy())         # fixed summation order
    ke[17, 0] = float('nan'); ew[400] = float('inf')
    st = sysd.energy_stats(ke, ew).cpu().numpy()
    assert st[4] == 2 and st[3] == 1000 and not np.isfinite(st[0])
    loss, aux = train.make_loss(net.apply, None, cell)(dp, torch.as_tensor(fx['x'], device='cuda'))
    assert float(aux.n_nonfinite) == 0.0 and np.isfinite(float(loss))


def test_bench_under_torchrun_with_rccl_one_rank():
    """The driver's launch line (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`) with N = 1 on the
    GPU box: the process group is RCCL (backend "nccl"), the packed statistics go through a real all-reduce on the device, and
    the JSON line reports the rank.  (N > 1 needs more GPUs than this box has; the CPU suite covers it with gloo.)"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', '29517', os.path.join(root, 'bench.py'), '--gpus', '1', '--system', 'lih', '--batch', '512', '--steps', '2',
           '--warmup', '1', '--no-cpu-baseline', '--no-mcmc']
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(line) == 1
    d = json.loads(line[0])
    assert d['n_gpus'] == 1 and d['ranks_seen'] == 1 and d['value'] > 0


@pytest.mark.parametrize('name', ['lih', 'bcc_li'])
def test_fused_one_electron_sampler(name):
    """`ds_mcmc_step_one_electron` (make_mcmc_step(one_electron_moves=True), qmc.py:227-287,355-358): with explicit noise it
    equals the per-move path (`mh_one_electron_update` called N * steps times from Python) bit for bit; with the in-kernel
    Philox stream a move is a pure function of its key, only the moved electron changes per move, and ranks / keys decorrelate."""
    from deepsolid_amd import qmc
    fx, cell, klist, net_kw, params = load_case(name)
    dp = dev_params(params)
    (slog,) = nets(cell, klist, net_kw, 'eval_slogdet')
    cu = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64, device='cuda')
    from deepsolid_amd import distance
    x0, _ = distance.enforce_pbc(cu(cell.a), cu(fx['mcmc_x0']))          # walkers inside the cell: an unmoved electron stays put
    B, n3 = x0.shape
    n = n3 // 3
    steps, width = 2, 0.25
    rng = np.random.default_rng(5)
    nz, un = rng.normal(size=(n * steps, B, 3)), rng.uniform(size=(n * steps, B))
    step = qmc.make_mcmc_step(slog.apply, B, cell.a, steps=steps, one_electron_moves=True)
    xf, pf = step(dp, x0, (cu(nz), cu(un)), width)
    # per-move reference path through the same kernels
    x, lp = x0.clone(), 2.0 * slog.apply(dp, x0)
    nacc = torch.zeros(1, dtype=torch.float64, device='cuda')
    for i in range(n * steps):
        x, _, lp, nacc = qmc.mh_one_electron_update(dp, slog.apply, x, None, lp, nacc, cell.a, stddev=width, i=i,
                                                    normal=cu(nz[i]), uniform=cu(un[i]))
    assert torch.equal(xf, x)
    assert abs(float(pf) - float(nacc[0]) / (n * steps * B)) < 1e-15
    # Philox mode
    xa, pa = step(dp, x0, 11, width)
    xb, pb = step(dp, x0, 11, width)
    xc, _ = step(dp, x0, 12, width)
    assert torch.equal(xa, xb) and float(pa) == float(pb) and not torch.equal(xa, xc)
    assert 0.0 < float(pa) <= 1.0
    # one move, first electron 0: electrons 1.. of every walker are untouched (up to the wrap of an already wrapped point)
    sysd = slog.apply.system
    x1 = x0.clone(); lp1 = torch.empty(B, dtype=torch.float64, device='cuda')
    sysd.mcmc_step(dp, x1, lp1, 1, width, seed=3, offset=0, first_electron=0)
    np.testing.assert_allclose(x1[:, 3:].cpu().numpy(), x0[:, 3:].cpu().numpy(), atol=1e-12)


@pytest.mark.parametrize('name', ['lih', 'bcc_li'])
def test_fused_importance_sampler(name):
    """`ds_mcmc_step_importance` (make_mcmc_step(importance_sampling=net.apply), qmc.py:83-150,324-325): with explicit noise it
    equals `importance_update` called per move from Python (the path the reference-executed imp_* vectors pin) bit for bit;
    with the in-kernel Philox stream a step is a pure function of its key."""
    from deepsolid_amd import qmc
    fx, cell, klist, net_kw, params = load_case(name)
    dp = dev_params(params)
    (slog,) = nets(cell, klist, net_kw, 'eval_slogdet')
    cu = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64, device='cuda')
    x0 = cu(fx['mcmc_x0'])
    B, n3 = x0.shape
    steps, width = 3, 0.12
    rng = np.random.default_rng(8)
    nz, un = rng.normal(size=(steps, B, n3)), rng.uniform(size=(steps, B))
    step = qmc.make_mcmc_step(slog.apply, B, cell.a, steps=steps, importance_sampling=slog.apply)
    xf, pf = step(dp, x0, (cu(nz), cu(un)), width)
    x, lp = x0.clone(), 2.0 * slog.apply(dp, x0)
    nacc = torch.zeros(1, dtype=torch.float64, device='cuda')
    for i in range(steps):
        x, _, lp, nacc = qmc.importance_update(dp, slog.apply.value_and_grad, x, None, lp, nacc, cell.a, stddev=width,
                                               normal=cu(nz[i]), uniform=cu(un[i]))
    assert torch.equal(xf, x)
    assert abs(float(pf) - float(nacc[0]) / (steps * B)) < 1e-15
    xa, pa = step(dp, x0, 21, width)
    xb, pb = step(dp, x0, 21, width)
    xc, _ = step(dp, x0, 22, width)
    assert torch.equal(xa, xb) and float(pa) == float(pb) and not torch.equal(xa, xc)
    assert 0.0 < float(pa) <= 1.0


@pytest.mark.parametrize('name', ['lih', 'bcc_li'])
def test_fused_asymmetric_sampler(name):
    """`ds_mcmc_step_asymmetric` (make_mcmc_step(atoms=...), mh_update's asymmetric branch qmc.py:197-215): with explicit noise
    it equals `mh_update(atoms=...)` called per move (the path the reference-executed mha_* vectors pin) bit for bit; Philox
    keys are pure."""
    from deepsolid_amd import qmc
    fx, cell, klist, net_kw, params = load_case(name)
    dp = dev_params(params)
    (slog,) = nets(cell, klist, net_kw, 'eval_slogdet')
    cu = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64, device='cuda')
    x0 = cu(fx['mcmc_x0'])
    B, n3 = x0.shape
    atoms = np.asarray(cell.atom_coords(), dtype=np.float64)
    steps, width = 3, 0.2
    rng = np.random.default_rng(9)
    nz, un = rng.normal(size=(steps, B, n3)), rng.uniform(size=(steps, B))
    step = qmc.make_mcmc_step(slog.apply, B, cell.a, steps=steps, atoms=atoms)
    xf, pf = step(dp, x0, (cu(nz), cu(un)), width)
    x, lp = x0.clone(), 2.0 * slog.apply(dp, x0)
    nacc = torch.zeros(1, dtype=torch.float64, device='cuda')
    for i in range(steps):
        x, _, lp, nacc = qmc.mh_update(dp, slog.apply, x, None, lp, nacc, cell.a, stddev=width, atoms=atoms,
                                       normal=cu(nz[i]), uniform=cu(un[i]))
    assert torch.equal(xf, x)
    assert abs(float(pf) - float(nacc[0]) / (steps * B)) < 1e-15
    xa, pa = step(dp, x0, 31, width)
    xb, pb = step(dp, x0, 31, width)
    xc, _ = step(dp, x0, 32, width)
    assert torch.equal(xa, xb) and float(pa) == float(pb) and not torch.equal(xa, xc)
    assert 0.0 < float(pa) <= 1.0
