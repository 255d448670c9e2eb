#!/usr/bin/env python3
"""Generate tests/golden/*.npz by EXECUTING the reference's own code.

Runs only in the build container (needs /root/reference, which does not exist
on the GPU box).  JAX and PySCF are not installable here, so the reference's
``network.py`` / ``ewaldsum.py`` / ``distance.py`` / ``supercell.py`` /
``qmc.py`` are imported with in-memory stand-in modules:

  * ``jax.numpy``   -> numpy (``sum`` accepts a list ``axis`` like jnp does)
  * ``jax.vmap``    -> python loop + stack (tuple outputs, ``None`` in_axes)
  * ``jax.lax.erfc``-> scipy.special.erfc ; ``jax.jit``/``jax.pmap`` -> identity
  * ``jax.lax.fori_loop`` -> python loop ; ``jax.random`` -> a recording numpy
    Generator (so the Metropolis noise can be replayed by the tests)
  * ``DeepSolid.curvature_tags_and_blocks`` -> identity tags (KFAC bookkeeping)
  * ``pyscf`` -> empty module; cells are a 20-line attribute bag

No autodiff exists in this stand-in, so ``hamiltonian.py`` cannot be executed;
kinetic-energy vectors are 4th-order finite differences of the
reference-executed ``eval_logdet`` (tolerance recorded in the file).

Nothing produced here except the small .npz numbers is shipped.  Parameters
are NOT stored: they are regenerated from ``oracle.testing.make_test_params``
(numpy default_rng) and guarded by a checksum.
"""
import os
import sys
import types
from functools import partial

import numpy as np
import scipy.special

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
REF = os.environ.get('DEEPSOLID_REFERENCE', '/root/reference')


# ----------------------------------------------------------------------------- shim
from tools import jax_torch_standin as standin          # noqa: E402
from tools.jax_torch_standin import RECORDER            # noqa: E402


def _install_shim():
    standin.install(REF)


class FakeCell:
    """The attributes of pyscf.pbc.gto.Cell the reference hot path touches."""
    def __init__(self, a, coords, charges, nelec):
        self.a = np.asarray(a, float)
        self._coords = np.asarray(coords, float).reshape(-1, 3)
        self._charges = np.asarray(charges)
        self.nelec = tuple(int(n) for n in nelec)
        self.nelectron = sum(self.nelec)
        self.original_cell = self
        self.S = np.eye(3)
        self.scale = 1

    def lattice_vectors(self): return self.a
    def reciprocal_vectors(self): return 2 * np.pi * np.linalg.inv(self.a).T
    def atom_coords(self): return self._coords
    def atom_charges(self): return self._charges


def ref_supercell(ref_sc, prim, S, nelec, sym_type='minimal'):
    """reference supercell.get_supercell (:64-95) minus the PySCF build()."""
    S = np.asarray(S, float)
    Rpts = ref_sc.get_supercell_copies(prim.lattice_vectors(), S)
    coords, charges = [], []
    for xyz, q in zip(prim.atom_coords(), prim.atom_charges()):
        for R in Rpts:
            coords.append(xyz + R)
            charges.append(q)
    sc = FakeCell(np.dot(S, prim.lattice_vectors()), coords, charges, nelec)
    sc.original_cell = prim
    sc.S = S
    sc.scale = abs(int(np.round(np.linalg.det(S))))
    return ref_sc.set_symmetry_lat(sc, sym_type)



# ----------------------------------------------------------------------------- autodiff legs
class TorchCell:
    """The same attribute bag with torch float64 leaves: what the reference's network.py reads
    when hamiltonian.py / train.py differentiate it under the torch-backed `jax` stand-in."""
    def __init__(self, c, original=None, energy_nuc=None):
        import torch
        T = lambda v: torch.as_tensor(np.asarray(v, dtype=np.float64))
        self.a, self.AV, self.BV = T(c.a), T(c.AV), T(c.BV)
        self._coords = T(c.atom_coords())
        self.nelec = c.nelec
        self.original_cell = original if original is not None else self
        self.scale = c.scale
        self._energy_nuc = energy_nuc

    def atom_coords(self): return self._coords
    def lattice_vectors(self): return self.a
    def energy_nuc(self): return self._energy_nuc


def to_torch_params(p):
    import torch
    return standin.tree_map(lambda a: torch.as_tensor(np.asarray(a, dtype=np.float64)), p)


def flatten_leaves(tree):
    """Leaves in the order of oracle.testing (sorted dict keys, list order)."""
    out = []

    def walk(o, path):
        if isinstance(o, dict):
            for k in sorted(o):
                walk(o[k], path + (k,))
        elif isinstance(o, (list, tuple)):
            for i, v in enumerate(o):
                walk(v, path + (i,))
        else:
            out.append(('/'.join(str(q) for q in path), o))
    walk(tree, ())
    return out

# ----------------------------------------------------------------------------- cases
def to_np_params(p):
    def conv(o):
        if isinstance(o, dict):
            return {k: conv(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return [conv(v) for v in o]
        return np.asarray(o, dtype=np.float64)
    return conv(p)


def fd_kinetic(f, x, h=2e-3):
    """-1/2 sum_i [d2f/dx_i^2 + (df/dx_i)^2] by 4th-order central differences of
    the complex f = log psi (branch of Im f unwrapped against the centre)."""
    f0 = f(x)
    tot = 0.0 + 0.0j

    def val(xx):
        v = f(xx)
        dphi = np.angle(np.exp(1j * (v.imag - f0.imag)))
        return v.real + 1j * (f0.imag + dphi)
    for i in range(x.size):
        e = np.zeros_like(x); e[i] = h
        fp1, fm1, fp2, fm2 = val(x + e), val(x - e), val(x + 2 * e), val(x - 2 * e)
        d1 = (-fp2 + 8 * fp1 - 8 * fm1 + fm2) / (12 * h)
        d2 = (-fp2 + 16 * fp1 - 30 * f0 + 16 * fm1 - fm2) / (12 * h * h)
        tot += d2 + d1 * d1
    return -0.5 * tot


def main():
    _install_shim()
    from DeepSolid import network as rnet, ewaldsum as rewald, distance as rdist
    from DeepSolid import supercell as rsc, qmc as rqmc, hamiltonian as rham, train as rtrain
    from deepsolid_amd import systems, supercell as my_sc
    from oracle.testing import make_test_params, params_checksum, klist_from_kpts, CASES

    out_dir = os.path.join(REPO, 'tests', 'golden')
    os.makedirs(out_dir, exist_ok=True)

    only = set(sys.argv[1:])
    for name, case in CASES.items():
        if only and name not in only:
            continue
        my_cell = systems.SYSTEMS[case['system']](**case.get('system_kw', {}))
        prim0 = my_cell.original_cell
        # rebuild the cell with the REFERENCE's supercell code from primitive data only
        prim = FakeCell(prim0.a, prim0.atom_coords(), prim0.atom_charges(), prim0.nelec)
        sim = ref_supercell(rsc, prim, my_cell.S, my_cell.nelec, case.get('sym_type', 'minimal'))
        kpts = rsc.get_supercell_kpts(sim)
        twist = np.asarray(case.get('twist', (0, 0, 0)), float)
        kpts_t = kpts + np.dot(np.linalg.inv(sim.a), np.mod(twist, 1.0)) * 2 * np.pi    # hf.py:61-62 (cell = the simulation cell)
        klist = klist_from_kpts(kpts_t, sim.nelec)   # grouping shaped like hf.py:99-104
        N = sum(sim.nelec)
        net_kw = dict(systems.DETNET_DEFAULTS); net_kw.update(case.get('net_kw', {}))
        params = make_test_params(case['seed'], prim.atom_coords(), sim.nelec, net_kw)
        pnp = to_np_params(params)
        nets = {m: rnet.make_solid_fermi_net(klist=klist, simulation_cell=sim, method_name=m, **net_kw)
                for m in ('eval_phase_and_slogdet', 'eval_mats', 'eval_logdet', 'eval_slogdet')}
        B = case.get('batch', 4)
        x = systems.synthetic_walkers(my_cell, B, seed=case['seed'] + 100)
        # push some walkers outside the cell so both wraps (primitive, simulation) are exercised
        rng = np.random.default_rng(case['seed'] + 200)
        shifts = rng.integers(-2, 3, size=(B, N, 3)) @ sim.a
        shifts[0] = 0
        x = x + shifts.reshape(B, -1)
        d = dict(prim_a=prim.a, prim_atoms=prim.atom_coords(), prim_charges=prim.atom_charges(),
                 S=sim.S, sim_a=sim.a, sim_atoms=sim.atom_coords(), sim_charges=sim.atom_charges(),
                 nelec=np.asarray(sim.nelec), prim_AV=prim.AV, prim_BV=prim.BV, sim_AV=sim.AV,
                 sim_BV=sim.BV, kpts=kpts, twist=twist, klist_up=klist[0], klist_dn=klist[1],
                 x=x, seed=case['seed'], params_checksum=params_checksum(params))
        ph, ls, mats = [], [], [[] for _ in range(2)]
        feats = [[] for _ in range(4)]
        for b in range(B):
            p_, l_ = nets['eval_phase_and_slogdet'].apply(pnp, x[b])
            ph.append(p_); ls.append(l_)
            m = nets['eval_mats'].apply(pnp, x[b])
            for s, mm in enumerate(m):
                mats[s].append(mm)
            ft = rnet.construct_periodic_input_features(x[b], prim.atom_coords(), simulation_cell=sim,
                                                        distance_type=net_kw['distance_type'])
            for k in range(4):
                feats[k].append(ft[k])
        d.update(phase=np.asarray(ph), logabs=np.asarray(ls))
        for s in range(2):
            if mats[s]:
                d[f'orbitals_{s}'] = np.asarray(mats[s])
        for k, nm in enumerate(('feat_ae', 'feat_ee', 'feat_r_ae', 'feat_r_ee')):
            d[nm] = np.asarray(feats[k])

        # --- Ewald (reference EwaldSum) -------------------------------------------------
        ew = rewald.EwaldSum(sim)
        en = np.asarray([[float(v) for v in ew.energy(x[b])] for b in range(B)])
        d.update(ewald=en, ewald_alpha=float(ew.alpha), ewald_ng=int(ew.gpoints.shape[0]),
                 ewald_gweight_sum=float(np.sum(ew.gweight)),
                 ewald_gnorm_sum=float(np.sum(np.linalg.norm(ew.gpoints, axis=1))),
                 ewald_ion_ion=float(ew.ion_ion), ewald_ii_const=float(ew.ii_const),
                 dist_mode={'diagonal_dist_i': 0, 'orthogonal_dist_i': 1, 'general_dist_i': 2}[ew.dist.dist_i.__name__])
        if ew.gpoints.shape[0] < 5000:
            d.update(ewald_gpoints=np.asarray(ew.gpoints), ewald_gweight=np.asarray(ew.gweight))

        # --- batched wrap (distance.enforce_pbc) ----------------------------------------
        wx, wrap = rdist.enforce_pbc(sim.a, x)
        d.update(pbc_x=wx, pbc_wrap=wrap)

        # --- Metropolis: one mh_update and a 3-step mcmc_step with recorded noise --------
        if case.get('mcmc', True):
            f_batch = lambda p, xs: np.asarray([nets['eval_slogdet'].apply(p, xx) for xx in xs])
            lp1 = 2.0 * f_batch(pnp, wx)
            RECORDER.reset(case['seed'] + 300)
            xn, _, lpn, nacc = rqmc.mh_update(pnp, f_batch, wx, None, lp1, 0.0, sim.a, stddev=0.05)
            d.update(mh_x1=wx, mh_lp1=lp1, mh_normal=RECORDER.normals[0], mh_uniform=RECORDER.uniforms[0],
                     mh_width=0.05, mh_x_new=xn, mh_lp_new=lpn, mh_num_accepts=float(nacc))
            # asymmetric proposal (atoms given: step width scaled by the harmonic mean of the nuclear distances, qmc.py:197-215)
            RECORDER.reset(case['seed'] + 350)
            xa, _, lpa, nacca = rqmc.mh_update(pnp, f_batch, wx, None, lp1, 0.0, sim.a, stddev=0.05, atoms=prim.atom_coords())
            d.update(mha_normal=RECORDER.normals[0], mha_uniform=RECORDER.uniforms[0], mha_x_new=xa, mha_lp_new=lpa,
                     mha_num_accepts=float(nacca))
            RECORDER.reset(case['seed'] + 400)
            step = rqmc.make_mcmc_step(f_batch, B, sim.a, steps=3)
            xs3, pmove = step(pnp, wx, None, 0.08)
            d.update(mcmc_x0=wx, mcmc_normals=np.asarray(RECORDER.normals),
                     mcmc_uniforms=np.asarray(RECORDER.uniforms), mcmc_width=0.08, mcmc_steps=3,
                     mcmc_x_out=xs3, mcmc_pmove=float(pmove))


        # --- the reference's other samplers (flagged "untested" in base_config.py:122-126), replayable noise ----------
        if case.get('mcmc', True) and case.get('samplers', True):
            import torch
            # one-electron moves (qmc.py:227-287) through make_mcmc_step: N moves, electron i = step % N (:355-356)
            RECORDER.reset(case['seed'] + 700)
            step1 = rqmc.make_mcmc_step(f_batch, B, sim.a, steps=1, one_electron_moves=True)
            x1e, pm1 = step1(pnp, wx.view(standin.AtArray), None, 0.3)
            d.update(mh1_normals=np.asarray(RECORDER.normals).reshape(N, B, 3), mh1_uniforms=np.asarray(RECORDER.uniforms),
                     mh1_width=0.3, mh1_x_out=np.asarray(x1e), mh1_pmove=float(pm1))
            # drift-biased importance sampling (qmc.py:83-124): f = value and gradient of log|psi| from the reference network
            tcell_p = TorchCell(prim); tcell = TorchCell(sim, original=tcell_p)
            with standin.torch_mode():
                tnet_s = rnet.make_solid_fermi_net(klist=[torch.as_tensor(k) for k in klist], simulation_cell=tcell,
                                                   method_name='eval_slogdet', **net_kw)
            tpar = to_torch_params(pnp)

            def f_vg(p_, xs):
                vals, grads = [], []
                with standin.torch_mode():
                    for xx in np.asarray(xs):
                        xt = torch.as_tensor(xx).clone().requires_grad_(True)
                        v = tnet_s.apply(tpar, xt)
                        g, = torch.autograd.grad(v, xt)
                        vals.append(float(v)); grads.append(g.numpy())
                return np.asarray(vals), np.asarray(grads)
            RECORDER.reset(case['seed'] + 800)
            lpi = 2.0 * f_vg(pnp, wx)[0]
            xi, _, lpin, nacci = rqmc.importance_update(pnp, f_vg, wx, None, lpi, 0.0, sim.a, stddev=0.2)
            d.update(imp_lp1=lpi, imp_normal=RECORDER.normals[0], imp_uniform=RECORDER.uniforms[0], imp_width=0.2,
                     imp_x_new=np.asarray(xi), imp_lp_new=np.asarray(lpin), imp_num_accepts=float(nacci))

        # --- kinetic energy: the reference's OWN hamiltonian.py over its own network.py -----
        # (torch-backed jax stand-in: jax.grad/jvp/hessian -> torch.func, float64)
        nke = case.get('ke_walkers', 0)
        if nke:
            import torch
            from types import SimpleNamespace
            from deepsolid_amd.ewaldsum import EwaldTables
            tab = EwaldTables(my_cell)
            e_nuc = float(tab.ion_ion + tab.ii_const)          # stands in for pyscf's cell.energy_nuc() (hamiltonian.py:170)
            tprim = TorchCell(prim)
            tsim = TorchCell(sim, original=tprim, energy_nuc=e_nuc)
            tklist = [torch.as_tensor(k) for k in klist]
            tparams = to_torch_params(pnp)

            class _RefEwald:
                """The reference's EwaldSum built on the numpy cell; only converts the walker to numpy."""
                def __init__(self, _cell):
                    self.ion_ion, self.ii_const = ew.ion_ion, ew.ii_const

                def energy(self, xt):
                    with standin.torch_mode(False):
                        return [torch.as_tensor(float(v), dtype=torch.float64) for v in ew.energy(standin.to_numpy(xt))]
            rham.ewaldsum = SimpleNamespace(EwaldSum=_RefEwald)
            with standin.torch_mode():
                tnet = rnet.make_solid_fermi_net(klist=tklist, simulation_cell=tsim, method_name='eval_logdet', **net_kw)
                modes = case.get('ke_modes', ('for',))
                for mode in modes:
                    pn = 3 if (3 * N) % 3 == 0 else 1
                    el = rham.local_energy_seperate(tnet.apply, tsim, mode=mode, partition_number=pn)
                    kes, ews = [], []
                    for b in range(nke if mode == 'for' else 1):
                        k_, e_ = el(tparams, torch.as_tensor(x[b]))
                        kes.append(complex(k_)); ews.append(float(e_))
                    key = 'ke_ref' if mode == 'for' else 'ke_ref_' + mode
                    d[key] = np.asarray(kes)
                    if mode == 'for':
                        d['ew_ref'] = np.asarray(ews)
                # --- energy gradient: the reference's train.make_loss + jax.value_and_grad (process.py:191-204)
                ngr = case.get('grad_walkers', 0)
                if ngr:
                    import jax
                    batch_net = jax.vmap(tnet.apply, in_axes=(None, 0), out_axes=0)
                    for clip_type in case.get('grad_clip_types', ('real',)):
                        total_energy = rtrain.make_loss(network=tnet.apply, batch_network=batch_net, simulation_cell=tsim,
                                                        clip_local_energy=5.0, clip_type=clip_type, mode='for')
                        vg = jax.value_and_grad(total_energy, argnums=0, has_aux=True)
                        (loss, aux), g = vg(tparams, torch.as_tensor(x[:ngr]))
                        from oracle.testing import make_test_direction
                        vdir = make_test_direction(case['seed'] + 600, params)
                        gl, vl = flatten_leaves(g), flatten_leaves(vdir)
                        sfx = '' if clip_type == 'real' else '_' + clip_type
                        d['grad_ref_loss' + sfx] = float(loss)
                        d['grad_ref_variance' + sfx] = float(aux.variance)
                        d['grad_ref_imag' + sfx] = float(aux.imaginary)
                        d['grad_ref_names'] = np.asarray([n for n, _ in gl])
                        d['grad_ref_norm' + sfx] = np.asarray([float(torch.linalg.vector_norm(t)) for _, t in gl])
                        d['grad_ref_dot' + sfx] = np.asarray([float((t * torch.as_tensor(v)).sum()) for (_, t), (_, v) in zip(gl, vl)])
                        d['grad_ref_seed'] = case['seed'] + 600
                        d['grad_ref_walkers'] = ngr
                        for n_, t in gl:
                            if t.numel() <= 4096:
                                d['grad_ref_leaf' + sfx + ':' + n_] = t.numpy()
        # --- kinetic energy by finite differences of the reference-executed forward -----
        nfd = case.get('fd_walkers', 0)
        if nfd:
            f = lambda xx: complex(nets['eval_logdet'].apply(pnp, xx))
            h = case.get('fd_h', 2e-3)
            d.update(ke_fd=np.asarray([fd_kinetic(f, x[b], h=h) for b in range(nfd)]),
                     ke_fd_h=h, ke_fd_tol=case.get('fd_tol', 5e-6))
        # --- parameter gradient: directional derivative of the reference-executed forward ----
        ngf = case.get('gradfd_walkers', 0)
        if ngf:
            from oracle.testing import make_test_direction, tree_axpy
            vdir = make_test_direction(case['seed'] + 500, params)
            h = 2.5e-4
            dl, da = [], []
            for b in range(ngf):
                vals = {}
                for m in (-2, -1, 1, 2):
                    vals[m] = nets['eval_phase_and_slogdet'].apply(tree_axpy(pnp, m * h, vdir), x[b])
                d4 = lambda k: (-vals[2][k] + 8 * vals[1][k] - 8 * vals[-1][k] + vals[-2][k]) / (12 * h)
                ph0 = complex(d['phase'][b])
                dl.append(float(np.real(d4(1))))
                da.append(float(np.imag(np.conj(ph0) * d4(0))))       # d arg = Im(conj(phase) d phase)
            d.update(gradfd_dlogabs=np.asarray(dl), gradfd_darg=np.asarray(da), gradfd_h=h, gradfd_seed=case['seed'] + 500)
        path = os.path.join(out_dir, name + '.npz')
        np.savez_compressed(path, **d)
        print(name, 'N=%d' % N, 'NG=%d' % d['ewald_ng'], 'logabs', d['logabs'][:2],
              '%.1f KB' % (os.path.getsize(path) / 1024))


if __name__ == '__main__':
    main()
