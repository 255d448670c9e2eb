"""Oracle (test infrastructure): shared recipes for golden vectors and tests.

``make_test_params`` is the single recipe for network parameters used by the
fixture generator (tools/make_golden.py, which feeds them to the reference's
own code) and by the tests (which feed them to the oracle and to the HIP
path), so the parameters themselves never need to be stored: a checksum in the
fixture guards the numpy stream.
"""
import numpy as np

from . import network


def make_test_params(seed, prim_atoms, nelec, net_kw, env_jitter=0.2):
    """Reference-shaped parameter tree (network.py:60-186) from numpy
    default_rng(seed).  The envelope pi/sigma (ones in the reference init) are
    jittered so that their indexing is exercised by the parity tests."""
    rng = np.random.default_rng(seed)
    kw = {k: net_kw[k] for k in ('envelope_type', 'bias_orbitals', 'use_last_layer', 'full_det',
                                 'hidden_dims', 'determinants', 'distance_type')}
    params = network.init_solid_fermi_net_params(rng, prim_atoms, tuple(int(n) for n in nelec), **kw)
    for env in params['envelope']:
        env['pi'] = env['pi'] * (1.0 + env_jitter * rng.uniform(-1, 1, size=env['pi'].shape))
        env['sigma'] = env['sigma'] * (1.0 + env_jitter * rng.uniform(-1, 1, size=env['sigma'].shape))
    return params


def make_test_direction(seed, params):
    """A parameter-space direction shaped like `params` (numpy default_rng(seed), leaves visited in
    sorted-key order, each scaled by the leaf's rms): the finite-difference fixtures of the parameter
    gradient (tools/make_golden.py 'gradfd_*') are directional derivatives along it."""
    rng = np.random.default_rng(seed)

    def walk(o):
        if isinstance(o, dict):
            return {k: walk(o[k]) for k in sorted(o)}
        if isinstance(o, (list, tuple)):
            return [walk(v) for v in o]
        a = np.asarray(o, dtype=np.float64)
        rms = float(np.sqrt(np.mean(a * a))) if a.size else 1.0
        return rng.normal(size=a.shape) * (rms if rms > 0 else 1.0)
    return walk(params)


def tree_axpy(params, h, direction):
    """params + h * direction, leaf by leaf (numpy)."""
    if isinstance(params, dict):
        return {k: tree_axpy(params[k], h, direction[k]) for k in params}
    if isinstance(params, (list, tuple)):
        return [tree_axpy(p, h, d) for p, d in zip(params, direction)]
    return np.asarray(params, dtype=np.float64) + h * np.asarray(direction)


def params_checksum(params):
    tot, tot2 = 0.0, 0.0

    def walk(o):
        nonlocal tot, tot2
        if isinstance(o, dict):
            for k in sorted(o):
                walk(o[k])
        elif isinstance(o, (list, tuple)):
            for v in o:
                walk(v)
        else:
            a = np.asarray(o, dtype=np.float64)
            tot += float(a.sum())
            tot2 += float((a * a).sum())
    walk(params)
    return np.asarray([tot, tot2])


def klist_from_kpts(kpts, nelec):
    """Occupied k list per spin: k-points filled in order, each repeated by its
    occupation (shape of hf.SCF.klist, reference hf.py:99-104)."""
    nk = kpts.shape[0]
    out = []
    for ns in nelec:
        base, rem = divmod(int(ns), nk)
        rows = [np.tile(k[None, :], (base + (1 if i < rem else 0), 1)) for i, k in enumerate(kpts)]
        out.append(np.concatenate(rows, axis=0) if int(ns) > 0 else np.zeros((0, 3)))
    return out


# name -> recipe.  `system` keys deepsolid_amd.systems.SYSTEMS.
#   ke_walkers   : walkers whose kinetic energy comes from the reference's own hamiltonian.py (modes in ke_modes)
#   grad_walkers : batch for the reference's own train.make_loss + jax.value_and_grad (clip types in grad_clip_types)
#   fd_walkers / gradfd_walkers : finite differences of the reference-executed forward (side checks)
_ALL_MODES = ('for', 'hessian', 'dim_batch', 'partition')
CASES = {
    'h2':            dict(system='h2', seed=11, batch=4, fd_walkers=2, ke_walkers=4, ke_modes=_ALL_MODES, grad_walkers=4),
    'lih':           dict(system='lih', seed=12, batch=6, fd_walkers=3, gradfd_walkers=3, ke_walkers=6,
                          ke_modes=_ALL_MODES, grad_walkers=6, grad_clip_types=('real', 'complex')),
    'lih_twist':     dict(system='lih', seed=13, batch=4, twist=(0.25, 0.1, 0.4), fd_walkers=2, gradfd_walkers=2,
                          ke_walkers=4, ke_modes=_ALL_MODES, grad_walkers=4, grad_clip_types=('real', 'complex')),
    'lih_2x1x1':     dict(system='lih', seed=14, batch=3, system_kw=dict(S=np.diag([2, 1, 1])), mcmc=False,
                          ke_walkers=3, grad_walkers=3),
    'bcc_li':        dict(system='bcc_li', seed=15, batch=4, fd_walkers=1, fd_h=5e-4, fd_tol=1e-5, gradfd_walkers=2,
                          ke_walkers=4, ke_modes=('for', 'hessian'), grad_walkers=4),
    'bcc_li_twist':  dict(system='bcc_li', seed=16, batch=2, twist=(0.3, 0.0, 0.15), mcmc=False, ke_walkers=2),
    'graphene':      dict(system='graphene', seed=17, batch=4, mcmc=False, ke_walkers=4, ke_modes=('for', 'hessian')),
    'diamond':       dict(system='diamond', seed=18, batch=4, mcmc=False, ke_walkers=4),
    'lih_fulldet':   dict(system='lih', seed=19, batch=3, net_kw=dict(full_det=True), mcmc=False, gradfd_walkers=2,
                          ke_walkers=3, grad_walkers=3),
    'lih_tri':       dict(system='lih', seed=20, batch=3, net_kw=dict(distance_type='tri'), mcmc=False, gradfd_walkers=2,
                          ke_walkers=3, grad_walkers=3),
    'lih_diagenv':   dict(system='lih', seed=21, batch=3, net_kw=dict(envelope_type='diagonal'), mcmc=False,
                          ke_walkers=3, grad_walkers=3),
    'lih_fullenv':   dict(system='lih', seed=22, batch=3, net_kw=dict(envelope_type='full'), mcmc=False,
                          ke_walkers=3, grad_walkers=3),
    'lih_lastlayer': dict(system='lih', seed=26, batch=3, net_kw=dict(use_last_layer=True), mcmc=False,
                          ke_walkers=3, grad_walkers=3),
    'lih_bias':      dict(system='lih', seed=23, batch=3, net_kw=dict(bias_orbitals=True), mcmc=False,
                          ke_walkers=3, grad_walkers=3),
    # the defaults of the make_solid_fermi_net SIGNATURE (network.py:609-621), not of base_config.py
    'lih_fn_defaults': dict(system='lih', seed=24, batch=3, mcmc=False, ke_walkers=3, grad_walkers=3,
                            net_kw=dict(envelope_type='full', full_det=True, determinants=16)),
    'bcc_li_fulldet': dict(system='bcc_li', seed=25, batch=2, net_kw=dict(full_det=True), mcmc=False, ke_walkers=2,
                           grad_walkers=2),
    # the non-minimal feature lattices of supercell.set_symmetry_lat (:103-129): 4 rows of AV/BV (fcc, hexagonal), 6 rows (bcc)
    'lih_fcc':       dict(system='lih', seed=27, batch=3, sym_type='fcc', mcmc=False, ke_walkers=3, grad_walkers=3),
    'graphene_hex':  dict(system='graphene', seed=28, batch=2, system_kw=dict(S=1), sym_type='hexagonal', mcmc=False,
                          ke_walkers=2, grad_walkers=2),
    'bcc_li_bcc':    dict(system='bcc_li', seed=29, batch=2, sym_type='bcc', mcmc=False, ke_walkers=2),
    # a fully spin-polarised cell: the empty spin channel is dropped everywhere (network.py:327-328, :537-541)
    # determinant counts that do not fill the orbital head's 8-orbital column tiles (3 x 2 = 6 orbitals per spin; 1 x 12)
    'lih_det3':      dict(system='lih', seed=31, batch=3, net_kw=dict(determinants=3), mcmc=False, samplers=False, ke_walkers=3, grad_walkers=3),
    'bcc_li_det1':   dict(system='bcc_li', seed=32, batch=2, net_kw=dict(determinants=1), mcmc=False, samplers=False, ke_walkers=2, grad_walkers=2),
    # other layer widths: narrow streams, and widths that change from layer to layer (no residual connection there, network.py:519)
    'lih_narrow':    dict(system='lih', seed=33, batch=3, net_kw=dict(hidden_dims=((128, 16), (128, 16), (128, 16))), mcmc=False,
                          samplers=False, ke_walkers=3, grad_walkers=3),
    'lih_mixed':     dict(system='lih', seed=34, batch=3, net_kw=dict(hidden_dims=((256, 32), (128, 16), (192, 32), (192, 32))), mcmc=False,
                          samplers=False, ke_walkers=3, grad_walkers=3),
    'li_polarized':  dict(system='bcc_li', seed=30, batch=3, system_kw=dict(S=1, nelec=(3, 0)), mcmc=False, samplers=False,
                          ke_walkers=3, grad_walkers=3),
    # supercells between and beyond the BASELINE sizes: 81 electrons (41 + 40: 16 jet-slot tiles, odd matrix size) and
    # 108 electrons (54 + 54: 21 jet-slot tiles) -- ordinary DeepSolid cells (supercell.py:64-95, network.py:60-186)
    'bcc_li_333':    dict(system='bcc_li', seed=35, batch=2, system_kw=dict(S=3), mcmc=False, samplers=False, ke_walkers=1),
    'graphene_331':  dict(system='graphene', seed=36, batch=2, system_kw=dict(S=3), mcmc=False, samplers=False, ke_walkers=1),
}
