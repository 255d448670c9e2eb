"""Oracle (test infrastructure): periodic FermiNet log-psi, torch CPU float64.

Restates /root/reference/DeepSolid/network.py function by function; every
function cites the reference lines it follows.  All functions are written per
walker (x is a flat (3N,) vector) exactly like the reference and are batched by
the caller.  Not imported by the product package.
"""
import math
from types import SimpleNamespace

import numpy as np
import torch

DT = torch.float64
CDT = torch.complex128


_WORK = {'dt': DT}


class working_dtype:
    """Context manager: run the restatement in another real dtype (float32 for the fp32 error budget of BASELINE config 5:
    what a straight float32 evaluation of the reference algorithm gives).  Build the network inside the context."""
    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        self.old = _WORK['dt']
        _WORK['dt'] = self.dtype

    def __exit__(self, *exc):
        _WORK['dt'] = self.old


def cdtype():
    return torch.complex128 if _WORK['dt'] == torch.float64 else torch.complex64


def _t(a, dtype=None):
    dtype = dtype or _WORK['dt']
    if isinstance(a, torch.Tensor):
        return a.to(dtype)
    return torch.as_tensor(np.asarray(a), dtype=dtype)


# --------------------------------------------------------------------------- #
# network.py:42-57
def enforce_pbc(latvec, epos):
    """frac = epos @ inv(latvec); wrap = floor(frac); back to Cartesian."""
    recpvecs = torch.linalg.inv(latvec)
    frac = epos @ recpvecs
    wrap = torch.floor(frac)          # `// 1`; zero derivative like JAX
    return (frac - wrap) @ latvec, wrap


# network.py:189-204
def scaled_f(w):
    return torch.abs(w) * (1 - torch.abs(w / math.pi) ** 3 / 4.0)


def scaled_g(w):
    aw = torch.abs(w / math.pi)
    return w * (1 - 1.5 * aw + 0.5 * aw ** 2)


# network.py:207-224
def nu_distance(xea, a, b):
    """xea (..., 3); a = AV (L,3), b = BV (L,3).  Returns sd (...,), rel (...,3)."""
    w = torch.einsum('...k,lk->...l', xea, b)
    mod = torch.floor((w + math.pi) / (2 * math.pi))
    w = w - mod * 2 * math.pi
    r1 = (torch.linalg.norm(a, dim=-1) * scaled_f(w)) ** 2
    sg = scaled_g(w)
    rel = torch.einsum('...i,ij->...j', sg, a)
    metric = a @ a.T
    r2 = metric * (sg[..., :, None] * sg[..., None, :])
    L = a.shape[0]
    offdiag = torch.ones(L, L, dtype=a.dtype) - torch.eye(L, dtype=a.dtype)
    result = r1.sum(-1) + (r2 * offdiag).sum((-1, -2))
    return result ** 0.5, rel


# network.py:227-246
def tri_distance(xea, a, b):
    w = torch.einsum('...k,lk->...l', xea, b)
    sg, cg = torch.sin(w), torch.cos(w)
    rel = torch.cat([torch.einsum('...i,ij->...j', sg, a),
                     torch.einsum('...i,ij->...j', cg, a)], dim=-1)
    metric = a @ a.T
    vec = (1 - cg[..., :, None]) * (1 - cg[..., None, :]) + sg[..., :, None] * sg[..., None, :]
    sd = torch.einsum('...ij,ij->...', vec, metric) ** 0.5
    return sd, rel


# network.py:249-302
def construct_periodic_input_features(x, atoms, simulation_cell, distance_type='nu'):
    if distance_type == 'nu':
        dist = nu_distance
    elif distance_type == 'tri':
        dist = tri_distance
    else:
        raise ValueError('Unrecognized distance function.')
    prim = simulation_cell.original_cell
    x = x.reshape(-1, 3)
    n = x.shape[0]
    prim_x, _ = enforce_pbc(_t(prim.a), x)
    prim_xea = prim_x[:, None, :] - atoms
    sea, xea = dist(prim_xea, _t(prim.AV), _t(prim.BV))
    sim_x, _ = enforce_pbc(_t(simulation_cell.a), x)
    xee = sim_x[:, None, :] - sim_x[None, :, :]
    eye = torch.eye(n, dtype=x.dtype)
    see, xee_p = dist(xee + eye[..., None], _t(simulation_cell.AV), _t(simulation_cell.BV))
    see = see * (1.0 - eye)
    xee_p = xee_p * (1.0 - eye)[..., None]
    return xea, xee_p, sea[..., None], see[..., None]


# network.py:305-332
def construct_symmetric_features(h_one, h_two, spins):
    nu = spins[0]
    h_ones = [h_one[:nu], h_one[nu:]]
    h_twos = [h_two[:nu], h_two[nu:]]
    g_one = [h.mean(0, keepdim=True) for h in h_ones if h.shape[0] > 0]
    g_two = [h.mean(0) for h in h_twos if h.shape[0] > 0]
    g_one = [g.expand(h_one.shape[0], -1) for g in g_one]
    return torch.cat([h_one] + g_one + g_two, dim=1)


# network.py:335-343
def isotropic_envelope(ae, params):
    return torch.sum(torch.exp(-torch.abs(ae * params['sigma'])) * params['pi'], dim=1)


def diagonal_envelope(ae, params):
    r_ae = torch.linalg.norm(ae[..., None] * params['sigma'], dim=2)
    return torch.sum(torch.exp(-r_ae) * params['pi'], dim=1)


# network.py:346-364  (einsum form given in the reference docstring :350)
def full_envelope(ae, params):
    r_ae = torch.einsum('ijk,kmjn->ijmn', ae, params['sigma'])
    r_ae = torch.linalg.norm(r_ae, dim=2)
    return torch.sum(torch.exp(-r_ae) * params['pi'], dim=1)


# network.py:375-392
def slogdet_op(x):
    if x.shape[-1] == 1:
        v = x[..., 0, 0]
        return torch.exp(1j * torch.angle(v)), torch.log(torch.abs(v))
    sign, logdet = torch.linalg.slogdet(x)
    return sign, logdet


# network.py:395-427
def logdet_matmul(xs):
    slogdets = [slogdet_op(x) for x in xs]
    sign_in, slogdet = slogdets[0]
    for s, l in slogdets[1:]:
        sign_in, slogdet = sign_in * s, slogdet + l
    slogdet_max = torch.max(slogdet)      # = slogdet[argmax]; the result is shift invariant
    det = sign_in * torch.exp(slogdet - slogdet_max)
    result = det.sum()
    sign_out = torch.exp(1j * torch.angle(result))
    slog_out = torch.log(torch.abs(result)) + slogdet_max
    return sign_out, slog_out


# network.py:449-458
def eval_phase(x, klist, spins, full_det=False):
    x = x.reshape(-1, 3)
    xs = [x[:spins[0]], x[spins[0]:]]
    if full_det:
        kcat = torch.cat([_t(k) for k in klist], dim=0)
        kd = [xx @ kcat.T for xx, ne in zip(xs, spins) if ne > 0]
    else:
        kd = [xx @ _t(k).T for xx, k, ne in zip(xs, klist, spins) if ne > 0]
    return [torch.exp(1j * k) for k in kd]


# network.py:461-560
def solid_fermi_net_orbitals(params, x, simulation_cell, klist, atoms, spins,
                             envelope_type='isotropic', full_det=False, distance_type='nu'):
    ae_, ee_, r_ae, r_ee = construct_periodic_input_features(
        x, atoms, simulation_cell, distance_type)
    ae = torch.cat((r_ae, ae_), dim=2).reshape(ae_.shape[0], -1)
    ee = torch.cat((r_ee, ee_), dim=2)
    to_env = r_ae if envelope_type == 'isotropic' else ae_
    envelope = {'isotropic': isotropic_envelope, 'diagonal': diagonal_envelope,
                'full': full_envelope}[envelope_type]

    def residual(a, b):
        return (a + b) / math.sqrt(2.0) if a.shape == b.shape else b

    h_one, h_two = ae, ee
    nd = len(params['double'])
    for i in range(nd):
        h_in = construct_symmetric_features(h_one, h_two, spins)
        h_one_next = torch.tanh(h_in @ params['single'][i]['w'] + params['single'][i]['b'])
        h_two_next = torch.tanh(h_two @ params['double'][i]['w'] + params['double'][i]['b'])
        h_one = residual(h_one, h_one_next)
        h_two = residual(h_two, h_two_next)
    if nd != len(params['single']):
        h_in = construct_symmetric_features(h_one, h_two, spins)
        h_one_next = torch.tanh(h_in @ params['single'][-1]['w'] + params['single'][-1]['b'])
        h_one = residual(h_one, h_one_next)
        h_to_orb = h_one
    else:
        h_to_orb = construct_symmetric_features(h_one, h_two, spins)
    hs = [h_to_orb[:spins[0]], h_to_orb[spins[0]:]]
    active = [s for s in spins if s > 0]
    hs = [h for h, s in zip(hs, spins) if s > 0]
    orbitals = []
    for h, p in zip(hs, params['orbital']):
        o = h @ p['w']
        if 'b' in p:
            o = o + p['b']
        nparams = p['w'].shape[-1] // 2
        orbitals.append(o[..., :nparams] + 1j * o[..., nparams:])
    envs = []
    off = 0
    for s in active:
        envs.append(to_env[off:off + s])
        off += s
    orbitals = [envelope(te, pe) * orb for te, orb, pe in zip(envs, orbitals, params['envelope'])]
    ncol = sum(spins) if full_det else None
    orbitals = [orb.reshape(s, -1, ncol if full_det else s).permute(1, 0, 2)
                for s, orb in zip(active, orbitals)]
    phases = eval_phase(x, klist, spins, full_det)
    orbitals = [orb * p[None, :, :] for orb, p in zip(orbitals, phases)]
    if full_det:
        orbitals = [torch.cat(orbitals, dim=1)]
    return orbitals, to_env


# network.py:563-606
def eval_func(params, x, klist, simulation_cell, atoms, spins, envelope_type='full',
              full_det=False, distance_type='nu', method_name='eval_slogdet'):
    orbitals, _ = solid_fermi_net_orbitals(params, x, simulation_cell, klist, atoms, spins,
                                           envelope_type, full_det, distance_type)
    if method_name == 'eval_slogdet':
        return logdet_matmul(orbitals)[1]
    if method_name == 'eval_logdet':
        sign, slogdet = logdet_matmul(orbitals)
        return torch.log(sign) + slogdet
    if method_name == 'eval_phase_and_slogdet':
        return logdet_matmul(orbitals)
    if method_name == 'eval_mats':
        return orbitals
    raise ValueError('Unrecognized method name')


# network.py:60-186 (shapes only; the JAX threefry stream cannot be reproduced,
# numpy default_rng is used as SURVEY.md section 8(d) prescribes)
def init_solid_fermi_net_params(rng, atoms, spins, envelope_type='full', bias_orbitals=False,
                                use_last_layer=False, full_det=True,
                                hidden_dims=((256, 32), (256, 32), (256, 32)),
                                determinants=16, distance_type='nu'):
    natom = np.asarray(atoms).shape[0]
    if distance_type == 'nu':
        in_dims = (natom * 4, 4)
    elif distance_type == 'tri':
        in_dims = (natom * 7, 7)
    else:
        raise ValueError('Unrecognized distance function.')
    active = [s for s in spins if s > 0]
    nch = len(active)
    dims_one_in = ([(nch + 1) * in_dims[0] + nch * in_dims[1]] +
                   [(nch + 1) * h[0] + nch * h[1] for h in hidden_dims])
    if not use_last_layer:
        dims_one_in[-1] = hidden_dims[-1][0]
    dims_one_out = [h[0] for h in hidden_dims]
    dims_two = [in_dims[1]] + [h[1] for h in hidden_dims]
    len_double = len(hidden_dims) if use_last_layer else len(hidden_dims) - 1
    params = {'single': [], 'double': [], 'orbital': [], 'envelope': []}
    for s in active:
        nparam = sum(spins) * determinants if full_det else s * determinants
        env = {'pi': np.ones((natom, nparam))}
        if envelope_type == 'isotropic':
            env['sigma'] = np.ones((natom, nparam))
        elif envelope_type == 'diagonal':
            env['sigma'] = np.ones((natom, 3, nparam))
        elif envelope_type == 'full':
            env['sigma'] = np.tile(np.eye(3)[..., None, None], [1, 1, natom, nparam])
        params['envelope'].append(env)
    for i in range(len(hidden_dims)):
        params['single'].append({
            'w': rng.standard_normal((dims_one_in[i], dims_one_out[i])) / math.sqrt(dims_one_in[i]),
            'b': rng.standard_normal((dims_one_out[i],))})
        if i < len_double:
            params['double'].append({
                'w': rng.standard_normal((dims_two[i], dims_two[i + 1])) / math.sqrt(dims_two[i]),
                'b': rng.standard_normal((dims_two[i + 1],))})
    for s in active:
        nparam = sum(spins) * determinants if full_det else s * determinants
        p = {'w': rng.standard_normal((dims_one_in[-1], 2 * nparam)) / math.sqrt(dims_one_in[-1])}
        if bias_orbitals:
            p['b'] = rng.standard_normal((2 * nparam,))
        params['orbital'].append(p)
    return params


def params_to_torch(params, dtype=None):
    def conv(o):
        if isinstance(o, dict):
            return {k: conv(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return [conv(v) for v in o]
        return _t(o, dtype)
    return conv(params)


# network.py:609-667
def make_solid_fermi_net(envelope_type='full', bias_orbitals=False, use_last_layer=False,
                         klist=None, simulation_cell=None, full_det=True,
                         hidden_dims=((256, 32), (256, 32), (256, 32)), determinants=16,
                         after_determinants=1, distance_type='nu', method_name='eval_logdet'):
    if method_name not in ['eval_slogdet', 'eval_logdet', 'eval_mats', 'eval_phase_and_slogdet']:
        raise ValueError('Method name is not in class dir.')
    atoms_np = np.asarray(simulation_cell.original_cell.atom_coords())
    atoms = _t(atoms_np)
    spins = tuple(int(s) for s in simulation_cell.nelec)

    def init(rng, data=None):
        return init_solid_fermi_net_params(rng, atoms_np, spins, envelope_type, bias_orbitals,
                                           use_last_layer, full_det, hidden_dims, determinants,
                                           distance_type)

    def apply(params, x):
        return eval_func(params, x, klist, simulation_cell, atoms, spins, envelope_type,
                         full_det, distance_type, method_name)

    return SimpleNamespace(init=init, apply=apply)
