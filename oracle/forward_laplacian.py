"""Oracle (test infrastructure): forward-Laplacian restatement of the local
kinetic energy -- the ALGORITHM the HIP kernels implement, in torch float64.

This is not in the reference.  The reference obtains
    E_kin = -1/2 sum_j [d_j^2 f + (d_j f)^2],  f = log psi   (hamiltonian.py:45-70)
from 2*3N forward-over-reverse Hessian-vector products.  Here the triple
(value, gradient wrt all 3N coordinates, Laplacian) is pushed forward through
network.py:461-560 and through the determinants in closed form:

    linear      y = xW + b      : every slot is mapped by W (bias on the value slot)
    tanh        y = tanh(z)     : y_d = y' z_d ;  y_L = y' z_L + y'' sum_d z_d^2
    product     (uv)_d = u_d v + u v_d ;  (uv)_L = u_L v + u v_L + 2 sum_d u_d v_d
    log det     d_d log det M = tr(M^-1 d_d M)
                lap log det M = tr(M^-1 lap M) - sum_d tr((M^-1 d_d M)^2)
    multi-det   lap psi / psi = sum_k w_k [ lap log D_k + |grad log D_k|^2 ],  w_k = D_k / psi

It is validated against the `for`-mode restatement (tests/test_oracle_golden.py)
and then used to check large batches and every intermediate stage of the HIP
path.  Slot layout of every "jet" tensor (last axis, length D = 3N + 2):
    slot 0 = value, slot 1 = Laplacian, slot 2 + 3*j + c = d / d x_{j,c}.
"""
import math

import torch
from torch.func import hessian, jacrev, vmap

from . import network as onet
from .network import _t


def _jet5(fun, r):
    """fun: (3,) -> (m,).  Returns value (.., m), grad (.., m, 3), lap (.., m) at every row of r."""
    flat = r.reshape(-1, 3)
    val = vmap(fun)(flat)
    jac = vmap(jacrev(fun))(flat)
    hes = vmap(hessian(fun))(flat)
    lap = hes.diagonal(dim1=-2, dim2=-1).sum(-1)
    shp = r.shape[:-1]
    return val.reshape(*shp, -1), jac.reshape(*shp, -1, 3), lap.reshape(*shp, -1)


def _tanh_jet(z):
    """z: (..., D) jets of the pre-activation -> jets of tanh(z)."""
    y = torch.tanh(z[..., 0])
    d1 = 1 - y * y
    d2 = -2 * y * d1
    out = d1[..., None] * z
    out[..., 0] = y
    out[..., 1] = d1 * z[..., 1] + d2 * (z[..., 2:] ** 2).sum(-1)
    return out


def stages(params, x, klist, simulation_cell, net_kw):
    """Forward-Laplacian through the network for ONE walker.  Returns a dict of
    every intermediate jet tensor (used for stage-by-stage HIP debugging)."""
    if net_kw.get('full_det', False) or net_kw.get('envelope_type', 'isotropic') != 'isotropic' or net_kw.get('bias_orbitals', False):
        raise NotImplementedError('forward-Laplacian oracle covers the tested default: isotropic, block-diagonal dets, no orbital bias')
    dist = {'nu': onet.nu_distance, 'tri': onet.tri_distance}[net_kw.get('distance_type', 'nu')]
    prim = simulation_cell.original_cell
    spins = tuple(int(s) for s in simulation_cell.nelec)
    atoms = _t(prim.atom_coords())
    x = _t(x).reshape(-1, 3)
    N, A = x.shape[0], atoms.shape[0]
    D = 3 * N + 2
    out = {}

    # --- input features (network.py:249-302) as 5-jets in the relative vector -------------
    prim_x, _ = onet.enforce_pbc(_t(prim.a), x)
    sim_x, _ = onet.enforce_pbc(_t(simulation_cell.a), x)
    AVp, BVp, AVs, BVs = _t(prim.AV), _t(prim.BV), _t(simulation_cell.AV), _t(simulation_cell.BV)

    def feat(a, b):
        def f(r):
            sd, rel = dist(r, a, b)
            return torch.cat([sd[None], rel])
        return f
    nf = 4 if net_kw.get('distance_type', 'nu') == 'nu' else 7
    r_ea = prim_x[:, None, :] - atoms                                    # (N,A,3)
    v, g, l = _jet5(feat(AVp, BVp), r_ea)                                # (N,A,nf) (N,A,nf,3) (N,A,nf)
    h1 = torch.zeros(N, A * nf, D, dtype=x.dtype)
    h1[:, :, 0] = v.reshape(N, -1)
    h1[:, :, 1] = l.reshape(N, -1)
    for i in range(N):
        h1[i, :, 2 + 3 * i:5 + 3 * i] = g[i].reshape(-1, 3)
    out['sd_ea_jet'] = (v[..., 0], g[..., 0, :], l[..., 0])              # envelope input
    eye = torch.eye(N, dtype=x.dtype)
    r_ee = sim_x[:, None, :] - sim_x[None, :, :] + eye[..., None]        # network.py:294
    v, g, l = _jet5(feat(AVs, BVs), r_ee)
    mask = (1.0 - eye)
    v, g, l = v * mask[..., None], g * mask[..., None, None], l * mask[..., None]
    h2 = torch.zeros(N, N, nf, D, dtype=x.dtype)
    h2[..., 0] = v
    h2[..., 1] = 2 * l                                                   # lap over x_i and x_j
    for i in range(N):
        for j in range(N):
            if i != j:
                h2[i, j, :, 2 + 3 * i:5 + 3 * i] = g[i, j]
                h2[i, j, :, 2 + 3 * j:5 + 3 * j] = -g[i, j]
    out['h1_0'], out['h2_0'] = h1, h2

    # --- equivariant layers (network.py:305-332, 517-533); everything linear acts on all slots
    def sym(h1, h2):
        nu = spins[0]
        parts = [h1]
        for sl in (slice(0, nu), slice(nu, N)):
            if sl.stop - sl.start > 0:
                parts.append(h1[sl].mean(0, keepdim=True).expand(N, -1, -1))
        for sl in (slice(0, nu), slice(nu, N)):
            if sl.stop - sl.start > 0:
                parts.append(h2[sl].mean(0))
        return torch.cat(parts, dim=1)

    def lin(h, w, b):
        z = torch.einsum('...kd,kn->...nd', h, w)
        z[..., 0] = z[..., 0] + b
        return z

    def res(a, b):
        return (a + b) / math.sqrt(2.0) if a.shape == b.shape else b

    nd = len(params['double'])
    for i in range(nd):
        g_in = sym(h1, h2)
        h1n = _tanh_jet(lin(g_in, params['single'][i]['w'], params['single'][i]['b']))
        h2n = _tanh_jet(lin(h2, params['double'][i]['w'], params['double'][i]['b']))
        h1, h2 = res(h1, h1n), res(h2, h2n)
        out[f'h1_{i + 1}'], out[f'h2_{i + 1}'] = h1, h2
    if nd != len(params['single']):
        g_in = sym(h1, h2)
        h1 = res(h1, _tanh_jet(lin(g_in, params['single'][-1]['w'], params['single'][-1]['b'])))
        h_orb = h1
        out[f'h1_{nd + 1}'] = h1
    else:
        h_orb = sym(h1, h2)

    # --- orbitals (network.py:539-557): phi * envelope * Bloch phase, product rule --------
    sd_v, sd_g, sd_l = out['sd_ea_jet']
    mats = []
    off = 0
    ch = 0
    for s, ns in enumerate(spins):
        if ns == 0:
            continue
        w = params['orbital'][ch]['w']
        nparam = w.shape[1] // 2
        o = torch.einsum('ikd,kp->ipd', h_orb[off:off + ns], w)
        phi = torch.complex(o[:, :nparam], o[:, nparam:])               # (ns, nparam, D)
        pi_, sg_ = params['envelope'][ch]['pi'], params['envelope'][ch]['sigma']   # (A, nparam)
        kpts = _t(klist[s])                                              # (ns, 3)
        M = torch.zeros(ns, nparam, D, dtype=onet.cdtype())
        for ii in range(ns):
            i = off + ii
            # envelope e[p] = sum_a pi exp(-|sd sigma|): 5-jet in x_i
            u = sd_v[i][:, None] * sg_                                   # (A, nparam)
            sgn = torch.sign(u)
            ex = torch.exp(-torch.abs(u)) * pi_
            de = -sgn * sg_ * ex                                         # d/d sd
            d2e = sg_ * sg_ * ex
            e_v = ex.sum(0)
            e_g = torch.einsum('ap,ac->pc', de, sd_g[i])                 # (nparam,3)
            e_l = (de * sd_l[i][:, None]).sum(0) + (d2e * (sd_g[i] ** 2).sum(-1)[:, None]).sum(0)
            # phase exp(i k_m . x_i) with the UNWRAPPED x (network.py:555)
            kd = kpts @ x[i]
            ph = torch.exp(1j * kd).repeat(nparam // ns)                 # p = k*ns + m
            kk = kpts.repeat(nparam // ns, 1).to(torch.complex128)       # (nparam,3)
            ph_g = 1j * kk * ph[:, None]
            ph_l = -(kk * kk).sum(-1) * ph
            q_v = e_v * ph
            q_g = e_g * ph[:, None] + e_v[:, None] * ph_g
            q_l = e_l * ph + e_v * ph_l + 2 * (e_g * ph_g).sum(-1)
            m = phi[ii] * q_v[:, None]
            own = slice(2 + 3 * i, 5 + 3 * i)
            m[:, own] = m[:, own] + phi[ii, :, 0:1] * q_g
            m[:, 1] = phi[ii, :, 1] * q_v + phi[ii, :, 0] * q_l + 2 * (phi[ii, :, own] * q_g).sum(-1)
            M[ii] = m
        # (ns, K*ns, D) -> (K, ns_elec, ns_orb, D)   (network.py:552-554)
        mats.append(M.reshape(ns, nparam // ns, ns, D).permute(1, 0, 2, 3))
        off += ns
        ch += 1
    out['mats'] = mats

    # --- determinants: log D_k, grad, lap  (closed form above) ---------------------------
    K = mats[0].shape[0]
    logD = torch.zeros(K, dtype=onet.cdtype())
    grad = torch.zeros(K, 3 * N, dtype=onet.cdtype())
    lap = torch.zeros(K, dtype=onet.cdtype())
    for M in mats:
        M0 = M[..., 0]
        sign, la = torch.linalg.slogdet(M0)
        logD = logD + la + torch.log(sign)
        Minv = torch.linalg.inv(M0)
        Y = torch.einsum('kab,kbcd->kacd', Minv, M[..., 2:])             # M^-1 d_dM
        grad = grad + torch.einsum('kaad->kd', Y)
        lap = lap + torch.einsum('kab,kba->k', Minv, M[..., 1]) - torch.einsum('kabd,kbad->k', Y, Y)
    out['logD'], out['gradD'], out['lapD'] = logD, grad, lap
    mx = logD.real.max()
    wk = torch.exp(logD - mx)
    S = wk.sum()
    wk = wk / S
    out['logabs'] = torch.log(torch.abs(S)) + mx
    out['phase'] = S / torch.abs(S)
    out['grad_f'] = (wk[:, None] * grad).sum(0)
    out['ke'] = -0.5 * (wk * (lap + (grad * grad).sum(-1))).sum()
    return out


def local_kinetic_energy_forward_laplacian(klist, simulation_cell, net_kw):
    def ke(params, x):
        return stages(params, x, klist, simulation_cell, net_kw)['ke']
    return ke
