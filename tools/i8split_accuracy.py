"""CPU experiment (round 5): what an int8 error-free split of the per-electron contractions would do to E_kin.

The forward-Laplacian oracle (oracle/forward_laplacian.py) is run on fixture walkers with its two per-electron contractions
(hidden layers: network.py:517-533; orbital head: network.py:539-545) replaced by an emulation of the split product:

  * jets X[k][slot] of one electron tile: one power-of-two scale per slot column (max over the per-electron rows k), fixed point with
    F = 8 s - 1 fractional bits, balanced radix-256 digits (s int8 slices);
  * weights W[k][n]: one scale per output column n, same digits;
  * products of digit planes (i, j) kept for i + j <= s + 1 (1-based) -- s (s + 1) / 2 int8 MFMA passes, int32 accumulation exact;
  * the shared spin-mean rows (contracted once per walker by k_shared_term) stay in float64.

Prints max |dE_kin| against the unmodified oracle per slice count.  Usage: python tools/i8split_accuracy.py [case] [walkers]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
from common import load_case                      # noqa: E402
from oracle import forward_laplacian as ofl       # noqa: E402
from oracle import network as onet                # noqa: E402


def digits(v, s):
    """v: float64 array with |v| <= 0.5 -> list of s int64 digit planes (balanced radix 256), value = sum_i d_i 2^(-8 i + 1) ... exact
    fixed point with F = 8 s - 1 fractional bits."""
    F = 8 * s - 1
    assert F <= 62
    x = np.rint(np.ldexp(v, F)).astype(np.int64)
    out = []
    for _ in range(s):
        d = ((x + 128) % 256) - 128
        out.append(np.asarray(d, dtype=np.int64))
        x = (x - d) // 256
    assert np.all(np.asarray(x == 0)), 'top digit overflow'
    return out[::-1]            # most significant first: value = sum_i d_i 2^(8 (s - i) - F), i = 1..s


def split_product(X, W, s, drop=True):
    """X: (K, D) float64, W: (K, n) float64 -> (n, D) emulated split product W^T X."""
    cs = np.abs(X).max(0)
    cs = np.where(cs > 0, 2.0 ** (np.ceil(np.log2(np.where(cs > 0, cs, 1.0))) + 1), 1.0)      # |X / cs| <= 0.5
    ws = np.abs(W).max(0)
    ws = np.where(ws > 0, 2.0 ** (np.ceil(np.log2(np.where(ws > 0, ws, 1.0))) + 1), 1.0)
    dx = digits(X / cs, s)
    dw = digits(W / ws, s)
    F = 8 * s - 1
    acc = np.zeros((W.shape[1], X.shape[1]), dtype=np.longdouble)
    # groups from the least significant upward, each an exact integer matrix product
    for g in range(2 * s, 1, -1):
        if drop and g > s + 1:
            continue
        tot = np.zeros((W.shape[1], X.shape[1]), dtype=np.int64)
        for i in range(1, s + 1):
            j = g - i
            if 1 <= j <= s:
                tot += dw[i - 1].T @ dx[j - 1]
        acc += np.ldexp(tot.astype(np.longdouble), 8 * (2 * s - g) - 2 * F)
    return (acc * ws[:, None].astype(np.longdouble) * cs[None, :].astype(np.longdouble)).astype(np.float64)


class Patch:
    """torch.einsum replaced for the two per-electron contraction patterns of oracle.forward_laplacian.stages."""

    def __init__(self, s, n_local, n_pair, layers, drop=True):
        self.s, self.n_local, self.n_pair, self.layers, self.drop = s, n_local, n_pair, layers, drop
        self.orig = torch.einsum
        self.calls = 0

    def __call__(self, eq, *ops):
        if eq == '...kd,kn->...nd' and ops[0].dim() == 3 and ops[0].shape[1] > 100:
            h, w = ops
            self.calls += 1
            if self.calls not in self.layers:
                return self.orig(eq, *ops)
            K = h.shape[1]
            loc = np.r_[0:self.n_local, K - self.n_pair:K]               # per-electron rows: h_i and the pair-mean rows
            sh = np.r_[self.n_local:K - self.n_pair]                     # shared spin means: exact
            hn, wn = h.numpy(), w.numpy()
            z = np.einsum('ikd,kn->ind', hn[:, sh], wn[sh])
            for i in range(hn.shape[0]):
                z[i] += split_product(hn[i, loc], wn[loc], self.s, self.drop)
            return torch.from_numpy(z)
        if eq == 'ikd,kp->ipd' and 'orb' in self.layers:
            h, w = ops
            hn, wn = h.numpy(), w.numpy()
            z = np.stack([split_product(hn[i], wn, self.s, self.drop) for i in range(hn.shape[0])])
            return torch.from_numpy(z)
        return self.orig(eq, *ops)

    def __enter__(self):
        torch.einsum = self
        return self

    def __exit__(self, *a):
        torch.einsum = self.orig


def main():
    case = sys.argv[1] if len(sys.argv) > 1 else 'bcc_li'
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    fx, cell, klist, net_kw, params = load_case(case)
    p = onet.params_to_torch(params)
    h1 = net_kw['hidden_dims'][0][0]
    h2 = net_kw['hidden_dims'][0][1]
    x = torch.as_tensor(fx['x'][:nb], dtype=torch.float64)
    ref = [complex(ofl.stages(p, x[b], klist, cell, net_kw)['ke']) for b in range(nb)]
    print(f'{case}: {nb} walkers, E_kin = {[round(r.real, 6) for r in ref]}')
    nl = len(params['single'])
    hidden = set(range(1, nl))                       # counted calls = the hidden layers with K > 100 (layer 0 is not counted: stays fp64)
    for s in (4, 5, 6, 7):
        for what, layers in (('hidden', hidden), ('hidden+orb', hidden | {'orb'})):
            errs = []
            for b in range(nb):
                with Patch(s, h1, 2 * h2, layers):
                    e = complex(ofl.stages(p, x[b], klist, cell, net_kw)['ke'])
                errs.append(abs(e - ref[b]))
            print(f's = {s} ({s * (s + 1) // 2:2d} products) {what:11s}: max |dE_kin| = {max(errs):.3e} Ha')


if __name__ == '__main__':
    main()
