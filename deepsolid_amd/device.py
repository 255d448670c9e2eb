"""DeviceSystem: one simulation cell + network architecture resident on one GPU.

Owns the C-ABI handle (``ds_system``), the packed parameter buffer and the
scratch workspace; every numeric call of the package ends in one of the
``ds_*`` entry points of libdeepsolid_hip.so here.  Walkers, parameters and
results are torch tensors on the ROCm device -- torch is used for memory,
streams and (elsewhere) torch.distributed, nothing else.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

_DTYPES = {torch.float64: 0, torch.float32: 1}


def _pd(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError('deepsolid_amd needs a ROCm GPU: there is no CPU path (the CPU oracle lives in '
                           'oracle/ and is test infrastructure only)')


def device_plan(hidden_dims, n_in_single, n_in_double=4, n_double=None):
    """-> (device widths, residual flags) for the reference's `hidden_dims` (network.py:111-132), from the library itself
    (`ds_device_widths`: host code of the C ABI, no GPU needed; `ds_system_create` applies the same plan to its descriptor).

    The kernels run one-electron widths in multiples of 64 and pair widths of 16 or 32; any other width runs with ZERO-PADDED
    weights and biases, which is exact: a padded feature is tanh(0) = 0 in every layer, has zero jets, adds nothing to the spin
    means and meets zero rows in the next layer.  A residual is added exactly where the reference adds one (network.py:525-528:
    in == out of the REFERENCE's widths) -- an explicit flag per layer and stream, independent of the padded widths.
    `n_in_single` / `n_in_double`: widths of the input features (nf x atoms, nf); `n_double`: number of pair layers that run
    (network.py:118-121: the last width is unused without `use_last_layer`).  Raises ValueError for what the kernels cannot run:
    widths beyond 1024 / 32.  A first layer as wide as its input features (either stream: the reference's residual there) runs for
    any width: the residual is added behind the layer."""
    import ctypes as C
    lib = _lib.load()
    n = len(hidden_dims)
    n_double = n if n_double is None else n_double
    arr = C.c_int32 * max(n, 1)
    hs, hd = arr(*[int(h[0]) for h in hidden_dims]), arr(*[int(h[1]) for h in hidden_dims])
    ps, pd, rs, rd = arr(), arr(), arr(), arr()
    if lib.ds_device_widths(hs, hd, n, int(n_in_single), int(n_in_double), int(n_double), ps, pd, rs, rd) != 0:
        raise ValueError(lib.ds_last_error().decode())
    return tuple(zip(ps[:n], pd[:n])), tuple(zip((bool(v) for v in rs[:n]), (bool(v) for v in rd[:n])))


def device_widths(hidden_dims, n_in_single, n_double=None, n_in_double=4):
    """The widths the kernels run for the reference's `hidden_dims` (see `device_plan`)."""
    return device_plan(hidden_dims, n_in_single, n_in_double, n_double)[0]


class DeviceSystem:
    def __init__(self, simulation_cell, klist, net_kw, tables, dtype=torch.float64, device=None):
        _require_gpu()
        self.lib = _lib.load()
        self.cell = simulation_cell
        self.dtype = dtype
        self.device = torch.device(device if device is not None else f'cuda:{torch.cuda.current_device()}')
        self.net_kw = dict(net_kw)
        self.tables = tables
        prim = simulation_cell.original_cell
        self.nelec = tuple(int(n) for n in simulation_cell.nelec)
        if self.nelec[0] == 0 and self.nelec[1] > 0:
            # only spin-down electrons -- an EXTENSION beyond the reference: its parameter tree drops the empty channel
            # (network.py:113-117) but its forward then raises (network.py:537-553 pairs the empty block with orbital[0]).  The
            # parameter tree is that of the mirrored cell (n_dn, 0), which is what runs here; the library does the same swap for a
            # C caller.  Checked against this repo's oracle only, not against reference-executed numbers
            self.nelec = (self.nelec[1], 0)
            klist = (klist[1], klist[0])
        self.n = sum(self.nelec)
        self.n_det = int(net_kw['determinants'])
        self.hidden_dims = tuple(tuple(int(v) for v in h) for h in net_kw['hidden_dims'])
        nf = 4 if net_kw.get('distance_type', 'nu') == 'nu' else 7
        nf_in = nf * np.asarray(prim.atom_coords()).reshape(-1, 3).shape[0]
        # what the kernels run (zero-padded weights) and where a residual is added: the library's own plan; the descriptor below
        # carries the REFERENCE's widths, ds_system_create pads them the same way
        self.device_dims, self.residuals = device_plan(self.hidden_dims, nf_in, nf,
                                                       len(self.hidden_dims) - (0 if net_kw.get('use_last_layer', False) else 1))
        d = _lib.SystemDesc()
        keep = []                       # host arrays must outlive ds_system_create

        def arr(a):
            a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
            keep.append(a)
            return _pd(a)
        d.dtype = _DTYPES[dtype]
        d.n_up, d.n_dn = self.nelec
        atoms = np.asarray(prim.atom_coords(), dtype=np.float64).reshape(-1, 3)
        d.n_atoms_prim = atoms.shape[0]
        d.prim_atoms = arr(atoms)
        d.prim_a[:] = np.asarray(prim.a, dtype=np.float64).reshape(-1).tolist()
        d.sim_a[:] = np.asarray(simulation_cell.a, dtype=np.float64).reshape(-1).tolist()
        n_sym = np.asarray(simulation_cell.AV).shape[0]
        d.n_sym = n_sym
        for name, src in (('prim_AV', prim.AV), ('prim_BV', prim.BV), ('sim_AV', simulation_cell.AV),
                          ('sim_BV', simulation_cell.BV)):
            flat = np.zeros(_lib.DS_MAX_SYM * 3)
            flat[:3 * n_sym] = np.asarray(src, dtype=np.float64).reshape(-1)
            getattr(d, name)[:] = flat.tolist()
        d.n_layers = len(self.hidden_dims)
        for i, (a, b) in enumerate(self.hidden_dims):
            d.hidden_single[i], d.hidden_double[i] = a, b
        d.n_det = self.n_det
        d.distance_type = {'nu': 0, 'tri': 1}.get(net_kw.get('distance_type', 'nu'), 99)
        d.envelope_type = {'isotropic': 0, 'diagonal': 1, 'full': 2}.get(net_kw.get('envelope_type'), 99)
        d.full_det = int(bool(net_kw.get('full_det', False)))
        d.use_last_layer = int(bool(net_kw.get('use_last_layer', False)))
        d.bias_orbitals = int(bool(net_kw.get('bias_orbitals', False)))
        kl = [np.asarray(k, dtype=np.float64).reshape(-1, 3) for k in klist]
        if kl[0].shape[0] != self.nelec[0] or kl[1].shape[0] != self.nelec[1]:
            raise ValueError('klist must hold one k vector per electron of each spin (hf.py:99-104)')
        d.klist_up, d.klist_dn = arr(kl[0]), arr(kl[1] if kl[1].size else np.zeros((1, 3)))
        t = tables
        d.n_atoms_sim = t.atom_coords.shape[0]
        d.sim_atoms, d.sim_charges = arr(t.atom_coords), arr(t.atom_charges)
        d.dist_mode = t.dist_mode
        d.n_g = t.gpoints.shape[0]
        d.gpoints, d.gweight = arr(t.gpoints), arr(t.gweight)
        d.ion_exp_re, d.ion_exp_im = arr(t.ion_exp.real), arr(t.ion_exp.imag)
        d.ewald_alpha = float(t.alpha)
        d.ee_const, d.ei_const = float(t.ee_const(self.n)), float(t.ei_const(self.n))
        d.ii_total = float(t.ion_ion + t.ii_const)
        handle = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.ds_system_create(C.byref(d), C.byref(handle)), 'ds_system_create')
        self.handle = handle
        nb = self.lib.ds_param_layout(handle, None, 0)
        blocks = (_lib.ParamBlock * nb)()
        self.lib.ds_param_layout(handle, blocks, nb)
        self.blocks = [(int(b.offset), int(b.rows), int(b.cols)) for b in blocks]
        self.param_count = int(self.lib.ds_param_count(handle))
        self._ws = None
        self._packed = None
        self._packed_key = None
        self._packed_leaves = []

    # ------------------------------------------------------------------ construction helpers
    @classmethod
    def for_network(cls, simulation_cell, klist, net_kw, dtype=torch.float64):
        from .ewaldsum import EwaldTables
        key = ('net', id(simulation_cell), dtype, repr(sorted(net_kw.items())),
               tuple(np.asarray(k).tobytes() for k in klist))
        cache = simulation_cell.__dict__.setdefault('_ds_cache', {})
        if key not in cache:
            tables = cache.get('tables')
            if tables is None:
                tables = cache['tables'] = EwaldTables(simulation_cell)
            cache[key] = cls(simulation_cell, klist, net_kw, tables, dtype)
        return cache[key]

    @classmethod
    def for_ewald(cls, simulation_cell, tables=None, dtype=torch.float64):
        """Ewald-only handle: a dummy minimal network description."""
        from .ewaldsum import EwaldTables
        from .supercell import make_klist
        cache = simulation_cell.__dict__.setdefault('_ds_cache', {})
        key = ('ewald', dtype)
        if key not in cache:
            tables = tables or cache.get('tables') or EwaldTables(simulation_cell)
            cache['tables'] = tables
            nk = dict(envelope_type='isotropic', bias_orbitals=False, use_last_layer=False, full_det=False,
                      hidden_dims=((64, 16), (64, 16)), determinants=8, distance_type='nu')
            if getattr(simulation_cell, 'AV', None) is None:
                from .supercell import set_symmetry_lat
                set_symmetry_lat(simulation_cell)
            klist = make_klist(simulation_cell)
            cache[key] = cls(simulation_cell, klist, nk, tables, dtype)
        return cache[key]

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                self.lib.ds_system_destroy(self.handle)
        except Exception:
            pass

    # ------------------------------------------------------------------ buffers
    def _check_x(self, x):
        if not x.is_cuda:
            raise RuntimeError('walkers must be a tensor on the ROCm device (no CPU path)')
        if x.dtype != self.dtype:
            raise TypeError(f'walkers are {x.dtype}, system was built for {self.dtype}')
        if x.dim() != 2 or x.shape[1] != 3 * self.n:
            raise ValueError(f'walkers must be (B, {3 * self.n}), got {tuple(x.shape)}')
        return x.contiguous()

    def workspace(self, B, max_bytes=None):
        need = int(self.lib.ds_workspace_bytes(self.handle, int(B)))
        if max_bytes is not None:
            need = min(need, int(max_bytes))
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def pack_params(self, params):
        """Reference parameter tree (network.py:135-186) -> the flat device buffer of
        include/deepsolid_hip.h (`ds_param_layout`).  Cached on the leaves' versions."""
        leaves = []

        def walk(o):
            if isinstance(o, dict):
                for k in sorted(o):
                    walk(o[k])
            elif isinstance(o, (list, tuple)):
                for v in o:
                    walk(v)
            else:
                leaves.append(o)
        walk(params)
        # cache only trees of tensors: the key holds the leaves themselves (so ids cannot be recycled), their
        # storage address and version counter; numpy leaves have no version counter and are packed every call
        cacheable = all(isinstance(t, torch.Tensor) for t in leaves)
        key = tuple((t.data_ptr(), t._version) for t in leaves) if cacheable else None
        if cacheable and self._packed is not None and key == self._packed_key and \
                len(leaves) == len(self._packed_leaves) and all(a is b for a, b in zip(leaves, self._packed_leaves)):
            return self._packed

        def dev(a):
            if not isinstance(a, torch.Tensor):
                a = torch.as_tensor(np.asarray(a))
            return a.to(device=self.device, dtype=pdt)
        pdt = getattr(self, '_pack_dtype', None) or self.dtype     # _grad_index packs entry numbers in float64
        flat = torch.zeros(self.param_count, dtype=pdt, device=self.device)
        nch = 2 if self.nelec[1] > 0 else 1
        bi = iter(self.blocks)

        def put(mat):
            off, rows, cols = next(bi)
            mat = mat.reshape(rows, cols) if mat.numel() == rows * cols else mat
            if tuple(mat.shape) != (rows, cols):
                raise ValueError(f'parameter block shape {tuple(mat.shape)} != expected {(rows, cols)}')
            flat[off:off + rows * cols] = mat.reshape(-1)
        natom = np.asarray(self.cell.original_cell.atom_coords()).reshape(-1, 3).shape[0]
        nf = 4 if self.net_kw.get('distance_type', 'nu') == 'nu' else 7
        r4 = lambda v: (v + 3) // 4 * 4
        # reference widths (network.py:111-132) and the device widths (layer-0 rows padded to the MFMA k-step, hidden widths
        # to what the kernels run: device_widths)
        h1_ref = [nf * natom] + [h[0] for h in self.hidden_dims]
        h2_ref = [nf] + [h[1] for h in self.hidden_dims]
        h1 = [r4(nf * natom)] + [h[0] for h in self.device_dims]
        h2 = [r4(nf)] + [h[1] for h in self.device_dims]

        def pad_rows(m, rows):
            if m.shape[0] == rows:
                return m
            return torch.cat([m, torch.zeros(rows - m.shape[0], m.shape[1], dtype=m.dtype, device=m.device)], dim=0)

        def pad_cols(m, cols):
            if m.shape[-1] == cols:
                return m
            return torch.cat([m, torch.zeros(*m.shape[:-1], cols - m.shape[-1], dtype=m.dtype, device=m.device)], dim=-1)
        for l in range(len(self.hidden_dims)):
            w = dev(params['single'][l]['w'])
            kh, k2, khp, k2p = h1_ref[l], h2_ref[l], h1[l], h2[l]
            if w.shape[0] != (nch + 1) * kh + nch * k2 or w.shape[1] != h1_ref[l + 1]:
                raise ValueError(f"single[{l}]['w'] has shape {tuple(w.shape)}, expected {((nch + 1) * kh + nch * k2, h1_ref[l + 1])}")
            w = pad_cols(w, h1[l + 1])
            m2 = [pad_rows(w[(nch + 1) * kh + c * k2:(nch + 1) * kh + (c + 1) * k2], k2p) for c in range(nch)]
            put(torch.cat([pad_rows(w[:kh], khp)] + m2, dim=0))                       # per-electron rows [h | m2_up | m2_dn]
            put(torch.cat([pad_rows(w[(1 + c) * kh:(2 + c) * kh], khp) for c in range(nch)], dim=0))   # spin-mean rows
            put(pad_cols(dev(params['single'][l]['b']).reshape(1, -1), h1[l + 1]))
        use_last = bool(self.net_kw.get('use_last_layer', False))
        for l in range(len(self.hidden_dims) - (0 if use_last else 1)):
            put(pad_cols(pad_rows(dev(params['double'][l]['w']), h2[l]), h2[l + 1]))
            put(pad_cols(dev(params['double'][l]['b']).reshape(1, -1), h2[l + 1]))
        full_det = bool(self.net_kw.get('full_det', False))
        for c in range(nch):
            norb = self.n if full_det else self.nelec[c]
            nparam = norb * self.n_det
            w = dev(params['orbital'][c]['w'])
            if w.shape[1] != 2 * nparam:
                raise ValueError(f"orbital[{c}]['w'] has {w.shape[1]} columns, expected {2 * nparam}")
            def put_packed(mat):
                off, rows, cols = next(bi)
                src = self._orbital_column_map(nparam, cols)
                if mat.shape[0] != rows:
                    raise ValueError(f"orbital[{c}]['w'] block has {mat.shape[0]} rows, expected {rows}")
                packed = torch.zeros(rows, cols, dtype=pdt, device=self.device)
                valid = src >= 0
                packed[:, torch.as_tensor(np.nonzero(valid)[0], device=self.device)] = \
                    mat[:, torch.as_tensor(src[valid], device=self.device)]
                flat[off:off + rows * cols] = packed.reshape(-1)
            kh, k2, khp, k2p = h1_ref[-1], h2_ref[-1], h1[-1], h2[-1]
            if use_last:      # rows [h | mean_up | mean_dn | m2_up | m2_dn] (network.py:126-128) -> per-electron / shared
                m2 = [pad_rows(w[(nch + 1) * kh + cc * k2:(nch + 1) * kh + (cc + 1) * k2], k2p) for cc in range(nch)]
                put_packed(torch.cat([pad_rows(w[:kh], khp)] + m2, dim=0))
                put_packed(torch.cat([pad_rows(w[(1 + cc) * kh:(2 + cc) * kh], khp) for cc in range(nch)], dim=0))
            else:
                put_packed(pad_rows(w, khp))
            if self.net_kw.get('bias_orbitals', False):
                put(dev(params['orbital'][c]['b']))
            put(dev(params['envelope'][c]['pi']))
            # sigma: (A, P) isotropic | (A, 3, P) diagonal -> rows a*3+c | (3, 3, A, P) full -> rows (k*3+m)*A+a
            put(dev(params['envelope'][c]['sigma']).reshape(-1, nparam))
        self._packed, self._packed_key, self._packed_leaves = (flat, key, leaves) if cacheable else (None, None, [])
        return flat

    def _orbital_column_map(self, nparam, cols):
        """Packed column c -> reference column of orbital[s]['w'] (or -1 for zero padding).
        Within every 16-column tile t the MFMA accumulator gives lane group q = lane >> 4 the rows
        {q, q+4, q+8, q+12} (f64) or {4q .. 4q+3} (f32) in registers r = 0..3; they are assigned
        (Re p, Im p, Re p+4, Im p+4) with p = 8t + q, so the complex product with the envelope/phase jet
        is lane-local in the fused orbital epilogue (csrc/ds_gemm.h: orb_col)."""
        src = -np.ones(cols, dtype=np.int64)
        for t in range(cols // 16):
            for q in range(4):
                for r in range(4):
                    row = q + 4 * r if self.dtype == torch.float64 else 4 * q + r
                    p = 8 * t + q + 4 * (r // 2)
                    if p < nparam:
                        src[16 * t + row] = p + (r % 2) * nparam
        return src

    # ------------------------------------------------------------------ calls
    def ewald(self, x):
        x = self._check_x(x)
        out = torch.empty(x.shape[0], 3, dtype=self.dtype, device=self.device)
        if x.shape[0] == 0:
            return out
        _lib.check(self.lib.ds_ewald(self.handle, _ptr(x), x.shape[0], _ptr(out), _stream()), 'ds_ewald')
        return out

    def local_energy(self, params, x, want_logpsi=False, ws_bytes=None):
        x = self._check_x(x)
        B = x.shape[0]
        p = self.pack_params(params)
        ws = self.workspace(B, ws_bytes)
        ke = torch.empty(B, 2, dtype=self.dtype, device=self.device)
        ew = torch.empty(B, dtype=self.dtype, device=self.device)
        la = torch.empty(B, dtype=self.dtype, device=self.device) if want_logpsi else None
        ph = torch.empty(B, 2, dtype=self.dtype, device=self.device) if want_logpsi else None
        if B == 0:
            return ke, ew, la, ph
        _lib.check(self.lib.ds_local_energy(self.handle, _ptr(p), _ptr(x), B, _ptr(ke), _ptr(ew), _ptr(la), _ptr(ph),
                                            _ptr(ws), ws.numel(), _stream()), 'ds_local_energy')
        return ke, ew, la, ph

    def logpsi(self, params, x, ws_bytes=None):
        x = self._check_x(x)
        B = x.shape[0]
        p = self.pack_params(params)
        ws = self.workspace(B, ws_bytes)
        la = torch.empty(B, dtype=self.dtype, device=self.device)
        ph = torch.empty(B, 2, dtype=self.dtype, device=self.device)
        if B == 0:
            return la, ph
        _lib.check(self.lib.ds_logpsi(self.handle, _ptr(p), _ptr(x), B, _ptr(la), _ptr(ph), _ptr(ws), ws.numel(),
                                      _stream()), 'ds_logpsi')
        return la, ph

    def logpsi_grad(self, params, x):
        """-> (log|psi| (B,), grad (B, 3N) complex: Re = grad log|psi|, Im = grad arg psi)."""
        x = self._check_x(x)
        B = x.shape[0]
        p = self.pack_params(params)
        ws = self.workspace(B)
        la = torch.empty(B, dtype=self.dtype, device=self.device)
        gr = torch.empty(B, 3 * self.n, 2, dtype=self.dtype, device=self.device)
        if B:
            _lib.check(self.lib.ds_logpsi_grad(self.handle, _ptr(p), _ptr(x), B, _ptr(la), _ptr(None), _ptr(gr), _ptr(ws),
                                               ws.numel(), _stream()), 'ds_logpsi_grad')
        return la, torch.view_as_complex(gr)

    def logpsi_vjp(self, params, x, cot, max_bytes=None):
        """Packed parameter gradient of sum_b Re(conj(cot_b) * log psi_b), log psi = log|psi| + i arg psi
        (`ds_logpsi_vjp`): cot (B,) complex or (B, 2) real.  -> (flat grad (param_count,), log|psi| (B,),
        phase (B,) complex).  `unpack_grad` turns the flat vector into the reference's parameter tree."""
        x = self._check_x(x)
        B = x.shape[0]
        if cot.is_complex():
            cot = torch.view_as_real(cot)
        cot = cot.to(device=self.device, dtype=self.dtype).contiguous()
        if tuple(cot.shape) != (B, 2):
            raise ValueError(f'cot must be ({B},) complex or ({B}, 2), got {tuple(cot.shape)}')
        p = self.pack_params(params)
        if B == 0:
            return (torch.zeros(self.param_count, dtype=self.dtype, device=self.device),
                    torch.empty(0, dtype=self.dtype, device=self.device),
                    torch.empty(0, dtype=torch.complex128 if self.dtype == torch.float64 else torch.complex64, device=self.device))
        need = int(self.lib.ds_vjp_workspace_bytes(self.handle, int(B)))
        if need < 0:
            _lib.check(1, 'ds_vjp_workspace_bytes')
        if max_bytes is not None:
            need = min(need, int(max_bytes))      # fewer walker groups per pass (the library chunks the batch)
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        ws = self._ws[:need]
        grad = torch.empty(self.param_count, dtype=self.dtype, device=self.device)
        la = torch.empty(B, dtype=self.dtype, device=self.device)
        ph = torch.empty(B, 2, dtype=self.dtype, device=self.device)
        _lib.check(self.lib.ds_logpsi_vjp(self.handle, _ptr(p), _ptr(x), B, _ptr(cot), _ptr(grad), _ptr(la), _ptr(ph), _ptr(ws),
                                          ws.numel(), _stream()), 'ds_logpsi_vjp')
        return grad, la, torch.view_as_complex(ph)

    def _grad_index(self, params):
        """For every leaf of the parameter tree: positions of its entries in the packed buffer (the packing
        is a gather with zero padding, so its transpose is one index_select per leaf).  Found by packing a
        tree whose leaves hold their own running entry number."""
        shapes = []

        def walk(o):
            if isinstance(o, dict):
                return {k: walk(o[k]) for k in sorted(o)}
            if isinstance(o, (list, tuple)):
                return [walk(v) for v in o]
            t = torch.as_tensor(np.asarray(o)) if not isinstance(o, torch.Tensor) else o
            shapes.append(tuple(t.shape))
            return None
        walk(params)
        key = tuple(shapes)
        if getattr(self, '_gidx_key', None) == key:
            return self._gidx
        counter = [1]

        def number(o):
            if isinstance(o, dict):
                return {k: number(o[k]) for k in sorted(o)}
            if isinstance(o, (list, tuple)):
                return [number(v) for v in o]
            t = torch.as_tensor(np.asarray(o)) if not isinstance(o, torch.Tensor) else o
            n = t.numel()
            out = torch.arange(counter[0], counter[0] + n, dtype=torch.float64).reshape(t.shape)
            counter[0] += n
            return out
        numbered = number(params)
        saved = (self._packed, self._packed_key, self._packed_leaves)
        try:
            self._pack_dtype = torch.float64    # entry numbers must survive the packing exactly
            self._packed = None
            flat = self.pack_params(numbered).round().to(torch.int64)
        finally:
            self._pack_dtype = None
            self._packed, self._packed_key, self._packed_leaves = saved
        total = counter[0] - 1
        pos = torch.full((total + 1,), -1, dtype=torch.int64, device=self.device)
        nz = torch.nonzero(flat > 0).reshape(-1)
        pos[flat[nz]] = nz
        self._gidx, self._gidx_key = pos[1:], key
        return self._gidx

    def unpack_grad(self, flat, params):
        """Packed gradient -> a tree shaped like `params` (reference layout, network.py:135-186)."""
        pos = self._grad_index(params)
        if bool((pos < 0).any()):
            raise RuntimeError('parameter entries without a place in the packed buffer')
        vals = flat[pos]
        off = [0]

        def build(o):
            if isinstance(o, dict):
                return {k: build(o[k]) for k in sorted(o)}
            if isinstance(o, (list, tuple)):
                return [build(v) for v in o]
            t = torch.as_tensor(np.asarray(o)) if not isinstance(o, torch.Tensor) else o
            n = t.numel()
            out = vals[off[0]:off[0] + n].reshape(t.shape)
            off[0] += n
            return out
        return build(params)

    def orbitals(self, params, x):
        x = self._check_x(x)
        B = x.shape[0]
        p = self.pack_params(params)
        ws = self.workspace(B)
        if self.net_kw.get('full_det', False):
            sizes = [self.n, 0]
        else:
            sizes = list(self.nelec)
        outs = [torch.empty(B, self.n_det, ns, ns, 2, dtype=self.dtype, device=self.device) if ns else None for ns in sizes]
        if B == 0:
            return [torch.view_as_complex(o) for o in outs if o is not None]
        _lib.check(self.lib.ds_orbitals(self.handle, _ptr(p), _ptr(x), B, _ptr(outs[0]), _ptr(outs[1]), _ptr(ws),
                                        ws.numel(), _stream()), 'ds_orbitals')
        return [torch.view_as_complex(o) for o in outs if o is not None]

    def mh_propose(self, x1, normal, width):
        x1 = self._check_x(x1)
        normal = self._check_x(normal)
        x2 = torch.empty_like(x1)
        _lib.check(self.lib.ds_mh_propose(self.handle, _ptr(x1), _ptr(normal), float(width), x1.shape[0], _ptr(x2),
                                          _stream()), 'ds_mh_propose')
        return x2

    def mh_accept(self, x1, lp1, x2, lp2, uniform, n_accept):
        """In-place select on x1 / lp1; n_accept (1,) is incremented."""
        _lib.check(self.lib.ds_mh_accept(self.handle, _ptr(x1), _ptr(lp1), _ptr(x2), _ptr(lp2), _ptr(uniform),
                                         x1.shape[0], _ptr(n_accept), _stream()), 'ds_mh_accept')

    def mh_propose_ex(self, mode, x1, normal, width, aux, scratch=None):
        """`ds_mh_propose_ex`: mode 1 asymmetric (aux = nuclei (A,3)), mode 2 drift (aux = grad log|psi| (B,3N))."""
        x1 = self._check_x(x1)
        normal = self._check_x(normal)
        aux = aux.to(device=self.device, dtype=self.dtype).contiguous()
        x2 = torch.empty_like(x1)
        n_aux = aux.shape[0] if mode == 1 else 0
        _lib.check(self.lib.ds_mh_propose_ex(self.handle, int(mode), _ptr(x1), _ptr(normal), float(width), _ptr(aux), int(n_aux),
                                             x1.shape[0], _ptr(x2), _ptr(scratch), _stream()), 'ds_mh_propose_ex')
        return x2

    def mh_accept_ex(self, mode, x1, lp1, x2, logabs2, uniform, normal, width, aux1, aux2, n_accept, scratch=None):
        """`ds_mh_accept_ex`: in-place select on x1 / lp1; n_accept (1,) is incremented."""
        cv = lambda t: None if t is None else t.to(device=self.device, dtype=self.dtype).contiguous()
        aux1, aux2, normal = cv(aux1), cv(aux2), cv(normal)
        n_aux = aux1.shape[0] if mode == 1 else 0
        _lib.check(self.lib.ds_mh_accept_ex(self.handle, int(mode), _ptr(x1), _ptr(lp1), _ptr(x2), _ptr(cv(logabs2)), _ptr(cv(uniform)),
                                            _ptr(normal), float(width), _ptr(aux1), _ptr(aux2), int(n_aux), x1.shape[0],
                                            _ptr(n_accept), _ptr(scratch), _stream()), 'ds_mh_accept_ex')

    def mcmc_step(self, params, x, lp, steps, width, seed=0, offset=0, normals=None, uniforms=None, lp_valid=False,
                  n_accept=None, first_electron=None, importance=False, atoms=None):
        """`ds_mcmc_step`: `steps` all-electron Metropolis moves on x (B,3N) / lp (B,) IN PLACE, enqueued without a host
        synchronisation.  Noise from the in-kernel Philox stream (seed, offset) or replayed from `normals`
        (steps,B,3N) / `uniforms` (steps,B).  -> n_accept (1,) device tensor (incremented).
        `first_electron` = e: `ds_mcmc_step_one_electron` instead -- move i displaces electron (e + i) % N only
        (explicit normals are then (steps,B,3)).  `importance=True`: `ds_mcmc_step_importance`, the drift-biased move.
        `atoms` (A,3): `ds_mcmc_step_asymmetric`, step widths scaled by the harmonic mean of the nuclear distances."""
        x = self._check_x(x)
        B = x.shape[0]
        p = self.pack_params(params)
        if n_accept is None:
            n_accept = torch.zeros(1, dtype=self.dtype, device=self.device)
        if B == 0 or steps == 0 and lp_valid:
            return n_accept
        if normals is not None:
            normals = normals.to(device=self.device, dtype=self.dtype).contiguous()
            uniforms = uniforms.to(device=self.device, dtype=self.dtype).contiguous()
            width3 = 3 * self.n if first_electron is None else 3
            if tuple(normals.shape) != (steps, B, width3) or tuple(uniforms.shape) != (steps, B):
                raise ValueError('explicit noise must be normals (steps, B, 3N) -- (steps, B, 3) for one-electron moves -- '
                                 'and uniforms (steps, B)')
        need = int(self.lib.ds_mcmc_workspace_bytes(self.handle, int(B)))
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        if atoms is not None:
            atoms = torch.as_tensor(atoms, dtype=self.dtype, device=self.device).reshape(-1, 3).contiguous()
            _lib.check(self.lib.ds_mcmc_step_asymmetric(
                self.handle, _ptr(p), _ptr(x), _ptr(lp), B, int(steps), float(width), _ptr(atoms), int(atoms.shape[0]),
                int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 64 - 1), _ptr(normals), _ptr(uniforms), int(bool(lp_valid)),
                _ptr(n_accept), _ptr(self._ws), self._ws.numel(), _stream()), 'ds_mcmc_step_asymmetric')
            return n_accept
        if importance:
            _lib.check(self.lib.ds_mcmc_step_importance(
                self.handle, _ptr(p), _ptr(x), _ptr(lp), B, int(steps), float(width), int(seed) & (2 ** 64 - 1),
                int(offset) & (2 ** 64 - 1), _ptr(normals), _ptr(uniforms), int(bool(lp_valid)), _ptr(n_accept), _ptr(self._ws),
                self._ws.numel(), _stream()), 'ds_mcmc_step_importance')
            return n_accept
        if first_electron is not None:
            _lib.check(self.lib.ds_mcmc_step_one_electron(
                self.handle, _ptr(p), _ptr(x), _ptr(lp), B, int(steps), int(first_electron), float(width), int(seed) & (2 ** 64 - 1),
                int(offset) & (2 ** 64 - 1), _ptr(normals), _ptr(uniforms), int(bool(lp_valid)), _ptr(n_accept), _ptr(self._ws),
                self._ws.numel(), _stream()), 'ds_mcmc_step_one_electron')
            return n_accept
        _lib.check(self.lib.ds_mcmc_step(self.handle, _ptr(p), _ptr(x), _ptr(lp), B, int(steps), float(width),
                                         int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 64 - 1), _ptr(normals), _ptr(uniforms),
                                         int(bool(lp_valid)), _ptr(n_accept), _ptr(self._ws), self._ws.numel(), _stream()),
                   'ds_mcmc_step')
        return n_accept

    def energy_stats(self, ke, ew):
        """`ds_energy_stats`: (8,) float64 = [sum Re E_L, sum Im E_L, sum |E_L|^2, n, n_nonfinite, sum Re ke, sum Im ke, sum ew]."""
        out = torch.empty(8, dtype=torch.float64, device=self.device)
        _lib.check(self.lib.ds_energy_stats(self.handle, _ptr(ke), _ptr(ew), ke.shape[0], _ptr(out), _stream()), 'ds_energy_stats')
        return out

    def profile(self, on=True, only=None):
        """Start (and reset) / stop per-kernel HIP-event timing inside the library; `only` = one
        kernel kind of _lib.PROF_KINDS (events around that kernel only)."""
        mode = 0 if not on else (1 if only is None else 2 + _lib.PROF_KINDS.index(only))
        _lib.check(self.lib.ds_profile_enable(self.handle, mode), 'ds_profile_enable')

    def profile_read(self):
        """-> {kernel kind: (total ms, launches)} for the launches since profile(True)."""
        n = len(_lib.PROF_KINDS)
        ms = (C.c_double * n)()
        cnt = (C.c_int64 * n)()
        _lib.check(self.lib.ds_profile_read(self.handle, ms, cnt), 'ds_profile_read')
        return {k: (float(ms[i]), int(cnt[i])) for i, k in enumerate(_lib.PROF_KINDS)}

    def int8_layers(self):
        """Number of dense hidden layers per local-energy evaluation whose per-electron contraction runs as an int8 split
        (csrc/ds_i8.h; 0 unless the handle was created with DS_I8=1, or for shapes without an instance)."""
        return int(self.lib.ds_int8_layers(self.handle))

    def profile_clock(self):
        """In-kernel clock probe of the hidden-layer kernel since profile(True): -> (shader cycles, 100 MHz reference ticks,
        GHz) summed over its workgroups; GHz is None when the probe did not run."""
        cyc, ticks = C.c_double(0), C.c_double(0)
        _lib.check(self.lib.ds_profile_read_clock(self.handle, C.byref(cyc), C.byref(ticks)), 'ds_profile_read_clock')
        return cyc.value, ticks.value, (cyc.value / ticks.value * 0.1 if ticks.value > 0 else None)

    def debug_stage(self, params, x, stage, n_elems):
        x = self._check_x(x)
        p = self.pack_params(params)
        ws = self.workspace(x.shape[0])
        out = torch.zeros(int(n_elems), dtype=self.dtype, device=self.device)
        n = self.lib.ds_debug_stage(self.handle, _ptr(p), _ptr(x), x.shape[0], stage.encode(), _ptr(out), out.numel(),
                                    _ptr(ws), ws.numel(), _stream())
        if n < 0:
            raise RuntimeError(f'ds_debug_stage({stage}): {self.lib.ds_last_error().decode()}')
        return out[:n]
