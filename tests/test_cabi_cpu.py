"""CPU checks of the C-ABI boundary: the library builds/loads and exports every symbol
include/deepsolid_hip.h declares; the product path refuses to run without a GPU."""
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, 'include', 'deepsolid_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(ds_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    from deepsolid_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), n
        assert n in _lib.SIGNATURES, f'{n} declared in the header but not bound in _lib.py'
    assert sorted(_lib.SIGNATURES) == names


def test_struct_matches_header_field_order():
    from deepsolid_amd import _lib
    src = open(os.path.join(ROOT, 'include', 'deepsolid_hip.h')).read()
    body = src[src.index('typedef struct ds_system_desc {'):src.index('} ds_system_desc;')]
    body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
    fields = []
    body = body.split('{', 1)[1]
    for decl in body.split(';'):
        decl = decl.strip()
        if not decl:
            continue
        for part in decl.split(','):
            m = re.search(r'([A-Za-z_][A-Za-z0-9_]*)\s*(\[[^\]]*\])?\s*$', part.strip())
            assert m, part
            fields.append(m.group(1))
    assert fields == [f[0] for f in _lib.SystemDesc._fields_]


@pytest.mark.skipif(torch.cuda.is_available(), reason='CPU-only behaviour')
def test_product_path_fails_loudly_without_gpu():
    from deepsolid_amd import hamiltonian, network, systems
    cell, klist = systems.build('lih')
    net = network.make_solid_fermi_net(klist=klist, simulation_cell=cell, method_name='eval_logdet',
                                       **systems.DETNET_DEFAULTS)
    params = net.init(0)
    x = torch.as_tensor(systems.synthetic_walkers(cell, 2))
    with pytest.raises(RuntimeError, match='GPU'):
        net.apply(params, x)
    with pytest.raises(RuntimeError, match='GPU'):
        hamiltonian.local_energy_seperate(net.apply, cell)(params, x)


def test_param_tree_shapes_match_reference_init():
    """network.py:126-186 shapes for the bcc-Li 24-electron default network."""
    from deepsolid_amd import network, systems
    cell, klist = systems.build('bcc_li')
    net = network.make_solid_fermi_net(klist=klist, simulation_cell=cell, **systems.DETNET_DEFAULTS)
    p = net.init(3)
    assert [tuple(l['w'].shape) for l in p['single']] == [(20, 256), (832, 256), (832, 256)]
    assert [tuple(l['w'].shape) for l in p['double']] == [(4, 32), (32, 32)]
    assert [tuple(l['w'].shape) for l in p['orbital']] == [(256, 192), (256, 192)]
    assert [tuple(l['pi'].shape) for l in p['envelope']] == [(1, 96), (1, 96)]
    n = sum(int(np.prod(t.shape)) for grp in p.values() for d in grp for t in d.values())
    assert n == 531776
    with pytest.raises(ValueError):
        network.make_solid_fermi_net(klist=klist, simulation_cell=cell, method_name='nope')


def test_philox_known_answers_and_moments():
    """ds_mcmc_step's generator is Philox4x32-10 (Salmon et al., SC'11): the three known-answer vectors of the
    Random123 distribution (kat_vectors: zeros, ones, digits of pi), and the uniform / Box-Muller maps the kernels
    apply to the raw words (csrc/ds_mcmc.h) have the right first moments."""
    import ctypes as C
    from deepsolid_amd import _lib
    lib = _lib.load()
    out = (C.c_uint32 * 4)()

    def block(c, k):            # counter words c0..c3, key words k0,k1 through the (seed, offset, index, stream) interface
        assert c[3] < 2 ** 30
        lib.ds_philox_host(k[0] | (k[1] << 32), c[2] | (c[3] << 32), 0, c[0] | (c[1] << 32), 0, out)
        return [int(v) for v in out]
    assert block([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert block([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    # the offset and the step add; the stream bits separate the three draws of one (seed, offset, index)
    lib.ds_philox_host(5, 7, 3, 11, 0, out); a = list(out)
    lib.ds_philox_host(5, 10, 0, 11, 0, out); assert list(out) == a
    lib.ds_philox_host(5, 10, 0, 11, 2, out); assert list(out) != a
    n = 20000
    words = np.zeros((n, 4), dtype=np.uint64)
    for i in range(n):
        lib.ds_philox_host(1234, 0, 0, i, 0, out)
        words[i] = list(out)
    to53 = lambda a, b: ((a << np.uint64(21)) ^ (b >> np.uint64(11))) & np.uint64(2 ** 53 - 1)
    u1 = (to53(words[:, 0], words[:, 1]).astype(np.float64) + 1) / 2.0 ** 53
    u2 = to53(words[:, 2], words[:, 3]).astype(np.float64) / 2.0 ** 53
    assert 0 < u1.min() and u1.max() <= 1 and 0 <= u2.min() and u2.max() < 1
    z = np.concatenate([np.sqrt(-2 * np.log(u1)) * np.cos(2 * np.pi * u2), np.sqrt(-2 * np.log(u1)) * np.sin(2 * np.pi * u2)])
    assert abs(z.mean()) < 0.02 and abs(z.var() - 1) < 0.03 and abs((z ** 4).mean() - 3) < 0.15
    assert abs(u2.mean() - 0.5) < 0.01


def test_shape_limits_are_rejected_at_create():
    """Every shape the kernels cannot run is refused by ds_system_create itself (check_arch runs before any device
    allocation, so this needs no GPU): a handle that was created never fails at its first launch.  The supported set is
    documented in DESIGN.md section 1: determinant matrices up to 64 x 64 (hence N <= 128 electrons, or 64 with full_det:
    jet-slot tiles 1..25 all have kernel instances), the reference's hidden_dims verbatim with one-electron widths 1..1024 and pair
    widths 1..32 (padded inside the library), up to 64 determinants."""
    import ctypes as C
    from deepsolid_amd import _lib
    lib = _lib.load()

    def desc(**kw):
        d = _lib.SystemDesc()
        d.dtype, d.n_up, d.n_dn, d.n_atoms_prim, d.n_sym, d.n_layers, d.n_det = 0, 2, 2, 1, 3, 3, 8
        for i in range(3):
            d.hidden_single[i], d.hidden_double[i] = 256, 32
        for k, v in kw.items():
            if isinstance(v, (list, tuple)):
                for i, x in enumerate(v):
                    getattr(d, k)[i] = x
            else:
                setattr(d, k, v)
        return d

    def err(**kw):
        h = C.c_void_p()
        rc = lib.ds_system_create(C.byref(desc(**kw)), C.byref(h))
        assert rc != 0, kw
        return lib.ds_last_error().decode()

    assert 'larger than 64' in err(n_up=65, n_dn=0)
    assert 'larger than 64' in err(n_up=40, n_dn=30, full_det=1)
    # N = 60 (12 jet-slot tiles), 81, 108, 128: every electron count whose matrices fit has kernel instances since round 4
    # (1..25 slot tiles, csrc/ds_tiles.h) -- these descriptors pass every shape check and are refused only for their missing arrays
    for n_up, n_dn in ((30, 30), (41, 40), (54, 54), (64, 64)):
        assert 'null array' in err(n_up=n_up, n_dn=n_dn)
    # the descriptor carries the REFERENCE's widths: 200 or 24 are fine (zero-padded inside), refused are widths beyond the kernels'
    # range; a first layer as wide as its input features (one-electron stream: nf x atoms, pair stream: nf -- the reference's
    # residual there, network.py:525-528) runs since round 6 (any width: the residual is added behind the layer)
    assert 'null array' in err(hidden_single=[256, 200, 256], hidden_double=[32, 24, 32])
    assert 'hidden_single' in err(hidden_single=[2048, 256, 256])
    assert 'hidden_single' in err(hidden_single=[256, 0, 256])
    assert 'hidden_double' in err(hidden_double=[32, 40, 32])
    assert 'null array' in err(hidden_single=[4, 256, 256])          # n_atoms_prim = 1, 'nu': 4 input features: residual at layer 0, accepted
    assert 'null array' in err(hidden_double=[4, 32, 32])
    # 16 atoms, 'nu': 64 input features = hidden_single[0] -> residual at layer 0 with K = 64 + 2 x 4 = 72 per-electron rows: accepted
    # (round 5 created the handle and failed at the first launch)
    assert 'null array' in err(n_atoms_prim=16, hidden_single=[64, 256, 256])
    assert 'null array' in err(n_atoms_prim=64, distance_type=1, hidden_single=[448, 256, 256], n_up=4, n_dn=0)
    assert 'n_det' in err(n_det=65)
    assert 'null array' in err(n_up=0, n_dn=4)                     # a spin-down-only cell runs as its mirror image (round 5)
    assert 'n_up' in err(n_up=0, n_dn=0)
    assert 'n_dn' in err(n_dn=-1)
    assert 'tri' in err(distance_type=1, envelope_type=2)
