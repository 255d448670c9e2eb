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
