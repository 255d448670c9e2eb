"""A torch-backed stand-in for the `jax` API surface the reference's hot path touches.

Fixture-generator infrastructure (build container only; see tools/make_golden.py).  JAX is not
installable here, so the reference's OWN ``network.py`` / ``hamiltonian.py`` / ``train.py`` are
imported against this module and executed verbatim:

  * ``jax.numpy``      -> per call: torch when any argument holds a torch tensor (or while
                          ``torch_mode()`` is active, for the array constructors), numpy otherwise,
                          so the forward/Ewald/PBC fixtures keep coming from plain numpy
  * ``jax.grad / jvp / hessian`` -> ``torch.func.grad / jvp / hessian`` (float64)
  * ``jax.vmap``       -> python loop over the mapped axis + stack (pytree arguments, ``None`` axes)
  * ``jax.lax.fori_loop / scan`` -> python loops
  * ``jax.custom_jvp`` -> the primal when called; ``jax.value_and_grad`` of such a function is the
                          linear transpose of its jvp rule (reverse-mode over the rule), which is how
                          JAX itself differentiates a custom_jvp function
  * ``jax.random``     -> a recording numpy Generator (the Metropolis noise is replayed by the tests)

Three numpy idioms of the reference that a torch.Tensor lacks are patched onto torch.Tensor for the
lifetime of the generator process only: ``a // b`` (floor with zero derivative, like jnp),
``t.transpose(tuple)`` and ``t.size`` used as an integer.
"""
import contextlib
import dataclasses
import sys
import types

import numpy as np
import scipy.special
import torch
import torch.func as tf

_TORCH_MODE = [False]
# Working precision of the torch legs: float64 / complex128 (every fixture of rounds 2-4), or float32 / complex64 for
# tools/make_f32_reference.py -- the reference's own hamiltonian.py run the way JAX runs it by default, in single precision.
WORK = [torch.float64, torch.complex128]


@contextlib.contextmanager
def working_dtype(real):
    old = list(WORK)
    WORK[0] = real
    WORK[1] = torch.complex64 if real == torch.float32 else torch.complex128
    try:
        yield
    finally:
        WORK[0], WORK[1] = old



@contextlib.contextmanager
def torch_mode(on=True):
    old = _TORCH_MODE[0]
    _TORCH_MODE[0] = on
    try:
        yield
    finally:
        _TORCH_MODE[0] = old


def _has_tensor(o):
    if isinstance(o, torch.Tensor):
        return True
    if isinstance(o, (list, tuple)):
        return any(_has_tensor(v) for v in o)
    return False


def _t(o):
    """numpy / python scalar / (nested) list -> torch tensor (float64 / complex128 / int64)."""
    if isinstance(o, torch.Tensor):
        return o
    if isinstance(o, (list, tuple)) and _has_tensor(o):
        return torch.stack([_t(v) for v in o])
    t = torch.as_tensor(np.asarray(o))
    if t.is_floating_point() and t.dtype != WORK[0]:
        t = t.to(WORK[0])
    elif t.is_complex() and t.dtype != WORK[1]:
        t = t.to(WORK[1])
    return t


def to_numpy(t):
    """Plain numpy value of a tensor, also from inside torch.func transforms (used for the Ewald term, which is
    not differentiated): the transform levels are popped while the wrappers are peeled off."""
    from torch._functorch.pyfunctorch import temporarily_pop_interpreter_stack
    fc = torch._C._functorch
    with contextlib.ExitStack() as st:
        while fc.peek_interpreter_stack() is not None:
            st.enter_context(temporarily_pop_interpreter_stack())
        while fc.is_functorch_wrapped_tensor(t):
            t = fc.get_unwrapped(t)
        return t.detach().numpy().copy()


# ------------------------------------------------------------------------------ torch.Tensor patches
class _SizeProxy(int):
    """`t.size` that still works as the method torch code calls and as numpy's element count."""
    def __new__(cls, tensor, orig):
        self = int.__new__(cls, tensor.numel())
        self._call = lambda *a, **k: orig(tensor, *a, **k)
        return self

    def __call__(self, *a, **k):
        return self._call(*a, **k)


def patch_tensor():
    if getattr(torch.Tensor, '_ds_numpy_idioms', False):
        return
    orig_size = torch.Tensor.size
    orig_transpose = torch.Tensor.transpose
    torch.Tensor.size = property(lambda self: _SizeProxy(self, orig_size))

    def transpose(self, *dims):
        if len(dims) == 1 and isinstance(dims[0], (tuple, list)):
            return self.permute(*dims[0])
        return orig_transpose(self, *dims)
    torch.Tensor.transpose = transpose
    torch.Tensor.__floordiv__ = lambda a, b: torch.floor(a / b)
    torch.Tensor.__rfloordiv__ = lambda a, b: torch.floor(b / a)
    torch.Tensor._ds_numpy_idioms = True


# ------------------------------------------------------------------------------ pytrees
def tree_map(f, tree, *rest):
    if isinstance(tree, dict):
        return {k: tree_map(f, tree[k], *[r[k] for r in rest]) for k in tree}
    if isinstance(tree, (list, tuple)):
        out = [tree_map(f, v, *[r[i] for r in rest]) for i, v in enumerate(tree)]
        return type(tree)(out) if isinstance(tree, tuple) else out
    if tree is None:
        return None
    return f(tree, *rest)


def tree_leaves(tree):
    out = []
    tree_map(lambda x: out.append(x), tree)
    return out


# ------------------------------------------------------------------------------ jax.numpy
def _make_jnp():
    jnp = types.ModuleType('jax.numpy')
    for name in dir(np):
        if not name.startswith('__'):
            setattr(jnp, name, getattr(np, name))
    jnp.DeviceArray = np.ndarray
    jnp.ndarray = np.ndarray

    def dispatch(name, torch_impl):
        np_impl = getattr(np, name)

        def fn(*a, **k):
            if _has_tensor(a) or _has_tensor(tuple(k.values())):
                return torch_impl(*a, **k)
            return np_impl(*a, **k)
        fn.__name__ = name
        setattr(jnp, name, fn)

    def _axis(axis):
        return tuple(axis) if isinstance(axis, list) else axis

    def _np_sum(a, axis=None, **kw):
        return np.sum(a, axis=_axis(axis), **kw)

    def _sum(a, axis=None, keepdims=False):
        a = _t(a)
        return a.sum() if axis is None else a.sum(dim=_axis(axis), keepdim=keepdims)
    jnp.sum = lambda a, axis=None, **kw: (_sum(a, axis, **kw) if _has_tensor(a) else _np_sum(a, axis, **kw))

    def _mean(a, axis=None, keepdims=False):
        a = _t(a)
        return a.mean() if axis is None else a.mean(dim=_axis(axis), keepdim=keepdims)
    dispatch('mean', _mean)

    for nm, tfn in (('abs', torch.abs), ('exp', torch.exp), ('log', torch.log), ('sin', torch.sin),
                    ('cos', torch.cos), ('tanh', torch.tanh), ('sqrt', torch.sqrt), ('angle', torch.angle),
                    ('conjugate', torch.conj), ('trace', torch.trace)):
        dispatch(nm, (lambda f: lambda a: f(_t(a)))(tfn))
    dispatch('argmax', lambda a: torch.argmax(_t(a)))
    dispatch('dot', lambda a, b: torch.matmul(_t(a), _promote(_t(b), _t(a))))
    dispatch('matmul', lambda a, b: torch.matmul(*_promote2(_t(a), _t(b))))
    dispatch('einsum', lambda spec, *ops: torch.einsum(spec, *_promote_all([_t(o) for o in ops])))
    dispatch('concatenate', lambda seq, axis=0: torch.cat(_promote_all([_t(s) for s in seq]), dim=axis))
    dispatch('stack', lambda seq, axis=0: torch.stack(_promote_all([_t(s) for s in seq]), dim=axis))
    dispatch('expand_dims', lambda a, axis: torch.unsqueeze(_t(a), axis))
    dispatch('squeeze', lambda a, axis=None: torch.squeeze(_t(a)) if axis is None else torch.squeeze(_t(a), axis))
    dispatch('reshape', lambda a, shape: torch.reshape(_t(a), tuple(shape)))
    dispatch('shape', lambda a: tuple(_t(a).shape))
    dispatch('transpose', lambda a, axes=None: _t(a).permute(*axes) if axes is not None else _t(a).T)
    dispatch('tile', lambda a, reps: _t(a).repeat(*reps))
    dispatch('clip', lambda a, lo, hi: torch.minimum(torch.maximum(_t(a), _t(lo)), _t(hi)))
    dispatch('median', lambda a: _np_median(_t(a)))
    dispatch('allclose', lambda a, b, rtol=1e-5, atol=1e-8: torch.allclose(_t(a), _t(b), rtol=rtol, atol=atol))

    jnp.where = lambda *a: (torch.where(*[_t(v) for v in a]) if _has_tensor(a) else np.where(*a).view(AtArray))

    def _split(a, idx, axis=0):
        a = _t(a)
        if isinstance(idx, int):
            return list(torch.tensor_split(a, idx, dim=axis))
        return list(torch.tensor_split(a, [int(i) for i in idx], dim=axis))
    dispatch('split', _split)
    dispatch('array_split', _split)

    def _asarray(a, dtype=None):
        if _has_tensor(a) or _TORCH_MODE[0]:
            return _t(a)
        return np.asarray(a, dtype=dtype)
    jnp.asarray = _asarray
    jnp.array = _asarray

    def ctor(name, timpl):
        np_impl = getattr(np, name)
        setattr(jnp, name, lambda *a, **k: (timpl(*a, **k) if _TORCH_MODE[0] else np_impl(*a, **k)))
    ctor('eye', lambda n: torch.eye(n, dtype=WORK[0]))
    ctor('ones', lambda shape: torch.ones(tuple(shape) if not isinstance(shape, int) else (shape,), dtype=WORK[0]))
    ctor('zeros', lambda shape: torch.zeros(tuple(shape) if not isinstance(shape, int) else (shape,), dtype=WORK[0]))

    linalg = types.ModuleType('jax.numpy.linalg')
    for name in dir(np.linalg):
        if not name.startswith('__'):
            setattr(linalg, name, getattr(np.linalg, name))

    def ldispatch(name, timpl):
        np_impl = getattr(np.linalg, name)
        setattr(linalg, name, lambda *a, **k: (timpl(*a, **k) if _has_tensor(a) else np_impl(*a, **k)))
    ldispatch('inv', lambda a: torch.linalg.inv(_t(a)))
    ldispatch('det', lambda a: torch.linalg.det(_t(a)))
    ldispatch('slogdet', lambda a: tuple(torch.linalg.slogdet(_t(a))))
    ldispatch('norm', lambda a, axis=None, keepdims=False: (
        torch.linalg.vector_norm(_t(a)) if axis is None else torch.linalg.vector_norm(_t(a), dim=axis, keepdim=keepdims)))
    jnp.linalg = linalg
    return jnp


def _np_median(t):
    """numpy's median (mean of the two middle values for an even count); torch.median takes the lower one."""
    s, _ = torch.sort(t.reshape(-1))
    n = s.numel()
    return s[n // 2] if n % 2 else 0.5 * (s[n // 2 - 1] + s[n // 2])


def _promote_all(ts):
    if any(t.is_complex() for t in ts):
        return [t.to(WORK[1]) for t in ts]
    if any(t.is_floating_point() for t in ts):
        return [t.to(WORK[0]) if not t.is_floating_point() else t for t in ts]
    return ts


def _promote2(a, b):
    return tuple(_promote_all([a, b]))


def _promote(b, a):
    return _promote_all([a, b])[1]


# ------------------------------------------------------------------------------ transforms
def _take(a, i, ax):
    if isinstance(a, torch.Tensor):
        return a.select(ax, i)
    return np.take(a, i, axis=ax)


def _stack(items, axis):
    if isinstance(items[0], torch.Tensor):
        return torch.stack(items, dim=axis)
    return np.stack(items, axis=axis)


def vmap(f, in_axes=0, out_axes=0):
    def wrapped(*args):
        ia = in_axes if isinstance(in_axes, (tuple, list)) else (in_axes,) * len(args)
        n = None
        for a, ax in zip(args, ia):
            if ax is not None:
                leaf = tree_leaves(a)[0]
                n = leaf.shape[ax]
                break
        outs = [f(*[a if ax is None else tree_map(lambda l: _take(l, i, ax), a) for a, ax in zip(args, ia)])
                for i in range(n)]
        return tree_map(lambda *ls: _stack(list(ls), out_axes), outs[0], *outs[1:])
    return wrapped


def fori_loop(lo, hi, body, val):
    for i in range(lo, hi):
        val = body(i, val)
    return val


def scan(body, init, xs):
    n = tree_leaves(xs)[0].shape[0]
    carry, ys = init, []
    for i in range(n):
        carry, y = body(carry, tree_map(lambda l: l[i], xs))
        ys.append(y)
    return carry, tree_map(lambda *ls: _stack(list(ls), 0), ys[0], *ys[1:])


class custom_jvp:
    def __init__(self, f):
        self.f = f
        self.jvp_rule = None

    def defjvp(self, rule):
        self.jvp_rule = rule
        return rule

    def __call__(self, *a):
        return self.f(*a)


def grad(f, argnums=0, holomorphic=False, has_aux=False):
    return tf.grad(f, argnums=argnums, has_aux=has_aux)


def value_and_grad(f, argnums=0, has_aux=False):
    if isinstance(f, custom_jvp):
        assert argnums == 0 and has_aux

        def vg(params, data):
            zero_d = torch.zeros_like(data)
            box = {}

            def lin(t):
                primal_out, (tdot, _) = f.jvp_rule((params, data), (t, zero_d))
                box['out'] = primal_out
                return tdot
            g = tf.grad(lin)(tree_map(torch.zeros_like, params))
            out = tree_map(lambda v: v.detach() if isinstance(v, torch.Tensor) else v, box['out'])
            return out, g
        return vg

    def vg(*args):
        if has_aux:
            g, (v, aux) = tf.grad_and_value(f, argnums=argnums, has_aux=True)(*args)
            return (v, aux), g
        g, v = tf.grad_and_value(f, argnums=argnums)(*args)
        return v, g
    return vg


class AtArray(np.ndarray):
    """numpy array with the `x.at[idx].add(v)` update of jax arrays (qmc.py:269); views and reshapes keep the type."""
    class _At:
        def __init__(self, arr):
            self.arr = arr

        def __getitem__(self, idx):
            arr = self.arr

            class _Upd:
                def add(self, v):
                    out = np.array(arr, copy=True)
                    out[idx] += v
                    return out.view(AtArray)
            return _Upd()

    @property
    def at(self):
        return AtArray._At(self)


class Recorder:
    """jax.random stand-in: numpy Generator whose draws are recorded."""
    def __init__(self):
        self.reset(0)

    def reset(self, seed):
        self.rng = np.random.default_rng(seed)
        self.normals, self.uniforms = [], []

    def split(self, key, num=2):
        return tuple(key for _ in range(num))

    def normal(self, key, shape=()):
        v = self.rng.standard_normal(shape)
        self.normals.append(v)
        return v

    def uniform(self, key, shape=()):
        v = self.rng.uniform(size=shape)
        self.uniforms.append(v)
        return v

    def PRNGKey(self, seed):
        return np.array([0, seed], dtype=np.uint32)


RECORDER = Recorder()


def install(reference_root):
    """Put the stand-in modules into sys.modules and the reference on sys.path."""
    patch_tensor()
    jnp = _make_jnp()

    lax = types.ModuleType('jax.lax')
    lax.erfc = lambda a: torch.erfc(a) if isinstance(a, torch.Tensor) else scipy.special.erfc(a)
    lax.fori_loop = fori_loop
    lax.scan = scan
    lax.pmean = lambda x, axis_name=None: x
    lax.psum = lambda x, axis_name=None: x

    core = types.ModuleType('jax.core')

    def axis_frame(name):
        raise NameError(name)
    core.axis_frame = axis_frame

    rnd = types.ModuleType('jax.random')
    for n in ('split', 'normal', 'uniform', 'PRNGKey'):
        setattr(rnd, n, getattr(RECORDER, n))

    jax = types.ModuleType('jax')
    jax.numpy, jax.lax, jax.core, jax.random = jnp, lax, core, rnd
    jax.vmap = vmap
    jax.jit = lambda f, **kw: f
    jax.pmap = lambda f, **kw: f
    jax.grad = grad
    jax.value_and_grad = value_and_grad
    jax.jvp = lambda f, primals, tangents: tf.jvp(f, tuple(primals), tuple(tangents))
    jax.hessian = lambda f, argnums=0: tf.hessian(f, argnums=argnums)
    jax.custom_jvp = custom_jvp
    jax.tree_map = tree_map
    sys.modules.update({'jax': jax, 'jax.numpy': jnp, 'jax.lax': lax, 'jax.core': core, 'jax.random': rnd})

    tags = types.ModuleType('DeepSolid.curvature_tags_and_blocks')
    tags.register_repeated_dense = lambda y, x, w, b: y
    tags.register_qmc1 = lambda y, x, w, **kw: y
    sys.modules['DeepSolid.curvature_tags_and_blocks'] = tags

    chex = types.ModuleType('chex')
    chex.dataclass = dataclasses.dataclass
    sys.modules['chex'] = chex
    for m in ('DeepSolid.utils', 'DeepSolid.utils.kfac_ferminet_alpha',
              'DeepSolid.utils.kfac_ferminet_alpha.loss_functions'):
        sys.modules[m] = types.ModuleType(m)
    lf = sys.modules['DeepSolid.utils.kfac_ferminet_alpha.loss_functions']
    lf.register_normal_predictive_distribution = lambda *a, **k: None     # KFAC tag: identity
    sys.modules['DeepSolid.utils.kfac_ferminet_alpha'].loss_functions = lf
    sys.modules['DeepSolid.utils'].kfac_ferminet_alpha = sys.modules['DeepSolid.utils.kfac_ferminet_alpha']

    for m in ('pyscf', 'pyscf.pbc', 'pyscf.pbc.gto'):
        sys.modules[m] = types.ModuleType(m)
    sys.modules['pyscf'].pbc = sys.modules['pyscf.pbc']
    sys.modules['pyscf.pbc'].gto = sys.modules['pyscf.pbc.gto']
    sys.modules['pyscf.pbc.gto'].Cell = object
    sys.path.insert(0, reference_root)
    return jax
