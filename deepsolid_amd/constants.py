"""Cross-device reductions (reference DeepSolid/constants.py:26-45).

The reference runs one process with ``jax.pmap`` over local devices and reduces
with ``lax.pmean``.  Here the unit is one process per GPU (torchrun); the
equivalent of ``pmean_if_pmap`` is an all-reduce over RCCL (backend "nccl" on
ROCm) when a process group exists and the identity otherwise -- the same
degrade-to-identity behaviour as constants.py:33-41.
"""
import torch
import torch.distributed as dist

PMAP_AXIS_NAME = 'qmc_pmap_axis'


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def pmean_if_pmap(obj, axis_name=PMAP_AXIS_NAME):
    del axis_name
    if world_size() == 1:
        return obj
    t = obj.clone()
    if t.is_complex():
        r = torch.view_as_real(t).contiguous()
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        return torch.view_as_complex(r) / world_size()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t / world_size()


def psum_if_pmap(obj, axis_name=PMAP_AXIS_NAME):
    del axis_name
    if world_size() == 1:
        return obj
    t = obj.clone()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def pmean_packed(*scalars):
    """One all-reduce for several scalars (energy mean Re/Im, variance, pmove ...):
    the message the reference sends as separate pmeans (train.py:78-80, qmc.py:361)."""
    packed = torch.stack([torch.as_tensor(s).reshape(()).to(torch.float64) for s in scalars])
    if world_size() > 1:
        dist.all_reduce(packed, op=dist.ReduceOp.SUM)
        packed = packed / world_size()
    return tuple(packed.unbind(0))
