"""Tensor-parallel process state: one process per GPU, torch.distributed over RCCL
(`backend="nccl"` on ROCm) or gloo on CPU.  Same functions as the reference's
core/parallel_config.py:14-79 (setup / get_group / get_rank / get_world_size / rank0 /
all_reduce / barrier)."""
import os

import torch
import torch.distributed as dist


class _State:
    group = None


_st = _State()


def setup():
    """Join the process group described by RANK / WORLD_SIZE (no-op for a single process)."""
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world > 1 and not dist.is_initialized():
        if torch.cuda.is_available() and os.environ.get("MVLLM_DIST_BACKEND", "nccl") != "gloo":
            local = int(os.environ.get("LOCAL_RANK", rank)) % torch.cuda.device_count()
            torch.cuda.set_device(local)
            dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    if dist.is_initialized() and _st.group is None:
        _st.group = dist.group.WORLD


def get_group():
    return _st.group if dist.is_initialized() else None


def get_rank():
    return dist.get_rank() if dist.is_initialized() else 0


def get_world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def rank0():
    rank = os.environ.get("RANK")
    if rank is not None:
        return int(rank) == 0
    return get_rank() == 0


def all_reduce(input_):
    """In-place SUM over the tensor-parallel ranks on the current stream; returns its argument."""
    if get_world_size() == 1:
        return input_
    group = get_group() or dist.group.WORLD
    if dist.get_backend(group) == "gloo":
        # gloo is the rehearsal backend (CPU ranks, or several ranks sharing one GPU in tests):
        # reduce on the host in fp32 and round once
        tmp = input_.detach().float().cpu()
        dist.all_reduce(tmp, group=group)
        input_.copy_(tmp)
        return input_
    dist.all_reduce(input_, group=group)
    return input_


def barrier():
    if get_world_size() > 1 and dist.is_initialized():
        dist.barrier(group=get_group() or dist.group.WORLD)
