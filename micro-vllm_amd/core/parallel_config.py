"""Tensor-parallel process state: one process per GPU, torch.distributed over RCCL
(`backend="nccl"` on ROCm) or gloo on CPU.  Same functions as the reference's
core/parallel_config.py:14-79 (setup / get_group / get_rank / get_world_size / rank0 /
all_reduce / barrier)."""
import os

import torch
import torch.distributed as dist


class _State:
    group = None
    reducer = None
    gatherer = None
    retired = []    # communicators replaced by larger ones: runners built earlier may still hold them


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


def host_side_backend():
    """True when the process group reduces on the host (gloo): such collectives cannot be captured
    into a device graph."""
    if not dist.is_initialized():
        return False
    return dist.get_backend(get_group() or dist.group.WORLD) == "gloo"


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


def tensor_parallel_reducer(max_rows: int, max_n: int):
    """The object that sums the decode step's [bs, hidden] partials over the ranks and applies the
    add + RMSNorm that follows (kernel/allreduce.py).  With every rank on a GPU of one node this is
    the one-shot all-reduce over mapped peer buffers, unless MVLLM_ALLREDUCE=dist asks for
    torch.distributed; whichever is chosen, all ranks choose the same (they vote).  One per process,
    sized for the library's row limit whatever the first caller asks for, so that a second runner
    with a larger batch shares it; one that is outgrown all the same (a wider hidden size) stays
    alive for the runners and captured graphs that hold it and a larger one is made beside it."""
    from ..kernel.allreduce import MAX_N, MAX_ROWS, DistAllReduce, IpcAllReduce

    world = get_world_size()
    assert world > 1, "a single rank has nothing to reduce"
    red = _st.reducer
    if red is not None and red.fits(max_rows, max_n):
        return red
    if red is not None:
        _st.retired.append(red)
    want_ipc = (os.environ.get("MVLLM_ALLREDUCE", "ipc") != "dist" and torch.cuda.is_available()
                and world <= 8 and max_rows <= MAX_ROWS and max_n <= MAX_N and max_n % 8 == 0)
    comm = _agree_on_ipc_comm(lambda: IpcAllReduce(get_rank(), world, MAX_ROWS, max_n), want_ipc,
                              "one-shot all-reduce")
    _st.reducer = comm if comm is not None else DistAllReduce(all_reduce)
    return _st.reducer


def _agree_on_ipc_comm(make, want: bool, what: str):
    """Set-up of a communicator over mapped peer buffers that either every rank ends up with or none:
    allocate, swap handles, map the peers, run its self-test, vote.  Every rank takes part in all
    three exchanges whatever happened to it locally, so that a failure on one rank cannot leave the
    others waiting in a collective.  Returns the communicator, or None when any rank said no."""
    world = get_world_size()
    group = get_group() or dist.group.WORLD
    log = __import__("logging").getLogger(__name__)
    comm, ok, handle = None, 0, b""
    if want:
        try:
            comm = make()
            handle = comm.handle()
        except Exception as e:
            log.warning("%s: no exchange buffer on rank %d: %r", what, get_rank(), e)
            comm = None
    handles = [None] * world
    dist.all_gather_object(handles, handle, group=group)
    if comm is not None and all(handles):
        try:
            comm.connect(handles)
            ok = 1
        except Exception as e:
            log.warning("%s: rank %d cannot map its peers: %r", what, get_rank(), e)
    mapped = [None] * world
    dist.all_gather_object(mapped, ok, group=group)
    if all(v == 1 for v in mapped):
        # everybody mapped everybody: one known exchange through the buffers before they are trusted
        if not comm.self_test():
            log.warning("%s: self-test failed on rank %d", what, get_rank())
            ok = 0
    else:
        ok = 0
    votes = [None] * world
    dist.all_gather_object(votes, ok, group=group)
    if all(v == 1 for v in votes):
        return comm
    if comm is not None:
        comm.close()
    return None


def tensor_parallel_gatherer(max_rows: int, n_local: int):
    """The object that collects the ranks' [bs, n_local] slices of the logits when the LM head is cut
    by vocabulary rows (kernel/allreduce.py::IpcAllGather), or None: then the head stays whole on
    every rank, as in the reference.  Only offered next to the one-shot all-reduce (the same mapped
    buffers, the same links), and switched off by MVLLM_LM_HEAD_SHARD=0; all ranks get the same
    answer (they vote).  One per process."""
    from ..kernel.allreduce import MAX_ROWS, IpcAllGather, IpcAllReduce

    world = get_world_size()
    if world <= 1 or not isinstance(_st.reducer, IpcAllReduce):
        return None
    g = _st.gatherer
    if g is not None and g.max_rows >= max_rows and g.max_n == n_local:
        return g
    if g is not None:
        _st.retired.append(g)   # never closed under a runner that may still hold it
    want = (os.environ.get("MVLLM_LM_HEAD_SHARD", "1") != "0" and max_rows <= MAX_ROWS and n_local % 8 == 0)
    _st.gatherer = _agree_on_ipc_comm(lambda: IpcAllGather(get_rank(), world, MAX_ROWS, n_local), want,
                                      "logits all-gather")
    return _st.gatherer


def barrier():
    if get_world_size() > 1 and dist.is_initialized():
        dist.barrier(group=get_group() or dist.group.WORLD)
