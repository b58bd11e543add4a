"""The one-shot tensor-parallel all-reduce (mv_allreduce_*, csrc/allreduce.hip) through the C ABI.

A test box has one GPU, so the ranks are processes that share it: each maps the others' exchange
buffers through HIP IPC handles exactly as ranks on different GPUs of a node would.  Expected values
are computed on the CPU: partials summed in fp32 in rank order and rounded to bf16 once (what
oracle/ref_model.py's `_allreduce` does for core/parallel_config.py:69-74), followed by the
reference's add + RMSNorm (oracle/ref_ops.py).  Integer/bit work: all ranks must hold the same bits,
and those must equal the expected bits."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _partial(rank, rows, n, seed):
    g = torch.Generator().manual_seed(1000 * seed + rank)
    return torch.randn(rows, n, generator=g).to(torch.bfloat16)


def _expected_sum(world, rows, n, seed):
    acc = torch.zeros(rows, n, dtype=torch.float32)
    for r in range(world):                      # rank order, fp32, one rounding
        acc += _partial(r, rows, n, seed).float()
    return acc.to(torch.bfloat16)


def test_single_rank_communicator_equals_the_norm_kernels(mv):
    """world = 1: the kernel's sum is its own partial, so slabs in -> (y, normed) out must equal
    mv_wgemm_postnorm bit for bit, and bf16 rows in must equal the stand-alone add / norm kernels."""
    from micro_vllm_amd import _native as nv
    from micro_vllm_amd.kernel.allreduce import IpcAllReduce
    from micro_vllm_amd.kernel.linear import (StreamedWeight, add_, gemm_workspace, wgemm, wgemm_postnorm,
                                              wgemm_slabs)
    from micro_vllm_amd.kernel.rmsnorm_add import rmsnorm_
    from micro_vllm_amd.kernel.rmsnorm_residual import rmsnorm_residual_gemm

    g = torch.Generator().manual_seed(3)
    comm = IpcAllReduce(0, 1, 64, 8192)
    for m, n, k in ((1, 4096, 4096), (3, 4096, 512), (16, 4096, 1792), (33, 8192, 3584), (8, 256, 96)):
        x = torch.randn(m, k, generator=g).to(torch.bfloat16).to(DEV)
        w = StreamedWeight.from_linear((torch.randn(n, k, generator=g) * 0.05).to(torch.bfloat16).to(DEV))
        res = torch.randn(m, n, generator=g).to(torch.bfloat16).to(DEV)
        lnw = (1 + 0.1 * torch.randn(n, generator=g)).to(torch.bfloat16).to(DEV)
        ws = gemm_workspace(DEV, nv.lib().mv_wgemm_postnorm_workspace_bytes(m, n, k))
        for vfr in (False, True):
            y0, n0 = torch.empty_like(res), torch.empty_like(res)
            wgemm_postnorm(x, w, y0, lnw, 1e-5, n0, residual=res, var_from_rounded=vfr, workspace=ws)
            y1, n1 = torch.empty_like(res), torch.empty_like(res)
            ks = wgemm_slabs(x, w, ws)
            assert ks == nv.lib().mv_wgemm_slab_count(m, n, k) >= 1
            comm.addnorm(m, n, slabs=ws, ksplit=ks, residual=res, y=y1, ln_weight=lnw, eps=1e-5, normed=n1,
                         var_from_rounded=vfr)
            assert torch.equal(y0, y1) and torch.equal(n0, n1), (m, n, k, vfr)
        # bf16 rows in: matmul -> (all_reduce) -> rmsnorm_residual_gemm, and -> add_ -> rmsnorm_
        part = wgemm(x, w, torch.empty_like(res), workspace=ws)
        y1, n1 = torch.empty_like(res), torch.empty_like(res)
        comm.addnorm(m, n, partial=part, residual=res, y=y1, ln_weight=lnw, eps=1e-5, normed=n1)
        n0, y0 = rmsnorm_residual_gemm(part, res, lnw, torch.empty_like(res), torch.empty_like(res), 1e-5)
        assert torch.equal(y0, y1) and torch.equal(n0, n1)
        comm.addnorm(m, n, partial=part, residual=res, y=y1, ln_weight=lnw, eps=1e-5, normed=n1,
                     var_from_rounded=True)
        y0 = add_(part.clone(), res)
        n0 = rmsnorm_(y0, lnw, torch.empty_like(res), 1e-5)
        assert torch.equal(y0, y1) and torch.equal(n0, n1)
        # plain sum, in place
        z = part.clone()
        comm.all_reduce(z)
        assert torch.equal(z, part)
    assert comm.status() == 0
    comm.close()
    # capacity and argument checks
    small = IpcAllReduce(0, 1, 4, 256)
    assert not small.fits(5, 256) and not small.fits(4, 512) and small.fits(4, 256)
    t = torch.zeros(5, 256, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(AssertionError):
        small.all_reduce(t)          # more rows than the communicator was made for
    small.close()


def _worker(rank, world, port, out_q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), MVLLM_DIST_BACKEND="gloo", MVLLM_ALLREDUCE_TIMEOUT_MS="20000")
    sys.path.insert(0, ROOT)
    import ctypes

    import micro_vllm_amd  # noqa: F401
    from micro_vllm_amd import _native as nv
    from micro_vllm_amd.core import parallel_config as pc
    from micro_vllm_amd.kernel.allreduce import IpcAllReduce

    torch.cuda.set_device(0)
    pc.setup()
    red = pc.tensor_parallel_reducer(16, 4096)
    res = {"ipc": isinstance(red, IpcAllReduce)}
    if not res["ipc"]:
        out_q.put((rank, res))
        return
    outs = []
    # plain sums of different sizes, back to back (the epoch of every row advances on its own)
    for seed, (rows, n) in enumerate([(1, 4096), (16, 4096), (3, 256), (16, 4096), (1, 4096), (5, 1024)]):
        x = _partial(rank, rows, n, seed).to(DEV)
        red.all_reduce(x)
        outs.append(x.cpu())
    # sum + residual + norm, both variance conventions
    g = torch.Generator().manual_seed(77)
    resid = torch.randn(8, 4096, generator=g).to(torch.bfloat16).to(DEV)
    lnw = (1 + 0.1 * torch.randn(4096, generator=g)).to(torch.bfloat16).to(DEV)
    for seed, vfr in ((20, False), (21, True)):
        y = torch.empty(8, 4096, dtype=torch.bfloat16, device=DEV)
        nrm = torch.empty_like(y)
        red.addnorm(8, 4096, partial=_partial(rank, 8, 4096, seed).to(DEV), residual=resid, y=y,
                    ln_weight=lnw, eps=1e-5, normed=nrm, var_from_rounded=vfr)
        outs += [y.cpu(), nrm.cpu()]
    # a captured sequence of four reductions, replayed: epochs keep growing across replays
    bufs = [_partial(rank, 4, 4096, 30 + i).to(DEV) for i in range(4)]
    src = [b.clone() for b in bufs]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for b in bufs:
            red.all_reduce(b)                      # warm-up, also advances every rank alike
        side.synchronize()
        for b, s in zip(bufs, src):
            b.copy_(s)
        side.synchronize()
        nv.check("begin", nv.lib().mv_graph_begin_capture(side.cuda_stream))
        for b in bufs:
            red.all_reduce(b)
        h = ctypes.c_void_p()
        nv.check("end", nv.lib().mv_graph_end_capture(side.cuda_stream, ctypes.byref(h)))
        for rep in range(5):
            for b, s in zip(bufs, src):
                b.copy_(s)
            nv.check("launch", nv.lib().mv_graph_launch(h, side.cuda_stream))
            side.synchronize()
            outs += [b.cpu().clone() for b in bufs]
        nv.lib().mv_graph_destroy(h)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    res["status"] = red.status()
    res["outs"] = [o.view(torch.int16).numpy() for o in outs]
    pc.barrier()
    out_q.put((rank, res))
    red.close()
    import torch.distributed as dist
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_ranks_sharing_one_gpu_agree_and_match_the_rank_ordered_sum(world):
    from oracle.ref_ops import rmsnorm_ref, rmsnorm_residual_ref

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, res = q.get(timeout=240)
        got[r] = res
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(got[r]["ipc"] for r in range(world)), "the one-shot all-reduce did not come up on this box"
    assert all(got[r]["status"] == 0 for r in range(world)), "a rank timed out waiting for a peer"
    outs = {r: [torch.from_numpy(o).view(torch.bfloat16) for o in got[r]["outs"]] for r in range(world)}
    for r in range(1, world):
        assert len(outs[r]) == len(outs[0])
        for a, b in zip(outs[0], outs[r]):
            assert torch.equal(a, b), "ranks disagree"
    it = iter(outs[0])
    for seed, (rows, n) in enumerate([(1, 4096), (16, 4096), (3, 256), (16, 4096), (1, 4096), (5, 1024)]):
        assert torch.equal(next(it), _expected_sum(world, rows, n, seed)), (seed, rows, n)
    g = torch.Generator().manual_seed(77)
    resid = torch.randn(8, 4096, generator=g).to(torch.bfloat16)
    lnw = (1 + 0.1 * torch.randn(4096, generator=g)).to(torch.bfloat16)
    t = _expected_sum(world, 8, 4096, 20)
    want_n, want_y = rmsnorm_residual_ref(t, resid, lnw, 1e-5)       # kernel/rmsnorm_residual.py:74-124
    y, nrm = next(it), next(it)
    assert torch.equal(y, want_y)
    assert (nrm.float() - want_n.float()).abs().max() <= 2 ** -7 * want_n.float().abs().max()
    assert (nrm != want_n).float().mean() < 0.005                     # single-ulp flips of the norm only
    t = _expected_sum(world, 8, 4096, 21)
    want_y = (t.float() + resid.float()).to(torch.bfloat16)           # model_graph.py:219, then :143
    want_n = rmsnorm_ref(want_y, lnw, 1e-5)
    y, nrm = next(it), next(it)
    assert torch.equal(y, want_y)
    assert (nrm != want_n).float().mean() < 0.005
    for rep in range(5):
        for i in range(4):
            assert torch.equal(next(it), _expected_sum(world, 4, 4096, 30 + i)), (rep, i)


def _gather_worker(rank, world, port, out_q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), MVLLM_DIST_BACKEND="gloo", MVLLM_ALLREDUCE_TIMEOUT_MS="20000")
    sys.path.insert(0, ROOT)
    import ctypes

    import micro_vllm_amd  # noqa: F401
    from micro_vllm_amd import _native as nv
    from micro_vllm_amd.core import parallel_config as pc
    from micro_vllm_amd.kernel.allreduce import IpcAllGather, IpcAllReduce

    torch.cuda.set_device(0)
    pc.setup()
    red = pc.tensor_parallel_reducer(16, 4096)           # the gatherer is only offered next to it
    n_local = 16032                                      # 128256 / 8: the 8B vocabulary cut eight ways
    gat = pc.tensor_parallel_gatherer(16, n_local)
    res = {"ipc": isinstance(red, IpcAllReduce) and isinstance(gat, IpcAllGather)}
    if not res["ipc"]:
        out_q.put((rank, res))
        return
    outs = []
    # different row counts and slice widths back to back, the all-reduce in between (as a step does)
    for seed, (rows, n) in enumerate([(1, n_local), (16, n_local), (3, 256), (1, n_local), (5, 1000)]):
        x = _partial(rank, rows, n, 100 + seed).to(DEV)
        out = torch.full((rows, world * n + 8), 7.0, dtype=torch.bfloat16, device=DEV)
        gat.gather(x, out[:, :world * n])
        red.all_reduce(torch.zeros(1, 256, dtype=torch.bfloat16, device=DEV))
        outs.append(out.cpu())
    # captured and replayed: epochs keep growing, both data slots come round
    x = torch.empty(4, n_local, dtype=torch.bfloat16, device=DEV)
    out = torch.zeros(4, world * n_local, dtype=torch.bfloat16, device=DEV)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        gat.gather(x, out)
        side.synchronize()
        nv.check("begin", nv.lib().mv_graph_begin_capture(side.cuda_stream))
        gat.gather(x, out)
        h = ctypes.c_void_p()
        nv.check("end", nv.lib().mv_graph_end_capture(side.cuda_stream, ctypes.byref(h)))
        for rep in range(5):
            x.copy_(_partial(rank, 4, n_local, 200 + rep).to(DEV))
            nv.check("launch", nv.lib().mv_graph_launch(h, side.cuda_stream))
            side.synchronize()
            outs.append(out.cpu().clone())
        nv.lib().mv_graph_destroy(h)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    res["status"] = max(red.status(), gat.status())
    res["outs"] = [o.view(torch.int16).numpy() for o in outs]
    pc.barrier()
    out_q.put((rank, res))
    gat.close()
    red.close()
    import torch.distributed as dist
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_all_gather_of_logit_slices_among_ranks_sharing_one_gpu(world):
    """Every rank ends with every rank's slice at that rank's columns, bit for bit, nothing written
    past the row, eager and under graph replay."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, res = q.get(timeout=240)
        got[r] = res
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(got[r]["ipc"] for r in range(world)), "the communicators over mapped buffers did not come up"
    assert all(got[r]["status"] == 0 for r in range(world)), "a rank timed out waiting for a peer"
    outs = {r: [torch.from_numpy(o).view(torch.bfloat16) for o in got[r]["outs"]] for r in range(world)}
    n_local = 16032
    for r in range(world):
        it = iter(outs[r])
        for seed, (rows, n) in enumerate([(1, n_local), (16, n_local), (3, 256), (1, n_local), (5, 1000)]):
            o = next(it)
            want = torch.cat([_partial(q_, rows, n, 100 + seed) for q_ in range(world)], dim=1)
            assert torch.equal(o[:, :world * n], want), (r, seed)
            assert torch.all(o[:, world * n:] == 7.0), "wrote past the row"
        for rep in range(5):
            want = torch.cat([_partial(q_, 4, n_local, 200 + rep) for q_ in range(world)], dim=1)
            assert torch.equal(next(it), want), (r, rep)
