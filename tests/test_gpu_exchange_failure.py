"""What the serving path does when a tensor-parallel exchange fails, and the exchange at the largest
group size.

The reference hangs when a rank stops taking part (its collectives are blocking and must be paired,
core/inference_context.py:54).  Here a reduction that waits longer than the communicator's timeout
finishes with a wrong sum and sets a host-visible status word; `InferenceEngine._decode` reads that
word behind the token read-back it does anyway and raises instead of handing the tokens on.

The test box has one GPU, so the ranks are processes sharing it (they map each other's exchange
buffers through IPC handles exactly as ranks on different GPUs would)."""
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


class Tok:
    eos_token_id = 2

    def __init__(self, vocab):
        self.vocab = vocab

    def encode(self, prompt, add_special_tokens=True):
        return [5 + (ord(c) * 7) % (self.vocab - 5) for c in prompt]

    def decode(self, ids, skip_special_tokens=True):
        return " ".join(str(i) for i in ids)


def _stall_worker(rank, world, port, out_q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), MVLLM_DIST_BACKEND="gloo", MVLLM_ALLREDUCE_TIMEOUT_MS="20000")
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    import micro_vllm_amd  # noqa: F401
    from micro_vllm_amd.core.engine import InferenceEngine
    from micro_vllm_amd.core.sequence import Sequence
    from micro_vllm_amd.kernel.allreduce import ExchangeTimeout, IpcAllReduce
    from micro_vllm_amd.models import SHAPES, build_synthetic_model

    torch.cuda.set_device(0)
    cfg = SHAPES["tiny-gqa"]
    tok = Tok(cfg.vocab_size)
    eng = InferenceEngine(model=build_synthetic_model(cfg, "cpu", seed=1234).to(DEV), tokenizer=tok,
                          max_batch_size=2, n_blocks=12, max_tokens=512, graph_batch_sizes=[1, 2],
                          prefill_timeout=0.0)
    runner = eng.graph_runner
    res = {"ipc": isinstance(runner._reducer, IpcAllReduce), "graph": 2 in runner._graphs}
    batch = [Sequence(100 + i, p, tok, 16) for i, p in enumerate(["the quick brown fox", "jumps over lazy dog"])]
    for s in batch:
        s.temperature = 0.0

    def share_tokens():
        box = [[s._next_token for s in batch]] if rank == 0 else [None]
        dist.broadcast_object_list(box, src=0)
        for s, t in zip(batch, box[0]):
            s._next_token = t
        eng.update_sequences(batch)

    with torch.inference_mode():
        eng._prefill(batch)
        share_tokens()
        eng._decode(batch)                      # a healthy step: both ranks take part
        share_tokens()
        res["healthy_tokens"] = [s.output_ids[-1] for s in batch]
        res["status_after_healthy_step"] = runner._reducer.status()
        torch.cuda.synchronize()
        dist.barrier()
        # ---- rank 1 stalls: it does not run the next step at all ------------------------------
        for comm in (runner._reducer, runner._gatherer):
            if comm is not None:
                comm.set_timeout_ms(50.0)
        before = [s._next_token for s in batch]
        res["raised"] = None
        if rank == 0:
            for s in batch:
                s._next_token = None
            try:
                eng._decode(batch)
                res["raised"] = False
            except ExchangeTimeout as e:
                res["raised"] = True
                res["message"] = str(e)
                res["waited_for"] = e.status - 1
            res["tokens_handed_on"] = [s._next_token for s in batch]
        del before
        torch.cuda.synchronize()
        dist.barrier()                          # rank 1's buffers stay mapped until rank 0 is done
    out_q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_a_stalled_rank_makes_the_step_raise_instead_of_returning_tokens():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_stall_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, res = q.get(timeout=280)
        got[r] = res
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(got[r]["ipc"] and got[r]["graph"] for r in range(world)), \
        "the one-shot all-reduce inside a captured step did not come up on this box"
    assert got[0]["healthy_tokens"] == got[1]["healthy_tokens"]
    assert got[0]["status_after_healthy_step"] == 0 and got[1]["status_after_healthy_step"] == 0
    assert got[0]["raised"] is True, "rank 0 returned tokens from a step whose reductions timed out"
    assert got[0]["waited_for"] == 1 and "rank 1" in got[0]["message"]
    assert got[0]["tokens_handed_on"] == [None, None]


def _world8_worker(rank, world, port, out_q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), MVLLM_DIST_BACKEND="gloo", MVLLM_ALLREDUCE_TIMEOUT_MS="60000")
    sys.path.insert(0, ROOT)
    import ctypes

    import torch.distributed as dist

    import micro_vllm_amd  # noqa: F401
    from micro_vllm_amd import _native as nv
    from micro_vllm_amd.core import parallel_config as pc
    from micro_vllm_amd.kernel.allreduce import IpcAllGather, IpcAllReduce

    torch.cuda.set_device(0)
    pc.setup()
    rows, n = 2, 1024
    red = pc.tensor_parallel_reducer(rows, n)
    gat = pc.tensor_parallel_gatherer(rows, n)
    res = {"ipc": isinstance(red, IpcAllReduce) and isinstance(gat, IpcAllGather)}
    if not res["ipc"]:
        out_q.put((rank, res))
        return
    g = torch.Generator().manual_seed(9)
    resid = torch.randn(rows, n, generator=g).to(torch.bfloat16).to(DEV)
    lnw = (1 + 0.1 * torch.randn(n, generator=g)).to(torch.bfloat16).to(DEV)
    part = torch.empty(rows, n, dtype=torch.bfloat16, device=DEV)
    y, nrm = torch.empty_like(part), torch.empty_like(part)
    gathered = torch.zeros(rows, world * n, dtype=torch.bfloat16, device=DEV)

    def step():
        red.addnorm(rows, n, partial=part, residual=resid, y=y, ln_weight=lnw, eps=1e-5, normed=nrm)
        gat.gather(part, gathered)

    def mine(seed):
        gg = torch.Generator().manual_seed(1000 * seed + rank)
        return torch.randn(rows, n, generator=gg).to(torch.bfloat16)

    outs = []
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        part.copy_(mine(0).to(DEV))
        step()                                   # eager
        side.synchronize()
        outs += [y.cpu().clone(), nrm.cpu().clone(), gathered.cpu().clone()]
        nv.check("begin", nv.lib().mv_graph_begin_capture(side.cuda_stream))
        step()
        h = ctypes.c_void_p()
        nv.check("end", nv.lib().mv_graph_end_capture(side.cuda_stream, ctypes.byref(h)))
        for rep in range(1, 4):                  # three replays, fresh partials each time
            part.copy_(mine(rep).to(DEV))
            nv.check("launch", nv.lib().mv_graph_launch(h, side.cuda_stream))
            side.synchronize()
            outs += [y.cpu().clone(), nrm.cpu().clone(), gathered.cpu().clone()]
        nv.lib().mv_graph_destroy(h)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    res["status"] = max(red.status(), gat.status())
    res["outs"] = [o.view(torch.int16).numpy() for o in outs]
    pc.barrier()
    out_q.put((rank, res))
    pc.barrier()
    dist.destroy_process_group()


def test_eight_ranks_through_the_one_shot_all_reduce_and_the_logits_gather():
    """world = 8 (the group size of the tensor-parallel configurations): one [2, 1024] message through
    mv_allreduce_addnorm and mv_allgather_rows, eager and three graph replays; every rank holds the
    rank-ordered fp32 sum rounded once (+ residual, norm) and every rank's slice, bit for bit."""
    from oracle.ref_ops import rmsnorm_residual_ref

    world, rows, n = 8, 2, 1024
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_world8_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, res = q.get(timeout=280)
        got[r] = res
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(got[r]["ipc"] for r in range(world)), "the communicators over mapped buffers did not come up"
    assert all(got[r]["status"] == 0 for r in range(world)), "a rank timed out waiting for a peer"
    outs = {r: [torch.from_numpy(o).view(torch.bfloat16) for o in got[r]["outs"]] for r in range(world)}
    for r in range(1, world):
        for a, b in zip(outs[0], outs[r]):
            assert torch.equal(a, b), "ranks disagree"

    def part_of(rank, seed):
        gg = torch.Generator().manual_seed(1000 * seed + rank)
        return torch.randn(rows, n, generator=gg).to(torch.bfloat16)

    g = torch.Generator().manual_seed(9)
    resid = torch.randn(rows, n, generator=g).to(torch.bfloat16)
    lnw = (1 + 0.1 * torch.randn(n, generator=g)).to(torch.bfloat16)
    it = iter(outs[0])
    for seed in range(4):
        acc = torch.zeros(rows, n, dtype=torch.float32)
        for r in range(world):                   # rank order, fp32, one rounding
            acc += part_of(r, seed).float()
        want_n, want_y = rmsnorm_residual_ref(acc.to(torch.bfloat16), resid, lnw, 1e-5)
        y, nrm, gathered = next(it), next(it), next(it)
        assert torch.equal(y, want_y), seed
        assert (nrm != want_n).float().mean() < 0.005
        assert (nrm.float() - want_n.float()).abs().max() <= 2 ** -7 * want_n.float().abs().max()
        assert torch.equal(gathered, torch.cat([part_of(r, seed) for r in range(world)], dim=1)), seed
