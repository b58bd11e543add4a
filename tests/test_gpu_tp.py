"""The decode runner's tensor-parallel wiring with 2 and 4 ranks sharing the one GPU of the test box:
HIP kernels compute each rank's shard; the two reductions per block go through the one-shot
all-reduce over IPC-mapped buffers (the path ranks on separate GPUs take: mv_allreduce_addnorm, slabs
in, add + norm fused) or, with MVLLM_ALLREDUCE=dist, through torch.distributed (gloo here, RCCL needs
one GPU per rank) followed by the norm kernels.  Checks slicing, the reduction, the captured step,
and that all ranks end with the oracle's logits."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu
LENS = [3, 300, 40, 255, 17, 64, 500, 129, 9, 31, 257, 100, 7, 450, 64, 200]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, shape, fused, bs, mode, graph, out_q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MVLLM_DIST_BACKEND="gloo",
                      MVLLM_ALLREDUCE=mode, MVLLM_ALLREDUCE_TIMEOUT_MS="20000")
    sys.path.insert(0, ROOT)
    import micro_vllm_amd  # noqa: F401
    from micro_vllm_amd.core import parallel_config as pc
    from micro_vllm_amd.core.cache_manager import KVCacheManager
    from micro_vllm_amd.core.layer.model_graph import ModelGraphRunner
    from micro_vllm_amd.kernel.allreduce import IpcAllReduce
    from micro_vllm_amd.models import SHAPES, build_synthetic_model

    torch.cuda.set_device(0)
    pc.setup()
    dev = "cuda"
    cfg = SHAPES[shape]
    model = build_synthetic_model(cfg, "cpu", seed=1234, qkv_bias=True).to(dev)
    hkv = cfg.num_key_value_heads // world
    n_pages = 2 * bs + 2
    runner = ModelGraphRunner(model, cfg.num_hidden_layers, cfg.num_attention_heads // world,
                              cfg.head_dim, hkv, cfg.hidden_size, cfg.intermediate_size // world, dev,
                              max_batch_size=16, fused=fused)
    assert runner.world_size == world and runner.rank == rank
    assert isinstance(runner._reducer, IpcAllReduce) == (mode == "ipc")
    # next to the one-shot all-reduce the LM head is cut by vocabulary rows and the logits are gathered
    assert (runner._gatherer is not None) == (mode == "ipc")
    mgr = KVCacheManager(n_pages, 256, cfg.num_hidden_layers, hkv, cfg.head_dim, dtype=torch.bfloat16,
                         device=dev, max_tokens=1024, max_batch_size=16)
    g = torch.Generator().manual_seed(5)
    lens = LENS[:bs]
    for li in range(cfg.num_hidden_layers):
        fk = torch.randn(n_pages, 256, cfg.num_key_value_heads, cfg.head_dim, generator=g).to(torch.bfloat16)
        fv = torch.randn(n_pages, 256, cfg.num_key_value_heads, cfg.head_dim, generator=g).to(torch.bfloat16)
        mgr.k_caches[li].copy_(fk[:, :, rank * hkv:(rank + 1) * hkv])
        mgr.v_caches[li].copy_(fv[:, :, rank * hkv:(rank + 1) * hkv])
    for sid, L in enumerate(lens):
        mgr.alloc(sid, L)
        mgr.append(sid)
    mgr.cache_batch_data(list(range(bs)), [L + 1 for L in lens])
    ids = torch.arange(bs, dtype=torch.long) * 13 + 5
    runner._input_ids[:bs] = ids.to(dev)
    logits = runner.decode(runner._input_ids[:bs], bs, mgr).float().cpu()
    replay_equal = None
    if graph:
        # the same step captured and replayed: the new key lands on the same slot with the same bits
        runner.capture(mgr, batch_sizes=[bs])
        assert bs in runner._graphs
        for _ in range(3):
            again = runner.forward(ids.to(dev), mgr, bs).float().cpu()
        replay_equal = bool(torch.equal(again, logits))
    bt = mgr._block_table_buffer[:bs].cpu()
    status = runner._reducer.status()
    if runner._gatherer is not None:
        status = max(status, runner._gatherer.status())
    pc.barrier()
    out_q.put((rank, logits.numpy(), bt.numpy(), replay_equal, status))
    runner._reducer.close()
    import torch.distributed as dist
    dist.destroy_process_group()


@pytest.mark.parametrize("world,shape,fused,bs,mode,graph", [
    (2, "tiny-gqa", True, 2, "ipc", True),      # bs <= 4: the dot-product projections leave the slabs
    (2, "tiny-gqa", True, 16, "ipc", True),     # the configuration's batch size, matrix-core projections
    (4, "tiny-mha", True, 6, "ipc", True),
    (2, "tiny-gqa", False, 2, "ipc", False),    # the reference's separate ops with the plain one-shot sum
    (2, "tiny-gqa", True, 6, "dist", False),    # torch.distributed + norm kernels behind the same interface
    (2, "tiny-gqa", False, 2, "dist", False),
])
def test_ranks_on_one_gpu_match_the_oracle(world, shape, fused, bs, mode, graph):
    sys.path.insert(0, ROOT)
    from micro_vllm_amd.models import SHAPES, build_synthetic_model
    from oracle.ref_model import RefRunner

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, shape, fused, bs, mode, graph, q))
             for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, lg, bt, replay_equal, status = q.get(timeout=300)
        got[r] = (torch.from_numpy(lg), torch.from_numpy(bt))
        assert status == 0, "a rank timed out waiting for a peer"
        if graph:
            assert replay_equal, f"rank {r}: captured step differs from the eager one"
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(1, world):
        assert torch.equal(got[0][0], got[r][0]), "ranks disagree"
        assert torch.equal(got[0][1], got[r][1])

    cfg = SHAPES[shape]
    model = build_synthetic_model(cfg, "cpu", seed=1234, qkv_bias=True)
    g = torch.Generator().manual_seed(5)
    n_pages = 2 * bs + 2
    fks, fvs = [], []
    for li in range(cfg.num_hidden_layers):
        fks.append(torch.randn(n_pages, 256, cfg.num_key_value_heads, cfg.head_dim, generator=g).to(torch.bfloat16))
        fvs.append(torch.randn(n_pages, 256, cfg.num_key_value_heads, cfg.head_dim, generator=g).to(torch.bfloat16))
    hkv = cfg.num_key_value_heads // world
    caches = [([t[:, :, r * hkv:(r + 1) * hkv].contiguous() for t in fks],
               [t[:, :, r * hkv:(r + 1) * hkv].contiguous() for t in fvs]) for r in range(world)]
    lens = LENS[:bs]
    sl = torch.tensor([L + 1 for L in lens], dtype=torch.int32)
    ids = torch.arange(bs, dtype=torch.long) * 13 + 5
    want = RefRunner(model, cfg.head_dim, world=world).decode(ids, caches, sl, got[0][1]).float()
    err = (got[0][0] - want).abs()
    tol = 2.0 ** -7 * float(want.abs().max()) + 2 ** -7 * want.abs()
    assert bool((err <= tol).all()), f"max err {err.max().item()}"
    assert err.mean().item() <= 2e-3
    assert bool((got[0][0].argmax(-1) == want.argmax(-1)).all())


def _prefill_worker(rank, world, port, S, out_q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MVLLM_DIST_BACKEND="gloo")
    sys.path.insert(0, ROOT)
    import micro_vllm_amd  # noqa: F401
    from micro_vllm_amd.core import parallel_config as pc
    from micro_vllm_amd.core.cache_manager import KVCacheManager
    from micro_vllm_amd.core.layer.model_prefill import ModelPrefillRunner
    from micro_vllm_amd.models import SHAPES, build_synthetic_model

    torch.cuda.set_device(0)
    pc.setup()
    dev = "cuda"
    cfg = SHAPES["tiny-gqa"]
    model = build_synthetic_model(cfg, "cpu", seed=1234, qkv_bias=True).to(dev)
    hkv = cfg.num_key_value_heads // world
    runner = ModelPrefillRunner(model, cfg.num_hidden_layers, cfg.num_attention_heads // world,
                                cfg.head_dim, hkv, cfg.hidden_size, cfg.intermediate_size // world, dev,
                                max_batch_size=2)
    mgr = KVCacheManager(8, 256, cfg.num_hidden_layers, hkv, cfg.head_dim, dtype=torch.bfloat16,
                         device=dev, max_tokens=1024, max_batch_size=2)
    for sid in range(2):
        assert mgr.alloc(sid, S)[0]
    mgr.cache_batch_data([0, 1], [S, S])
    ids = torch.randint(0, cfg.vocab_size, (2, S), generator=torch.Generator().manual_seed(11))
    logits = runner.forward(ids.to(dev), mgr, 2).float().cpu()
    bt = mgr._block_table_buffer[:2].cpu()
    pc.barrier()
    out_q.put((rank, logits.numpy(), bt.numpy()))
    import torch.distributed as dist
    dist.destroy_process_group()


def test_two_rank_prefill_matches_the_oracle():
    """ModelPrefillRunner under TP=2: column/row-parallel matrix-core GEMMs, per-rank kv heads and
    pages, all-reduce of [B*S, hid] after O and after down."""
    sys.path.insert(0, ROOT)
    from micro_vllm_amd.models import SHAPES, build_synthetic_model
    from oracle.ref_model import RefRunner

    S = 70
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_prefill_worker, args=(r, 2, port, S, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        r, lg, bt = q.get(timeout=300)
        got[r] = (torch.from_numpy(lg), torch.from_numpy(bt))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert torch.equal(got[0][0], got[1][0]), "ranks disagree"

    cfg = SHAPES["tiny-gqa"]
    model = build_synthetic_model(cfg, "cpu", seed=1234, qkv_bias=True)
    hkv = cfg.num_key_value_heads // 2
    caches = [([torch.zeros(8, 256, hkv, cfg.head_dim, dtype=torch.bfloat16) for _ in range(cfg.num_hidden_layers)],
               [torch.zeros(8, 256, hkv, cfg.head_dim, dtype=torch.bfloat16) for _ in range(cfg.num_hidden_layers)])
              for _ in range(2)]
    ids = torch.randint(0, cfg.vocab_size, (2, S), generator=torch.Generator().manual_seed(11))
    want = RefRunner(model, cfg.head_dim, world=2, p_bf16=True).prefill(ids, caches, got[0][1]).float()
    err = (got[0][0] - want).abs()
    assert err.mean().item() <= 4e-3
    assert err.max().item() <= 4 * 2.0 ** -7 * float(want.abs().max())


@pytest.mark.gpu
def test_rccl_collectives_are_captured_with_the_decode_step():
    """The path several ranks take (all_reduce after the O and the down projection, step captured
    through torch's graph) on one GPU: a one-rank RCCL group, the runner sharded as rank 0 of 2.
    Runs in a child process so that this process never joins a process group."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "rehearse_rccl_graph.py"), "tiny-d128", "4"],
                       capture_output=True, text=True, timeout=110, cwd=root, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "all_reduce calls per eager step 4 (expected 4)" in r.stdout
    assert "graph captured: True (CUDAGraph)" in r.stdout and "bit for bit: True" in r.stdout
