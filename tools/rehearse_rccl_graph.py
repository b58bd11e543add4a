"""Rehearsal, on ONE GPU, of the decode path several tensor-parallel ranks take: an RCCL process
group of a single rank, the runner told that it is rank 0 of 2 (so it cuts half the heads, calls
all_reduce after the O and the down projection and captures the step through torch's graph, the
way collectives have to be captured), eager step against graph replay.  A sum over one rank is the
identity, so nothing is compared with the full model: what this shows is that RCCL calls are
captured and replayed together with the library's kernels on this software stack."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
import torch
import torch.distributed as dist
import micro_vllm_amd  # noqa: F401
from micro_vllm_amd.core import parallel_config as pc
from micro_vllm_amd.core.cache_manager import KVCacheManager
from micro_vllm_amd.core.layer.model_graph import ModelGraphRunner
from micro_vllm_amd.models import SHAPES, build_synthetic_model

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
pc.setup()
calls = {"n": 0}
real_all_reduce = dist.all_reduce


def counted(t, *a, **k):
    calls["n"] += 1
    return real_all_reduce(t, *a, **k)


dist.all_reduce = counted
shape = sys.argv[1] if len(sys.argv) > 1 else "tiny-d128"
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
cfg = SHAPES[shape]
layers = min(cfg.num_hidden_layers, 4)
model = build_synthetic_model(cfg, "cuda", seed=1, num_layers=layers)
hq, hkv = cfg.num_attention_heads // 2, cfg.num_key_value_heads // 2
mgr = KVCacheManager(n_blocks=4 * bs + 4, block_size=256, n_layers=layers, n_heads=hkv, head_size=cfg.head_dim,
                     dtype=torch.bfloat16, device="cuda", max_tokens=1024, max_batch_size=bs)
from micro_vllm_amd.kernel.allreduce import DistAllReduce


def rccl_sum(t):                        # what pc.all_reduce does with more than one rank
    dist.all_reduce(t, group=dist.group.WORLD)
    return t


# the runner shards as rank 0 of 2 and takes the several-ranks path, reducing through RCCL
runner = ModelGraphRunner(model, layers, hq, cfg.head_dim, hkv, cfg.hidden_size, cfg.intermediate_size // 2,
                          "cuda", max_batch_size=bs, reducer=DistAllReduce(rccl_sum), tp_rank=0, tp_world=2)
assert runner.world_size == 2
lens = [300 + 17 * i for i in range(bs)]
for i, L in enumerate(lens):
    assert mgr.alloc(i, L)[0]
for kc, vc in zip(mgr.k_caches, mgr.v_caches):
    kc.normal_(); vc.normal_()
ids = torch.randint(0, cfg.vocab_size, (bs,), device="cuda")
for i in range(bs):
    mgr.append(i)
mgr.cache_batch_data(list(range(bs)), [L + 1 for L in lens])
runner._input_ids[:bs] = ids
eager = runner.decode(runner._input_ids[:bs], bs, mgr).clone()
n_eager = calls["n"]
runner.capture(mgr, batch_sizes=[bs])
captured = bs in runner._graphs
n_capture = calls["n"] - n_eager
out = runner.forward(ids, mgr, bs).clone()
torch.cuda.synchronize()
print(f"{shape} x{layers} layers, bs {bs}: all_reduce calls per eager step {n_eager} (expected {2 * layers}), "
      f"during warm-up + capture {n_capture}, graph captured: {captured} ({type(runner._graphs.get(bs)).__name__}), "
      f"replay equals eager bit for bit: {torch.equal(out, eager)}, finite: {bool(torch.isfinite(out.float()).all())}")
dist.destroy_process_group()
