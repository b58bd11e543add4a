"""A/B inside one process of the decode step for more than four rows: attention adding up the QKV
projection's k-slabs itself (runner.qkv_slabs) against the projection reducing them in its own
launch.  Two runners on one weight image, a KV pool and a captured graph each, timed in
alternation so that clock and temperature drift hit both alike:
    python tools/ab_qkv_slabs.py [bs] [ctx] [rounds]"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import micro_vllm_amd  # noqa: F401
from micro_vllm_amd.core.cache_manager import KVCacheManager
from micro_vllm_amd.core.layer.model_graph import ModelGraphRunner
from micro_vllm_amd.models import SHAPES, build_synthetic_model

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ctx = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 6
steps = 30
dev = "cuda"
cfg = SHAPES["llama3-8b"]
layers = cfg.num_hidden_layers
model = build_synthetic_model(cfg, dev, seed=1234, num_layers=layers)
bps = (ctx + rounds * steps + 64 + 255) // 256 + 1
legs = []
for slabs in (True, False):
    r = ModelGraphRunner(model, layers, cfg.num_attention_heads, cfg.head_dim, cfg.num_key_value_heads,
                         cfg.hidden_size, cfg.intermediate_size, dev, max_batch_size=bs, max_blocks=max(32, bps),
                         release_source_weights=True)
    r.qkv_slabs = slabs
    m = KVCacheManager(bs * bps + 2, 256, layers, cfg.num_key_value_heads, cfg.head_dim, dtype=torch.bfloat16,
                       device=dev, max_tokens=bps * 256, max_batch_size=bs)
    order = list(m._free); random.Random(2).shuffle(order); m._free.clear(); m._free.extend(order)
    g = torch.Generator(device=dev); g.manual_seed(1)
    m._k_store.normal_(0.0, 1.0, generator=g); m._v_store.normal_(0.0, 1.0, generator=g)
    for sid in range(bs):
        assert m.alloc(sid, ctx)[0]
    r.capture(m, batch_sizes=[bs])
    legs.append(dict(name="slabs" if slabs else "own launch", r=r, m=m, pos=1,
                     ids=torch.randint(0, cfg.vocab_size, (bs,), generator=torch.Generator().manual_seed(0)).to(dev), t=[]))

def run(leg, n):
    r, m = leg["r"], leg["m"]
    for _ in range(n):
        for sid in range(bs):
            m.append(sid)
        m.cache_batch_data(list(range(bs)), [ctx + leg["pos"]] * bs)
        leg["ids"] = r.greedy_ids(r.forward(leg["ids"], m, bs))
        leg["pos"] += 1

for leg in legs:
    run(leg, 5)
torch.cuda.synchronize()
for rd in range(rounds):
    for leg in (legs if rd % 2 == 0 else legs[::-1]):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        run(leg, steps)
        torch.cuda.synchronize(); leg["t"].append((time.perf_counter() - t0) / steps * 1e3)
for leg in legs:
    ts = sorted(leg["t"])
    print(f"{leg['name']:12s} ms/step: median {ts[len(ts) // 2]:.4f}  min {ts[0]:.4f}  all {' '.join(f'{t:.3f}' for t in leg['t'])}", flush=True)
# the two routes are the same arithmetic: same tokens
assert torch.equal(legs[0]["ids"], legs[1]["ids"]), "the two routes picked different tokens"
