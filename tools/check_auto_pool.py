"""8B-shaped engine with the KV pool sized from free device memory: pool size, set-up time, and a
short generation over it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import micro_vllm_amd  # noqa: F401
from micro_vllm_amd.core.engine import InferenceEngine
from micro_vllm_amd.models import SHAPES, build_synthetic_model


class Tok:
    eos_token_id = 1
    def encode(self, prompt, add_special_tokens=True):
        return [5 + (ord(c) * 7919) % 100000 for c in prompt]
    def decode(self, ids, skip_special_tokens=True):
        return " ".join(str(i) for i in ids)


cfg = SHAPES["llama3-8b"]
model = build_synthetic_model(cfg, "cuda", seed=1234)
t0 = time.time()
eng = InferenceEngine(model=model, tokenizer=Tok(), max_batch_size=8, n_blocks="auto", max_tokens=8192,
                      graph_batch_sizes=[1, 8], prefill_timeout=0.0)
torch.cuda.synchronize()
free, total = torch.cuda.mem_get_info()
st = eng.cache_manager.stats
print(f"pool: {st['free']} pages = {st['free'] * 256 / 1e6:.2f} M tokens, engine set up in {time.time() - t0:.1f} s, "
      f"device memory free after it {free / 2**30:.1f} of {total / 2**30:.1f} GiB", flush=True)
t0 = time.time()
out = eng.generate(["x" * 1500 + str(i) for i in range(8)], max_tokens=32, temperature=0.0)
assert len(out) == 8 and all(len(v.split()) == 1501 + 32 for v in out.values()), {k[-2:]: len(v.split()) for k, v in out.items()}
print(f"8 prompts of 1501 tokens + 32 new: {time.time() - t0:.2f} s, pages free again: {eng.cache_manager.stats['free']}")
