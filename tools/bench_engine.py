"""InferenceEngine end to end (scheduler -> prefill -> graph decode -> sampler -> update_sequences) on a
Llama-3-8B-shaped random-weight model: N requests of PROMPT tokens, GEN new tokens each.
   python tools/bench_engine.py [N=32] [PROMPT=512] [GEN=128] [TEMPERATURE=0]
Reports generated tokens/s over the whole job and the split between prefill and decode steps, so
that host time per step (Python, the token read-back) can be read off against the device time
of the same step (bench.py --bs N)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import micro_vllm_amd  # noqa: F401
from micro_vllm_amd.core.engine import InferenceEngine
from micro_vllm_amd.models import SHAPES, build_synthetic_model

n, plen, gen = (int(a) for a in (sys.argv[1:4] + ["32", "512", "128"][len(sys.argv[1:4]):]))
temp = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0


class Tok:
    """One token per character; no end-of-sequence token among them."""
    eos_token_id = 1

    def __init__(self, vocab):
        self.vocab = vocab

    def encode(self, prompt, add_special_tokens=True):
        return [5 + (ord(c) * 7919) % (self.vocab - 5) for c in prompt]

    def decode(self, ids, skip_special_tokens=True):
        return " ".join(str(i) for i in ids)


cfg = SHAPES["llama3-8b"]
model = build_synthetic_model(cfg, "cuda", seed=1234)
blocks = n * ((plen + gen + 255) // 256 + 1) + 4
eng = InferenceEngine(model=model, tokenizer=Tok(cfg.vocab_size), max_batch_size=max(n, 1),
                      max_prefill_tokens=max(2048, plen), n_blocks=blocks, max_tokens=plen + gen + 256,
                      graph_batch_sizes=[b for b in (1, 2, 4, 8, 16, 32, 40) if b <= n] or [n],
                      prefill_timeout=0.0)
prompts = ["".join(chr(33 + (i * 31 + j * 7) % 90) for j in range(plen)) for i in range(n)]

kinds = {"prefill": [0, 0.0, []], "decode": [0, 0.0, []]}
orig_step = eng.step
def timed_step(ctx):
    torch.cuda.synchronize()
    t = time.perf_counter()
    r = orig_step(ctx)
    torch.cuda.synchronize()
    k = kinds[ctx.batch_type if ctx.batch_type in kinds else "decode"]
    k[0] += 1
    k[1] += time.perf_counter() - t
    k[2].append(time.perf_counter() - t)
    return r
eng.step = timed_step

for rep in range(2):
    for k in kinds.values():
        k[0], k[1], k[2] = 0, 0.0, []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = eng.generate(prompts, max_tokens=gen, temperature=temp, top_p=0.9)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ntok = sum(len(v.split()) for v in out.values()) - n * plen   # the decoded text holds the prompt too
    print(f"rep {rep}: {len(out)} requests, {ntok} tokens generated in {dt*1e3:.0f} ms = {ntok/dt:.0f} tok/s "
          f"(prompt {plen}, temperature {temp})")
    for name, (cnt, tt, each) in kinds.items():
        if cnt:
            each = sorted(each)
            print(f"    {name}: {cnt} steps, {tt*1e3:.0f} ms in step() = {tt/cnt*1e3:.2f} ms/step "
                  f"(median {each[len(each)//2]*1e3:.2f}, slowest {each[-1]*1e3:.2f})")
    print(f"    outside step() (scheduler, update_sequences): {(dt - sum(k[1] for k in kinds.values()))*1e3:.0f} ms")
