"""Prefill tokens/s of the 8B shape (random weights): python tools/bench_prefill.py [S] [B] [layers]
Prints whole-forward time plus a per-op breakdown of one layer (HIP events)."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import micro_vllm_amd  # noqa: F401
from micro_vllm_amd.core.cache_manager import KVCacheManager
from micro_vllm_amd.core.layer.model_prefill import ModelPrefillRunner
from micro_vllm_amd.models import SHAPES, build_synthetic_model

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
layers = int(sys.argv[3]) if len(sys.argv) > 3 else 32
dev = "cuda"
cfg = SHAPES["llama3-8b"]
model = build_synthetic_model(cfg, dev, seed=1234, num_layers=layers)
blocks = (S + 255) // 256 + 1
runner = ModelPrefillRunner(model, layers, cfg.num_attention_heads, cfg.head_dim, cfg.num_key_value_heads,
                            cfg.hidden_size, cfg.intermediate_size, dev, max_batch_size=B,
                            max_blocks=max(32, blocks), release_source_weights=True)
mgr = KVCacheManager(B * blocks + 2, 256, layers, cfg.num_key_value_heads, cfg.head_dim,
                     dtype=torch.bfloat16, device=dev, max_tokens=blocks * 256, max_batch_size=B)
order = list(mgr._free); random.Random(2).shuffle(order); mgr._free.clear(); mgr._free.extend(order)
for sid in range(B):
    assert mgr.alloc(sid, S)[0]
mgr.cache_batch_data(list(range(B)), [S] * B)
ids = torch.randint(0, cfg.vocab_size, (B, S), device=dev)

def timed(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3

t = timed(lambda: runner.forward(ids, mgr, B, last_token_only=True))
T = B * S
hid, I, V, L = cfg.hidden_size, cfg.intermediate_size // 2, cfg.vocab_size, layers
qkv_dim = (cfg.num_attention_heads + 2 * cfg.num_key_value_heads) * cfg.head_dim
flops_gemm = 2 * T * L * (hid * qkv_dim + hid * hid + 3 * hid * I)
flops_attn = 4 * L * cfg.num_attention_heads * cfg.head_dim * B * S * S / 2
print(f"prefill B={B} S={S} layers={L}: {t*1e3:.2f} ms  {T/t:,.0f} tok/s  "
      f"{(flops_gemm+flops_attn)/t/1e12:.0f} TFLOP/s ({(flops_gemm+flops_attn)/t/2.5e15*100:.1f}% of 2.5 PF)", flush=True)

# per-op breakdown of layer 0
from micro_vllm_amd.core.prefill_attention import paged_attn_prefill_rows, rope_store_prefill
from micro_vllm_amd.kernel.linear import gemm_prefill
from micro_vllm_amd.kernel.rmsnorm_add import rmsnorm, rmsnorm_residual_fused
blk = model.transformer.h[0]
hq, hkv, d = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
h = torch.randn(T, hid, device=dev).to(torch.bfloat16)
lens = mgr._cache_seqlens_buffer[:B].zero_(); bt = mgr._block_table_buffer[:B]
kc, vc = mgr.get(0)
qkv = torch.empty(T, qkv_dim, dtype=torch.bfloat16, device=dev)
attn = torch.empty(T, hq * d, dtype=torch.bfloat16, device=dev)
o = torch.empty(T, hid, dtype=torch.bfloat16, device=dev)
act = torch.empty(T, I, dtype=torch.bfloat16, device=dev)
ops = [
    ("rmsnorm", lambda: rmsnorm(h, blk.ln_1.weight, 1e-5), 0),
    ("qkv gemm", lambda: gemm_prefill(h, blk.attn._qkv_w, qkv), 2 * T * hid * qkv_dim),
    ("rope+store", lambda: rope_store_prefill(qkv, B, S, hq, hkv, d, kc, vc, lens, bt, runner.attention._cos_pool, runner.attention._sin_pool), 0),
    ("attention", lambda: paged_attn_prefill_rows(qkv[:, :hq * d], B, S, hq, d, kc, vc, lens, bt, attn), 4 * hq * d * B * S * S / 2),
    ("o gemm", lambda: gemm_prefill(attn, blk.attn._o, o), 2 * T * hid * hid),
    ("add+rmsnorm", lambda: rmsnorm_residual_fused(o, h, blk.ln_2.weight, 1e-5), 0),
    ("gate/up gemm+swiglu", lambda: gemm_prefill(h, blk.mlp._gu, act, swiglu=True), 2 * T * hid * 2 * I),
    ("down gemm", lambda: gemm_prefill(act, blk.mlp._d, o, residual=h), 2 * T * hid * I),
]
tot = 0
for name, fn, fl in ops:
    tt = timed(fn, reps=10)
    tot += tt
    extra = f"  {fl/tt/1e12:7.0f} TFLOP/s" if fl else ""
    print(f"  {name:22s} {tt*1e6:9.1f} us{extra}", flush=True)
print(f"  layer sum {tot*1e6:.1f} us -> x{L} = {tot*L*1e3:.2f} ms")
