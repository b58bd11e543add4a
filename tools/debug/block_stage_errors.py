"""Where one block of the decode step at Llama-3-8B widths differs from the CPU restatement, stage
by stage (teacher-forced input): prints max / mean abs error of qkv, attention output, residual sum,
SwiGLU output and block output."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
import micro_vllm_amd  # noqa
from test_gpu_full_size import _build
from oracle.ref_ops import linear_ref, rmsnorm_ref, rmsnorm_residual_ref, swiglu_ref, BF16
from oracle.ref_attention import flash_attn_with_kvcache_ref

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cfg, runner, mgr, ref, ref_cache, lens = _build(bs)
DEV = "cuda"
sids = list(range(bs))
ids = torch.randint(0, cfg.vocab_size, (bs,), generator=torch.Generator().manual_seed(40 + bs))
for sid in sids:
    mgr.append(sid); ref_cache.append(sid)
pos = [L + 1 for L in lens]
mgr.cache_batch_data(sids, pos)
rbt, rsl = ref_cache.cache_batch_data(sids, pos)
h = ref.model.transformer.wte.weight[ids]
block = ref.model.transformer.h[0]
w = ref.shards[0].layers[0]
normed = rmsnorm_ref(h, block.ln_1.weight, block.ln_1.eps)
qkv = linear_ref(normed, w["qkv"], w["qkv_b"])
q, k, v = ref._split_qkv(qkv, (bs,))
attn = flash_attn_with_kvcache_ref(q.unsqueeze(1), ref_cache.k_caches[0], ref_cache.v_caches[0], k.unsqueeze(1),
                                   v.unsqueeze(1), ref.cos, ref.sin, rsl, rbt, causal=True, p_bf16=False).squeeze(1)
o = linear_ref(attn.reshape(bs, -1), w["o"])
normed2, residual = rmsnorm_residual_ref(o, h, block.ln_2.weight, block.ln_2.eps)
gu = linear_ref(normed2, w["gu"])
act = swiglu_ref(gu)
mlp = linear_ref(act, w["d"])
out = (mlp.float() + residual.float()).to(BF16)

runner._hidden0[:bs] = h.to(DEV)
got = runner._layer(runner._hidden0[:bs], 0, bs, mgr)
torch.cuda.synchronize()
def rep(name, g, wnt):
    g, wnt = g.float().cpu().reshape(wnt.shape), wnt.float()
    e = (g - wnt).abs()
    ulp = (e / (2.0 ** -8 * wnt.abs().clamp_min(1e-6)))
    print(f"{name:10s} max|x| {wnt.abs().max():8.3f} mean|x| {wnt.abs().mean():8.4f}  max err {e.max():.3e} mean err {e.mean():.3e}  frac != {float((g != wnt).float().mean()):.4f}")
rep("qkv", runner._qkv[:bs], qkv)
rep("attn", runner._attn[:bs], attn)
rep("residual", runner._residual[:bs], residual)
rep("act", runner._swiglu_out[:bs], act)
rep("mlp(f32)", got, out)
print("h", h.abs().max(), "o", o.abs().max(), "mlp", mlp.abs().max(), mlp.abs().mean())
