"""CPU restatement of the reference's decode and prefill runners (test infrastructure).

Follows core/layer/model_graph.py:87-223 (prepare_weights + decode, explicit-kernel route) and
core/layer/model_prefill.py:23-62.  The reference's own runners cannot execute on a CPU (their
ops assert CUDA and flash_attn is absent), so this restatement is what "the reference's eager
CPU path" means wherever it is timed or compared against.

Tensor parallelism is simulated in-process: `world` ranks are evaluated one after the other and
their partial outputs summed where the reference calls all_reduce (model_graph.py:210,219); the
sum is taken in fp32 and rounded to bf16 once.
"""
import torch

from .ref_attention import flash_attn_with_kvcache_ref
from .ref_ops import (BF16, linear_ref, rmsnorm_ref, rmsnorm_residual_ref, rope_prefill_ref,
                      rotary_tables_ref, swiglu_ref)


class RefShard:
    """One rank's weights, sliced as model_graph.py:95-116 (kept [out,in]; linear_ref transposes)."""

    def __init__(self, model, rank, world, head_size, fp32_weights=False):
        cfg = model.config
        gh = cfg.num_attention_heads
        gkv = getattr(cfg, "num_key_value_heads", gh)
        q_dim, kv_dim = gh * head_size, gkv * head_size
        cv = (lambda t: t.float()) if fp32_weights else (lambda t: t)
        self.layers = []
        for block in model.transformer.h:
            w1, w2 = block.mlp.w1.weight, block.mlp.w2.weight
            gu = torch.cat([w1.chunk(world, 0)[rank], w2.chunk(world, 0)[rank]], 0)   # :98-101
            d = block.mlp.c_proj.weight.chunk(world, 1)[rank]                          # :102
            o = block.attn.c_proj.weight.chunk(world, 1)[rank]                         # :105
            wq, wk, wv = block.attn.c_attn.weight.split([q_dim, kv_dim, kv_dim], 0)    # :109
            qkv = torch.cat([w.chunk(world, 0)[rank] for w in (wq, wk, wv)], 0)        # :110-111
            b = None
            if block.attn.c_attn.bias is not None:                                     # :113-116
                bq, bk, bv = block.attn.c_attn.bias.split([q_dim, kv_dim, kv_dim], 0)
                b = torch.cat([x.chunk(world, 0)[rank] for x in (bq, bk, bv)], 0)
            self.layers.append(dict(gu=cv(gu), d=cv(d), o=cv(o), qkv=cv(qkv), qkv_b=b))


class RefRunner:
    def __init__(self, model, head_size, world=1, max_blocks=32, fp32_weights=False, p_bf16=False):
        """p_bf16: round the softmax probabilities to bf16 before P.V, as the tensor-core kernels
        of the flash-attn package do internally.  Off by default: an eager restatement of
        attention is softmax(QK^T)V in fp32, and that internal rounding is not reproducible by
        any other kernel (its realisation depends on the package's tile order)."""
        self.p_bf16 = p_bf16
        cfg = model.config
        self.model, self.world, self.D = model, world, head_size
        self.hq = cfg.num_attention_heads // world          # core/engine.py:112
        self.hkv = getattr(cfg, "num_key_value_heads", cfg.num_attention_heads) // world  # :113
        self.shards = [RefShard(model, r, world, head_size, fp32_weights) for r in range(world)]
        # core/paged_attention.py:172-190: 256 positions per block regardless of block_size
        self.cos, self.sin = rotary_tables_ref(head_size, max_blocks * 256)
        self.lm_head = model.lm_head.weight.float() if fp32_weights else model.lm_head.weight
        self.timing = None

    @staticmethod
    def _allreduce(parts):
        if len(parts) == 1:
            return parts[0]
        return torch.stack([p.float() for p in parts]).sum(0).to(BF16)

    def _split_qkv(self, qkv, lead):
        hq, hkv, D = self.hq, self.hkv, self.D
        q = qkv[..., :hq * D].reshape(*lead, hq, D)
        k = qkv[..., hq * D:(hq + hkv) * D].reshape(*lead, hkv, D)
        v = qkv[..., (hq + hkv) * D:].reshape(*lead, hkv, D)
        return q, k, v

    def decode_layer(self, h, li, caches, seqlens, block_table, parts_out=None):
        """One block of model_graph.py:202-219 on hidden state h [bs, hid] bf16.  `parts_out` (a
        dict) receives what each rank holds before the two all-reduces and their sums:
        "o" / "mlp" = lists of per-rank partials, "o_sum" / "mlp_sum" = the reduced tensors."""
        block = self.model.transformer.h[li]
        bs = h.shape[0]
        normed = rmsnorm_ref(h, block.ln_1.weight, block.ln_1.eps)                 # :143
        parts = []
        for r, sh in enumerate(self.shards):
            w = sh.layers[li]
            qkv = linear_ref(normed, w["qkv"], w["qkv_b"])                         # :145-147
            q, k, v = self._split_qkv(qkv, (bs,))
            attn = flash_attn_with_kvcache_ref(                                    # :161-168
                q.unsqueeze(1), caches[r][0][li], caches[r][1][li], k.unsqueeze(1),
                v.unsqueeze(1), self.cos, self.sin, seqlens, block_table, causal=True,
                p_bf16=self.p_bf16).squeeze(1)
            parts.append(linear_ref(attn.reshape(bs, -1), w["o"]))                 # :171
        attn_out = self._allreduce(parts)                                          # :210
        if parts_out is not None:
            parts_out["o"], parts_out["o_sum"] = parts, attn_out
        normed, residual = rmsnorm_residual_ref(attn_out, h, block.ln_2.weight, block.ln_2.eps)  # :213
        parts = []
        for sh in self.shards:
            w = sh.layers[li]
            gu = linear_ref(normed, w["gu"])                                       # :183
            parts.append(linear_ref(swiglu_ref(gu), w["d"]))                       # :184-185
        mlp_out = self._allreduce(parts)
        if parts_out is not None:
            parts_out["mlp"], parts_out["mlp_sum"] = parts, mlp_out
        return (mlp_out.float() + residual.float()).to(BF16)                       # :219

    def head(self, h, fp32=False):
        """model_graph.py:222-223; fp32=True returns the logits before their bf16 rounding."""
        tr = self.model.transformer
        hn = rmsnorm_ref(h, tr.ln_f.weight, tr.ln_f.eps)
        if fp32:
            return hn.float() @ self.lm_head.float().t()
        return linear_ref(hn, self.lm_head)

    def decode(self, input_ids, caches, seqlens, block_table, trace=None, parts_trace=None):
        """model_graph.py:192-223.  caches[r] = (k_caches, v_caches) lists per layer for rank r;
        seqlens int32 [bs]; block_table int32 [bs, max_blocks].  Returns logits [bs,V] bf16.
        `trace` (a list) receives the hidden state entering each layer and the final one;
        `parts_trace` (a list) one `parts_out` dict of decode_layer per layer."""
        h = self.model.transformer.wte.weight[input_ids]                           # :196
        for li in range(len(self.model.transformer.h)):
            if trace is not None:
                trace.append(h)
            po = None
            if parts_trace is not None:
                po = {}
                parts_trace.append(po)
            h = self.decode_layer(h, li, caches, seqlens, block_table, parts_out=po)
        if trace is not None:
            trace.append(h)
        return self.head(h)

    def prefill(self, input_ids, caches, block_table, trace=None):
        """model_prefill.py:23-62: fixed-length batch [B,S], RoPE outside the attention call with
        positions 0..S-1, cache_seqlens zeroed, logits for every position.  `trace` (a list)
        receives the hidden state [B,S,hid] entering each layer and the final one."""
        tr = self.model.transformer
        B, S = input_ids.shape
        h = tr.wte.weight[input_ids]                                               # :29
        zeros = torch.zeros(B, dtype=torch.int32)                                  # :32
        for li, block in enumerate(tr.h):
            if trace is not None:
                trace.append(h)
            normed = rmsnorm_ref(h, block.ln_1.weight, block.ln_1.eps)             # :41
            parts = []
            for r, sh in enumerate(self.shards):
                w = sh.layers[li]
                qkv = linear_ref(normed, w["qkv"], w["qkv_b"])                     # :42-43
                q, k, v = self._split_qkv(qkv, (B, S))
                q, k = rope_prefill_ref(q, k, self.cos, self.sin)                  # :46
                attn = flash_attn_with_kvcache_ref(                                # :49-52
                    q, caches[r][0][li], caches[r][1][li], k, v, None, None, zeros, block_table,
                    causal=True, p_bf16=self.p_bf16)
                parts.append(linear_ref(attn.reshape(B, S, -1), w["o"]))           # :54
            out = self._allreduce(parts)
            normed, residual = rmsnorm_residual_ref(out, h, block.ln_2.weight, block.ln_2.eps)  # :55
            parts = []
            for sh in self.shards:
                w = sh.layers[li]
                gu = linear_ref(normed, w["gu"])                                   # :58 (_mlp)
                parts.append(linear_ref(swiglu_ref(gu), w["d"]))
            h = (self._allreduce(parts).float() + residual.float()).to(BF16)       # :59
        if trace is not None:
            trace.append(h)
        h = rmsnorm_ref(h, tr.ln_f.weight, tr.ln_f.eps)                            # :61
        return linear_ref(h, self.lm_head)                                         # :62
