"""Fixed-length batched prefill on the HIP library: same call surface as the reference's
ModelPrefillRunner (core/layer/model_prefill.py:16-62): `forward(input_ids[B,S], cache_manager,
batch_size) -> logits[B,S,V]`, KV pages filled for positions 0..S-1 of every sequence.

Per layer: rmsnorm -> QKV projection (+bias) -> rotate q,k / append k,v to the pages (one
launch) -> causal attention over the pages -> O projection -> all_reduce -> add+rmsnorm ->
gate/up projection with SwiGLU in its epilogue -> down projection (+residual) -> all_reduce.
All projections run on the matrix cores over the same weight image the decode runner streams,
so both runners share one copy of the weights.
"""
import logging

import torch

from .. import parallel_config as pc
from ..prefill_attention import paged_attn_prefill_rows, rope_store_prefill
from ...kernel.linear import add_, embedding, gemm_prefill, wgemm
from ...kernel.rmsnorm_add import rmsnorm, rmsnorm_residual_fused
from ...kernel.swiglu import swiglu_fused
from .model_graph import ModelGraphRunner
from .rope import RoPE

logger = logging.getLogger(__name__)


class ModelPrefillRunner(ModelGraphRunner):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.rope = RoPE()

    def _fast_mlp(self, x, gu, d, residual=None):
        """swiglu(x @ Wgu) @ Wd, the reference's compiled `_mlp` (model_graph.py:75-81): gate_up
        is rounded to bf16 before the activation, the activation output to bf16 before Wd."""
        T = x.shape[0]
        if gu.swiglu:
            act = gemm_prefill(x, gu, torch.empty(T, gu.out_features, dtype=torch.bfloat16, device=x.device),
                               swiglu=True)
        else:
            act = swiglu_fused(gemm_prefill(x, gu, torch.empty(T, gu.n, dtype=torch.bfloat16, device=x.device)))
        return gemm_prefill(act, d, torch.empty(T, d.n, dtype=torch.bfloat16, device=x.device),
                            residual=residual)

    def _prefill_layer(self, h, layer_idx, B, S, cache_manager, cache_lens, block_table):
        """One transformer block on hidden rows h [B*S, hid] (model_prefill.py:36-59)."""
        T, dev = B * S, h.device
        hq, hkv, d = self.num_heads, self.kv_num_heads, self.head_size
        block = self.model.transformer.h[layer_idx]
        normed = rmsnorm(h, block.ln_1.weight, block.ln_1.eps)
        w_qkv = block.attn._qkv_w
        qkv = gemm_prefill(normed, w_qkv, torch.empty(T, w_qkv.n, dtype=torch.bfloat16, device=dev),
                           bias=block.attn._qkv_b)
        k_cache, v_cache = cache_manager.get(layer_idx)
        rope_store_prefill(qkv, B, S, hq, hkv, d, k_cache, v_cache, cache_lens, block_table,
                           self.attention._cos_pool, self.attention._sin_pool)
        attn = paged_attn_prefill_rows(qkv[:, :hq * d], B, S, hq, d, k_cache, v_cache, cache_lens,
                                       block_table,
                                       torch.empty(T, hq * d, dtype=torch.bfloat16, device=dev))
        w_o = block.attn._o
        out = pc.all_reduce(gemm_prefill(attn, w_o, torch.empty(T, w_o.n, dtype=torch.bfloat16, device=dev)))
        normed, residual = rmsnorm_residual_fused(out, h, block.ln_2.weight, block.ln_2.eps)
        if self.world_size == 1:
            return self._fast_mlp(normed, block.mlp._gu, block.mlp._d, residual=residual)
        return add_(pc.all_reduce(self._fast_mlp(normed, block.mlp._gu, block.mlp._d)), residual)

    def forward(self, input_ids: torch.Tensor, cache_manager, batch_size: int,
                last_token_only: bool = False, last_token_index=None) -> torch.Tensor:
        """input_ids [B,S] (right-padded to a common S by the caller).  Returns logits [B,S,V], or
        [B,1,V] holding only the last position when `last_token_only` (what the engine samples
        from, core/engine.py:233) - that skips S-1 of every S rows of the LM head.
        `last_token_index` (B integers) returns [B,1,V] too, row b taken at position
        last_token_index[b]: the last REAL token of a right-padded prompt instead of the pad
        position the reference samples from."""
        B, S = input_ids.shape
        assert B == batch_size
        T = B * S
        dev = input_ids.device
        tr = self.model.transformer
        h = embedding(tr.wte.weight, input_ids.reshape(-1).contiguous(),
                      torch.empty(T, self.hidden_dim, dtype=torch.bfloat16, device=dev))

        cache_lens = cache_manager._cache_seqlens_buffer[:batch_size].zero_()
        block_table = cache_manager._block_table_buffer[:batch_size]

        for layer_idx in range(self.num_layers):
            h = self._prefill_layer(h, layer_idx, B, S, cache_manager, cache_lens, block_table)

        if last_token_index is not None:
            idx = torch.as_tensor(last_token_index, dtype=torch.long).to(dev)
            assert idx.numel() == B
            h = h.view(B, S, self.hidden_dim)[torch.arange(B, device=dev), idx.clamp(0, S - 1)].contiguous()
            S = 1
        elif last_token_only:
            h = h.view(B, S, self.hidden_dim)[:, -1].contiguous()
            S = 1
        h = rmsnorm(h, tr.ln_f.weight, tr.ln_f.eps)
        rows = h.shape[0]
        logits = torch.empty(rows, self.vocab_size, dtype=torch.bfloat16, device=dev)
        if rows <= 64:
            wgemm(h, self._lm_head, logits, workspace=self._gemm_ws)
        else:
            gemm_prefill(h, self._lm_head, logits)
        return logits.view(B, S, self.vocab_size)
