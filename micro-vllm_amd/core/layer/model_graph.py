"""Decode runner: one single-token forward over all layers, captured once per batch size into a
hipGraph and replayed.  Same constructor, attributes and methods as the reference's
ModelGraphRunner (core/layer/model_graph.py:25-260): prepare_weights / _alloc_bufs /
_qkv_proj / _flash_attn / _ffn / decode / capture / forward.

Differences that are deliberate:
  * weights are relaid out into the wave-tile format of csrc/wgemm.hip instead of `[in,out]`;
  * q/k/v are split as [Hq*D | Hkv*D | Hkv*D], which equals the reference's
    `reshape(bs, 3, H, D)` when Hkv == Hq and also covers grouped-query shapes;
  * the eager (uncaptured) path uses the real block table, not the all-zero one of
    model_graph.py:155-159;
  * `fused=True` (default) folds bias, SwiGLU and the final residual add into GEMM epilogues
    and, for batches of up to 4 rows, the two norms into the prologue of the GEMM that consumes
    them (5 launches per layer instead of 9); every rounding still happens where the
    reference's separate ops round, and the fused and unfused paths agree bit for bit.
"""
import logging
import os
from typing import Dict, List

import torch

from .. import parallel_config as pc
from ..paged_attention import PagedAttention, paged_attn_decode, paged_attn_decode_slabs, qkv_attn_decode
from ... import _native as nv
from ...kernel.linear import (StreamedWeight, add_, embedding, gemm_workspace,
                              wgemm, wgemm_norm, wgemm_postnorm, wgemm_slabs)
from ...kernel.rmsnorm_add import rmsnorm_
from ...kernel.rmsnorm_residual import rmsnorm_residual_gemm
from ...kernel.swiglu import swiglu_gemm
from .sampler import greedy, sample

logger = logging.getLogger(__name__)


def shard_block_weights(block, num_heads_total, kv_heads_total, head_size, rank, world):
    """One rank's slice of a transformer block, exactly as the reference cuts it
    (core/layer/model_graph.py:95-116): column-parallel QKV and gate/up = rows of the [out,in]
    weights (q, k and v each chunked separately), row-parallel O and down = their columns; the
    QKV bias is sliced like the QKV rows and nothing is added on the row-parallel side.
    Returns [out,in] tensors (views where possible); no device work, usable on CPU."""
    q_dim, kv_dim = num_heads_total * head_size, kv_heads_total * head_size

    def rows(w):
        return w.chunk(world, dim=0)[rank]

    def cols(w):
        return w.chunk(world, dim=1)[rank]

    w_q, w_k, w_v = block.attn.c_attn.weight.split([q_dim, kv_dim, kv_dim], dim=0)
    out = {
        "qkv": torch.cat([rows(w_q), rows(w_k), rows(w_v)], dim=0),
        "qkv_b": None,
        "o": cols(block.attn.c_proj.weight),
        "up": rows(block.mlp.w1.weight),
        "gate": rows(block.mlp.w2.weight),
        "d": cols(block.mlp.c_proj.weight),
    }
    b = block.attn.c_attn.bias
    if b is not None:
        b_q, b_k, b_v = b.split([q_dim, kv_dim, kv_dim], dim=0)
        out["qkv_b"] = torch.cat([rows(b_q), rows(b_k), rows(b_v)], dim=0)
    return out


class HipGraph:
    """An instantiated hipGraph (mv_graph_* of the C ABI)."""

    def __init__(self, handle):
        self.handle = handle

    def replay(self):
        nv.check("mv_graph_launch", nv.lib().mv_graph_launch(self.handle, nv.stream_ptr()))

    def __del__(self):
        try:
            if self.handle:
                nv.lib().mv_graph_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class ModelGraphRunner:
    def __init__(self, model, num_layers: int, num_heads: int, head_size: int, kv_num_heads: int,
                 hidden_dim: int, intermediate_size: int, device: str, max_batch_size: int = 32,
                 dtype: torch.dtype = torch.bfloat16, top_k: int = 1000, fused: bool = True,
                 max_blocks: int = 32, exact_rope: bool = False, release_source_weights: bool = False,
                 reducer=None, tp_rank: int = None, tp_world: int = None):
        """num_heads / kv_num_heads / intermediate_size are the PER-RANK values (already divided
        by the tensor-parallel world size), intermediate_size counting gate and up together,
        exactly as the reference's engine passes them (core/engine.py:108-114).

        `reducer` (kernel/allreduce.py interface) replaces the process group's reduction, and
        `tp_rank` / `tp_world` the process group's rank and size: a single process can then run
        the step of any one rank of a tensor-parallel group (tests, per-rank benchmarks)."""
        assert dtype == torch.bfloat16, "the HIP path computes in bf16 storage / fp32 accumulate"
        self.model = model
        self.num_layers = num_layers
        self.rank, self.world_size = pc.get_rank(), pc.get_world_size()
        if tp_world is not None:
            assert reducer is not None and 0 <= tp_rank < tp_world
            self.rank, self.world_size = tp_rank, tp_world
        self._reducer = reducer
        # batches of more than four rows: attention adds up the QKV projection's k-slabs itself
        # (MVLLM_QKV_SLABS=0: the projection reduces them in a launch of its own, as before)
        self.qkv_slabs = os.environ.get("MVLLM_QKV_SLABS", "1") != "0"
        # batches of 1, 2 or 4 rows on one GPU: QKV projection and attention as one launch whose
        # key/value rows are requested while the projection streams (MVLLM_QKV_ATTN=1).  The same
        # bits either way; off by default because it measured slower on MI355X (3.16 against 3.02 ms
        # per step of the 8B shape): weights and rows share one HBM pipe, so the rows lengthen the
        # projection phase by what they shorten the attention phase, and the hand-over of q costs
        # what the launch boundary did.
        self.qkv_attn = os.environ.get("MVLLM_QKV_ATTN", "0") == "1"
        self._qkv_attn_ok: Dict[tuple, bool] = {}
        self.num_heads = num_heads
        self.kv_num_heads = kv_num_heads
        self.intermediate_size = intermediate_size
        self.head_size = head_size
        self.hidden_dim = hidden_dim
        self.vocab_size = model.config.vocab_size
        self.device = device
        self.max_bs = max_batch_size
        self.dtype = dtype
        self.top_k = top_k
        self.fused = fused
        self.release_source_weights = release_source_weights

        self.attention = PagedAttention(num_heads, head_size, kv_num_heads, device, max_batch_size,
                                        max_blocks=max_blocks, exact_positions=exact_rope,
                                        rope_base=float(getattr(model.config, "rope_theta", 10000.0)))
        self.prepare_weights()
        self._alloc_bufs()
        self._gatherer = None
        if self.world_size > 1 and self._reducer is None:
            self._reducer = pc.tensor_parallel_reducer(min(max_batch_size, 64), hidden_dim)
            if self._lm_head_shard is not None and max_batch_size <= 64:
                # every rank asks (the set-up is a vote among the ranks), whatever it will answer
                self._gatherer = pc.tensor_parallel_gatherer(max_batch_size, self.vocab_size // self.world_size)
        if self._gatherer is not None:
            self._logits_local = torch.empty(max_batch_size, self.vocab_size // self.world_size,
                                             dtype=dtype, device=device)
        self._graphs: Dict[int, HipGraph] = {}
        # second graph per batch size, captured with a finer attention split: replayed when the
        # rows of a step differ much in length (the longest row alone then sets the pace of an
        # unsplit launch: 32 rows uniform in [256, 4096] run 79.7 us unsplit, 70.3 us in two splits)
        self._graphs_uneven: Dict[int, HipGraph] = {}
        self._uneven = False
        # longest row / mean row above which the finer split is used; None = by measurement
        # (tools/sweep_attn_uneven_ratio.py, rows uniform in [lo, 4096]: 32 rows break even at 1.5
        # and gain 12 % at 1.9, below 1.5 two splits LOSE 6-7 %; 64 rows break even near 1.27)
        self.uneven_ratio = None
        self._is_graph_ready = False
        self._num_splits: Dict[int, int] = {}

    # ------------------------------------------------------------------ init
    def prepare_weights(self):
        """Tensor-parallel slicing as in the reference (column-parallel QKV and gate/up: rows of
        the [out,in] weight; row-parallel O and down: its columns), then the one-time relayout
        into wave tiles."""
        cfg = self.model.config
        g_heads = cfg.num_attention_heads
        g_kv = getattr(cfg, "num_key_value_heads", g_heads)
        dev = self.device
        for block in self.model.transformer.h:
            if isinstance(getattr(block.attn, "_qkv_w", None), StreamedWeight):
                continue  # another runner on this model (decode + prefill) already did the relayout
            sh = shard_block_weights(block, g_heads, g_kv, self.head_size, self.rank, self.world_size)
            up, gate = sh["up"].to(dev), sh["gate"].to(dev)
            if self.fused:
                block.mlp._gu = StreamedWeight.from_up_gate(up, gate)
            else:
                block.mlp._gu = StreamedWeight.from_linear(torch.cat([up, gate], dim=0))
            block.mlp._d = StreamedWeight.from_linear(sh["d"].to(dev).contiguous())
            block.attn._o = StreamedWeight.from_linear(sh["o"].to(dev).contiguous())
            block.attn._qkv_w = StreamedWeight.from_linear(sh["qkv"].to(dev))
            block.attn._qkv_b = None if sh["qkv_b"] is None else sh["qkv_b"].to(dev).contiguous()
            block.ln_1.weight = block.ln_1.weight.to(dev)
            block.ln_2.weight = block.ln_2.weight.to(dev)
            if self.release_source_weights:
                block.mlp.w1.weight = block.mlp.w2.weight = block.mlp.c_proj.weight = None
                block.attn.c_attn.weight = block.attn.c_proj.weight = None
        tr = self.model.transformer
        tr.wte.weight = tr.wte.weight.to(dev).contiguous()
        tr.ln_f.weight = tr.ln_f.weight.to(dev)
        # embedding, final norm and LM head are replicated on every rank (model_graph.py:196,222-223)
        shared = getattr(self.model, "_mv_lm_head_tiles", None)
        shards = getattr(self.model, "_mv_lm_head_shards", None)
        if shards is None:
            shards = self.model._mv_lm_head_shards = {}
        # ... and, with several ranks, this rank's slice of the head by vocabulary rows next to it: the
        # decode step multiplies the slice and collects the ranks' logits when a gatherer is available
        # (kernel/allreduce.py::IpcAllGather), the whole head otherwise - decided after the weights
        v_local = self.vocab_size // self.world_size
        key = (self.rank, self.world_size)
        if (self.world_size > 1 and key not in shards and self.vocab_size % (16 * self.world_size) == 0
                and getattr(self.model.lm_head, "weight", None) is not None):
            w = self.model.lm_head.weight[self.rank * v_local:(self.rank + 1) * v_local]
            shards[key] = StreamedWeight.from_linear(w.to(dev).contiguous())
        if shared is None:
            shared = StreamedWeight.from_linear(self.model.lm_head.weight.to(dev))
            self.model._mv_lm_head_tiles = shared
            if self.release_source_weights:
                self.model.lm_head.weight = None
        self._lm_head = shared
        self._lm_head_shard = shards.get(key)
        torch.cuda.synchronize()
        logger.info("weights relaid out (rank %d)", self.rank)

    def _alloc_bufs(self):
        max_b, dev, dt = self.max_bs, self.device, self.dtype
        _b = self.model.transformer.h[0]
        qkv_dim = _b.attn._qkv_w.shape[1]
        o_dim = _b.attn._o.shape[1]
        self._input_ids = torch.zeros(max_b, dtype=torch.long, device=dev)
        self._logits = torch.empty(max_b, self.vocab_size, dtype=dt, device=dev)
        # argmax of every row of the last decode's logits: part of the captured step, so greedy
        # decoding needs no launch of its own after the graph
        self._greedy = torch.zeros(max_b, dtype=torch.long, device=dev)
        # Stochastic pick inside the captured step as well: per-row temperature / top-p / uniform
        # number live in static buffers the engine rewrites between steps; rows with temperature 0
        # (the buffers' initial state) take the greedy id and cost nothing.  `_picked` is what a
        # caller reads back: greedy or sampled, row by row.
        self.decode_top_k = 50       # core/engine.py:272 (`self._sample(logits, temps, topp, 50)`)
        self._samp_temp = torch.zeros(max_b, dtype=torch.float32, device=dev)
        self._samp_topp = torch.ones(max_b, dtype=torch.float32, device=dev)
        self._samp_u = torch.zeros(max_b, dtype=torch.float32, device=dev)
        self._picked = torch.zeros(max_b, dtype=torch.long, device=dev)
        self._hidden0 = torch.empty((max_b, self.hidden_dim), dtype=dt, device=dev)
        self._normed = torch.empty((max_b, self.hidden_dim), dtype=dt, device=dev)
        self._qkv = torch.empty(max_b, qkv_dim, dtype=dt, device=dev)
        self._attn = torch.empty(max_b, self.num_heads, self.head_size, dtype=dt, device=dev)
        self._attn_out = torch.empty(max_b, o_dim, dtype=dt, device=dev)
        self._residual = torch.empty((max_b, self.hidden_dim), dtype=dt, device=dev)
        self._gate_up = torch.empty((max_b, self.intermediate_size), dtype=dt, device=dev)
        self._swiglu_out = torch.empty((max_b, self.intermediate_size // 2), dtype=dt, device=dev)
        self._mlp_out = torch.empty((max_b, self.hidden_dim), dtype=dt, device=dev)
        self._hsum = torch.empty((max_b, self.hidden_dim), dtype=dt, device=dev)
        # split-k slabs of the widest projection and split-KV partials: sized once, never regrown
        L = nv.lib()
        need = 0
        shapes = [(qkv_dim, self.hidden_dim), (o_dim, _b.attn._o.k), (_b.mlp._gu.n, self.hidden_dim),
                  (self.hidden_dim, _b.mlp._d.k), (self.vocab_size, self.hidden_dim)]
        # (over every batch size: fewer rows may be cut into more k-slices)
        for n, k in shapes:
            need = max([need] + [L.mv_wgemm_workspace_bytes(m, n, k) for m in range(1, max_b + 1)])
        # projections that hand k-slabs to the kernel after them: O and down (their slab reduction
        # carries a norm, or is the tensor-parallel reduction)
        # and QKV, whose slabs the attention kernel adds up
        for n, k in ((self.hidden_dim, _b.attn._o.k), (self.hidden_dim, _b.mlp._d.k), (qkv_dim, self.hidden_dim)):
            need = max([need] + [L.mv_wgemm_postnorm_workspace_bytes(m, n, k) for m in range(1, max_b + 1)])
        self._gemm_ws = gemm_workspace(dev, max(need, 16))
        self._attn_ws = None
        # flags of the fused QKV + attention launch (zero-initialised, never reset) and the word it
        # raises if a wait inside it gives up: pinned host memory, read behind the token read-back
        self._qa_ws = torch.zeros((L.mv_qkv_attn_workspace_bytes() + 3) // 4, dtype=torch.int32, device=dev)
        self._qa_status = torch.zeros(1, dtype=torch.int32).pin_memory()
        # this runner's own meeting place for the parts of the greedy pick (zero-initialised tickets)
        self._argmax_ws = torch.zeros((L.mv_argmax_workspace_bytes() + 3) // 4, dtype=torch.float32, device=dev)

    # ------------------------------------------------------------------ pieces
    def _splits_uneven(self, bs, even):
        """Split count for rows of very different lengths.  With at least one (row, kv head) pair
        per CU the even-length rule leaves rows whole or nearly; uneven rows are cut in two (pairs
        up to the CU count) or four (more) so that the dispatcher can balance them."""
        pairs = bs * self.kv_num_heads
        cus = torch.cuda.get_device_properties(self.device).multi_processor_count
        if 2 * pairs <= cus:
            return even
        return max(even, 2 if pairs <= cus else 4)

    def _splits_for(self, bs, cache_manager):
        ns = self._num_splits.get(bs)
        if self._uneven and ns is not None:
            return self._splits_uneven(bs, ns)
        if ns is None:
            ctx = cache_manager._block_table_buffer.shape[1] * cache_manager.block_size
            ns = nv.lib().mv_paged_attn_decode_pick_splits(bs, self.kv_num_heads, self.head_size, ctx)
            self._num_splits[bs] = ns
            if self._attn_ws is None:
                # sized once for every batch size: captured graphs keep this pointer
                L = nv.lib()
                need = max(L.mv_paged_attn_decode_workspace_bytes(
                    b, self.num_heads, self.head_size,
                    max(L.mv_paged_attn_decode_pick_splits(b, self.kv_num_heads, self.head_size, ctx), 4))
                    for b in range(1, self.max_bs + 1))
                # zero-filled: the workspace starts with the arrival tickets of the split merge
                self._attn_ws = torch.zeros((need + 3) // 4, dtype=torch.float32, device=self.device)
        return ns

    def _qkv_proj(self, h, block, bs):
        rmsnorm_(h, block.ln_1.weight, self._normed[:bs], block.ln_1.eps)
        qkv_buf = self._qkv[:bs]
        # bs > 4: k cut as in the batched step, whose attention adds the QKV slabs up itself
        if self.fused:
            wgemm(self._normed[:bs], block.attn._qkv_w, qkv_buf, bias=block.attn._qkv_b,
                  workspace=self._gemm_ws, plan_slabs=bs > 4)
        else:
            wgemm(self._normed[:bs], block.attn._qkv_w, qkv_buf, workspace=self._gemm_ws, plan_slabs=bs > 4)
            if block.attn._qkv_b is not None:
                add_(qkv_buf, block.attn._qkv_b)
        return qkv_buf

    def _flash_attn(self, qkv, block, layer_idx, bs, cache_manager, use_graph_cache=True):
        hq, hkv, d = self.num_heads, self.kv_num_heads, self.head_size
        q = qkv[:, :hq * d].unflatten(1, (hq, d))
        k = qkv[:, hq * d:(hq + hkv) * d].unflatten(1, (hkv, d))
        v = qkv[:, (hq + hkv) * d:].unflatten(1, (hkv, d))
        k_cache, v_cache = cache_manager.get(layer_idx)
        attn = paged_attn_decode(
            q, k, v, k_cache, v_cache, cache_manager._cache_seqlens_buffer[:bs],
            cache_manager._block_table_buffer[:bs], self.attention._cos_pool,
            self.attention._sin_pool, num_splits=self._splits_for(bs, cache_manager),
            out=self._attn[:bs], workspace=self._attn_ws)
        out_buf = self._attn_out[:bs]
        # bs > 4: k cut as in the batched step, whose O and down projections hand k-slabs to the
        # kernel that follows them (same sums, bit for bit)
        wgemm(attn.reshape(bs, -1), block.attn._o, out_buf, workspace=self._gemm_ws, plan_slabs=bs > 4)
        return out_buf

    def _ffn(self, x_normed, block, bs, residual=None):
        if self.fused:
            wgemm(x_normed, block.mlp._gu, self._swiglu_out[:bs], swiglu=True, workspace=self._gemm_ws)
        else:
            wgemm(x_normed, block.mlp._gu, self._gate_up[:bs], workspace=self._gemm_ws)
            swiglu_gemm(self._gate_up[:bs], self._swiglu_out[:bs])
        wgemm(self._swiglu_out[:bs], block.mlp._d, self._mlp_out[:bs], residual=residual,
              workspace=self._gemm_ws, plan_slabs=bs > 4)
        return self._mlp_out[:bs]

    # ------------------------------------------------------------------ decode
    def _all_reduce(self, t):
        """In-place sum of [bs, hidden] over the ranks (model_graph.py:210,219)."""
        if self.world_size == 1:
            return t
        return self._reducer.all_reduce(t)

    def _layer(self, h, layer_idx, bs, cache_manager):
        """One transformer block on hidden state `h` [bs, hid]; returns the next hidden state
        (a view of a static buffer)."""
        if self.fused and bs <= 4:
            out, pend = self._layer_norm_fused(h, None, layer_idx, bs, cache_manager)
            return out if pend is None else add_(out, pend)
        block = self.model.transformer.h[layer_idx]
        qkv = self._qkv_proj(h, block, bs)
        attn_out = self._flash_attn(qkv, block, layer_idx, bs, cache_manager)
        attn_out = self._all_reduce(attn_out)
        rmsnorm_residual_gemm(attn_out, h, block.ln_2.weight, self._normed[:bs],
                              self._residual[:bs], block.ln_2.eps)
        if self.fused and self.world_size == 1:
            return self._ffn(self._normed[:bs], block, bs, residual=self._residual[:bs])
        mlp_out = self._ffn(self._normed[:bs], block, bs)
        return add_(self._all_reduce(mlp_out), self._residual[:bs])

    def _layer_norm_fused(self, h, pend, layer_idx, bs, cache_manager):
        """Block with both norms folded into the consuming projections (bs <= 4).

        `pend` is a residual that still has to be added to `h` (tensor-parallel runs: the sum
        `all_reduce(mlp_out) + residual` of the previous block is formed in this block's QKV
        prologue).  Returns (out, pend_out) with the same meaning for the next block."""
        block = self.model.transformer.h[layer_idx]
        ws = self._gemm_ws
        if pend is None and self.world_size == 1 and self.qkv_attn:
            splits = self._splits_for(bs, cache_manager)
            if self._qkv_attn_ok.get((bs, splits)) is not False:
                k_cache, v_cache = cache_manager.get(layer_idx)
                ok = qkv_attn_decode(
                    h, block.ln_1.weight, block.ln_1.eps, block.attn._qkv_w, block.attn._qkv_b, self._qkv[:bs],
                    self.num_heads, k_cache, v_cache, cache_manager._cache_seqlens_buffer[:bs],
                    cache_manager._block_table_buffer[:bs], self.attention._cos_pool, self.attention._sin_pool,
                    num_splits=splits, out=self._attn[:bs], workspace=self._attn_ws,
                    sync_workspace=self._qa_ws, status_word=self._qa_status)
                self._qkv_attn_ok[(bs, splits)] = ok
                if ok:
                    attn_out = wgemm(self._attn[:bs].reshape(bs, -1), block.attn._o, self._attn_out[:bs],
                                     workspace=ws)
                    wgemm_norm(attn_out, block.ln_2.weight, block.ln_2.eps, block.mlp._gu, self._swiglu_out[:bs],
                               add_r=h, sum_out=self._residual[:bs], swiglu=True, workspace=ws)
                    wgemm(self._swiglu_out[:bs], block.mlp._d, self._mlp_out[:bs],
                          residual=self._residual[:bs], workspace=ws)
                    return self._mlp_out[:bs], None
        if pend is None:
            wgemm_norm(h, block.ln_1.weight, block.ln_1.eps, block.attn._qkv_w, self._qkv[:bs],
                       bias=block.attn._qkv_b, workspace=ws)
        else:
            # h <- bf16(h + pend) (model_graph.py:219), then rmsnorm_ of the ROUNDED sum (:143)
            wgemm_norm(h, block.ln_1.weight, block.ln_1.eps, block.attn._qkv_w, self._qkv[:bs],
                       add_r=pend, sum_out=self._hsum[:bs], var_from_rounded=True,
                       bias=block.attn._qkv_b, workspace=ws)
            h = self._hsum[:bs]
        attn_out = self._flash_attn(self._qkv[:bs], block, layer_idx, bs, cache_manager)
        attn_out = self._all_reduce(attn_out)
        wgemm_norm(attn_out, block.ln_2.weight, block.ln_2.eps, block.mlp._gu, self._swiglu_out[:bs],
                   add_r=h, sum_out=self._residual[:bs], swiglu=True, workspace=ws)
        if self.world_size == 1:
            wgemm(self._swiglu_out[:bs], block.mlp._d, self._mlp_out[:bs],
                  residual=self._residual[:bs], workspace=ws)
            return self._mlp_out[:bs], None
        wgemm(self._swiglu_out[:bs], block.mlp._d, self._mlp_out[:bs], workspace=ws)
        return self._all_reduce(self._mlp_out[:bs]), self._residual[:bs]

    def _head(self, h, bs, out=None, pend=None):
        """Final norm + LM head; `out` may be an fp32 [bs,V] tensor to get un-rounded logits."""
        tr = self.model.transformer
        dst = self._logits[:bs] if out is None else out
        if self.fused and bs <= 4:
            if pend is None:
                return wgemm_norm(h, tr.ln_f.weight, tr.ln_f.eps, self._lm_head, dst,
                                  workspace=self._gemm_ws)
            return wgemm_norm(h, tr.ln_f.weight, tr.ln_f.eps, self._lm_head, dst, add_r=pend,
                              sum_out=self._hsum[:bs], var_from_rounded=True, workspace=self._gemm_ws)
        if pend is not None:
            h = add_(h, pend)
        rmsnorm_(h, tr.ln_f.weight, self._normed[:bs], tr.ln_f.eps)
        return wgemm(self._normed[:bs], self._lm_head, dst, workspace=self._gemm_ws)

    def decode(self, input_ids, bs, cache_manager, use_graph_cache=True):
        """One token for each of `bs` rows; returns logits [bs, V] (a view of a static buffer).
        Allocation-free after the first call for a given bs, so it can be stream-captured."""
        h = embedding(self.model.transformer.wte.weight, input_ids, self._hidden0[:bs])
        if self.fused and self.world_size > 1 and self._reducer.fits(bs, self.hidden_dim):
            logits = self._decode_tp(h, bs, cache_manager)
        elif self.fused and bs <= 4:
            pend = None
            for layer_idx in range(self.num_layers):
                h, pend = self._layer_norm_fused(h, pend, layer_idx, bs, cache_manager)
            logits = self._head(h, bs, pend=pend)
        elif self.fused and self.world_size == 1 and self.hidden_dim <= 8192:  # mv_wgemm_postnorm's row limit
            logits = self._decode_batched(h, bs, cache_manager)
        else:
            for layer_idx in range(self.num_layers):
                h = self._layer(h, layer_idx, bs, cache_manager)
            logits = self._head(h, bs)
        greedy(logits, out=self._greedy, workspace=self._argmax_ws)
        if self.vocab_size % 8 == 0:
            sample(logits, self._samp_temp[:bs], self._samp_topp[:bs], min(self.decode_top_k, self.vocab_size),
                   uniforms=self._samp_u[:bs], out=self._picked, greedy_ids=self._greedy)
        else:                         # the device sampler wants rows of a multiple of 8
            self._picked[:bs].copy_(self._greedy[:bs])
        return logits

    def picked_ids(self, logits: torch.Tensor):
        """The token of every row of the last decode - drawn with the row's temperature / top-p /
        uniform number as they stood in the static buffers when the step ran, or the argmax for
        temperature 0 - if `logits` are what that decode returned, else None."""
        if logits.data_ptr() == self._logits.data_ptr() and logits.shape[1] == self.vocab_size:
            return self._picked[:logits.shape[0]]
        return None

    def set_sampling(self, temperatures, top_ps):
        """Per-row temperature and top-p for the steps to come (host lists; rows not named keep
        temperature 0 = greedy).  Call when the batch composition changes."""
        n = len(temperatures)
        self._samp_temp.zero_()
        self._samp_topp.fill_(1.0)
        if n:
            self._samp_temp[:n] = torch.tensor(temperatures, dtype=torch.float32)
            self._samp_topp[:n] = torch.tensor(top_ps, dtype=torch.float32)

    def draw_uniforms(self, bs: int, generator=None):
        """Fresh uniform numbers for the next step's draws (queued ahead of the step)."""
        self._samp_u[:bs].uniform_(0.0, 1.0, generator=generator)

    def _qkv_attention(self, normed, block, layer_idx, bs, cache_manager):
        """QKV projection of `normed` and attention over the pages -> [bs, Hq, D].  More than four
        rows: the projection is cut along k and leaves its slabs in the workspace, which the
        attention kernel adds up (with the bias) as it loads q, k and v - the same values, bit for
        bit, as the projection's own slab reduction would have stored, without that launch."""
        hq, hkv, d = self.num_heads, self.kv_num_heads, self.head_size
        k_cache, v_cache = cache_manager.get(layer_idx)
        seqlens = cache_manager._cache_seqlens_buffer[:bs]
        table = cache_manager._block_table_buffer[:bs]
        splits = self._splits_for(bs, cache_manager)
        if bs > 4 and self.qkv_slabs:
            ks = wgemm_slabs(normed, block.attn._qkv_w, self._gemm_ws)
            return paged_attn_decode_slabs(
                self._gemm_ws, ks, block.attn._qkv_b, hq, k_cache, v_cache, seqlens, table,
                self.attention._cos_pool, self.attention._sin_pool, num_splits=splits,
                out=self._attn[:bs], workspace=self._attn_ws)
        # (k cut as the slab-leaving call cuts it: the same sums either way)
        wgemm(normed, block.attn._qkv_w, self._qkv[:bs], bias=block.attn._qkv_b, workspace=self._gemm_ws,
              plan_slabs=bs > 4)
        qkv = self._qkv[:bs]
        return paged_attn_decode(
            qkv[:, :hq * d].unflatten(1, (hq, d)), qkv[:, hq * d:(hq + hkv) * d].unflatten(1, (hkv, d)),
            qkv[:, (hq + hkv) * d:].unflatten(1, (hkv, d)), k_cache, v_cache, seqlens, table,
            self.attention._cos_pool, self.attention._sin_pool, num_splits=splits,
            out=self._attn[:bs], workspace=self._attn_ws)

    def _decode_batched(self, h, bs, cache_manager):
        """bs > 4 on one GPU: the norm that follows the O and the down projection rides on their
        slab reduction (`wgemm_postnorm`), so a block is QKV, attention, O(+add+norm), gate/up,
        down(+residual+next norm): five projections' worth of launches instead of ten.  Same
        arithmetic, bit for bit, as `_layer` (model_graph.py:141-220 of the reference)."""
        tr = self.model.transformer
        blocks = tr.h
        ws = self._gemm_ws
        normed = self._normed[:bs]
        rmsnorm_(h, blocks[0].ln_1.weight, normed, blocks[0].ln_1.eps)
        for layer_idx in range(self.num_layers):
            block = blocks[layer_idx]
            attn = self._qkv_attention(normed, block, layer_idx, bs, cache_manager)
            # residual <- bf16(attn @ Wo^T + h); normed <- rmsnorm(that sum, unrounded) * ln_2
            wgemm_postnorm(attn.reshape(bs, -1), block.attn._o, self._residual[:bs], block.ln_2.weight,
                           block.ln_2.eps, normed, residual=h, workspace=ws)
            wgemm(normed, block.mlp._gu, self._swiglu_out[:bs], swiglu=True, workspace=ws)
            # h <- bf16(act @ Wd^T + residual); normed <- rmsnorm(h) * (next block's ln_1 | ln_f)
            nxt = blocks[layer_idx + 1].ln_1 if layer_idx + 1 < self.num_layers else tr.ln_f
            wgemm_postnorm(self._swiglu_out[:bs], block.mlp._d, self._mlp_out[:bs], nxt.weight, nxt.eps,
                           normed, residual=self._residual[:bs], var_from_rounded=True, workspace=ws)
            h = self._mlp_out[:bs]
        return wgemm(normed, self._lm_head, self._logits[:bs], workspace=ws)

    def _decode_tp(self, h, bs, cache_manager):
        """Several ranks, any batch size: each of the two reductions of a block (after the O and
        after the down projection, core/layer/model_graph.py:210,219) is ONE launch that also adds
        the residual and applies the RMSNorm that follows (:213, and :143 of the next block), so a
        block is QKV, attention, O, reduce+add+norm, gate/up, down, reduce+add+norm.  With the
        one-shot reducer the row-parallel projections leave their k-slabs in the workspace and
        the reduction adds them on the way in."""
        tr = self.model.transformer
        blocks = tr.h
        ws = self._gemm_ws
        red = self._reducer
        n = self.hidden_dim
        normed = self._normed[:bs]
        hq, hkv, d = self.num_heads, self.kv_num_heads, self.head_size

        def row_parallel(x, w, residual, ln, y, var_from_rounded):
            if red.takes_slabs:
                ks = wgemm_slabs(x, w, ws)
                red.addnorm(bs, n, slabs=ws, ksplit=ks, residual=residual, y=y, ln_weight=ln.weight,
                            eps=ln.eps, normed=normed, var_from_rounded=var_from_rounded)
            else:
                wgemm(x, w, self._attn_out[:bs], workspace=ws)
                red.addnorm(bs, n, partial=self._attn_out[:bs], residual=residual, y=y,
                            ln_weight=ln.weight, eps=ln.eps, normed=normed,
                            var_from_rounded=var_from_rounded)

        rmsnorm_(h, blocks[0].ln_1.weight, normed, blocks[0].ln_1.eps)
        for layer_idx in range(self.num_layers):
            block = blocks[layer_idx]
            attn = self._qkv_attention(normed, block, layer_idx, bs, cache_manager)
            # residual <- bf16(sum_ranks(attn @ Wo^T) + h); normed <- rmsnorm(unrounded) * ln_2
            row_parallel(attn.reshape(bs, -1), block.attn._o, h, block.ln_2, self._residual[:bs], False)
            wgemm(normed, block.mlp._gu, self._swiglu_out[:bs], swiglu=True, workspace=ws)
            # h <- bf16(sum_ranks(act @ Wd^T) + residual); normed <- rmsnorm(h) * (next ln_1 | ln_f)
            nxt = blocks[layer_idx + 1].ln_1 if layer_idx + 1 < self.num_layers else tr.ln_f
            row_parallel(self._swiglu_out[:bs], block.mlp._d, self._residual[:bs], nxt, self._mlp_out[:bs], True)
            h = self._mlp_out[:bs]
        if self._gatherer is not None:
            # the head cut by vocabulary rows: 1 / world of its bytes per rank, then one exchange
            wgemm(normed, self._lm_head_shard, self._logits_local[:bs], workspace=ws)
            return self._gatherer.gather(self._logits_local[:bs], self._logits[:bs])
        return wgemm(normed, self._lm_head, self._logits[:bs], workspace=ws)

    def raise_if_exchange_failed(self):
        """Raises kernel.allreduce.ExchangeTimeout if a reduction or logits gather of a finished
        step gave up waiting for a peer (its results are wrong).  A read of two host words: call it
        after waiting for the step, e.g. behind the read-back of its tokens."""
        st = int(self._qa_status[0])
        if st:
            raise RuntimeError(f"a wait inside the fused QKV + attention launch gave up (workgroup {st - 1}): "
                               "the launch needs the whole GPU; this step's results are invalid")
        if self.world_size > 1:
            for comm, what in ((self._reducer, "all-reduce of a decode step"),
                               (self._gatherer, "logits all-gather of a decode step")):
                check = getattr(comm, "raise_if_timed_out", None)
                if check is not None:
                    check(what)

    def greedy_ids(self, logits: torch.Tensor):
        """Row-wise argmax of `logits` if they are what the last decode returned (then it is already
        there), else None."""
        if logits.data_ptr() == self._logits.data_ptr() and logits.shape[1] == self.vocab_size:
            return self._greedy[:logits.shape[0]]
        return None

    def capture(self, cache_manager, batch_sizes: List[int] = [1, 2, 4, 8, 16, 32], warmup: int = 2):
        """Capture one graph per batch size.  The static block-table / seqlens buffers are read
        by the captured kernels at replay time, so their contents may change between replays."""
        if self._is_graph_ready:
            return
        # torch.distributed collectives are captured through torch's RCCL hooks; the one-shot
        # reducer is an ordinary kernel of the library
        use_torch_graph = self.world_size > 1 and not getattr(self._reducer, "takes_slabs", False)
        if use_torch_graph and pc.host_side_backend():
            # a host all-reduce (gloo rehearsal) cannot be part of a graph: every rank decodes eagerly
            logger.warning("host-side all-reduce backend: decode steps are not captured")
            self._is_graph_ready = True
            return
        for bs in batch_sizes:
            if bs > self.max_bs:
                continue
            even = self._splits_for(bs, cache_manager)
            for uneven in (False, True):
                if uneven and self._splits_uneven(bs, even) == even:
                    continue  # same launch either way
                self._uneven = uneven
                try:
                    g = self._capture_one(bs, cache_manager, warmup, use_torch_graph)
                finally:
                    self._uneven = False
                if g is not None:
                    (self._graphs_uneven if uneven else self._graphs)[bs] = g
            logger.info("captured decode graph for batch size %d", bs)
        self._is_graph_ready = True

    def _capture_one(self, bs, cache_manager, warmup, use_torch_graph):
        """One decode step for `bs` rows as a replayable graph."""
        g = None
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            # warm-up on the buffers as they are: every kernel guards its page lookups, so
            # stale or -1 entries are skipped, never dereferenced
            for _ in range(warmup):
                self.decode(self._input_ids[:bs], bs, cache_manager)
            side.synchronize()
            if use_torch_graph:
                # a failure here is raised, never swallowed: ranks that captured and ranks that did
                # not would otherwise issue different sequences of collectives
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    self.decode(self._input_ids[:bs], bs, cache_manager)
            else:
                import ctypes
                nv.check("mv_graph_begin_capture", nv.lib().mv_graph_begin_capture(side.cuda_stream))
                try:
                    self.decode(self._input_ids[:bs], bs, cache_manager)
                finally:
                    handle = ctypes.c_void_p()
                    rc = nv.lib().mv_graph_end_capture(side.cuda_stream, ctypes.byref(handle))
                nv.check("mv_graph_end_capture", rc)
                g = HipGraph(handle)
        torch.cuda.current_stream().wait_stream(side)
        return g

    def _rows_uneven(self, cache_manager, batch_size) -> bool:
        """Whether the lengths the cache manager last wrote for this batch differ enough for the
        finer attention split (host-side knowledge only: nothing is read back from the device)."""
        n = getattr(cache_manager, "step_rows", 0)
        if n != batch_size or n < 2:
            return False
        ratio = self.uneven_ratio
        if ratio is None:
            cus = torch.cuda.get_device_properties(self.device).multi_processor_count
            ratio = 1.5 if batch_size * self.kv_num_heads <= cus else 1.3
        return cache_manager.step_len_max * n > ratio * cache_manager.step_len_sum

    def forward(self, input_ids: torch.Tensor, cache_manager, batch_size: int) -> torch.Tensor:
        self._input_ids[:batch_size] = input_ids
        g = None
        if self._graphs_uneven and self._rows_uneven(cache_manager, batch_size):
            g = self._graphs_uneven.get(batch_size)
        if g is None:
            g = self._graphs.get(batch_size)
        if g is None:
            return self.decode(self._input_ids[:batch_size], batch_size, cache_manager)
        g.replay()
        return self._logits[:batch_size]
