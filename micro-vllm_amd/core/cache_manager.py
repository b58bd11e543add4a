"""Paged KV-cache bookkeeping with the reference's KVCacheManager surface
(core/cache_manager.py:262-691): alloc / append / put / get / get_blocks / cache_batch_data /
get_buffer_data / get_slots / free / reset / stats, plus the attributes the runners read
directly (`k_caches`, `v_caches`, `block_size`, `_block_table_buffer`, `_cache_seqlens_buffer`).

Storage is one torch tensor per K and V (sliced per layer) laid out
[layer][page][slot][kv_head][D]; the bookkeeping is host-side integers.  The block table and
sequence-length buffers keep fixed device addresses (they are rewritten in place each step) so
captured graphs stay valid.
"""
import collections
from typing import List

import torch

from .. import _native as nv


def store_kvcache(key: torch.Tensor, value: torch.Tensor, k_cache: torch.Tensor,
                  v_cache: torch.Tensor, slot_mapping: torch.Tensor, block_size: int):
    """key/value [T,H,D] -> cache[slot // block_size, slot % block_size]; slot -1 is skipped.
    Same contract as the reference's store_kvcache (core/cache_manager.py:143)."""
    num_tokens, num_heads, head_size = key.shape
    assert key.dim() == 3 and value.dim() == 3
    assert value.shape == (num_tokens, num_heads, head_size)
    assert slot_mapping.numel() == num_tokens
    assert key.is_cuda, "store_kvcache runs on the GPU library only"
    assert key.stride(2) == 1 and key.stride(1) == head_size
    assert value.stride() == key.stride(), "key and value must share strides"
    assert k_cache.is_contiguous() and v_cache.is_contiguous()
    slots = slot_mapping.to(device=key.device, dtype=torch.int32).contiguous()
    stride_t = key.stride(0) if num_tokens > 1 else num_heads * head_size
    nv.check("mv_store_kv", nv.lib().mv_store_kv(
        key.data_ptr(), value.data_ptr(), stride_t, k_cache.data_ptr(), v_cache.data_ptr(),
        slots.data_ptr(), num_tokens, num_heads, head_size, block_size, k_cache.shape[0],
        nv.stream_ptr()))


def store_kvcache_batch(key, value, k_cache, v_cache, slot_mapping_batch, block_size):
    """[B,1,H,D] form of store_kvcache (core/cache_manager.py:196)."""
    batch_size, num_tokens, num_heads, head_size = key.shape
    assert num_tokens == 1, "Only support num_tokens=1 for decoding stage"
    assert slot_mapping_batch.shape == (batch_size, num_tokens)
    store_kvcache(key.reshape(-1, num_heads, head_size), value.reshape(-1, num_heads, head_size),
                  k_cache, v_cache, slot_mapping_batch.reshape(-1), block_size)


class KVCacheManager:
    def __init__(self, n_blocks: int, block_size: int, n_layers: int, n_heads: int, head_size: int,
                 dtype=torch.float16, device="cuda", max_tokens: int = 1024,
                 max_batch_size: int = 32, strict_reference: bool = False):
        """`strict_reference=True` keeps the reference's accounting for prompts whose length is
        an exact multiple of block_size (the last block is recorded as empty, so the next
        page is never opened - core/cache_manager.py:384-387); the default records it as
        full, which is what lets a sequence of 4096 tokens keep decoding."""
        self.n_blocks, self.block_size, self.n_layers = n_blocks, block_size, n_layers
        self.dtype, self.device = dtype, device
        self.strict_reference = strict_reference

        shape = (n_layers, n_blocks, block_size, n_heads, head_size)
        self._k_store = torch.zeros(shape, dtype=dtype, device=device)
        self._v_store = torch.zeros(shape, dtype=dtype, device=device)
        self.k_caches = [self._k_store[i] for i in range(n_layers)]
        self.v_caches = [self._v_store[i] for i in range(n_layers)]

        self._free = collections.deque(range(n_blocks))
        self._blocks = {}  # seq_id -> [block ids]
        self._pos = {}     # block id -> used slots
        self._opened_by_append = {}   # seq_id -> whether its latest append took a page from the free list

        self.cache_seqlens = torch.tensor([1], dtype=torch.int32, device=self.device)
        self.max_tokens = max_tokens
        self.max_seq_blocks = (max_tokens + block_size - 1) // block_size
        # One int32 allocation [max_batch seqlens | max_batch x max_seq_blocks block table], so that
        # the per-step rewrite of both is a single host-to-device copy.
        W = self.max_seq_blocks
        self._step_buf = torch.full((max_batch_size * (1 + W),), -1, dtype=torch.int32, device=device)
        self._step_buf[:max_batch_size] = 0
        self._cache_seqlens_buffer = self._step_buf[:max_batch_size]
        self._block_table_buffer = self._step_buf[max_batch_size:].view(max_batch_size, W)
        # Pinned host staging for the per-step rewrite: one asynchronous copy instead of one per row.  The copies are stream-ordered behind earlier decode steps, so the host
        # may be several steps ahead of the device; a small ring of staging buffers, each guarded
        # by an event recorded after its copy, keeps the host from overwriting bytes a queued
        # copy has not read yet.
        self._on_gpu = torch.cuda.is_available() and str(device).startswith("cuda")
        self._ring = 4 if self._on_gpu else 1
        self._step_host = [torch.full((max_batch_size * (1 + W),), -1, dtype=torch.int32,
                                      pin_memory=self._on_gpu) for _ in range(self._ring)]
        for h in self._step_host:
            h[:max_batch_size] = 0
        self._sl_host = [h[:max_batch_size] for h in self._step_host]
        self._bt_host = [h[max_batch_size:].view(max_batch_size, W) for h in self._step_host]
        self._ring_events = [None] * self._ring
        self.step_len_max = self.step_len_sum = self.step_rows = 0
        self._ring_next = 0

    # ------------------------------------------------------------------ allocation
    def alloc(self, seq_id: int, n_tokens: int):
        """Reserve pages for an n_tokens prompt; returns (ok, slot_mapping int32 [n_tokens])."""
        bs = self.block_size
        n_needed = (n_tokens + bs - 1) // bs
        if len(self._free) < n_needed:
            return False, None
        blocks = [self._free.popleft() for _ in range(n_needed)]
        for i, b in enumerate(blocks):
            if i < n_needed - 1:
                self._pos[b] = bs
            else:
                tail = n_tokens % bs
                self._pos[b] = tail if (tail or self.strict_reference) else bs
        self._blocks[seq_id] = blocks
        idx = torch.arange(n_tokens, dtype=torch.int64)
        blk = torch.tensor(blocks, dtype=torch.int64)
        slot_mapping = (blk[idx // bs] * bs + idx % bs).to(torch.int32) if n_tokens else \
            torch.zeros(0, dtype=torch.int32)
        return True, slot_mapping.to(self.device)

    def append(self, seq_id: int):
        """One more token for seq_id; returns its slot or -1.  Like the reference
        (core/cache_manager.py:401-438) a page is considered full at block_size - 1 used
        slots, which is what keeps it in step with the engine's position numbering."""
        if seq_id not in self._blocks:
            return -1
        blocks = self._blocks[seq_id]
        last = blocks[-1]
        used = self._pos[last]
        if used < self.block_size - 1:
            self._pos[last] = used + 1
            self._opened_by_append[seq_id] = False
            return last * self.block_size + used
        if self._free:
            nb = self._free.popleft()
            blocks.append(nb)
            self._pos[nb] = 1
            self._opened_by_append[seq_id] = True
            return nb * self.block_size
        print(f"cache_manager.append failed for seq {seq_id}: no free block "
              f"(blocks={blocks}, used={used}, block_size={self.block_size})")
        return -1

    def unappend(self, seq_id: int, slot: int):
        """Take back the `append` that returned `slot` (the step it was made for will not run):
        a page opened by it goes back to the head of the free list, otherwise the page's fill
        count drops by one.  Whether the append opened a page is what it recorded, not what the
        slot number suggests: offset 0 of an EXISTING page is also what the first append after a
        prompt of a whole number of pages returns under `strict_reference` (that page holds
        prompt rows and must stay)."""
        blocks = self._blocks.get(seq_id)
        if not blocks or slot < 0:
            return
        last = blocks[-1]
        assert slot // self.block_size == last, "not this sequence's latest slot"
        if self._opened_by_append.pop(seq_id, False):
            assert slot % self.block_size == 0 and self._pos.get(last) == 1 and len(blocks) > 1
            blocks.pop()
            self._pos.pop(last, None)
            self._free.appendleft(last)
        else:
            self._pos[last] -= 1

    def clear_slots(self, seq_ids: list, positions: list):
        """Zero the K and V rows (every layer, every kv head) at token index positions[i] of sequence
        seq_ids[i].  Pages are recycled without being cleared, so a slot that the position numbering
        makes attention read before anything writes it (the one right after a prompt, see
        InferenceEngine._decode) would otherwise hold another request's keys.  Positions whose page
        is not allocated are skipped.  Two small launches for the whole batch."""
        pages, offs = [], []
        for sid, pos in zip(seq_ids, positions):
            blocks = self._blocks.get(sid)
            if blocks is None or pos < 0 or pos // self.block_size >= len(blocks):
                continue
            pages.append(blocks[pos // self.block_size])
            offs.append(pos % self.block_size)
        if not pages:
            return 0
        pg = torch.tensor(pages, dtype=torch.long).to(self.device, non_blocking=True)
        of = torch.tensor(offs, dtype=torch.long).to(self.device, non_blocking=True)
        self._k_store[:, pg, of] = 0
        self._v_store[:, pg, of] = 0
        return len(pages)

    def put(self, seq_id: int, key: torch.Tensor, value: torch.Tensor, layer: int,
            slot_map: torch.Tensor):
        store_kvcache(key, value, self.k_caches[layer], self.v_caches[layer], slot_map,
                      self.block_size)

    def get(self, layer: int, block_id: int = None):
        k, v = self.k_caches[layer], self.v_caches[layer]
        return (k[block_id], v[block_id]) if block_id is not None else (k, v)

    def get_blocks(self, seq_id: int):
        return self._blocks.get(seq_id, [])

    # ------------------------------------------------------------------ per-step buffers
    def cache_batch_data(self, seq_ids: list, context_lens: list):
        """Rewrite rows [0, len(seq_ids)) of the static block table / seqlens buffers in place."""
        n = len(seq_ids)
        if n > 0:
            slot = self._ring_next
            self._ring_next = (slot + 1) % self._ring
            if self._ring_events[slot] is not None:
                self._ring_events[slot].synchronize()
            bt, sl = self._bt_host[slot], self._sl_host[slot]
            bt_np = bt.numpy()   # a view: plain integer stores instead of one tensor per row
            bt_np[:n] = -1
            for i, sid in enumerate(seq_ids):
                blocks = self._blocks.get(sid, [])
                if blocks:
                    if len(blocks) > self.max_seq_blocks:
                        raise RuntimeError(
                            f"sequence {sid} spans {len(blocks)} blocks but the block table is "
                            f"{self.max_seq_blocks} wide (max_tokens={self.max_tokens})")
                    bt_np[i, :len(blocks)] = blocks
            sl.numpy()[:n] = context_lens
            # what the host knows about this step's lengths: the decode runner picks its
            # attention split count from it (uneven rows want finer work units)
            self.step_len_max, self.step_len_sum, self.step_rows = max(context_lens), sum(context_lens), n
            upto = self._cache_seqlens_buffer.numel() + n * self.max_seq_blocks
            self._step_buf[:upto].copy_(self._step_host[slot][:upto], non_blocking=True)
            if self._on_gpu:
                ev = self._ring_events[slot] or torch.cuda.Event()
                ev.record()
                self._ring_events[slot] = ev
        return self._block_table_buffer[:n], self._cache_seqlens_buffer[:n]

    def get_buffer_data(self, seq_ids: list):
        n = len(seq_ids)
        return self._block_table_buffer[:n], self._cache_seqlens_buffer[:n]

    def get_slots(self, seq_id: int, token_positions: list) -> list:
        if seq_id not in self._blocks:
            return [-1] * len(token_positions)
        blocks, bs = self._blocks[seq_id], self.block_size
        out = []
        for pos in token_positions:
            bi, off = divmod(pos, bs)
            if bi >= len(blocks) or off >= self._pos.get(blocks[bi], 0):
                out.append(-1)
            else:
                out.append(blocks[bi] * bs + off)
        return out

    def free(self, seq_id: int):
        blocks = self._blocks.pop(seq_id, None)
        self._opened_by_append.pop(seq_id, None)
        if blocks:
            for b in blocks:
                self._free.append(b)
                self._pos.pop(b, None)

    def reset(self):
        self._free = collections.deque(range(self.n_blocks))
        self._blocks.clear()
        self._pos.clear()
        self._opened_by_append.clear()

    @property
    def stats(self):
        total, free = self.n_blocks, len(self._free)
        longest = run = 0
        prev = None
        for b in sorted(self._free):
            run = run + 1 if (prev is not None and b == prev + 1) else 1
            longest = max(longest, run)
            prev = b
        frag = 1.0 - longest / max(free, 1) if free > 0 else 0.0
        return {"total": total, "free": free, "used": total - free, "frag": round(frag, 3)}
