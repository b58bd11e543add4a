"""CPU restatement of the reference's KV-cache bookkeeping (test infrastructure).

Integer arithmetic only; follows core/cache_manager.py:351-438 (alloc / append),
:509-546 (cache_batch_data) and :616-636 (free).
"""
import collections

import torch


class RefKVCache:
    def __init__(self, n_blocks, block_size, n_layers, n_heads, head_size, max_tokens=1024,
                 max_batch_size=32, strict_reference=True, dtype=torch.bfloat16):
        self.n_blocks, self.block_size, self.n_layers = n_blocks, block_size, n_layers
        self.strict_reference = strict_reference
        # core/cache_manager.py:311-323: zero-initialised [n_blocks, block_size, heads, D] per layer
        self.k_caches = [torch.zeros(n_blocks, block_size, n_heads, head_size, dtype=dtype)
                         for _ in range(n_layers)]
        self.v_caches = [torch.zeros(n_blocks, block_size, n_heads, head_size, dtype=dtype)
                         for _ in range(n_layers)]
        self.free_blocks = collections.deque(range(n_blocks))  # :327
        self.blocks = {}                                       # :330
        self.fill = {}                                         # :333 (_pos)
        self.max_seq_blocks = (max_tokens + block_size - 1) // block_size  # :341
        # :343-349 static buffers, -1 / 0 filled
        self.block_table = torch.full((max_batch_size, self.max_seq_blocks), -1, dtype=torch.int32)
        self.seqlens = torch.zeros(max_batch_size, dtype=torch.int32)

    def alloc(self, seq_id, n_tokens):
        """:351-399.  With strict_reference the last block of an exact multiple of block_size is
        recorded with fill 0 (n_tokens % block_size, :385); otherwise as full."""
        bs = self.block_size
        need = (n_tokens + bs - 1) // bs
        if len(self.free_blocks) < need:
            return False, None
        blocks = [self.free_blocks.popleft() for _ in range(need)]
        for i, b in enumerate(blocks):
            if i == len(blocks) - 1:
                tail = n_tokens % bs
                self.fill[b] = tail if (tail or self.strict_reference) else bs
            else:
                self.fill[b] = bs
        self.blocks[seq_id] = blocks
        slots = [blocks[i // bs] * bs + i % bs for i in range(n_tokens)]
        return True, slots

    def append(self, seq_id):
        """:401-438: room is judged against block_size - 1 (:425)."""
        if seq_id not in self.blocks:
            return -1
        blocks = self.blocks[seq_id]
        last = blocks[-1]
        cur = self.fill[last]
        if cur < self.block_size - 1:
            self.fill[last] += 1
            return last * self.block_size + cur
        if self.free_blocks:
            nb = self.free_blocks.popleft()
            blocks.append(nb)
            self.fill[nb] = 1
            return nb * self.block_size
        return -1

    def cache_batch_data(self, seq_ids, context_lens):
        """:509-546: rows [0,n) reset to -1, then each row's block list; seqlens = context_lens."""
        n = len(seq_ids)
        self.block_table[:n] = -1
        for i, sid in enumerate(seq_ids):
            blocks = self.blocks.get(sid, [])
            if blocks:
                self.block_table[i, :len(blocks)] = torch.tensor(blocks, dtype=torch.int32)
        if n:
            self.seqlens[:n] = torch.tensor(context_lens, dtype=torch.int32)
        return self.block_table[:n], self.seqlens[:n]

    def free(self, seq_id):
        """:616-636."""
        if seq_id in self.blocks:
            for b in self.blocks[seq_id]:
                self.free_blocks.append(b)
                self.fill.pop(b, None)
            del self.blocks[seq_id]
