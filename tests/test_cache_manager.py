"""The product KVCacheManager's integer bookkeeping against traces recorded from the reference's own
KVCacheManager (tests/golden/cache_traces.json).  Runs on CPU tensors: no kernel is involved."""
import json
import os

import pytest
import torch

from test_oracle_golden import _replay


def make(mv, tr, strict):
    from micro_vllm_amd.core.cache_manager import KVCacheManager

    return KVCacheManager(tr["n_blocks"], tr["block_size"], 1, 1, 8, dtype=torch.float32,
                          device="cpu", max_tokens=tr["max_tokens"],
                          max_batch_size=tr["max_batch_size"], strict_reference=strict)


def test_block_tables_bit_exact_vs_reference(mv, golden_dir):
    traces = json.load(open(os.path.join(golden_dir, "cache_traces.json")))
    for tr in traces:
        _replay(tr, make(mv, tr, strict=True), is_ref=False)


def test_default_mode_matches_reference_off_the_page_boundary(mv, golden_dir):
    """With the page-boundary fix on (default), every trace whose prompt length is not a multiple
    of the block size is still identical to the reference."""
    traces = json.load(open(os.path.join(golden_dir, "cache_traces.json")))
    for tr in traces:
        if tr["name"].startswith("degenerate"):
            continue
        _replay(tr, make(mv, tr, strict=False), is_ref=False)


@pytest.mark.parametrize("L", [256, 4096])
def test_full_last_page_opens_the_next_one(mv, L):
    """Prompt of exactly k pages: the reference records the last page as empty and never opens the
    next, leaving block_table[L // 256] = -1 where the first decoded token is written.  The default
    mode opens it on the first append."""
    from micro_vllm_amd.core.cache_manager import KVCacheManager

    mgr = KVCacheManager(40, 256, 1, 1, 8, dtype=torch.float32, device="cpu", max_tokens=8192,
                         max_batch_size=2)
    ok, slots = mgr.alloc(1, L)
    assert ok and slots.numel() == L
    mgr.append(1)
    bt, sl = mgr.cache_batch_data([1], [L + 1])
    assert int(bt[0, (L + 1) // 256]) >= 0
    assert sl.tolist() == [L + 1]


def test_oom_and_table_width(mv):
    from micro_vllm_amd.core.cache_manager import KVCacheManager

    mgr = KVCacheManager(4, 16, 1, 1, 8, dtype=torch.float32, device="cpu", max_tokens=32,
                         max_batch_size=2)
    assert mgr._block_table_buffer.shape == (2, 2)
    ok, _ = mgr.alloc(1, 100)
    assert not ok
    ok, _ = mgr.alloc(1, 48)  # 3 pages, table is 2 wide
    assert ok
    with pytest.raises(RuntimeError):
        mgr.cache_batch_data([1], [48])
    assert mgr.append(99) == -1
    mgr.free(1)
    assert mgr.stats == {"total": 4, "free": 4, "used": 0, "frag": 0.0}
    a = mgr._block_table_buffer.data_ptr()
    mgr.alloc(2, 5)
    mgr.cache_batch_data([2], [5])
    assert mgr._block_table_buffer.data_ptr() == a  # static address


def test_unappend_restores_the_state_before_the_append(mv):
    """A decode step that is not run after all gives its slots back: inside a page the fill count
    drops, across a page boundary the page returns to the head of the free list, and the next
    append hands out the same slot again."""
    import copy

    from micro_vllm_amd.core.cache_manager import KVCacheManager

    mgr = KVCacheManager(6, 16, 1, 1, 8, dtype=torch.float32, device="cpu", max_tokens=64, max_batch_size=2)
    assert mgr.alloc(1, 13)[0] and mgr.alloc(2, 20)[0]
    for _ in range(40):                      # crosses page boundaries of both sequences
        before = (copy.deepcopy(mgr._blocks), dict(mgr._pos), list(mgr._free))
        s1 = mgr.append(1)
        if s1 == -1:
            break
        s2 = mgr.append(2)
        if s2 == -1:                         # sequence 2 starved: sequence 1 gives its slot back
            mgr.unappend(1, s1)
            assert (mgr._blocks, dict(mgr._pos), list(mgr._free)) == before
            break
        mgr.unappend(2, s2)
        mgr.unappend(1, s1)
        assert (mgr._blocks, dict(mgr._pos), list(mgr._free)) == before
        assert (mgr.append(1), mgr.append(2)) == (s1, s2)
    else:
        raise AssertionError("the pool never ran out")


def test_unappend_after_a_whole_page_prompt_under_strict_reference_keeps_the_prompt_page(mv):
    """strict_reference records the last page of a prompt of exactly k pages as empty (the reference's
    alloc), so the first append returns offset 0 of that EXISTING page.  Taking that append back must
    only lower the fill count: the page holds prompt rows and stays with the sequence."""
    import copy

    from micro_vllm_amd.core.cache_manager import KVCacheManager

    mgr = KVCacheManager(6, 16, 1, 1, 8, dtype=torch.float32, device="cpu", max_tokens=64, max_batch_size=2,
                         strict_reference=True)
    assert mgr.alloc(1, 32)[0]              # two whole pages
    pages = list(mgr._blocks[1])
    before = (copy.deepcopy(mgr._blocks), dict(mgr._pos), list(mgr._free))
    slot = mgr.append(1)
    assert slot == pages[-1] * 16           # offset 0 of the prompt's last page, no page opened
    mgr.unappend(1, slot)
    assert (mgr._blocks, dict(mgr._pos), list(mgr._free)) == before
    assert mgr._blocks[1] == pages and pages[-1] not in mgr._free
    assert mgr.append(1) == slot
