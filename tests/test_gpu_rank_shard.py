"""One rank of a tensor-parallel group at the full widths of the multi-GPU configurations, run by a
single process on one GPU: Llama-3-8B widths as a rank of 2, 4 and 8 at batch 16 / context 2048, and
Llama-3-70B widths (hidden 8192, 8 query heads on 1 kv head per rank) as ranks 0 and 7 of 8 at batch 8 /
context 8191 (two blocks each, so that the CPU restatement finishes in seconds).

The runner is told its rank and group size and given a reducer that stands where the all-reduce
stands (core/layer/model_graph.py:210,219): it keeps what the rank hands in - the rank's partial,
before any reduction - and answers with the ORACLE's reduced tensor, so every block runs on the
oracle's inputs.  Compared per block: the rank's partial of the attention branch and of the MLP branch
against the oracle's partial for that rank (the path's 1e-3 + one bf16 unit), the appended K/V rows,
every untouched cache slot (bit exact), block tables (bit exact), and the logits at the end."""
import copy
import dataclasses
import functools
import random

import pytest
import torch

from conftest import assert_close_bf16
from oracle.ref_cache import RefKVCache
from oracle.ref_model import RefRunner

pytestmark = pytest.mark.gpu
DEV = "cuda"


class TeacherReducer:
    """Stands where the all-reduce stands: records this rank's partial, continues on the oracle's sum."""

    takes_slabs = True

    def __init__(self, sums):
        from micro_vllm_amd.kernel.allreduce import IpcAllReduce
        self.one = IpcAllReduce(0, 1, 64, 8192)      # a group of one: the kernels' own arithmetic
        self.sums = list(sums)                       # the oracle's reduced tensors, in call order
        self.partials = []

    def fits(self, rows, n):
        return self.one.fits(rows, n)

    def status(self):
        return self.one.status()

    def addnorm(self, rows, n, partial=None, slabs=None, ksplit=0, residual=None, y=None, ln_weight=None,
                eps=0.0, normed=None, var_from_rounded=False):
        mine = torch.empty(rows, n, dtype=torch.bfloat16, device=DEV)
        self.one.addnorm(rows, n, partial=partial, slabs=slabs, ksplit=ksplit, y=mine)
        self.partials.append(mine)
        t = self.sums[len(self.partials) - 1].to(DEV)
        self.one.addnorm(rows, n, partial=t, residual=residual, y=y, ln_weight=ln_weight, eps=eps,
                         normed=normed, var_from_rounded=var_from_rounded)


@functools.lru_cache(maxsize=1)
def _cpu_model(shape):
    """Two blocks at the shape's full widths; the vocabulary (replicated on every rank, not part of
    what is sharded; its full size is covered by tests/test_gpu_full_size.py) is cut to 16384 so that
    drawing the random weights does not dominate the test."""
    from micro_vllm_amd.models import SHAPES, build_synthetic_model

    cfg = dataclasses.replace(SHAPES[shape], num_hidden_layers=2, vocab_size=16384)
    return cfg, build_synthetic_model(cfg, "cpu", seed=1234, norm_jitter=0.1)


def _run(shape, world, ranks, bs, ctx, seed=11):
    from micro_vllm_amd import _native as nv
    from micro_vllm_amd.core.cache_manager import KVCacheManager
    from micro_vllm_amd.core.layer.model_graph import ModelGraphRunner

    torch.set_num_threads(16)
    cfg, cpu_model = _cpu_model(shape)
    D, hq, hkv = cfg.head_dim, cfg.num_attention_heads // world, cfg.num_key_value_heads // world
    pages_per_seq = ctx // 256 + 2
    n_blocks, max_tokens = bs * pages_per_seq + 2, pages_per_seq * 256
    # the matrix-core form of the attention kernel rounds P to bf16: use the oracle form that matches
    p_bf16 = nv.lib().mv_paged_attn_decode_form(hq, hkv, D, 256) == 1
    ref = RefRunner(cpu_model, D, world=world, max_blocks=max_tokens // 256, p_bf16=p_bf16)
    # one block table for all ranks (identical on every rank), each rank its own kv heads' pages
    ref_cache = RefKVCache(n_blocks, 256, 2, hkv, D, max_tokens=max_tokens, max_batch_size=bs,
                           strict_reference=False)
    order = list(ref_cache.free_blocks)
    random.Random(seed).shuffle(order)
    ref_cache.free_blocks.clear()
    ref_cache.free_blocks.extend(order)
    lens = [ctx - 5 * i for i in range(bs)]
    for sid, L in enumerate(lens):
        assert ref_cache.alloc(sid, L)[0]
        ref_cache.append(sid)
    pos = [L + 1 for L in lens]
    rbt, rsl = ref_cache.cache_batch_data(list(range(bs)), pos)
    g = torch.Generator().manual_seed(seed)
    full_k = [torch.randn(n_blocks, 256, cfg.num_key_value_heads, D, generator=g).to(torch.bfloat16) for _ in range(2)]
    full_v = [torch.randn(n_blocks, 256, cfg.num_key_value_heads, D, generator=g).to(torch.bfloat16) for _ in range(2)]
    caches = [([t[:, :, r * hkv:(r + 1) * hkv].contiguous() for t in full_k],
               [t[:, :, r * hkv:(r + 1) * hkv].contiguous() for t in full_v]) for r in range(world)]
    before = {r: ([t.clone() for t in caches[r][0]], [t.clone() for t in caches[r][1]]) for r in ranks}
    ids = torch.randint(0, cfg.vocab_size, (bs,), generator=g)
    parts = []
    want_logits = ref.decode(ids, caches, rsl, rbt, parts_trace=parts).float()
    sums = [t for po in parts for t in (po["o_sum"], po["mlp_sum"])]

    for rank in ranks:
        gpu_model = copy.deepcopy(cpu_model).to(DEV)
        teacher = TeacherReducer(sums)
        runner = ModelGraphRunner(gpu_model, 2, hq, D, hkv, cfg.hidden_size, cfg.intermediate_size // world,
                                  DEV, max_batch_size=bs, max_blocks=max_tokens // 256,
                                  release_source_weights=True, reducer=teacher, tp_rank=rank, tp_world=world)
        mgr = KVCacheManager(n_blocks, 256, 2, hkv, D, dtype=torch.bfloat16, device=DEV,
                             max_tokens=max_tokens, max_batch_size=bs)
        mgr._free.clear()
        mgr._free.extend(order)
        for sid, L in enumerate(lens):
            assert mgr.alloc(sid, L)[0]
            mgr.append(sid)
        bt, sl = mgr.cache_batch_data(list(range(bs)), pos)
        assert torch.equal(bt.cpu(), rbt) and torch.equal(sl.cpu(), rsl)         # indexing: bit exact
        for li in range(2):
            mgr.k_caches[li].copy_(before[rank][0][li])
            mgr.v_caches[li].copy_(before[rank][1][li])
        runner._input_ids[:bs] = ids.to(DEV)
        logits = runner.decode(runner._input_ids[:bs], bs, mgr).float().cpu()
        assert teacher.status() == 0 and len(teacher.partials) == 4
        for li in range(2):
            for j, key in enumerate(("o", "mlp")):
                got, want = teacher.partials[2 * li + j], parts[li][key][rank]
                assert_close_bf16(got, want, atol=1e-3, rtol=2 ** -7,
                                  what=f"{shape} rank {rank}/{world} block {li} {key} partial")
            # this rank's pages: only the appended slots changed, and those hold the oracle's rows
            for got_c, want_c, old_c in ((mgr.k_caches[li].cpu(), caches[rank][0][li], before[rank][0][li]),
                                         (mgr.v_caches[li].cpu(), caches[rank][1][li], before[rank][1][li])):
                changed = (got_c != old_c).flatten(2).any(-1)
                hit = {(int(rbt[b, p // 256]), p % 256) for b, p in enumerate(pos)}
                assert {tuple(x) for x in changed.nonzero().tolist()} <= hit
                assert_close_bf16(got_c, want_c, atol=1e-3, rtol=2 ** -7, what=f"appended KV block {li}")
        # the head runs on the oracle's last reduced hidden state: bf16 logits to one unit
        assert_close_bf16(logits, want_logits, atol=1e-3, rtol=2 ** -7, what=f"{shape} rank {rank} logits")
        del runner, mgr, gpu_model, teacher
        torch.cuda.empty_cache()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_llama3_8b_rank_of_a_group_at_batch_16(mv, world):
    _run("llama3-8b", world, ranks=[world - 1], bs=16, ctx=2048)


def test_llama3_70b_ranks_0_and_7_of_8_at_context_8191(mv):
    _run("llama3-70b", 8, ranks=[0, 7], bs=8, ctx=8191)
