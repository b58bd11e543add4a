"""Engine glue around the two runners: request intake, one scheduler decision per step, prefill /
decode drivers, token bookkeeping.  Mirrors the reference's InferenceEngine
(core/engine.py:35-309) method for method; the model comes from a safetensors checkpoint
directory (`model_path`, core/model_loader.py) or is handed in as an object with its tokenizer,
and the KV pool geometry is a constructor argument instead of the reference's fixed 81 blocks /
4-column block table.
"""
import logging
import time
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch

from .cache_manager import KVCacheManager
from .inference_context import BatchInferenceContext
from .layer.model_graph import ModelGraphRunner
from .layer.model_prefill import ModelPrefillRunner
from .layer.sampler import Sampler, greedy
from .parallel_config import get_rank, get_world_size, rank0, setup
from .scheduler import Scheduler
from .sequence import Sequence

logger = logging.getLogger("InferenceEngine")


class KVPoolExhausted(RuntimeError):
    """A decode step could not get a KV slot for `seq_ids`; nothing of the step was run and the other
    sequences of the batch are as they were before it."""

    def __init__(self, seq_ids):
        super().__init__("OOM: no free KV page for sequence(s) %s" % ", ".join(str(i) for i in seq_ids))
        self.seq_ids = list(seq_ids)


@dataclass
class InferenceStats:
    prep_time: float = 0.0
    gpu_time: float = 0.0
    sample_time: float = 0.0
    total_time: float = 0.0


class InferenceEngine:
    DEFAULT_BLOCK_SIZE = 256
    DEFAULT_MAX_BLOCKS = 81

    def __init__(self, model_path: Optional[str] = None, max_batch_size: int = 40,
                 max_prefill_tokens: int = 2048, *, model=None, tokenizer=None,
                 n_blocks: Optional[int] = None, max_tokens: int = 1024,
                 graph_batch_sizes: Optional[List[int]] = None, prefill_timeout: float = 0.02,
                 exact_rope: bool = False, trust_remote_code: bool = False, kv_fraction: float = 0.9,
                 clear_unwritten_slot: bool = True, sample_last_real_token: bool = False):
        """`max_tokens` is the longest sequence a block-table row can describe (the reference's
        KVCacheManager default, 1024, gives four columns - pass more for long contexts).
        `n_blocks`: None = the reference's 81 pages (core/engine.py:53-57), a number, or "auto" =
        `kv_fraction` of the device memory that is free once the weights are in place (on a
        288 GB device an 8B model leaves room for more than a million cached tokens).
        `exact_rope` builds the rotary tables from fp32 positions; the default keeps the
        reference's bf16 positions (core/paged_attention.py:50-52), which merge neighbouring
        positions above 256 - with a trained checkpoint and long contexts pass True.
        `clear_unwritten_slot`: the reference's position numbering (core/sequence.py:17,42 with
        core/engine.py:262) leaves token index S of a sequence prefilled with S (padded) tokens
        unwritten - the first generated token lands at S + 1 - and attends to it ever after.  In
        the reference that slot holds zeros on a fresh pool and another request's keys once pages
        have been recycled; here it is zeroed on the sequence's first decode step, so outputs are
        those of the reference on a fresh pool and never depend on what a page held before.  False
        leaves the slot as it is.
        `sample_last_real_token`: take the first generated token of a right-padded prompt from the
        logits of its last REAL token instead of the batch's last (pad) position, which is what the
        reference samples from (core/engine.py:233 with core/scheduler.py:157-163)."""
        setup()
        self.rank = get_rank()
        if model is None:
            if model_path is None:
                raise ValueError("pass model_path= (a safetensors checkpoint directory) or model=")
            from .model_loader import load_model
            logger.info("Loading model %s on rank %d", model_path, self.rank)
            model, loaded_tok = load_model(model_path, tokenizer=tokenizer is None,
                                           trust_remote_code=trust_remote_code)
            tokenizer = tokenizer if tokenizer is not None else loaded_tok
        assert torch.cuda.is_available(), "the HIP library is the only compute path"
        self.device_index = self.rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(self.device_index)   # per thread: a serving thread sets it again
        self.device, self.dtype = "cuda", torch.bfloat16
        self.model, self.tokenizer = model, tokenizer
        self.config = model.config

        # per-rank geometry (core/engine.py:101-114)
        cfg = self.config
        world = get_world_size()
        self.num_layers = cfg.num_hidden_layers
        heads = cfg.num_attention_heads
        kv_heads = getattr(cfg, "num_key_value_heads", heads)
        self.head_size = getattr(cfg, "head_dim", cfg.hidden_size // heads)
        assert heads % world == 0 and kv_heads % world == 0
        # what the attention kernels are built for - checked before any weight moves to the device
        if self.head_size not in (64, 128) or heads % kv_heads or not 1 <= heads // kv_heads <= 8:
            raise ValueError(
                f"unsupported attention shape: head size {self.head_size} (supported: 64, 128), "
                f"{heads} query heads on {kv_heads} kv heads = {heads / kv_heads:g} per kv head "
                f"(supported: 1 to 8)")
        self.num_heads, self.kv_num_heads = heads // world, kv_heads // world
        self.intermediate_size = cfg.intermediate_size // world

        max_blocks = max(32, (max_tokens + 255) // 256)
        runner_args = dict(model=model, num_layers=self.num_layers, num_heads=self.num_heads,
                           head_size=self.head_size, kv_num_heads=self.kv_num_heads,
                           hidden_dim=cfg.hidden_size, intermediate_size=self.intermediate_size,
                           device=self.device, max_batch_size=max_batch_size, dtype=self.dtype,
                           max_blocks=max_blocks, exact_rope=exact_rope)
        # the runners first: they put this rank's weights on the device, and a pool sized from
        # what is left has to see that
        self.graph_runner = ModelGraphRunner(**runner_args)
        self.prefill_runner = ModelPrefillRunner(**runner_args)
        if n_blocks == "auto":
            blocks = self.auto_kv_blocks(kv_fraction)
        else:
            blocks = n_blocks if n_blocks is not None else self.DEFAULT_MAX_BLOCKS
        self.cache_manager = KVCacheManager(
            n_blocks=blocks, block_size=self.DEFAULT_BLOCK_SIZE, n_layers=self.num_layers,
            n_heads=self.kv_num_heads, head_size=self.head_size, dtype=self.dtype, device=self.device,
            max_tokens=max_tokens, max_batch_size=max_batch_size)
        self.scheduler = Scheduler(max_batch_size, max_prefill_tokens, tokenizer, prefill_timeout)
        self.sampler = Sampler()
        self.clear_unwritten_slot = clear_unwritten_slot
        self.sample_last_real_token = sample_last_real_token

        # The synchronous token read-back at the end of a step means a staging row is never
        # overwritten before its copy has run; two rows keep that true if a caller pipelines.
        self._ids_host = [torch.zeros(max_batch_size, dtype=torch.long, pin_memory=True) for _ in range(2)]
        self._ids_slot = 0
        self._sample_params = None   # (temperature, top_p) per row as last sent to the runner
        self.eos_token_id = getattr(tokenizer, "eos_token_id", None)
        self.stream_callbacks = {}
        self.max_position = getattr(cfg, "max_position_embeddings", 4096)
        sizes = graph_batch_sizes or [b for b in (1, 2, 4, 8, 16, 32, 40) if b <= max_batch_size]
        self.scheduler.batch_sizes = [b for b in sizes if b <= max_batch_size] or [max_batch_size]
        self.graph_runner.capture(self.cache_manager, batch_sizes=self.scheduler.batch_sizes)

    def auto_kv_blocks(self, fraction: float) -> int:
        """Pages of DEFAULT_BLOCK_SIZE tokens (K and V, all layers, this rank's kv heads) that fit
        into `fraction` of the currently free device memory; never fewer than the reference's pool."""
        torch.cuda.synchronize()
        free, _ = torch.cuda.mem_get_info()
        per_block = 2 * self.DEFAULT_BLOCK_SIZE * self.kv_num_heads * self.head_size * 2 * self.num_layers
        blocks = max(self.DEFAULT_MAX_BLOCKS, int(free * fraction) // per_block)
        if get_world_size() > 1:
            # every rank mirrors the same allocator state, so all take the smallest pool
            from .inference_context import _comm_device
            t = torch.tensor([blocks], dtype=torch.int64, device=_comm_device())
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MIN)
            blocks = int(t.item())
        return blocks

    def shutdown(self):
        try:
            if torch.distributed.is_initialized():
                torch.distributed.destroy_process_group()
        except Exception as e:  # best effort at exit
            logger.warning("error during shutdown: %s", e)

    # ------------------------------------------------------------------ public API
    def add_request(self, prompt: str, max_tokens: int = 128, temperature: float = 0.7,
                    top_p: float = 0.9) -> int:
        seq_id = hash(prompt + str(time.time())) % (2 ** 32)
        seq = Sequence(seq_id, prompt, self.tokenizer, max_tokens)
        seq.temperature, seq.top_p = temperature, top_p
        # A prompt no prefill step may hold, or one whose continuation would run past the last
        # block-table column, can never be served; the reference fails on it later, inside the
        # scheduler (TypeError) or the cache manager.  Refused here, with the limits named.
        n = len(seq.input_ids)
        rows = self.cache_manager.max_seq_blocks * self.cache_manager.block_size
        if n == 0 or n > self.scheduler.max_prefill_tokens or n + max_tokens + 1 > rows:
            raise ValueError(
                f"prompt of {n} tokens with max_tokens={max_tokens} cannot be scheduled: "
                f"max_prefill_tokens={self.scheduler.max_prefill_tokens}, block table covers {rows} positions")
        self.scheduler.add_request(seq)
        return seq_id

    def register_stream_callback(self, seq_id: int, callback):
        self.stream_callbacks[seq_id] = callback

    def unregister_stream_callback(self, seq_id: int):
        self.stream_callbacks.pop(seq_id, None)

    def get_next_batch(self) -> Tuple[List[Sequence], str]:
        return self.scheduler.get_next_batch()

    @torch.inference_mode()
    def step(self, ctx: BatchInferenceContext) -> bool:
        if not ctx.sequences:
            return False
        if ctx.batch_type == "prefill":
            self._prefill(ctx.sequences)
        else:
            self._decode(ctx.sequences)
        return True

    def generate(self, prompts: List[str], max_tokens: int = 100, temperature: float = 0.7,
                 top_p: float = 0.9) -> Dict[str, str]:
        """Run the prompts to completion on this process (single-rank event loop).  Unlike the
        reference's `generate` (core/engine.py:178-199), which never applies the sampled tokens
        and so returns nothing, this loop calls `update_sequences` after every step the way the
        server loop does (api_server.py:64-76)."""
        seq_ids = [self.add_request(p, max_tokens, temperature, top_p) for p in prompts]
        by_id = dict(zip(seq_ids, prompts))
        results = {}
        budget = (max_tokens + 2) * max(1, len(prompts)) + 16
        for _ in range(budget * 50):
            batch, kind = self.get_next_batch()
            if kind == "waiting":
                time.sleep(self.scheduler.prefill_timeout / 4)
                continue
            if not batch:
                break
            ctx = BatchInferenceContext(len(batch), kind, batch)
            self.step(ctx)
            self.update_sequences(ctx.sequences)
        for seq, ids in self.scheduler.get_finished_results():
            try:
                results[by_id[seq.seq_id]] = self.tokenizer.decode(ids, skip_special_tokens=True)
            except Exception:
                results[by_id.get(seq.seq_id, str(seq.seq_id))] = "[Error]"
        self.scheduler.running_sequences.clear()
        return results

    # ------------------------------------------------------------------ step drivers
    def _sample(self, logits, temps, topp, top_k):
        if bool((temps <= 0).all()):
            return greedy(logits.contiguous()).tolist()          # deterministic limit, on device
        # one launch: temperature, top-k, softmax, top-p and the draw (rows with temperature 0 are
        # resolved greedily inside the kernel)
        return self.sampler(logits, temps, topp, min(top_k, logits.shape[-1])).tolist()

    def _prefill(self, batch: List[Sequence]):
        """core/engine.py:205-239: the batch arrives right-padded to one length; pages for that
        length are allocated for every sequence; the token is sampled from the LAST position."""
        device = self.device
        input_ids = torch.tensor([seq.get_next_input_ids() for seq in batch], dtype=torch.long, device=device)
        temps = torch.tensor([seq.temperature for seq in batch], device=device)
        topp = torch.tensor([seq.top_p for seq in batch], device=device)
        seq_lens = [input_ids.shape[1]] * len(batch)
        taken = []
        for seq in batch:
            ok, _ = self.cache_manager.alloc(seq.seq_id, seq_lens[0])
            if not ok:
                # give back what this batch took so far: the caller may retry or fail the requests,
                # the pool must not shrink either way (the reference leaks these pages)
                for sid in taken:
                    self.cache_manager.free(sid)
                raise RuntimeError("OOM")
            taken.append(seq.seq_id)
        self.cache_manager.cache_batch_data([s.seq_id for s in batch], seq_lens)
        last = None
        if self.sample_last_real_token:
            # full_ids still holds the prompt as it was before the scheduler padded input_ids
            last = [len(seq.full_ids) - len(seq.output_ids) - 1 for seq in batch]
        logits = self.prefill_runner.forward(input_ids, self.cache_manager, len(batch), last_token_only=True,
                                             last_token_index=last)
        if rank0():
            next_tokens = self._sample(logits[:, -1, :], temps, topp, 1000)
            for i, seq in enumerate(batch):
                seq._next_token = next_tokens[i]

    def _decode(self, batch: List[Sequence]):
        """core/engine.py:241-285: padded batches repeat sequences, so appends are de-duplicated;
        cache_seqlens = current_position (already advanced past the prompt)."""
        device = self.device
        batch_size = len(batch)
        seen, first = set(), []
        for seq in batch:
            first.append(seq.seq_id not in seen)
            seen.add(seq.seq_id)
        # one pinned staging row per step in flight; the copy into the runner's static input buffer is
        # the only host-to-device traffic besides the block table
        ids_host = self._ids_host[self._ids_slot]
        self._ids_slot = (self._ids_slot + 1) % len(self._ids_host)
        for i, seq in enumerate(batch):
            ids_host[i] = seq.get_next_input_ids()[0]
        input_ids = ids_host[:batch_size].to(device, non_blocking=True)
        all_greedy = all(seq.temperature <= 0 for seq in batch)
        taken, starved = [], []
        for i, seq in enumerate(batch):
            if first[i]:
                slot = self.cache_manager.append(seq.seq_id)
                (starved if slot == -1 else taken).append((seq.seq_id, slot))
        if starved:
            # the step will not run: the sequences that did get a slot give it back, so that they
            # can take part in the next step as if this one had not been tried, and the caller
            # learns which sequences the pool had no page for (the reference fails the whole batch
            # and leaves the others one slot ahead, core/engine.py:262-266)
            for sid, slot in reversed(taken):
                self.cache_manager.unappend(sid, slot)
            raise KVPoolExhausted([sid for sid, _ in starved])
        if self.clear_unwritten_slot:
            # first decode step of a sequence (one token generated so far, by prefill): index
            # current_position - 1 has never been written and is attended from now on
            fresh = [seq for i, seq in enumerate(batch) if first[i] and len(seq.output_ids) == 1]
            if fresh:
                self.cache_manager.clear_slots([s.seq_id for s in fresh],
                                               [s.current_position - 1 for s in fresh])
        self.cache_manager.cache_batch_data([s.seq_id for s in batch], [s.current_position for s in batch])
        # The pick (argmax, or temperature / top-k / top-p draw) is part of the captured step: the
        # runner's static per-row parameters are rewritten when the batch composition changes, the
        # uniform numbers before every step that draws, and the only wait of the step is the
        # read-back of the ids.  (Launching the sampler behind the graph instead costs 0.3 ms per
        # step at bs 8: the launch has to wait for the graph's own submission.)
        key = tuple((max(float(seq.temperature), 0.0), float(seq.top_p)) for seq in batch)
        if self._sample_params != key:
            self.graph_runner.set_sampling([t for t, _ in key], [p for _, p in key])
            self._sample_params = key
        if not all_greedy:
            self.graph_runner.draw_uniforms(batch_size)
        logits = self.graph_runner.forward(input_ids, self.cache_manager, batch_size)
        if rank0():
            ids = self.graph_runner.picked_ids(logits)
            if ids is not None:
                next_tokens = ids.tolist()
            else:   # logits from somewhere else than the runner's static buffer
                temps = torch.tensor([t for t, _ in key], device=device)
                topp = torch.tensor([p for _, p in key], device=device)
                next_tokens = self._sample(logits, temps, topp, 50)
            # the read-back above waited for the step: a reduction or gather of it that gave up on a
            # peer has set its communicator's (host-visible) status word by now, and the tokens are
            # thrown away before anybody sees them
            self.graph_runner.raise_if_exchange_failed()
            for i, seq in enumerate(batch):
                seq._next_token = next_tokens[i]
        else:
            # no read-back on this rank: the word speaks for the steps that have finished so far,
            # so a failed exchange is noticed a step late at worst
            self.graph_runner.raise_if_exchange_failed()

    def update_sequences(self, sequences: List[Sequence]):
        """core/engine.py:287-309: apply each sequence's sampled token once, stream it, retire
        finished sequences and return their pages."""
        done = set()
        for seq in sequences:
            if seq._next_token is None or seq.seq_id in done:
                continue
            done.add(seq.seq_id)
            seq.update_state(seq._next_token, None)
            if rank0():
                cb = self.stream_callbacks.get(seq.seq_id)
                if cb is not None:
                    try:
                        cb(seq._next_token, self.tokenizer.decode([seq._next_token], skip_special_tokens=True))
                    except Exception:
                        pass
            if seq.is_finished():
                self.cache_manager.free(seq.seq_id)
                if rank0():
                    self.scheduler.mark_finished(seq)
