"""Continuous batching policy of the reference (core/scheduler.py:11-206), decision for decision:

* prefill first: waiting prompts are grouped into length buckets of 50 tokens; the shortest
  bucket that is ready (max_batch_size prompts, or its oldest prompt has waited prefill_timeout)
  is launched, longest prompt first, within max_prefill_tokens, right-padded with token 0 to the
  longest length of the batch;
* otherwise decode: running sequences are grouped by current_position, the group with the
  smallest position runs (at most max_batch_size of them), padded up to the next captured batch
  size by repeating its members cyclically (the engine de-duplicates the cache appends).
"""
import logging
import time
from collections import defaultdict, deque
from typing import Callable, List, Tuple

from .sequence import Sequence

logger = logging.getLogger(__name__)


class Scheduler:
    def __init__(self, max_batch_size: int = 32, max_prefill_tokens: int = 2048, tokenizer=None,
                 prefill_timeout: float = 0.02, clock: Callable[[], float] = None):
        self.tokenizer = tokenizer
        self.max_batch_size = max_batch_size
        self.max_prefill_tokens = max_prefill_tokens
        self.prefill_timeout = prefill_timeout
        self.bucket_size = 50
        self.waiting_queue = deque()
        self.running_sequences: List[Sequence] = []
        self.finished_sequences: List[Sequence] = []
        self.batch_sizes = [1, 2, 4, 8, 16, 32]   # graph-captured decode batch sizes
        self._clock = clock or time.time

    def _get_bucket_key(self, length: int) -> int:
        return length - length % self.bucket_size

    def add_request(self, seq: Sequence):
        self.waiting_queue.append(seq)

    def get_next_batch(self) -> Tuple[List[Sequence], str]:
        self.running_sequences = [s for s in self.running_sequences if not s.is_finished()]
        if self.waiting_queue:
            batch, kind = self._get_prefill_batch()
            if kind != "idle":
                return batch, kind           # "prefill", or "waiting" (let the caller yield)
        return self._get_decode_batch()

    def _get_decode_batch(self) -> Tuple[List[Sequence], str]:
        by_position = defaultdict(list)
        for seq in self.running_sequences:
            if seq.state == "decode" and not seq.is_finished():
                by_position[seq.current_position].append(seq)
        if not by_position:
            return [], "idle"
        group = by_position[min(by_position)][:self.max_batch_size]
        target = next((b for b in self.batch_sizes if b >= len(group)), self.batch_sizes[-1])
        padded = list(group)
        i = 0
        while len(padded) < target:
            padded.append(group[i % len(group)])
            i += 1
        return padded, "decode"

    def _get_prefill_batch(self) -> Tuple[List[Sequence], str]:
        buckets = defaultdict(list)
        for seq in self.waiting_queue:
            if seq.state == "prefill":
                buckets[self._get_bucket_key(len(seq.input_ids))].append(seq)
        if not buckets:
            return [], "idle"
        for key in sorted(buckets):
            members = sorted(buckets[key], key=lambda s: len(s.input_ids), reverse=True)  # stable
            chosen, tokens, oldest = [], 0, None
            for seq in members:
                if len(chosen) >= self.max_batch_size:
                    break
                n = len(seq.input_ids)
                if tokens + n > self.max_prefill_tokens:
                    continue
                chosen.append(seq)
                tokens += n
                oldest = seq.timestamp if oldest is None else min(oldest, seq.timestamp)
            if not chosen:
                # every prompt of this bucket exceeds max_prefill_tokens on its own; the reference
                # raises a TypeError here (timestamp None, core/scheduler.py:152) - skip the bucket
                continue
            waited = self._clock() - oldest
            if len(chosen) >= self.max_batch_size or waited >= self.prefill_timeout:
                longest = max(len(s.input_ids) for s in chosen)
                for seq in chosen:
                    seq.input_ids = seq.input_ids + [0] * (longest - len(seq.input_ids))
                for seq in chosen:
                    self.waiting_queue.remove(seq)
                    self.running_sequences.append(seq)
                logger.info("prefill bucket %d: %d sequences, max_len %d, tokens %d, waited %.3fs",
                            key, len(chosen), longest, tokens, waited)
                return chosen, "prefill"
        return [], "waiting"

    def mark_finished(self, seq: Sequence):
        if seq in self.running_sequences:
            self.running_sequences.remove(seq)
        self.finished_sequences.append(seq)

    def get_finished_results(self):
        done = [(seq, seq.full_ids) for seq in self.finished_sequences]
        self.finished_sequences.clear()
        return done

    def is_finished(self, seq_id: int) -> bool:
        """True when the id is neither waiting nor running."""
        return all(s.seq_id != seq_id for s in self.waiting_queue) and \
            all(s.seq_id != seq_id for s in self.running_sequences)
