"""Per-step control message from rank 0 to the other tensor-parallel ranks.

Same surface as the reference's BatchInferenceContext (core/inference_context.py:9-77:
`broadcast()` on rank 0, `receive(tokenizer)` elsewhere, called in matching pairs).  The
reference pickles a dict per sequence through `broadcast_object_list` (two object broadcasts per
call, each a size exchange plus a payload); here the message is ONE flat int64 tensor with a fixed
header, sent as a 3-word header broadcast plus one payload broadcast, no pickling:

  header : [batch_size, type_code, payload_words]
  payload: per sequence 13 scalar words (floats as IEEE-754 bit patterns), then its
           input_ids and output_ids back to back (full_ids = input_ids + output_ids)
"""
import struct
from dataclasses import dataclass, field
from typing import List

import torch
import torch.distributed as dist

from .parallel_config import get_rank, get_world_size, rank0
from .sequence import Sequence

_TYPES = ["waiting", "prefill", "decode", "idle"]
_STATES = ["prefill", "decode", "finished"]
_NONE = -(1 << 62)


def _f2i(x: float) -> int:
    return struct.unpack("<q", struct.pack("<d", float(x)))[0]


def _i2f(i: int) -> float:
    return struct.unpack("<d", struct.pack("<q", int(i)))[0]


def pack_sequences(seqs: List[Sequence]) -> List[int]:
    words: List[int] = []
    for s in seqs:
        nxt = _NONE if s._next_token is None else int(s._next_token)
        eos = _NONE if s.eos_token_id is None else int(s.eos_token_id)
        # the prompt part of full_ids differs from input_ids once the scheduler has padded them
        n_prompt = len(s.full_ids) - len(s.output_ids)
        words += [int(s.seq_id), int(s.max_tokens), _STATES.index(s.state), int(s.current_position),
                  _f2i(s.temperature), _f2i(s.top_p), eos, int(s.priority), _f2i(s.timestamp), nxt,
                  len(s.input_ids), len(s.output_ids), n_prompt]
        words += [int(t) for t in s.input_ids] + [int(t) for t in s.output_ids]
    return words


def unpack_sequences(words: List[int], count: int, tokenizer) -> List[Sequence]:
    out, p = [], 0
    for _ in range(count):
        (sid, max_tokens, state, pos, temp, top_p, eos, prio, ts, nxt, n_in, n_out, n_prompt) = words[p:p + 13]
        p += 13
        ids = words[p:p + n_in]
        outs = words[p + n_in:p + n_in + n_out]
        p += n_in + n_out
        data = {"seq_id": sid, "max_tokens": max_tokens, "state": _STATES[state], "current_position": pos,
                "temperature": _i2f(temp), "top_p": _i2f(top_p), "eos_token_id": None if eos == _NONE else eos,
                "priority": prio, "timestamp": _i2f(ts), "_next_token": None if nxt == _NONE else nxt,
                "input_ids": ids, "output_ids": outs, "full_ids": ids[:n_prompt] + outs}
        out.append(Sequence.from_dict(data, tokenizer))
    return out


@dataclass
class BatchInferenceContext:
    batch_size: int
    batch_type: str  # "waiting" / "prefill" / "decode"
    sequences: List[Sequence] = field(default_factory=list)
    _device: torch.device = None

    def __post_init__(self):
        self._device = _comm_device()

    def broadcast(self):
        if get_world_size() <= 1:
            return
        if not rank0():
            raise RuntimeError("broadcast can only be called on main rank (Rank 0)")
        payload = [] if self.batch_type == "waiting" else pack_sequences(self.sequences)
        head = torch.tensor([self.batch_size, _TYPES.index(self.batch_type), len(payload)],
                            dtype=torch.int64, device=self._device)
        dist.broadcast(head, src=0)
        if payload:
            dist.broadcast(torch.tensor(payload, dtype=torch.int64, device=self._device), src=0)

    @classmethod
    def receive(cls, _tokenizer):
        if rank0():
            raise RuntimeError("receive can only be called on non-main rank")
        dev = _comm_device()
        head = torch.zeros(3, dtype=torch.int64, device=dev)
        dist.broadcast(head, src=0)
        batch_size, type_code, n_words = (int(v) for v in head.tolist())
        ctx = cls(batch_size=batch_size, batch_type=_TYPES[type_code])
        if n_words:
            payload = torch.zeros(n_words, dtype=torch.int64, device=dev)
            dist.broadcast(payload, src=0)
            ctx.sequences = unpack_sequences(payload.tolist(), batch_size, _tokenizer)
        return ctx


def _comm_device() -> torch.device:
    """Tensors of a collective live where the backend works: host for gloo, this rank's GPU for RCCL."""
    if dist.is_initialized() and dist.get_backend() == "gloo":
        return torch.device("cpu")
    if torch.cuda.is_available():
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")
