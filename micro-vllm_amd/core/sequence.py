"""One generation request and its token state.  Same fields, state machine and numbering as the
reference's Sequence (core/sequence.py:6-109); in particular `current_position` starts at the
prompt length and is advanced BEFORE the first decode step reads it (:17, :42)."""
import time
from typing import List, Optional


class Sequence:
    # result of the last step, kept on every rank until update_state consumes it
    _next_token: Optional[int] = None

    def __init__(self, seq_id: int, prompt: str, tokenizer, max_tokens: int = 128):
        self.seq_id = seq_id
        self.prompt = prompt
        self.tokenizer = tokenizer
        self.max_tokens = max_tokens
        self.input_ids: List[int] = tokenizer.encode(prompt, add_special_tokens=True)
        self.output_ids: List[int] = []
        self.full_ids: List[int] = list(self.input_ids)
        self.state = "prefill"           # prefill -> decode -> finished
        self.past_key_values = None      # unused by the paged path; kept for interface parity
        self.current_position = len(self.input_ids)
        self.temperature = 0.7
        self.top_p = 0.9
        self.eos_token_id = tokenizer.eos_token_id
        self.priority = 0
        self.timestamp = time.time()     # arrival time, read by the prefill batching timeout

    def is_finished(self):
        if len(self.output_ids) >= self.max_tokens:
            return True
        return bool(self.output_ids) and self.output_ids[-1] == self.eos_token_id

    def get_next_input_ids(self):
        if self.state == "prefill":
            return self.input_ids
        if self.state == "decode":
            return [self.output_ids[-1]]
        return None

    def update_state(self, next_token: int, new_past_key_values=None):
        self.output_ids.append(next_token)
        self.full_ids.append(next_token)
        self.past_key_values = new_past_key_values
        self.current_position += 1
        if self.is_finished():
            self.state = "finished"
        elif self.state == "prefill":
            self.state = "decode"

    # ---- plain-data form (tensor-parallel control messages, core/sequence.py:49-109) ----
    _FIELDS = ("seq_id", "max_tokens", "state", "current_position", "temperature", "top_p",
               "eos_token_id", "priority", "timestamp", "_next_token")

    def to_dict(self) -> dict:
        d = {k: getattr(self, k) for k in self._FIELDS}
        d["input_ids"] = list(self.input_ids)
        d["output_ids"] = list(self.output_ids)
        d["full_ids"] = list(self.full_ids)
        return d

    @classmethod
    def from_dict(cls, data: dict, dummy_tokenizer):
        seq = cls(data["seq_id"], "", dummy_tokenizer, data["max_tokens"])
        seq.input_ids = list(data["input_ids"])
        seq.output_ids = list(data["output_ids"])
        seq.full_ids = list(data["full_ids"])
        for k in cls._FIELDS:
            if k not in ("seq_id", "max_tokens"):
                setattr(seq, k, data[k])
        return seq
