"""Host-side glue without a GPU: the scheduler against the trace recorded from the reference's
Scheduler/Sequence (tests/golden/scheduler_trace.json, scripted arrivals, stub tokenizer, patched
clock), and the fixed-layout control message (pack/unpack and a 2-rank gloo broadcast)."""
import json
import os

import pytest
import torch
import torch.multiprocessing as mp

GOLD = os.path.join(os.path.dirname(__file__), "golden", "scheduler_trace.json")


class Tok:
    eos_token_id = 2

    def encode(self, prompt, add_special_tokens=True):
        return [5 + (ord(c) % 50) for c in prompt]

    def decode(self, ids, skip_special_tokens=True):
        return " ".join(str(i) for i in ids)


def test_scheduler_reproduces_the_reference_trace():
    import micro_vllm_amd  # noqa: F401
    from micro_vllm_amd.core.scheduler import Scheduler
    from micro_vllm_amd.core.sequence import Sequence

    gold = json.load(open(GOLD))
    clock = [1000.0]
    tok = Tok()
    sch = Scheduler(max_batch_size=gold["max_batch_size"], max_prefill_tokens=gold["max_prefill_tokens"],
                    tokenizer=tok, clock=lambda: clock[0])
    log = []

    def snap(tag):
        batch, kind = sch.get_next_batch()
        log.append({"tag": tag, "t": clock[0], "type": kind, "seq_ids": [s.seq_id for s in batch],
                    "lens": [len(s.input_ids) for s in batch],
                    "padded_tail": [s.input_ids[-3:] for s in batch],
                    "running": [s.seq_id for s in sch.running_sequences],
                    "waiting": [s.seq_id for s in sch.waiting_queue]})
        return batch, kind

    for i, L in enumerate(gold["lens"]):
        seq = Sequence(i, "a" * L, tok, max_tokens=3)
        seq.timestamp = clock[0]
        sch.add_request(seq)
        clock[0] += 0.001
    snap("before_timeout")
    clock[0] += 0.05
    for tag in ("after_timeout_bucket0", "next_bucket", "bucket100"):
        b, _ = snap(tag)
        for s in b:
            s.update_state(7, None)
    for step in range(4):
        b, _ = snap(f"decode{step}")
        seen = set()
        for s in b:
            if s.seq_id in seen:
                continue
            seen.add(s.seq_id)
            s.update_state(7 if step < 1 else 2, None)
    snap("drained")
    assert len(log) == len(gold["log"])
    for got, want in zip(log, gold["log"]):
        assert got == want, f"{got['tag']}: {got} != {want}"


def test_sequence_state_machine_and_position_numbering():
    import micro_vllm_amd  # noqa: F401
    from micro_vllm_amd.core.sequence import Sequence

    s = Sequence(9, "abcd", Tok(), max_tokens=2)
    assert s.current_position == 4 and s.state == "prefill" and s.get_next_input_ids() == s.input_ids
    s.update_state(11, None)
    assert s.current_position == 5 and s.state == "decode" and s.get_next_input_ids() == [11]
    s.update_state(2, None)            # eos
    assert s.is_finished() and s.state == "finished" and s.get_next_input_ids() is None
    d = s.to_dict()
    t = Sequence.from_dict(d, Tok())
    assert t.to_dict() == d


def _mk_seqs():
    from micro_vllm_amd.core.sequence import Sequence

    a = Sequence(2 ** 31 + 5, "hello", Tok(), max_tokens=7)
    a.temperature, a.top_p = 0.0, 0.95
    a.input_ids = a.input_ids + [0, 0]          # scheduler padding
    a._next_token = 17
    b = Sequence(3, "xy", Tok(), max_tokens=4)
    b.update_state(9, None)
    b.update_state(8, None)
    return [a, b]


def test_control_message_round_trip():
    import micro_vllm_amd  # noqa: F401
    from micro_vllm_amd.core.inference_context import pack_sequences, unpack_sequences

    seqs = _mk_seqs()
    back = unpack_sequences(pack_sequences(seqs), len(seqs), Tok())
    for s, t in zip(seqs, back):
        assert s.to_dict() == t.to_dict()


def _rank_main(rank, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      MVLLM_DIST_BACKEND="gloo")
    import micro_vllm_amd  # noqa: F401
    from micro_vllm_amd.core import parallel_config as pc
    from micro_vllm_amd.core.inference_context import BatchInferenceContext

    pc.setup()
    if rank == 0:
        BatchInferenceContext(0, "waiting").broadcast()
        seqs = _mk_seqs()
        BatchInferenceContext(len(seqs), "decode", seqs).broadcast()
        q.put(("sent", [s.to_dict() for s in seqs]))
    else:
        c0 = BatchInferenceContext.receive(Tok())
        c1 = BatchInferenceContext.receive(Tok())
        q.put(("got", [c0.batch_type, c0.batch_size, c1.batch_type, c1.batch_size,
                       [s.to_dict() for s in c1.sequences]]))
    pc.barrier()
    torch.distributed.destroy_process_group()


def test_control_message_over_gloo_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_rank_main, args=(r, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    msgs = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    kinds = msgs["got"]
    assert kinds[:4] == ["waiting", 0, "decode", 2]
    assert kinds[4] == msgs["sent"]
