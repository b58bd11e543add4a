"""Tensor-parallel reduction of the decode step's [bs, hidden] partials.

The reference sums them with `all_reduce` (core/parallel_config.py:69-74, called at
core/layer/model_graph.py:210,219) and then runs add + RMSNorm as separate kernels.  Two
interchangeable reducers do that here, behind one method, `addnorm`:

  IpcAllReduce   mv_allreduce_* of the C ABI: one launch per rank pushes the partial into every
                 rank's exchange buffer over the direct GPU links, sums what arrived in rank order
                 and applies residual add + RMSNorm (csrc/allreduce.hip).  Needs the ranks on one
                 node, one process per GPU (or, in tests, several processes on one GPU).
  DistAllReduce  torch.distributed (RCCL on GPUs, gloo in rehearsals) followed by the library's
                 norm kernels: the always-available path, also used for prefill-sized messages.

Both produce  t = bf16(sum of partials);  y = bf16(t + residual);  normed = rmsnorm(.) * w  with
the variance from the unrounded t + residual, or from y as stored (`var_from_rounded`).
"""
import ctypes
from typing import Callable, List, Optional

import torch

from .. import _native as nv
from .linear import add_
from .rmsnorm_add import rmsnorm_
from .rmsnorm_residual import rmsnorm_residual_gemm

MAX_ROWS = 64
MAX_N = 8192


class ExchangeTimeout(RuntimeError):
    """A rank waited longer than the communicator's timeout for a peer during a decode step: the
    step's sums (or gathered logits) are wrong and must not be used.  The reference hangs in this
    situation (core/inference_context.py:54); here the step finishes and the host raises this."""

    def __init__(self, what: str, status: int, rank: int):
        super().__init__(f"{what}: rank {rank} gave up waiting for rank {status - 1} "
                         f"(status word {status}); this step's results are invalid")
        self.status, self.rank = status, rank


class IpcAllReduce:
    """Communicator of the one-shot all-reduce.  `exchange(handle: bytes) -> List[bytes]` must
    return every rank's handle in rank order (a host-side all-gather)."""

    takes_slabs = True

    def __init__(self, rank: int, world: int, max_rows: int, max_n: int,
                 exchange: Optional[Callable[[bytes], List[bytes]]] = None):
        """Allocates this rank's exchange buffer; with `exchange` given (and world > 1) also swaps
        handles and maps the peers.  Without it, call `handle()` and `connect()` yourself."""
        L = nv.lib()
        self.rank, self.world, self.max_rows, self.max_n = rank, world, max_rows, max_n
        self._comm = ctypes.c_void_p()
        nv.check("mv_allreduce_init", L.mv_allreduce_init(rank, world, max_rows, max_n,
                                                          ctypes.byref(self._comm)))
        if world > 1 and exchange is not None:
            try:
                self.connect(exchange(self.handle()))
            except Exception:
                self.close()
                raise

    def handle(self) -> bytes:
        """What the other ranks need to map this rank's buffer."""
        L = nv.lib()
        mine = ctypes.create_string_buffer(L.mv_allreduce_handle_bytes())
        nv.check("mv_allreduce_get_handle", L.mv_allreduce_get_handle(self._comm, mine))
        return mine.raw

    def connect(self, handles: List[bytes]):
        """`handles`: every rank's `handle()` in rank order."""
        L = nv.lib()
        nb = L.mv_allreduce_handle_bytes()
        assert len(handles) == self.world and all(isinstance(h, bytes) and len(h) == nb for h in handles)
        blob = ctypes.create_string_buffer(b"".join(handles), nb * self.world)
        nv.check("mv_allreduce_connect", L.mv_allreduce_connect(self._comm, blob))

    def fits(self, rows: int, n: int) -> bool:
        return rows <= self.max_rows and n <= self.max_n and n % 8 == 0

    def set_timeout_ms(self, ms: float):
        nv.check("mv_allreduce_set_timeout_ms", nv.lib().mv_allreduce_set_timeout_ms(self._comm, float(ms)))

    SELF_TEST_ROUNDS = 6

    @staticmethod
    def _self_test_rows(rank: int, rnd: int, n: int) -> torch.Tensor:
        """What `rank` contributes in round `rnd` (CPU, fp32): small integers that differ per rank,
        per round and per column, all exactly representable in bf16, and so is every sum of up to 8
        of them - a stale slot of an earlier round or another rank's slot read twice gives a wrong sum."""
        col = torch.arange(n, dtype=torch.float32) % 7
        return (col + float(3 * rank + 1) + float(5 * (rnd % 3)))[None]

    def self_test(self, timeout_ms: float = 3000.0) -> bool:
        """Known sums through the mapped buffers, every rank at once, SELF_TEST_ROUNDS times in a row
        with no host synchronisation in between (so each of the two data slots is reused while peers
        may still be a round behind) and with data that differs per rank, round and column; every
        rank must read the exact rank-ordered sum every time, with no wait timing out.  Run by all
        ranks right after `connect`, before the communicator is trusted with a decode step: a link
        that maps but does not deliver, or delivers a flag before its data, ends here, in seconds,
        instead of in a step that waits out the full timeout 64 times or returns wrong sums."""
        import os
        n = min(self.max_n, 1024)
        self.set_timeout_ms(timeout_ms)
        ok = True
        try:
            xs = [self._self_test_rows(self.rank, r, n).to(torch.bfloat16).to("cuda")
                  for r in range(self.SELF_TEST_ROUNDS)]
            for x in xs:
                self.all_reduce(x)
            torch.cuda.synchronize()
            for r, x in enumerate(xs):
                want = sum(self._self_test_rows(q, r, n) for q in range(self.world))
                ok = ok and bool((x.float().cpu() == want).all())
            ok = ok and self.status() == 0
        except Exception:
            ok = False
        default = float(os.environ.get("MVLLM_ALLREDUCE_TIMEOUT_MS", "10000") or 10000)
        self.set_timeout_ms(default if default > 0 else 10000.0)
        return ok

    def status(self) -> int:
        """0 unless a wait for a peer timed out since the communicator was made.  A plain read of a
        host word the kernels write: it does not wait, so it speaks for the steps the caller has
        already waited for (e.g. through the read-back of their tokens)."""
        st = ctypes.c_int(0)
        nv.check("mv_allreduce_status", nv.lib().mv_allreduce_status(self._comm, ctypes.byref(st)))
        return st.value

    def raise_if_timed_out(self, what: str = "tensor-parallel exchange"):
        st = self.status()
        if st:
            raise ExchangeTimeout(what, st, self.rank)

    def close(self):
        if self._comm:
            nv.lib().mv_allreduce_destroy(self._comm)
            self._comm = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def all_reduce(self, x: torch.Tensor) -> torch.Tensor:
        """In-place sum over the ranks of a [rows, n] bf16 tensor."""
        assert x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 2 and x.stride(1) == 1
        rows, n = x.shape
        ld = x.stride(0) if rows > 1 else max(n, x.stride(0))
        nv.check("mv_allreduce_step", nv.lib().mv_allreduce_step(
            self._comm, x.data_ptr(), ld, x.data_ptr(), ld, rows, n, nv.stream_ptr()))
        return x

    def addnorm(self, rows: int, n: int, partial: Optional[torch.Tensor] = None,
                slabs: Optional[torch.Tensor] = None, ksplit: int = 0,
                residual: Optional[torch.Tensor] = None, y: Optional[torch.Tensor] = None,
                ln_weight: Optional[torch.Tensor] = None, eps: float = 0.0,
                normed: Optional[torch.Tensor] = None, var_from_rounded: bool = False):
        def ld(t):
            return 0 if t is None else (t.stride(0) if rows > 1 else max(n, t.stride(0)))

        for t in (partial, residual, y, normed):
            assert t is None or (t.dtype == torch.bfloat16 and t.shape == (rows, n) and t.stride(1) == 1)
        assert (partial is None) != (slabs is None)
        if slabs is not None:
            assert slabs.dtype == torch.float32 and slabs.numel() >= ksplit * rows * n and ksplit >= 1
        if ln_weight is not None:
            assert ln_weight.dtype == torch.bfloat16 and ln_weight.numel() == n and ln_weight.is_contiguous()
        nv.check("mv_allreduce_addnorm", nv.lib().mv_allreduce_addnorm(
            self._comm, nv.ptr(partial), ld(partial), nv.ptr(slabs), int(ksplit), nv.ptr(residual),
            ld(residual), nv.ptr(y), ld(y), nv.ptr(ln_weight), float(eps), nv.ptr(normed), ld(normed),
            nv.MV_NORM_VAR_FROM_ROUNDED if var_from_rounded else 0, rows, n, nv.stream_ptr()))


class IpcAllGather(IpcAllReduce):
    """Communicator of the all-gather of [rows, n_local] slices (the logits of an LM head cut by
    vocabulary rows): same buffers, handles, flags and time-out as the all-reduce, nothing summed."""

    takes_slabs = False

    def __init__(self, rank: int, world: int, max_rows: int, n_local: int,
                 exchange: Optional[Callable[[bytes], List[bytes]]] = None):
        L = nv.lib()
        self.rank, self.world, self.max_rows, self.max_n = rank, world, max_rows, n_local
        self._comm = ctypes.c_void_p()
        nv.check("mv_allgather_init", L.mv_allgather_init(rank, world, max_rows, n_local,
                                                          ctypes.byref(self._comm)))
        if world > 1 and exchange is not None:
            try:
                self.connect(exchange(self.handle()))
            except Exception:
                self.close()
                raise

    def gather(self, local: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
        """out[:, r * n_local : (r + 1) * n_local] = rank r's `local`, on every rank."""
        assert local.is_cuda and local.dtype == torch.bfloat16 and local.dim() == 2 and local.stride(1) == 1
        rows, n_local = local.shape
        assert out.dtype == torch.bfloat16 and out.shape == (rows, self.world * n_local) and out.stride(1) == 1
        ld_in = local.stride(0) if rows > 1 else max(n_local, local.stride(0))
        ld_out = out.stride(0) if rows > 1 else max(self.world * n_local, out.stride(0))
        nv.check("mv_allgather_rows", nv.lib().mv_allgather_rows(
            self._comm, local.data_ptr(), ld_in, out.data_ptr(), ld_out, rows, n_local, nv.stream_ptr()))
        return out

    def self_test(self, timeout_ms: float = 3000.0) -> bool:
        """Rank r sends rows that differ per rank, round and column (`_self_test_rows`); every rank
        must read rank q's row in slice q, SELF_TEST_ROUNDS times in a row without host
        synchronisation in between (both data slots reused), with no wait timing out."""
        import os
        n = min(self.max_n, 1024)
        self.set_timeout_ms(timeout_ms)
        ok = True
        try:
            xs = [self._self_test_rows(self.rank, r, n).to(torch.bfloat16).to("cuda")
                  for r in range(self.SELF_TEST_ROUNDS)]
            outs = [torch.zeros(1, self.world * n, dtype=torch.bfloat16, device="cuda") for _ in xs]
            for x, out in zip(xs, outs):
                self.gather(x, out)
            torch.cuda.synchronize()
            for r, out in enumerate(outs):
                want = torch.cat([self._self_test_rows(q, r, n) for q in range(self.world)], dim=1)
                ok = ok and bool((out.float().cpu() == want).all())
            ok = ok and self.status() == 0
        except Exception:
            ok = False
        default = float(os.environ.get("MVLLM_ALLREDUCE_TIMEOUT_MS", "10000") or 10000)
        self.set_timeout_ms(default if default > 0 else 10000.0)
        return ok

    def all_reduce(self, x):
        raise TypeError("an all-gather communicator sums nothing")

    def addnorm(self, *a, **k):
        raise TypeError("an all-gather communicator sums nothing")


class DistAllReduce:
    """The reference's own sequence: all_reduce through torch.distributed, then the norm kernels."""

    takes_slabs = False

    def __init__(self, all_reduce: Callable[[torch.Tensor], torch.Tensor]):
        self._all_reduce = all_reduce

    def fits(self, rows: int, n: int) -> bool:
        return True

    def status(self) -> int:
        return 0

    def raise_if_timed_out(self, what: str = ""):
        pass    # torch.distributed raises (or aborts the process) by itself

    def close(self):
        pass

    def all_reduce(self, x: torch.Tensor) -> torch.Tensor:
        return self._all_reduce(x)

    def addnorm(self, rows: int, n: int, partial=None, slabs=None, ksplit=0, residual=None, y=None,
                ln_weight=None, eps=0.0, normed=None, var_from_rounded=False):
        assert partial is not None and slabs is None, "this reducer takes bf16 rows"
        t = self._all_reduce(partial)
        if residual is None:
            if y is not None and y.data_ptr() != t.data_ptr():
                y.copy_(t)
            if ln_weight is not None:
                rmsnorm_(t, ln_weight, normed, eps)
            return
        if ln_weight is not None and not var_from_rounded:
            # model_graph.py:213: rmsnorm_residual_gemm(attn_out, h, ...) -> (normed, residual)
            rmsnorm_residual_gemm(t, residual, ln_weight, normed, y, eps)
            return
        # model_graph.py:219 `.add_(residual)`, then the next block's rmsnorm_ (:143)
        t = add_(t, residual.contiguous())
        if y is not None and y.data_ptr() != t.data_ptr():
            y.copy_(t)
        if ln_weight is not None:
            rmsnorm_(t, ln_weight, normed, eps)
