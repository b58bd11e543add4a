"""Time mv_wgemm for one (M, N, K) over distinct weight copies (no cache reuse), per tuning setting.

  python tools/bench_wgemm.py M N K [copies]
Reports device time per launch from HIP events around a hipGraph of `copies` launches.
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import micro_vllm_amd  # noqa: E402,F401
from micro_vllm_amd import _native as nv  # noqa: E402
from micro_vllm_amd.kernel.linear import StreamedWeight, wgemm, wgemm_norm  # noqa: E402


def timed_graph(fn, reps=20):
    L = nv.lib()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        fn()
        side.synchronize()
        nv.check("b", L.mv_graph_begin_capture(side.cuda_stream))
        fn()
        h = ctypes.c_void_p()
        nv.check("e", L.mv_graph_end_capture(side.cuda_stream, ctypes.byref(h)))
        for _ in range(3):
            L.mv_graph_launch(h, side.cuda_stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(side)
        for _ in range(reps):
            L.mv_graph_launch(h, side.cuda_stream)
        e1.record(side)
        side.synchronize()
        L.mv_graph_destroy(h)
    return e0.elapsed_time(e1) * 1e-3 / reps


def main():
    m, n, k = (int(v) for v in sys.argv[1:4])
    copies = int(sys.argv[4]) if len(sys.argv) > 4 else max(4, int(600e6 / (n * k * 2)) + 1)
    dev = "cuda"
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    ws = [StreamedWeight(torch.empty(n * k, dtype=torch.bfloat16, device=dev).normal_(0, 0.02, generator=g), n, k)
          for _ in range(copies)]
    x = torch.randn(m, k, device=dev).to(torch.bfloat16)
    lw = torch.ones(k, dtype=torch.bfloat16, device=dev)
    y = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
    L = nv.lib()
    wsb = torch.empty(8 * m * n + 16, dtype=torch.float32, device=dev)
    nbytes = n * k * 2
    print(f"M={m} N={n} K={k} copies={copies} ({nbytes/1e6:.1f} MB each)")
    for norm in (False, True):
        if norm and m > 4:
            continue
        for ks in (1, 2, 4):
            for bpc in (1, 2, 3, 4):
                L.mv_wgemm_tune(1, ks)
                L.mv_wgemm_tune(0, bpc)

                def fn():
                    for w in ws:
                        if norm:
                            wgemm_norm(x, lw, 1e-5, w, y, workspace=wsb)
                        else:
                            wgemm(x, w, y, workspace=wsb)
                t = timed_graph(fn) / copies
                print(f"  norm={int(norm)} ksplit={ks} bpc={bpc}: {t*1e6:8.2f} us  {nbytes/t/1e9:8.1f} GB/s", flush=True)
    L.mv_wgemm_tune(1, 0)
    L.mv_wgemm_tune(0, 0)


if __name__ == "__main__":
    main()
