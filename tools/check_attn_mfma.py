"""Matrix-core form of the decode attention against the dot-product form (same inputs, forced by
mv_paged_attn_tune key 2), then both timed on the shapes that matter."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import micro_vllm_amd
from micro_vllm_amd import _native as nv
from test_gpu_attention import make_case, run_both
from bench_attn import run
L = nv.lib()
worst = 0.0
for (hq, hkv) in [(8, 2), (8, 1), (16, 2), (4, 1)]:
    for ctxs in ([0, 1, 15, 16, 17, 31, 32, 33, 255, 256, 257, 700], [4095, 4096, 1000, 129]):
        for splits in (1, 3, 0, 8):
            case = make_case(len(ctxs), hq, hkv, 128, 256, ctxs, seed=hq + hkv + splits)
            outs = []
            for mode in (1, 2):
                assert L.mv_paged_attn_tune(2, mode) == 0
                got, want, (kc, vc), (kc_ref, vc_ref) = run_both(case, splits)
                torch.cuda.synchronize()
                assert torch.equal(kc.cpu(), kc_ref) and torch.equal(vc.cpu(), vc_ref), (hq, hkv, splits, mode, "append")
                outs.append(got.float().cpu())
            err = (outs[0] - outs[1]).abs().max().item()
            ref_err = (outs[1] - want.float()).abs().max().item()
            worst = max(worst, err)
            print(f"hq={hq} hkv={hkv} ctx[0..]={ctxs[:3]} splits={splits}: dot-vs-mfma {err:.3e}  mfma-vs-oracle {ref_err:.3e}", flush=True)
print("worst", worst)
assert worst < 3e-2
if len(sys.argv) > 1 and sys.argv[1] == "time":
    for mode in (1, 2):
        L.mv_paged_attn_tune(2, mode)
        print("dot form" if mode == 1 else "matrix-core form", flush=True)
        run(1, 32, 8, 128, 4096)
        run(8, 32, 8, 128, 4096, layers=16)
        run(32, 32, 8, 128, 4096, layers=8)
        run(8, 8, 1, 128, 8192, layers=16)
        run(16, 4, 1, 128, 2048, layers=16)
L.mv_paged_attn_tune(2, 0)
