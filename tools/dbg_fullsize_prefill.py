"""Layer 0 of the full-size prefill test (600 tokens, Llama-3-8B widths) with the plain and the
pipelined attention kernel, each in a process of its own; where the two outputs differ."""
import dataclasses, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch, micro_vllm_amd
    from micro_vllm_amd.core.cache_manager import KVCacheManager
    from micro_vllm_amd.core.layer.model_prefill import ModelPrefillRunner
    from micro_vllm_amd.models import SHAPES, build_synthetic_model
    cfg = dataclasses.replace(SHAPES["llama3-8b"], num_hidden_layers=1)
    model = build_synthetic_model(cfg, "cpu", seed=1234, norm_jitter=0.1).to("cuda")
    prefill = ModelPrefillRunner(model, 1, cfg.num_attention_heads, cfg.head_dim, cfg.num_key_value_heads,
                                 cfg.hidden_size, cfg.intermediate_size, "cuda", max_batch_size=1)
    mgr = KVCacheManager(6, 256, 1, cfg.num_key_value_heads, cfg.head_dim, dtype=torch.bfloat16, device="cuda",
                         max_tokens=1024, max_batch_size=1)
    mgr._free.clear(); mgr._free.extend([4, 1, 5, 0, 2, 3])
    S = 600
    assert mgr.alloc(0, S)[0]
    mgr.cache_batch_data([0], [S])
    h = (torch.randn(S, cfg.hidden_size, generator=torch.Generator().manual_seed(3)) * 0.5).to(torch.bfloat16).cuda()
    lens = mgr._cache_seqlens_buffer[:1].zero_()
    import micro_vllm_amd.core.layer.model_prefill as mp
    keep = {}
    orig = mp.paged_attn_prefill_rows
    def spy(q_rows, B, S_, hq, d, kc, vc, lens_, bt_, *a, **k):
        r = orig(q_rows, B, S_, hq, d, kc, vc, lens_, bt_, *a, **k); keep["attn"] = r.clone()
        # fp32 reference on the same rotated q and the pages as written
        hkv = kc.shape[2]
        pages = bt_[0, :(S_ + 255) // 256].long()
        kk = kc[pages].reshape(-1, hkv, d)[:S_].float(); vv = vc[pages].reshape(-1, hkv, d)[:S_].float()
        q = q_rows.reshape(S_, hq, d).float()
        kk = kk.repeat_interleave(hq // hkv, dim=1); vv = vv.repeat_interleave(hq // hkv, dim=1)
        sc = torch.einsum("qhd,khd->hqk", q, kk) * d ** -0.5
        sc = sc + torch.full((S_, S_), float("-inf"), device=sc.device).triu(1)
        keep["ref"] = torch.einsum("hqk,khd->qhd", sc.softmax(-1), vv).reshape(S_, hq * d)
        return r
    mp.paged_attn_prefill_rows = spy
    out = prefill._prefill_layer(h, 0, 1, S, mgr, lens, mgr._block_table_buffer[:1])
    torch.cuda.synchronize()
    torch.save({"out": out.cpu(), "attn": keep["attn"].cpu(), "ref": keep["ref"].cpu()}, sys.argv[2])
else:
    import torch
    for pipe in ("0", "1"):
        subprocess.run([sys.executable, __file__, "child", f"/tmp/fs_out{pipe}.pt"],
                       env=dict(os.environ, MVLLM_PREFILL_ATTN_PIPE=pipe, MVLLM_PREFILL_ATTN_WAVES="8"), check=True)
    a, b = torch.load("/tmp/fs_out0.pt"), torch.load("/tmp/fs_out1.pt")
    for key in ("attn", "out"):
        d = (a[key].float() - b[key].float()).abs()
        print(key, tuple(d.shape), "max diff", d.max().item(), "rows with diff > 0.02:", (d.amax(dim=-1) > 0.02).nonzero().flatten().tolist()[:30])
    for name, t in (("plain", a), ("pipelined", b)):
        e = (t["attn"].float() - t["ref"]).abs()
        print(f"{name}: attention against fp32 softmax: mean abs err {e.mean().item():.3e}, max {e.max().item():.3e}, "
              f"mean |ref| {t['ref'].abs().mean().item():.3f}")
    d = (a["attn"].float() - b["attn"].float()).abs().reshape(600, 32, 128)
    bad = (d.amax(dim=-1) > 0.02).nonzero()
    print("bad (pos, head):", bad[:20].tolist(), "count", len(bad))
    if len(bad):
        p, hh = bad[0].tolist()
        print("plain", a["attn"].reshape(600, 32, 128)[p, hh, :8].tolist())
        print("piped", b["attn"].reshape(600, 32, 128)[p, hh, :8].tolist())
