import os, sys, subprocess
sys.path.insert(0, "/root/repo")
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch, micro_vllm_amd
    from micro_vllm_amd.core.prefill_attention import paged_attn_prefill_rows
    S, B, hq, hkv, d = int(sys.argv[3]), 1, 32, int(sys.argv[4]), 128
    g = torch.Generator().manual_seed(5)
    pages = (S + 255) // 256
    kc = torch.randn(pages + 3, 256, hkv, d, generator=g).to(torch.bfloat16).cuda()
    vc = torch.randn(pages + 3, 256, hkv, d, generator=g).to(torch.bfloat16).cuda()
    qkv = torch.randn(B * S, (hq + 2 * hkv) * d, generator=g).to(torch.bfloat16).cuda()
    out = torch.zeros(B * S, hq * d, dtype=torch.bfloat16, device="cuda")
    bt = torch.tensor([[4, 1, 5, 0][:pages] + [-1] * (4 - pages)], dtype=torch.int32).cuda() if pages <= 4 else torch.arange(pages, dtype=torch.int32).reshape(1, -1).cuda()
    sl = torch.zeros(B, dtype=torch.int32, device="cuda")
    paged_attn_prefill_rows(qkv[:, :hq * d], B, S, hq, d, kc, vc, sl, bt, out)
    torch.cuda.synchronize()
    torch.save(out.cpu(), sys.argv[2])
else:
    import torch
    S, hkv = sys.argv[1], sys.argv[2]
    for pipe in ("0", "1"):
        subprocess.run([sys.executable, __file__, "child", f"/tmp/pf_out{pipe}.pt", S, hkv], env=dict(os.environ, MVLLM_PREFILL_ATTN_PIPE=pipe, MVLLM_PREFILL_ATTN_WAVES="8"), check=True)
    a, b = torch.load("/tmp/pf_out0.pt").float(), torch.load("/tmp/pf_out1.pt").float()
    diff = (a - b).abs().reshape(int(S), 32, 128)
    print("max diff", diff.max().item())
    bad = (diff.amax(dim=-1) > 0.02).nonzero()
    print("bad (pos, head) count", len(bad), "first", bad[:10].tolist(), "last", bad[-5:].tolist())
    if len(bad):
        print("bad positions:", sorted(set(bad[:, 0].tolist()))[:40])
        print("bad heads:", sorted(set(bad[:, 1].tolist())))
