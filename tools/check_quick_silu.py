"""How often the fused SwiGLU epilogue of the prefill GEMM differs from the same product formed in
fp64 from the unfused projection's bf16 outputs (same accumulators, same first rounding):
python tools/check_quick_silu.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from micro_vllm_amd.kernel.linear import StreamedWeight, gemm_prefill
torch.manual_seed(0)
m, inter, k = 2048, 14336, 4096
up = torch.randn(inter, k, device="cuda").mul_(0.02).to(torch.bfloat16)
gate = torch.randn(inter, k, device="cuda").mul_(0.02).to(torch.bfloat16)
x = torch.randn(m, k, device="cuda").to(torch.bfloat16)
fused = gemm_prefill(x, StreamedWeight.from_up_gate(up, gate), torch.empty(m, inter, dtype=torch.bfloat16, device="cuda"), swiglu=True)
u = gemm_prefill(x, StreamedWeight.from_linear(up), torch.empty(m, inter, dtype=torch.bfloat16, device="cuda"))
g = gemm_prefill(x, StreamedWeight.from_linear(gate), torch.empty(m, inter, dtype=torch.bfloat16, device="cuda"))
g64, u64 = g.double(), u.double()
exact = (g64 * torch.sigmoid(g64) * u64).to(torch.bfloat16)
g32, u32 = g.float(), u.float()
f32 = ((g32 * (1.0 / (1.0 + torch.exp(-g32)))) * u32).to(torch.bfloat16)
n = fused.numel()
d = (fused.view(torch.int16).int() - exact.view(torch.int16).int()).abs()
d32 = (f32.view(torch.int16).int() - exact.view(torch.int16).int()).abs()
print(f"{n} outputs; fused vs fp64: {int((d != 0).sum())} differ ({float((d != 0).float().mean()):.2e}), max {int(d.max())} ulp; "
      f"fp32 library form vs fp64: {int((d32 != 0).sum())} differ ({float((d32 != 0).float().mean()):.2e}), max {int(d32.max())} ulp")
