"""CPU restatement of the reference's paged-KV decode path.  TEST INFRASTRUCTURE ONLY.

Nothing under this directory is part of the product: only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import it, and only as the checker.  Every function
cites the reference file:line it restates (paths relative to the micro-vllm checkout).

How the restatement is pinned (tests/test_oracle_golden.py):
  * rmsnorm / add+rmsnorm / swiglu: against outputs of the reference's own Triton kernels run
    under the Triton interpreter on CPU (tests/golden/*.npz, made by tests/golden/generate_golden.py);
  * rotary tables, RoPE.forward, KVCacheManager block-table traces, _mlp, scheduler
    decisions: against outputs of the reference's own Python run on CPU (same fixtures);
  * flash_attn_with_kvcache: PARITY UNPINNED.  The arithmetic lives in the third-party
    flash-attn package (requirements.txt:21, `flash-attn>=2.0.0`, unpinned, not installed and
    not vendored); the reference holds no test or golden vector at that boundary.  The
    restatement follows the package's documented contract and is itself the oracle;
  * GEMMs (cuBLAS via torch.matmul), ln_f / wte / lm_head (HuggingFace remote-code modules):
    restated as fp32-accumulate matmul / RMSNorm / row gather with one bf16 rounding each.
"""
