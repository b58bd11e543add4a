/*
 * mvllm.h - C ABI of the MI355X (gfx950) paged-KV decode path.
 *
 * Every entry point is a plain C function: raw device pointers, sizes and
 * strides in ELEMENTS, an explicit hipStream_t (passed as void*), and an int
 * return code (0 = ok, negative = mv_status below).  No STEP call (anything that
 * takes a stream) allocates, frees or synchronises: the caller owns every buffer,
 * and all launches are legal under HIP stream capture.  The set-up and tear-down
 * calls of the communicators (mv_allreduce_init / _connect / _destroy,
 * mv_allgather_init) do allocate, map and free device memory and wait for the
 * device.  They say so where they are declared.  The entry points keep no state on the device
 * between calls except in the buffers the caller passes, so calls on different
 * streams (or threads) are independent as long as they are given different
 * workspaces; the one exception is mv_paged_attn_decode_noalloc, which says so.  All tensors are bf16 unless a parameter
 * says otherwise; accumulation is fp32.
 *
 * Each declaration names the reference call it stands in for
 * (paths relative to the micro-vllm checkout).
 */
#ifndef MVLLM_H
#define MVLLM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mv_stream_t; /* hipStream_t */

enum mv_status {
    MV_OK = 0,
    MV_ERR_INVALID_ARG = -1, /* null pointer, non-positive size, misalignment */
    MV_ERR_UNSUPPORTED = -2, /* shape outside what the kernels are built for  */
    MV_ERR_LAUNCH = -3,      /* hipGetLastError() after the launch was not ok */
    MV_ERR_WORKSPACE = -4,   /* caller workspace too small                    */
    MV_ERR_GRAPH = -5        /* stream-capture / graph API failure            */
};

/* Library identity and error text. */
int mv_version(void);
const char* mv_status_string(int status);
/* Last HIP error string seen by this thread's most recent failing call. */
const char* mv_last_hip_error(void);

/* ------------------------------------------------------------------ norms */

/* y[r,:] = bf16( (x[r,:] * rsqrt(mean(x[r,:]^2) + eps)) * w ), fp32 math.
 * Replaces kernel/rmsnorm_add.py:170 rmsnorm_ (decode, writes a static
 * buffer) and kernel/rmsnorm_add.py:37 rmsnorm (prefill). */
int mv_rmsnorm(const void* x, const void* w, void* y, int rows, int n,
               int64_t stride_x, int64_t stride_y, float eps, mv_stream_t stream);

/* s = x + r (fp32); res_out = bf16(s); var from the UNROUNDED s;
 * y = bf16( (float(res_out) * rrms) * w ).
 * Replaces kernel/rmsnorm_residual.py:127 rmsnorm_residual_gemm and
 * kernel/rmsnorm_add.py:89 rmsnorm_residual_fused. */
int mv_rmsnorm_residual(const void* x, const void* r, const void* w, void* y, void* res_out,
                        int rows, int n, int64_t stride_x, int64_t stride_r, int64_t stride_y,
                        int64_t stride_res, float eps, mv_stream_t stream);

/* out[r,c] = bf16( gate*sigmoid(gate)*up ), up = gate_up[r,c], gate = gate_up[r,c+d].
 * Replaces kernel/swiglu.py:100 swiglu_gemm and :49 swiglu_fused. */
int mv_swiglu(const void* gate_up, void* out, int64_t rows, int d, mv_stream_t stream);

/* out = bf16(float(a) + float(b)), n elements (in place allowed: out == a).
 * Replaces the `.add_(self._residual[:bs])` of core/layer/model_graph.py:219
 * and the bias add of :147 (broadcast form: b has `n_b` elements, n % n_b == 0). */
int mv_add(const void* a, const void* b, void* out, int64_t n, int64_t n_b, mv_stream_t stream);

/* out[r,:] = table[ids[r],:].  Replaces model.transformer.wte(input_ids),
 * core/layer/model_graph.py:196.  ids are int64 (torch.long). */
int mv_embedding(const void* table, const int64_t* ids, void* out, int rows, int n,
                 int64_t vocab, mv_stream_t stream);

/* ------------------------------------------------- weight-streaming GEMM */

/* One-time relayout of a [N,K] row-major bf16 weight (torch Linear.weight,
 * y = x @ W^T) into wave tiles: tile (nt,kt) holds rows 16nt..16nt+15 and
 * columns 32kt..32kt+31 as 1 KiB, lane l owning W[16nt + (l&15)][32kt + 8(l>>4) + 0..7].
 * N % 16 == 0, K % 32 == 0.  Replaces the `.t().contiguous()` relayout of
 * core/layer/model_graph.py:87-118 prepare_weights. */
int mv_shuffle_weight(const void* w_nk, void* w_tiles, int n, int k, mv_stream_t stream);

/* Rows of `a` and `b` interleaved in groups of 8: out rows [16j,16j+8) = a rows
 * [8j,8j+8), out rows [16j+8,16j+16) = b rows [8j,8j+8).  Used to put `up`
 * and `gate` rows of one output column into the same wave tile. */
int mv_interleave_rows8(const void* a, const void* b, void* out, int rows_each, int k,
                        mv_stream_t stream);

/* Bytes of workspace mv_wgemm needs for the given problem (0 if none). */
size_t mv_wgemm_workspace_bytes(int m, int n, int k);

/* y[m, :] = bf16( bf16( bf16(x[m,:] @ W^T) + bias ) + residual[m,:] ), the two
 * optional terms each adding one bf16 rounding exactly where the reference's
 * `torch.matmul(out=)`, `.add_(bias)` and `.add_(residual)` round.
 * x: [M,K] (row stride ldx), W: tiles from mv_shuffle_weight, y: [M,N].
 * 1 <= M <= 64.  flags: MV_GEMM_* below.
 * Replaces the decode projections torch.matmul(..., out=) at
 * core/layer/model_graph.py:145,171,183,185 and model.lm_head (:223). */
enum {
    MV_GEMM_DEFAULT = 0,
    MV_GEMM_SWIGLU = 1, /* W rows are up/gate interleaved by 8 (mv_interleave_rows8);
                           y is [M, N/2] = swiglu(bf16(x @ W^T)), i.e. the matmul
                           + swiglu_gemm pair of model_graph.py:183-184 in one pass */
    MV_GEMM_OUT_F32 = 2, /* y is fp32 [M,N]: the un-rounded accumulators (tests) */
    MV_GEMM_SLABS = 4,   /* y is not written (may be NULL): the product is left in `workspace` as
                            mv_wgemm_slab_count(m, n, k) fp32 slabs [slab][M][N] whose sum, added in
                            slab order, is x @ W^T - for a caller whose next kernel adds them
                            (mv_allreduce_addnorm).  Workspace: mv_wgemm_postnorm_workspace_bytes. */
    MV_GEMM_PLAN_SLABS = 8 /* an ordinary call (y written, epilogue applied) that cuts k exactly as the
                            MV_GEMM_SLABS call of the same shape does, so that its sums are formed in
                            the same order: what a step built from separate kernels uses where the
                            fused step hands slabs to the next kernel (tests compare the two bit
                            for bit) */
};
int mv_wgemm(const void* x, int64_t ldx, const void* w_tiles, const void* bias,
             const void* residual, int64_t ld_res, void* y, int64_t ldy, int m, int n, int k,
             void* workspace, size_t workspace_bytes, int flags, mv_stream_t stream);

/* Number of slabs an MV_GEMM_SLABS call leaves for this problem (>= 1). */
int mv_wgemm_slab_count(int m, int n, int k);

/* mv_wgemm with the input transform of the preceding norm kernel folded in (1 <= M <= 4):
 *   add_r == NULL:  x' = rmsnorm(x) * ln_w                       (mv_rmsnorm, then mv_wgemm)
 *   add_r != NULL:  s = x + add_r; sum_out = bf16(s); x' = rmsnorm(s) * ln_w
 *                                                    (mv_rmsnorm_residual, then mv_wgemm)
 * with exactly the rounding of those two-kernel sequences (results are bit-identical to them).
 * norm_flags & MV_NORM_VAR_FROM_ROUNDED takes the variance from bf16(s) instead of the fp32 s:
 * that is `h = a.add_(r)` followed by rmsnorm_(h) (core/layer/model_graph.py:219 then :143).
 * Replaces rmsnorm_ + matmul (model_graph.py:143-145), rmsnorm_residual_gemm + matmul
 * (:213, :183) and ln_f + lm_head (:222-223) as single launches. */
enum { MV_NORM_VAR_FROM_ROUNDED = 1 };
int mv_wgemm_norm(const void* x, int64_t ldx, const void* add_r, int64_t ld_add_r,
                  const void* ln_w, float eps, void* sum_out, int64_t ld_sum_out, int norm_flags,
                  const void* w_tiles, const void* bias, const void* residual, int64_t ld_res,
                  void* y, int64_t ldy, int m, int n, int k, void* workspace,
                  size_t workspace_bytes, int flags, mv_stream_t stream);

/* mv_wgemm with the norm kernel that FOLLOWS the projection folded into the output (any M <= 64,
 * n <= 8192; the product goes through fp32 slabs in `workspace`, one launch sums them, applies
 * the epilogue and normalises the row):
 *   t = bf16(x @ W^T); t = bf16(t + bias); s = t + residual; y = bf16(s);
 *   normed = bf16((float(y) * rrms) * ln_w), rrms = 1/sqrt(mean(s^2) + eps)
 * i.e. matmul followed by rmsnorm_residual_gemm (core/layer/model_graph.py:171 then :213: the
 * variance is taken from the unrounded s), or, with MV_NORM_VAR_FROM_ROUNDED, matmul + residual
 * add followed by the next block's rmsnorm_ (:185, :219, :143: variance from y as stored).
 * Bit-identical to mv_wgemm followed by mv_rmsnorm_residual / mv_rmsnorm. */
size_t mv_wgemm_postnorm_workspace_bytes(int m, int n, int k);
int mv_wgemm_postnorm(const void* x, int64_t ldx, const void* w_tiles, const void* bias,
                      const void* residual, int64_t ld_res, void* y, int64_t ldy, const void* ln_w,
                      float eps, void* normed, int64_t ld_normed, int norm_flags, int m, int n, int k,
                      void* workspace, size_t workspace_bytes, mv_stream_t stream);

/* The QKV projection (with its RMSNorm prologue) and the paged attention of a decode step of 1, 2 or
 * 4 rows in ONE launch: what mv_wgemm_norm (x, ln_w, eps, w_tiles, bias -> qkv [m, n], n = (hq + 2 hkv) d)
 * followed by mv_paged_attn_decode (q | k | v = the columns of qkv) computes, bit for bit - the same
 * strips in the same wave shape, the same row assignment, softmax arithmetic and split merge - with
 * the attention's page ids and key/value rows requested at the very start of the launch, before q
 * exists.  core/layer/model_graph.py:143-147 and :161-168 of one block.  No counterpart in the
 * reference (which cannot start flash-attn's loads before cuBLAS has finished).
 * One 512-thread workgroup per (row, kv head, split), all resident at once: MV_ERR_UNSUPPORTED (issue
 * the two launches instead) unless CUs / 2 < m * hkv * num_splits <= CUs, num_splits >= 2, the head
 * shape is one the dot-product attention form serves with (d, hq / hkv) in {128} x {1, 2, 4} or
 * {64} x {1, 4}, and the stand-alone projection would run this shape as whole strips on 8 waves.
 * qkv is written (write-through) and may be read by later launches.  attn_workspace: as for
 * mv_paged_attn_decode.  sync_workspace: mv_qkv_attn_workspace_bytes() bytes, 16-byte aligned,
 * zero-filled once by the caller, for launches that are stream-ordered with each other; the launch
 * needs the whole GPU to itself (a workgroup waits for the others of its kv head).  A wait of more
 * than 2 s gives up, the results are then wrong, and *status_word - any int the device can write,
 * e.g. pinned host memory, zero before the first launch - is set to 1 + the workgroup that gave up. */
size_t mv_qkv_attn_workspace_bytes(void);
int mv_qkv_attn_decode(const void* x, int64_t ldx, const void* ln_w, float eps, const void* w_tiles,
                       const void* bias, void* qkv, int64_t ld_qkv, int m, int n, int k, void* k_cache,
                       void* v_cache, const void* rot_cos, const void* rot_sin, int rot_max_pos,
                       const int32_t* cache_seqlens, const int32_t* block_table, int bt_stride, void* out,
                       void* attn_workspace, size_t attn_workspace_bytes, void* sync_workspace,
                       size_t sync_workspace_bytes, int* status_word, int hq, int hkv, int d, int page,
                       int num_pages, float scale, int num_splits, mv_stream_t stream);

/* Launch-shape knobs for experiments (no counterpart in the reference; process-wide, read once
 * per call, safe to set from another thread):
 * key 0 = workgroups per CU the grid is sized for (1..8; 0 = planner, the default);
 * key 1 = forced number of k-slices per strip (0 = pick automatically);
 * key 2 = waves per workgroup of the M <= 4 kernel (8 or 16; 0 = planner);
 * key 3 = strips a workgroup of the M > 4 kernel multiplies together (1, 2, 4, 7; 0 = planner);
 * key 4 = tiles of a wave of the M <= 4 kernel consecutive (1) or interleaved (0, the default). */
int mv_wgemm_tune(int key, int value);

/* y[M,N] = x[M,K] . W^T for many rows (prefill) on the matrix cores, reading the same tile image
 * as mv_wgemm (mv_shuffle_weight / mv_interleave_rows8); flags, bias, residual, rounding and the
 * SWIGLU output width as for mv_wgemm.  K must be a multiple of 64, N of 16; any M >= 0.
 * Replaces torch.matmul at core/layer/model_prefill.py:42,54, the compiled _mlp (:58,
 * core/layer/model_graph.py:75-81) and lm_head (:62). */
int mv_gemm_prefill(const void* x, int64_t ldx, const void* w_tiles, const void* bias,
                    const void* residual, int64_t ld_res, void* y, int64_t ldy, int m, int n,
                    int k, int flags, mv_stream_t stream);

/* The same with a workspace for sums over parts of k.  When M x N alone gives fewer output tiles
 * than the chip has CUs (prompts of up to a few hundred tokens), k is cut into slices that run side
 * by side and a second launch adds them (fp32, in order) and applies the epilogue.  When there are
 * more 256-wide tiles than CUs and the last round of tiles would leave half the CUs or more idle,
 * the tiles of that round are cut along k between the workgroups of the one launch: the lower parts
 * leave their accumulators in the workspace (256 KiB per workgroup) and the workgroup with the last
 * part adds them, in part order, as it applies the epilogue.  The workspace starts with 4 KiB of
 * counters through which those parts count themselves in: zero-fill the workspace ONCE, when it is
 * allocated; every launch returns its counters to zero and writes its data behind them, so the
 * same workspace serves any later call of any shape that fits - one workspace per stream that
 * issues such calls.  (A part that has not arrived after about 20 s - a workgroup of the launch
 * lost - ends the launch with a trap, i.e. the process, rather than returning sums that lack a
 * part of k.)  Without the workspace (or with too small a one) the call behaves like
 * mv_gemm_prefill. */
size_t mv_gemm_prefill_workspace_bytes(int m, int n, int k);
int mv_gemm_prefill_ws(const void* x, int64_t ldx, const void* w_tiles, const void* bias,
                       const void* residual, int64_t ld_res, void* y, int64_t ldy, int m, int n,
                       int k, int flags, void* workspace, size_t workspace_bytes, mv_stream_t stream);
/* Which form a call of that shape takes on the current device (for tests and tools; no counterpart
 * in the reference): 0 = 128 x 256 tiles, two workgroups per CU, weights straight into registers;
 * 1 = 256 x 256 tiles, one workgroup per CU, both operands through LDS (taken when its tiles fill
 * at least half the CUs; MVLLM_PREFILL_WIDE=0/1 forces the choice for a process); 2 and 3 = the same
 * with tiles of 192 and of 224 columns, where those fill the CUs in fewer or fuller rounds
 * (MVLLM_PREFILL_WIDE_STRIPS=12/14/16 forces the width, in 16-column strips). */
int mv_gemm_prefill_form(int m, int n, int k);
/* Overrides for tests and A/B runs inside one process (no counterpart in the reference): cut = 0 / 1
 * switches the cutting of left-over tiles off / on, strips = 12 / 14 / 16 forces the tile width, for
 * every call that follows; -1 returns a choice to the environment (MVLLM_PREFILL_CUT,
 * MVLLM_PREFILL_WIDE_STRIPS) and the planner.  Process-wide; not for use beside running calls. */
void mv_debug_prefill_options(int cut, int strips);

/* ---------------------------------------------------- paged attention */

/* Bytes of workspace for the split-KV merge (0 for one split): a 1 MiB header of arrival
 * tickets, then the splits' log-sum-exps and fp32 partial outputs.  The caller zero-fills the
 * workspace ONCE, when it allocates it; every launch returns the tickets to zero, so the same
 * workspace serves any later call, of any shape that fits. */
size_t mv_paged_attn_decode_workspace_bytes(int bs, int hq, int d, int num_splits);

/* How many KV splits the library would pick for this shape (ctx_hint = typical
 * context length; <= 0 means unknown). */
int mv_paged_attn_decode_pick_splits(int bs, int hkv, int d, int ctx_hint);

/* Which form of the kernel mv_paged_attn_decode runs for this shape: 0 = dot-product form (scores,
 * softmax, P and P.V in fp32), 1 = matrix-core form (P rounded to bf16 before P.V, as
 * flash_attn_with_kvcache does: model_graph.py:161-168 is the call it stands in for); -1 = bad shape.
 * The two agree within that rounding: |diff| <= 2^-9 * sum_t p_t |v_t| + one bf16 unit. */
int mv_paged_attn_decode_form(int hq, int hkv, int d, int page);

/* Experiment knobs (no counterpart in the reference): key 0 = merge the KV splits inside the
 * attention kernel (1, default) or in a second launch (0); key 1 = workgroups per CU the
 * automatic split count aims at (1..8; 0 = the built-in rule, default); key 2 = the matrix-core
 * form of the kernel (head size 128, 4 or 8 query heads per kv head): 0 = built-in rule,
 * 1 = never, 2 = whenever it applies. */
int mv_paged_attn_tune(int key, int value);

/* Single-token paged attention with in-kernel NeoX RoPE and KV append.
 * For each row b with p = cache_seqlens[b]:
 *   q,k <- bf16(RoPE(q,k) at position p) using cos/sin[p, :D/2] (skipped when cos == NULL);
 *   k_cache/v_cache[block_table[b, p / page], p % page, h, :] <- k, v   (in place);
 *   out[b,hq,:] = bf16( softmax(q . K[0..p]^T * scale) V[0..p] ), q head hq reading kv head hq / (Hq/Hkv),
 *   scores, softmax, P and the P.V accumulation in fp32.
 * q: [bs,Hq,D] (row stride q_stride_b, heads contiguous), k_new/v_new: [bs,Hkv,D]
 * (kv_stride_b), caches: [num_pages, page, Hkv, D] contiguous, cache_seqlens: int32 [bs],
 * block_table: int32 [bs, bt_stride] (-1 = unallocated, never dereferenced for
 * indices <= p / page unless the caller broke the contract, in which case the key is
 * skipped), out: [bs,Hq,D] contiguous.  D in {64,128}; Hq/Hkv from 1 to 8.
 * workspace: see mv_paged_attn_decode_workspace_bytes (zero-filled once by the caller; calls
 * that may overlap on one GPU need a workspace each).
 * Replaces the flash_attn_with_kvcache call at core/layer/model_graph.py:161-168
 * (and core/paged_attention.py:281). */
int mv_paged_attn_decode(const void* q, int64_t q_stride_b, const void* k_new, const void* v_new,
                         int64_t kv_stride_b, void* k_cache, void* v_cache, const void* rot_cos,
                         const void* rot_sin, int rot_max_pos, const int32_t* cache_seqlens,
                         const int32_t* block_table, int bt_stride, void* out, void* workspace,
                         size_t workspace_bytes, int bs, int hq, int hkv, int d, int page,
                         int num_pages, float scale, int num_splits, mv_stream_t stream);

/* mv_paged_attn_decode with q | k_new | v_new taken as the QKV projection left them when it was cut
 * along k: qkv_slabs = fp32 [ksplit][bs][(Hq + 2 Hkv) D] (an MV_GEMM_SLABS call of mv_wgemm), whose
 * sum in slab order, rounded to bf16, plus qkv_bias ([(Hq + 2 Hkv) D] bf16 or NULL) rounded again,
 * is the row the projection's own slab reduction would have stored - bit for bit - so a decode
 * step for more than four rows saves that launch.  Everything else as mv_paged_attn_decode.  No
 * counterpart in the reference (its QKV projection is one cuBLAS call, core/layer/model_graph.py:
 * 141-148). */
int mv_paged_attn_decode_slabs(const void* qkv_slabs, int ksplit, const void* qkv_bias, void* k_cache,
                               void* v_cache, const void* rot_cos, const void* rot_sin, int rot_max_pos,
                               const int32_t* cache_seqlens, const int32_t* block_table, int bt_stride,
                               void* out, void* workspace, size_t workspace_bytes, int bs, int hq, int hkv,
                               int d, int page, int num_pages, float scale, int num_splits,
                               mv_stream_t stream);

/* mv_paged_attn_decode for callers that cannot allocate device memory (code under HIP stream
 * capture, tight serving loops, harnesses that forbid hipMalloc): the same kernels and the same
 * results, bit for bit, with the split-KV workspace in an 8 MiB array that is part of the
 * library's device image instead of a caller buffer.  It has no counterpart in the reference.
 * The call allocates and frees nothing.  Limit: mv_paged_attn_decode_workspace_bytes(bs, hq, d,
 * num_splits) <= 8 MiB (e.g. 32 rows x 32 heads x 4 splits at d = 128), else MV_ERR_WORKSPACE
 * with the sizes in mv_last_hip_error().  Every GPU holds one copy of the array for the life of
 * the process and every call on a GPU uses it: calls that overlap on one GPU (two streams, two
 * threads) race and must not be made.  The first call on a device looks the array's address up
 * (make it outside stream capture). */
int mv_paged_attn_decode_noalloc(const void* q, int64_t q_stride_b, const void* k_new,
                                 const void* v_new, int64_t kv_stride_b, void* k_cache, void* v_cache,
                                 const void* rot_cos, const void* rot_sin, int rot_max_pos,
                                 const int32_t* cache_seqlens, const int32_t* block_table,
                                 int bt_stride, void* out, int bs, int hq, int hkv, int d, int page,
                                 int num_pages, float scale, int num_splits, mv_stream_t stream);

/* ---- prefill (many tokens per sequence) ---- */

/* NeoX rotation of `heads` heads of width d per row, y may alias x; row t uses cos/sin[t % seq].
 * Evaluated like the reference's bf16 tensor expression (every product and the sum/difference
 * round to bf16): core/layer/rope.py:10-39 (half-width table branch :28-33). */
int mv_rope_prefill(const void* x, int64_t x_stride, void* y, int64_t y_stride, const void* rot_cos,
                    const void* rot_sin, int rot_max_pos, int64_t tokens, int seq, int heads, int d,
                    mv_stream_t stream);

/* One launch for what the prefill runner does between the QKV projection and attention
 * (core/layer/model_prefill.py:45-52): rows of qkv are [q heads | k heads | v heads]; token i of
 * sequence b sits at position p = cache_seqlens[b] + i; q and k are rotated in place with
 * cos/sin[p] (mv_rope_prefill arithmetic; skipped when cos == NULL) and the rotated k and v are
 * written to k_cache/v_cache[block_table[b, p / page], p % page, h, :]. */
int mv_rope_store_prefill(void* qkv, int64_t row_stride, const void* rot_cos, const void* rot_sin,
                          int rot_max_pos, const int32_t* cache_seqlens, const int32_t* block_table,
                          int bt_stride, void* k_cache, void* v_cache, int bs, int seq, int hq,
                          int hkv, int d, int page, int num_pages, mv_stream_t stream);

/* Causal attention of `seq` new tokens per sequence over the paged cache, which must already
 * hold their keys and values (mv_rope_store_prefill / mv_store_kv): query i of sequence b sees
 * keys 0 .. cache_seqlens[b] + i (bottom-right aligned mask), q head h reads kv head h / (Hq/Hkv).
 * q rows: [bs*seq][q_stride] with heads contiguous; out rows likewise.  fp32 scores and softmax,
 * P rounded to bf16 as the matrix-core operand, fp32 accumulation, bf16 output.
 * D in {64,128}; Hq/Hkv from 1 to 8 (3, 5, 6 and 7 take the row stack of the next power of two: the rows of
 * the heads that do not exist are computed and dropped); page a power of two >= 64.
 * With the store above this replaces flash_attn_with_kvcache(q, k_cache, v_cache, k=, v=,
 * cache_seqlens=, block_table=, causal=True) at core/layer/model_prefill.py:49-52. */
int mv_paged_attn_prefill(const void* q, int64_t q_stride, const void* k_cache, const void* v_cache,
                          const int32_t* cache_seqlens, const int32_t* block_table, int bt_stride,
                          void* out, int64_t out_stride, int bs, int seq, int hq, int hkv, int d,
                          int page, int num_pages, float scale, mv_stream_t stream);
/* Waves per workgroup the call above takes for this shape on the current device (4 = 128 stacked query
 * rows per workgroup, 8 = 256; for tests and tools, no counterpart in the reference;
 * MVLLM_PREFILL_ATTN_WAVES=4/8 forces one for a process). */
int mv_paged_attn_prefill_form(int bs, int seq, int hq, int hkv);

/* key[t,h,:] -> k_cache[slot/page, slot%page, h, :] (same for value); slot -1 = skip.
 * key/value: [T,H,D] with token stride kv_stride_t; slot_mapping int32 [T].
 * Replaces core/cache_manager.py:143 store_kvcache (Triton store_kvcache_kernel :75). */
int mv_store_kv(const void* key, const void* value, int64_t kv_stride_t, void* k_cache,
                void* v_cache, const int32_t* slot_mapping, int tokens, int heads, int d,
                int page, int num_pages, mv_stream_t stream);

/* ------------------------------------------------------------- sampling */

/* ids[r] = argmax_c logits[r, c] over bf16 logits [rows, n] (row stride ld), lowest index on
 * ties; ids are int64.  The greedy (temperature -> 0) case of core/layer/sampler.py:21-43,
 * done on the device so the step needs no [bs,V] sort and no host round trip.
 * workspace (optional, mv_argmax_workspace_bytes() bytes, 16-byte aligned, zero-filled once by
 * the caller): where the parts of a long row meet when few rows are spread over many
 * workgroups; without it every row gets one workgroup (same result, slower). */
size_t mv_argmax_workspace_bytes(void);
int mv_argmax(const void* logits, int64_t ld, int rows, int n, int64_t* ids, void* workspace,
              size_t workspace_bytes, mv_stream_t stream);

/* Stochastic pick, one launch per batch: for each row, scale by 1/temperature[r], keep the
 * min(top_k, n, 1024) largest logits, softmax over them, keep the sorted prefix whose inclusive
 * cumulative probability is < top_p[r] (always at least the first), renormalise, and return the
 * entry at which the cumulative probability first exceeds uniforms[r] (a number in [0,1) the caller
 * draws).  Equal logits are ordered by ascending index, so the result is a deterministic function of
 * its inputs; temperature <= 0 returns the argmax.  n must be a multiple of 8.
 * The recipe of core/layer/sampler.py:21-43 (_sample_impl: topk, softmax, sort, cumsum < top_p,
 * multinomial), with the draw expressed through an explicit uniform number. */
int mv_sample_topk_topp(const void* logits, int64_t ld, int rows, int n, const float* temperatures,
                        const float* top_ps, int top_k, const float* uniforms, int64_t* ids,
                        mv_stream_t stream);

/* The same, for use inside a captured decode step: rows with temperature <= 0 take
 * greedy_ids[r] (the caller's mv_argmax result) without looking at the logits, so a step that
 * always contains this launch costs greedy batches nothing.  greedy_ids == NULL: as above. */
int mv_sample_or_greedy(const void* logits, int64_t ld, int rows, int n, const float* temperatures,
                        const float* top_ps, int top_k, const float* uniforms,
                        const int64_t* greedy_ids, int64_t* ids, mv_stream_t stream);

/* ------------------------------------------- tensor-parallel all-reduce */

/* One-shot all-reduce for the small [rows, n] bf16 messages of a tensor-parallel decode step, among
 * the (at most 8) GPUs of one node, one process per GPU.  Every rank pushes its partial into every
 * rank's exchange buffer over the direct xGMI links and sums what arrived in its own buffer in rank
 * order (fp32, rounded to bf16 once): all ranks get the same bits.  Stands in for
 * core/parallel_config.py:69-74 all_reduce at core/layer/model_graph.py:210,219; messages beyond the
 * communicator's capacity (prefill) stay with RCCL.
 *
 * Set-up, once per process (these three allocate / map device memory; the step calls do not):
 *   mv_allreduce_init        exchange buffer for messages of up to max_rows x max_n (max_rows <= 64,
 *                            max_n <= 8192, a multiple of 8) on the current device;
 *   mv_allreduce_get_handle  mv_allreduce_handle_bytes() bytes to pass to the other ranks by any host
 *                            channel (e.g. torch.distributed.all_gather_object);
 *   mv_allreduce_connect     `handles` = all ranks' handles in rank order; maps the peers' buffers.
 * The step calls take the stream, never synchronise and are legal under stream capture: flags carry an
 * epoch kept in device memory that only grows, so a captured step can be replayed.  All ranks must
 * issue the same sequence of calls with the same `rows`.  A rank that waits longer than
 * MVLLM_ALLREDUCE_TIMEOUT_MS (default 10 s) for a peer gives up, finishes with a wrong sum and sets
 * the communicator's status word (1 + the rank that did not arrive; 0 = no timeout so far).  The word
 * is host memory mapped into the device: mv_allreduce_status is a plain load that neither waits nor
 * touches the device, and tells about the step calls the caller has already waited for (so a serving
 * loop reads it right after the token read-back of a step, before it uses the tokens).  After a
 * timeout the ranks' epochs may be out of step: destroy the communicator, do not go on with it. */
size_t mv_allreduce_handle_bytes(void);
int mv_allreduce_init(int rank, int world, int max_rows, int max_n, void** comm_out);
int mv_allreduce_get_handle(void* comm, void* handle_out);
int mv_allreduce_connect(void* comm, const void* handles);
int mv_allreduce_status(void* comm, int* status_out);
/* How long later step calls wait for a peer (overrides MVLLM_ALLREDUCE_TIMEOUT_MS for this communicator). */
int mv_allreduce_set_timeout_ms(void* comm, double ms);
int mv_allreduce_destroy(void* comm);

/* out = bf16(sum over ranks of in), [rows, n] bf16; out may alias in. */
int mv_allreduce_step(void* comm, const void* in, int64_t ld_in, void* out, int64_t ld_out, int rows,
                      int n, mv_stream_t stream);

/* All-gather of [rows, n_local] bf16 slices into [rows, world * n_local] rows on every rank, through
 * a communicator of its own (same handle / connect / status / set_timeout / destroy calls as above):
 *   mv_allgather_init   exchange buffer for up to max_rows rows of world x n_local columns
 *                       (n_local a multiple of 8);
 *   mv_allgather_rows   rank r's `in` lands at columns [r * n_local, (r + 1) * n_local) of `out` on every
 *                       rank; takes the stream, never synchronises, legal under stream capture.
 * No counterpart in the reference, which keeps the LM head whole on every rank (core/layer/
 * model_graph.py:222-223): with the head cut by vocabulary rows each rank multiplies 1/world of it and
 * the ranks exchange their logits instead (at 8 ranks of the 8B shape 1/8 of the largest single
 * projection of a decode step). */
int mv_allgather_init(int rank, int world, int max_rows, int n_local, void** comm_out);
int mv_allgather_rows(void* comm, const void* in, int64_t ld_in, void* out, int64_t ld_out, int rows,
                      int n_local, mv_stream_t stream);

/* The all-reduce together with what a decoder block does next (core/layer/model_graph.py:210-213 and
 * :219 + :143 of the following block), one launch:
 *   t = bf16(sum over ranks of partial);  s = t + residual;  y = bf16(s);
 *   normed = bf16((float(y) * rrms) * ln_w),  rrms = 1 / sqrt(mean(s^2) + eps)
 * (mean over the unrounded s, or over y as stored with MV_NORM_VAR_FROM_ROUNDED: the two cases of
 * mv_wgemm_postnorm).  This rank's partial is either `in` (bf16 rows) or, with in == NULL, the sum of
 * the `ksplit` fp32 slabs [ksplit][rows][n] an MV_GEMM_SLABS projection left, rounded to bf16 - what
 * that projection's bf16 output would have been.  residual, y and (ln_w, normed) are optional. */
int mv_allreduce_addnorm(void* comm, const void* in, int64_t ld_in, const float* slabs, int ksplit,
                         const void* residual, int64_t ld_res, void* y, int64_t ldy, const void* ln_w,
                         float eps, void* normed, int64_t ld_normed, int norm_flags, int rows, int n,
                         mv_stream_t stream);

/* ------------------------------------------------------------ hipGraph */

/* Capture everything launched on `stream` between begin and end into an
 * executable hipGraph; replay it with mv_graph_launch.  Replaces the
 * torch.cuda.CUDAGraph capture/replay of core/layer/model_graph.py:241-259. */
int mv_graph_begin_capture(mv_stream_t stream);
int mv_graph_end_capture(mv_stream_t stream, void** graph_exec_out);
int mv_graph_launch(void* graph_exec, mv_stream_t stream);
int mv_graph_destroy(void* graph_exec);

#ifdef __cplusplus
}
#endif
#endif /* MVLLM_H */
