// Single-token paged attention (decode) with in-kernel NeoX RoPE and KV append.
//
// Bandwidth-bound dot-product kernel: each key/value row (one token, one kv head,
// D bf16 = 2D bytes contiguous) is read exactly once by D/8 adjacent lanes, 16 B per
// lane, so one wave instruction moves 64/(D/8) whole rows in full 128-byte lines.
// All G = Hq/Hkv query heads that share the kv head are scored against the row
// while it is in registers (GQA reuse: KV bytes are read once per kv head).
//
// Work split: workgroup = (sequence b, kv head, kv split); its W = 4 waves interleave
// over the split's rows.  Every 16-lane (D=128) / 8-lane (D=64) row group keeps
// its own online-softmax state (m, l, acc[G][8]); states meet once in LDS at the
// end.  Splits write normalised fp32 partials + log2-sum-exp with write-through stores
// and take a ticket on a per-(b, kv head) arrival counter; the workgroup that draws the
// last ticket merges the partials in split order (deterministic) and writes the bf16
// output, so no second launch is needed.  num_splits == 1 writes the output directly.
//
// Numerics: q and k are rotated in fp32 and rounded to bf16 (k is stored rotated);
// scores, softmax, P and the P.V accumulation are all fp32.  (A tensor-core kernel
// rounds P to bf16 to feed the matrix unit; a dot-product kernel has no reason to,
// and keeping P exact halves the expected distance to any such kernel's result.)
#include "common.h"

#include <atomic>
#include <stdio.h>

#ifdef MV_ATTN_STAMP
namespace mv {
// Phase stamps (build variant for tools/stamp_attn_decode.py): the constant 100 MHz counter at
// up to sixteen points of every workgroup, written by its first thread.
static __device__ unsigned long long g_astamp[16 * 4096];
}  // namespace mv
#define MV_ASTAMP(slot)                                                             \
    do {                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                          \
        if (threadIdx.x == 0 && blockIdx.x < 4096)                                  \
            mv::g_astamp[16 * blockIdx.x + (slot)] = __builtin_amdgcn_s_memrealtime();   \
        __builtin_amdgcn_sched_barrier(0);                                          \
    } while (0)
#endif
#include "attn_decode_shared.h"

namespace mv {

// q (G heads) | k | v of row b, kv head kvh, from the projection's slabs into LDS as bf16
// [(G + 2) * D]: what the projection's own slab reduction would have stored - the slabs added in order,
// one rounding, then the bias and a second rounding (wgemm.hip: wgemm_finalize_kernel,
// epilogue_value).  The workgroup shares the work, four columns a thread, every slab of a column
// requested before the first is added: one trip to L2 for the lot.  Ends with a barrier.
template <int D, int G>
__device__ __forceinline__ void stage_qkv_from_slabs(const AttnArgs& a, uint16_t* stage, int b, int kvh) {
    constexpr int NC = (G + 2) * D;
    constexpr int kUnroll = 8;
    const int64_t slab_stride = (int64_t)a.bs * a.slab_n;
    for (int c = threadIdx.x * 4; c < NC; c += blockDim.x * 4) {
        const int seg = c / D, within = c - seg * D;
        const int head = seg < G ? kvh * G + seg : seg == G ? a.hq + kvh : a.hq + a.hkv + kvh;
        const int col = head * D + within;
        const float* sp = a.slabs + (int64_t)b * a.slab_n + col;
        f32x4_t v[kUnroll];
#pragma unroll
        for (int ks = 0; ks < kUnroll; ++ks) {
            if (ks >= a.slab_ks) break;   // uniform
            v[ks] = *reinterpret_cast<const f32x4_t*>(sp + ks * slab_stride);
        }
        f32x4_t t = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < kUnroll; ++ks)
            if (ks < a.slab_ks) t += v[ks];
        for (int ks = kUnroll; ks < a.slab_ks; ++ks) t += *reinterpret_cast<const f32x4_t*>(sp + ks * slab_stride);
        float o[4] = {round_bf16(t[0]), round_bf16(t[1]), round_bf16(t[2]), round_bf16(t[3])};
        if (a.slab_bias) {
            const uint2 bb = *reinterpret_cast<const uint2*>(a.slab_bias + col);
            o[0] = round_bf16(o[0] + bf16_lo(bb.x));
            o[1] = round_bf16(o[1] + bf16_hi(bb.x));
            o[2] = round_bf16(o[2] + bf16_lo(bb.y));
            o[3] = round_bf16(o[3] + bf16_hi(bb.y));
        }
        uint2 w;
        w.x = pack_bf16x2(o[0], o[1]);
        w.y = pack_bf16x2(o[2], o[3]);
        *reinterpret_cast<uint2*>(stage + c) = w;
    }
    __syncthreads();
}

template <int D, int G, int W>
__global__ __launch_bounds__(W * 64) void paged_attn_decode_kernel(AttnArgs a) {
    constexpr int kAW = W;          // waves per workgroup
    constexpr int kAC = 4;          // row loads per wave per pipeline stage
    constexpr int LPR = D / 8;      // lanes per row
    constexpr int RPI = 64 / LPR;   // rows per wave instruction
    constexpr int NS = kAW * RPI;   // softmax states per workgroup
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* st_m = lds;                 // [NS][G]
    float* st_l = lds + NS * G;        // [NS][G]
    float* st_acc = lds + 2 * NS * G;  // [NS][G][D]

    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane % LPR, tg = lane / LPR;
    const int d0 = lr * 8;
    MV_ASTAMP(0);

    const int split = blockIdx.x % a.nsplit;
    const int bh = blockIdx.x / a.nsplit;
    const int kvh = bh % a.hkv, b = bh / a.hkv;
    const bool last_split = (split == a.nsplit - 1);

    // keys 0..pos-1 are cached, the new key lands at pos; requested first, needed only after the
    // loads that do not depend on it are on their way
    const int pos = a.seqlens[b];
    const int32_t* bt = a.block_table + (int64_t)b * a.bt_stride;
    const int64_t head_off = (int64_t)kvh * D + d0;
    const int64_t row_stride = (int64_t)a.hkv * D;

    // Rows are dealt to the splits in chunks of CR = NS * kAC consecutive rows, round robin:
    // chunk c of this split is chunk c * nsplit + split of the sequence.  Which rows a workgroup
    // reads therefore does not depend on the sequence length, only how many of them count, so
    // the page ids of its first chunks (and q, k, v) are requested before the length has arrived.
    // Rows at or past the length are read from wherever their block-table entry points (page 0
    // when it points nowhere) and masked out of the scores and of the value sum.
    auto row_of = [&](int c, int i) {
        return (((c * a.nsplit + split) * kAC + i) * kAW + wid) * RPI + tg;
    };
    auto load_pages = [&](int (&pg)[kAC], int c) {
#pragma unroll
        for (int i = 0; i < kAC; ++i) pg[i] = bt[min(row_of(c, i) >> a.page_shift, a.bt_stride - 1)];
    };
    int pgA[kAC], pgB[kAC];
    load_pages(pgA, 0);
    load_pages(pgB, 1);

    const bool rotate = (a.rot_cos != nullptr);
    const bool first_half = d0 < D / 2;
    const int dpart = first_half ? d0 + D / 2 : d0 - D / 2;
    uint4 qx[G], qo[G], kx, ko, vnew;
    ko = make_uint4(0, 0, 0, 0);
    if (a.slabs) {
        // (the state arrays this borrows are first written after the main loop)
        uint16_t* stage = reinterpret_cast<uint16_t*>(lds);
        stage_qkv_from_slabs<D, G>(a, stage, b, kvh);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            qx[g] = *reinterpret_cast<const uint4*>(stage + g * D + d0);
            qo[g] = rotate ? *reinterpret_cast<const uint4*>(stage + g * D + dpart) : qx[g];
        }
        kx = *reinterpret_cast<const uint4*>(stage + G * D + d0);
        if (rotate) ko = *reinterpret_cast<const uint4*>(stage + G * D + dpart);
        vnew = *reinterpret_cast<const uint4*>(stage + (G + 1) * D + d0);
        __syncthreads();
    } else {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const uint16_t* qh = a.q + (int64_t)b * a.q_stride_b + (int64_t)(kvh * G + g) * D;
            qx[g] = ld16(qh + d0);
            qo[g] = rotate ? ld16(qh + dpart) : qx[g];
        }
        const uint16_t* kh = a.k_new + (int64_t)b * a.kv_stride_b + (int64_t)kvh * D;
        const uint16_t* vh = a.v_new + (int64_t)b * a.kv_stride_b + (int64_t)kvh * D;
        kx = ld16(kh + d0);
        if (rotate) ko = ld16(kh + dpart);
        vnew = ld16(vh + d0);
    }

#ifdef MV_ATTN_STAMP
    if (pos >= 0) MV_ASTAMP(11);
#endif
    uint4 cosv = make_uint4(0, 0, 0, 0), sinv = cosv;
    if (rotate) {
        const int rp = min(max(pos, 0), a.rot_max_pos - 1);
        const int64_t ro = (int64_t)rp * (D / 2) + (first_half ? d0 : d0 - D / 2);
        cosv = ld16(a.rot_cos + ro);
        sinv = ld16(a.rot_sin + ro);
    }
    constexpr int CR = NS * kAC;  // rows per chunk
    const int seq_chunks = (max(pos, 0) + CR - 1) / CR;
    const int nchunk = seq_chunks > split ? (seq_chunks - split + a.nsplit - 1) / a.nsplit : 0;

    auto load_rows = [&](uint4(&kb)[kAC], uint4(&vb)[kAC], bool(&ok)[kAC], const int (&pg)[kAC], int c) {
#pragma unroll
        for (int i = 0; i < kAC; ++i) {
            const int t = row_of(c, i);
            const bool page_ok = pg[i] >= 0 && pg[i] < a.num_pages;
            ok[i] = (t < pos) && page_ok;
            const int64_t off =
                (((int64_t)(page_ok ? pg[i] : 0) << a.page_shift) + (t & (a.page - 1))) * row_stride + head_off;
            kb[i] = ld16(a.k_cache + off);
            vb[i] = ld16(a.v_cache + off);
        }
    };

    // Both stages go out as soon as the page ids are in; the page ids of the stages after them
    // follow each stage, so that they are always older than the rows that will wait for them.
    uint4 kA[kAC], vA[kAC], kB[kAC], vB[kAC];
    bool okA[kAC], okB[kAC];
#ifdef MV_ATTN_STAMP
    if (pgA[0] >= -1) MV_ASTAMP(12);
#endif
    load_rows(kA, vA, okA, pgA, 0);
    load_pages(pgA, 2);
    load_rows(kB, vB, okB, pgB, 1);
    load_pages(pgB, 3);

    // ---- q (G heads) and the new k/v: rotate, round -------------------------------
    uint32_t qp[G][4];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        if (rotate) {
            rope8(qx[g], qo[g], cosv, sinv, first_half, qp[g]);
        } else {
            qp[g][0] = qx[g].x; qp[g][1] = qx[g].y; qp[g][2] = qx[g].z; qp[g][3] = qx[g].w;
        }
    }
    uint32_t kp[4];
    if (rotate) {
        rope8(kx, ko, cosv, sinv, first_half, kp);
    } else {
        kp[0] = kx.x; kp[1] = kx.y; kp[2] = kx.z; kp[3] = kx.w;
    }
    // append: the last split owns the new row; one row group writes it
    if (last_split && wid == 0 && tg == 0) {
        const int pi = pos >> a.page_shift;
        if (pos >= 0 && pi < a.bt_stride) {
            const int pg = bt[pi];
            if (pg >= 0 && pg < a.num_pages) {
                const int64_t off =
                    (((int64_t)pg << a.page_shift) + (pos & (a.page - 1))) * a.hkv * D + (int64_t)kvh * D + d0;
                st16(a.k_cache + off, make_uint4(kp[0], kp[1], kp[2], kp[3]));
                st16(a.v_cache + off, vnew);
            }
        }
    }

    // ---- online softmax state of this row group ---------------------------------
    float m[G], l[G], acc[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        m[g] = kNegBig;
        l[g] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[g][j] = 0.f;
    }

    // scores of up to kAC rows + their values -> state
    auto absorb = [&](const uint4(&kb)[kAC], const uint4(&vb)[kAC], const bool(&ok)[kAC]) {
        // values of rows that do not count may be any bit pattern (0 * NaN): zero them
        uint4 vm[kAC];
#pragma unroll
        for (int i = 0; i < kAC; ++i) {
            const uint32_t keep = ok[i] ? 0xFFFFFFFFu : 0u;
            vm[i] = make_uint4(vb[i].x & keep, vb[i].y & keep, vb[i].z & keep, vb[i].w & keep);
        }
        float s[kAC][G];
#pragma unroll
        for (int i = 0; i < kAC; ++i) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float d = 0.f;
                d = dot2_bf16(qp[g][0], kb[i].x, d);
                d = dot2_bf16(qp[g][1], kb[i].y, d);
                d = dot2_bf16(qp[g][2], kb[i].z, d);
                d = dot2_bf16(qp[g][3], kb[i].w, d);
                s[i][g] = row_sum<LPR>(d) * a.scale_log2;
            }
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float mx = m[g];
#pragma unroll
            for (int i = 0; i < kAC; ++i) mx = ok[i] ? fmaxf(mx, s[i][g]) : mx;
            const float alpha = __builtin_amdgcn_exp2f(m[g] - mx);
            m[g] = mx;
            float p[kAC], psum = 0.f;
#pragma unroll
            for (int i = 0; i < kAC; ++i) {
                p[i] = ok[i] ? __builtin_amdgcn_exp2f(s[i][g] - mx) : 0.f;
                psum += p[i];
            }
            l[g] = l[g] * alpha + psum;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[g][j] *= alpha;
#pragma unroll
            for (int i = 0; i < kAC; ++i) {
                const uint32_t vs[4] = {vm[i].x, vm[i].y, vm[i].z, vm[i].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[g][2 * j] = __builtin_fmaf(p[i], bf16_lo(vs[j]), acc[g][2 * j]);
                    acc[g][2 * j + 1] = __builtin_fmaf(p[i], bf16_hi(vs[j]), acc[g][2 * j + 1]);
                }
            }
        }
    };

    MV_ASTAMP(1);
    // Rows of chunks c, c + 1 and page ids of c + 2, c + 3 are in flight at the top of every pass.
    // The steady state has no branch around a request, so the compiler's wait counts are exact
    // (with the requests under `if`s every wait also covered the other stage, and a long split
    // had barely one stage in flight); the last three chunks at most are handled after it.
    int c = 0;
    for (; c + 3 < nchunk; c += 2) {
        absorb(kA, vA, okA);
#ifdef MV_ATTN_STAMP
        if (c == 0) MV_ASTAMP(8);
#endif
        // (the fences keep the scheduler from gathering both stages' arithmetic in front of
        // both stages' requests, which would leave nothing in flight while it runs)
        __builtin_amdgcn_sched_barrier(0);
        load_rows(kA, vA, okA, pgA, c + 2);
        load_pages(pgA, c + 4);
        __builtin_amdgcn_sched_barrier(0);
        absorb(kB, vB, okB);
        __builtin_amdgcn_sched_barrier(0);
        load_rows(kB, vB, okB, pgB, c + 3);
        load_pages(pgB, c + 5);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (c < nchunk) {
        absorb(kA, vA, okA);
        if (c + 2 < nchunk) load_rows(kA, vA, okA, pgA, c + 2);
        if (c + 1 < nchunk) absorb(kB, vB, okB);
        if (c + 2 < nchunk) absorb(kA, vA, okA);
    }

    // ---- the new row (held in registers; never read back from the cache) --------
    // One row, so no chunk machinery: only row group 0 of wave 0 of the last split adds it.
    if (last_split && wid == 0) {  // wave-uniform
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float d = 0.f;
            d = dot2_bf16(qp[g][0], kp[0], d);
            d = dot2_bf16(qp[g][1], kp[1], d);
            d = dot2_bf16(qp[g][2], kp[2], d);
            d = dot2_bf16(qp[g][3], kp[3], d);
            const float sc = row_sum<LPR>(d) * a.scale_log2;
            if (tg == 0) {
                const float mx = fmaxf(m[g], sc);
                const float alpha = __builtin_amdgcn_exp2f(m[g] - mx);
                const float p = __builtin_amdgcn_exp2f(sc - mx);
                m[g] = mx;
                l[g] = l[g] * alpha + p;
                const uint32_t vs[4] = {vnew.x, vnew.y, vnew.z, vnew.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[g][2 * j] = __builtin_fmaf(p, bf16_lo(vs[j]), acc[g][2 * j] * alpha);
                    acc[g][2 * j + 1] = __builtin_fmaf(p, bf16_hi(vs[j]), acc[g][2 * j + 1] * alpha);
                }
            }
        }
    }

    // ---- merge the NS states ----------------------------------------------------
    MV_ASTAMP(2);
    const int sidx = wid * RPI + tg;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        if (lr == 0) {
            st_m[sidx * G + g] = m[g];
            st_l[sidx * G + g] = l[g];
        }
        float* dst = st_acc + ((int64_t)(sidx * G + g)) * D + d0;
        *reinterpret_cast<f32x4_t*>(dst) = (f32x4_t){acc[g][0], acc[g][1], acc[g][2], acc[g][3]};
        *reinterpret_cast<f32x4_t*>(dst + 4) = (f32x4_t){acc[g][4], acc[g][5], acc[g][6], acc[g][7]};
    }
    finish_states<D, G, NS, kAW>(a, lds, b, kvh, bh, split, wid, lane);
}

// ------------------------------------------------------------ matrix-core form (D = 128, G >= 4)
// The dot-product kernel above spends its time in the vector unit once four or more query heads
// share a kv head (G dots, G lane reductions and G x 8 multiply-adds per row).  Here a wave takes
// 32 rows at a time and gives both products to the matrix cores:
//   S^T[t][g] = K[t][:] . Q[g][:]      A = 16 key rows as loaded (lane (r, q): row r, 8 values of
//                                      k-slice q), B = the rotated query heads, padded to 16
//   O^T[d][g] += V^T[d][t] . P^T[t][g] A = values transposed through LDS, B = P rounded to bf16
// The k index of the second product is the order the first one leaves its scores in (lane (g, q')
// holds rows 4q'..4q'+3 of both 16-row blocks), and the values are read in that order, so P never
// moves between lanes.  A lane reads two neighbouring d of a value row as one dword and gives the
// low halves to one product and the high halves to another: output row m of pair p is
// d = 32p + 2m (+1), which leaves lane (g, q') with O[g][32p + 8q' .. + 7].
// The softmax state is per wave (4 per workgroup); everything after it is finish_states.
// Where it stops: 32 sequences x 8 kv heads at ctx 4096 run 95-98 us (5.5-5.7 TB/s) in this form
// and in the dot-product form alike, with 4 or 8 waves per workgroup, one or two workgroups per
// CU, and with the kv heads' rows interleaved (2 KB stride) or back to back
// (tools/attn_stride_probe.py): neither arithmetic, bytes in flight nor the row stride set that
// pace.
constexpr int kMRows = 32;        // rows per wave per stage
constexpr int kMVRow = 272;       // bytes per staged value row (256 + 16: conflict-free both ways)

template <int G>
__global__ __launch_bounds__(256) void paged_attn_decode_mfma_kernel(AttnArgs a) {
    constexpr int D = 128, kAW = 4, NS = kAW;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, q = lane >> 4;
    unsigned char* vst = reinterpret_cast<unsigned char*>(lds) + wid * (kMRows * kMVRow);
    MV_ASTAMP(0);

    const int split = blockIdx.x % a.nsplit;
    const int bh = blockIdx.x / a.nsplit;
    const int kvh = bh % a.hkv, b = bh / a.hkv;
    const bool last_split = (split == a.nsplit - 1);

    const int pos = a.seqlens[b];
    const int32_t* bt = a.block_table + (int64_t)b * a.bt_stride;
    const int64_t row_stride = (int64_t)a.hkv * D;
    const int64_t head_off = (int64_t)kvh * D;

    // 32-row blocks of the sequence are dealt round robin over (split, wave): which rows a wave
    // reads does not depend on the sequence length, only how many of them count.
    const int blk_first = split * kAW + wid, blk_step = a.nsplit * kAW;
    auto page_of = [&](int c) {
        const int t0 = (blk_first + c * blk_step) * kMRows;
        return bt[min(t0 >> a.page_shift, a.bt_stride - 1)];
    };
    int pgA = page_of(0), pgB = page_of(1);   // pages of the current and the next stage

    // ---- query heads (B operand: lane (r, q) = head r, 8 values at 32 kk + 8 q), new k / v ----
    const bool rotate = (a.rot_cos != nullptr);
    const int gq = r < G ? r : 0;
    const uint16_t* qh = a.q + (int64_t)b * a.q_stride_b + (int64_t)(kvh * G + gq) * D + q * 8;
    const uint16_t* kh = a.k_new + (int64_t)b * a.kv_stride_b + (int64_t)kvh * D + q * 8;
    const uint16_t* vh = a.v_new + (int64_t)b * a.kv_stride_b + (int64_t)kvh * D + q * 8;
    uint4 qx[4], kx[4], vnew[4];
    if (a.slabs) {
        // (the value staging this borrows is first written in the main loop, past the barrier below)
        uint16_t* stage = reinterpret_cast<uint16_t*>(lds);
        stage_qkv_from_slabs<128, G>(a, stage, b, kvh);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            qx[kk] = *reinterpret_cast<const uint4*>(stage + gq * 128 + q * 8 + kk * 32);
            kx[kk] = *reinterpret_cast<const uint4*>(stage + G * 128 + q * 8 + kk * 32);
            vnew[kk] = *reinterpret_cast<const uint4*>(stage + (G + 1) * 128 + q * 8 + kk * 32);
        }
        __syncthreads();
    } else {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            qx[kk] = ld16(qh + kk * 32);
            kx[kk] = ld16(kh + kk * 32);
            vnew[kk] = ld16(vh + kk * 32);
        }
    }

    auto load_k = [&](uint4(&kb)[8], int pg, int c) {
        const int t0 = (blk_first + c * blk_step) * kMRows;
        const bool page_ok = pg >= 0 && pg < a.num_pages;
        const int64_t row0 = ((int64_t)(page_ok ? pg : 0) << a.page_shift) + (t0 & (a.page - 1));
        const uint16_t* kp0 = a.k_cache + (row0 + r) * row_stride + head_off + q * 8;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) kb[blk * 4 + kk] = ld16(kp0 + blk * 16 * row_stride + kk * 32);
    };
    auto load_v = [&](uint4(&vb)[8], int pg, int c) {
        const int t0 = (blk_first + c * blk_step) * kMRows;
        const bool page_ok = pg >= 0 && pg < a.num_pages;
        const int64_t row0 = ((int64_t)(page_ok ? pg : 0) << a.page_shift) + (t0 & (a.page - 1));
        const uint16_t* vp0 = a.v_cache + (row0 + q) * row_stride + head_off + r * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) vb[i] = ld16(vp0 + i * 4 * row_stride);
    };
    // One stage of rows in registers: its value rows go to LDS and its key rows into the matrix
    // unit at the top of a pass, and the next stage is requested into the same registers right
    // behind them, so it has the softmax and the second product to arrive (other waves of the CU
    // cover the rest).  Two register stages cost 64 more registers and a wave per SIMD, and
    // were measured slower even where there is one workgroup per CU anyway (8 sequences x 1 kv
    // head at ctx 8192: 16.4 against 16.2 us; one sequence x 8 kv heads at ctx 4096: 12.7 against
    // 11.3): the second stage's requests are wasted on short splits and lengthen the prologue.
    uint4 kR[8], vR[8];
    load_k(kR, pgA, 0);
    load_v(vR, pgA, 0);
    uint4 cosv[2], sinv[2];
    cosv[0] = cosv[1] = sinv[0] = sinv[1] = make_uint4(0, 0, 0, 0);
    if (rotate) {
        const int rp = min(max(pos, 0), a.rot_max_pos - 1);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            cosv[h] = ld16(a.rot_cos + (int64_t)rp * (D / 2) + h * 32 + q * 8);
            sinv[h] = ld16(a.rot_sin + (int64_t)rp * (D / 2) + h * 32 + q * 8);
        }
    }
    const int nblk = (max(pos, 0) + kMRows - 1) / kMRows;
    const int nchunk = nblk > blk_first ? (nblk - blk_first + blk_step - 1) / blk_step : 0;

    // pieces kk and kk ^ 2 are half a head apart: the rotation needs no other lane
    bf16x8_t qf[4];
    uint32_t kp[4][4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        uint32_t o[4];
        if (rotate) {
            rope8(qx[kk], qx[kk ^ 2], cosv[kk & 1], sinv[kk & 1], kk < 2, o);
            rope8(kx[kk], kx[kk ^ 2], cosv[kk & 1], sinv[kk & 1], kk < 2, kp[kk]);
        } else {
            o[0] = qx[kk].x; o[1] = qx[kk].y; o[2] = qx[kk].z; o[3] = qx[kk].w;
            kp[kk][0] = kx[kk].x; kp[kk][1] = kx[kk].y; kp[kk][2] = kx[kk].z; kp[kk][3] = kx[kk].w;
        }
        const uint32_t keep = r < G ? 0xFFFFFFFFu : 0u;   // padding heads score zero
        qf[kk] = __builtin_bit_cast(bf16x8_t, make_uint4(o[0] & keep, o[1] & keep, o[2] & keep, o[3] & keep));
    }
    // append: the last split owns the new row; the four lanes with r == 0 of wave 0 hold all of it
    if (last_split && wid == 0 && r == 0) {
        const int pi = pos >> a.page_shift;
        if (pos >= 0 && pi < a.bt_stride) {
            const int pg = bt[pi];
            if (pg >= 0 && pg < a.num_pages) {
                const int64_t off =
                    (((int64_t)pg << a.page_shift) + (pos & (a.page - 1))) * row_stride + head_off + q * 8;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    st16(a.k_cache + off + kk * 32, make_uint4(kp[kk][0], kp[kk][1], kp[kk][2], kp[kk][3]));
                    st16(a.v_cache + off + kk * 32, vnew[kk]);
                }
            }
        }
    }

    // ---- state of this wave: lane (g = r, q') holds m[g], a quarter of l[g], O[g][32p + 8q' ..] ----
    float m = kNegBig, l = 0.f;
    f32x4_t oacc[4][2];
#pragma unroll
    for (int p = 0; p < 4; ++p) oacc[p][0] = oacc[p][1] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    // The new row comes from registers and goes in FIRST (the order of rows does not matter to the
    // running softmax), so that its registers are free during the loop: wave 0 of the last split.
    if (last_split && wid == 0) {  // wave-uniform
        float d = 0.f;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const uint4 qv = __builtin_bit_cast(uint4, qf[kk]);
            d = dot2_bf16(qv.x, kp[kk][0], d);
            d = dot2_bf16(qv.y, kp[kk][1], d);
            d = dot2_bf16(qv.z, kp[kk][2], d);
            d = dot2_bf16(qv.w, kp[kk][3], d);
        }
        d += __shfl_xor(d, 16);
        d += __shfl_xor(d, 32);
        m = d * a.scale_log2;
        l = q == 0 ? 1.f : 0.f;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const uint32_t vs[4] = {vnew[p].x, vnew[p].y, vnew[p].z, vnew[p].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                oacc[p][0][j] = bf16_lo(vs[j]);
                oacc[p][1][j] = bf16_hi(vs[j]);
            }
        }
    }

    auto absorb = [&](uint4(&kb)[8], uint4(&vb)[8], int pg, int pg_next, int c, bool more) {
        const int t0 = (blk_first + c * blk_step) * kMRows;
        const bool page_ok = pg >= 0 && pg < a.num_pages;
        // values of rows that do not count may be any bit pattern: zero them on the way into LDS
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = i * 4 + q;
            const uint32_t keep = (page_ok && t0 + row < pos) ? 0xFFFFFFFFu : 0u;
            *reinterpret_cast<uint4*>(vst + row * kMVRow + r * 16) =
                make_uint4(vb[i].x & keep, vb[i].y & keep, vb[i].z & keep, vb[i].w & keep);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (more) load_v(vb, pg_next, c + 1);
        __builtin_amdgcn_sched_barrier(0);
        f32x4_t s0 = (f32x4_t){0.f, 0.f, 0.f, 0.f}, s1 = s0;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kb[kk]), qf[kk], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kb[4 + kk]), qf[kk], s1, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (more) load_k(kb, pg_next, c + 1);
        __builtin_amdgcn_sched_barrier(0);
        // lane (g, q'): s0[j] = row t0 + 4q' + j, s1[j] = row t0 + 16 + 4q' + j
        float sc[8];
        bool ok[8];
        float mx = kNegBig;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int t = t0 + (j < 4 ? 0 : 16) + 4 * q + (j & 3);
            ok[j] = page_ok && t < pos;
            sc[j] = (j < 4 ? s0[j & 3] : s1[j & 3]) * a.scale_log2;
            mx = ok[j] ? fmaxf(mx, sc[j]) : mx;
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        mx = fmaxf(mx, m);
        const float alpha = __builtin_amdgcn_exp2f(m - mx);
        m = mx;
        float pv[8], psum = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            pv[j] = ok[j] ? __builtin_amdgcn_exp2f(sc[j] - mx) : 0.f;
            psum += pv[j];
        }
        l = l * alpha + psum;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            oacc[p][0] *= alpha;
            oacc[p][1] *= alpha;
        }
        const bf16x8_t pf = __builtin_bit_cast(
            bf16x8_t, make_uint4(pack_bf16x2(pv[0], pv[1]), pack_bf16x2(pv[2], pv[3]),
                                 pack_bf16x2(pv[4], pv[5]), pack_bf16x2(pv[6], pv[7])));
        __builtin_amdgcn_wave_barrier();  // LDS operations of a wave run in order: no wait needed
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            uint32_t w[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = (i < 4 ? 0 : 16) + 4 * q + (i & 3);
                w[i] = *reinterpret_cast<const uint32_t*>(vst + row * kMVRow + (16 * p + r) * 4);
            }
            uint32_t lo[4], hi[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                lo[j] = __builtin_amdgcn_perm(w[2 * j + 1], w[2 * j], 0x05040100u);
                hi[j] = __builtin_amdgcn_perm(w[2 * j + 1], w[2 * j], 0x07060302u);
            }
            oacc[p][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                __builtin_bit_cast(bf16x8_t, make_uint4(lo[0], lo[1], lo[2], lo[3])), pf, oacc[p][0], 0, 0, 0);
            oacc[p][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                __builtin_bit_cast(bf16x8_t, make_uint4(hi[0], hi[1], hi[2], hi[3])), pf, oacc[p][1], 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
    };

    MV_ASTAMP(1);
    for (int c = 0; c < nchunk; ++c) {
        const int pgC = page_of(c + 2);
        absorb(kR, vR, pgA, pgB, c, c + 1 < nchunk);
        pgA = pgB;
        pgB = pgC;
    }

    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);

    // ---- the four wave states into LDS (over the value staging: everybody is done with it) ----
    MV_ASTAMP(2);
    __syncthreads();
    float* st_m = lds;
    float* st_l = lds + NS * G;
    float* st_acc = lds + 2 * NS * G;
    if (r < G) {
        if (q == 0) {
            st_m[wid * G + r] = m;
            st_l[wid * G + r] = l;
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float* dst = st_acc + (int64_t)(wid * G + r) * D + 32 * p + 8 * q;
            *reinterpret_cast<f32x4_t*>(dst) = (f32x4_t){oacc[p][0][0], oacc[p][1][0], oacc[p][0][1], oacc[p][1][1]};
            *reinterpret_cast<f32x4_t*>(dst + 4) = (f32x4_t){oacc[p][0][2], oacc[p][1][2], oacc[p][0][3], oacc[p][1][3]};
        }
    }
    finish_states<D, G, NS, kAW>(a, lds, b, kvh, bh, split, wid, lane);
}

// Second-launch form of the split merge (used when the arrival counters are unavailable or
// switched off).  Same arithmetic and summation order as the in-kernel merge: PARTS interleaved
// partial sums per output, added in order.
template <int D, int G, int W>
__global__ __launch_bounds__(D) void paged_attn_combine_kernel(const float* __restrict__ part_lse,
                                                               const float* __restrict__ part_o,
                                                               uint16_t* __restrict__ out,
                                                               int nsplit) {
    constexpr int ITEMS = G * D / 4;
    constexpr int PARTS = W * 64 / ITEMS >= 1 ? W * 64 / ITEMS : 1;
    const int64_t bh = blockIdx.x;  // (b, query head)
    const int d = threadIdx.x;
    const float* lse = part_lse + bh * nsplit;
    __shared__ float wts[128];
    float total = 0.f;
    if (nsplit <= 128) {  // same weight arithmetic as the in-kernel merge: one split per lane
        if (threadIdx.x < 64) {
            const int lane = threadIdx.x;
            const float lse0 = lane < nsplit ? lse[lane] : kNegBig;
            const float lse1 = lane + 64 < nsplit ? lse[lane + 64] : kNegBig;
            const float mx = wave_max(fmaxf(lse0, lse1));
            const float w0 = lane < nsplit ? __builtin_amdgcn_exp2f(lse0 - mx) : 0.f;
            const float w1 = lane + 64 < nsplit ? __builtin_amdgcn_exp2f(lse1 - mx) : 0.f;
            const float sum = wave_sum(w0 + w1);
            const float inv = sum > 0.f ? 1.0f / sum : 0.f;
            wts[lane] = w0 * inv;
            wts[lane + 64] = w1 * inv;
        }
        __syncthreads();
        for (int p = 0; p < PARTS; ++p) {
            float o = 0.f;
            for (int s = p; s < nsplit; s += PARTS)
                o = __builtin_fmaf(wts[s], part_o[(bh * nsplit + s) * D + d], o);
            total = p == 0 ? o : total + o;
        }
    } else {
        float mx = kNegBig;
        for (int s = 0; s < nsplit; ++s) mx = fmaxf(mx, lse[s]);
        float wsum = 0.f;
        for (int s = 0; s < nsplit; ++s) wsum += __builtin_amdgcn_exp2f(lse[s] - mx);
        const float inv = wsum > 0.f ? 1.0f / wsum : 0.f;
        for (int p = 0; p < PARTS; ++p) {
            float o = 0.f;
            for (int s = p; s < nsplit; s += PARTS)
                o = __builtin_fmaf(__builtin_amdgcn_exp2f(lse[s] - mx) * inv, part_o[(bh * nsplit + s) * D + d], o);
            total = p == 0 ? o : total + o;
        }
    }
    out[bh * D + d] = (uint16_t)f32_to_bf16_bits(total);
}

template <int D, int G, int W>
static int launch_decode_w(const AttnArgs& a, hipStream_t s) {
    constexpr int kAT = W * 64;
    constexpr int NS = W * (64 / (D / 8));
    const size_t lds = (size_t)(2 * NS * G + NS * G * D) * sizeof(float);
    static_assert((size_t)(2 * NS * G + NS * G * D) * sizeof(float) <= 160 * 1024 - 1024, "state merge fits LDS");
    static_assert((size_t)(G * 128 + kAT * 4) * sizeof(float) <= (size_t)(2 * NS * G + NS * G * D) * sizeof(float),
                  "split merge reuses the state arrays");
    static LdsOptIn opted;  // per instantiation
    if (int rc = ensure_dynamic_lds((const void*)paged_attn_decode_kernel<D, G, W>, lds, opted)) return rc;
    hipLaunchKernelGGL((paged_attn_decode_kernel<D, G, W>), dim3(a.bs * a.hkv * a.nsplit), dim3(kAT),
                       lds, s, a);
    int rc = check_launch();
    if (rc != MV_OK || a.nsplit == 1 || a.arrive != nullptr) return rc;
    hipLaunchKernelGGL((paged_attn_combine_kernel<D, G, W>), dim3(a.bs * a.hq), dim3(D), 0, s,
                       a.part_lse, a.part_o, a.out, a.nsplit);
    return check_launch();
}

// Four waves per workgroup.  Eight or sixteen (with one workgroup per CU, so that more waves share
// the score arithmetic) were measured on MI355X at bs = 1: 11.8 us against 11.9 at ctx 4096,
// 14.9 against 15.4 at ctx 8192 with eight; slower with sixteen (the state merge grows with the
// number of row groups).
template <int D, int G>
static int launch_decode(const AttnArgs& a, hipStream_t s) {
    return launch_decode_w<D, G, 4>(a, s);
}

template <int G>
static int launch_decode_mfma(const AttnArgs& a, hipStream_t s) {
    constexpr int D = 128, NS = 4, kAT = 256;
    constexpr size_t stage = (size_t)4 * kMRows * kMVRow;
    constexpr size_t state = (size_t)(2 * NS * G + NS * G * D) * sizeof(float);
    constexpr size_t merge = (size_t)(G * 128 + kAT * 4) * sizeof(float);
    constexpr size_t lds = stage > state ? (stage > merge ? stage : merge) : (state > merge ? state : merge);
    static_assert(lds <= 64 * 1024, "no opt-in needed");
    hipLaunchKernelGGL((paged_attn_decode_mfma_kernel<G>), dim3(a.bs * a.hkv * a.nsplit), dim3(kAT), lds, s, a);
    int rc = check_launch();
    if (rc != MV_OK || a.nsplit == 1 || a.arrive != nullptr) return rc;
    hipLaunchKernelGGL((paged_attn_combine_kernel<D, G, 4>), dim3(a.bs * a.hq), dim3(D), 0, s,
                       a.part_lse, a.part_o, a.out, a.nsplit);
    return check_launch();
}

static std::atomic<int> g_tune_mfma{0};  // matrix-core form: 0 = built-in rule, 1 = never, 2 = whenever it applies

// Measured on MI355X (tools/sweep_attn_mfma.py): with eight query heads per kv head the
// matrix-core form wins everywhere (8 sequences x 1 kv head at ctx 8192: 16.1 us against 21.6);
// with four the two forms are within 3 % of each other at every batch size (both stop at about
// 5.6 TB/s), and the dot-product form keeps P in fp32.
static bool use_mfma_form(int d, int g, int page) {
    const int forced = g_tune_mfma.load(std::memory_order_relaxed);
    if (d != 128 || (g != 4 && g != 8) || (page % kMRows) != 0 || forced == 1) return false;
    return forced == 2 || g == 8;
}

template <int D>
static int dispatch_g(const AttnArgs& a, int g, hipStream_t s) {
    if (use_mfma_form(D, g, a.page)) return g == 4 ? launch_decode_mfma<4>(a, s) : launch_decode_mfma<8>(a, s);
    switch (g) {
        case 1: return launch_decode<D, 1>(a, s);
        case 2: return launch_decode<D, 2>(a, s);
        case 3: return launch_decode<D, 3>(a, s);
        case 4: return launch_decode<D, 4>(a, s);
        case 5: return launch_decode<D, 5>(a, s);
        case 6: return launch_decode<D, 6>(a, s);
        case 7: return launch_decode<D, 7>(a, s);
        case 8: return launch_decode<D, 8>(a, s);
        default: return MV_ERR_UNSUPPORTED;
    }
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
// Experiment knobs (mv_paged_attn_tune): atomics, so a server thread reading them while another
// thread flips one is not a data race; a call reads each of them once.
static std::atomic<int> g_tune_inkernel_merge{1};
static std::atomic<int> g_tune_target_blocks_x{0};  // workgroups per CU the split count aims at (0 = built-in rule)

}  // namespace mv

extern "C" {

size_t mv_paged_attn_decode_workspace_bytes(int bs, int hq, int d, int num_splits) {
    if (bs <= 0 || hq <= 0 || d <= 0 || num_splits <= 1) return 0;
    const size_t rows = (size_t)bs * hq * num_splits;
    // arrival counters | log-sum-exps | partial outputs
    return mv::kArriveBytes + mv::align_up(rows * sizeof(float), 256) + rows * d * sizeof(float);
}

int mv_paged_attn_decode_pick_splits(int bs, int hkv, int d, int ctx_hint) {
    if (bs <= 0 || hkv <= 0) return 1;
    // Every split adds a fixed cost (state merge, partial write, arrival ticket) and the merging
    // workgroup reads all of them, so the count aims at one workgroup per CU: measured on MI355X,
    // bs=1 / 8 kv heads runs 12.3 us at 32 splits against 14.8 at 16 and 14.9 at 64 for ctx 4096,
    // 16.8 / 21.6 / 17.4 us for ctx 8192; 8 sequences x 1 kv head at ctx 8192 run 23.3 us at 32
    // splits, 29.7 at 16 and 37.6 at 64.  Large batches get what the same rule gives them: 16
    // sequences x 8 kv heads at ctx 4096 run 53.2 us at 2 splits against 58.1 at 4, 32 sequences
    // 94.9 us unsplit against 105.0 at 2.
    const int pairs = bs * hkv;
    const int per_cu = mv::g_tune_target_blocks_x.load(std::memory_order_relaxed);  // 0 = the rule above
    const int cus = mv::num_cus();
    int ns;
    if (per_cu > 0) {
        ns = (per_cu * cus + pairs - 1) / pairs;
    } else {
        // ... which, when the pairs do not divide the CUs, means: the split count whose
        // workgroups load the fullest CU least, ceil(pairs * ns / CUs) / ns of a sequence, with a
        // small charge per split (40 sequences x 8 kv heads: 1 split leaves 64 CUs with two whole
        // sequences, 4 splits leave every CU with 1.25)
        float best = 1e30f;
        ns = 1;
        for (int c = 1; c <= 32; ++c) {
            const float cost = (float)((pairs * c + cus - 1) / cus) / (float)c + 0.0005f * (float)c;
            if (cost < best) {
                best = cost;
                ns = c;
            }
        }
    }
    if (ctx_hint > 0) {
        // rows are dealt out in chunks of 64 (d = 128) or 128 (d = 64): no more splits than chunks
        const int chunk_rows = d == 64 ? 128 : 64;
        const int cap = ctx_hint / chunk_rows > 1 ? ctx_hint / chunk_rows : 1;
        if (ns > cap) ns = cap;
    }
    if (ns > 32) ns = 32;  // 64 never beat 32 (one kv head at ctx 4096: 9.5 vs 9.2 us; ctx 16384: equal)
    if (ns < 1) ns = 1;
    return ns;
}

int mv_paged_attn_decode_form(int hq, int hkv, int d, int page) {
    if (hq <= 0 || hkv <= 0 || hq % hkv) return -1;
    return mv::use_mfma_form(d, hq / hkv, page) ? 1 : 0;
}

/* Experiment knobs: key 0 = merge splits inside the kernel (1) or in a second launch (0);
 * key 1 = workgroups per CU the automatic split count aims at (0 = built-in rule). */
int mv_paged_attn_tune(int key, int value) {
    if (key == 0 && (value == 0 || value == 1)) {
        mv::g_tune_inkernel_merge.store(value, std::memory_order_relaxed);
        return MV_OK;
    }
    if (key == 1 && value >= 0 && value <= 8) {
        mv::g_tune_target_blocks_x.store(value, std::memory_order_relaxed);
        return MV_OK;
    }
    if (key == 2 && value >= 0 && value <= 2) {
        mv::g_tune_mfma.store(value, std::memory_order_relaxed);
        return MV_OK;
    }
    return MV_ERR_INVALID_ARG;
}

#ifdef MV_ATTN_STAMP
int mv_debug_read_attn_stamps(void* dst, size_t bytes) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(mv::g_astamp), bytes) == hipSuccess ? 0 : -3;
}
int mv_debug_clear_attn_stamps(void) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(mv::g_astamp)) != hipSuccess) return -3;
    return hipMemset(p, 0, sizeof(unsigned long long) * 16 * 4096) == hipSuccess ? 0 : -3;
}
#endif

}  // extern "C"

namespace mv {
// The decode entry points: q / k_new / v_new as bf16 rows, or (slabs != null) as the k-slabs of the
// QKV projection.
static int decode_entry(const void* q, int64_t q_stride_b, const void* k_new, const void* v_new,
                        int64_t kv_stride_b, const float* slabs, int slab_ks, const void* slab_bias,
                        void* k_cache, void* v_cache, const void* rot_cos, const void* rot_sin,
                        int rot_max_pos, const int32_t* cache_seqlens, const int32_t* block_table,
                        int bt_stride, void* out, void* workspace, size_t workspace_bytes, int bs, int hq,
                        int hkv, int d, int page, int num_pages, float scale, int num_splits,
                        mv_stream_t stream) {
    if (!q || !k_new || !v_new || !k_cache || !v_cache || !cache_seqlens || !block_table || !out)
        return MV_ERR_INVALID_ARG;
    if (bs <= 0 || hq <= 0 || hkv <= 0 || page <= 0 || num_pages <= 0 || bt_stride <= 0 ||
        num_splits <= 0)
        return MV_ERR_INVALID_ARG;
    if ((rot_cos == nullptr) != (rot_sin == nullptr)) return MV_ERR_INVALID_ARG;
    if (rot_cos && rot_max_pos <= 0) return MV_ERR_INVALID_ARG;
    if (d != 64 && d != 128) return MV_ERR_UNSUPPORTED;
    if (hq % hkv) return MV_ERR_UNSUPPORTED;
    if (page & (page - 1)) return MV_ERR_UNSUPPORTED;  // rows are located with shifts
    if (!aligned16(q) || !aligned16(k_new) || !aligned16(v_new) || !aligned16(k_cache) ||
        !aligned16(v_cache) || !aligned16(out) || (q_stride_b % 8) || (kv_stride_b % 8))
        return MV_ERR_INVALID_ARG;
    if (rot_cos && (!aligned16(rot_cos) || !aligned16(rot_sin))) return MV_ERR_INVALID_ARG;

    AttnArgs a;
    a.q = (const uint16_t*)q;
    a.q_stride_b = q_stride_b;
    a.k_new = (const uint16_t*)k_new;
    a.v_new = (const uint16_t*)v_new;
    a.kv_stride_b = kv_stride_b;
    a.k_cache = (uint16_t*)k_cache;
    a.v_cache = (uint16_t*)v_cache;
    a.rot_cos = (const uint16_t*)rot_cos;
    a.rot_sin = (const uint16_t*)rot_sin;
    a.rot_max_pos = rot_max_pos;
    a.seqlens = cache_seqlens;
    a.block_table = block_table;
    a.bt_stride = bt_stride;
    a.out = (uint16_t*)out;
    a.bs = bs;
    a.hq = hq;
    a.hkv = hkv;
    a.page = page;
    a.page_shift = __builtin_ctz((unsigned)page);
    a.num_pages = num_pages;
    a.scale_log2 = scale * 1.4426950408889634f;
    a.nsplit = num_splits;
    a.part_lse = nullptr;
    a.part_o = nullptr;
    if (num_splits > 1) {
        const size_t need = mv_paged_attn_decode_workspace_bytes(bs, hq, d, num_splits);
        if (!workspace || workspace_bytes < need || !aligned16(workspace)) return MV_ERR_WORKSPACE;
        const size_t rows = (size_t)bs * hq * num_splits;
        a.part_lse = (float*)((char*)workspace + kArriveBytes);
        a.part_o = (float*)((char*)workspace + kArriveBytes + align_up(rows * sizeof(float), 256));
    }
    a.arrive = nullptr;
    if (num_splits > 1 && num_splits <= 128 && g_tune_inkernel_merge.load(std::memory_order_relaxed) &&
        bs * hkv <= kMaxArrive)
        a.arrive = (unsigned int*)workspace;
    a.slabs = slabs;
    a.slab_ks = slab_ks;
    a.slab_n = (hq + 2 * hkv) * d;
    a.slab_bias = (const uint16_t*)slab_bias;
    const int g = hq / hkv;
    hipStream_t s = (hipStream_t)stream;
    return d == 128 ? dispatch_g<128>(a, g, s) : dispatch_g<64>(a, g, s);
}
}  // namespace mv

extern "C" {

int mv_paged_attn_decode(const void* q, int64_t q_stride_b, const void* k_new, const void* v_new,
                         int64_t kv_stride_b, void* k_cache, void* v_cache, const void* rot_cos,
                         const void* rot_sin, int rot_max_pos, const int32_t* cache_seqlens,
                         const int32_t* block_table, int bt_stride, void* out, void* workspace,
                         size_t workspace_bytes, int bs, int hq, int hkv, int d, int page,
                         int num_pages, float scale, int num_splits, mv_stream_t stream) {
    return mv::decode_entry(q, q_stride_b, k_new, v_new, kv_stride_b, nullptr, 0, nullptr, k_cache, v_cache,
                            rot_cos, rot_sin, rot_max_pos, cache_seqlens, block_table, bt_stride, out,
                            workspace, workspace_bytes, bs, hq, hkv, d, page, num_pages, scale, num_splits,
                            stream);
}

int mv_paged_attn_decode_slabs(const void* qkv_slabs, int ksplit, const void* qkv_bias, void* k_cache,
                               void* v_cache, const void* rot_cos, const void* rot_sin, int rot_max_pos,
                               const int32_t* cache_seqlens, const int32_t* block_table, int bt_stride,
                               void* out, void* workspace, size_t workspace_bytes, int bs, int hq, int hkv,
                               int d, int page, int num_pages, float scale, int num_splits,
                               mv_stream_t stream) {
    if (!qkv_slabs || ksplit < 1 || ksplit > 64 || !mv::aligned16(qkv_slabs)) return MV_ERR_INVALID_ARG;
    if (qkv_bias && !mv::aligned16(qkv_bias)) return MV_ERR_INVALID_ARG;
    // (the row pointers are not read in this mode; they only have to pass the checks)
    return mv::decode_entry(qkv_slabs, 0, qkv_slabs, qkv_slabs, 0, (const float*)qkv_slabs, ksplit, qkv_bias,
                            k_cache, v_cache, rot_cos, rot_sin, rot_max_pos, cache_seqlens, block_table,
                            bt_stride, out, workspace, workspace_bytes, bs, hq, hkv, d, page, num_pages,
                            scale, num_splits, stream);
}

}  // extern "C"

// ---------------------------------------------------------------- the entry point that allocates nothing
// For callers that cannot allocate device memory (code under stream capture, serving loops,
// harnesses that forbid it): mv_paged_attn_decode with the split-KV workspace (arrival counters,
// log-sum-exps, partial outputs) in a fixed array of the library instead of a caller buffer.
// Every GPU the library runs on holds its own copy of the array, at its own address, until the
// process ends; every call on a GPU uses that GPU's copy, so calls that overlap on one GPU (two
// streams, two threads) race on it and must not be made.
namespace mv {

constexpr size_t kScratchBytes = (size_t)8 << 20;  // 8 MiB: 1 MiB of counters + 7 MiB of partials
static __device__ __align__(256) unsigned char scratch[kScratchBytes];

// Device address of `scratch` on the current device (the array's name is only a host-side handle:
// it must never be passed to a kernel or to a memset).  Resolved once per device.
static unsigned char* scratch_on_current_device() {
    static std::atomic<unsigned char*> cached[64];
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= 64) {
        set_last_hip_error(e != hipSuccess ? e : hipErrorInvalidDevice);
        return nullptr;
    }
    unsigned char* p = cached[dev].load(std::memory_order_acquire);
    if (!p) {
        void* sym = nullptr;
        e = hipGetSymbolAddress(&sym, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !sym) {
            set_last_hip_error(e != hipSuccess ? e : hipErrorInvalidSymbol);
            return nullptr;
        }
        p = (unsigned char*)sym;
        cached[dev].store(p, std::memory_order_release);
    }
    return p;
}

}  // namespace mv

extern "C" int mv_paged_attn_decode_noalloc(const void* q, int64_t q_stride_b, const void* k_new,
                                            const void* v_new, int64_t kv_stride_b, void* k_cache,
                                            void* v_cache, const void* rot_cos, const void* rot_sin,
                                            int rot_max_pos, const int32_t* cache_seqlens,
                                            const int32_t* block_table, int bt_stride, void* out, int bs,
                                            int hq, int hkv, int d, int page, int num_pages, float scale,
                                            int num_splits, mv_stream_t stream) {
    using namespace mv;
    if (bs <= 0 || hq <= 0 || hkv <= 0 || d <= 0 || num_splits <= 0) return MV_ERR_INVALID_ARG;
    const size_t need = mv_paged_attn_decode_workspace_bytes(bs, hq, d, num_splits);
    if (need > kScratchBytes) {
        char msg[200];
        snprintf(msg, sizeof(msg),
                 "mv_paged_attn_decode_noalloc: %d rows x %d heads x %d splits need %zu bytes of split "
                 "partials, the built-in scratch holds %zu; use fewer splits or mv_paged_attn_decode",
                 bs, hq, num_splits, need, kScratchBytes);
        set_last_error_text(msg);
        return MV_ERR_WORKSPACE;
    }
    unsigned char* ws = scratch_on_current_device();
    if (!ws) return MV_ERR_LAUNCH;
    if (num_splits > 1) {
        // the array keeps what the last call left: the counters this call uses must start at zero
        // (a completed call leaves them so; one that was cut short may not have)
        const size_t pairs = (size_t)bs * hkv < (size_t)kMaxArrive ? (size_t)bs * hkv : (size_t)kMaxArrive;
        hipError_t e = hipMemsetAsync(ws, 0, pairs * kArriveStride * sizeof(unsigned int), (hipStream_t)stream);
        if (e != hipSuccess) {
            set_last_hip_error(e);
            return MV_ERR_LAUNCH;
        }
    }
    return mv_paged_attn_decode(q, q_stride_b, k_new, v_new, kv_stride_b, k_cache, v_cache, rot_cos, rot_sin,
                                rot_max_pos, cache_seqlens, block_table, bt_stride, out, ws, kScratchBytes, bs,
                                hq, hkv, d, page, num_pages, scale, num_splits, stream);
}
