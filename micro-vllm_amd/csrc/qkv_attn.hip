// QKV projection and single-token paged attention of a decode step in ONE launch (1-4 rows).
//
// As two launches the attention is a chain of dependent trips to memory that starts only when
// the projection has ended: sequence length and page ids -> key/value rows -> state merge ->
// partial hand-over -> ticket -> merge of the splits; 16.8 MB of rows take 12-13 us of an
// 8B-shape step at 4096 tokens of context where streaming them would take 3.  The first two
// trips need nothing the projection computes.  Here one 8-wave workgroup per CU is BOTH a split of
// the attention (sequence b, kv head, split) and a worker of the projection:
//
//   1. waves 0-3 request the page ids of their rows and, as soon as those are in, the rows
//      themselves (two stages, kept in the registers the attention body holds them in anyway);
//   2. all eight waves multiply their share of the QKV strips - the 16-row strips of the
//      workgroup's OWN kv head (its G query heads, its key and its value head), dealt over the
//      workgroups of that head - in exactly the shape the stand-alone kernel gives a strip (eight
//      waves interleaved over its tiles, partial sums added in wave order, RMSNorm prologue), so
//      the sums are the same bits;
//   3. the strips' outputs are handed over with write-through stores and one flag per workgroup;
//      a workgroup waits only for the workgroups of its own head (with round-robin placement
//      they share an XCD, which makes the wait cheaper; nothing depends on that placement);
//   4. waves 0-3 run the stand-alone attention body on rows that have long landed (same row
//      assignment, same arithmetic, same split merge: the same bits), waves 4-7 only keep the
//      barriers company.
//
// Every workgroup must be resident at once (grid <= CUs; the launch shape allows one workgroup per
// CU): a wait that outlasts the timeout sets the status word and gives up instead of hanging.
// Flags carry an epoch kept in the workspace, advanced once per launch by the workgroup that
// is last to pass its wait, so nothing is reset and a captured launch can be replayed.
#include "attn_decode_shared.h"

#include <atomic>
#include <type_traits>

#ifdef MV_QA_STAMP
namespace mv {
// Phase stamps (build variant for tools/stamp_qkv_attn.py): the constant 100 MHz counter at up to
// sixteen points of every workgroup, written by its first thread.
static __device__ unsigned long long g_qstamp[16 * 1024];
}  // namespace mv
#define MV_QSTAMP(slot)                                                                  \
    do {                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                               \
        if (threadIdx.x == 0 && blockIdx.x < 1024)                                       \
            mv::g_qstamp[16 * blockIdx.x + (slot)] = __builtin_amdgcn_s_memrealtime();   \
        __builtin_amdgcn_sched_barrier(0);                                               \
    } while (0)
#else
#define MV_QSTAMP(slot) do {} while (0)
#endif

namespace mv {

int dot_form_ksplit(int m, int nt, int kt);   // wgemm.hip: how the stand-alone projection would be cut
int dot_form_waves(int m, int nt, int kt);

// workspace words (uint32): epoch | passed counter | flags, each on 256-byte lines of its own
constexpr int kQaEpoch = 0, kQaPassed = 64, kQaFlags = 128;
constexpr int kQaMaxWgs = 1024;
constexpr size_t kQaBytes = (size_t)(kQaFlags + kQaMaxWgs) * sizeof(unsigned int);

struct QkvAttnArgs {
    // projection: qkv[m, :] = bf16(bf16(rmsnorm(x[m, :]) * ln_w) @ W^T) (+ bias, rounded again)
    const uint16_t* x;
    int64_t ldx;
    const uint16_t* ln_w;
    float eps;
    const uint4* w;
    const uint16_t* bias;
    uint16_t* qkv;
    int64_t ld_qkv;
    int m, k, kt;
    unsigned int* sync;
    int* status;
    unsigned long long timeout_ticks;
    AttnArgs at;   // q / k_new / v_new point into qkv
};

struct QaUnit {
    int strip, ntile, nchunk, last;
    const uint4* wbase;
};

template <int D, int G, int M>
__global__ __launch_bounds__(512) void qkv_attn_kernel(QkvAttnArgs qa) {
    constexpr int GW = 8;                    // waves of the projection (all of the workgroup)
    // weight tiles per stage and wave: twice the stand-alone kernel's (the registers of the attention
    // phase are idle meanwhile, and half the workgroups stream two strips one after the other; the
    // order of a wave's sums does not depend on it)
    constexpr int U = (M <= 2) ? 8 : 4;
    constexpr int kAW = 4;                   // waves of the attention body
    constexpr int kAC = 4;                   // row loads per wave per stage
    constexpr int LPR = D / 8, RPI = 64 / LPR, NS = kAW * RPI;
    constexpr int CR = NS * kAC;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const AttnArgs& a = qa.at;

    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool attn_wave = wid < kAW;
    MV_QSTAMP(0);

    // ---- who this workgroup is -----------------------------------------------------------
    const int kvh = blockIdx.x % a.hkv;
    const int jm = blockIdx.x / a.hkv;          // member of the head's group
    const int J = gridDim.x / a.hkv;            // members per head = bs * nsplit
    const int b = jm / a.nsplit, split = jm - b * a.nsplit;
    const int bh = b * a.hkv + kvh;
    const bool last_split = (split == a.nsplit - 1);

    // the epoch of this launch: one more than the word the last launch left (written through at its
    // end, so a plain - scalar - load across the kernel boundary sees it); nothing waits for it here
    const unsigned int e = qa.sync[kQaEpoch] + 1u;

    // ---- attention, part one: everything that does not need q, k, v -------------------------
    const int lr = lane % LPR, tg = lane / LPR;
    const int d0 = lr * 8;
    const int pos = a.seqlens[b];
    const int32_t* bt = a.block_table + (int64_t)b * a.bt_stride;
    const int64_t head_off = (int64_t)kvh * D + d0;
    const int64_t row_stride = (int64_t)a.hkv * D;
    auto row_of = [&](int c, int i) { return (((c * a.nsplit + split) * kAC + i) * kAW + wid) * RPI + tg; };
    auto load_pages = [&](int (&pg)[kAC], int c) {
#pragma unroll
        for (int i = 0; i < kAC; ++i) pg[i] = bt[min(row_of(c, i) >> a.page_shift, a.bt_stride - 1)];
    };
    // Waves 4-7 run the same instruction stream with every row address replaced by a line that is
    // always cache-hot (their lanes' 16 bytes of x): a branch around these requests would make the
    // compiler's wait for x, further down, cover them too (it counts requests per program path),
    // and the input transform of every workgroup would sit behind 16.8 MB of cold rows.
    const uint16_t* hot = qa.x + (threadIdx.x & 255) * 8;
    auto load_rows = [&](uint4(&kb)[kAC], uint4(&vb)[kAC], bool(&ok)[kAC], const int (&pg)[kAC], int c) {
#pragma unroll
        for (int i = 0; i < kAC; ++i) {
            const int t = row_of(c, i);
            const bool page_ok = pg[i] >= 0 && pg[i] < a.num_pages;
            ok[i] = (t < pos) && page_ok;
            const int64_t off =
                (((int64_t)(page_ok ? pg[i] : 0) << a.page_shift) + (t & (a.page - 1))) * row_stride + head_off;
            kb[i] = ld16(attn_wave ? a.k_cache + off : hot);
            vb[i] = ld16(attn_wave ? a.v_cache + off : hot);
        }
    };
    int pgA[kAC], pgB[kAC];
    load_pages(pgA, 0);
    load_pages(pgB, 1);

    // ---- projection: x and the norm weight are requested first (memory answers a wave in order) --
    const int r = lane & 15, q = lane >> 4;
    uint16_t* xs = reinterpret_cast<uint16_t*>(smem);
    const int zero_off = M * qa.k * 2;
    float* red = reinterpret_cast<float*>(smem + zero_off + 16);
    const int kvec = qa.k >> 3;
    constexpr int kPT = 256;
    constexpr int PC = (M == 1) ? 4 : (M == 2 ? 2 : 1);
    uint4 pxe[M][PC], pwe[PC];
#pragma unroll
    for (int j = 0; j < PC; ++j) {
        if (j * kPT < kvec) {   // uniform
            const int c = min((int)(threadIdx.x & (kPT - 1)) + j * kPT, kvec - 1);
#pragma unroll
            for (int m = 0; m < M; ++m) pxe[m][j] = ld16(qa.x + (int64_t)m * qa.ldx + c * 8);
            pwe[j] = ld16(qa.ln_w + c * 8);
        }
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);

    // strips of this head: its G query heads, its key head, its value head
    constexpr int SQ = G * D / 16, SK = D / 16, SH = SQ + 2 * SK;
    auto setup = [&](int s) {
        QaUnit c;
        c.strip = s < SQ ? kvh * SQ + s
                         : (s < SQ + SK ? a.hq * SK + kvh * SK + (s - SQ)
                                        : (a.hq + a.hkv) * SK + kvh * SK + (s - SQ - SK));
        c.ntile = (qa.kt - wid + GW - 1) / GW;   // wave w owns tiles w, w + 8, ... (may be <= 0)
        c.nchunk = (c.ntile + U - 1) / U;
        c.last = max(c.ntile - 1, 0);
        c.wbase = qa.w + ((int64_t)c.strip * qa.kt + (c.ntile > 0 ? wid : 0)) * 64 + lane;
        return c;
    };
    constexpr int wstride = GW * 64;
    auto load = [&](uint4(&buf)[U], const QaUnit& c, int chunk) {
#pragma unroll
        for (int i = 0; i < U; ++i) {
            const int t = min(chunk * U + i, c.last);
            buf[i] = ld16_nt(c.wbase + (int64_t)t * wstride);
        }
    };
    auto compute = [&](const uint4(&buf)[U], const QaUnit& c, int chunk, float(&acc)[M]) {
#pragma unroll
        for (int i = 0; i < U; ++i) {
            const int t = chunk * U + i;
            const int xoff = (t < c.ntile) ? ((wid + t * GW) * 32 + q * 8) * 2 : -1;
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const int off = (xoff < 0) ? zero_off : m * qa.k * 2 + xoff;
                const uint4 xv = *reinterpret_cast<const uint4*>(smem + off);
                acc[m] = dot2_bf16(buf[i].x, xv.x, acc[m]);
                acc[m] = dot2_bf16(buf[i].y, xv.y, acc[m]);
                acc[m] = dot2_bf16(buf[i].z, xv.z, acc[m]);
                acc[m] = dot2_bf16(buf[i].w, xv.w, acc[m]);
            }
        }
    };

    // member jm of the head takes strip jm and, where there is one, strip jm + J (the launcher
    // only admits shapes with one or two strips per member)
    const QaUnit cur = setup(jm);
    uint4 bufA[U], bufB[U];
#pragma unroll
    for (int i = 0; i < U; ++i) {
        const int t = min(i, cur.last);
        bufA[i] = ld16_nt(cur.wbase + (int64_t)t * wstride);
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("" ::: "memory");

    // ---- x -> LDS as rmsnorm(x) * ln_w: chunk assignment and summation order of the stand-alone
    // norm kernels and of the stand-alone projection's prologue ---------------------------------
    {
        float ss[M];
#pragma unroll
        for (int m = 0; m < M; ++m) ss[m] = 0.f;
        auto absorb_x = [&](int m, int c, const uint4& xv) {
            const uint32_t x4[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v0 = bf16_lo(x4[j]), v1 = bf16_hi(x4[j]);
                ss[m] += v0 * v0;
                ss[m] += v1 * v1;
            }
            st16(xs + m * qa.k + c * 8, xv);
        };
        if (threadIdx.x < kPT) {
#pragma unroll
            for (int m = 0; m < M; ++m) {
#pragma unroll
                for (int j = 0; j < PC; ++j) {
                    const int c = threadIdx.x + j * kPT;
                    if (c < kvec) absorb_x(m, c, pxe[m][j]);
                }
                for (int c = threadIdx.x + PC * kPT; c < kvec; c += kPT)
                    absorb_x(m, c, ld16(qa.x + (int64_t)m * qa.ldx + c * 8));
            }
        }
        if (threadIdx.x == 0) *reinterpret_cast<uint4*>(smem + zero_off) = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const float v = wave_sum(ss[m]);
            if (lane == 0 && wid < kPT / 64) red[m * (kPT / 64) + wid] = v;
        }
        __syncthreads();
        float rrms[M];
#pragma unroll
        for (int m = 0; m < M; ++m) {
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < kPT / 64; ++i) t += red[m * (kPT / 64) + i];
            rrms[m] = 1.0f / sqrtf(t / (float)qa.k + qa.eps);
        }
        auto scale = [&](int m, int c, const uint4& wv) {
            uint4* px = reinterpret_cast<uint4*>(xs + m * qa.k + c * 8);
            const uint4 sv = *px;
            const uint32_t s4[4] = {sv.x, sv.y, sv.z, sv.w}, w4[4] = {wv.x, wv.y, wv.z, wv.w};
            uint32_t o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                o[j] = pack_bf16x2((bf16_lo(s4[j]) * rrms[m]) * bf16_lo(w4[j]),
                                   (bf16_hi(s4[j]) * rrms[m]) * bf16_hi(w4[j]));
            *px = make_uint4(o[0], o[1], o[2], o[3]);
        };
        if (threadIdx.x < kPT) {
#pragma unroll
            for (int j = 0; j < PC; ++j) {
                const int c = threadIdx.x + j * kPT;
                if (c < kvec) {
#pragma unroll
                    for (int m = 0; m < M; ++m) scale(m, c, pwe[j]);
                }
            }
            for (int c = threadIdx.x + PC * kPT; c < kvec; c += kPT) {
                const uint4 wv = ld16(qa.ln_w + c * 8);
#pragma unroll
                for (int m = 0; m < M; ++m) scale(m, c, wv);
            }
        }
    }
    __syncthreads();
    MV_QSTAMP(1);   // x staged
#pragma unroll
    for (int i = U; i < 2 * U; ++i) {
        const int t = min(i, cur.last);
        bufB[i - U] = ld16_nt(cur.wbase + (int64_t)t * wstride);
        __builtin_amdgcn_sched_barrier(0);
    }
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
        qa.qkv, 0, (int)(((int64_t)(M - 1) * qa.ld_qkv + (int64_t)(a.hq + 2 * a.hkv) * D) * 2), 0x00020000);
    // ---- the rows: declared here, requested (issue_rows) once this workgroup's LAST strip is
    // multiplied.  Weights and rows share HBM; the strips are what the head's other workgroups wait
    // for, the rows are needed only when q has come back, so the rows queue behind the weights.
    // Measured on MI355X (tools/stamp_qkv_attn.py, 8B head shape, 4096 rows, cold caches): the
    // last strip of a workgroup is multiplied 11.6 us after the launch starts when the rows are
    // requested only after the hand-over (-DMV_QA_ROWS_LATE), 15-17 us with the rows requested here or
    // anywhere before - they are 16.8 MB in front of or among 50 MB of weights and cost their share
    // of the pipe whenever they go - and the launch ends at 24 and 26-27 us against 23 us for the
    // two stand-alone launches.  That is why the runner leaves this launch switched off. ----------
    uint4 kA[kAC], vA[kAC], kB[kAC], vB[kAC];
    bool okA[kAC], okB[kAC];
    uint4 cosv = make_uint4(0, 0, 0, 0), sinv = cosv;
    const bool rotate = (a.rot_cos != nullptr);
    const bool first_half = d0 < D / 2;
    const int dpart = first_half ? d0 + D / 2 : d0 - D / 2;
    auto issue_rows = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        load_rows(kA, vA, okA, pgA, 0);
        load_pages(pgA, 2);
        load_rows(kB, vB, okB, pgB, 1);
        load_pages(pgB, 3);
        // (uniform: one table or none for the whole launch)
        const int rp = min(max(pos, 0), max(a.rot_max_pos, 1) - 1);
        const int64_t ro = (int64_t)rp * (D / 2) + (first_half ? d0 : d0 - D / 2);
        cosv = ld16(rotate ? a.rot_cos + ro : hot);
        sinv = ld16(rotate ? a.rot_sin + ro : hot);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };

    constexpr int kPast = 1 << 24;
    // One strip: its sums, and - with another strip to follow - that strip's first two stages
    // requested on the way.  Wave 4 (which has no rows outstanding, so its wait for its own stores
    // is short) forms the strip's 16 outputs per row and stores them, two to a dword, write-through.
    auto run_unit = [&](const QaUnit& cu, const QaUnit& nx, auto more_c) {
        constexpr bool more = decltype(more_c)::value;
        float acc[M];
#pragma unroll
        for (int m = 0; m < M; ++m) acc[m] = 0.f;
        const int npair = (cu.nchunk + 1) >> 1;
        for (int p = 0; p < npair - 1; ++p) {
            compute(bufA, cu, 2 * p, acc);
            load(bufA, cu, 2 * p + 2);
            compute(bufB, cu, 2 * p + 1, acc);
            load(bufB, cu, 2 * p + 3);
        }
        const int cl = npair > 0 ? 2 * (npair - 1) : kPast;
        if constexpr (more) {
            compute(bufA, cu, cl, acc);
            load(bufA, nx, 0);
            compute(bufB, cu, cl + 1, acc);
            load(bufB, nx, 1);
        } else {
            // The rows go out here, behind the last tiles of the last strip and behind the loop
            // above: requested in front of a loop that also requests, they would be "the oldest
            // requests" on one path into these waits and absent on the other, and the compiler
            // would then wait for everything - rows included - before the last tile's product.
            compute(bufA, cu, cl, acc);
            compute(bufB, cu, cl + 1, acc);
#ifndef MV_QA_ROWS_LATE
            issue_rows();
#endif
        }
#pragma unroll
        for (int m = 0; m < M; ++m) {
            acc[m] += __shfl_xor(acc[m], 16, 64);
            acc[m] += __shfl_xor(acc[m], 32, 64);
        }
        if (lane < 16) {
#pragma unroll
            for (int m = 0; m < M; ++m) red[(wid * M + m) * 16 + r] = acc[m];
        }
        __syncthreads();
        if (wid == 4) {
            const int m = min(lane >> 4, M - 1), rr = lane & 15;
            float s = 0.f;
#pragma unroll
            for (int w8 = 0; w8 < GW; ++w8) s += red[(w8 * M + m) * 16 + rr];
            const int n = cu.strip * 16 + rr;
            float t = round_bf16(s);
            if (qa.bias) t = round_bf16(t + bf16_bits_to_f32(qa.bias[n]));
            const uint32_t bits = f32_to_bf16_bits(t);
            const uint32_t nb = __shfl_down(bits, 1, 64);
            if (lane < 16 * M && !(rr & 1))
                __builtin_amdgcn_raw_buffer_store_b32(bits | (nb << 16), yrs,
                                                      (int)(((int64_t)m * qa.ld_qkv + n) * 2), 0, 16 /*sc1*/);
        }
        if constexpr (more) __syncthreads();   // `red` is reused by the next strip
    };
    // (two straight-line paths, so that the compiler's request counts stay exact in each)
    if (jm + J < SH) {
        const QaUnit nxt = setup(jm + J);
        run_unit(cur, nxt, std::true_type{});
        run_unit(nxt, nxt, std::false_type{});
    } else {
        run_unit(cur, cur, std::false_type{});
    }

    // ---- hand-over: this workgroup's outputs are in memory, then its flag goes up; the storing
    // wave does both, and then watches the flags of the head's workgroups while the others sleep
    // at the barrier ---------------------------------------------------------------------------
    MV_QSTAMP(2);   // strips done
    if (wid == 4) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0)
            __hip_atomic_store(qa.sync + kQaFlags + blockIdx.x, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (int iter = 1;; ++iter) {
            bool there = true;
            for (int i = lane; i < J; i += 64) {
                const unsigned int f = __hip_atomic_load(qa.sync + kQaFlags + kvh + a.hkv * i, __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_AGENT);
                there = there && (int)(f - e) >= 0;
            }
            if (__all(there)) break;
            if ((iter & 31) == 0 && __builtin_amdgcn_s_memrealtime() - t0 > qa.timeout_ticks) {
                if (lane == 0) __hip_atomic_store(qa.status, 1 + (int)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
#ifdef MV_QA_ROWS_LATE
    issue_rows();   // timing experiment: rows requested only once q is ready
#endif
    MV_QSTAMP(4);   // head's flags all up
    // the epoch moves on once every workgroup is past its wait (nobody compares against it again)
    unsigned int passed_before = 0;
    if (threadIdx.x == 0)
        passed_before = __hip_atomic_fetch_add(qa.sync + kQaPassed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    // ---- attention, part two: q, k, v of this head (write-through data: L1-bypassing loads) ------
    float* lds = reinterpret_cast<float*>(smem);
    float* st_m = lds;
    float* st_l = lds + NS * G;
    float* st_acc = lds + 2 * NS * G;
    if (attn_wave) {
        const __amdgpu_buffer_rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc(
            qa.qkv + (int64_t)b * qa.ld_qkv, 0, (a.hq + 2 * a.hkv) * D * 2, 0x00020000);
        auto ldq = [&](int col) {
            const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(qrs, col * 2, 0, 16 /*sc1*/);
            return make_uint4(v.x, v.y, v.z, v.w);
        };
        uint4 qx[G], qo[G], kx, ko, vnew;
        ko = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            qx[g] = ldq((kvh * G + g) * D + d0);
            qo[g] = rotate ? ldq((kvh * G + g) * D + dpart) : qx[g];
        }
        kx = ldq((a.hq + kvh) * D + d0);
        if (rotate) ko = ldq((a.hq + kvh) * D + dpart);
        vnew = ldq((a.hq + a.hkv + kvh) * D + d0);

#ifdef MV_QA_STAMP
        if (vnew.x != 0x7fc07fc1u) MV_QSTAMP(5);   // q, k, v landed
#endif
        const int seq_chunks = (max(pos, 0) + CR - 1) / CR;
        const int nchunk = seq_chunks > split ? (seq_chunks - split + a.nsplit - 1) / a.nsplit : 0;

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

        float m[G], l[G], acc[G][8];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            m[g] = kNegBig;
            l[g] = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[g][j] = 0.f;
        }
        auto absorb = [&](const uint4(&kb)[kAC], const uint4(&vb)[kAC], const bool(&ok)[kAC]) {
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

        int c = 0;
        for (; c + 3 < nchunk; c += 2) {
            absorb(kA, vA, okA);
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

        if (last_split && wid == 0) {
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
    }
    MV_QSTAMP(6);   // rows absorbed, states in LDS
    if (threadIdx.x == 0 && passed_before == gridDim.x - 1u) {
        __hip_atomic_store(qa.sync + kQaPassed, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(qa.sync + kQaEpoch, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // (threads past the attention waves take no part in the per-head work of the merge)
    finish_states<D, G, NS, kAW>(a, lds, b, kvh, bh, split, attn_wave ? wid : 4096, lane);
    MV_QSTAMP(7);   // end (last arrivers: output written)
}

template <int D, int G, int M>
static int launch_qkv_attn(const QkvAttnArgs& qa, int grid, hipStream_t s) {
    constexpr int NS = 4 * (64 / (D / 8));
    const size_t lds_attn = (size_t)(2 * NS * G + NS * G * D) * sizeof(float);
    const size_t lds_proj = (size_t)M * qa.k * 2 + 16 + (size_t)8 * M * 16 * 4;
    const size_t lds = lds_attn > lds_proj ? lds_attn : lds_proj;
    if (lds > 159 * 1024) return MV_ERR_UNSUPPORTED;
    static LdsOptIn opted;
    if (int rc = ensure_dynamic_lds((const void*)qkv_attn_kernel<D, G, M>, lds, opted)) return rc;
    hipLaunchKernelGGL((qkv_attn_kernel<D, G, M>), dim3(grid), dim3(512), lds, s, qa);
    return check_launch();
}

template <int D, int G>
static int launch_qkv_attn_m(const QkvAttnArgs& qa, int grid, hipStream_t s) {
    switch (qa.m) {
        case 1: return launch_qkv_attn<D, G, 1>(qa, grid, s);
        case 2: return launch_qkv_attn<D, G, 2>(qa, grid, s);
        case 4: return launch_qkv_attn<D, G, 4>(qa, grid, s);
        default: return MV_ERR_UNSUPPORTED;
    }
}

}  // namespace mv

#ifdef MV_QA_STAMP
extern "C" int mv_debug_read_qa_stamps(void* dst, size_t bytes) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(mv::g_qstamp), bytes) == hipSuccess ? 0 : -3;
}
#endif

extern "C" {

size_t mv_qkv_attn_workspace_bytes(void) { return mv::kQaBytes; }

int mv_qkv_attn_decode(const void* x, int64_t ldx, const void* ln_w, float eps, const void* w_tiles,
                       const void* bias, void* qkv, int64_t ld_qkv, int m, int n, int k, void* k_cache,
                       void* v_cache, const void* rot_cos, const void* rot_sin, int rot_max_pos,
                       const int32_t* cache_seqlens, const int32_t* block_table, int bt_stride, void* out,
                       void* attn_workspace, size_t attn_workspace_bytes, void* sync_workspace,
                       size_t sync_workspace_bytes, int* status_word, int hq, int hkv, int d, int page,
                       int num_pages, float scale, int num_splits, mv_stream_t stream) {
    using namespace mv;
    if (!x || !ln_w || !w_tiles || !qkv || !k_cache || !v_cache || !cache_seqlens || !block_table || !out ||
        !sync_workspace || !status_word)
        return MV_ERR_INVALID_ARG;
    if (m <= 0 || n <= 0 || k <= 0 || hq <= 0 || hkv <= 0 || page <= 0 || num_pages <= 0 || bt_stride <= 0 ||
        num_splits <= 0)
        return MV_ERR_INVALID_ARG;
    if ((rot_cos == nullptr) != (rot_sin == nullptr) || (rot_cos && rot_max_pos <= 0)) return MV_ERR_INVALID_ARG;
    if (!aligned16(x) || !aligned16(ln_w) || !aligned16(w_tiles) || !aligned16(qkv) || !aligned16(k_cache) ||
        !aligned16(v_cache) || !aligned16(out) || !aligned16(sync_workspace) || (ldx % 8) || ldx < k ||
        (ld_qkv % 8) || ld_qkv < n || (bias && !aligned16(bias)) ||
        (rot_cos && (!aligned16(rot_cos) || !aligned16(rot_sin))))
        return MV_ERR_INVALID_ARG;
    if (sync_workspace_bytes < kQaBytes) return MV_ERR_WORKSPACE;
    // ---- what is fused: shapes whose two stand-alone launches this reproduces bit for bit ---------
    if (m > 4 || m == 3 || (d != 64 && d != 128) || (hq % hkv) || (page & (page - 1)) || (k % 32) ||
        n != (hq + 2 * hkv) * d)
        return MV_ERR_UNSUPPORTED;
    const int g = hq / hkv;
    if (mv_paged_attn_decode_form(hq, hkv, d, page) != 0) return MV_ERR_UNSUPPORTED;   // matrix-core form
    const int nt = n / 16, kt = k / 32;
    if (dot_form_waves(m, nt, kt) != 8 || dot_form_ksplit(m, nt, kt) != 1) return MV_ERR_UNSUPPORTED;
    const int grid = m * hkv * num_splits;
    const int cus = num_cus();
    // every workgroup resident at once, most CUs busy, and one or two strips of the head for each of
    // the head's workgroups
    if (num_splits < 2 || num_splits > 128 || grid > cus || 2 * grid <= cus || grid > kQaMaxWgs ||
        m * hkv > kMaxArrive || (g + 2) * d / 16 < m * num_splits || (g + 2) * d / 16 > 2 * m * num_splits)
        return MV_ERR_UNSUPPORTED;
    const size_t need = mv_paged_attn_decode_workspace_bytes(m, hq, d, num_splits);
    if (!attn_workspace || attn_workspace_bytes < need || !aligned16(attn_workspace)) return MV_ERR_WORKSPACE;

    QkvAttnArgs qa = {};
    qa.x = (const uint16_t*)x;
    qa.ldx = ldx;
    qa.ln_w = (const uint16_t*)ln_w;
    qa.eps = eps;
    qa.w = (const uint4*)w_tiles;
    qa.bias = (const uint16_t*)bias;
    qa.qkv = (uint16_t*)qkv;
    qa.ld_qkv = ld_qkv;
    qa.m = m;
    qa.k = k;
    qa.kt = kt;
    qa.sync = (unsigned int*)sync_workspace;
    qa.status = status_word;
    qa.timeout_ticks = 200000000ull;   // 2 s of the 100 MHz counter
    AttnArgs& a = qa.at;
    a.q = qa.qkv;
    a.q_stride_b = ld_qkv;
    a.k_new = qa.qkv + (int64_t)hq * d;
    a.v_new = qa.qkv + (int64_t)(hq + hkv) * d;
    a.kv_stride_b = ld_qkv;
    a.k_cache = (uint16_t*)k_cache;
    a.v_cache = (uint16_t*)v_cache;
    a.rot_cos = (const uint16_t*)rot_cos;
    a.rot_sin = (const uint16_t*)rot_sin;
    a.rot_max_pos = rot_max_pos;
    a.seqlens = cache_seqlens;
    a.block_table = block_table;
    a.bt_stride = bt_stride;
    a.out = (uint16_t*)out;
    a.bs = m;
    a.hq = hq;
    a.hkv = hkv;
    a.page = page;
    a.page_shift = __builtin_ctz((unsigned)page);
    a.num_pages = num_pages;
    a.scale_log2 = scale * 1.4426950408889634f;
    a.nsplit = num_splits;
    const size_t rows = (size_t)m * hq * num_splits;
    a.part_lse = (float*)((char*)attn_workspace + kArriveBytes);
    a.part_o = (float*)((char*)attn_workspace + kArriveBytes + (rows * sizeof(float) + 255) / 256 * 256);
    a.arrive = (unsigned int*)attn_workspace;
    a.slabs = nullptr;
    a.slab_bias = nullptr;
    a.slab_ks = 0;
    a.slab_n = n;
    hipStream_t s = (hipStream_t)stream;
    if (d == 128) {
        switch (g) {
            case 1: return launch_qkv_attn_m<128, 1>(qa, grid, s);
            case 2: return launch_qkv_attn_m<128, 2>(qa, grid, s);
            case 4: return launch_qkv_attn_m<128, 4>(qa, grid, s);
            default: return MV_ERR_UNSUPPORTED;
        }
    }
    switch (g) {
        case 1: return launch_qkv_attn_m<64, 1>(qa, grid, s);
        case 4: return launch_qkv_attn_m<64, 4>(qa, grid, s);
        default: return MV_ERR_UNSUPPORTED;   // (eight heads of 64 on one kv head do not fit the register file here)
    }
}

}  // extern "C"
