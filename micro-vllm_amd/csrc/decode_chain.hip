// Up to four dependent weight-streaming projections of a decode step (1 <= M <= 4 rows) in ONE
// launch: the O projection, gate/up (+ add + RMSNorm in front, SwiGLU behind), down (+ residual)
// and the next block's QKV projection (RMSNorm in front) - or the LM head - separated by grid
// barriers instead of launch boundaries.
//
// Why: every one of these projections is a stream of weights from HBM that takes 7-37 us, and a
// launch boundary drains that stream for about 4 us (last workgroup out, next kernel in, first
// touch of the previous result, first tiles back from HBM).  Weights do not depend on the previous
// projection's result, so here every wave requests the first two stages of the NEXT projection's
// tiles (8 KiB per wave, 32 MiB over the chip) before it waits at the barrier: HBM keeps streaming
// while the result of the current projection becomes visible.
//
// One workgroup of 16 waves per CU; all workgroups are resident at once (that is what makes a grid
// barrier legal), and a wait that outlasts the timeout sets the status word and gives up instead
// of hanging.  A projection runs in one of the two shapes the stand-alone kernel
// (wgemm_dot_kernel) uses for it - 16 waves on one strip of 16 output rows, or two strips at a time
// with 8 waves each - with the same tile-to-wave assignment and the same fixed-order reductions,
// so the results are bit-identical to the separate launches.
//
// Hand-over between projections: outputs are written with write-through (sc1) stores; a
// workgroup arrives at the barrier after its stores are acknowledged; after the barrier whatever
// another workgroup wrote in this launch (x, the residual sum) is read with sc1 loads, which go
// past this XCD's L2.  (An agent-scope acquire fence in front of plain loads does the same job
// and was measured at 26 us per barrier: it invalidates the whole L2.)  The barrier has no read-modify-write in
// it: every workgroup owns a flag into which it stores the number of the barrier it has reached
// (a write-through store), and one wave of every workgroup watches all flags, four per lane.  (One shared
// arrival counter was measured at 27 us per barrier on MI355X: 256 agent-scope atomic adds on one
// address queue up behind each other at about 100 ns apiece.)  Barrier numbers only grow; each
// launch reads the number it starts from out of a word that workgroup 0 advances after the
// launch's last barrier, so launches (which must be stream-ordered when they share a workspace)
// need no reset and a captured graph can be replayed.
#include "common.h"

namespace mv {

constexpr int kCW = 16;              // waves per workgroup
constexpr int kCT = kCW * kWave;     // 1024 threads
constexpr int kChainMaxPhases = 4;
constexpr int kCPT = 256;            // threads of the input transform (as in the stand-alone kernels)

enum { kCProNone = 0, kCProNorm = 1, kCProAddNorm = 2 };

struct ChainPhase {
    const uint4* w;
    int nt, kt, k, n;
    int halves;                 // 2: two strips at a time, 8 waves each; 1: one strip, 16 waves
    const uint16_t* x;
    int64_t ldx;
    int prologue, var_from_rounded;
    const uint16_t* add_r;
    int64_t ld_add_r;
    const uint16_t* ln_w;
    float eps;
    uint16_t* sum_out;
    int64_t ld_sum_out;
    const uint16_t* bias;
    const uint16_t* residual;
    int64_t ld_res;
    uint16_t* y;
    int64_t ldy;
    int flags;
};

struct ChainArgs {
    ChainPhase ph[kChainMaxPhases];
    int nphase;
    int slab_off;                // byte offset of the per-wave tile slabs in LDS (0 = none: no deep prefetch)
    unsigned int* flags;         // [workgroup]: number of the last barrier it reached
    unsigned int* base;          // barrier number the current launch starts from
    int* status;                 // set non-zero when a wait timed out
    unsigned long long timeout_ticks;
};
constexpr int kMaxChainWgs = 512;

#ifdef MV_CHAIN_STAMP
// Phase stamps (build variant for tools/stamp_chain.py): the constant 100 MHz counter at four points
// of every phase of every workgroup of the LAST launch, written by the workgroup's first thread.
static __device__ unsigned long long g_cstamp[512 * 4 * 4];
#define MV_CSTAMP(phase, slot)                                                                  \
    do {                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                      \
        if (threadIdx.x == 0) g_cstamp[(blockIdx.x * 4 + (phase)) * 4 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
        __builtin_amdgcn_sched_barrier(0);                                                      \
    } while (0)
#else
#define MV_CSTAMP(phase, slot) do {} while (0)
#endif

struct ChainUnit {
    int valid;                  // (an int: a bool would leave padding that struct copies drag through scratch)
    int strip, ntile, last, wv, gw;
    const uint4* wbase;
};

template <int M>
__global__ __launch_bounds__(kCT) void decode_chain_kernel(ChainArgs a) {
    constexpr int U = (M <= 2) ? 4 : 2;   // tiles per stage, as the stand-alone kernel
    constexpr bool kDeep = (M == 1);      // four stages across a barrier (LDS has room for the slabs at one row)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // layout: 16 zero bytes | red [kCW][M][16] fp32 | epi [2 halves][2][M][16] fp32 | x rows [M][K] bf16
    float* red = reinterpret_cast<float*>(smem + 16);
    float* epi = red + kCW * M * 16;     // bias and residual of a phase's first unit, fetched while x is staged
    constexpr int kXsOff = 16 + kCW * M * 16 * 4 + 2 * 2 * M * 16 * 4;
    uint16_t* xs = reinterpret_cast<uint16_t*>(smem + kXsOff);
    __shared__ unsigned int s_base;

    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int nwg = gridDim.x;

    if (threadIdx.x == 0) {
        s_base = __hip_atomic_load(a.base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *reinterpret_cast<uint4*>(smem) = make_uint4(0, 0, 0, 0);
    }

    auto setup = [&](const ChainPhase& P, int step) {
        ChainUnit c;
        c.gw = P.halves == 2 ? 8 : 16;
        const int half = P.halves == 2 ? (wid >> 3) : 0;
        c.wv = P.halves == 2 ? (wid & 7) : wid;
        c.strip = step * (P.halves * nwg) + half * nwg + (int)blockIdx.x;
        c.valid = c.strip < P.nt;
        c.ntile = (P.kt - c.wv + c.gw - 1) / c.gw;
        c.last = max(c.ntile - 1, 0);
        // a unit past the end (or a wave without tiles) still issues loads: point it at valid tiles
        c.wbase = P.w + ((int64_t)min(c.strip, P.nt - 1) * P.kt + (c.ntile > 0 ? c.wv : 0)) * 64 + lane;
        return c;
    };
    auto load = [&](uint4(&buf)[U], const ChainUnit& c, int chunk) {
#pragma unroll
        for (int i = 0; i < U; ++i) {
            const int t = min(chunk * U + i, c.last);
            buf[i] = ld16_nt(c.wbase + (int64_t)t * (c.gw * 64));
        }
    };
    auto compute_tile = [&](const uint4& w, const ChainUnit& c, int k, int t, float(&acc)[M]) {
        const int xoff = (t < c.ntile) ? ((c.wv + t * c.gw) * 32 + q * 8) * 2 : -1;
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const int off = (xoff < 0) ? 0 : kXsOff + m * k * 2 + xoff;
            const uint4 xv = *reinterpret_cast<const uint4*>(smem + off);
            acc[m] = dot2_bf16(w.x, xv.x, acc[m]);
            acc[m] = dot2_bf16(w.y, xv.y, acc[m]);
            acc[m] = dot2_bf16(w.z, xv.z, acc[m]);
            acc[m] = dot2_bf16(w.w, xv.w, acc[m]);
        }
    };
    auto compute = [&](const uint4(&buf)[U], const ChainUnit& c, int k, int chunk, float(&acc)[M]) {
#pragma unroll
        for (int i = 0; i < U; ++i) compute_tile(buf[i], c, k, chunk * U + i, acc);
    };

    // ---- the first projection: its x is requested first (memory requests of a wave return in
    // order, and x is what the wave needs first), the first two stages of its tiles right behind ----
    constexpr int XE = 2;   // early 16-byte chunks of x per thread
    uint4 x_early[XE];
    const bool x0_early = a.ph[0].prologue == kCProNone && M * (a.ph[0].k >> 3) <= XE * kCT;
    if (x0_early) {
        const int kv0 = a.ph[0].k >> 3;
#pragma unroll
        for (int j = 0; j < XE; ++j) {
            const int c = min((int)threadIdx.x + j * kCT, M * kv0 - 1);
            const int m = c / kv0, cv = c - m * kv0;
            x_early[j] = ld16(a.ph[0].x + (int64_t)m * a.ph[0].ldx + cv * 8);
        }
    }
    ChainUnit cur = setup(a.ph[0], 0);
    uint4 bufA[U], bufB[U];
    load(bufA, cur, 0);
    load(bufB, cur, 1);
    const bool DEEP = kDeep && a.slab_off > 0;
    uint4* slab = reinterpret_cast<uint4*>(smem + a.slab_off) + (size_t)wid * (2 * U * 64) + lane;

    int barriers_done = 0;
    for (int p = 0; p < a.nphase; ++p) {
        const ChainPhase& P = a.ph[p];
        const int kvec = P.k >> 3;
        if (p > 0) {
            // ---- grid barrier: every workgroup's outputs of projection p-1 are in memory ----------
            // (this workgroup's stores were acknowledged, and the arrival made, at the end of p-1)
            if (wid == 0) {   // one wave watches all flags (four per lane and load); the others sleep at the barrier below
                const unsigned int gen = s_base + (unsigned int)barriers_done;
                const __amdgpu_buffer_rsrc_t frs =
                    __builtin_amdgcn_make_buffer_rsrc(a.flags, 0, nwg * 4, 0x00020000);
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                for (int iter = 1;; ++iter) {
                    bool there = true;
                    for (int c = lane; c * 4 < nwg; c += 64) {
                        const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(frs, c * 16, 0, 16 /*sc1*/);
                        const unsigned int f4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (c * 4 + j < nwg) there = there && (int)(f4[j] - gen) >= 0;
                    }
                    if (__all(there)) break;
                    if ((iter & 31) == 0 && __builtin_amdgcn_s_memrealtime() - t0 > a.timeout_ticks) {
                        if (lane == 0) __hip_atomic_store(a.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
            }
            __syncthreads();
        }
        MV_CSTAMP(p, 0);
        // x (and, in the epilogue, the residual) may have been written by other workgroups of this launch
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint16_t*>(P.x), 0, (int)(((int64_t)(M - 1) * P.ldx + P.k) * 2), 0x00020000);
        auto ld_x = [&](int m, int c) {
            const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)(((int64_t)m * P.ldx + c * 8) * 2), 0,
                                                                    16 /*sc1*/);
            return make_uint4(v.x, v.y, v.z, v.w);
        };

        // What the epilogue of this workgroup's first unit adds (bias, residual) is fetched here, in
        // front of x: the residual sum was written by another workgroup of this launch, so it is an
        // sc1 load - a trip to memory that would otherwise sit at the end of the unit.
        {
            const int half0 = P.halves == 2 ? (wid >> 3) : 0;
            const int th0 = (int)threadIdx.x - half0 * (8 * kWave);
            const int strip0 = half0 * nwg + (int)blockIdx.x;
            if (th0 >= 0 && th0 < 16 * M && strip0 < P.nt && !(P.flags & MV_GEMM_SWIGLU)) {
                const int n = strip0 * 16 + (th0 & 15);
                float bv = 0.f, rv = 0.f;
                if (P.bias) bv = bf16_bits_to_f32(P.bias[n]);
                if (P.residual) {
                    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(
                        const_cast<uint16_t*>(P.residual), 0, (int)(((int64_t)(M - 1) * P.ld_res + P.n) * 2),
                        0x00020000);
                    rv = bf16_bits_to_f32(__builtin_amdgcn_raw_buffer_load_b16(
                        rrs, (int)(((int64_t)(th0 >> 4) * P.ld_res + n) * 2), 0, 16 /*sc1*/));
                }
                epi[(half0 * 2 + 0) * M * 16 + th0] = bv;
                epi[(half0 * 2 + 1) * M * 16 + th0] = rv;
            }
        }

        // ---- stage x into LDS, optionally as rmsnorm(x [+ add_r]) * ln_w ---------------------------
        // Same chunk assignment (256 threads, stride 256) and the same reduction order as the
        // stand-alone norm kernels and the stand-alone projection's prologue: bit-identical.
        if (p == 0 && x0_early) {
#pragma unroll
            for (int j = 0; j < XE; ++j) {
                const int c = threadIdx.x + j * kCT;
                if (c < M * kvec) {
                    const int m = c / kvec, cv = c - m * kvec;
                    st16(xs + m * P.k + cv * 8, x_early[j]);
                }
            }
        } else if (P.prologue == kCProNone) {
            for (int c = threadIdx.x; c < M * kvec; c += kCT) {
                const int m = c / kvec, cv = c - m * kvec;
                st16(xs + m * P.k + cv * 8, ld_x(m, cv));
            }
        } else {
            const bool add = P.prologue == kCProAddNorm;
            float ss[M];
#pragma unroll
            for (int m = 0; m < M; ++m) ss[m] = 0.f;
            if (threadIdx.x < kCPT) {
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    for (int c = threadIdx.x; c < kvec; c += kCPT) {
                        const uint4 xv = ld_x(m, c);
                        const uint32_t x4[4] = {xv.x, xv.y, xv.z, xv.w};
                        if (add) {
                            const uint4 rv = ld16(P.add_r + (int64_t)m * P.ld_add_r + c * 8);
                            const uint32_t r4[4] = {rv.x, rv.y, rv.z, rv.w};
                            uint32_t o[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                float s0 = bf16_lo(x4[j]) + bf16_lo(r4[j]);
                                float s1 = bf16_hi(x4[j]) + bf16_hi(r4[j]);
                                o[j] = pack_bf16x2(s0, s1);
                                if (P.var_from_rounded) {
                                    s0 = bf16_lo(o[j]);
                                    s1 = bf16_hi(o[j]);
                                }
                                ss[m] += s0 * s0;
                                ss[m] += s1 * s1;
                            }
                            const uint4 ov = make_uint4(o[0], o[1], o[2], o[3]);
                            st16(xs + m * P.k + c * 8, ov);
                            if (blockIdx.x == 0) {   // the sum is a result other workgroups read later
                                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                                    P.sum_out + (int64_t)m * P.ld_sum_out, 0, P.k * 2, 0x00020000);
                                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, ov), rs, c * 16, 0,
                                                                       16 /*sc1*/);
                            }
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const float v0 = bf16_lo(x4[j]), v1 = bf16_hi(x4[j]);
                                ss[m] += v0 * v0;
                                ss[m] += v1 * v1;
                            }
                            st16(xs + m * P.k + c * 8, xv);
                        }
                    }
                }
            }
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const float v = wave_sum(ss[m]);
                if (lane == 0 && wid < kCPT / 64) red[m * (kCPT / 64) + wid] = v;
            }
            __syncthreads();
            float rrms[M];
#pragma unroll
            for (int m = 0; m < M; ++m) {
                float t = 0.f;
#pragma unroll
                for (int i = 0; i < kCPT / 64; ++i) t += red[m * (kCPT / 64) + i];
                rrms[m] = 1.0f / sqrtf(t / (float)P.k + P.eps);
            }
            if (threadIdx.x < kCPT) {
                for (int c = threadIdx.x; c < kvec; c += kCPT) {
                    const uint4 wv = ld16(P.ln_w + c * 8);
                    const uint32_t w4[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
                    for (int m = 0; m < M; ++m) {
                        uint4* px = reinterpret_cast<uint4*>(xs + m * P.k + c * 8);
                        const uint4 sv = *px;
                        const uint32_t s4[4] = {sv.x, sv.y, sv.z, sv.w};
                        uint32_t o[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            o[j] = pack_bf16x2((bf16_lo(s4[j]) * rrms[m]) * bf16_lo(w4[j]),
                                               (bf16_hi(s4[j]) * rrms[m]) * bf16_hi(w4[j]));
                        *px = make_uint4(o[0], o[1], o[2], o[3]);
                    }
                }
            }
        }
        __syncthreads();
        MV_CSTAMP(p, 1);

        // ---- the projection: units of this workgroup, two stages of tiles always in flight ---------
        const int per_step = P.halves * nwg;
        const int nsteps = (P.nt + per_step - 1) / per_step;
        const int gw = P.halves == 2 ? 8 : 16;
        const int half = P.halves == 2 ? (wid >> 3) : 0;
        const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
            P.y, 0, (int)(((int64_t)(M - 1) * P.ldy + ((P.flags & MV_GEMM_SWIGLU) ? P.n / 2 : P.n)) * 2), 0x00020000);
        const bool last_phase = p + 1 == a.nphase;
        for (int step = 0; step < nsteps; ++step) {
            const bool more_here = step + 1 < nsteps;
            ChainUnit nxt = cur;
            if (more_here) nxt = setup(P, step + 1);

            const int th = (int)threadIdx.x - half * (8 * kWave);   // thread index inside its half
            const bool ep_thread = th >= 0 && th < 16 * M && cur.valid;

            float acc[M];
#pragma unroll
            for (int m = 0; m < M; ++m) acc[m] = 0.f;
            const int nchunk = (cur.ntile + U - 1) / U;
            // The first unit behind a barrier has four stages waiting: chunks 0 and 1 in the LDS slab,
            // 2 and 3 in (or on their way to) the registers.
            int cb = 0;
            if (DEEP && p > 0 && step == 0) {
                cb = 2;
                if (cur.valid) {
#pragma unroll
                    for (int i = 0; i < 2 * U; ++i) {
                        const uint4 w = slab[i * 64];
                        compute_tile(w, cur, P.k, i, acc);
                    }
                }
            }
            const int rem = max(nchunk - cb, 0);
            const int npair = (rem + 1) >> 1;
            if (cur.valid) {   // wave-uniform; a unit past the end is skipped (its stages are dropped)
                for (int pr = 0; pr < npair - 1; ++pr) {
                    compute(bufA, cur, P.k, cb + 2 * pr, acc);
                    load(bufA, cur, cb + 2 * pr + 2);
                    compute(bufB, cur, P.k, cb + 2 * pr + 1, acc);
                    load(bufB, cur, cb + 2 * pr + 3);
                }
            }
            const int cl = (cur.valid && npair > 0) ? cb + 2 * (npair - 1) : (1 << 24);
            if (more_here) {   // the next unit of this projection follows without a gap
                compute(bufA, cur, P.k, cl, acc);
                load(bufA, nxt, 0);
                compute(bufB, cur, P.k, cl + 1, acc);
                load(bufB, nxt, 1);
            } else {
                compute(bufA, cur, P.k, cl, acc);
                compute(bufB, cur, P.k, cl + 1, acc);
            }

            if (!more_here) MV_CSTAMP(p, 2);
            // lanes r, r+16, r+32, r+48 hold the four k-quarters of row r
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
            if (ep_thread) {
                const int m = th >> 4, rr = th & 15;
                const int w0 = half * 8;
                float s = 0.f;
                for (int w8 = 0; w8 < gw; ++w8) s += red[((w0 + w8) * M + m) * 16 + rr];
                const int n = cur.strip * 16 + rr;
                if (P.flags & MV_GEMM_SWIGLU) {
                    if (rr < 8) {
                        float g = 0.f;
                        for (int w8 = 0; w8 < gw; ++w8) g += red[((w0 + w8) * M + m) * 16 + rr + 8];
                        __builtin_amdgcn_raw_buffer_store_b16(
                            (unsigned short)f32_to_bf16_bits(silu_mul_rounded(g, s)), yrs,
                            (int)(((int64_t)m * P.ldy + cur.strip * 8 + rr) * 2), 0, 16 /*sc1*/);
                    }
                } else {
                    float t = round_bf16(s);
                    float bias_v, res_v;
                    if (step == 0) {            // fetched while x was staged
                        bias_v = epi[(half * 2 + 0) * M * 16 + th];
                        res_v = epi[(half * 2 + 1) * M * 16 + th];
                    } else {
                        bias_v = P.bias ? bf16_bits_to_f32(P.bias[n]) : 0.f;
                        res_v = 0.f;
                        if (P.residual) {
                            const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(
                                const_cast<uint16_t*>(P.residual), 0,
                                (int)(((int64_t)(M - 1) * P.ld_res + P.n) * 2), 0x00020000);
                            res_v = bf16_bits_to_f32(__builtin_amdgcn_raw_buffer_load_b16(
                                rrs, (int)(((int64_t)m * P.ld_res + n) * 2), 0, 16 /*sc1*/));
                        }
                    }
                    if (P.bias) t = round_bf16(t + bias_v);
                    if (P.residual) t = round_bf16(t + res_v);
                    __builtin_amdgcn_raw_buffer_store_b16((unsigned short)f32_to_bf16_bits(t), yrs,
                                                          (int)(((int64_t)m * P.ldy + n) * 2), 0, 16 /*sc1*/);
                }
            }
            if (more_here) __syncthreads();   // `red` is reused by the next unit
            cur = nxt;
        }

        if (!last_phase) {
            // ---- end of projection p: make its outputs visible, arrive, and put the next projection's
            // first tiles in flight before anybody waits ----------------------------------------------
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's write-through stores are acknowledged
            __syncthreads();
            MV_CSTAMP(p, 3);
            ++barriers_done;
            if (threadIdx.x == 0)
                __hip_atomic_store(a.flags + blockIdx.x, s_base + (unsigned int)barriers_done, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            cur = setup(a.ph[p + 1], 0);
            load(bufA, cur, 0);
            load(bufB, cur, 1);
            if (DEEP) {
                // Four stages for the wait ahead instead of two, without more registers: the first
                // two are parked in this wave's LDS slab as soon as they land, and the registers
                // are sent for the next two.
#pragma unroll
                for (int i = 0; i < U; ++i) {
                    slab[i * 64] = bufA[i];
                    slab[(U + i) * 64] = bufB[i];
                }
                load(bufA, cur, 2);
                load(bufB, cur, 3);
            }
        }
    }
    // every workgroup read the starting number before it reached the launch's last barrier, and
    // workgroup 0 is past that barrier here: the next launch starts behind this one's numbers
    if (blockIdx.x == 0 && threadIdx.x == 0 && barriers_done > 0)
        __hip_atomic_store(a.base, s_base + (unsigned int)barriers_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

int dot_form_waves(int m, int nt, int kt);   // wgemm.hip: waves per workgroup the stand-alone kernel uses
int dot_form_ksplit(int m, int nt, int kt);  // wgemm.hip: k-slices the stand-alone planner picks

struct ChainWorkspace {
    unsigned int base;
    int status;
    int pad[14];
    unsigned int flags[kMaxChainWgs];
};
static_assert(sizeof(ChainWorkspace) == 64 + kMaxChainWgs * 4, "layout");

template <int M>
static int launch_chain(const ChainArgs& a, int grid, size_t lds, hipStream_t s) {
    static LdsOptIn opted;
    if (int rc = ensure_dynamic_lds((const void*)decode_chain_kernel<M>, lds, opted)) return rc;
    hipLaunchKernelGGL((decode_chain_kernel<M>), dim3(grid), dim3(kCT), lds, s, a);
    return check_launch();
}

}  // namespace mv

extern "C" {

size_t mv_decode_chain_workspace_bytes(void) { return sizeof(mv::ChainWorkspace); }

int mv_decode_chain_status(const void* workspace, int* status_out) {
    if (!workspace || !status_out) return MV_ERR_INVALID_ARG;
    hipError_t e = hipMemcpy(status_out, &((const mv::ChainWorkspace*)workspace)->status, sizeof(int),
                             hipMemcpyDeviceToHost);
    if (e != hipSuccess) {
        mv::set_last_hip_error(e);
        return MV_ERR_LAUNCH;
    }
    return MV_OK;
}

int mv_decode_chain(const mv_chain_phase* phases, int nphase, int m, void* workspace, size_t workspace_bytes,
                    mv_stream_t stream) {
    using namespace mv;
    if (!phases || nphase < 1 || nphase > kChainMaxPhases || m < 1 || m > 4) return MV_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < sizeof(ChainWorkspace) || !aligned16(workspace)) return MV_ERR_WORKSPACE;
    ChainArgs a = {};
    a.nphase = nphase;
    int kmax = 0;
    for (int p = 0; p < nphase; ++p) {
        const mv_chain_phase& s = phases[p];
        ChainPhase& P = a.ph[p];
        if (!s.x || !s.w_tiles || !s.y || s.n <= 0 || s.k <= 0) return MV_ERR_INVALID_ARG;
        if ((s.n % 16) != 0 || (s.k % 32) != 0) return MV_ERR_UNSUPPORTED;
        if (!aligned16(s.x) || !aligned16(s.w_tiles) || (s.ldx % 8) != 0 || s.ldx < s.k) return MV_ERR_INVALID_ARG;
        const bool sw = (s.flags & MV_GEMM_SWIGLU) != 0;
        if (s.flags & ~MV_GEMM_SWIGLU) return MV_ERR_UNSUPPORTED;
        if (sw && (s.bias || s.residual)) return MV_ERR_UNSUPPORTED;
        if (s.ldy < (sw ? s.n / 2 : s.n)) return MV_ERR_INVALID_ARG;
        if (s.residual && s.ld_res < s.n) return MV_ERR_INVALID_ARG;
        if ((s.add_r == nullptr) != (s.sum_out == nullptr)) return MV_ERR_INVALID_ARG;
        if (s.add_r && !s.ln_w) return MV_ERR_INVALID_ARG;
        if (s.ln_w && !aligned16(s.ln_w)) return MV_ERR_INVALID_ARG;
        if (s.add_r && (!aligned16(s.add_r) || !aligned16(s.sum_out) || (s.ld_add_r % 8) || (s.ld_sum_out % 8) ||
                        s.ld_add_r < s.k || s.ld_sum_out < s.k))
            return MV_ERR_INVALID_ARG;
        P.w = (const uint4*)s.w_tiles;
        P.n = s.n;
        P.k = s.k;
        P.nt = s.n / 16;
        P.kt = s.k / 32;
        // the shape (and with it the summation order) the stand-alone kernel uses for this projection
        if (dot_form_ksplit(m, P.nt, P.kt) != 1) return MV_ERR_UNSUPPORTED;
        P.halves = dot_form_waves(m, P.nt, P.kt) == 16 ? 1 : 2;
        P.x = (const uint16_t*)s.x;
        P.ldx = s.ldx;
        P.prologue = s.add_r ? kCProAddNorm : (s.ln_w ? kCProNorm : kCProNone);
        P.var_from_rounded = (s.norm_flags & MV_NORM_VAR_FROM_ROUNDED) ? 1 : 0;
        P.add_r = (const uint16_t*)s.add_r;
        P.ld_add_r = s.ld_add_r;
        P.ln_w = (const uint16_t*)s.ln_w;
        P.eps = s.eps;
        P.sum_out = (uint16_t*)s.sum_out;
        P.ld_sum_out = s.ld_sum_out;
        P.bias = (const uint16_t*)s.bias;
        P.residual = (const uint16_t*)s.residual;
        P.ld_res = s.ld_res;
        P.y = (uint16_t*)s.y;
        P.ldy = s.ldy;
        P.flags = s.flags;
        if (s.k > kmax) kmax = s.k;
    }
    size_t lds = 16 + (size_t)kCW * m * 16 * 4 + (size_t)2 * 2 * m * 16 * 4 + (size_t)m * kmax * 2;
    if (lds > 160 * 1024 - 512) return MV_ERR_UNSUPPORTED;
    a.slab_off = 0;
    if (m == 1 && nphase > 1) {   // two parked stages of four 1 KiB tiles per wave
        const size_t off = (lds + 15) / 16 * 16, slabs = (size_t)kCW * 2 * 4 * 1024;
        if (off + slabs <= 160 * 1024 - 512) {
            a.slab_off = (int)off;
            lds = off + slabs;
        }
    }
    ChainWorkspace* ws = (ChainWorkspace*)workspace;
    a.flags = ws->flags;
    a.base = &ws->base;
    a.status = &ws->status;
    a.timeout_ticks = 200000000ull;   // 2 s of the 100 MHz counter
    // one workgroup per CU: all of them resident at once, which the grid barrier relies on
    const int grid = num_cus() < kMaxChainWgs ? num_cus() : kMaxChainWgs;
    hipStream_t s = (hipStream_t)stream;
    switch (m) {
        case 1: return launch_chain<1>(a, grid, lds, s);
        case 2: return launch_chain<2>(a, grid, lds, s);
        case 3: return launch_chain<3>(a, grid, lds, s);
        default: return launch_chain<4>(a, grid, lds, s);
    }
}

}  // extern "C"

#ifdef MV_CHAIN_STAMP
extern "C" int mv_debug_read_chain_stamps(void* dst, size_t bytes) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(mv::g_cstamp), bytes) == hipSuccess ? 0 : -3;
}
#endif
