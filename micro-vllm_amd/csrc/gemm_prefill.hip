// Many-token projection  y[M,N] = x[M,K] . W^T  on the matrix cores, over the same wave-tile
// weight image the single-token kernels stream (tile (nt,kt) = 16 rows x 32 k = 1 KiB, lane l
// owns W[16nt + (l&15)][32kt + 8(l>>4) + 0..7]): that image IS the A-operand fragment order of
// v_mfma_f32_16x16x32_bf16, so weight fragments go from global memory straight into registers,
// one fully coalesced 1 KiB request per tile, and only x passes through LDS.
//
// Workgroup = 4 waves = 256 (n) x 128 (m) outputs; wave w owns n-tiles 4w..4w+3 against all
// eight m-tiles (128 fp32 accumulators per lane).  x tiles of 128 rows x 64 k go global -> LDS
// directly, three stages deep, in fragment order; half of a k-tile's x fragments are read while
// the previous k-tile is still being multiplied; weight fragments are requested two k-tiles
// before they are multiplied.
//
// Workgroups are numbered so that the ones resident together on an XCD (blockIdx mod 8) share
// few n-blocks and many m-blocks: each weight fragment is then fetched into that XCD's L2 once
// and hit by the other m-blocks.
//
// Epilogue rounding as in the single-token kernels: t = bf16(acc); t = bf16(t + bias);
// t = bf16(t + residual); SWIGLU: out = bf16(silu(gate_t) * up_t) with up/gate rows interleaved
// by 8 inside each 16-row tile.
#include "common.h"

#include <stdlib.h>

#include <algorithm>

namespace mv {

constexpr int kPM = 128;            // rows of x per workgroup
constexpr int kPN = 256;            // output columns per workgroup
constexpr int kPK = 64;             // k per LDS stage
constexpr int kPStage = kPM * kPK * 2; // bytes of one x stage in LDS
constexpr int kPMT = kPM / 16;      // m-tiles per wave
constexpr int kPNT = 4;             // n-tiles per wave
constexpr int kWSlotBytes = 8 * 32 * 64 * 16;   // a workgroup's accumulators: 8 waves x 32 registers of 16 bytes x 64 lanes

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;

// s_waitcnt vmcnt(n) alone (gfx9 encoding: vmcnt in bits 3:0 and 15:14, the other counters at
// their maxima)
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | (7 << 4) | (15 << 8));
}

struct PGemmArgs {
    const uint16_t* x;
    int64_t ldx;
    const uint4* w;
    const uint16_t* bias;
    const uint16_t* residual;
    int64_t ld_res;
    void* y;
    int64_t ldy;
    int m, n, k;
    int nt, kt;       // 16-row strips, 32-wide k-tiles
    int mblocks, nblocks, nb_per_xcd;
    int mband;        // wide form: m-blocks walked together against all n-blocks of an XCD
    int flags;
    // k-split: slice ks of a tile covers k-tiles [ks * ktps, (ks + 1) * ktps) and writes its fp32
    // accumulators to partial[ks][m][n]; a second launch adds the slices and applies the epilogue
    int ksplit, ktps;
    float* partial;
    // wide form: slots of kWSlotBytes, one per workgroup, for the accumulators of a tile that several
    // workgroups cut along k between them (null: tiles are never cut inside the launch), and the
    // counters through which the parts of such a tile count themselves in (zero between launches)
    float* cut_ws;
    unsigned* cut_count;
};

#ifdef MV_GEMM_STAMP
// Phase stamps of one wave (build variant for tools/stamp_gemm_prefill.py): s_memtime ticks at four
// points of every k stage, for the workgroup selected by kStampBlock.  MV_EXP_NO_WLOAD /
// MV_EXP_NO_XREAD additionally remove one operand path (wrong results, timing only).
constexpr int kStampBlock = 777;
static __device__ unsigned long long g_stamp[4 * 512];
#define MV_STAMP(slot)                                                            \
    do {                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                        \
        if (blockIdx.x == kStampBlock && threadIdx.x == 0 && s < 512)             \
            g_stamp[4 * s + (slot)] = __builtin_readcyclecounter();               \
        __builtin_amdgcn_sched_barrier(0);                                        \
    } while (0)
#else
#define MV_STAMP(slot) do {} while (0)
#endif
#ifdef MV_GEMM_WSTAMP
// Wide form, tools/stamp_gemm_wide.py: wall-clock ticks (100 MHz) of wave 0 of workgroup kWStampBlock at
// the top of every second k-tile and on either side of a tile's output stores; slot = tile-local
// index + 512 * piece (k-tiles 0, 2, .. then nkt, nkt + 1 for the two sides of the epilogue).
#ifndef MV_WSTAMP_BLOCK
#define MV_WSTAMP_BLOCK 77
#endif
constexpr int kWStampBlock = MV_WSTAMP_BLOCK;
static __device__ unsigned long long g_wstamp[4096];
#define MV_WSTAMP(i)                                                                         \
    do {                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        const int si_ = (i) - g0 + tile_no * 512;                                      \
        if (blockIdx.x == kWStampBlock && threadIdx.x == 0 && si_ < 2048) {                   \
            g_wstamp[si_] = wall_clock64();                                                  \
            g_wstamp[2048 + si_] = clock64();   /* shader clock: the ratio of the two gives its frequency */ \
        }                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                   \
    } while (0)
#else
#define MV_WSTAMP(i) do {} while (0)
#endif
// MV_EXP_NO_MFMA (timing experiment, wrong results): operands are delivered and consumed by an empty
// asm statement instead of the matrix instruction; MV_EXP_NO_BARRIER drops the per-stage barrier.
#ifdef MV_EXP_NO_MFMA
#define MV_MFMA(c, a_, b_) asm volatile("" : "+v"(c) : "v"(a_), "v"(b_))
#else
#define MV_MFMA(c, a_, b_) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_, b_, c, 0, 0, 0)
#endif

// One wave's 64 (n) x 128 (m) block of accumulators to memory, with the projection epilogue:
// acc[i][t][j] = C[n = 16 (nt0+i) + 4q + j][m = m0 + 16 t + r]
// The epilogue loops below are fully unrolled and go row by row (t outside, the tiles of the row
// inside), with a scheduling fence per row: the row's addresses are then formed once, used and dropped;
// run side by side, the rows' temporaries do not fit beside 128 accumulators and the operand
// fragments already fetched for the next tile.
#define MV_EPI_FENCE __builtin_amdgcn_sched_barrier(0)

// silu(gate) * up on two accumulators rounded to bf16, as silu_mul_rounded, with the hardware's
// exp2 and reciprocal (1 ulp each) in place of the library exp and the IEEE division: 10 vector
// instructions instead of 33.  A tile's epilogue forms 64 of these per lane while the matrix pipe
// waits.  Against the same product formed in fp64 from the same rounded accumulators (29 M outputs
// of the 8B gate/up shape, tools/check_quick_silu.py): 16 outputs in 10^6 differ, by one bf16 ulp;
// with the library exp and the IEEE division 7 in 10^6 do.
__device__ __forceinline__ float silu_mul_rounded_quick(float gate_acc, float up_acc) {
    const float g = round_bf16(gate_acc), u = round_bf16(up_acc);
    const float e = __builtin_amdgcn_exp2f(g * -1.44269504088896341f);
    const float sig = __builtin_amdgcn_rcpf(1.0f + e);
    return (g * sig) * u;
}
template <int NT>
__device__ __forceinline__ void store_wave_tile(const PGemmArgs& a, const f32x4_t (&acc)[NT][kPMT], int nt0,
                                                int m0, int ks, int r, int q, const unsigned char* lower = nullptr,
                                                int nlower = 0) {
    const bool sw = (a.flags & MV_GEMM_SWIGLU) != 0, of32 = (a.flags & MV_GEMM_OUT_F32) != 0;
    // `lower`: this wave's 32 KiB in the first of `nlower` accumulator slots (8 slots apart) that hold
    // the sums over the lower parts of k, register by register as they lie here; added in part order.
    // They were written by other workgroups of this launch: read with sc1 loads (served by L2 or
    // memory, never by this CU's L1), the loads of one row of tiles in flight together.
    const unsigned lane16 = (threadIdx.x & 63) * 16;
    auto sum = [&](int i, int t) __attribute__((always_inline)) {
        f32x4_t c = acc[i][t];
        for (int e = 0; e < nlower; ++e) {
            f32x4_t part;
            asm volatile("global_load_dwordx4 %0, %1, %2 sc1\n\ts_waitcnt vmcnt(0)"
                         : "=v"(part) : "v"(lane16), "s"(lower + (size_t)e * 8 * kWSlotBytes + (i * 8 + t) * 1024) : "memory");
            c += part;
        }
        return c;
    };
    if (a.partial) {
#pragma unroll
        for (int t = 0; t < kPMT; ++t) {
            MV_EPI_FENCE;
            const int m = m0 + t * 16 + r;
            float* const row = a.partial + ((int64_t)ks * a.m + min(m, a.m - 1)) * a.n + nt0 * 16 + q * 4;
#pragma unroll
            for (int i = 0; i < NT; ++i)
                if (nt0 + i < a.nt && m < a.m) *reinterpret_cast<f32x4_t*>(row + i * 16) = acc[i][t];
        }
        return;
    }
    if (sw) {
        // Rows 0..7 of a tile are up, 8..15 the matching gates: lane l < 32 holds four up values and
        // lane l + 32 their gates.  Two v_permlane32_swap leave the lower lane (up, gate) of columns
        // 0, 1 of its four and the upper lane those of columns 2, 3, so that every lane forms two of
        // the products instead of half the lanes forming four.  A lane then holds one dword (two
        // outputs) of each of the wave's tiles, and the four lanes of a row hold the 16 bytes of a
        // tile between them: a 4 x 4 transposition across those lanes (tiles 0 <-> 2, 1 <-> 3 by
        // v_permlane32_swap, then 0 <-> 1, 2 <-> 3 by v_permlane16_swap) gives lane (r, q) all of tile
        // q - one 16-byte store per row and lane, 64 contiguous bytes per row and instruction, where
        // dword stores covered 16.
        const bool whole = (a.ldy % 8) == 0 && (reinterpret_cast<uintptr_t>(a.y) & 15) == 0;
#pragma unroll
        for (int t = 0; t < kPMT; ++t) {
            MV_EPI_FENCE;
            const int m = m0 + t * 16 + r;
            uint16_t* const row = reinterpret_cast<uint16_t*>(a.y) + (int64_t)min(m, a.m - 1) * a.ldy + nt0 * 8;
            uint32_t o[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const f32x4_t c = sum(i, t);
                const auto s0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(c[0]), __float_as_uint(c[2]), false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(c[1]), __float_as_uint(c[3]), false, false);
                o[i] = pack_bf16x2(silu_mul_rounded_quick(__uint_as_float(s0[1]), __uint_as_float(s0[0])),
                                   silu_mul_rounded_quick(__uint_as_float(s1[1]), __uint_as_float(s1[0])));
            }
#ifdef MV_EXP_EPI_NO_WRITE
            asm volatile("" : : "v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3]));   // timing experiment only
#else
            if (whole) {
                const auto a02 = __builtin_amdgcn_permlane32_swap(o[0], o[2], false, false);
                const auto a13 = __builtin_amdgcn_permlane32_swap(o[1], o[3], false, false);
                const auto lo = __builtin_amdgcn_permlane16_swap(a02[0], a13[0], false, false);   // columns 0,1 | 4,5
                const auto hi = __builtin_amdgcn_permlane16_swap(a02[1], a13[1], false, false);   // columns 2,3 | 6,7
                if (m < a.m && q < NT && nt0 + q < a.nt)
                    *reinterpret_cast<uint4*>(row + q * 8) = make_uint4(lo[0], hi[0], lo[1], hi[1]);
            } else {
#pragma unroll
                for (int i = 0; i < NT; ++i)
                    if (m < a.m && nt0 + i < a.nt)
                        *reinterpret_cast<uint32_t*>(row + i * 8 + (q & 1) * 4 + (q >> 1) * 2) = o[i];
            }
#endif
        }
        return;
    }
    if (of32) {
#pragma unroll
        for (int t = 0; t < kPMT; ++t) {
            MV_EPI_FENCE;
            const int m = m0 + t * 16 + r;
            float* const row = reinterpret_cast<float*>(a.y) + (int64_t)min(m, a.m - 1) * a.ldy + nt0 * 16 + q * 4;
#pragma unroll
            for (int i = 0; i < NT; ++i)
                if (nt0 + i < a.nt && m < a.m) *reinterpret_cast<f32x4_t*>(row + i * 16) = sum(i, t);
        }
        return;
    }
    // bf16 output: four columns (8 bytes) of one row per lane and tile; t = bf16(acc), + bias, + residual
    // with a rounding after each
    auto rounded = [&](const f32x4_t& c, const uint16_t* bias, const uint16_t* res) __attribute__((always_inline)) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float tv = round_bf16(c[j]);
            if (a.bias) tv = round_bf16(tv + bf16_bits_to_f32(bias[j]));
            if (a.residual) tv = round_bf16(tv + bf16_bits_to_f32(res[j]));
            v[j] = tv;
        }
        uint2 o;
        o.x = pack_bf16x2(v[0], v[1]);
        o.y = pack_bf16x2(v[2], v[3]);
        return o;
    };
    // Two neighbouring tiles at a time where the rows allow 16-byte stores: v_permlane16_swap
    // exchanges the odd 16-lane rows (q = 1, 3) of the first tile's registers with the even rows of
    // the second's, after which lane (r, q) holds columns 8 (q >> 1) .. + 7 of tile i + (q & 1) - a
    // store instruction then covers 64 contiguous bytes of each of its 16 rows instead of 32.
    const bool pairs = (a.ldy % 8) == 0 && (reinterpret_cast<uintptr_t>(a.y) & 15) == 0;
#pragma unroll
    for (int t = 0; t < kPMT; ++t) {
        MV_EPI_FENCE;
        const int m = m0 + t * 16 + r, mc = min(m, a.m - 1);
        uint16_t* const row = reinterpret_cast<uint16_t*>(a.y) + (int64_t)mc * a.ldy + nt0 * 16;
        const uint16_t* const res = a.residual ? a.residual + (int64_t)mc * a.ld_res + nt0 * 16 + q * 4 : nullptr;
        const uint16_t* const bias = a.bias ? a.bias + nt0 * 16 + q * 4 : nullptr;
#pragma unroll
        for (int i = 0; i < NT; i += 2) {
            if (nt0 + i >= a.nt) continue;  // uniform
            if (i + 1 < NT && pairs && nt0 + i + 1 < a.nt) {
                constexpr int kLast = NT - 1;
                const uint2 p0 = rounded(sum(i, t), bias + i * 16, res + i * 16);
                const uint2 p1 = rounded(sum(i + 1 < NT ? i + 1 : kLast, t), bias + (i + 1) * 16, res + (i + 1) * 16);
                const auto rx = __builtin_amdgcn_permlane16_swap(p0.x, p1.x, false, false);
                const auto ry = __builtin_amdgcn_permlane16_swap(p0.y, p1.y, false, false);
                if (m < a.m)
                    *reinterpret_cast<uint4*>(row + (i + (q & 1)) * 16 + (q >> 1) * 8) = make_uint4(rx[0], ry[0], rx[1], ry[1]);
            } else {
#pragma unroll
                for (int ii = i; ii < i + 2 && ii < NT; ++ii)
                    if (nt0 + ii < a.nt) {  // uniform
                        const uint2 o = rounded(sum(ii, t), bias + ii * 16, res + ii * 16);
                        if (m < a.m) *reinterpret_cast<uint2*>(row + ii * 16 + q * 4) = o;
                    }
            }
        }
    }
}

__global__ __launch_bounds__(256, 2) void gemm_prefill_kernel(PGemmArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char xs[3][kPStage];

    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, q = lane >> 4;

    // blockIdx -> (m-block, n-block): an XCD takes a contiguous range of n-blocks and walks the
    // m-blocks fastest
    const int tiles = 8 * a.nb_per_xcd * a.mblocks;
    const int ks = blockIdx.x / tiles, bid = blockIdx.x - ks * tiles;
    const int kt0 = ks * a.ktps;               // first k-tile of this slice
    const int xcd = bid & 7, idx = bid >> 3;
    const int nb = xcd * a.nb_per_xcd + idx / a.mblocks;
    const int mb = idx % a.mblocks;
    if (nb >= a.nblocks || idx / a.mblocks >= a.nb_per_xcd) return;
    const int m0 = mb * kPM;
    const int nt0 = nb * (kPN / 16) + wid * kPNT;

    // weight fragment streams of this wave (strips past the end are clamped and never stored)
    const uint4* wp[kPNT];
#pragma unroll
    for (int i = 0; i < kPNT; ++i)
        wp[i] = a.w + ((int64_t)min(nt0 + i, a.nt - 1) * a.kt + kt0) * 64 + lane;

    // ---- x: global -> LDS directly, three stages deep, in fragment order ---------------------
    // The LDS image of a stage is 16 fragment tiles (8 m-tiles x 2 k-tiles) of 1 KiB in MFMA
    // operand order: lane l of the tile (mt, kk) holds x[16 mt + (l & 15)][32 kk + 8 (l >> 4) ..+7],
    // so a fragment read is linear in the lane (no bank conflicts).  Wave w fills tiles 4w..4w+3
    // with global_load_lds (no staging registers, no LDS store instructions).  These loads are
    // issued through inline assembly: when the compiler sees them it stops trusting request
    // order and waits for ALL outstanding loads before every weight use and every LDS read;
    // hidden from it they only make its counted waits wait for more (requests complete in
    // order), never for less.  The wait that makes the data usable is placed by hand.
    const int nstage = a.ktps >> 1;  // a slice is a multiple of 64 k (checked on the host)
    const uint16_t* xg[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int tile = wid * 4 + j;
        xg[j] = a.x + (int64_t)min(m0 + (tile >> 1) * 16 + r, a.m - 1) * a.ldx + kt0 * 32 + ((tile & 1) * 4 + q) * 8;
    }
    const unsigned xs_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)&xs[0][0];
    auto x_fetch = [&](int s, int buf) {
        const int sc = min(s, nstage - 1);  // past the end: a harmless re-fetch keeps the counts fixed
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint16_t* src = xg[j] + sc * kPK;
            const unsigned dst = xs_base + buf * kPStage + (wid * 4 + j) * 1024;
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(dst) : "memory");
        }
    };
    auto x_frag = [&](int buf, int t, int kk) {
#ifdef MV_EXP_NO_XREAD
        return __builtin_bit_cast(bf16x8_t, make_uint4(lane, t, kk, buf));  // timing experiment only
#endif
        return __builtin_bit_cast(
            bf16x8_t, *reinterpret_cast<const uint4*>(&xs[buf][(t * 2 + kk) * 1024 + lane * 16]));
    };

    f32x4_t acc[kPNT][kPMT];
#pragma unroll
    for (int i = 0; i < kPNT; ++i)
#pragma unroll
        for (int t = 0; t < kPMT; ++t) acc[i][t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // weight fragments: a ring of three k-tiles; a slot is refilled (two k-tiles ahead) at the start
    // of the k-tile after the one that used it
    uint4 wA[kPNT], wB[kPNT], wC[kPNT];
    auto w_load = [&](uint4(&w)[kPNT], int kt) {
        const int ktc = min(kt, a.ktps - 1);
#pragma unroll
        for (int i = 0; i < kPNT; ++i) {
#ifdef MV_EXP_NO_WLOAD
            if (kt > 2) continue;  // timing experiment only: weights never refreshed
#endif
            w[i] = ld16(wp[i] + (int64_t)ktc * 64);
        }
    };
    x_fetch(0, 0);
    x_fetch(1, 1);
    w_load(wA, 0);
    w_load(wB, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // x fragments of two k-tiles alternate between X0 and X1; the first half of a k-tile's
    // fragments is read while the previous k-tile is still being multiplied
    bf16x8_t X0[kPMT], X1[kPMT];
#pragma unroll
    for (int t = 0; t < kPMT / 2; ++t) X0[t] = x_frag(0, t, 0);

    // one k-tile: (buf, kk) = where it lives; (nbuf, nkk) = the k-tile after it
    // `w` = this k-tile's weight fragments; `wr` = the ring slot the previous k-tile used, refilled
    // here, first thing, with the k-tile two ahead (left to itself the scheduler sinks these
    // requests to the end of the stage, and the next stage then starts by waiting for them).
    // `fetch_stage` >= 0: also request that stage of x, after the first half of the MFMAs.
    auto ktile = [&](uint4(&w)[kPNT], uint4(&wr)[kPNT], bf16x8_t(&xc)[kPMT], bf16x8_t(&xn)[kPMT], int buf,
                     int kk, int nbuf, int nkk, int kt_refill, int fetch_stage, int fetch_buf) {
        bf16x8_t wf[kPNT];
#pragma unroll
        for (int i = 0; i < kPNT; ++i) wf[i] = __builtin_bit_cast(bf16x8_t, w[i]);
        __builtin_amdgcn_sched_barrier(0);
        w_load(wr, kt_refill);
        __builtin_amdgcn_sched_barrier(0);
        // (pinning this order with scheduling fences measured 3 % slower than letting the
        // scheduler interleave reads and MFMAs itself)
#pragma unroll
        for (int t = kPMT / 2; t < kPMT; ++t) xc[t] = x_frag(buf, t, kk);
#pragma unroll
        for (int t = 0; t < kPMT / 2; ++t)
#pragma unroll
            for (int i = 0; i < kPNT; ++i)
                MV_MFMA(acc[i][t], wf[i], xc[t]);
        // The x requests are invisible to the compiler's request counting, so they go out behind
        // the waits for this k-tile's weights: in front of them they would make those waits cover
        // the previous refill as well.
        if (fetch_stage >= 0) {
            __builtin_amdgcn_sched_barrier(0);
            x_fetch(fetch_stage, fetch_buf);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int t = 0; t < kPMT / 2; ++t) xn[t] = x_frag(nbuf, t, nkk);
#pragma unroll
        for (int t = kPMT / 2; t < kPMT; ++t)
#pragma unroll
            for (int i = 0; i < kPNT; ++i)
                MV_MFMA(acc[i][t], wf[i], xc[t]);
    };
    // One stage: request stage s+2 of x (its buffer was last read in stage s-1), two k-tiles, wait
    // for this wave's part of stage s+2 (two weight refills were requested after it), barrier: the
    // next stage's later reads and the stage after's early reads both see complete data.
    auto stage = [&](uint4(&w0)[kPNT], uint4(&w1)[kPNT], uint4(&w2)[kPNT], int s, int b0, int b1, int b2) {
        MV_STAMP(0);
        ktile(w0, w2, X0, X1, b0, 0, b0, 1, 2 * s + 2, s + 2, b2);
        ktile(w1, w0, X1, X0, b0, 1, b1, 0, 2 * s + 3, -1, 0);
        MV_STAMP(1);
        wait_vmcnt<kPNT>();   // only the second k-tile's refill was requested after the x stage
        MV_STAMP(2);
#ifndef MV_EXP_NO_BARRIER
        __syncthreads();
#endif
        MV_STAMP(3);
    };
    // three stages per pass and no branch inside the pass (a branch between stages makes the
    // compiler wait for every outstanding request where the paths meet); the one or two stages
    // left over follow
    int s = 0;
    for (; s + 2 < nstage; s += 3) {
        stage(wA, wB, wC, s, 0, 1, 2);
        stage(wC, wA, wB, s + 1, 1, 2, 0);
        stage(wB, wC, wA, s + 2, 2, 0, 1);
    }
    if (s < nstage) stage(wA, wB, wC, s, 0, 1, 2);
    if (s + 1 < nstage) stage(wC, wA, wB, s + 1, 1, 2, 0);

    store_wave_tile(a, acc, nt0, m0, ks, r, q);
}


// ---- the wide form: 256 (n) x 256 (m) outputs per workgroup, both operands through LDS -----------
// The form above pulls (128 + 256) operand rows through a CU's vector memory path per 128 x 256
// outputs, and at thousands of tokens that path, not the matrix pipe, sets its pace.  Here eight
// waves share a 256 x 256 tile - (256 + 256) rows per twice the outputs, a third fewer bytes per
// multiply-add out of L2 - and nothing goes through registers on the way in: weight tiles (already
// in fragment order) and x rows are both written to LDS by global_load_lds, the weights four
// k-tiles ahead of their use, x three 64-wide stages ahead.  Wave w owns n-tiles 4 (w & 3) .. +3
// against m-tiles 8 (w >> 2) .. +7: the same 64 x 128 block of accumulators as above.
//
// LDS (all 160 KiB): x ring 3 x 32 KiB + weight ring 4 x 16 KiB.
//   weight slot: 16 n-tiles x 1 KiB in fragment order (a tile is copied as it lies in memory);
//   x slot: 16 m-tiles x 2 KiB, an m-tile = 16 rows x 128 bytes (64 k); the 16-byte chunk c of row
//   r sits at chunk position c ^ ((r >> 1) & 7) of its row, which makes the fragment reads (lane
//   (r, q) reads chunk 4 kk + q of row r) free of bank conflicts while every copy instruction
//   still moves whole 128-byte row segments (8 rows per instruction; the permutation is applied to
//   the global address of each lane, the LDS side of such a copy is always linear in the lane).
// Schedule of k-tile kt (stage s = kt / 2): request weight tile kt + 4 into the slot whose
// fragments were read during kt - 1 and, on odd kt, x stage s + 3 likewise; read the fragments of
// kt + 1 into the second register set; 32 MFMAs on the first (copies, reads and MFMAs interleaved
// in eight groups); wait until this wave's pieces of weight tile kt + 2 and of the x stage it
// belongs to have landed (8 younger requests may be outstanding, on either parity); barrier.
//
// Measured on the gate/up projection at 4096 tokens (962 GFLOP): this schedule with no copies
// and no fragment reads (MFMAs and the barrier alone) 610 us, with the reads 683, with the copies as
// well 795-810 (1.19-1.21 PFLOP/s; the 128 x 256 form 940).  Each fragment read costs the SIMD about 7
// cycles of matrix-pipe time and each copy instruction about 27 wherever they stand in the k-tile,
// and a tile of this size needs 24 + 8 of them per SIMD per 1024 cycles of MFMAs.
constexpr int kWM = 256, kWN = 256;
constexpr int kWXSlot = 32768, kWWSlot = 16384;
constexpr int kWLds = 3 * kWXSlot + 4 * kWWSlot;
constexpr int kWMaxGrid = 1024;
// The first kWCountBytes of a caller's workspace: one counter per XCD and left-over tile.  The caller
// zero-fills them once, when it allocates the workspace; the last part of a tile returns its counter
// to zero, and everything else the library keeps in a workspace (k-slices, accumulator slots) lies
// behind them.
constexpr size_t kWCountBytes = kWMaxGrid * sizeof(unsigned);

// Parts a left-over tile is cut into along k: at most as many as there are workgroups per left-over
// tile, each a whole number of 64-wide stages and at least 8 k-tiles long.  A part costs its share of
// the tile; every part more costs the last one about an eighth of a tile (it reads that part's 256 KiB of
// sums inside its epilogue, measured 8 us per part against 100 us tiles), so the count that
// makes 1/p + (p - 1)/8 smallest is taken - two to four in practice.
__host__ __device__ inline int wide_cut_parts(int rest, int wgs, int nkt) {
    if (rest <= 0) return 1;
    const int most = wgs / rest < 8 ? wgs / rest : 8;
    int best = 1;
    float best_cost = 1.0f;
    for (int p = 2; p <= most; ++p) {
        if (nkt % (2 * p) != 0 || nkt / p < 8) continue;
        const float cost = 1.0f / p + 0.125f * (p - 1);
        if (cost < best_cost) {
            best_cost = cost;
            best = p;
        }
    }
    return best;
}

// One wave's share of a tile: NT n-tiles from strip `wcol` of a block that is `width` strips wide.
template <int NT>
__device__ __forceinline__ void wide_tile(const PGemmArgs& a, unsigned char* lds, int wid, int wm, int wcol,
                                          int width) {
    unsigned char* const xring = lds;
    unsigned char* const wring = lds + 3 * kWXSlot;

    const int lane = threadIdx.x & 63;
    const int r = lane & 15, q = lane >> 4;

    // The grid is one workgroup per CU (or one per tile, with MVLLM_PREFILL_PERSIST=0).  A workgroup
    // that has a further tile never drains its pipeline between the two: the copies issued during the
    // last four k-tiles of a tile already bring the first weight tiles and x stages of the next one
    // into the ring slots that fall free, the output stores of the finished tile are left in flight,
    // and the accumulators start again from zero.
    //
    // Tiles of an XCD (blockIdx mod 8; it owns a contiguous range of n-blocks) are numbered in bands
    // of `mband` m-blocks: m fastest inside the band, then n, then the next band - so the workgroups
    // resident on the XCD at one time hold mband x W/mband tiles (8 x 4 at 2048 tokens and more)
    // rather than one long column, and the operand rows they pull through that XCD's L2 together are
    // as few as that many tiles allow.  With W workgroups on the XCD and T tiles, workgroup j takes
    // tiles j, j + W, ... of the first W floor(T / W); the R = T mod W tiles left over would keep R
    // workgroups busy for a whole tile while the others wait, so where R <= W / 2 each of them is cut
    // along k into up to floor(W / R) parts (wide_cut_parts), one per workgroup.  The workgroups
    // with the lower parts take them first, write their accumulators to their slot of `cut_ws` and
    // count themselves in; the one with the last part - the highest numbered, so the ones it waits
    // for were dispatched before it - takes it after its whole tiles and adds those slots to its
    // accumulators as it runs the epilogue.
    //
    // Copy sources of this wave in a tile: weight tiles 2 wid, 2 wid + 1 of the block; x m-tiles
    // likewise.  Every copy is addressed as a wave-uniform 64-bit base (scalar registers) plus a
    // 32-bit per-lane offset that changes only with the tile: no vector arithmetic and half the
    // address registers per copy instruction.
    struct Tile {
        const unsigned char* wbase[2];
        const unsigned char* xbase[2];
        unsigned xlim[2];   // byte offset of the last row of x inside each of the wave's two m-tiles
        int ks, m0, ntb;
        int nkt;          // k-tiles of this piece of work (a slice, or a part of one)
        int part, parts;  // part of a cut tile (parts == 1: the whole tile or slice)
        int rest;         // cut tile: its number among the XCD's left-over tiles
    };
    const int xcd = blockIdx.x & 7, wj = blockIdx.x >> 3, wgs = gridDim.x >> 3;
    const int owned = max(0, min(a.nb_per_xcd, a.nblocks - xcd * a.nb_per_xcd));   // n-blocks of this XCD
    const int per_slice = owned * a.mblocks;
    const int ntiles = per_slice * a.ksplit;
    const int full = ntiles / wgs, rest = ntiles - full * wgs;
    const int parts = a.cut_ws && full > 0 ? wide_cut_parts(rest, wgs, a.ktps) : 1;
    // An x copy moves rows (lane >> 3) and 8 + (lane >> 3) of an m-tile, 16 bytes per lane: the chunk
    // that lives at position lane & 7 of the row in LDS.  Rows past the end of x repeat its last row.
    const unsigned row_bytes = (unsigned)(a.ldx * 2);
    const unsigned xrow = (unsigned)(lane >> 3) * row_bytes;
    const unsigned xchunk = (unsigned)(((lane & 7) ^ ((lane >> 4) & 7)) * 16);
    auto x_lane = [&](const Tile& tl, int j) __attribute__((always_inline)) {   // j: m-tile 2 wid + (j >> 1), rows 8 (j & 1) ..
        return min(xrow + (unsigned)(j & 1) * 8u * row_bytes, tl.xlim[j >> 1]) + (xchunk ^ ((unsigned)(j & 1) * 64u));
    };
    // piece of work number n of this workgroup -> tile; false past the last one
    auto decode = [&](int n, Tile& tl) __attribute__((always_inline)) -> bool {
        // a workgroup with a piece of the left-over round takes it first - except the last part of a
        // cut tile, which comes after the workgroup's whole tiles: by then the lower parts are done
        const bool has_rest = wj < rest * parts;
        const int my_part = wj - (wj / parts) * parts;
        const bool last_part = has_rest && parts > 1 && my_part == parts - 1;
        const int rest_at = last_part ? full : 0;
        int index, part = 0;
        const bool is_rest = has_rest && n == rest_at;
        if (is_rest) {
            index = full * wgs + wj / parts;
            part = my_part;
        } else {
            const int round = n - (has_rest && n > rest_at ? 1 : 0);
            if (round >= full) return false;
            index = wj + round * wgs;
        }
        const bool is_cut = is_rest && parts > 1;
        const int ks = index / per_slice, idx = index - ks * per_slice;   // k-slice (see the host side), tile
        const int per_band = a.mband * owned;
        const int band = idx / per_band;
        const int rows = min(a.mband, a.mblocks - band * a.mband);
        const int j = idx - band * per_band;
        const int nb = xcd * a.nb_per_xcd + j / rows;
        const int mb = band * a.mband + j % rows;
        tl.nkt = is_cut ? a.ktps / parts : a.ktps;
        tl.part = part;
        tl.parts = is_cut ? parts : 1;
        tl.rest = wj / parts;
        const int kt0 = ks * a.ktps + part * tl.nkt;
        tl.ks = ks;
        tl.m0 = mb * kWM;
        tl.ntb = nb * width;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            tl.wbase[j] = reinterpret_cast<const unsigned char*>(a.w + ((int64_t)min(tl.ntb + 2 * wid + j, a.nt - 1) * a.kt + kt0) * 64);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int first = min(tl.m0 + (2 * wid + j) * 16, a.m - 1);   // first row of the m-tile (clamped)
            tl.xbase[j] = reinterpret_cast<const unsigned char*>(a.x + (int64_t)first * a.ldx + kt0 * 32);
            tl.xlim[j] = (unsigned)min(15, a.m - 1 - first) * row_bytes;
        }
        return true;
    };
    Tile cur, nxt;
    if (!decode(0, cur)) return;
    // with no piece left `nxt` repeats `cur`, whose first k-tiles are then fetched once more into
    // slots nobody reads (the copy counts per k-tile stay fixed)
    int piece = 0;
    bool more = decode(1, nxt);
    if (!more) nxt = cur;
    const unsigned wlane = lane * 16;
    const unsigned x_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)xring;
    const unsigned w_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)wring;
    auto copy = [&](const unsigned char* base, unsigned lane_off, unsigned dst) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                     : : "v"(lane_off), "s"(base), "s"(dst) : "memory");
    };
    // the ring fill in front of a workgroup's first piece (a piece has at least 8 k-tiles)
    auto copy_w = [&](int kt, int slot) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
            copy(cur.wbase[j] + (int64_t)kt * 1024, wlane, w_lds + slot * kWWSlot + (2 * wid + j) * 1024);
    };
    auto copy_x = [&](int st, int slot) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            copy(cur.xbase[j >> 1] + st * 128, x_lane(cur, j), x_lds + slot * kWXSlot + (2 * wid + (j >> 1)) * 2048 + (j & 1) * 1024);
    };
    // fragment reads: per-lane byte offsets inside a slot
    const int xoff0 = (wm * 8) * 2048 + r * 128 + (((0 + q) ^ ((r >> 1) & 7)) * 16);   // kk = 0
    const int xoff1 = (wm * 8) * 2048 + r * 128 + (((4 + q) ^ ((r >> 1) & 7)) * 16);   // kk = 1
    const int woff = wcol * 1024 + lane * 16;
    auto read_frags = [&](bf16x8_t(&wf)[NT], bf16x8_t(&xf)[kPMT], int kt) {
#if defined(MV_EXP_MFMA_ONLY) || defined(MV_EXP_NO_FRAG_READ)
        // timing experiment only: operands made up in registers, nothing read
#pragma unroll
        for (int i = 0; i < NT; ++i) wf[i] = __builtin_bit_cast(bf16x8_t, make_uint4(lane * 77 + kt, i * 0x3f803f80u, kt * 0x01010101u, 0x3c003c00u));
#pragma unroll
        for (int t = 0; t < kPMT; ++t) xf[t] = __builtin_bit_cast(bf16x8_t, make_uint4(lane * 31 + t, kt * 0x00810081u, 0x3f003f00u + t, 0x3d803d80u));
        return;
#endif
        const int st = kt >> 1;
        const unsigned char* wp = wring + (kt & 3) * kWWSlot + woff;
        const unsigned char* xp = xring + (st % 3) * kWXSlot + ((kt & 1) ? xoff1 : xoff0);
#pragma unroll
        for (int i = 0; i < NT; ++i)
            wf[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(wp + i * 1024));
#pragma unroll
        for (int t = 0; t < kPMT; ++t)
            xf[t] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(xp + t * 2048));
    };

    f32x4_t acc[NT][kPMT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int t = 0; t < kPMT; ++t) acc[i][t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    copy_x(0, 0);
    copy_x(1, 1);
    copy_x(2, 2);
#pragma unroll
    for (int j = 0; j < 4; ++j) copy_w(j, j);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    bf16x8_t wA[NT], xA[kPMT], wB[NT], xB[kPMT];
    read_frags(wA, xA, 0);
    __syncthreads();   // every wave holds k-tile 0 before its slot is written again

    // One k-tile = eight groups of four MFMAs (one m-tile each); the wave's copy instructions and its
    // twelve fragment reads for the next k-tile are dealt out between the groups instead of standing
    // in front of them: a copy instruction holds the wave's issue for 60-180 cycles and a burst of
    // fragment reads for about as long, and with both waves of a SIMD in step (the barrier keeps
    // them so) the matrix pipe idles through every such block; between short MFMA groups the
    // partner's MFMAs fill it.
    // `kt` counts inside the tile, `g` over all k-tiles of this workgroup (the ring slots go by g);
    // what lies past the end of the tile is taken from the start of the next one
    auto copy_piece = [&](int g, int kt, bool odd, int p) {
        // pieces of this k-tile in issue order: on odd k-tiles the four x pieces first
        const int nx = odd ? 4 : 0;
        if (p < nx) {
            const int s3 = (kt >> 1) + 3, slot = (g >> 1) % 3, nstage = cur.nkt >> 1;
            const bool over = s3 >= nstage;
            const unsigned char* base = over ? nxt.xbase[p >> 1] : cur.xbase[p >> 1];
            const int st = over ? s3 - nstage : s3;
            copy(base + st * 128, x_lane(over ? nxt : cur, p),
                 x_lds + slot * kWXSlot + (2 * wid + (p >> 1)) * 2048 + (p & 1) * 1024);
        } else if (p < nx + 2) {
            const int j = p - nx, k4 = kt + 4;
            const bool over = k4 >= cur.nkt;
            const unsigned char* base = over ? nxt.wbase[j] : cur.wbase[j];
            const int ktc = over ? k4 - cur.nkt : k4;
            copy(base + (int64_t)ktc * 1024, wlane, w_lds + (g & 3) * kWWSlot + (2 * wid + j) * 1024);
        }
    };
    auto ktile = [&](bf16x8_t(&wc)[NT], bf16x8_t(&xc)[kPMT], bf16x8_t(&wnx)[NT], bf16x8_t(&xnx)[kPMT], int g,
                     int kt, bool odd) {
        const int gn = g + 1;
        const unsigned char* wp = wring + (gn & 3) * kWWSlot + woff;
        const unsigned char* xp = xring + ((gn >> 1) % 3) * kWXSlot + ((gn & 1) ? xoff1 : xoff0);
#pragma unroll
        for (int t = 0; t < kPMT; ++t) {
            __builtin_amdgcn_sched_barrier(0);
#ifndef MV_EXP_NO_COPY
            if (t < 6) copy_piece(g, kt, odd, t);
#endif
            __builtin_amdgcn_sched_barrier(0);
#ifndef MV_EXP_NO_FRAG_READ
            if (t < 2) {
                wnx[2 * t] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(wp + (2 * t) * 1024));
                if (2 * t + 1 < NT)
                    wnx[2 * t + 1] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(wp + (2 * t + 1) * 1024));
            } else if (t < 6) {
                xnx[2 * t - 4] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(xp + (2 * t - 4) * 2048));
                xnx[2 * t - 3] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(xp + (2 * t - 3) * 2048));
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NT; ++i)
                MV_MFMA(acc[i][t], wc[i], xc[t]);
        }
        __builtin_amdgcn_sched_barrier(0);
        // this wave's pieces of k-tile kt + 2 have landed (8 younger copies may be outstanding), and
        // its fragment reads of kt + 1 are complete before that slot can be written again
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
#ifndef MV_EXP_NO_BARRIER
        __syncthreads();
#endif
    };
    [[maybe_unused]] int tile_no = 0;
    for (int g0 = 0;;) {
        const int nkt = cur.nkt;
        for (int kt = 0; kt < cur.nkt; kt += 2) {   // a piece is a multiple of 64 k (checked on the host)
            MV_WSTAMP(g0 + kt);
            ktile(wA, xA, wB, xB, g0 + kt, kt, false);
            ktile(wB, xB, wA, xA, g0 + kt + 1, kt + 1, true);
        }
        MV_WSTAMP(g0 + nkt);
#ifdef MV_EXP_NO_STORE
        // timing experiment only: the accumulators are consumed, nothing is written
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int t = 0; t < kPMT; ++t) asm volatile("" : : "v"(acc[i][t]));
#else
        if (cur.parts > 1 && cur.part + 1 < cur.parts) {
            // a lower part of a cut tile: the accumulators as they lie in the registers (1 KiB per
            // register and wave), then one count per workgroup once every wave's stores are visible
            // (a wave-uniform base and a 32-bit lane offset: per-lane 64-bit addresses for 32 stores
            // that lie further apart than an instruction's offset field reaches cost 64 registers)
            size_t my_offset = (size_t)blockIdx.x * kWSlotBytes + (size_t)wid * 32768;
            asm volatile("" : "+s"(my_offset));   // formed here: 32 addresses hoisted out of the loop over pieces would be held, or spilled, throughout
            unsigned char* const mine = reinterpret_cast<unsigned char*>(a.cut_ws) + my_offset;
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int t = 0; t < kPMT; ++t)
                    asm volatile("global_store_dwordx4 %0, %1, %2 sc1" : : "v"(wlane), "v"(acc[i][t]), "s"(mine + (i * 8 + t) * 1024) : "memory");
            // Hand-off without fences: sc1 (write-through) stores, each writing whole 128-byte lines;
            // every wave waits for its own; behind the barrier one lane counts the workgroup in with a
            // device-scope atomic; the reader polls the counter with an sc1 load, passes a barrier and
            // reads the sums with sc1 loads only.  A device-wide release / acquire pair here
            // (__threadfence) writes back the XCD's whole L2 and empties every L1: measured 227 -> 305 us.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0)
                __hip_atomic_fetch_add(&a.cut_count[xcd * wgs + cur.rest], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            // The last part of a cut tile adds the lower parts as it runs the epilogue.  Their
            // workgroups took them first and were dispatched before this one, so the wait is over
            // before it begins unless this launch shares the GPU; the counter is left at zero for the
            // next launch.  (Adding them into the accumulators here instead, under this condition,
            // makes the compiler keep a second set of the 128 registers and move and spill between the
            // two at every tile: with the path compiled in and never taken, 4096^3 119 -> 144 us.)
            const unsigned char* lower = nullptr;
            int nlower = 0;
            if (cur.parts > 1) {
                if (threadIdx.x == 0) {
                    // bounded (about 20 s of 0.6-us naps): parts that never arrive mean a lost workgroup,
                    // and a trap that ends the process beats sums that silently lack a part of k
                    unsigned* const count = &a.cut_count[xcd * wgs + cur.rest];
                    int spin = 0;
                    while (__hip_atomic_load(count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(cur.parts - 1)) {
                        if (++spin > (1 << 25)) __builtin_trap();
                        __builtin_amdgcn_s_sleep(16);
                    }
                    __hip_atomic_store(count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __syncthreads();
                asm volatile("" ::: "memory");
                lower = reinterpret_cast<const unsigned char*>(a.cut_ws) + (size_t)(((cur.rest * cur.parts) << 3) | xcd) * kWSlotBytes +
                        (size_t)wid * 32768;
                nlower = cur.parts - 1;
            }
            if (cur.ntb + wcol < a.nt)
                store_wave_tile<NT>(a, acc, cur.ntb + wcol, cur.m0 + wm * 128, cur.ks, r, q, lower, nlower);
        }
#endif
        MV_WSTAMP(g0 + nkt + 1);
        if (!more) break;
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int t = 0; t < kPMT; ++t) acc[i][t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        g0 += nkt;
        cur = nxt;
        ++piece;
        more = decode(piece + 1, nxt);
        if (!more) nxt = cur;
        ++tile_no;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // nothing may land in LDS after this workgroup is gone
}

// Tile width: wave column c of wave row wm takes NTA n-tiles when c + wm is even and NTB otherwise, so
// a block is 2 (NTA + NTB) strips wide - 16, 14 or 12 - and the two waves that share a SIMD (same
// column, the two rows) always have NTA + NTB n-tiles between them: the matrix pipes stay evenly
// loaded at the odd width too.  Both kinds of wave run the same copies, waits and barriers per
// k-tile; only the number of fragments and MFMAs differs.
template <int NTA, int NTB>
__global__ __launch_bounds__(512, 1) void gemm_prefill_wide_kernel(PGemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wn = wid & 3, wm = wid >> 2;
    constexpr int kWidth = 2 * (NTA + NTB);
    if constexpr (NTA == NTB) {
        wide_tile<NTA>(a, lds, wid, wm, wn * NTA, kWidth);
    } else {
        // row 0: A B A B, row 1: B A B A
        const int first = wm ? NTB : NTA;
        const int wcol = (wn >> 1) * (NTA + NTB) + ((wn & 1) ? first : 0);
        if (((wn + wm) & 1) == 0)
            wide_tile<NTA>(a, lds, wid, wm, wcol, kWidth);
        else
            wide_tile<NTB>(a, lds, wid, wm, wcol, kWidth);
    }
}

}  // namespace mv

#ifdef MV_GEMM_WSTAMP
extern "C" int mv_debug_read_wide_stamps(void* dst, size_t bytes) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(mv::g_wstamp), bytes) == hipSuccess ? 0 : -3;
}
#endif
#ifdef MV_GEMM_STAMP
extern "C" int mv_debug_read_stamps(void* dst, size_t bytes) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(mv::g_stamp), bytes) == hipSuccess ? 0 : -3;
}
#endif

namespace mv {
// k-slices for an M x N product over kt k-tiles: none when the output tiles alone fill the chip.
// Otherwise a small cost model picks the count (measured on MI355X: a workgroup walks k at about
// 1.1 us per 64-k stage when the chip is mostly idle; the slices cost a second launch, about 4 us,
// plus writing and re-reading ks x M x N floats at about 3 TB/s).  One 256 x 128 tile per
// workgroup walks k serially: at 128 tokens the QKV projection of the 8B shape has 24 tiles and
// took 70 us with nine tenths of the chip idle, 29 us split; the gate/up projection at 256 tokens
// (224 tiles) is faster unsplit because its slices would be 117 MB.
static int prefill_ksplit(int m, int n, int kt) {
    const int cus = num_cus();
    const long tiles = (long)((m + kPM - 1) / kPM) * ((n + kPN - 1) / kPN);
    if (tiles >= cus) return 1;
    int best = 1;
    double best_us = 1e30;
    for (int ks = 1; ks <= 16; ks *= 2) {
        if (ks > 1 && (kt % (2 * ks) != 0 || kt / ks < 8)) break;
        const long rounds = (tiles * ks + 2L * cus - 1) / (2L * cus);
        double us = (double)rounds * (kt / (2.0 * ks)) * 1.1;
        if (ks > 1) us += 4.0 + 8.0 * ks * (double)m * n / 3.0e6;
        if (us < best_us) {
            best_us = us;
            best = ks;
        }
    }
    return best;
}

// Which form a product takes and in how many k-slices.  The wide form (256 x 256 tiles, one
// workgroup per CU) when the 128 x 256 form would not cut k and the wide tiles fill at least half the
// CUs; and, in TWO k-slices, where its tiles fill three eighths to half of the CUs and k is long (the
// down projection at 1536-2048 tokens: 96-128 tiles of 224 stages; one workgroup per CU then walks
// half the stages, and only a long k pays for the 8 bytes per output the slices write and read back
// plus the reduction launch: 2048 tokens 268 -> 216 us; at 2304 tokens, 144 tiles, 344 against 300).
// Since the slab reduction reads 16 bytes a lane (15 us for 67 MB of slabs where it took 43) the same
// holds from k = 4096: the O projection at 1536 tokens 85 -> 67 us, at 2048 tokens 93 -> 89.
// MVLLM_PREFILL_WIDE = 0 / 1 forces the choice of form (read once).
//
// Width of a wide tile: 16 strips (256 columns; four n-tiles per wave), 14 (224; four and three) or
// 12 (192; three).  A CU holds one tile at a time and an XCD's 32 CUs take that XCD's tiles in
// rounds, so a product costs rounds x tile width; a narrower tile does more operand traffic per
// multiply-add and has fewer MFMAs to hide each copy and fragment read behind (priced at 12 %: the
// gate/up projection at 2048 tokens, four rounds of 256 columns or five of 192, measured 425 against
// 441 us) and wins where it fills a round the wider one leaves part empty - the QKV projection of the
// 8B shape (6144 columns) at 4096 tokens is 384 tiles of 256 columns, two rounds with the second half
// empty, or 512 tiles of 192 in two full rounds of three quarters the length: 194 -> 171 us; at 1536
// tokens 106 -> 95.  The 224-column tile gains less than its width promises (priced at 10 %): the wave
// with three n-tiles waits at every barrier for its SIMD partner with four, which then runs without a
// partner to fill its issue gaps - gate/up (28672 columns = 128 such tiles per m-block) at 2048 tokens
// 419 -> 415 us, at 3072 tokens 630 -> 605.  MVLLM_PREFILL_WIDE_STRIPS = 12 / 14 / 16 forces the
// width (read once).
struct PrefillPlan {
    bool wide;
    int ks;
    int strips;   // wide form: 16-column strips per tile (16, 14 or 12)
};
// MVLLM_PREFILL_PERSIST = 0 launches one wide-form workgroup per tile instead of one per CU (read once)
static bool wide_persistent() {
    static const bool on = [] {
        const char* e = getenv("MVLLM_PREFILL_PERSIST");
        return !(e && *e) || atoi(e) != 0;
    }();
    return on;
}
// workgroups of a wide-form launch over `ids` tile slots (8 x the slots of the fullest XCD)
static long wide_grid(long ids) {
    const long resident = std::max(8, num_cus() / 8 * 8);
    return wide_persistent() ? std::min(ids, resident) : ids;
}
// MVLLM_PREFILL_CUT = 0: left-over tiles are never cut along k inside the launch (read once)
static int g_debug_cut = -1, g_debug_strips = -1;   // mv_debug_prefill_options: overrides for A/B runs in one process
static bool wide_cut_enabled() {
    static const bool on = [] {
        const char* e = getenv("MVLLM_PREFILL_CUT");
        return !(e && *e) || atoi(e) != 0;
    }();
    return (g_debug_cut >= 0 ? g_debug_cut != 0 : on) && wide_persistent();
}
// Does any XCD of an unsliced wide-form launch cut its left-over tiles?  (Then the launch needs one
// accumulator slot per workgroup.)
static bool wide_cuts(int mblocks, int nblocks, int kt) {
    if (!wide_cut_enabled()) return false;
    const int per_xcd = (nblocks + 7) / 8;
    const long grid = wide_grid(8L * per_xcd * mblocks);
    const int wgs = (int)(grid / 8);
    for (int x = 0; x < 8; ++x) {
        const int owned = std::max(0, std::min(per_xcd, nblocks - x * per_xcd));
        const int ntiles = owned * mblocks;
        if (ntiles >= wgs && wide_cut_parts(ntiles % wgs, wgs, kt) > 1) return true;
    }
    return false;
}
// Length of an unsliced wide-form launch in whole tiles (the fullest XCD sets it)
static double wide_rounds(int mblocks, int nblocks, int kt) {
    const int per_xcd = (nblocks + 7) / 8;
    const long grid = wide_grid(8L * per_xcd * mblocks);
    const int wgs = (int)(grid / 8);
    const int ntiles = per_xcd * mblocks;
    const int rest = ntiles % wgs;
    const int parts = wide_cut_enabled() && ntiles >= wgs ? wide_cut_parts(rest, wgs, kt) : 1;
    // a cut costs a quarter of a tile beyond the parts themselves (the slots written and read, two
    // epilogues and a second ring fill per tile: 4096 x 6144 x 4096 in 256-column tiles, 2 rounds
    // 196 us, 1.5 rounds 178)
    return ntiles / wgs + (rest == 0 ? 0.0 : parts == 1 ? 1.0 : 1.0 / parts + 0.125 * (parts - 1) + 0.25);
}
static int wide_tile_strips(int m, int n, int kt) {
    static const int forced = [] {
        const char* e = getenv("MVLLM_PREFILL_WIDE_STRIPS");
        return e && *e ? atoi(e) : 0;
    }();
    if (g_debug_strips == 12 || g_debug_strips == 14 || g_debug_strips == 16) return g_debug_strips;
    if (forced == 12 || forced == 14 || forced == 16) return forced;
    const int mblocks = (m + kWM - 1) / kWM;
    const int nt = n / 16;
    int best = 16;
    double best_cost = 1e30;
    for (int w = 16; w >= 12; w -= 2) {
        const int nblocks = (nt + w - 1) / w;
        const double cost = wide_rounds(mblocks, nblocks, kt) * w * (w == 16 ? 1.0 : w == 14 ? 1.10 : 1.12);
        if (cost < best_cost) {
            best_cost = cost;
            best = w;
        }
    }
    return best;
}
static PrefillPlan prefill_plan(int m, int n, int kt) {
    static const int forced = [] {
        const char* e = getenv("MVLLM_PREFILL_WIDE");
        return e && *e ? atoi(e) : -1;
    }();
    const int narrow_ks = prefill_ksplit(m, n, kt);
    if (kt < 8 || forced == 0) return {false, narrow_ks, 16};
    const long tiles = (long)((m + kWM - 1) / kWM) * ((n + kWN - 1) / kWN);
    const int cus = num_cus();
    if (forced < 0 && kt >= 128 && kt % 4 == 0 && 8 * tiles >= 3L * cus && 2 * tiles <= cus) return {true, 2, 16};
    if (narrow_ks == 1 && (forced == 1 || 2 * tiles >= cus)) return {true, 1, wide_tile_strips(m, n, kt)};
    return {false, narrow_ks, 16};
}
}  // namespace mv

// Experiments only (tools/ab_gemm_prefill.py): whether left-over tiles are cut (0 / 1) and the tile width
// in strips (12 / 14 / 16) for the calls that follow; -1 returns a choice to the environment and the plan.
extern "C" void mv_debug_prefill_options(int cut, int strips) {
    mv::g_debug_cut = cut;
    mv::g_debug_strips = strips;
}

extern "C" size_t mv_gemm_prefill_workspace_bytes(int m, int n, int k) {
    using namespace mv;
    if (m <= 0 || n <= 0 || k <= 0 || (n % 16) != 0 || (k % 64) != 0) return 0;
    const PrefillPlan plan = prefill_plan(m, n, k / 32);
    if (plan.ks > 1) return kWCountBytes + (size_t)plan.ks * m * n * sizeof(float);
    if (!plan.wide) return 0;
    const int mblocks = (m + kWM - 1) / kWM, nblocks = (n / 16 + plan.strips - 1) / plan.strips;
    if (!wide_cuts(mblocks, nblocks, k / 32)) return 0;
    return kWCountBytes + (size_t)wide_grid(8L * ((nblocks + 7) / 8) * mblocks) * kWSlotBytes;
}

extern "C" int mv_gemm_prefill_form(int m, int n, int k) {
    using namespace mv;
    if (m <= 0 || n <= 0 || k <= 0 || (n % 16) != 0 || (k % 64) != 0) return 0;
    const PrefillPlan plan = prefill_plan(m, n, k / 32);
    return !plan.wide ? 0 : plan.ks > 1 || plan.strips == 16 ? 1 : plan.strips == 12 ? 2 : 3;
}

extern "C" int mv_gemm_prefill(const void* x, int64_t ldx, const void* w_tiles, const void* bias,
                               const void* residual, int64_t ld_res, void* y, int64_t ldy, int m,
                               int n, int k, int flags, mv_stream_t stream) {
    return mv_gemm_prefill_ws(x, ldx, w_tiles, bias, residual, ld_res, y, ldy, m, n, k, flags, nullptr, 0,
                              stream);
}

extern "C" int mv_gemm_prefill_ws(const void* x, int64_t ldx, const void* w_tiles, const void* bias,
                                  const void* residual, int64_t ld_res, void* y, int64_t ldy, int m,
                                  int n, int k, int flags, void* workspace, size_t workspace_bytes,
                                  mv_stream_t stream) {
    using namespace mv;
    if (m == 0) return MV_OK;
    if (!x || !w_tiles || !y || m < 0 || n <= 0 || k <= 0) return MV_ERR_INVALID_ARG;
    if ((n % 16) != 0 || (k % 64) != 0) return MV_ERR_UNSUPPORTED;
    if (!aligned16(x) || !aligned16(w_tiles) || (ldx % 8) != 0 || ldx < k) return MV_ERR_INVALID_ARG;
    const bool sw = (flags & MV_GEMM_SWIGLU) != 0, of32 = (flags & MV_GEMM_OUT_F32) != 0;
    if (sw && (bias || residual || of32)) return MV_ERR_UNSUPPORTED;
    if (of32 && (bias || residual)) return MV_ERR_UNSUPPORTED;
    const int ncols = sw ? n / 2 : n;
    if (ldy < ncols || (ldy % 4) != 0) return MV_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(y) & (of32 ? 15 : 7)) != 0) return MV_ERR_INVALID_ARG;
    if (residual && ld_res < n) return MV_ERR_INVALID_ARG;

    PGemmArgs a;
    a.x = (const uint16_t*)x;
    a.ldx = ldx;
    a.w = (const uint4*)w_tiles;
    a.bias = (const uint16_t*)bias;
    a.residual = (const uint16_t*)residual;
    a.ld_res = ld_res;
    a.y = y;
    a.ldy = ldy;
    a.m = m;
    a.n = n;
    a.k = k;
    a.nt = n / 16;
    a.kt = k / 32;
    a.mblocks = (m + kPM - 1) / kPM;
    a.nblocks = (n + kPN - 1) / kPN;
    a.nb_per_xcd = (a.nblocks + 7) / 8;
    a.mband = 1;
    a.flags = flags;
    const PrefillPlan plan = prefill_plan(m, n, a.kt);
    a.ksplit = plan.ks;
    a.partial = nullptr;
    a.cut_ws = nullptr;
    a.cut_count = nullptr;
    // the workspace: kWCountBytes of counters, then the data
    void* const data = workspace && workspace_bytes > kWCountBytes && aligned16(workspace)
                           ? (void*)((char*)workspace + kWCountBytes) : nullptr;
    const size_t data_bytes = data ? workspace_bytes - kWCountBytes : 0;
    bool wide = plan.wide;
    if (a.ksplit > 1) {
        const size_t need = (size_t)a.ksplit * m * n * sizeof(float);
        if (data && data_bytes >= need) {
            a.partial = (float*)data;
        } else {
            a.ksplit = 1;  // no room for the slices: whole k per tile (slower, same result class)
            wide = false;
        }
    }
    a.ktps = a.kt / a.ksplit;
    if (wide) {
        static LdsOptIn lds_done[3];
        const int strips = a.ksplit == 1 ? plan.strips : 16;
        const void* kernel = strips == 12   ? (const void*)gemm_prefill_wide_kernel<3, 3>
                             : strips == 14 ? (const void*)gemm_prefill_wide_kernel<4, 3>
                                            : (const void*)gemm_prefill_wide_kernel<4, 4>;
        int rc = ensure_dynamic_lds(kernel, kWLds, lds_done[(strips - 12) / 2]);
        if (rc != MV_OK) return rc;
        a.mblocks = (m + kWM - 1) / kWM;
        a.nblocks = (a.nt + strips - 1) / strips;
        a.nb_per_xcd = (a.nblocks + 7) / 8;
        a.mband = a.mblocks < 8 ? a.mblocks : 8;
        if (const char* e = getenv("MVLLM_PREFILL_MBAND")) {   // experiments: 0 = one band (the whole column)
            const int v = atoi(e);
            a.mband = v <= 0 || v > a.mblocks ? a.mblocks : v;
        }
        const long ids = 8L * a.nb_per_xcd * a.mblocks * a.ksplit;
        if (ids > 0x7fffffffL) return MV_ERR_UNSUPPORTED;
        // one workgroup per CU takes its share of every XCD's tiles (a multiple of 8 workgroups: a
        // workgroup's XCD is its number mod 8)
        const long wgrid = wide_grid(ids);
        // left-over tiles are cut along k inside the launch when the caller's workspace holds a slot per
        // workgroup; without it they are taken whole (slower, same result class)
        if (a.ksplit == 1 && wgrid <= kWMaxGrid && wide_cuts(a.mblocks, a.nblocks, a.kt) && data &&
            data_bytes >= (size_t)wgrid * kWSlotBytes) {
            a.cut_ws = (float*)data;
            a.cut_count = (unsigned*)workspace;
        }
        if (strips == 12)
            hipLaunchKernelGGL((gemm_prefill_wide_kernel<3, 3>), dim3((unsigned)wgrid), dim3(512), kWLds, (hipStream_t)stream, a);
        else if (strips == 14)
            hipLaunchKernelGGL((gemm_prefill_wide_kernel<4, 3>), dim3((unsigned)wgrid), dim3(512), kWLds, (hipStream_t)stream, a);
        else
            hipLaunchKernelGGL((gemm_prefill_wide_kernel<4, 4>), dim3((unsigned)wgrid), dim3(512), kWLds, (hipStream_t)stream, a);
        rc = check_launch();
        if (rc != MV_OK || a.ksplit == 1) return rc;
        return launch_wgemm_finalize(a.partial, a.ksplit, m, n, flags, a.bias, a.residual, ld_res, y, ldy,
                                     (hipStream_t)stream);
    }
    const long grid = 8L * a.nb_per_xcd * a.mblocks * a.ksplit;
    if (grid > 0x7fffffffL) return MV_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(gemm_prefill_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
    int rc = check_launch();
    if (rc != MV_OK || a.ksplit == 1) return rc;
    return launch_wgemm_finalize(a.partial, a.ksplit, m, n, flags, a.bias, a.residual, ld_res, y, ldy,
                                 (hipStream_t)stream);
}
