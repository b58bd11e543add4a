// Weight-streaming skinny GEMM for decode:  y[M,N] = x[M,K] @ W[N,K]^T,  1 <= M <= 64.
//
// The op is HBM-bound on W (2*N*K bytes read once; x and y are M/N-th of that), so
// the whole design serves one goal: every CU keeps >= 64 KiB of perfectly coalesced
// weight loads in flight, and every CU gets the same number of bytes.
//
//  * W is stored as wave tiles (mv_shuffle_weight): tile (nt,kt) = 16 output rows x
//    32 k = 1 KiB, lane l owning W[16nt + (l&15)][32kt + 8(l>>4) + 0..7].  One
//    global_load_dwordx4 per wave moves one tile and leaves it already laid out as
//    the A operand of v_mfma_f32_16x16x32_bf16 (rows on lanes&15, k on lanes>>4).
//    The tiles of one 16-row strip are contiguous along K.
//  * A unit of work is (strip nt, k-slice ks).  A 512-thread workgroup takes whole
//    units; its 8 waves interleave over the unit's tiles (wave w takes tile kt0+w,
//    kt0+w+8, ...) so the workgroup reads one contiguous stream.  Partial sums of
//    the 8 waves meet in LDS in a fixed order: results are run-to-run identical.
//  * M <= 4: dot-product form (v_dot2c_f32_bf16), x staged once per workgroup in
//    LDS.  M > 4: v_mfma_f32_16x16x32_bf16 with the x fragment (B operand) loaded
//    from L2 in the same lane layout.  Both read the same weight tiles.
//  * ksplit > 1 (picked when strips alone cannot balance the CUs) writes fp32
//    slabs; a small second kernel adds them in slab order and applies the epilogue.
//
// Epilogue rounding follows the reference's op boundaries: t = bf16(acc);
// t = bf16(t + bias); t = bf16(t + residual); SWIGLU: out = bf16(silu(gate_t) * up_t).
#include "common.h"

#include <algorithm>

#include <atomic>

namespace mv {

constexpr int kGW = 8;           // waves per workgroup, MFMA form
constexpr int kGT = kGW * kWave;
constexpr int kMaxKsplit = 8;

// Experiment knobs (mv_wgemm_tune).  Atomics: a thread may set one while another is inside a call.
static std::atomic<int> g_tune_bpc{0};      // workgroups per CU the grid is sized for (0 = planner)
static std::atomic<int> g_tune_waves{0};    // waves per workgroup of the dot form: 8 or 16 (0 = planner)
static std::atomic<int> g_tune_sg{0};       // strips per workgroup of the MFMA form: 1, 2 or 4 (0 = planner)
static std::atomic<int> g_tune_ksplit{0};   // 0 = heuristic
static std::atomic<int> g_tune_contig{0};   // dot form: consecutive tiles per wave

enum { kProNone = 0, kProNorm = 1, kProAddNorm = 2 };

struct GemmArgs {
    const uint16_t* x;
    int64_t ldx;
    const uint4* w;
    const uint16_t* bias;
    const uint16_t* residual;
    int64_t ld_res;
    void* y;
    int64_t ldy;
    float* partial;  // [ksplit][M][N] fp32 when ksplit > 1
    int m, n, k;
    int nt, kt;      // strips, k-tiles per strip
    int ksplit, ktps;  // k-slices, k-tiles per slice
    int contig;        // dot form: a wave's tiles are consecutive (1) or interleaved with the other waves' (0)
    int nunits;
    int flags;
    // fused input transform (dot form only): x <- rmsnorm(x [+ add_r]) * ln_w
    int prologue;
    int var_from_rounded;       // kProAddNorm: variance from bf16(x + add_r) instead of the fp32 sum
    const uint16_t* add_r;
    int64_t ld_add_r;
    const uint16_t* ln_w;
    float eps;
    uint16_t* sum_out;          // kProAddNorm: bf16(x + add_r), written by workgroup 0
    int64_t ld_sum_out;
    // output transform of the slab reduction: normed = rmsnorm(y) * post_ln_w next to y itself
    const uint16_t* post_ln_w;
    uint16_t* post_out;
    int64_t ld_post;
    float post_eps;
    int post_var_from_rounded;  // variance from y as stored (bf16) instead of the unrounded t + residual
#ifdef MV_WGEMM_STAMP
    int stamp_slot;             // which of the 32 stamp records this launch writes
#endif
};

#ifdef MV_WGEMM_STAMP
// Phase stamps (build variant for tools/stamp_wgemm.py): the constant 100 MHz counter at up to
// eight points of every workgroup of the last 32 launches, written by the workgroup's first thread.
static __device__ unsigned long long g_wstamp[32 * 256 * 8];
static int g_stamp_counter = 0;
#define MV_WSTAMP(slot)                                                                          \
    do {                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        if (threadIdx.x == 0 && blockIdx.x < 256)                                                \
            g_wstamp[((a.stamp_slot & 31) * 256 + blockIdx.x) * 8 + (slot)] =                    \
                __builtin_amdgcn_s_memrealtime();                                                \
        __builtin_amdgcn_sched_barrier(0);                                                       \
    } while (0)
#else
#define MV_WSTAMP(slot) do {} while (0)
#endif

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;

// Final value of one output element from its fp32 accumulator.
__device__ __forceinline__ float epilogue_value(const GemmArgs& a, float acc, int m, int n) {
    float t = round_bf16(acc);
    if (a.bias) t = round_bf16(t + bf16_bits_to_f32(a.bias[n]));
    if (a.residual) t = round_bf16(t + bf16_bits_to_f32(a.residual[(int64_t)m * a.ld_res + n]));
    return t;
}

// ------------------------------------------------------------ M <= 4: dot form
struct UnitCtx {
    int nt, ks, kt0, ntile, nchunk, last, wfirst;
    const uint4* wbase;
};

// GW waves per workgroup, each keeping two stages of U weight tiles in flight; PRO = the
// input transform (kProNone / kProNorm / kProAddNorm).
template <int M, int GW, int U, int PRO>
__global__ __launch_bounds__(GW * kWave) void wgemm_dot_kernel(GemmArgs a) {
    constexpr int GT = GW * kWave;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // layout: x rows [M][K] bf16 | 16 zero bytes | red [GW][M][16] fp32
    uint16_t* xs = reinterpret_cast<uint16_t*>(smem);
    const int zero_off = M * a.k * 2;
    float* red = reinterpret_cast<float*>(smem + zero_off + 16);

    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, q = lane >> 4;
    MV_WSTAMP(0);

    const int tstride = a.contig ? 1 : GW;   // tiles between two consecutive tiles of a wave
    const int wstride = tstride * 64;        // the same in 16-byte units
    auto setup = [&](int u) {
        UnitCtx c;
        c.nt = u / a.ksplit;
        c.ks = u - c.nt * a.ksplit;
        c.kt0 = c.ks * a.ktps;
        const int kt1 = min(a.kt, c.kt0 + a.ktps);
        if (a.contig) {  // wave w owns tiles [w * tpw, (w + 1) * tpw) of the slice
            const int tpw = (kt1 - c.kt0 + GW - 1) / GW;
            c.wfirst = wid * tpw;
            c.ntile = min(tpw, kt1 - c.kt0 - c.wfirst);
        } else {         // wave w owns tiles w, w + GW, w + 2 GW, ...
            c.wfirst = wid;
            c.ntile = (kt1 - c.kt0 - wid + GW - 1) / GW;  // tiles of this wave (may be <= 0)
        }
        c.nchunk = (c.ntile + U - 1) / U;
        c.last = max(c.ntile - 1, 0);
        // a wave without tiles still issues (clamped) loads: point it at the slice's first tile
        c.wbase = a.w + ((int64_t)c.nt * a.kt + c.kt0 + (c.ntile > 0 ? c.wfirst : 0)) * 64 + lane;
        return c;
    };
    auto load = [&](uint4(&buf)[U], const UnitCtx& c, int chunk) {
#pragma unroll
        for (int i = 0; i < U; ++i) {
            const int t = min(chunk * U + i, c.last);  // clamp: always a valid address (chunk < 2^24)
            buf[i] = ld16_nt(c.wbase + (int64_t)t * wstride);
        }
    };
    auto compute = [&](const uint4(&buf)[U], const UnitCtx& c, int chunk, float(&acc)[M]) {
#pragma unroll
        for (int i = 0; i < U; ++i) {
            const int t = chunk * U + i;
            // byte offset of this lane's 8 x values; out-of-range tiles read zeros
            const int xoff = (t < c.ntile) ? ((c.kt0 + c.wfirst + t * tstride) * 32 + q * 8) * 2 : -1;
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const int off = (xoff < 0) ? zero_off : m * a.k * 2 + xoff;
                const uint4 xv = *reinterpret_cast<const uint4*>(smem + off);
                acc[m] = dot2_bf16(buf[i].x, xv.x, acc[m]);
                acc[m] = dot2_bf16(buf[i].y, xv.y, acc[m]);
                acc[m] = dot2_bf16(buf[i].z, xv.z, acc[m]);
                acc[m] = dot2_bf16(buf[i].w, xv.w, acc[m]);
            }
        }
    };

    // Memory requests of a wave return in issue order, and two stages of weight tiles per wave
    // are a couple of microseconds of HBM traffic: x (and the norm weight) are therefore
    // requested FIRST, into registers, the weight stream is started behind them, and only
    // then is x consumed.  Requested after the tiles, x would arrive behind them and the
    // stream would stall until the input transform is through.
    // Units go to workgroups by fixed stride.  Handing them out by ticket instead (an atomic
    // counter, drawn one unit ahead so that its round trip is hidden) was measured on MI355X to be
    // slower although workgroups on the odd XCDs finish a gate/up launch about 12 % later than
    // those on the even ones: 37.3 -> 41.1 us for N = 28672, 153 -> 166 us for N = 128256.
    int u = blockIdx.x;
    UnitCtx cur = setup(u);
    uint4 bufA[U], bufB[U];
    const int kvec = a.k >> 3;

    constexpr int kPT = 256;                              // threads of the input transform
    constexpr int XC = 2;                                 // early chunks per thread, plain staging
    constexpr int PC = (M == 1) ? 4 : (M == 2 ? 2 : 1);   // early chunks per thread and row, norm
    // Straight-line code with clamped addresses (no branches around the loads), so the compiler
    // can wait for exactly the x loads and leave the tile loads in flight.
    uint4 xe[XC];
    uint4 pxe[M][PC], pre[M][PC], pwe[PC];
    if (PRO == kProNone) {
#pragma unroll
        for (int j = 0; j < XC; ++j) {
            const int c = min((int)threadIdx.x + j * GT, M * kvec - 1);
            const int m = c / kvec, cv = c - m * kvec;
            xe[j] = ld16(a.x + (int64_t)m * a.ldx + cv * 8);
        }
    } else {
#pragma unroll
        for (int j = 0; j < PC; ++j) {
            if (j * kPT < kvec) {  // uniform
                const int c = min((int)(threadIdx.x & (kPT - 1)) + j * kPT, kvec - 1);
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    pxe[m][j] = ld16(a.x + (int64_t)m * a.ldx + c * 8);
                    if (PRO == kProAddNorm)
                        pre[m][j] = ld16(a.add_r + (int64_t)m * a.ld_add_r + c * 8);
                }
                pwe[j] = ld16(a.ln_w + c * 8);
            }
        }
    }
    // Keep the issue order x, A0..A3 (B0..B3 follow the transform): the loop below waits by
    // counting requests, and a first pass issued in any other order would make every later
    // wait conservative.
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < U; ++i) {
        const int t = min(i, cur.last);  // indices past the end are clamped to the wave's last tile
        const uint4 v = ld16_nt(cur.wbase + (int64_t)t * wstride);
        bufA[i] = v;
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("" ::: "memory");

    // ---- stage x into LDS, optionally as rmsnorm(x [+ add_r]) * ln_w ------------------
    if (PRO == kProNone) {
#pragma unroll
        for (int j = 0; j < XC; ++j) {
            const int c = threadIdx.x + j * GT;
            if (c < M * kvec) {
                const int m = c / kvec, cv = c - m * kvec;
                st16(xs + m * a.k + cv * 8, xe[j]);
            }
        }
        for (int c = threadIdx.x + XC * GT; c < M * kvec; c += GT) {
            const int m = c / kvec, cv = c - m * kvec;
            st16(xs + m * a.k + cv * 8, ld16(a.x + (int64_t)m * a.ldx + cv * 8));
        }
        if (threadIdx.x == 0) *reinterpret_cast<uint4*>(smem + zero_off) = make_uint4(0, 0, 0, 0);
    } else {
        // Same chunk assignment (256 threads, stride 256) and the same reduction order as the
        // stand-alone norm kernels, so fused and unfused paths agree bit for bit.
        float ss[M];
#pragma unroll
        for (int m = 0; m < M; ++m) ss[m] = 0.f;
        // one chunk of one row: (sum,) squares, LDS image, copy of the sum for the next block
        auto absorb = [&](int m, int c, const uint4& xv, const uint4& rv) {
            const uint32_t x4[4] = {xv.x, xv.y, xv.z, xv.w};
            if (PRO == kProAddNorm) {
                const uint32_t r4[4] = {rv.x, rv.y, rv.z, rv.w};
                uint32_t o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float s0 = bf16_lo(x4[j]) + bf16_lo(r4[j]);
                    float s1 = bf16_hi(x4[j]) + bf16_hi(r4[j]);
                    o[j] = pack_bf16x2(s0, s1);
                    if (a.var_from_rounded) {
                        s0 = bf16_lo(o[j]);
                        s1 = bf16_hi(o[j]);
                    }
                    ss[m] += s0 * s0;
                    ss[m] += s1 * s1;
                }
                const uint4 ov = make_uint4(o[0], o[1], o[2], o[3]);
                st16(xs + m * a.k + c * 8, ov);
                if (blockIdx.x == 0) st16(a.sum_out + (int64_t)m * a.ld_sum_out + c * 8, ov);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v0 = bf16_lo(x4[j]), v1 = bf16_hi(x4[j]);
                    ss[m] += v0 * v0;
                    ss[m] += v1 * v1;
                }
                st16(xs + m * a.k + c * 8, xv);
            }
        };
        if (threadIdx.x < kPT) {
#pragma unroll
            for (int m = 0; m < M; ++m) {
#pragma unroll
                for (int j = 0; j < PC; ++j) {
                    const int c = threadIdx.x + j * kPT;
                    if (c < kvec) absorb(m, c, pxe[m][j], pre[m][j]);
                }
                for (int c = threadIdx.x + PC * kPT; c < kvec; c += kPT) {
                    const uint4 xv = ld16(a.x + (int64_t)m * a.ldx + c * 8);
                    uint4 rv = xv;
                    if (PRO == kProAddNorm) rv = ld16(a.add_r + (int64_t)m * a.ld_add_r + c * 8);
                    absorb(m, c, xv, rv);
                }
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
            rrms[m] = 1.0f / sqrtf(t / (float)a.k + a.eps);
        }
        auto scale = [&](int m, int c, const uint4& wv) {
            uint4* px = reinterpret_cast<uint4*>(xs + m * a.k + c * 8);
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
                const uint4 wv = ld16(a.ln_w + c * 8);
#pragma unroll
                for (int m = 0; m < M; ++m) scale(m, c, wv);
            }
        }
    }
    __syncthreads();
    MV_WSTAMP(1);

    // Stage B goes out only now.  Requesting both stages before the input transform measured
    // slower on MI355X for every decode projection (O: 8.0 vs 7.3 us).
#pragma unroll
    for (int i = U; i < 2 * U; ++i) {
        const int t = min(i, cur.last);
        bufB[i - U] = ld16_nt(cur.wbase + (int64_t)t * wstride);
        __builtin_amdgcn_sched_barrier(0);
    }
    // Chunks 0 and 1 of the current unit are in flight in bufA / bufB here and at the top of
    // every pass.  The stream is strictly A, B, A, B, ... with no branch around a load (a chunk
    // index past the end re-reads the wave's last tile, a cache hit; its product is masked to
    // zero), so every wait in the loop is for exactly one stage and the other stays in flight.
    constexpr int kPast = 1 << 24;
    for (;;) {
        const int un = u + gridDim.x;
        const bool more = un < a.nunits;
        UnitCtx nxt = cur;
        if (more) nxt = setup(un);

        float acc[M];
#pragma unroll
        for (int m = 0; m < M; ++m) acc[m] = 0.f;

        const int npair = (cur.nchunk + 1) >> 1;
#ifdef MV_WGEMM_STAMP
        if (u == (int)blockIdx.x && npair > 1) {
            compute(bufA, cur, 0, acc);   // timing only: first stage landed (results are wrong)
            MV_WSTAMP(2);
        }
#endif
        for (int p = 0; p < npair - 1; ++p) {
            compute(bufA, cur, 2 * p, acc);
            load(bufA, cur, 2 * p + 2);
            compute(bufB, cur, 2 * p + 1, acc);
            load(bufB, cur, 2 * p + 3);
        }
        const int cl = npair > 0 ? 2 * (npair - 1) : kPast;  // nothing to add if the wave has no tiles
        if (more) {  // two copies so that each path keeps exact request counts
            compute(bufA, cur, cl, acc);
            load(bufA, nxt, 0);
            compute(bufB, cur, cl + 1, acc);
            load(bufB, nxt, 1);
        } else {
            compute(bufA, cur, cl, acc);
            compute(bufB, cur, cl + 1, acc);
        }
        if (!more) MV_WSTAMP(3);

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
        if (threadIdx.x < 16 * M) {
            const int m = threadIdx.x >> 4, rr = threadIdx.x & 15;
            float s = 0.f;
#pragma unroll
            for (int w8 = 0; w8 < GW; ++w8) s += red[(w8 * M + m) * 16 + rr];
            const int n = cur.nt * 16 + rr;
            if (a.partial) {
                a.partial[((int64_t)cur.ks * a.m + m) * a.n + n] = s;
            } else if (a.flags & MV_GEMM_OUT_F32) {
                reinterpret_cast<float*>(a.y)[(int64_t)m * a.ldy + n] = s;
            } else if (a.flags & MV_GEMM_SWIGLU) {
                if (rr < 8) {
                    float g = 0.f;
#pragma unroll
                    for (int w8 = 0; w8 < GW; ++w8) g += red[(w8 * M + m) * 16 + rr + 8];
                    reinterpret_cast<uint16_t*>(a.y)[(int64_t)m * a.ldy + cur.nt * 8 + rr] =
                        (uint16_t)f32_to_bf16_bits(silu_mul_rounded(g, s));
                }
            } else {
                reinterpret_cast<uint16_t*>(a.y)[(int64_t)m * a.ldy + n] =
                    (uint16_t)f32_to_bf16_bits(epilogue_value(a, s, m, n));
            }
        }
        if (!more) break;
        __syncthreads();
        u = un;
        cur = nxt;
    }
    MV_WSTAMP(4);
}

// ------------------------------------------------------------ M > 4: MFMA form
// MB = number of 16-row blocks of x (M <= 16*MB); SG = strips a workgroup multiplies together.
// A wave loads the x fragments of a k-tile once and uses them against the SG weight tiles of
// that k-tile: x comes out of L2 for every workgroup, so without the sharing the x requests
// (MB per weight tile) outnumber the weight requests and the L2 path, not HBM, sets the pace.
template <int MB, int SG>
__global__ __launch_bounds__(kGT) void wgemm_mfma_kernel(GemmArgs a) {
    constexpr int U = (SG + MB <= 6) ? 2 : 1;  // k-tiles per stage (register budget)
    // cross-wave reduction buffer [kGW][SG][MB][64 lanes][4]: 8 KiB per (strip, row block) pair,
    // up to 128 KiB (dynamic: past the 64 KiB a static array may have)
    extern __shared__ __attribute__((aligned(16))) float red[];

    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, q = lane >> 4;

    // this lane's x rows (B operand column), clamped into [0, M)
    const uint16_t* xrow[MB];
#pragma unroll
    for (int b = 0; b < MB; ++b) xrow[b] = a.x + (int64_t)min(b * 16 + r, a.m - 1) * a.ldx + q * 8;

    for (int u = blockIdx.x; u < a.nunits; u += gridDim.x) {
        const int g = u / a.ksplit, ks = u - g * a.ksplit;
        const int kt0 = ks * a.ktps;
        const int kt1 = min(a.kt, kt0 + a.ktps);
        const int ntile = (kt1 - kt0 - wid + kGW - 1) / kGW;
        const int last = max(ntile - 1, 0);
        const int ktw = kt0 + (ntile > 0 ? wid : 0);  // a wave without tiles still loads (clamped)
        const uint4* wbase[SG];
#pragma unroll
        for (int sI = 0; sI < SG; ++sI)
            wbase[sI] = a.w + ((int64_t)min(g * SG + sI, a.nt - 1) * a.kt + ktw) * 64 + lane;

        f32x4_t acc[SG][MB];
#pragma unroll
        for (int sI = 0; sI < SG; ++sI)
#pragma unroll
            for (int b = 0; b < MB; ++b) acc[sI][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

        uint4 wA[U][SG], wB[U][SG], xA[U][MB], xB[U][MB];
        auto load = [&](uint4(&wb)[U][SG], uint4(&xb)[U][MB], int c) {
#pragma unroll
            for (int i = 0; i < U; ++i) {
                const int t = min(c * U + i, last);
#pragma unroll
                for (int sI = 0; sI < SG; ++sI) wb[i][sI] = ld16_nt(wbase[sI] + (int64_t)t * (kGW * 64));
                const int kcol = (ktw + t * kGW) * 32;
#pragma unroll
                // (with these requests removed the four Llama-3-8B projections at M = 32 take 90 us
                // instead of 104: the x path costs about 13 %)
                for (int b = 0; b < MB; ++b) xb[i][b] = ld16(xrow[b] + kcol);
            }
        };
        auto compute = [&](const uint4(&wb)[U][SG], const uint4(&xb)[U][MB], int c, bool all) {
#pragma unroll
            for (int i = 0; i < U; ++i) {
                if (all || c * U + i < ntile) {  // wave-uniform
#pragma unroll
                    for (int sI = 0; sI < SG; ++sI) {
                        const bf16x8_t wf = __builtin_bit_cast(bf16x8_t, wb[i][sI]);
#pragma unroll
                        for (int b = 0; b < MB; ++b)
                            acc[sI][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                wf, __builtin_bit_cast(bf16x8_t, xb[i][b]), acc[sI][b], 0, 0, 0);
                    }
                }
            }
        };

        const int nchunk = (ntile + U - 1) / U;
        // Both first stages go out unconditionally (indices past the end are clamped) and the steady
        // state has no branch around a request, so the compiler's wait counts are exact: with a
        // request under an `if` anywhere before or inside the loop every wait covers both stages
        // and only one is ever in flight.  The last three chunks at most are handled after it.
        // (The fences keep the two first stages in the order the loop issues them: interleaved by
        // the scheduler they would make the loop's first wait, and through the back edge all of
        // them, cover both stages.)
        __builtin_amdgcn_sched_barrier(0);
        load(wA, xA, 0);
        __builtin_amdgcn_sched_barrier(0);
        load(wB, xB, 1);
        __builtin_amdgcn_sched_barrier(0);
        int c = 0;
        for (; c + 3 < nchunk; c += 2) {
            compute(wA, xA, c, true);
            __builtin_amdgcn_sched_barrier(0);
            load(wA, xA, c + 2);
            __builtin_amdgcn_sched_barrier(0);
            compute(wB, xB, c + 1, true);
            __builtin_amdgcn_sched_barrier(0);
            load(wB, xB, c + 3);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (c < nchunk) {
            compute(wA, xA, c, false);
            if (c + 2 < nchunk) load(wA, xA, c + 2);
            if (c + 1 < nchunk) compute(wB, xB, c + 1, false);
            if (c + 2 < nchunk) compute(wA, xA, c + 2, false);
        }

        // acc[s][b][j] = C[n = 16 (g SG + s) + 4q + j][m = 16b + r]
#pragma unroll
        for (int sI = 0; sI < SG; ++sI)
#pragma unroll
            for (int b = 0; b < MB; ++b)
                *reinterpret_cast<f32x4_t*>(&red[(((wid * SG + sI) * MB + b) * 64 + lane) * 4]) = acc[sI][b];
        __syncthreads();
        // one wave per (strip, row block) pair; more pairs than waves take a second turn
        for (int sb = wid; sb < MB * SG; sb += kGW) {
            const int l = lane;
            const int sI = sb / MB, b = sb - sI * MB;
            const int rr = l & 15, qq = l >> 4;
            const int m = b * 16 + rr;
            const int nt = g * SG + sI;
            auto sum_at = [&](int ll) {
                f32x4_t s = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int w8 = 0; w8 < kGW; ++w8)
                    s += *reinterpret_cast<const f32x4_t*>(&red[(((w8 * SG + sI) * MB + b) * 64 + ll) * 4]);
                return s;
            };
            if (m < a.m && nt < a.nt) {
                const f32x4_t s = sum_at(l);
                const int n0 = nt * 16 + qq * 4;
                if (a.partial) {
                    *reinterpret_cast<f32x4_t*>(&a.partial[((int64_t)ks * a.m + m) * a.n + n0]) = s;
                } else if (a.flags & MV_GEMM_OUT_F32) {
                    *reinterpret_cast<f32x4_t*>(
                        &reinterpret_cast<float*>(a.y)[(int64_t)m * a.ldy + n0]) = s;
                } else if (a.flags & MV_GEMM_SWIGLU) {
                    if (qq < 2) {  // rows 0..7 = up; their gates sit 8 rows (32 lanes) further
                        const f32x4_t gt = sum_at(l + 32);
                        uint2 o;
                        o.x = pack_bf16x2(silu_mul_rounded(gt[0], s[0]), silu_mul_rounded(gt[1], s[1]));
                        o.y = pack_bf16x2(silu_mul_rounded(gt[2], s[2]), silu_mul_rounded(gt[3], s[3]));
                        *reinterpret_cast<uint2*>(&reinterpret_cast<uint16_t*>(
                            a.y)[(int64_t)m * a.ldy + nt * 8 + qq * 4]) = o;
                    }
                } else {
                    uint2 o;
                    o.x = pack_bf16x2(epilogue_value(a, s[0], m, n0), epilogue_value(a, s[1], m, n0 + 1));
                    o.y = pack_bf16x2(epilogue_value(a, s[2], m, n0 + 2), epilogue_value(a, s[3], m, n0 + 3));
                    *reinterpret_cast<uint2*>(
                        &reinterpret_cast<uint16_t*>(a.y)[(int64_t)m * a.ldy + n0]) = o;
                }
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------ slab reduction
// Eight consecutive outputs of a row per thread, for bf16 outputs with bias / residual epilogues on
// 16-byte-aligned rows: the slabs of many-token products are tens of megabytes (the down projection
// at 2048 tokens in two k-slices: 67 MB of slabs, 17 of residual in, 17 out), and one element per
// thread read them at 2.3 TB/s (43 us).  Same sums in the same order, same rounding points.
__global__ __launch_bounds__(256) void wgemm_finalize_vec8_kernel(GemmArgs a) {
    const int nvec = a.n >> 3;
    const int64_t total = (int64_t)a.m * nvec;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int m = (int)(i / nvec), n0 = (int)(i - (int64_t)m * nvec) * 8;
        f32x4_t lo[kMaxKsplit], hi[kMaxKsplit];
#pragma unroll
        for (int ks = 0; ks < kMaxKsplit; ++ks) {
            if (ks >= a.ksplit) break;   // uniform
            const float* p = a.partial + ((int64_t)ks * a.m + m) * a.n + n0;
            lo[ks] = *reinterpret_cast<const f32x4_t*>(p);
            hi[ks] = *reinterpret_cast<const f32x4_t*>(p + 4);
        }
        const uint4 bv = a.bias ? ld16(a.bias + n0) : make_uint4(0, 0, 0, 0);
        const uint4 rv = a.residual ? ld16(a.residual + (int64_t)m * a.ld_res + n0) : make_uint4(0, 0, 0, 0);
        f32x4_t slo = (f32x4_t){0.f, 0.f, 0.f, 0.f}, shi = slo;
#pragma unroll
        for (int ks = 0; ks < kMaxKsplit; ++ks)
            if (ks < a.ksplit) {
                slo += lo[ks];
                shi += hi[ks];
            }
        const float acc[8] = {slo[0], slo[1], slo[2], slo[3], shi[0], shi[1], shi[2], shi[3]};
        const uint32_t b4[4] = {bv.x, bv.y, bv.z, bv.w}, r4[4] = {rv.x, rv.y, rv.z, rv.w};
        uint32_t o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float t0 = round_bf16(acc[2 * e]), t1 = round_bf16(acc[2 * e + 1]);
            if (a.bias) {
                t0 = round_bf16(t0 + bf16_lo(b4[e]));
                t1 = round_bf16(t1 + bf16_hi(b4[e]));
            }
            if (a.residual) {
                t0 = round_bf16(t0 + bf16_lo(r4[e]));
                t1 = round_bf16(t1 + bf16_hi(r4[e]));
            }
            o[e] = pack_bf16x2(t0, t1);
        }
        st16(reinterpret_cast<uint16_t*>(a.y) + (int64_t)m * a.ldy + n0, make_uint4(o[0], o[1], o[2], o[3]));
    }
}

// One thread per output element (per up/gate pair for SWIGLU); slabs added in order.
__global__ __launch_bounds__(256) void wgemm_finalize_kernel(GemmArgs a) {
    const bool sw = (a.flags & MV_GEMM_SWIGLU) != 0;
    const int ncols = sw ? a.n / 2 : a.n;
    const int64_t total = (int64_t)a.m * ncols;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * 256) {
        const int m = (int)(i / ncols), c = (int)(i - (int64_t)m * ncols);
        // all slabs of the element requested before the first is added (they are added in order)
        auto slab_sum = [&](int n) {
            float v[kMaxKsplit];
#pragma unroll
            for (int ks = 0; ks < kMaxKsplit; ++ks) {
                if (ks >= a.ksplit) break;   // uniform
                v[ks] = a.partial[((int64_t)ks * a.m + m) * a.n + n];
            }
            float s = 0.f;
#pragma unroll
            for (int ks = 0; ks < kMaxKsplit; ++ks)
                if (ks < a.ksplit) s += v[ks];
            for (int ks = kMaxKsplit; ks < a.ksplit; ++ks) s += a.partial[((int64_t)ks * a.m + m) * a.n + n];
            return s;
        };
        if (sw) {
            const int nu = (c >> 3) * 16 + (c & 7);
            const float up = slab_sum(nu), gate = slab_sum(nu + 8);
            reinterpret_cast<uint16_t*>(a.y)[(int64_t)m * a.ldy + c] =
                (uint16_t)f32_to_bf16_bits(silu_mul_rounded(gate, up));
        } else if (a.flags & MV_GEMM_OUT_F32) {
            reinterpret_cast<float*>(a.y)[(int64_t)m * a.ldy + c] = slab_sum(c);
        } else {
            reinterpret_cast<uint16_t*>(a.y)[(int64_t)m * a.ldy + c] =
                (uint16_t)f32_to_bf16_bits(epilogue_value(a, slab_sum(c), m, c));
        }
    }
}

// Slab reduction + output epilogue + RMSNorm of the result, one workgroup per row:
//   t = bf16(sum of slabs); t = bf16(t + bias); s = t + residual (fp32); y = bf16(s);
//   normed = bf16((float(y) * rrms) * ln_w),  rrms = 1 / sqrt(mean(s^2) + eps)
// with the mean over the unrounded s (what add+RMSNorm after a projection computes) or, with
// post_var_from_rounded, over y as stored (what RMSNorm of a stored hidden state computes).
// Chunk assignment and summation order are those of the stand-alone norm kernels (256 threads,
// stride 256, block_sum), so the fused and the separate path agree bit for bit.
constexpr int kPostChunks = 4;  // 16-byte chunks of y a thread keeps: rows up to 8192 wide
__global__ __launch_bounds__(256) void wgemm_finalize_norm_kernel(GemmArgs a) {
    __shared__ float red[256 / 64];
    const int m = blockIdx.x;
    const int nvec = a.n >> 3;
    uint4 keep[kPostChunks];
    float ss = 0.f;
    // Every operand of every chunk is requested before the first is used - the first kPostAhead slabs
    // included (the usual splits are 2 and 4): the slabs were written by another kernel, so each
    // dependent round of loads is a trip to memory, and this kernel is nothing but such trips.
    // (Chunk rows and slabs that do not exist are skipped: both conditions are uniform.)
    constexpr int kPostAhead = 4;
    f32x4_t p_lo[kPostChunks][kPostAhead], p_hi[kPostChunks][kPostAhead];
    uint4 b_in[kPostChunks], r_in[kPostChunks], w_in[kPostChunks];
#pragma unroll
    for (int j = 0; j < kPostChunks; ++j) {
        if (j * 256 >= nvec) break;
        const int n0 = min(threadIdx.x + j * 256, nvec - 1) * 8;
#pragma unroll
        for (int ks = 0; ks < kPostAhead; ++ks) {
            if (ks >= a.ksplit) break;
            const float* p = a.partial + ((int64_t)ks * a.m + m) * a.n + n0;
            p_lo[j][ks] = *reinterpret_cast<const f32x4_t*>(p);
            p_hi[j][ks] = *reinterpret_cast<const f32x4_t*>(p + 4);
        }
        b_in[j] = a.bias ? ld16(a.bias + n0) : make_uint4(0, 0, 0, 0);
        r_in[j] = a.residual ? ld16(a.residual + (int64_t)m * a.ld_res + n0) : make_uint4(0, 0, 0, 0);
        w_in[j] = ld16(a.post_ln_w + n0);
    }
#pragma unroll
    for (int j = 0; j < kPostChunks; ++j) {
        const int c = threadIdx.x + j * 256;
        if (c < nvec) {
            const int n0 = c * 8;
            f32x4_t lo = (f32x4_t){0.f, 0.f, 0.f, 0.f}, hi = lo;
#pragma unroll
            for (int ks = 0; ks < kPostAhead; ++ks) {
                if (ks < a.ksplit) {   // slabs are added in order, whichever way they were fetched
                    lo += p_lo[j][ks];
                    hi += p_hi[j][ks];
                }
            }
            for (int ks = kPostAhead; ks < a.ksplit; ++ks) {
                const float* p = a.partial + ((int64_t)ks * a.m + m) * a.n + n0;
                lo += *reinterpret_cast<const f32x4_t*>(p);
                hi += *reinterpret_cast<const f32x4_t*>(p + 4);
            }
            const float acc[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            const uint4 bv = b_in[j], rv = r_in[j];
            const uint32_t b4[4] = {bv.x, bv.y, bv.z, bv.w}, r4[4] = {rv.x, rv.y, rv.z, rv.w};
            uint32_t o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t0 = round_bf16(acc[2 * e]), t1 = round_bf16(acc[2 * e + 1]);
                if (a.bias) {
                    t0 = round_bf16(t0 + bf16_lo(b4[e]));
                    t1 = round_bf16(t1 + bf16_hi(b4[e]));
                }
                float s0 = t0, s1 = t1;
                if (a.residual) {
                    s0 = t0 + bf16_lo(r4[e]);
                    s1 = t1 + bf16_hi(r4[e]);
                }
                o[e] = pack_bf16x2(s0, s1);
                if (a.post_var_from_rounded) {
                    s0 = bf16_lo(o[e]);
                    s1 = bf16_hi(o[e]);
                }
                ss += s0 * s0;
                ss += s1 * s1;
            }
            keep[j] = make_uint4(o[0], o[1], o[2], o[3]);
            st16(reinterpret_cast<uint16_t*>(a.y) + (int64_t)m * a.ldy + n0, keep[j]);
        }
    }
    const float total = block_sum(ss, red);
    const float rrms = 1.0f / sqrtf(total / (float)a.n + a.post_eps);
#pragma unroll
    for (int j = 0; j < kPostChunks; ++j) {
        const int c = threadIdx.x + j * 256;
        if (c < nvec) {
            const uint4 wv = w_in[j];
            const uint32_t s4[4] = {keep[j].x, keep[j].y, keep[j].z, keep[j].w};
            const uint32_t w4[4] = {wv.x, wv.y, wv.z, wv.w};
            uint32_t o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                o[e] = pack_bf16x2((bf16_lo(s4[e]) * rrms) * bf16_lo(w4[e]),
                                   (bf16_hi(s4[e]) * rrms) * bf16_hi(w4[e]));
            st16(a.post_out + (int64_t)m * a.ld_post + c * 8, make_uint4(o[0], o[1], o[2], o[3]));
        }
    }
}

// The slab reduction for other kernels of the library that split k (the prefill GEMM): same
// layout partial[ks][m][n], same epilogues.
int launch_wgemm_finalize(float* partial, int ksplit, int m, int n, int flags, const uint16_t* bias,
                          const uint16_t* residual, int64_t ld_res, void* y, int64_t ldy, hipStream_t s) {
    GemmArgs a = {};
    a.partial = partial;
    a.ksplit = ksplit;
    a.m = m;
    a.n = n;
    a.flags = flags;
    a.bias = bias;
    a.residual = residual;
    a.ld_res = ld_res;
    a.y = y;
    a.ldy = ldy;
    const int ncols = (flags & MV_GEMM_SWIGLU) ? n / 2 : n;
    const int most = num_cus() * 8;
    const bool vec8 = !(flags & (MV_GEMM_SWIGLU | MV_GEMM_OUT_F32)) && ksplit <= kMaxKsplit && (n % 8) == 0 &&
                      (ldy % 8) == 0 && aligned16(y) && aligned16(partial) && (!bias || aligned16(bias)) &&
                      (!residual || (aligned16(residual) && (ld_res % 8) == 0)) && (int64_t)m * n >= (1 << 16);
    if (vec8) {
        int64_t blocks = ((int64_t)m * (n / 8) + 255) / 256;
        if (blocks > most) blocks = most;
        hipLaunchKernelGGL(wgemm_finalize_vec8_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
        return check_launch();
    }
    const int64_t total = (int64_t)m * ncols;
    int64_t blocks = (total + 255) / 256;
    if (blocks > most) blocks = most;
    hipLaunchKernelGGL(wgemm_finalize_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
    return check_launch();
}

// ------------------------------------------------------------ host planning
// k-slices per strip: balance whole units over the CUs.  Cost in tiles per CU,
// with a split paying roughly one extra kernel boundary.
static int mfma_strip_group(int m, int nt, int kt, bool free_reduce);

// k-slices for `units` whole-k work items: split while fewer than half the CUs would have one and
// every wave keeps at least 4 tiles
static int ksplit_for_units(int units, int kt) {
    const int cus = num_cus();
    int best = 1;
    for (int ks = 2; ks <= kMaxKsplit; ks *= 2) {
        if (kt % ks) break;
        if (kt / ks < 4 * kGW) break;             // keep >= 4 tiles per wave
        if ((long)units * best * 2 > cus) break;  // already at least half the CUs busy
        best = ks;
    }
    return best;
}

// `nt` = work items along N: strips for the dot form, strip groups for the MFMA form.
// `free_reduce`: the slabs are added by a launch that runs anyway (the caller's next kernel for
// MV_GEMM_SLABS, the norm kernel of mv_wgemm_postnorm), so a k-split costs no launch of its own.
static int pick_ksplit(int m, int nt, int kt, bool free_reduce = false) {
    if (m > 4) {
        const int sg = mfma_strip_group(m, nt, kt, free_reduce);
        nt = (nt + sg - 1) / sg;
    }
    if (const int forced = g_tune_ksplit; forced > 0) {
        int ks = forced;
        while (ks > 1 && (kt % ks) != 0) --ks;
        return ks;
    }
    // Measured on MI355X (tools/bench_wgemm.py, M=1): whole strips beat every k-split for all
    // four decode projections (QKV 11.8 vs 14.8 us, O 8.0 vs 11.4, gate/up 38.8 vs 48.5, down
    // 20.8 vs 23.7): the second launch that adds the slabs costs more than the imbalance of
    // 1.5 strips per CU.  Splitting only pays when there are far fewer strips than CUs.
    // ... and for the dot form not even then: shards of tensor-parallel runs (tools/sweep_small_gemv.py,
    // M=1) run 5.4 us unsplit against 6.5-8.7 us split at 48 strips (N=768, K=4096), 5.8 against
    // 7.2-10.0 at 96 strips.
    if (m <= 4 && nt >= 32) return 1;
    return ksplit_for_units(nt, kt);
}

// Waves per workgroup of the dot form for a projection of `nunits` units of `kt` k-tiles each (the
// rule above wgemm_impl's launch explains the numbers).  qkv_attn.hip runs the QKV projection in
// the same shape, so that its sums are formed in the same order.
static int dot_form_waves_for_units(int nunits, int kt) {
    const int cus = num_cus();
    const bool one_long_strip_per_cu = nunits <= cus && nunits * 8 >= cus * 7 && kt >= 128;
    const int forced = g_tune_waves;
    return forced ? forced : (one_long_strip_per_cu ? 16 : 8);
}
int dot_form_ksplit(int m, int nt, int kt) { return g_tune_contig ? 0 : pick_ksplit(m, nt, kt); }
int dot_form_waves(int m, int nt, int kt) { return dot_form_waves_for_units(nt * pick_ksplit(m, nt, kt), kt); }

template <int M, int GW, int PRO>
static int launch_dot(const GemmArgs& a, int grid, hipStream_t s) {
    constexpr int U = (M <= 2) ? 4 : 2;  // tiles per stage (register budget: 128 VGPRs at 16 waves per CU)
    const size_t lds = (size_t)M * a.k * 2 + 16 + (size_t)GW * M * 16 * 4;
    if (lds > 160 * 1024) return MV_ERR_UNSUPPORTED;
    static LdsOptIn opted;  // per instantiation
    if (int rc = ensure_dynamic_lds((const void*)wgemm_dot_kernel<M, GW, U, PRO>,
                                    lds > 64 * 1024 ? 160 * 1024 : lds, opted))
        return rc;
#ifdef MV_WGEMM_STAMP
    GemmArgs as = a;
    as.stamp_slot = g_stamp_counter++;
    hipLaunchKernelGGL((wgemm_dot_kernel<M, GW, U, PRO>), dim3(grid), dim3(GW * kWave), lds, s, as);
    return check_launch();
#endif
    hipLaunchKernelGGL((wgemm_dot_kernel<M, GW, U, PRO>), dim3(grid), dim3(GW * kWave), lds, s, a);
    return check_launch();
}

template <int M, int GW>
static int launch_dot_pro(const GemmArgs& a, int grid, hipStream_t s) {
    switch (a.prologue) {
        case kProNone: return launch_dot<M, GW, kProNone>(a, grid, s);
        case kProNorm: return launch_dot<M, GW, kProNorm>(a, grid, s);
        default: return launch_dot<M, GW, kProAddNorm>(a, grid, s);
    }
}

template <int M>
static int launch_dot_waves(const GemmArgs& a, int waves, int grid, hipStream_t s) {
    return waves == 16 ? launch_dot_pro<M, 16>(a, grid, s) : launch_dot_pro<M, 8>(a, grid, s);
}

template <int MB, int SG>
static int launch_mfma(const GemmArgs& a, int grid, hipStream_t s) {
    constexpr size_t lds = (size_t)kGW * SG * MB * 64 * 4 * sizeof(float);
    static_assert(lds <= 128 * 1024, "reduction buffer fits LDS");
    static LdsOptIn opted;  // per instantiation
    if (int rc = ensure_dynamic_lds((const void*)wgemm_mfma_kernel<MB, SG>, lds, opted)) return rc;
    hipLaunchKernelGGL((wgemm_mfma_kernel<MB, SG>), dim3(grid), dim3(kGT), lds, s, a);
    return check_launch();
}

// strips per workgroup of the MFMA form: the cross-wave reduction buffer is 8 KiB per
// (strip, row block) pair, 128 KiB for four strips of four row blocks
static int mfma_strip_group(int m, int nt, int kt, bool free_reduce) {
    const int mb = (m + 15) / 16;
    const int most = 4;
    if (const int forced = g_tune_sg;
        forced == 1 || forced == 2 || forced == 4 || ((forced == 7 || forced == 3) && mb <= 2))
        return forced;
    // Seven strips per workgroup where that deals the strips out to the CUs exactly (Llama-3-8B
    // gate/up: 1792 strips = 7 x 256): in groups of four 448 units fall on 256 CUs as two rounds, the
    // second three quarters full, and every workgroup re-reads x for four strips instead of seven.
    if (mb <= 2 && nt % 7 == 0 && (nt / 7) % num_cus() == 0) return 7;
    // Three or four row blocks of x: always the widest group - the x traffic out of L2 dominates
    // otherwise (tools/sweep_mfma_gemv.py).
    if (mb > 2) return most;
    // One or two row blocks (5-32 rows).  A small model of the three costs, fitted to the Llama-3-8B
    // projections at 8, 16 and 32 rows (tools/bench_mfma_gemv.py, tools/bench_slabs.py): weights
    // stream at 6.45 MB/us over the CUs that have a unit, every unit re-reads its k-slice of x out of
    // L2 at about 12.9 MB/us, and a k-split pays a slab-reduction launch of 3.6 us (QKV at 32 rows:
    // groups of three in two k-slices, one unit per CU, 11.8 + 3.0 us against 16.2 for pairs without
    // a split, which leave a quarter of the CUs idle).  (Down projection, 14336 wide: single
    // strips win at 8 rows, 21.6 against 22.8 us, and lose at 16, 27.3 against 22.9, because x is
    // then as many bytes as the weights; the 4096-wide O projection keeps single strips at both,
    // 9.1 against 11.2 us, its split costing more than its x.)
    // With a free slab reduction the same model decides with that term dropped.
    const int cus = num_cus();
    int best = 1;
    double best_us = 1e30;
    for (int sg = most; sg >= 1; --sg) {
        const int groups = (nt + sg - 1) / sg;
        const int ks = ksplit_for_units(groups, kt);
        const long units = (long)groups * ks;
        const double fill = (double)units / (double)(((units + cus - 1) / cus) * cus);
        const double w_mb = (double)nt * kt * 1024e-6, x_mb = (double)units * m * (kt / ks) * 64e-6;
        const double us = w_mb / (6.45 * fill) + x_mb / 12.9 + (ks > 1 && !free_reduce ? 3.6 : 0.0);
        if (us < best_us) {
            best_us = us;
            best = sg;
        }
    }
    return best;
}

static int wgemm_impl(const void* x, int64_t ldx, const void* add_r, int64_t ld_add_r,
                      const void* ln_w, float eps, void* sum_out, int64_t ld_sum_out, int prologue,
                      int var_from_rounded, const void* w_tiles, const void* bias,
                      const void* residual, int64_t ld_res, void* y, int64_t ldy, int m, int n,
                      int k, void* workspace, size_t workspace_bytes, int flags,
                      mv_stream_t stream, const void* post_ln_w = nullptr, float post_eps = 0.f,
                      void* post_out = nullptr, int64_t ld_post = 0, int post_var_from_rounded = 0) {
    const bool slabs_only = (flags & MV_GEMM_SLABS) != 0;
    if (!x || !w_tiles || (!y && !slabs_only) || m <= 0 || n <= 0 || k <= 0) return MV_ERR_INVALID_ARG;
    if (m > 64 || (n % 16) != 0 || (k % 32) != 0) return MV_ERR_UNSUPPORTED;
    if (!aligned16(x) || !aligned16(w_tiles) || (ldx % 8) != 0 || ldx < k) return MV_ERR_INVALID_ARG;
    const bool sw = (flags & MV_GEMM_SWIGLU) != 0, of32 = (flags & MV_GEMM_OUT_F32) != 0;
    if (sw && (bias || residual || of32)) return MV_ERR_UNSUPPORTED;
    if (of32 && (bias || residual)) return MV_ERR_UNSUPPORTED;
    if (slabs_only && (sw || of32 || bias || residual || post_ln_w)) return MV_ERR_UNSUPPORTED;
    const int ncols = sw ? n / 2 : n;
    if (!slabs_only) {
        if (ldy < ncols) return MV_ERR_INVALID_ARG;
        if (!of32 && ((reinterpret_cast<uintptr_t>(y) & 7) != 0 || (ldy % 4) != 0))
            return MV_ERR_INVALID_ARG;
        if (of32 && (!aligned16(y) || (ldy % 4) != 0)) return MV_ERR_INVALID_ARG;
    }
    if (residual && ld_res < n) return MV_ERR_INVALID_ARG;
    if (prologue != kProNone) {
        if (!aligned16(ln_w)) return MV_ERR_INVALID_ARG;
        if (prologue == kProAddNorm &&
            (!aligned16(add_r) || !aligned16(sum_out) || (ld_add_r % 8) || (ld_sum_out % 8) ||
             ld_add_r < k || ld_sum_out < k))
            return MV_ERR_INVALID_ARG;
    }

    GemmArgs a = {};
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
    a.prologue = prologue;
    a.var_from_rounded = var_from_rounded;
    a.add_r = (const uint16_t*)add_r;
    a.ld_add_r = ld_add_r;
    a.ln_w = (const uint16_t*)ln_w;
    a.eps = eps;
    a.sum_out = (uint16_t*)sum_out;
    a.ld_sum_out = ld_sum_out;
    const bool free_reduce = slabs_only || post_ln_w != nullptr || (flags & MV_GEMM_PLAN_SLABS) != 0;
    a.ksplit = pick_ksplit(m, a.nt, a.kt, free_reduce);
    a.partial = nullptr;
    a.post_ln_w = (const uint16_t*)post_ln_w;
    a.post_out = (uint16_t*)post_out;
    a.ld_post = ld_post;
    a.post_eps = post_eps;
    a.post_var_from_rounded = post_var_from_rounded;
    if (slabs_only) {
        // the caller's next kernel adds the slabs (one slab when the planner would not split)
        const size_t need = (size_t)a.ksplit * m * n * sizeof(float);
        if (!workspace || workspace_bytes < need || !aligned16(workspace)) return MV_ERR_WORKSPACE;
        a.partial = (float*)workspace;
    } else if (post_ln_w) {
        // the norm lives in the slab reduction, so the product always goes through slabs (one
        // slab when the planner would not split)
        if (sw || of32 || (n % 8) != 0 || n > 8 * 256 * kPostChunks) return MV_ERR_UNSUPPORTED;
        if (!post_out || !aligned16(post_ln_w) || !aligned16(post_out) || !aligned16(y) ||
            (ld_post % 8) != 0 || ld_post < n || (ldy % 8) != 0)
            return MV_ERR_INVALID_ARG;
        if ((bias && !aligned16(bias)) || (residual && (!aligned16(residual) || (ld_res % 8) != 0)))
            return MV_ERR_INVALID_ARG;
        const size_t need = (size_t)a.ksplit * m * n * sizeof(float);
        if (!workspace || workspace_bytes < need || !aligned16(workspace)) return MV_ERR_WORKSPACE;
        a.partial = (float*)workspace;
    } else if (a.ksplit > 1) {
        const size_t need = (size_t)a.ksplit * m * n * sizeof(float);
        if (!workspace || workspace_bytes < need || !aligned16(workspace)) {
            a.ksplit = 1;  // no room for slabs: fall back to whole strips (slower, same result class)
        } else {
            a.partial = (float*)workspace;
        }
    }
    a.ktps = a.kt / a.ksplit;
    a.contig = g_tune_contig;
    const int sg = m > 4 ? mfma_strip_group(m, a.nt, a.kt, free_reduce) : 1;
    a.nunits = ((a.nt + sg - 1) / sg) * a.ksplit;
    a.flags = flags;

    const int cus = num_cus();
    // Dot-form shape, from sweeps on MI355X (tools/sweep_wgemm.py, M=1, 4 tiles per stage):
    // with at most one strip per CU a single 16-wave workgroup hides HBM latency best
    // (O 8.0 -> 7.4 us, down 20.6 -> 20.4); with more strips than CUs two 8-wave workgroups
    // per CU even out the tail (QKV 12.8 -> 11.3 us, gate/up 39.1 -> 37.6).
    // Sixteen waves only pay when nearly every CU has a strip and the strip is long: shards with
    // short k (K=1792: 4.8 vs 5.8 us) or few strips (48: 5.4 vs 5.8) are faster with eight.
    const int forced_bpc = g_tune_bpc;
    int waves = dot_form_waves_for_units(a.nunits, a.kt);
    int bpc = forced_bpc ? forced_bpc : (m <= 4 && waves == 8 ? 2 : 1);
    if (m <= 4) {
        // LDS limits co-residency; never size the grid past what fits
        const size_t lds = (size_t)m * k * 2 + 16 + (size_t)waves * m * 64;
        const int fit = (int)((160 * 1024) / (lds + 512));
        if (fit < 1) return MV_ERR_UNSUPPORTED;
        if (bpc > fit) bpc = fit;
    }
    if (bpc > 4) bpc = 4;  // 4 x 512 threads = 32 waves per CU
    int grid = a.nunits < cus * bpc ? a.nunits : cus * bpc;
    hipStream_t s = (hipStream_t)stream;

    int rc;
    switch (m) {
        case 1: rc = launch_dot_waves<1>(a, waves, grid, s); break;
        case 2: rc = launch_dot_waves<2>(a, waves, grid, s); break;
        case 3: rc = launch_dot_waves<3>(a, waves, grid, s); break;
        case 4: rc = launch_dot_waves<4>(a, waves, grid, s); break;
        default: {
            const int mb = (m + 15) / 16;
            if (mb == 1) rc = sg == 7 ? launch_mfma<1, 7>(a, grid, s) : sg == 3 ? launch_mfma<1, 3>(a, grid, s) : sg == 4 ? launch_mfma<1, 4>(a, grid, s) : sg == 2 ? launch_mfma<1, 2>(a, grid, s) : launch_mfma<1, 1>(a, grid, s);
            else if (mb == 2) rc = sg == 7 ? launch_mfma<2, 7>(a, grid, s) : sg == 3 ? launch_mfma<2, 3>(a, grid, s) : sg == 4 ? launch_mfma<2, 4>(a, grid, s) : sg == 2 ? launch_mfma<2, 2>(a, grid, s) : launch_mfma<2, 1>(a, grid, s);
            else if (mb == 3) rc = sg == 4 ? launch_mfma<3, 4>(a, grid, s) : sg == 2 ? launch_mfma<3, 2>(a, grid, s) : launch_mfma<3, 1>(a, grid, s);
            else rc = sg == 4 ? launch_mfma<4, 4>(a, grid, s) : sg == 2 ? launch_mfma<4, 2>(a, grid, s) : launch_mfma<4, 1>(a, grid, s);
        }
    }
    if (rc != MV_OK || slabs_only) return rc;
    if (post_ln_w) {
        hipLaunchKernelGGL(wgemm_finalize_norm_kernel, dim3(m), dim3(256), 0, s, a);
        rc = check_launch();
    } else if (a.ksplit > 1) {
        const int64_t total = (int64_t)m * ncols;
        int blocks = (int)((total + 255) / 256);
        if (blocks > cus * 4) blocks = cus * 4;
        hipLaunchKernelGGL(wgemm_finalize_kernel, dim3(blocks), dim3(256), 0, s, a);
        rc = check_launch();
    }
    return rc;
}

}  // namespace mv

extern "C" {

// Tuning hook for experiments: key 0 = workgroups per CU, key 1 = forced ksplit,
// key 2 = waves per workgroup of the dot form (8 or 16), key 3 = strips per workgroup of the
// MFMA form (1, 2, 3, 4, 7); 0 = let the planner choose.
int mv_wgemm_tune(int key, int value) {
    if (key == 2 && (value == 0 || value == 8 || value == 16)) {
        mv::g_tune_waves = value;
        return MV_OK;
    }
    if (key == 3 && (value == 0 || value == 1 || value == 2 || value == 3 || value == 4 || value == 7)) {
        mv::g_tune_sg = value;
        return MV_OK;
    }
    if (key == 0 && value >= 0 && value <= 8) {
        mv::g_tune_bpc = value;
        return MV_OK;
    }
    if (key == 1 && value >= 0 && value <= mv::kMaxKsplit) {
        mv::g_tune_ksplit = value;
        return MV_OK;
    }
    if (key == 4 && (value == 0 || value == 1)) {
        mv::g_tune_contig = value;
        return MV_OK;
    }
    return MV_ERR_INVALID_ARG;
}

size_t mv_wgemm_workspace_bytes(int m, int n, int k) {
    if (m <= 0 || n <= 0 || k <= 0 || (n % 16) || (k % 32)) return 0;
    // slabs for the split the planner picks now; a call that later finds the
    // workspace too small (tuning changed) runs unsplit instead of failing
    // (the larger of the plain plan and the MV_GEMM_PLAN_SLABS one: the flag is not known here)
    const int ks = std::max(mv::pick_ksplit(m, n / 16, k / 32), mv::pick_ksplit(m, n / 16, k / 32, true));
    return ks > 1 ? (size_t)ks * (size_t)m * (size_t)n * sizeof(float) : 0;
}

int mv_wgemm(const void* x, int64_t ldx, const void* w_tiles, const void* bias,
             const void* residual, int64_t ld_res, void* y, int64_t ldy, int m, int n, int k,
             void* workspace, size_t workspace_bytes, int flags, mv_stream_t stream) {
    return mv::wgemm_impl(x, ldx, nullptr, 0, nullptr, 0.f, nullptr, 0, mv::kProNone, 0, w_tiles, bias,
                          residual, ld_res, y, ldy, m, n, k, workspace, workspace_bytes, flags,
                          stream);
}

int mv_wgemm_slab_count(int m, int n, int k) {
    if (m <= 0 || n <= 0 || k <= 0 || (n % 16) || (k % 32)) return 0;
    return mv::pick_ksplit(m, n / 16, k / 32, true);
}

size_t mv_wgemm_postnorm_workspace_bytes(int m, int n, int k) {
    if (m <= 0 || n <= 0 || k <= 0 || (n % 16) || (k % 32)) return 0;
    const int ks = mv::pick_ksplit(m, n / 16, k / 32, true);
    return (size_t)(ks > 1 ? ks : 1) * (size_t)m * (size_t)n * sizeof(float);
}

int mv_wgemm_postnorm(const void* x, int64_t ldx, const void* w_tiles, const void* bias,
                      const void* residual, int64_t ld_res, void* y, int64_t ldy, const void* ln_w,
                      float eps, void* normed, int64_t ld_normed, int norm_flags, int m, int n, int k,
                      void* workspace, size_t workspace_bytes, mv_stream_t stream) {
    if (!ln_w || !normed) return MV_ERR_INVALID_ARG;
    return mv::wgemm_impl(x, ldx, nullptr, 0, nullptr, 0.f, nullptr, 0, mv::kProNone, 0, w_tiles, bias,
                          residual, ld_res, y, ldy, m, n, k, workspace, workspace_bytes, 0, stream,
                          ln_w, eps, normed, ld_normed,
                          (norm_flags & MV_NORM_VAR_FROM_ROUNDED) ? 1 : 0);
}

int mv_wgemm_norm(const void* x, int64_t ldx, const void* add_r, int64_t ld_add_r,
                  const void* ln_w, float eps, void* sum_out, int64_t ld_sum_out, int norm_flags,
                  const void* w_tiles, const void* bias, const void* residual, int64_t ld_res,
                  void* y, int64_t ldy, int m, int n, int k, void* workspace,
                  size_t workspace_bytes, int flags, mv_stream_t stream) {
    if (!ln_w) return MV_ERR_INVALID_ARG;
    if ((add_r == nullptr) != (sum_out == nullptr)) return MV_ERR_INVALID_ARG;
    if (m > 4) return MV_ERR_UNSUPPORTED;  // the fused input transform lives in the dot form
    return mv::wgemm_impl(x, ldx, add_r, ld_add_r, ln_w, eps, sum_out, ld_sum_out,
                          add_r ? mv::kProAddNorm : mv::kProNorm,
                          (norm_flags & MV_NORM_VAR_FROM_ROUNDED) ? 1 : 0, w_tiles, bias, residual,
                          ld_res, y, ldy, m, n, k, workspace, workspace_bytes, flags, stream);
}

}  // extern "C"

#ifdef MV_WGEMM_STAMP
extern "C" int mv_debug_read_wgemm_stamps(void* dst, size_t bytes) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(mv::g_wstamp), bytes) == hipSuccess ? 0 : -3;
}
#endif
