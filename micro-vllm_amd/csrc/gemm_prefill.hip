// Many-token projection  y[M,N] = x[M,K] . W^T  on the matrix cores, over the same wave-tile
// weight image the single-token kernels stream (tile (nt,kt) = 16 rows x 32 k = 1 KiB, lane l
// owns W[16nt + (l&15)][32kt + 8(l>>4) + 0..7]): that image IS the A-operand fragment order of
// v_mfma_f32_16x16x32_bf16, so weight fragments go from global memory straight into registers,
// one fully coalesced 1 KiB request per tile, and only x passes through LDS.
//
// Workgroup = 4 waves = 256 (n) x 128 (m) outputs; wave w owns n-tiles 4w..4w+3 against all
// eight m-tiles (128 fp32 accumulators per lane).  x tiles of 128 rows x 64 k are staged in
// LDS twice buffered (row stride 144 B: conflict-free 16-byte fragment reads); the weight
// fragments of the next k-tile are requested while the current one is multiplied.
//
// Workgroups are numbered so that the ones resident together on an XCD (blockIdx mod 8) share
// few n-blocks and many m-blocks: each weight fragment is then fetched into that XCD's L2 once
// and hit by the other m-blocks.
//
// Epilogue rounding as in the single-token kernels: t = bf16(acc); t = bf16(t + bias);
// t = bf16(t + residual); SWIGLU: out = bf16(silu(gate_t) * up_t) with up/gate rows interleaved
// by 8 inside each 16-row tile.
#include "common.h"

namespace mv {

constexpr int kPM = 128;            // rows of x per workgroup
constexpr int kPN = 256;            // output columns per workgroup
constexpr int kPK = 64;             // k per LDS stage
constexpr int kPRow = kPK * 2 + 16; // LDS row stride in bytes
constexpr int kPMT = kPM / 16;      // m-tiles per wave
constexpr int kPNT = 4;             // n-tiles per wave

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;

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
    int flags;
};

__global__ __launch_bounds__(256, 2) void gemm_prefill_kernel(PGemmArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char xs[2][kPM * kPRow];

    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, q = lane >> 4;

    // blockIdx -> (m-block, n-block): an XCD takes a contiguous range of n-blocks and walks the
    // m-blocks fastest
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int nb = xcd * a.nb_per_xcd + idx / a.mblocks;
    const int mb = idx % a.mblocks;
    if (nb >= a.nblocks || idx / a.mblocks >= a.nb_per_xcd) return;
    const int m0 = mb * kPM;
    const int nt0 = nb * (kPN / 16) + wid * kPNT;

    // weight fragment streams of this wave (strips past the end are clamped and never stored)
    const uint4* wp[kPNT];
#pragma unroll
    for (int i = 0; i < kPNT; ++i)
        wp[i] = a.w + (int64_t)min(nt0 + i, a.nt - 1) * a.kt * 64 + lane;

    // x staging: thread t moves vectors t + 256 j (j < 4) of the 128 x 8 vector tile
    const uint16_t* xg[4];
    int xl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int v = threadIdx.x + 256 * j, row = v >> 3, k8 = v & 7;
        xg[j] = a.x + (int64_t)min(m0 + row, a.m - 1) * a.ldx + k8 * 8;
        xl[j] = row * kPRow + k8 * 16;
    }
    const int nstage = a.kt >> 1;  // K is a multiple of 64 (checked on the host)
    auto x_load = [&](uint4(&xr)[4], int s) {
#pragma unroll
        for (int j = 0; j < 4; ++j) xr[j] = ld16(xg[j] + s * kPK);
    };

    f32x4_t acc[kPNT][kPMT];
#pragma unroll
    for (int i = 0; i < kPNT; ++i)
#pragma unroll
        for (int t = 0; t < kPMT; ++t) acc[i][t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    uint4 xr[4], wc[kPNT], wn[kPNT];
    x_load(xr, 0);
#pragma unroll
    for (int i = 0; i < kPNT; ++i) wc[i] = ld16(wp[i]);
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<uint4*>(&xs[0][xl[j]]) = xr[j];
    __syncthreads();

    const int frag_off = r * kPRow + q * 16;
    for (int s = 0; s < nstage; ++s) {
        const unsigned char* xb = xs[s & 1];
        const bool more = s + 1 < nstage;
        if (more) x_load(xr, s + 1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int kt = 2 * s + kk;
            {
                const int ktn = min(kt + 1, a.kt - 1);
#pragma unroll
                for (int i = 0; i < kPNT; ++i) wn[i] = ld16(wp[i] + (int64_t)ktn * 64);
                bf16x8_t xf[kPMT];
#pragma unroll
                for (int t = 0; t < kPMT; ++t)
                    xf[t] = __builtin_bit_cast(
                        bf16x8_t, *reinterpret_cast<const uint4*>(xb + t * 16 * kPRow + kk * 64 + frag_off));
#pragma unroll
                for (int i = 0; i < kPNT; ++i) {
                    const bf16x8_t wf = __builtin_bit_cast(bf16x8_t, wc[i]);
#pragma unroll
                    for (int t = 0; t < kPMT; ++t)
                        acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf[t], acc[i][t], 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < kPNT; ++i) wc[i] = wn[i];
            }
        }
        if (more) {
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<uint4*>(&xs[(s + 1) & 1][xl[j]]) = xr[j];
        }
        __syncthreads();
    }

    // acc[i][t][j] = C[n = 16 (nt0+i) + 4q + j][m = m0 + 16 t + r]
    const bool sw = (a.flags & MV_GEMM_SWIGLU) != 0, of32 = (a.flags & MV_GEMM_OUT_F32) != 0;
#pragma unroll
    for (int i = 0; i < kPNT; ++i) {
        const int nt = nt0 + i;
        if (nt >= a.nt) continue;  // uniform
        const int n0 = nt * 16 + q * 4;
#pragma unroll
        for (int t = 0; t < kPMT; ++t) {
            const int m = m0 + t * 16 + r;
            const f32x4_t c = acc[i][t];
            if (sw) {
                // rows 0..7 of the tile are up, 8..15 the matching gates: 32 lanes further
                f32x4_t g;
#pragma unroll
                for (int j = 0; j < 4; ++j) g[j] = __shfl_xor(c[j], 32, 64);
                if (q < 2 && m < a.m) {
                    uint2 o;
                    o.x = pack_bf16x2(silu_mul_rounded(g[0], c[0]), silu_mul_rounded(g[1], c[1]));
                    o.y = pack_bf16x2(silu_mul_rounded(g[2], c[2]), silu_mul_rounded(g[3], c[3]));
                    *reinterpret_cast<uint2*>(
                        &reinterpret_cast<uint16_t*>(a.y)[(int64_t)m * a.ldy + nt * 8 + q * 4]) = o;
                }
            } else if (m < a.m) {
                if (of32) {
                    *reinterpret_cast<f32x4_t*>(&reinterpret_cast<float*>(a.y)[(int64_t)m * a.ldy + n0]) = c;
                } else {
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float tv = round_bf16(c[j]);
                        if (a.bias) tv = round_bf16(tv + bf16_bits_to_f32(a.bias[n0 + j]));
                        if (a.residual)
                            tv = round_bf16(tv + bf16_bits_to_f32(a.residual[(int64_t)m * a.ld_res + n0 + j]));
                        v[j] = tv;
                    }
                    uint2 o;
                    o.x = pack_bf16x2(v[0], v[1]);
                    o.y = pack_bf16x2(v[2], v[3]);
                    *reinterpret_cast<uint2*>(&reinterpret_cast<uint16_t*>(a.y)[(int64_t)m * a.ldy + n0]) = o;
                }
            }
        }
    }
}

}  // namespace mv

extern "C" int mv_gemm_prefill(const void* x, int64_t ldx, const void* w_tiles, const void* bias,
                               const void* residual, int64_t ld_res, void* y, int64_t ldy, int m,
                               int n, int k, int flags, mv_stream_t stream) {
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
    a.flags = flags;
    const long grid = 8L * a.nb_per_xcd * a.mblocks;
    if (grid > 0x7fffffffL) return MV_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(gemm_prefill_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch();
}
