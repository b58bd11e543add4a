// SwiGLU, bf16 add, embedding row gather and the one-time weight relayouts.
// All are HBM/L2-bound byte movers: 16 bytes per lane, grid-stride, no LDS.
#include "common.h"

namespace mv {

constexpr int kEwThreads = 256;

__device__ __forceinline__ float silu_mul(float gate, float up) {
    // same evaluation order as the reference kernel: (gate * sigmoid(gate)) * up
    const float sig = 1.0f / (1.0f + expf(-gate));
    return (gate * sig) * up;
}

// gate_up: [rows, 2d], up = first half, gate = second half.  d % 8 == 0.
__global__ __launch_bounds__(kEwThreads) void swiglu_kernel(const uint16_t* __restrict__ gu,
                                                          uint16_t* __restrict__ out,
                                                          int64_t nvec_total, int dvec, int d) {
    for (int64_t v = (int64_t)blockIdx.x * kEwThreads + threadIdx.x; v < nvec_total;
         v += (int64_t)gridDim.x * kEwThreads) {
        const int64_t row = v / dvec;
        const int c8 = (int)(v - row * dvec) * 8;
        const uint16_t* base = gu + row * 2 * (int64_t)d + c8;
        const uint4 u = ld16(base), g = ld16(base + d);
        const uint32_t us[4] = {u.x, u.y, u.z, u.w}, gs[4] = {g.x, g.y, g.z, g.w};
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            o[j] = pack_bf16x2(silu_mul(bf16_lo(gs[j]), bf16_lo(us[j])),
                               silu_mul(bf16_hi(gs[j]), bf16_hi(us[j])));
        st16(out + row * (int64_t)d + c8, make_uint4(o[0], o[1], o[2], o[3]));
    }
}

// scalar form for d % 8 != 0 or unaligned pointers
__global__ __launch_bounds__(kEwThreads) void swiglu_scalar_kernel(const uint16_t* __restrict__ gu,
                                                                 uint16_t* __restrict__ out,
                                                                 int64_t n_out, int d) {
    for (int64_t i = (int64_t)blockIdx.x * kEwThreads + threadIdx.x; i < n_out;
         i += (int64_t)gridDim.x * kEwThreads) {
        const int64_t row = i / d;
        const int64_t col = i - row * d;
        const float up = bf16_bits_to_f32(gu[row * 2 * d + col]);
        const float gate = bf16_bits_to_f32(gu[row * 2 * d + col + d]);
        out[i] = (uint16_t)f32_to_bf16_bits(silu_mul(gate, up));
    }
}

// out = bf16(a + b[i % n_b]); n, n_b multiples of 8
__global__ __launch_bounds__(kEwThreads) void add_kernel(const uint16_t* __restrict__ a,
                                                       const uint16_t* __restrict__ b,
                                                       uint16_t* __restrict__ out, int64_t nvec,
                                                       int64_t nbvec) {
    for (int64_t v = (int64_t)blockIdx.x * kEwThreads + threadIdx.x; v < nvec;
         v += (int64_t)gridDim.x * kEwThreads) {
        const uint4 x = ld16(a + v * 8), y = ld16(b + (v % nbvec) * 8);
        const uint32_t xs[4] = {x.x, x.y, x.z, x.w}, ys[4] = {y.x, y.y, y.z, y.w};
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            o[j] = pack_bf16x2(bf16_lo(xs[j]) + bf16_lo(ys[j]), bf16_hi(xs[j]) + bf16_hi(ys[j]));
        st16(out + v * 8, make_uint4(o[0], o[1], o[2], o[3]));
    }
}

// one workgroup per output row
__global__ __launch_bounds__(kEwThreads) void embedding_kernel(const uint16_t* __restrict__ table,
                                                             const int64_t* __restrict__ ids,
                                                             uint16_t* __restrict__ out, int n,
                                                             int64_t vocab) {
    int64_t id = ids[blockIdx.x];
    if (id < 0) id = 0;
    if (id >= vocab) id = vocab - 1;  // never read outside the table
    const uint16_t* src = table + id * (int64_t)n;
    uint16_t* dst = out + (int64_t)blockIdx.x * n;
    for (int c = threadIdx.x; c < (n >> 3); c += kEwThreads) st16(dst + c * 8, ld16(src + c * 8));
    for (int c = ((n >> 3) << 3) + threadIdx.x; c < n; c += kEwThreads) dst[c] = src[c];
}

// [N,K] row-major -> wave tiles.  One 16-byte chunk per thread: chunk id = output
// position, so the stores are perfectly coalesced.
__global__ __launch_bounds__(kEwThreads) void shuffle_weight_kernel(const uint16_t* __restrict__ w,
                                                                  uint16_t* __restrict__ out,
                                                                  int64_t nchunks, int kt_count,
                                                                  int k) {
    for (int64_t c = (int64_t)blockIdx.x * kEwThreads + threadIdx.x; c < nchunks;
         c += (int64_t)gridDim.x * kEwThreads) {
        const int lane = (int)(c & 63);
        const int64_t tile = c >> 6;
        const int64_t nt = tile / kt_count;
        const int kt = (int)(tile - nt * kt_count);
        const int64_t row = nt * 16 + (lane & 15);
        const int col = kt * 32 + (lane >> 4) * 8;
        st16(out + c * 8, ld16(w + row * (int64_t)k + col));
    }
}

__global__ __launch_bounds__(kEwThreads) void interleave_rows8_kernel(
    const uint16_t* __restrict__ a, const uint16_t* __restrict__ b, uint16_t* __restrict__ out,
    int64_t nchunks, int kvec) {
    for (int64_t c = (int64_t)blockIdx.x * kEwThreads + threadIdx.x; c < nchunks;
         c += (int64_t)gridDim.x * kEwThreads) {
        const int64_t orow = c / kvec;
        const int cv = (int)(c - orow * kvec);
        const int64_t grp = orow >> 4;
        const int within = (int)(orow & 15);
        const uint16_t* src = (within < 8 ? a : b) + ((grp * 8 + (within & 7)) * (int64_t)kvec + cv) * 8;
        st16(out + c * 8, ld16(src));
    }
}

static int ew_grid(int64_t work_items) {
    int64_t blocks = (work_items + kEwThreads - 1) / kEwThreads;
    const int64_t cap = (int64_t)num_cus() * 8;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

}  // namespace mv

extern "C" {

int mv_swiglu(const void* gate_up, void* out, int64_t rows, int d, mv_stream_t stream) {
    if (rows == 0 && d > 0) return MV_OK;
    if (!gate_up || !out || rows < 0 || d <= 0) return MV_ERR_INVALID_ARG;
    if ((d % 8) == 0 && mv::aligned16(gate_up) && mv::aligned16(out)) {
        const int64_t nvec = rows * (int64_t)(d / 8);
        hipLaunchKernelGGL(mv::swiglu_kernel, dim3(mv::ew_grid(nvec)), dim3(mv::kEwThreads), 0,
                           (hipStream_t)stream, (const uint16_t*)gate_up, (uint16_t*)out, nvec,
                           d / 8, d);
    } else {
        const int64_t n_out = rows * (int64_t)d;
        hipLaunchKernelGGL(mv::swiglu_scalar_kernel, dim3(mv::ew_grid(n_out)),
                           dim3(mv::kEwThreads), 0, (hipStream_t)stream,
                           (const uint16_t*)gate_up, (uint16_t*)out, n_out, d);
    }
    return mv::check_launch();
}

int mv_add(const void* a, const void* b, void* out, int64_t n, int64_t n_b, mv_stream_t stream) {
    if (n == 0 && n_b > 0) return MV_OK;
    if (!a || !b || !out || n < 0 || n_b <= 0) return MV_ERR_INVALID_ARG;
    if ((n % n_b) != 0 || (n_b % 8) != 0) return MV_ERR_UNSUPPORTED;
    if (!mv::aligned16(a) || !mv::aligned16(b) || !mv::aligned16(out)) return MV_ERR_INVALID_ARG;
    hipLaunchKernelGGL(mv::add_kernel, dim3(mv::ew_grid(n / 8)), dim3(mv::kEwThreads), 0,
                       (hipStream_t)stream, (const uint16_t*)a, (const uint16_t*)b, (uint16_t*)out,
                       n / 8, n_b / 8);
    return mv::check_launch();
}

int mv_embedding(const void* table, const int64_t* ids, void* out, int rows, int n, int64_t vocab,
                 mv_stream_t stream) {
    if (rows == 0 && n > 0) return MV_OK;
    if (!table || !ids || !out || rows < 0 || n <= 0 || vocab <= 0) return MV_ERR_INVALID_ARG;
    if ((n % 8) != 0) return MV_ERR_UNSUPPORTED;
    if (!mv::aligned16(table) || !mv::aligned16(out)) return MV_ERR_INVALID_ARG;
    hipLaunchKernelGGL(mv::embedding_kernel, dim3(rows), dim3(mv::kEwThreads), 0,
                       (hipStream_t)stream, (const uint16_t*)table, ids, (uint16_t*)out, n, vocab);
    return mv::check_launch();
}

int mv_shuffle_weight(const void* w_nk, void* w_tiles, int n, int k, mv_stream_t stream) {
    if (!w_nk || !w_tiles || n <= 0 || k <= 0) return MV_ERR_INVALID_ARG;
    if ((n % 16) != 0 || (k % 32) != 0) return MV_ERR_UNSUPPORTED;
    if (!mv::aligned16(w_nk) || !mv::aligned16(w_tiles) || w_nk == w_tiles)
        return MV_ERR_INVALID_ARG;
    const int64_t nchunks = (int64_t)n * k / 8;
    hipLaunchKernelGGL(mv::shuffle_weight_kernel, dim3(mv::ew_grid(nchunks)),
                       dim3(mv::kEwThreads), 0, (hipStream_t)stream, (const uint16_t*)w_nk,
                       (uint16_t*)w_tiles, nchunks, k / 32, k);
    return mv::check_launch();
}

int mv_interleave_rows8(const void* a, const void* b, void* out, int rows_each, int k,
                        mv_stream_t stream) {
    if (!a || !b || !out || rows_each <= 0 || k <= 0) return MV_ERR_INVALID_ARG;
    if ((rows_each % 8) != 0 || (k % 8) != 0) return MV_ERR_UNSUPPORTED;
    if (!mv::aligned16(a) || !mv::aligned16(b) || !mv::aligned16(out)) return MV_ERR_INVALID_ARG;
    const int64_t nchunks = 2 * (int64_t)rows_each * (k / 8);
    hipLaunchKernelGGL(mv::interleave_rows8_kernel, dim3(mv::ew_grid(nchunks)),
                       dim3(mv::kEwThreads), 0, (hipStream_t)stream, (const uint16_t*)a,
                       (const uint16_t*)b, (uint16_t*)out, nchunks, k / 8);
    return mv::check_launch();
}

}  // extern "C"
