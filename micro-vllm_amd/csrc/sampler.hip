// Greedy token pick: ids[r] = argmax_c logits[r, c] (lowest index on ties), one workgroup per row.
// Reads the bf16 logits row once with 16-byte loads.
#include "common.h"

namespace mv {

constexpr int kAmThreads = 1024;

__device__ __forceinline__ void amax_merge(float& v, int& i, float ov, int oi) {
    if (ov > v || (ov == v && oi < i)) {
        v = ov;
        i = oi;
    }
}

__global__ __launch_bounds__(kAmThreads) void argmax_kernel(const uint16_t* __restrict__ logits,
                                                          int64_t ld, int n,
                                                          int64_t* __restrict__ ids) {
    __shared__ float sv[kAmThreads / 64];
    __shared__ int si[kAmThreads / 64];
    const uint16_t* row = logits + (int64_t)blockIdx.x * ld;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    const int nvec = n >> 3;
    for (int c = threadIdx.x; c < nvec; c += kAmThreads) {
        const uint4 v = ld16(row + c * 8);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            amax_merge(best, bi, bf16_lo(w[j]), c * 8 + 2 * j);
            amax_merge(best, bi, bf16_hi(w[j]), c * 8 + 2 * j + 1);
        }
    }
    for (int c = (nvec << 3) + threadIdx.x; c < n; c += kAmThreads)
        amax_merge(best, bi, bf16_bits_to_f32(row[c]), c);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        amax_merge(best, bi, __shfl_xor(best, o, 64), __shfl_xor(bi, o, 64));
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) {
        sv[wid] = best;
        si[wid] = bi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kAmThreads / 64; ++w) amax_merge(best, bi, sv[w], si[w]);
        ids[blockIdx.x] = bi == 0x7fffffff ? 0 : bi;  // all-NaN row: index 0
    }
}

}  // namespace mv

extern "C" int mv_argmax(const void* logits, int64_t ld, int rows, int n, int64_t* ids,
                         mv_stream_t stream) {
    if (rows == 0 && n > 0) return MV_OK;
    if (!logits || !ids || rows < 0 || n <= 0 || ld < n) return MV_ERR_INVALID_ARG;
    if (!mv::aligned16(logits) || (ld % 8) != 0) return MV_ERR_INVALID_ARG;
    hipLaunchKernelGGL(mv::argmax_kernel, dim3(rows), dim3(mv::kAmThreads), 0, (hipStream_t)stream,
                       (const uint16_t*)logits, ld, n, ids);
    return mv::check_launch();
}
