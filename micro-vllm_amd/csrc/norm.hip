// RMSNorm and add+RMSNorm: one workgroup per row, 16-byte vector access, fp32 math.
//
// Rounding contract (what the callers' parity tests pin):
//   rmsnorm:          y = bf16( (x * rrms) * w )
//   rmsnorm_residual: s = x + r in fp32; res_out = bf16(s); the variance is taken
//                     from the UNROUNDED s; y = bf16( (float(res_out) * rrms) * w )
//   rrms = 1 / sqrt(sum(.)/N + eps)
#include "common.h"

namespace mv {

constexpr int kNormThreads = 256;

template <bool kResidual>
__global__ __launch_bounds__(kNormThreads) void rmsnorm_kernel(
    const uint16_t* __restrict__ x, const uint16_t* __restrict__ r, const uint16_t* __restrict__ w,
    uint16_t* __restrict__ y, uint16_t* __restrict__ res_out, int n, int64_t stride_x,
    int64_t stride_r, int64_t stride_y, int64_t stride_res, float eps) {
    __shared__ float red[kNormThreads / 64];
    const int row = blockIdx.x;
    const uint16_t* xr = x + (int64_t)row * stride_x;
    const uint16_t* rr = kResidual ? r + (int64_t)row * stride_r : nullptr;
    uint16_t* yr = y + (int64_t)row * stride_y;
    uint16_t* sr = kResidual ? res_out + (int64_t)row * stride_res : nullptr;

    const int nvec = n >> 3;  // 8 bf16 per 16-byte chunk
    float ss = 0.f;
    for (int c = threadIdx.x; c < nvec; c += kNormThreads) {
        uint4 xv = ld16(xr + c * 8);
        const uint32_t xs[4] = {xv.x, xv.y, xv.z, xv.w};
        if (kResidual) {
            uint4 rv = ld16(rr + c * 8);
            const uint32_t rs[4] = {rv.x, rv.y, rv.z, rv.w};
            uint32_t o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float s0 = bf16_lo(xs[j]) + bf16_lo(rs[j]);
                float s1 = bf16_hi(xs[j]) + bf16_hi(rs[j]);
                ss += s0 * s0;
                ss += s1 * s1;
                o[j] = pack_bf16x2(s0, s1);
            }
            st16(sr + c * 8, make_uint4(o[0], o[1], o[2], o[3]));
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float a = bf16_lo(xs[j]), b = bf16_hi(xs[j]);
                ss += a * a;
                ss += b * b;
            }
        }
    }
    // tail (n not a multiple of 8): scalar
    for (int c = (nvec << 3) + threadIdx.x; c < n; c += kNormThreads) {
        float s = bf16_bits_to_f32(xr[c]);
        if (kResidual) {
            s += bf16_bits_to_f32(rr[c]);
            sr[c] = (uint16_t)f32_to_bf16_bits(s);
        }
        ss += s * s;
    }
    const float total = block_sum(ss, red);
    const float rrms = 1.0f / sqrtf(total / (float)n + eps);

    // second pass: each thread re-reads exactly the columns it wrote
    const uint16_t* src = kResidual ? sr : xr;
    for (int c = threadIdx.x; c < nvec; c += kNormThreads) {
        uint4 sv = ld16(src + c * 8);
        uint4 wv = ld16(w + c * 8);
        const uint32_t s4[4] = {sv.x, sv.y, sv.z, sv.w};
        const uint32_t w4[4] = {wv.x, wv.y, wv.z, wv.w};
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a = (bf16_lo(s4[j]) * rrms) * bf16_lo(w4[j]);
            float b = (bf16_hi(s4[j]) * rrms) * bf16_hi(w4[j]);
            o[j] = pack_bf16x2(a, b);
        }
        st16(yr + c * 8, make_uint4(o[0], o[1], o[2], o[3]));
    }
    for (int c = (nvec << 3) + threadIdx.x; c < n; c += kNormThreads) {
        float a = (bf16_bits_to_f32(src[c]) * rrms) * bf16_bits_to_f32(w[c]);
        yr[c] = (uint16_t)f32_to_bf16_bits(a);
    }
}

static bool rows_vec_ok(const void* p, int64_t stride) { return aligned16(p) && (stride % 8) == 0; }

}  // namespace mv

extern "C" {

int mv_rmsnorm(const void* x, const void* w, void* y, int rows, int n, int64_t stride_x,
               int64_t stride_y, float eps, mv_stream_t stream) {
    if (rows == 0 && n > 0) return MV_OK;  // nothing to do (empty tensors carry null pointers)
    if (!x || !w || !y || rows < 0 || n <= 0) return MV_ERR_INVALID_ARG;
    if (!mv::rows_vec_ok(x, stride_x) || !mv::rows_vec_ok(y, stride_y) || !mv::aligned16(w))
        return MV_ERR_INVALID_ARG;
    hipLaunchKernelGGL(mv::rmsnorm_kernel<false>, dim3(rows), dim3(mv::kNormThreads), 0,
                       (hipStream_t)stream, (const uint16_t*)x, nullptr, (const uint16_t*)w,
                       (uint16_t*)y, nullptr, n, stride_x, 0, stride_y, 0, eps);
    return mv::check_launch();
}

int mv_rmsnorm_residual(const void* x, const void* r, const void* w, void* y, void* res_out,
                        int rows, int n, int64_t stride_x, int64_t stride_r, int64_t stride_y,
                        int64_t stride_res, float eps, mv_stream_t stream) {
    if (rows == 0 && n > 0) return MV_OK;
    if (!x || !r || !w || !y || !res_out || rows < 0 || n <= 0) return MV_ERR_INVALID_ARG;
    if (!mv::rows_vec_ok(x, stride_x) || !mv::rows_vec_ok(r, stride_r) ||
        !mv::rows_vec_ok(y, stride_y) || !mv::rows_vec_ok(res_out, stride_res) ||
        !mv::aligned16(w))
        return MV_ERR_INVALID_ARG;
    hipLaunchKernelGGL(mv::rmsnorm_kernel<true>, dim3(rows), dim3(mv::kNormThreads), 0,
                       (hipStream_t)stream, (const uint16_t*)x, (const uint16_t*)r,
                       (const uint16_t*)w, (uint16_t*)y, (uint16_t*)res_out, n, stride_x, stride_r,
                       stride_y, stride_res, eps);
    return mv::check_launch();
}

}  // extern "C"
