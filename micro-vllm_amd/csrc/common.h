// Shared device/host helpers for the gfx950 kernels.  Wavefront = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mvllm.h"

namespace mv {

constexpr int kWave = 64;

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;

// ---- bf16 <-> f32 -------------------------------------------------------
__device__ __forceinline__ float bf16_bits_to_f32(uint32_t bits16) {
    return __uint_as_float(bits16 << 16);
}
// low / high bf16 of a packed dword
__device__ __forceinline__ float bf16_lo(uint32_t packed) { return __uint_as_float(packed << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t packed) {
    return __uint_as_float(packed & 0xffff0000u);
}
// round-to-nearest-even; a plain cast lowers to v_cvt_pk_bf16_f32 and keeps NaN a NaN
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
    __bf16 b = (__bf16)f;
    return (uint32_t)__builtin_bit_cast(unsigned short, b);
}
// both halves in one v_cvt_pk_bf16_f32 (the scalar cast above costs a convert, a shift and an or per pair)
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
// value after a bf16 round trip
__device__ __forceinline__ float round_bf16(float f) { return bf16_bits_to_f32(f32_to_bf16_bits(f)); }

// acc += a.lo*b.lo + a.hi*b.hi   (bf16 pairs, fp32 accumulate: v_dot2c_f32_bf16)
__device__ __forceinline__ float dot2_bf16(uint32_t a, uint32_t b, float acc) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a),
                                           __builtin_bit_cast(bf16x2_t, b), acc, false);
}

// SwiGLU on two projection accumulators: both are rounded to bf16 first (the reference
// materialises gate_up in bf16, kernel/swiglu.py:35-45), the product is formed in fp32.
__device__ __forceinline__ float silu_mul_rounded(float gate_acc, float up_acc) {
    const float g = round_bf16(gate_acc), u = round_bf16(up_acc);
    const float sig = 1.0f / (1.0f + expf(-g));
    return (g * sig) * u;
}

// ---- cross-lane ----------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// block-wide sum; `red` needs (blockDim.x/64) floats; every thread gets the total
__device__ __forceinline__ float block_sum(float v, float* red) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    if (lane == 0) red[wid] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];  // fixed order: deterministic
    __syncthreads();
    return t;
}

// ---- 16-byte global access -----------------------------------------------
__device__ __forceinline__ uint4 ld16(const void* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ void st16(void* p, uint4 v) { *reinterpret_cast<uint4*>(p) = v; }
// streamed-once data (weights): non-temporal so it does not displace reused lines
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
__device__ __forceinline__ uint4 ld16_nt(const void* p) {
    const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}

// ---- host side -------------------------------------------------------------
void set_last_hip_error(hipError_t e);
void set_last_error_text(const char* text);  // what mv_last_hip_error() returns next
inline int check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_last_hip_error(e);
        return MV_ERR_LAUNCH;
    }
    return MV_OK;
}
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
int num_cus();  // compute units of the current device (cached)
// Opt a kernel in to `lds` bytes of dynamic LDS (needed past 64 KiB) on the current device.
// `done` is the caller's per-kernel record of what each device was last configured for.
struct LdsOptIn { size_t bytes[64] = {}; };
int ensure_dynamic_lds(const void* kernel, size_t lds, LdsOptIn& done);
// wgemm.hip: adds k-slices partial[ks][m][n] (fp32) in order and applies the projection epilogue
int launch_wgemm_finalize(float* partial, int ksplit, int m, int n, int flags, const uint16_t* bias,
                          const uint16_t* residual, int64_t ld_res, void* y, int64_t ldy, hipStream_t s);

}  // namespace mv
