// Scatter new key/value rows into the paged cache by slot number.
// One 16-byte chunk per thread; a (token, head) row of D bf16 is D/8 adjacent lanes.
#include "common.h"

namespace mv {

__global__ __launch_bounds__(256) void store_kv_kernel(
    const uint16_t* __restrict__ key, const uint16_t* __restrict__ value, int64_t kv_stride_t,
    uint16_t* __restrict__ k_cache, uint16_t* __restrict__ v_cache,
    const int32_t* __restrict__ slot_mapping, int64_t nchunks, int heads, int dvec, int page,
    int num_pages) {
    const int hd = heads * dvec;  // chunks per token
    for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < nchunks;
         c += (int64_t)gridDim.x * 256) {
        const int64_t t = c / hd;
        const int within = (int)(c - t * hd);
        const int slot = slot_mapping[t];
        if (slot < 0 || slot >= (int64_t)num_pages * page) continue;  // -1 = skip
        const int64_t src = t * kv_stride_t + (int64_t)within * 8;
        const int64_t dst = ((int64_t)slot * hd + within) * 8;
        st16(k_cache + dst, ld16(key + src));
        st16(v_cache + dst, ld16(value + src));
    }
}

}  // namespace mv

extern "C" int mv_store_kv(const void* key, const void* value, int64_t kv_stride_t, void* k_cache,
                           void* v_cache, const int32_t* slot_mapping, int tokens, int heads,
                           int d, int page, int num_pages, mv_stream_t stream) {
    if (!key || !value || !k_cache || !v_cache || !slot_mapping || tokens < 0 || heads <= 0 ||
        d <= 0 || page <= 0 || num_pages <= 0)
        return MV_ERR_INVALID_ARG;
    if (tokens == 0) return MV_OK;
    if ((d % 8) != 0 || (kv_stride_t % 8) != 0) return MV_ERR_UNSUPPORTED;
    if (kv_stride_t < (int64_t)heads * d) return MV_ERR_INVALID_ARG;
    if (!mv::aligned16(key) || !mv::aligned16(value) || !mv::aligned16(k_cache) ||
        !mv::aligned16(v_cache))
        return MV_ERR_INVALID_ARG;
    const int64_t nchunks = (int64_t)tokens * heads * (d / 8);
    int64_t blocks = (nchunks + 255) / 256;
    const int64_t cap = (int64_t)mv::num_cus() * 8;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(mv::store_kv_kernel, dim3((unsigned)blocks), dim3(256), 0,
                       (hipStream_t)stream, (const uint16_t*)key, (const uint16_t*)value,
                       kv_stride_t, (uint16_t*)k_cache, (uint16_t*)v_cache, slot_mapping, nchunks,
                       heads, d / 8, page, num_pages);
    return mv::check_launch();
}
