// Token selection on the device, one workgroup per row of bf16 logits.
//   argmax_kernel : greedy pick (lowest index on ties); reads the row once with 16-byte loads.
//   sample_kernel : temperature -> top-k -> softmax -> top-p -> draw, the recipe of the reference's
//                   Sampler._sample_impl (core/layer/sampler.py:21-43), as one launch instead of
//                   topk + softmax + sort + cumsum + multinomial + two gathers.  The k largest
//                   logits are found by a two-pass radix select on the 16 bf16 bits (division by a
//                   positive temperature does not change the order), gathered into LDS (ties at
//                   the threshold by ascending index, so the result is deterministic), sorted there
//                   (bitonic), and the draw is the inverse CDF of the caller's uniform number.
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

// Few rows: one workgroup per row would leave the chip idle and read a 128k-entry row through a
// single CU (about 15 us).  The row is cut into parts, one workgroup each; partial results go to the
// caller's workspace (tickets | values | indices) with write-through stores, and the workgroup that
// draws the last arrival ticket of a row merges them.  The tickets are zero when the caller hands
// the workspace over for the first time and are returned to zero by every launch.  The merge is a maximum under a total order (value, then lowest
// index), so the result does not depend on arrival order.
constexpr int kAmPartThreads = 256;
constexpr int kAmMaxParts = 64;
constexpr int kAmMaxRows = 64;
constexpr size_t kAmTicketBytes = (size_t)kAmMaxRows * 64 * sizeof(unsigned int);  // 256 bytes apart
constexpr size_t kAmPartBytes = (size_t)kAmMaxRows * kAmMaxParts * 4;
constexpr size_t kAmWorkspaceBytes = kAmTicketBytes + 2 * kAmPartBytes;

struct AmScratch {
    float* val;
    int* idx;
    unsigned int* ticket;
};

__global__ __launch_bounds__(kAmPartThreads) void argmax_parts_kernel(const uint16_t* __restrict__ logits,
                                                                      int64_t ld, int n, int parts,
                                                                      int vec_per_part, AmScratch sc,
                                                                      int64_t* __restrict__ ids) {
    __shared__ float sv[kAmPartThreads / 64];
    __shared__ int si[kAmPartThreads / 64];
    __shared__ int is_last;
    const int rowi = blockIdx.x / parts, part = blockIdx.x - rowi * parts;
    const uint16_t* row = logits + (int64_t)rowi * ld;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    const int nvec = n >> 3;
    const int c0 = part * vec_per_part, c1 = min(nvec, c0 + vec_per_part);
    for (int c = c0 + threadIdx.x; c < c1; c += kAmPartThreads) {
        const uint4 v = ld16(row + c * 8);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            amax_merge(best, bi, bf16_lo(w[j]), c * 8 + 2 * j);
            amax_merge(best, bi, bf16_hi(w[j]), c * 8 + 2 * j + 1);
        }
    }
    if (part == parts - 1)
        for (int c = (nvec << 3) + threadIdx.x; c < n; c += kAmPartThreads)
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
        for (int w = 1; w < kAmPartThreads / 64; ++w) amax_merge(best, bi, sv[w], si[w]);
        const int slot = rowi * kAmMaxParts + part;
        __hip_atomic_store(reinterpret_cast<unsigned int*>(sc.val + slot), __float_as_uint(best),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(sc.idx + slot, bi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // both stores have left before the ticket
        const unsigned old = __hip_atomic_fetch_add(sc.ticket + rowi * 64, 1u, __ATOMIC_RELAXED,
                                                    __HIP_MEMORY_SCOPE_AGENT);
        is_last = old == (unsigned)parts - 1u;
        if (is_last)  // back to zero for the next launch; nobody else touches it in this one
            __hip_atomic_store(sc.ticket + rowi * 64, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!is_last || threadIdx.x >= 64) return;
    best = -INFINITY;
    bi = 0x7fffffff;
    if (lane < parts) {
        const int slot = rowi * kAmMaxParts + lane;
        best = __uint_as_float(__hip_atomic_load(reinterpret_cast<unsigned int*>(sc.val + slot),
                                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        bi = __hip_atomic_load(sc.idx + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        amax_merge(best, bi, __shfl_xor(best, o, 64), __shfl_xor(bi, o, 64));
    if (lane == 0) ids[rowi] = bi == 0x7fffffff ? 0 : bi;  // all-NaN row: index 0
}

static AmScratch am_scratch(void* workspace) {
    unsigned char* p = (unsigned char*)workspace;
    return AmScratch{(float*)(p + kAmTicketBytes), (int*)(p + kAmTicketBytes + kAmPartBytes),
                     (unsigned int*)p};
}

// ---------------------------------------------------------------- top-k / top-p sampling
constexpr int kSmThreads = 1024;
constexpr int kSmMaxK = 1024;
constexpr int kSmKeep = 16;   // 16-byte chunks of the row a thread keeps in registers

// order-preserving 16-bit key of a bf16 value (larger key = larger value); NaN sorts lowest
__device__ __forceinline__ unsigned bf16_key(unsigned bits) {
    if ((bits & 0x7f80u) == 0x7f80u && (bits & 0x7fu)) return 0u;
    return (bits & 0x8000u) ? (~bits & 0xffffu) : (bits | 0x8000u);
}

// inclusive scan over the workgroup (one value per thread); `ws` holds one slot per wave
template <typename T>
__device__ __forceinline__ T block_scan_incl(T v, T* ws, T& total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const T u = __shfl_up(v, o, 64);
        if (lane >= o) v += u;
    }
    __syncthreads();  // ws may still be read from a previous scan
    if (lane == 63) ws[wid] = v;
    __syncthreads();
    T base = 0, tot = 0;
    for (int w = 0; w < kSmThreads / 64; ++w) {
        const T x = ws[w];
        if (w < wid) base += x;
        tot += x;
    }
    total = tot;
    return v + base;
}

__global__ __launch_bounds__(kSmThreads) void sample_kernel(const uint16_t* __restrict__ logits,
                                                          int64_t ld, int n,
                                                          const float* __restrict__ temps,
                                                          const float* __restrict__ top_ps, int top_k,
                                                          const float* __restrict__ uniforms,
                                                          const int64_t* __restrict__ greedy_ids,
                                                          int64_t* __restrict__ ids) {
    // A row whose greedy pick is already known (temperature <= 0 and the caller has the argmax)
    // is done at once: the kernel can then sit in a captured decode step at no cost for greedy
    // batches.
    if (greedy_ids != nullptr && !(temps[blockIdx.x] > 0.f)) {  // uniform
        if (threadIdx.x == 0) ids[blockIdx.x] = greedy_ids[blockIdx.x];
        return;
    }
    // Sixteen copies of the 256-bin histogram, one per lane group: logits share a handful of
    // exponents, so with a single copy nearly every lane of a wave adds to the same LDS word and the
    // adds are served one after the other.  A thread also counts runs of equal bins in a register
    // and adds a whole run at once.
    __shared__ unsigned hist[16][256];
    __shared__ unsigned sh_u[8];
    __shared__ float c_val[kSmMaxK];
    __shared__ int c_idx[kSmMaxK];
    __shared__ float ws_f[kSmThreads / 64];
    __shared__ unsigned ws_u[kSmThreads / 64];

    const int tid = threadIdx.x;
    const uint16_t* row = logits + (int64_t)blockIdx.x * ld;
    const int nvec = n >> 3;
    const int k = min(min(top_k, n), kSmMaxK);

    // The row is read ONCE: the first kSmKeep chunks of every thread (rows up to 131 072 entries
    // whole) stay in registers for all three passes, requested back to back so that the read costs
    // one memory round trip instead of one per chunk and pass; longer rows stream the rest.
    uint4 keep[kSmKeep];
#pragma unroll
    for (int i = 0; i < kSmKeep; ++i) {
        const int c = tid + i * kSmThreads;
        keep[i] = ld16(row + (int64_t)min(c, nvec - 1) * 8);
    }
    auto for_each_chunk = [&](auto&& fn) {  // fn(chunk index, 8 values), every thread the same trip count
#pragma unroll
        for (int i = 0; i < kSmKeep; ++i) fn(tid + i * kSmThreads, keep[i]);
        for (int c0 = kSmKeep * kSmThreads; c0 < nvec; c0 += kSmThreads) {
            const int c = c0 + tid;
            fn(c, ld16(row + (int64_t)min(c, nvec - 1) * 8));
        }
    };
    auto for_each = [&](auto&& fn) {  // fn(key, index) over the row
        for_each_chunk([&](int c, const uint4& v) {
            if (c >= nvec) return;
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                fn(bf16_key(w[j] & 0xffffu), c * 8 + 2 * j);
                fn(bf16_key(w[j] >> 16), c * 8 + 2 * j + 1);
            }
        });
    };
    // the bin (scanning from the top) in which the running count reaches `want`
    auto pick_bin = [&](unsigned want, unsigned& bin, unsigned& above) {
        __syncthreads();
        if (tid < 256) {
            unsigned t = 0;
#pragma unroll
            for (int c = 0; c < 16; ++c) t += hist[c][tid];
            hist[0][tid] = t;
        }
        __syncthreads();
        // suffix sums from bin 255 down, all bins at once: thread t holds bin 255 - t; the first
        // thread whose running count reaches `want` names the bin (bin 0 if none does)
        const unsigned h = tid < 256 ? hist[0][255 - tid] : 0u;
        unsigned total;
        const unsigned incl = block_scan_incl(h, ws_u, total);
        if (tid == 0) sh_u[0] = 255u;
        __syncthreads();
        if (tid < 255 && incl >= want) atomicMin(&sh_u[0], (unsigned)tid);
        __syncthreads();
        if ((unsigned)tid == sh_u[0]) sh_u[1] = incl - h;
        __syncthreads();
        if (tid == 0) sh_u[0] = 255u - sh_u[0];
        __syncthreads();
        bin = sh_u[0];
        above = sh_u[1];
        __syncthreads();
    };

    // ---- radix select of the k-th largest key: high byte, then low byte
    unsigned* my_hist = hist[tid & 15];
    unsigned run_bin = 0, run_len = 0;
    auto count = [&](unsigned bin) {
        if (bin != run_bin) {
            if (run_len) atomicAdd(&my_hist[run_bin], run_len);
            run_bin = bin;
            run_len = 0;
        }
        ++run_len;
    };
    auto flush = [&]() {
        if (run_len) atomicAdd(&my_hist[run_bin], run_len);
        run_len = 0;
    };
    for (int i = tid; i < 16 * 256; i += kSmThreads) (&hist[0][0])[i] = 0;
    __syncthreads();
    for_each([&](unsigned key, int) { count(key >> 8); });
    flush();
    unsigned b1, above1;
    pick_bin((unsigned)k, b1, above1);
    for (int i = tid; i < 16 * 256; i += kSmThreads) (&hist[0][0])[i] = 0;
    __syncthreads();
    for_each([&](unsigned key, int) {
        if ((key >> 8) == b1) count(key & 255u);
    });
    flush();
    unsigned b2, above2;
    pick_bin((unsigned)k - above1, b2, above2);
    const unsigned thr = (b1 << 8) | b2;
    const unsigned n_gt = above1 + above2;           // keys strictly above the threshold (< k)
    const unsigned need_eq = (unsigned)k - n_gt;     // threshold-valued entries to take, lowest indices first

    // ---- gather: keys above the threshold in any order, threshold ties in index order
    if (tid == 0) sh_u[2] = 0;
    __syncthreads();
    unsigned eq_base = 0;
    for_each_chunk([&](int c, const uint4& v) {
        if (c - tid >= nvec) return;  // uniform: this pass of the workgroup lies past the row
        unsigned keys[8];
        unsigned my_eq = 0;
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        if (c < nvec) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                keys[2 * j] = bf16_key(w[j] & 0xffffu);
                keys[2 * j + 1] = bf16_key(w[j] >> 16);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (keys[j] > thr) {
                    const unsigned slot = atomicAdd(&sh_u[2], 1u);
                    const uint32_t wj = w[j >> 1];
                    c_val[slot] = (j & 1) ? bf16_hi(wj) : bf16_lo(wj);
                    c_idx[slot] = c * 8 + j;
                }
                my_eq += (keys[j] == thr);
            }
        }
        if (eq_base < need_eq) {  // uniform: eq_base is the same in every thread
            unsigned total;
            const unsigned incl = block_scan_incl(my_eq, ws_u, total);
            unsigned rank = eq_base + incl - my_eq;
            if (c < nvec && my_eq) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (keys[j] == thr) {
                        if (rank < need_eq) {
                            const uint32_t wj = w[j >> 1];
                            c_val[n_gt + rank] = (j & 1) ? bf16_hi(wj) : bf16_lo(wj);
                            c_idx[n_gt + rank] = c * 8 + j;
                        }
                        ++rank;
                    }
                }
            }
            eq_base += total;
        }
    });
    // pad to a power of two for the sort
    int p2 = 1;
    while (p2 < k) p2 <<= 1;
    __syncthreads();
    if (tid >= k && tid < p2) {
        c_val[tid] = -INFINITY;
        c_idx[tid] = 0x7fffffff;
    }
    // ---- bitonic sort, descending by value, ascending index among equals
    for (int size = 2; size <= p2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            const int i = tid, j = tid ^ stride;
            if (i < p2 && j > i) {
                const float vi = c_val[i], vj = c_val[j];
                const int ii = c_idx[i], ij = c_idx[j];
                const bool i_first = (vi > vj) || (vi == vj && ii < ij);  // i belongs before j
                const bool want_first = (i & size) == 0;
                if (i_first != want_first) {
                    c_val[i] = vj;
                    c_val[j] = vi;
                    c_idx[i] = ij;
                    c_idx[j] = ii;
                }
            }
        }
    }
    __syncthreads();

    // ---- softmax over the k candidates, top-p prefix, inverse-CDF draw
    const float temp = temps[blockIdx.x];
    if (!(temp > 0.f)) {  // uniform: greedy limit
        if (tid == 0) ids[blockIdx.x] = c_idx[0];
        return;
    }
    const float smax = c_val[0] / temp;
    const float e = tid < k ? expf(c_val[tid] / temp - smax) : 0.f;
    float total;
    const float cum = block_scan_incl(e, ws_f, total);
    const float cump = cum / total;
    const float top_p = top_ps[blockIdx.x];
    const unsigned valid = (tid < k) && (cump < top_p || tid == 0);
    unsigned m;
    block_scan_incl(valid, ws_u, m);               // the valid set is a prefix: m = its length
    if (tid == (int)m - 1) ws_f[0] = cump;           // mass kept
    __syncthreads();
    const float target = uniforms[blockIdx.x] * ws_f[0];
    __syncthreads();
    const unsigned below = (tid < (int)m) && (cump <= target);
    unsigned nb;
    block_scan_incl(below, ws_u, nb);
    if (tid == 0) ids[blockIdx.x] = c_idx[min(nb, m - 1)];
}

}  // namespace mv

extern "C" int mv_sample_topk_topp(const void* logits, int64_t ld, int rows, int n,
                                   const float* temperatures, const float* top_ps, int top_k,
                                   const float* uniforms, int64_t* ids, mv_stream_t stream) {
    return mv_sample_or_greedy(logits, ld, rows, n, temperatures, top_ps, top_k, uniforms, nullptr, ids,
                               stream);
}

extern "C" int mv_sample_or_greedy(const void* logits, int64_t ld, int rows, int n,
                                   const float* temperatures, const float* top_ps, int top_k,
                                   const float* uniforms, const int64_t* greedy_ids, int64_t* ids,
                                   mv_stream_t stream) {
    if (rows == 0 && n > 0) return MV_OK;
    if (!logits || !ids || !temperatures || !top_ps || !uniforms || rows < 0 || n <= 0 || ld < n ||
        top_k <= 0)
        return MV_ERR_INVALID_ARG;
    if (!mv::aligned16(logits) || (ld % 8) != 0) return MV_ERR_INVALID_ARG;
    if ((n % 8) != 0) return MV_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(mv::sample_kernel, dim3(rows), dim3(mv::kSmThreads), 0, (hipStream_t)stream,
                       (const uint16_t*)logits, ld, n, temperatures, top_ps, top_k, uniforms, greedy_ids, ids);
    return mv::check_launch();
}

extern "C" size_t mv_argmax_workspace_bytes(void) { return mv::kAmWorkspaceBytes; }

extern "C" int mv_argmax(const void* logits, int64_t ld, int rows, int n, int64_t* ids, void* workspace,
                         size_t workspace_bytes, mv_stream_t stream) {
    if (rows == 0 && n > 0) return MV_OK;
    if (!logits || !ids || rows < 0 || n <= 0 || ld < n) return MV_ERR_INVALID_ARG;
    if (!mv::aligned16(logits) || (ld % 8) != 0) return MV_ERR_INVALID_ARG;
    // few long rows: several workgroups per row, meeting in the caller's workspace (without one,
    // or with too small a one, every row gets a single workgroup: same result, slower)
    const int nvec = n >> 3;
    if (workspace && workspace_bytes >= mv::kAmWorkspaceBytes && mv::aligned16(workspace) &&
        rows <= mv::kAmMaxRows && rows * 8 <= mv::num_cus() && nvec >= 4 * mv::kAmPartThreads) {
        const mv::AmScratch sc = mv::am_scratch(workspace);
        int parts = mv::num_cus() / (2 * rows);
        if (parts > mv::kAmMaxParts) parts = mv::kAmMaxParts;
        const int most = nvec / mv::kAmPartThreads;  // at least one 16-byte load per thread
        if (parts > most) parts = most;
        if (parts >= 2) {
            const int vec_per_part = (nvec + parts - 1) / parts;
            hipLaunchKernelGGL(mv::argmax_parts_kernel, dim3(rows * parts), dim3(mv::kAmPartThreads), 0,
                               (hipStream_t)stream, (const uint16_t*)logits, ld, n, parts, vec_per_part,
                               sc, ids);
            return mv::check_launch();
        }
    }
    hipLaunchKernelGGL(mv::argmax_kernel, dim3(rows), dim3(mv::kAmThreads), 0, (hipStream_t)stream,
                       (const uint16_t*)logits, ld, n, ids);
    return mv::check_launch();
}
