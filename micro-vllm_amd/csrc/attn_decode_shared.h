// Pieces of the single-token paged attention that more than one kernel is built from: the argument
// block, the row reduction, the NeoX rotation and everything that happens after a workgroup's
// softmax states are in LDS (state merge, split hand-over, merge of all splits by the last
// arriver).  Used by attn_decode.hip (the stand-alone kernels) and qkv_attn.hip (QKV projection and
// attention in one launch).
#pragma once
#include "common.h"

#ifndef MV_ASTAMP
#define MV_ASTAMP(slot) do {} while (0)
#endif

namespace mv {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;

constexpr float kNegBig = -1.0e30f;
// Arrival counters sit 256 bytes apart: tickets of different (sequence, kv head) pairs are then
// served by different L2 channels instead of queueing on one cache line.  They live in a
// fixed-size header at the start of the caller's workspace (so two calls with two workspaces
// never share a counter): zero when the caller hands the workspace over for the first time,
// returned to zero by the merging workgroup of every launch.
constexpr int kArriveStride = 64;
constexpr int kMaxArrive = 4096;   // (sequence, kv head) pairs with a counter; more: second-launch merge
constexpr size_t kArriveBytes = (size_t)kMaxArrive * kArriveStride * sizeof(unsigned int);  // 1 MiB

struct AttnArgs {
    const uint16_t* q;
    int64_t q_stride_b;
    const uint16_t* k_new;
    const uint16_t* v_new;
    int64_t kv_stride_b;
    uint16_t* k_cache;
    uint16_t* v_cache;
    const uint16_t* rot_cos;
    const uint16_t* rot_sin;
    int rot_max_pos;
    const int32_t* seqlens;
    const int32_t* block_table;
    int bt_stride;
    uint16_t* out;
    float* part_lse;  // [bs*Hq*nsplit]
    float* part_o;    // [bs*Hq*nsplit*D]
    unsigned int* arrive;  // [bs*Hkv] arrival counters in the workspace header (zero between launches);
                           // null = merge in a second kernel
    int bs, hq, hkv;
    int page, page_shift, num_pages;
    float scale_log2;
    int nsplit;
    // q, k_new, v_new as the QKV projection left them when it was cut along k: fp32 slabs
    // [slab_ks][bs][slab_n] whose sum, in slab order, is a row of q | k | v (null: the bf16 rows above)
    const float* slabs;
    const uint16_t* slab_bias;   // [slab_n] or null
    int slab_ks, slab_n;
};

// sum over the LPR adjacent lanes that hold one row; every lane gets the total
template <int LPR>
__device__ __forceinline__ float row_sum(float v) {
    // xor-1, xor-2 inside quads, then mirror within 8, then mirror within 16
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    if (LPR == 16)
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    return v;
}

// NeoX rotation of this lane's 8 elements (x) given the 8 elements half a head away
// (partner) and cos/sin of the pair index; `first_half` says which side x is on.
__device__ __forceinline__ void rope8(const uint4& x, const uint4& partner, const uint4& c,
                                      const uint4& s, bool first_half, uint32_t (&out)[4]) {
    const uint32_t xs[4] = {x.x, x.y, x.z, x.w}, ps[4] = {partner.x, partner.y, partner.z, partner.w};
    const uint32_t cs[4] = {c.x, c.y, c.z, c.w}, ss[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float lo, hi;
        if (first_half) {
            lo = bf16_lo(xs[j]) * bf16_lo(cs[j]) - bf16_lo(ps[j]) * bf16_lo(ss[j]);
            hi = bf16_hi(xs[j]) * bf16_hi(cs[j]) - bf16_hi(ps[j]) * bf16_hi(ss[j]);
        } else {
            lo = bf16_lo(xs[j]) * bf16_lo(cs[j]) + bf16_lo(ps[j]) * bf16_lo(ss[j]);
            hi = bf16_hi(xs[j]) * bf16_hi(cs[j]) + bf16_hi(ps[j]) * bf16_hi(ss[j]);
        }
        out[j] = pack_bf16x2(lo, hi);
    }
}

// Everything after a workgroup's NS softmax states (m, l, acc[D] per query head) are in LDS as
// st_m [NS][G] | st_l [NS][G] | st_acc [NS][G][D]: merge them, hand the split's partial over (or
// write the output when there is one split) and, as the last-arriving split, merge all splits.
template <int D, int G, int NS, int kAW>
__device__ __forceinline__ void finish_states(const AttnArgs& a, float* lds, int b, int kvh, int bh,
                                              int split, int wid, int lane) {
    constexpr int kAT = kAW * 64;
    float* st_m = lds;                 // [NS][G]
    float* st_l = lds + NS * G;        // [NS][G]
    float* st_acc = lds + 2 * NS * G;  // [NS][G][D]
    __syncthreads();
    // One thread per (head, 4 consecutive d): it works out the weights 2^(m_s - max) of its head's
    // NS states itself (every thread of a head repeats that, which is cheaper than a phase and a
    // barrier of its own), then adds the states in order.
    if (threadIdx.x < G * D / 4) {
        const int g = threadIdx.x / (D / 4), d4 = (threadIdx.x - g * (D / 4)) * 4;
        float mx = kNegBig;
#pragma unroll
        for (int s = 0; s < NS; ++s) mx = fmaxf(mx, st_m[s * G + g]);
        float lt = 0.f;
        f32x4_t at = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const float w = __builtin_amdgcn_exp2f(st_m[s * G + g] - mx);
            lt = __builtin_fmaf(st_l[s * G + g], w, lt);
            const f32x4_t v = *reinterpret_cast<const f32x4_t*>(st_acc + (int64_t)(s * G + g) * D + d4);
            at[0] = __builtin_fmaf(v[0], w, at[0]);
            at[1] = __builtin_fmaf(v[1], w, at[1]);
            at[2] = __builtin_fmaf(v[2], w, at[2]);
            at[3] = __builtin_fmaf(v[3], w, at[3]);
        }
        const int hq = kvh * G + g;
        f32x4_t val = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        if (lt > 0.f) {
            val[0] = at[0] / lt;
            val[1] = at[1] / lt;
            val[2] = at[2] / lt;
            val[3] = at[3] / lt;
        }
        if (a.nsplit == 1) {
            uint2 o2;
            o2.x = pack_bf16x2(val[0], val[1]);
            o2.y = pack_bf16x2(val[2], val[3]);
            *reinterpret_cast<uint2*>(a.out + ((int64_t)b * a.hq + hq) * D + d4) = o2;
        } else {
            // write-through (sc1) stores: visible to the merging workgroup without a release fence
            const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(
                a.part_o + ((int64_t)b * a.hq + kvh * G) * a.nsplit * D, 0, G * a.nsplit * D * 4, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, val), prs,
                                                   ((g * a.nsplit + split) * D + d4) * 4, 0, 16 /*sc1*/);
            if (d4 == 0) {
                const int64_t pi = ((int64_t)b * a.hq + hq) * a.nsplit + split;
                __hip_atomic_store(reinterpret_cast<unsigned int*>(a.part_lse + pi),
                                   __float_as_uint(lt > 0.f ? mx + __builtin_amdgcn_logf(lt) : kNegBig),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    MV_ASTAMP(3);
    if (a.nsplit == 1 || a.arrive == nullptr) return;

    // ---- last-arriving split of this (b, kv head) merges all splits ---------------------
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its stores
    __syncthreads();
    MV_ASTAMP(4);
    int* flag = reinterpret_cast<int*>(lds);          // the state arrays are dead past the barrier
    if (threadIdx.x == 0) {
        // One counter per pair.  (Two levels of counters, groups of eight splits first, measured
        // slower: 12.3 vs 11.9 us at 32 splits; a ticket costs its round trip, not its queue.)
        unsigned int* cnt = a.arrive + (int64_t)bh * kArriveStride;
        const bool last = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ==
                          (unsigned)a.nsplit - 1u;
        if (last)  // back to zero for the next launch: nobody else touches it in this one
            __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag = last;
    }
    __syncthreads();
    MV_ASTAMP(5);
    if (!*flag) return;
    __syncthreads();  // everyone has read the flag before LDS is reused
    // Every load of handed-off data below is an sc1 (L1-bypassing) load.  Each thread owns one
    // (head, 4 consecutive d) item and every PARTS-th split of it; its 16-byte loads and the
    // lse loads are all issued before anything waits, so the merge costs about one memory
    // round trip.  Partial sums meet in LDS in a fixed order (results are deterministic).
    // (With 3, 5, 6 or 7 query heads per kv head the items do not divide the threads: the threads
    // past the last whole part sit the merge out.)
    constexpr int ITEMS = G * D / 4;
    constexpr int PARTS = kAT / ITEMS;
    static_assert(PARTS >= 1, "at least one thread per (head, 4 consecutive d)");
    constexpr int kMU = PARTS >= 8 ? 4 : 16;  // loads in flight per thread
    const int ns = a.nsplit;
    float* wts = lds;                 // [G][nsplit] weights 2^(lse - max) / sum
    float* psum = lds + G * 128;      // [PARTS][ITEMS][4] (nsplit <= 128)
    const int item = threadIdx.x % ITEMS, part = threadIdx.x / ITEMS;
    const bool active = part < PARTS;
    const int mg = item / (D / 4), md4 = (item - mg * (D / 4)) * 4;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        a.part_o + ((int64_t)b * a.hq + kvh * G) * ns * D, 0, G * ns * D * 4, 0x00020000);
    auto load_batch = [&](u32x4_t(&v)[kMU], int s0) {
#pragma unroll
        for (int j = 0; j < kMU; ++j) {
            const int sp = min(s0 + j * PARTS, ns - 1);
            v[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, ((mg * ns + sp) * D + md4) * 4, 0, 16 /*sc1*/);
        }
    };
    // One wave per head turns its log-sum-exps into normalised weights, one split per lane (two
    // when there are more than 64 splits); the partial loads are in flight meanwhile.
    float lse0 = kNegBig, lse1 = kNegBig;
    const int wg0 = wid;  // first head of this wave
    if (wg0 < G) {
        const int64_t pi = ((int64_t)b * a.hq + kvh * G + wg0) * ns;
        if (lane < ns)
            lse0 = __uint_as_float(__hip_atomic_load(reinterpret_cast<unsigned int*>(a.part_lse + pi + lane),
                                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (lane + 64 < ns)
            lse1 = __uint_as_float(__hip_atomic_load(reinterpret_cast<unsigned int*>(a.part_lse + pi + lane + 64),
                                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    u32x4_t v0[kMU];
    if (active) load_batch(v0, part);
    for (int g = wg0; g < G; g += kAW) {
        if (g != wg0) {  // only when there are more heads than waves
            const int64_t pi = ((int64_t)b * a.hq + kvh * G + g) * ns;
            lse0 = lane < ns ? __uint_as_float(__hip_atomic_load(
                                   reinterpret_cast<unsigned int*>(a.part_lse + pi + lane), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT))
                             : kNegBig;
            lse1 = lane + 64 < ns ? __uint_as_float(__hip_atomic_load(
                                        reinterpret_cast<unsigned int*>(a.part_lse + pi + lane + 64),
                                        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                                  : kNegBig;
        }
        const float mx = wave_max(fmaxf(lse0, lse1));
        const float w0 = lane < ns ? __builtin_amdgcn_exp2f(lse0 - mx) : 0.f;
        const float w1 = lane + 64 < ns ? __builtin_amdgcn_exp2f(lse1 - mx) : 0.f;
        const float sum = wave_sum(w0 + w1);
        const float inv = sum > 0.f ? 1.0f / sum : 0.f;
        if (lane < ns) wts[g * ns + lane] = w0 * inv;
        if (lane + 64 < ns) wts[g * ns + lane + 64] = w1 * inv;
    }
    MV_ASTAMP(9);
    __syncthreads();
    MV_ASTAMP(10);
    if (active) {
        const float* w = wts + mg * ns;
        f32x4_t acc_o = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        auto consume = [&](const u32x4_t(&v)[kMU], int s0) {
#pragma unroll
            for (int j = 0; j < kMU; ++j) {
                const int sp = s0 + j * PARTS;
                const float wj = sp < ns ? w[sp] : 0.f;
                acc_o[0] = __builtin_fmaf(wj, __uint_as_float(v[j].x), acc_o[0]);
                acc_o[1] = __builtin_fmaf(wj, __uint_as_float(v[j].y), acc_o[1]);
                acc_o[2] = __builtin_fmaf(wj, __uint_as_float(v[j].z), acc_o[2]);
                acc_o[3] = __builtin_fmaf(wj, __uint_as_float(v[j].w), acc_o[3]);
            }
        };
        consume(v0, part);
        for (int s0 = part + PARTS * kMU; s0 < ns; s0 += PARTS * kMU) {
            u32x4_t v[kMU];
            load_batch(v, s0);
            consume(v, s0);
        }
        *reinterpret_cast<f32x4_t*>(psum + (part * ITEMS + item) * 4) = acc_o;
    }
    __syncthreads();
    MV_ASTAMP(6);
    if (threadIdx.x < ITEMS) {
        f32x4_t t = *reinterpret_cast<const f32x4_t*>(psum + item * 4);
#pragma unroll
        for (int p = 1; p < PARTS; ++p) t += *reinterpret_cast<const f32x4_t*>(psum + (p * ITEMS + item) * 4);
        uint2 o2;
        o2.x = pack_bf16x2(t[0], t[1]);
        o2.y = pack_bf16x2(t[2], t[3]);
        *reinterpret_cast<uint2*>(a.out + ((int64_t)b * a.hq + kvh * G + mg) * D + md4) = o2;
    }
    MV_ASTAMP(7);
}

}  // namespace mv
