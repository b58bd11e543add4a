// Prefill attention over the paged KV cache: a rotate-and-store kernel that appends the
// chunk's keys and values to their pages, and a causal flash-attention kernel on the matrix
// cores that reads keys and values back through the block table.
//
// Attention kernel.  Workgroup = (sequence, kv head, block of 128/G query positions): the G
// query heads that share the kv head are stacked into 128 rows, so every K/V tile staged in
// LDS serves all of them.  Wave w owns rows 32w..32w+31 as two 16-row tiles.
//   S^T = K . Q^T   v_mfma_f32_16x16x32_bf16, A = K rows from LDS (ds_read_b128), B = Q from
//                   registers; a lane holds 4 keys x 1 query per tile, so a query's running
//                   max and sum live in one lane column (reduced over lanes l, l^16, l^32).
//   O^T = V^T . P^T the two S^T tiles of a 32-key step, rounded to bf16, ARE the B operand
//                   (keys in the order 4q..4q+3, 16+4q..16+4q+3 for lane group q); V^T comes
//                   from the row-major V tile with the transposing LDS read
//                   (ds_read_b64_tr_b16), which delivers those same keys.
// Softmax in fp32 with exp2 and a folded scale; the normaliser sums the unrounded P; P is
// rounded to bf16 as the matrix operand (as in the flash-attn package this call stands for).
// K/V tiles of 64 keys are staged twice buffered; key positions past the last valid key are
// clamped on load (finite data, masked to zero weight).
#include "common.h"

#include <stdlib.h>

#include <type_traits>

namespace mv {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;

constexpr float kPfNegBig = -1.0e30f;

// Maximum over lanes l, l ^ 16, l ^ 32, l ^ 48 (the four lanes that hold one query's scores), in the
// vector unit: v_permlane16_swap exchanges the odd 16-lane rows of one register with the even rows of
// the other, v_permlane32_swap the upper half of one with the lower half of the other - given the
// same value twice, the two results hold a lane's own row (half) and its partner's.  The shuffle
// intrinsic goes through the LDS crossbar instead (ds_bpermute_b32 and a wait for every LDS operation
// in flight), twice per query tile and key tile: 213 -> 198 us per layer at 4096 tokens.
__device__ __forceinline__ float max_over_lane_quarters(float v) {
    const unsigned u = __float_as_uint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float m = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const unsigned w = __float_as_uint(m);
    const auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
constexpr int kPfWideFrom = 512;   // chunk length from which 8-wave workgroups are taken
// waves per workgroup of the attention kernel forced by MVLLM_PREFILL_ATTN_WAVES (4 or 8; read once)
static const int g_pf_waves = [] {
    const char* e = getenv("MVLLM_PREFILL_ATTN_WAVES");
    const int v = e && *e ? atoi(e) : 0;
    return v == 4 || v == 8 ? v : 0;
}();
// MVLLM_PREFILL_ATTN_PIPE = 0 keeps the eight-wave workgroups on the plain (un-pipelined) kernel
static const bool g_pf_pipe = [] {
    const char* e = getenv("MVLLM_PREFILL_ATTN_PIPE");
    return !(e && *e == '0');
}();
// MVLLM_PREFILL_ATTN_RAISE = x (powers of two, 0 .. 16; default 0): the pipelined kernel raises a row's
// running maximum only when some row of the wave outgrew its own by more than 2^x.  0 is the exact
// running maximum of the flash-attn package (the largest weight of a row is then exactly 1 when it
// is rounded to bf16).  8 skips nearly every rescale of the output accumulators - 4096 tokens of the
// 8B shape 152 us per layer against 174 - at 12 % more mean error against an fp32 softmax (4.4e-4
// against 3.9e-4 on outputs of mean magnitude 0.22): a lagging maximum leaves the largest weight
// above 1 and inexact in bf16.
static const float g_pf_raise = [] {
    const char* e = getenv("MVLLM_PREFILL_ATTN_RAISE");
    const float v = e && *e ? (float)atof(e) : 0.0f;
    return v >= 0.0f && v <= 16.0f ? v : 0.0f;
}();

static int pf_group_slots(int g) {   // query heads per kv head, rounded up to a power of two
    int gp = 1;
    while (gp < g) gp <<= 1;
    return gp;
}

// 256-row (8-wave) workgroups from 512 tokens a chunk: the pipelined kernel they run is ahead of the
// 128-row kernel at every shape measured from there (8B head shape, tools/sweep_prefill_attn_forms.sh:
// 8 x 512 tokens 36 against 44 us, 4 x 1024 51 against 68, 2048 67 against 76 - half the CUs idle - ,
// 3072 103 against 122, 4096 159 against 220) and level with it below (256 tokens: 14.6 against 13.2).
static bool pf_eight_waves(int bs, int seq, int hkv, int g) {
    if (g_pf_waves) return g_pf_waves == 8;
    return seq >= kPfWideFrom;
}

// ------------------------------------------------------------------ rotate + store
struct RopeStoreArgs {
    uint16_t* qkv;          // [B*S][row_stride]: q heads | k heads | v heads
    int64_t row_stride;
    const uint16_t* cos;    // [max_pos][D/2] or null
    const uint16_t* sin;
    int rot_max_pos;
    const int32_t* start;   // [B] tokens already cached
    const int32_t* block_table;
    int bt_stride;
    uint16_t* k_cache;
    uint16_t* v_cache;
    int B, S, hq, hkv, d;
    int page, page_shift, num_pages;
};

// bf16 tensor arithmetic of core/layer/rope.py:29-33: every product and the sum round to bf16
__device__ __forceinline__ void rope_pair(uint32_t x1, uint32_t x2, uint32_t c, uint32_t s,
                                          uint32_t& o1, uint32_t& o2) {
    const float a_lo = round_bf16(bf16_lo(x1) * bf16_lo(c)), a_hi = round_bf16(bf16_hi(x1) * bf16_hi(c));
    const float b_lo = round_bf16(bf16_lo(x2) * bf16_lo(s)), b_hi = round_bf16(bf16_hi(x2) * bf16_hi(s));
    const float c_lo = round_bf16(bf16_lo(x2) * bf16_lo(c)), c_hi = round_bf16(bf16_hi(x2) * bf16_hi(c));
    const float d_lo = round_bf16(bf16_lo(x1) * bf16_lo(s)), d_hi = round_bf16(bf16_hi(x1) * bf16_hi(s));
    o1 = pack_bf16x2(a_lo - b_lo, a_hi - b_hi);
    o2 = pack_bf16x2(c_lo + d_lo, c_hi + d_hi);
}

// one thread = one 16-byte chunk pair (first half / second half) of one head of one token
__global__ __launch_bounds__(256) void rope_store_kernel(RopeStoreArgs a) {
    const int cph = a.d >> 4;                       // chunk pairs per head
    const int heads = a.hq + 2 * a.hkv;
    const int64_t total = (int64_t)a.B * a.S * heads * cph;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * 256) {
        const int c = (int)(idx % cph);
        const int h = (int)((idx / cph) % heads);
        const int64_t tok = idx / ((int64_t)cph * heads);
        const int b = (int)(tok / a.S), i = (int)(tok - (int64_t)b * a.S);
        const int pos = a.start[b] + i;
        uint16_t* base = a.qkv + tok * a.row_stride + (int64_t)h * a.d;
        uint4 v1 = ld16(base + c * 8), v2 = ld16(base + (a.d >> 1) + c * 8);
        if (h < a.hq + a.hkv && a.cos) {
            const int rp = min(max(pos, 0), a.rot_max_pos - 1);
            const uint4 cv = ld16(a.cos + (int64_t)rp * (a.d >> 1) + c * 8);
            const uint4 sv = ld16(a.sin + (int64_t)rp * (a.d >> 1) + c * 8);
            rope_pair(v1.x, v2.x, cv.x, sv.x, v1.x, v2.x);
            rope_pair(v1.y, v2.y, cv.y, sv.y, v1.y, v2.y);
            rope_pair(v1.z, v2.z, cv.z, sv.z, v1.z, v2.z);
            rope_pair(v1.w, v2.w, cv.w, sv.w, v1.w, v2.w);
            st16(base + c * 8, v1);
            st16(base + (a.d >> 1) + c * 8, v2);
        }
        if (h >= a.hq) {
            // a position before the table or past its last column belongs to no page of this
            // sequence: reading that entry would take the next row's page ids (or run off the
            // buffer) and the store would land in another sequence's page
            const int pcol = pos >> a.page_shift;
            if (pos < 0 || pcol >= a.bt_stride) continue;
            const int pg = a.block_table[(int64_t)b * a.bt_stride + pcol];
            if (pg < 0 || pg >= a.num_pages) continue;  // unallocated: nothing to write to
            const bool is_v = h >= a.hq + a.hkv;
            const int kvh = h - a.hq - (is_v ? a.hkv : 0);
            uint16_t* dst = (is_v ? a.v_cache : a.k_cache) +
                            (((int64_t)pg * a.page + (pos & (a.page - 1))) * a.hkv + kvh) * a.d;
            st16(dst + c * 8, v1);
            st16(dst + (a.d >> 1) + c * 8, v2);
        }
    }
}

// RoPE alone on [tokens][heads*d] rows (position = token index inside its sequence)
__global__ __launch_bounds__(256) void rope_rows_kernel(const uint16_t* x, int64_t x_stride, uint16_t* y,
                                                        int64_t y_stride, const uint16_t* cos,
                                                        const uint16_t* sin, int rot_max_pos,
                                                        int64_t tokens, int seq, int heads, int d) {
    const int cph = d >> 4;
    const int64_t total = tokens * heads * cph;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * 256) {
        const int c = (int)(idx % cph);
        const int h = (int)((idx / cph) % heads);
        const int64_t tok = idx / ((int64_t)cph * heads);
        const int pos = min((int)(tok % seq), rot_max_pos - 1);
        const uint16_t* src = x + tok * x_stride + (int64_t)h * d;
        uint16_t* dst = y + tok * y_stride + (int64_t)h * d;
        uint4 v1 = ld16(src + c * 8), v2 = ld16(src + (d >> 1) + c * 8);
        const uint4 cv = ld16(cos + (int64_t)pos * (d >> 1) + c * 8);
        const uint4 sv = ld16(sin + (int64_t)pos * (d >> 1) + c * 8);
        rope_pair(v1.x, v2.x, cv.x, sv.x, v1.x, v2.x);
        rope_pair(v1.y, v2.y, cv.y, sv.y, v1.y, v2.y);
        rope_pair(v1.z, v2.z, cv.z, sv.z, v1.z, v2.z);
        rope_pair(v1.w, v2.w, cv.w, sv.w, v1.w, v2.w);
        st16(dst + c * 8, v1);
        st16(dst + (d >> 1) + c * 8, v2);
    }
}

// ------------------------------------------------------------------ attention
struct PfArgs {
    const uint16_t* q;      // [B*S][q_stride] rows, heads contiguous
    int64_t q_stride;
    const uint16_t* k_cache;
    const uint16_t* v_cache;
    const int32_t* start;
    const int32_t* block_table;
    int bt_stride;
    uint16_t* out;          // [B*S][out_stride]
    int64_t out_stride;
    int B, S, hq, hkv, g;
    int qt, qt_shift;       // query positions per workgroup = (128 or 256) / (g rounded up to a power of two)
    int nqb;
    int page_shift, num_pages, page;
    float scale_log2;
    float raise;            // pipelined kernel: powers of two a row's maximum may lag (0 = never)
};

// NW waves per workgroup = 32 NW stacked query rows against every staged K/V tile: 4 (128 rows, two
// workgroups per CU) or 8 (256 rows, one per CU: each tile fetched and stored to LDS once for twice
// the matrix work - taken when that still leaves every CU a workgroup and a half).
template <int D, int NW>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void paged_attn_prefill_kernel(PfArgs a) {
    constexpr int NT = NW * 64;        // threads
    constexpr int KT = 64;             // keys per LDS tile
    constexpr int KS = D / 32;         // k-steps of K.Q^T
    constexpr int DT = D / 16;         // 16-wide slices of the head dimension
    constexpr int KROW = D * 2 + 16;   // LDS row strides (bytes): conflict-free row reads ...
    constexpr int VROW = D * 2 + 32;   // ... and conflict-free transposed reads
    constexpr int VPR = D / 8;         // 16-byte vectors per row
    constexpr int NV = KT * VPR / NT;  // vectors per thread per tile
    __shared__ __attribute__((aligned(16))) unsigned char kL[2][KT * KROW];
    __shared__ __attribute__((aligned(16))) unsigned char vL[2][KT * VROW];

    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, q = lane >> 4;

    // heaviest query blocks (most keys) first
    const int per = a.hkv * a.B;
    const int qb = a.nqb - 1 - (int)(blockIdx.x / per);
    const int rem = blockIdx.x % per;
    const int kvh = rem % a.hkv, b = rem / a.hkv;
    const int p0 = max(a.start[b], 0);   // a negative start is a contract breach: treated as 0
    const int i0 = qb * a.qt;
    const int last_key = p0 + min(i0 + a.qt, a.S) - 1;   // last key any row of this block sees
    const int ntiles = last_key / KT + 1;
    const int32_t* bt = a.block_table + (int64_t)b * a.bt_stride;

    // this lane's query rows (C-layout column r of each of the wave's two tiles)
    // (3, 5, 6 or 7 query heads per kv head: the stack has room for the next power of two; the rows
    // of the heads that do not exist repeat the last real head and are never stored)
    int limit[2], qi[2], head[2];
    bf16x8_t qf[2][KS];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int row = wid * 32 + mt * 16 + r;
        const int gi = row >> a.qt_shift, i = i0 + (row & (a.qt - 1));
        head[mt] = kvh * a.g + min(gi, a.g - 1);
        qi[mt] = gi < a.g ? i : a.S;   // a position past the chunk: not stored
        const int ic = min(i, a.S - 1);
        limit[mt] = p0 + ic;
        const uint16_t* qp = a.q + ((int64_t)b * a.S + ic) * a.q_stride + (int64_t)head[mt] * D + q * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            qf[mt][ks] = __builtin_bit_cast(bf16x8_t, ld16(qp + ks * 32));
    }

    // the smallest last-key over the wave's 32 rows (its first query position; rows past S repeat S - 1)
    int wave_min_limit = min(limit[0], limit[1]);
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) wave_min_limit = min(wave_min_limit, __shfl_xor(wave_min_limit, o, 64));
    wave_min_limit = __builtin_amdgcn_readfirstlane(wave_min_limit);

    // K/V staging: thread moves vectors t + 256 j of the 64 x VPR tile
    auto tile_page = [&](int t) {
        // column clamped into the table: keys past its last column have no page (a contract breach:
        // such keys read page-table garbage before; now they read the last column's page)
        const int pg = bt[min((t * KT) >> a.page_shift, a.bt_stride - 1)];
        return (pg >= 0 && pg < a.num_pages) ? pg : 0;   // unallocated entry: stay inside the cache
    };
    auto g_load = [&](uint4(&kr)[NV], uint4(&vr)[NV], int t, int pg) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int v = threadIdx.x + NT * j, row = v / VPR, ch = v % VPR;
            const int kp = min(t * KT + row, last_key);
            const int64_t off = (((int64_t)pg * a.page + (kp & (a.page - 1))) * a.hkv + kvh) * D + ch * 8;
            kr[j] = ld16(a.k_cache + off);
            vr[j] = ld16(a.v_cache + off);
        }
    };
    auto l_store = [&](const uint4(&kr)[NV], const uint4(&vr)[NV], int buf) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int v = threadIdx.x + NT * j, row = v / VPR, ch = v % VPR;
            *reinterpret_cast<uint4*>(&kL[buf][row * KROW + ch * 16]) = kr[j];
            *reinterpret_cast<uint4*>(&vL[buf][row * VROW + ch * 16]) = vr[j];
        }
    };

    f32x4_t oacc[2][DT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) oacc[mt][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    float mrun[2] = {kPfNegBig, kPfNegBig}, lrun[2] = {0.f, 0.f};

    uint4 kr[NV], vr[NV];
    int pg_next = tile_page(0);
    g_load(kr, vr, 0, pg_next);
    pg_next = ntiles > 1 ? tile_page(1) : 0;
    l_store(kr, vr, 0);
    __syncthreads();

    // transposed-read lane address inside a 16-key x 16-column block pair
    const int li = lane & 15;
    const int tr_off = (4 * q + (li >> 2)) * VROW + (li & 3) * 8;

    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
#ifdef MV_EXP_NO_KVLOAD
        const bool more = false;   // timing experiment only (wrong results): every tile re-uses the first two
#else
        const bool more = t + 1 < ntiles;
#endif
        if (more) {
            g_load(kr, vr, t + 1, pg_next);
            if (t + 2 < ntiles) pg_next = tile_page(t + 2);
        }
        // The 64 keys of the tile in one pass: 32 MFMAs for S^T, one softmax update over 16 scores
        // per lane and row (one cross-lane maximum, one rescale check), 32 MFMAs for O^T.  (In 32-key
        // steps the serial parts of the update - lane exchange, exponentials, the running sums -
        // came round twice as often between MFMA bursts half as long.)
        const int kb = t * KT;
        f32x4_t sacc[2][4];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int x = 0; x < 4; ++x) sacc[mt][x] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int x = 0; x < 4; ++x) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8_t kf = __builtin_bit_cast(
                    bf16x8_t, *reinterpret_cast<const uint4*>(&kL[buf][(x * 16 + r) * KROW + ks * 64 + q * 16]));
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    sacc[mt][x] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[mt][ks], sacc[mt][x], 0, 0, 0);
            }
        }
        bf16x8_t pf[2][2];
        // tiles wholly at or below every row's last key (all but the one or two on the diagonal)
        // need no mask; keys past the last valid one were clamped on load and are masked here
        const bool all_valid = kb + KT - 1 <= wave_min_limit;   // uniform
#ifdef MV_EXP_NO_SOFTMAX
        // timing experiment only (wrong results): the scores go to the second product as they are
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                uint4 pk;
                pk.x = pack_bf16x2(sacc[mt][2 * h][0], sacc[mt][2 * h][1]);
                pk.y = pack_bf16x2(sacc[mt][2 * h][2], sacc[mt][2 * h][3]);
                pk.z = pack_bf16x2(sacc[mt][2 * h + 1][0], sacc[mt][2 * h + 1][1]);
                pk.w = pack_bf16x2(sacc[mt][2 * h + 1][2], sacc[mt][2 * h + 1][3]);
                pf[mt][h] = __builtin_bit_cast(bf16x8_t, pk);
            }
        if (false)
#endif
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            // raw scores; the scale (positive) is applied to the maximum and inside the exponent
            float s[16];
            if (all_valid) {
#pragma unroll
                for (int x = 0; x < 4; ++x)
#pragma unroll
                    for (int j = 0; j < 4; ++j) s[x * 4 + j] = sacc[mt][x][j];
            } else {
#pragma unroll
                for (int x = 0; x < 4; ++x)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int key = kb + x * 16 + q * 4 + j;
                        s[x * 4 + j] = key <= limit[mt] ? sacc[mt][x][j] : kPfNegBig;
                    }
            }
            float m4[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) m4[x] = fmaxf(fmaxf(s[4 * x], s[4 * x + 1]), fmaxf(s[4 * x + 2], s[4 * x + 3]));
            float mloc = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3])) * a.scale_log2;
            mloc = max_over_lane_quarters(mloc);
            // the running maximum moves in few tiles once the first are in: the rescale of the 32
            // output accumulators is skipped while it stands for every row of the wave (a factor
            // of exactly 1 otherwise: same result)
            if (__builtin_amdgcn_ballot_w64(mloc > mrun[mt]) != 0) {
                const float mnew = fmaxf(mrun[mt], mloc);
                const float alpha = __builtin_amdgcn_exp2f(mrun[mt] - mnew);
                mrun[mt] = mnew;
                lrun[mt] *= alpha;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) oacc[mt][dt] *= alpha;
            }
            const float mnew = mrun[mt];
            float p[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) p[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[e], a.scale_log2, -mnew));
            // four partial sums: no 16-long chain of dependent adds
            lrun[mt] += ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7])) +
                        (((p[8] + p[9]) + (p[10] + p[11])) + ((p[12] + p[13]) + (p[14] + p[15])));
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                uint4 pk;
                pk.x = pack_bf16x2(p[8 * h + 0], p[8 * h + 1]);
                pk.y = pack_bf16x2(p[8 * h + 2], p[8 * h + 3]);
                pk.z = pack_bf16x2(p[8 * h + 4], p[8 * h + 5]);
                pk.w = pack_bf16x2(p[8 * h + 6], p[8 * h + 7]);
                pf[mt][h] = __builtin_bit_cast(bf16x8_t, pk);
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const unsigned char* vb = &vL[buf][h * 32 * VROW + dt * 32 + tr_off];
                const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (s16x4_t __attribute__((address_space(3)))*)(vb));
                const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (s16x4_t __attribute__((address_space(3)))*)(vb + 16 * VROW));
                typedef __attribute__((ext_vector_type(8))) short s16x8;
                const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                const bf16x8_t vf = __builtin_bit_cast(bf16x8_t, both);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    oacc[mt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[mt][h], oacc[mt][dt], 0, 0, 0);
            }
        }
        if (more) l_store(kr, vr, buf ^ 1);
        __syncthreads();
    }

    // oacc[mt][dt][j] = O^T[d = 16 dt + 4q + j][query row r]
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        float l = lrun[mt];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const float inv = 1.0f / l;
        if (qi[mt] < a.S) {
            uint16_t* op = a.out + ((int64_t)b * a.S + qi[mt]) * a.out_stride + (int64_t)head[mt] * D + q * 4;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const f32x4_t o = oacc[mt][dt];
                uint2 w;
                w.x = pack_bf16x2(o[0] * inv, o[1] * inv);
                w.y = pack_bf16x2(o[2] * inv, o[3] * inv);
                *reinterpret_cast<uint2*>(op + dt * 16) = w;
            }
        }
    }
}

// ---- the pipelined form: 256 stacked query rows per workgroup, 32 per wave ---------------------
// Same workgroup shape and tiles as the eight-wave form above, three differences:
//   * both products run on v_mfma_f32_32x32x16_bf16.  S^T = K . Q^T leaves a lane (query l & 31,
//     half h = l >> 5) 16 scores per 32-key block, keys (r & 3) + 8 (r >> 2) + 4 h for register r:
//     a query's maximum is one exchange between lanes l and l + 32, and a 32-cycle MFMA holds the
//     SIMD's vector issue for 8 of its cycles where the 16x16x32 form holds it for 8 of 16 - the
//     softmax arithmetic (two thirds as many issue cycles as the MFMAs themselves) only fits beside
//     MFMAs of the wide form.  Registers 8 s .. 8 s + 7 of a block, rounded to bf16, are the B
//     operand of k-step s of O^T += V^T . P^T as they stand: contraction slot (h, j) is key
//     16 s + 4 h + j for j < 4 and 16 s + 8 + 4 h + (j - 4) above, and the transposing reads of V
//     deliver exactly those keys (two 4-key blocks 8 keys apart per lane group).
//   * software pipeline over 32-key blocks: while the MFMAs of S(u + 1) run, the vector unit turns
//     S(u) into P(u); while those of O += V(u)^T P(u) run, it takes the row maxima of S(u + 1).
//     The MFMAs, the LDS reads that feed them (issued a few MFMAs ahead) and the softmax pieces are
//     dealt out slot by slot between scheduling fences, so the order written here is the order
//     issued - and a score pair's three stages (scale and subtract, exponential, sum and round)
//     stand in three consecutive slots: a wave issues in order, and with the stages of one pair
//     back to back every slot waited out the latencies of its own chain (55 cycles per slot where
//     the MFMA takes 32).  One barrier per 64-key tile: tile step t reads K(t) (upper half),
//     K(t + 1) (lower half) and V(t), and fills the buffers of K(t + 2) and V(t + 1), which nothing
//     reads before that barrier - K therefore sits in a ring of three tiles, V in two.
//   * the rescale of the output accumulators (taken when some row's maximum rose - or, if the
//     caller allows the maximum to lag, rose by more than that) happens at the top of the next
//     block step, off the MFMA-paired stretches.
// Rows of V are padded to a stride of 16 banks mod 64 (the transposing reads of a 32-lane half
// touch four rows x two column groups: conflict-free), rows of K by 16 bytes as above.
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

// one instruction where fmaxf(fmaxf(a, b), c) is three (a canonicalising self-maximum per operand
// that came out of an MFMA, then two maxima)
__device__ __forceinline__ float max3_f32(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

#ifdef MV_PFX_STAMP
// Stamp build (tools/stamp_attn_prefill.py): shader-clock ticks at eight points of tile steps
// kStampFrom .. + 15, for every wave of workgroup 0 (the heaviest query block).  The stamps are kept in
// LDS and copied out at the end.
constexpr int kStampFrom = 24;
static __device__ unsigned long long g_pfx_stamp[8 * 16 * 8];
// per workgroup: start and end on the constant 100 MHz counter, and the hardware id words
static __device__ unsigned long long g_pfx_wg[1024 * 3];
#define MV_PSTAMP(slot)                                                                        \
    do {                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        if (blockIdx.x == 0 && lane == 0 && t >= kStampFrom && t < kStampFrom + 16)            \
            stampL[(wid * 16 + t - kStampFrom) * 8 + (slot)] = __builtin_readcyclecounter();   \
        __builtin_amdgcn_sched_barrier(0);                                                     \
    } while (0)
#else
#define MV_PSTAMP(slot)
#endif

#ifndef MV_PFX_PF
#define MV_PFX_PF 3
#endif
#ifndef MV_PFX_PFV
#define MV_PFX_PFV 7
#endif

template <int D>
__global__ __launch_bounds__(512, 1) void paged_attn_prefill_pipe_kernel(PfArgs a) {
    constexpr int NT = 512, KT = 64;
    constexpr int KS = D / 16;          // MFMAs of one 32-key block of S^T (k-steps of 16)
    constexpr int DB = D / 32;          // 32-row blocks of O^T
    constexpr int NP = 2 * DB;          // MFMAs of one 32-key block of O^T (two k-steps of 16 keys)
    constexpr int PF = MV_PFX_PF < KS ? MV_PFX_PF : KS - 1;     // K fragment reads run this many MFMAs ahead
    constexpr int PFV = MV_PFX_PFV < NP ? MV_PFX_PFV : NP - 1;  // V fragment reads likewise
    // A row's running maximum is raised only when some row of the wave outgrows its own by more
    // than this many powers of two (the caller's choice, 0 = whenever any row's maximum rose: the
    // exact running maximum).  Above 0 the weights stay below 2^kRaise instead of 1 and the 16 DB
    // multiplications of the output accumulators are skipped in all but the first blocks of a
    // sequence.
    const float kRaise = a.raise;
    constexpr int KROW = D * 2 + 16, VROW = D * 2 + 64;
    constexpr int VPR = D / 8, NV = KT * VPR / NT;
    static_assert(NV == 1 || NV == 2, "one or two vectors per thread and tile");
    __shared__ __attribute__((aligned(16))) unsigned char kL[3][KT * KROW];
    __shared__ __attribute__((aligned(16))) unsigned char vL[2][KT * VROW];
#ifdef MV_PFX_STAMP
    __shared__ unsigned long long stampL[8 * 16 * 8];
#endif

    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lq = lane & 31, hi = lane >> 5;
#ifdef MV_PFX_STAMP
    if (threadIdx.x == 0 && blockIdx.x < 1024) {
        g_pfx_wg[blockIdx.x * 3] = wall_clock64();
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_pfx_wg[blockIdx.x * 3 + 2] = ((unsigned long long)xcc << 32) | hwid;
    }
#endif

    const int per = a.hkv * a.B;
    const int qb = a.nqb - 1 - (int)(blockIdx.x / per);   // heaviest query blocks first
    const int rem = blockIdx.x % per;
    const int kvh = rem % a.hkv, b = rem / a.hkv;
    const int p0 = max(a.start[b], 0);
    const int i0 = qb * a.qt;
    const int last_key = p0 + min(i0 + a.qt, a.S) - 1;
    const int ntiles = last_key / KT + 1;
    // The block table does not change under the kernel: read through the constant address space its
    // entries are scalar loads whatever else the kernel stores (a vector load of a page id would sit
    // behind the tile requests in the in-order counter and wait for them).
    typedef const __attribute__((address_space(4))) int32_t* const_i32_ptr;
    const_i32_ptr bt = (const_i32_ptr)(uintptr_t)(a.block_table + (int64_t)b * a.bt_stride);

    // this lane's query row
    const int row = wid * 32 + lq;
    const int gi = row >> a.qt_shift, pos_i = i0 + (row & (a.qt - 1));
    const int head = kvh * a.g + min(gi, a.g - 1);
    const int qi = gi < a.g ? pos_i : a.S;   // a head that does not exist: never stored
    const int ic = min(pos_i, a.S - 1);
    const int limit = p0 + ic;
    bf16x8_t qf[KS];
    {
        const uint16_t* qp = a.q + ((int64_t)b * a.S + ic) * a.q_stride + (int64_t)head * D + hi * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = __builtin_bit_cast(bf16x8_t, ld16(qp + ks * 16));
    }
    int wave_min_limit = limit;
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) wave_min_limit = min(wave_min_limit, __shfl_xor(wave_min_limit, o, 64));
    wave_min_limit = __builtin_amdgcn_readfirstlane(wave_min_limit);
    const int n_full = (wave_min_limit + 1) / 32;   // leading 32-key blocks no row of this wave has to mask

    // ---- staging: thread moves vectors tid, tid + 512 of a 64 x VPR tile.  A tile's rows lie
    // hkv * D elements apart from a base that is uniform over the workgroup (page, first row in the
    // page, kv head): the per-lane part is 32 bits wide and almost constant - only the last tile
    // clamps rows past the last key (finite data in place of rows never written).
    // (Staging registers are native vectors: copies of HIP's uint4 struct between address spaces are
    // memcpy calls to the compiler, which then kept the staging area in memory - in LDS, next to the
    // tiles - and every request waited for its data on the spot.)
    struct Stage {
        u32x4_t r0, r1;
    };
    const int srow0 = threadIdx.x / VPR, srow1 = (threadIdx.x + NT * (NV - 1)) / VPR, sch16 = (threadIdx.x % VPR) * 16;
    const int row_bytes = a.hkv * D * 2;
    const unsigned page_bytes = (unsigned)a.page * (unsigned)row_bytes;   // (checked on the host: below 2^32)
    auto page_of = [&](int t) __attribute__((always_inline)) {   // (range-checked where it is used)
        return bt[min((min(t, ntiles - 1) * KT) >> a.page_shift, a.bt_stride - 1)];
    };
    // byte offset of tile t in either cache: from its page id where a page begins, 64 rows after the
    // tile before it otherwise (two scalar additions instead of a multiplication chain per step);
    // tiles past the last one repeat it (a harmless re-fetch that is never read)
    auto tile_offset = [&](int t, int pg_raw, uint64_t prev) __attribute__((always_inline)) -> uint64_t {
        if (t >= ntiles) return prev;
        if (t > 0 && ((t * KT) & (a.page - 1)) != 0) return prev + (uint64_t)(KT * row_bytes);
        const unsigned pg = (pg_raw >= 0 && pg_raw < a.num_pages) ? pg_raw : 0;   // unallocated entry: stay inside the cache
        return (uint64_t)pg * page_bytes + (unsigned)(kvh * D * 2);
    };
    auto g_load = [&](Stage& sr, const uint16_t* cache, uint64_t toff, int t) __attribute__((always_inline)) {
        const unsigned char* base = reinterpret_cast<const unsigned char*>(cache) + toff;
        const int last_rel = last_key - min(t, ntiles - 1) * KT;
        sr.r0 = *reinterpret_cast<const u32x4_t*>(base + (unsigned)(min(srow0, last_rel) * row_bytes + sch16));
        if constexpr (NV > 1)
            sr.r1 = *reinterpret_cast<const u32x4_t*>(base + (unsigned)(min(srow1, last_rel) * row_bytes + sch16));
    };
    auto k_store = [&](const Stage& sr, int buf, int which) __attribute__((always_inline)) {
        if (which == 0) *reinterpret_cast<u32x4_t*>(&kL[buf][srow0 * KROW + sch16]) = sr.r0;
        if (NV > 1 && which == 1) *reinterpret_cast<u32x4_t*>(&kL[buf][srow1 * KROW + sch16]) = sr.r1;
    };
    auto v_store = [&](const Stage& sr, int buf, int which) __attribute__((always_inline)) {
        if (which == 0) *reinterpret_cast<u32x4_t*>(&vL[buf][srow0 * VROW + sch16]) = sr.r0;
        if (NV > 1 && which == 1) *reinterpret_cast<u32x4_t*>(&vL[buf][srow1 * VROW + sch16]) = sr.r1;
    };

    // ---- fragment reads
    const int k_off = lq * KROW + hi * 16;
    auto k_frag = [&](const unsigned char* kblock, int ks) __attribute__((always_inline)) {   // kblock: first row of a 32-key block
        return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(kblock + ks * 32 + k_off));
    };
    const int li = lane & 15;
    const int tr_off = (4 * hi + (li >> 2)) * VROW + (16 * ((lane >> 4) & 1) + 4 * (li & 3)) * 2;
    auto v_frag = [&](const unsigned char* vblock, int j) __attribute__((always_inline)) {   // MFMA j of a block: k-step j / DB, rows 32 (j % DB) of O^T
        const unsigned char* vb = vblock + (j / DB) * 16 * VROW + (j % DB) * 64 + tr_off;
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(vb));
        const s16x4_t up = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (s16x4_t __attribute__((address_space(3)))*)(vb + 8 * VROW));
        const s16x8_t both = {lo[0], lo[1], lo[2], lo[3], up[0], up[1], up[2], up[3]};
        return __builtin_bit_cast(bf16x8_t, both);
    };

    f32x16_t oacc[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[db][r] = 0.f;
    float mrun = kPfNegBig, lrun0 = 0.f, lrun1 = 0.f, alpha = 1.f;
    bool rescale = false;
    uint32_t pw[8];   // P of the current block rounded to bf16, two scores per word

    // A score pair (registers 2 pr, 2 pr + 1 of the block) goes through three stages, one slot apart
    float fa[8][2], eb[8][2];
    auto stage_a = [&](const f32x16_t& sc, int pr) __attribute__((always_inline)) {
        fa[pr][0] = __builtin_fmaf(sc[2 * pr], a.scale_log2, -mrun);
        fa[pr][1] = __builtin_fmaf(sc[2 * pr + 1], a.scale_log2, -mrun);
    };
    auto stage_b = [&](int pr) __attribute__((always_inline)) {
        eb[pr][0] = __builtin_amdgcn_exp2f(fa[pr][0]);
        eb[pr][1] = __builtin_amdgcn_exp2f(fa[pr][1]);
    };
    auto stage_c = [&](int pr) __attribute__((always_inline)) {
        lrun0 += eb[pr][0];
        lrun1 += eb[pr][1];
        pw[pr] = pack_bf16x2(eb[pr][0], eb[pr][1]);
    };
    // the stages that stand at pipeline position q (0 .. KS + NP - 1 over the two stretches of a
    // block step): pair pr enters at kEnter[pr], pairs 0-3 early enough to be words of P when the
    // second stretch begins
    auto softmax_slot = [&](const f32x16_t& sc, int q) __attribute__((always_inline)) {
        constexpr int kEnter128[8] = {1, 2, 3, 4, 5, 6, 7, 8}, kEnter64[8] = {1, 1, 1, 1, 2, 2, 3, 3};
#pragma unroll
        for (int pr = 0; pr < 8; ++pr)
            if (q == (D == 128 ? kEnter128[pr] : kEnter64[pr]) + 2) stage_c(pr);
#pragma unroll
        for (int pr = 0; pr < 8; ++pr)
            if (q == (D == 128 ? kEnter128[pr] : kEnter64[pr]) + 1) stage_b(pr);
#pragma unroll
        for (int pr = 0; pr < 8; ++pr)
            if (q == (D == 128 ? kEnter128[pr] : kEnter64[pr])) stage_a(sc, pr);
    };
    auto p_frag = [&](int s) __attribute__((always_inline)) {   // B operand of k-step s (0, 1) of the block's share of O^T
        const uint4 w = make_uint4(pw[4 * s], pw[4 * s + 1], pw[4 * s + 2], pw[4 * s + 3]);
        return __builtin_bit_cast(bf16x8_t, w);
    };
    // Row maximum of a block in four pieces over two accumulators: registers 0-5, 6-9, 10-13, 14-15.
    // lim_rel = this row's last key relative to the block's first, less 4 hi; masked: keys past it
    // are set to the floor, in the scores too.
    float m2[2];
    auto max_piece = [&](f32x16_t& sn, int mi, int lim_rel, auto masked) __attribute__((always_inline)) {
        constexpr int first[5] = {0, 6, 10, 14, 16};
        if constexpr (decltype(masked)::value) {
#pragma unroll
            for (int r = first[mi]; r < first[mi + 1]; ++r)
                if ((r & 3) + 8 * (r >> 2) > lim_rel) sn[r] = kPfNegBig;
        }
        if (mi == 0) {
            m2[0] = max3_f32(sn[0], sn[1], sn[2]);
            m2[1] = max3_f32(sn[3], sn[4], sn[5]);
        } else if (mi == 1) {
            m2[0] = max3_f32(m2[0], sn[6], sn[7]);
            m2[1] = max3_f32(m2[1], sn[8], sn[9]);
        } else if (mi == 2) {
            m2[0] = max3_f32(m2[0], sn[10], sn[11]);
            m2[1] = max3_f32(m2[1], sn[12], sn[13]);
        } else {
            m2[0] = max3_f32(m2[0], sn[14], sn[15]);
        }
    };
    // after the four pieces: the row's maximum over both lane halves, the new running maximum and
    // the factor the accumulators owe it
    auto max_finish = [&]() __attribute__((always_inline)) {
        float mloc = fmaxf(m2[0], m2[1]) * a.scale_log2;
        const unsigned u = __float_as_uint(mloc);
        const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        mloc = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        rescale = __builtin_amdgcn_ballot_w64(mloc > mrun + kRaise) != 0;
        if (rescale) {
            const float mnew = fmaxf(mrun, mloc);
            alpha = __builtin_amdgcn_exp2f(mrun - mnew);
            mrun = mnew;
        }
    };
    auto apply_rescale = [&]() __attribute__((always_inline)) {
        if (rescale) {
            lrun0 *= alpha;
            lrun1 *= alpha;
            // (two accumulators per instruction: v_pk_mul_f32)
            typedef __attribute__((ext_vector_type(2))) float f32x2_t;
            const f32x2_t a2 = {alpha, alpha};
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2_t o2 = f32x2_t{oacc[db][r], oacc[db][r + 1]} * a2;
                    oacc[db][r] = o2[0];
                    oacc[db][r + 1] = o2[1];
                }
        }
    };

    // One block step: scores `sc` of block u (maximum pieces taken) -> P(u) -> O from the V rows at
    // `vblock`; scores of block u + 1 (K rows at `kblock`, first key `next_key0`) -> `sn`.  The LDS
    // reads of a stretch's first MFMAs are issued beside the last MFMAs of the stretch before it
    // (their latency would stand exposed at every stretch start): `kpre` holds the first PF K
    // fragments of this step, read during the step before, and is refilled from `kblock_after` (the
    // K rows of the step that follows) if `has_after`.  `tail(j)` is called in slot j of the
    // second stretch, `hook(q)` in every slot q = 0 .. KS + NP - 1 of the step (the tile step's
    // requests and LDS stores ride there).
    bf16x8_t kpre[PF];
    auto block_step = [&](const f32x16_t& sc, f32x16_t& sn, const unsigned char* kblock, const unsigned char* vblock,
                          const unsigned char* kblock_after, int next_key0, auto has_next, auto has_after,
                          auto mask_next, int t, int slot0, auto&& hook) __attribute__((always_inline)) {
        constexpr bool kNext = decltype(has_next)::value, kAfter = decltype(has_after)::value;
        max_finish();
        apply_rescale();
        MV_PSTAMP(slot0);
        // stretch 1: S(u + 1) on the matrix pipe, the exponentials of S(u) beside it.  (The end of the
        // row maximum and the rescale placed beside the first MFMA instead of in front of the
        // stretch measured 2 % slower; cut into four branch-free parts over the last slots of the
        // second stretch, no faster either.)
        bf16x8_t kf[KS], vf[NP];
#pragma unroll
        for (int i = 0; i < KS; ++i) {
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (kNext) {
                if (i < PF) kf[i] = kpre[i];
                if (i + PF < KS) kf[i + PF] = k_frag(kblock, i + PF);
                const f32x16_t zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                sn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[i], qf[i], i == 0 ? zero : sn, 0, 0, 0);
            }
            if (i + PFV >= KS) vf[i + PFV - KS] = v_frag(vblock, i + PFV - KS);
            softmax_slot(sc, i);
            hook(i);
        }
        MV_PSTAMP(slot0 + 1);
        // stretch 2: O += V(u)^T P(u); the tail of the softmax pipeline in its first slots (the
        // second k-step needs all of P), the maximum pieces of S(u + 1) in its second half
        const int lim_rel = limit - next_key0 - 4 * hi;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            __builtin_amdgcn_sched_barrier(0);
            if (j + PFV < NP) vf[j + PFV] = v_frag(vblock, j + PFV);
            if constexpr (kAfter) {
                if (j + PF >= NP) kpre[j + PF - NP] = k_frag(kblock_after, j + PF - NP);
            }
            oacc[j % DB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[j], p_frag(j / DB), oacc[j % DB], 0, 0, 0);
            softmax_slot(sc, KS + j);
            if constexpr (kNext) {
                if (j >= DB) {
#pragma unroll
                    for (int mi = (j - DB) * 4 / DB; mi < (j - DB + 1) * 4 / DB; ++mi) max_piece(sn, mi, lim_rel, mask_next);
                }
            }
            hook(KS + j);
        }
        __builtin_amdgcn_sched_barrier(0);
        MV_PSTAMP(slot0 + 2);
    };
    auto no_hook = [](int) __attribute__((always_inline)) {};

    // ---- prologue: K(0), V(0), K(1) staged; S of block 0 and its maxima
    Stage sr, sv;
    int pg3 = page_of(3);
    uint64_t off1, off2;   // byte offsets of tiles t + 1 and t + 2
    {
        const uint64_t off0 = tile_offset(0, page_of(0), 0);
        off1 = tile_offset(1, page_of(1), off0);
        off2 = tile_offset(2, page_of(2), off1);
        Stage s2;   // all three tiles requested before the first is waited for
        g_load(sr, a.k_cache, off0, 0);
        g_load(sv, a.v_cache, off0, 0);
        g_load(s2, a.k_cache, off1, 1);
        k_store(sr, 0, 0);
        k_store(sr, 0, 1);
        v_store(sv, 0, 0);
        v_store(sv, 0, 1);
        k_store(s2, 1, 0);
        k_store(s2, 1, 1);
    }
    __syncthreads();
    f32x16_t s0, s1;
#pragma unroll
    for (int r = 0; r < 16; ++r) s0[r] = 0.f;
#pragma unroll
    for (int i = 0; i < KS; ++i) s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k_frag(&kL[0][0], i), qf[i], s0, 0, 0, 0);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) max_piece(s0, mi, limit - 4 * hi, std::true_type{});
#pragma unroll
    for (int i = 0; i < PF; ++i) kpre[i] = k_frag(&kL[0][0] + 32 * KROW, i);

    // the second-dispatched half of the workgroup loses every arbitration against the first on
    // its SIMD (age); one static priority step evens that out (about 1 %)
    if (wid >= 4) __builtin_amdgcn_s_setprio(1);
    // One tile step: blocks 2 t (scores in s0) and 2 t + 1; leaves the scores of block 2 t + 2 in s0.
    int kc = 0, kn = 1, kw = 2;   // ring slots of K(t), K(t + 1), K(t + 2)
    auto tile_step = [&](int t, auto masks) __attribute__((always_inline)) {
        const unsigned char* kcur = &kL[kc][0];
        const unsigned char* knext = &kL[kn][0];
        const unsigned char* vcur = &vL[t & 1][0];
        MV_PSTAMP(0);
        // K(t + 2) and V(t + 1) requested now; they are written to LDS beside the last MFMAs of the
        // step, a whole step for the requests to land.  (Dealt out over the slots of the first block
        // step - wave w in slot w, so that the CU's address path, 16 cycles per 1-KiB request, never
        // has a queue - the uniform branches around them broke up the slots' instruction order and
        // the step took 10 % longer.)
        g_load(sr, a.k_cache, off2, t + 2);
        g_load(sv, a.v_cache, off1, t + 1);
        off1 = off2;
        off2 = tile_offset(t + 3, pg3, off2);
        pg3 = page_of(t + 4);   // its scalar load has a whole step to land
        block_step(s0, s1, kcur + 32 * KROW, vcur, knext, t * KT + 32, std::true_type{}, std::true_type{}, masks, t, 1,
                   no_hook);
        const int kdst = kw, vdst = (t + 1) & 1;
        block_step(s1, s0, knext, vcur + 32 * VROW, knext + 32 * KROW, (t + 1) * KT, std::true_type{}, std::true_type{},
                   masks, t, 4, [&](int q) __attribute__((always_inline)) {
                       // four (two) stores over the last slots of the step
                       // (behind the last MFMA, in front of the barrier, they cost 3 % more)
                       const int k = q - (KS + NP - 2 * NV);
                       if (k >= 0 && k < NV) k_store(sr, kdst, k);
                       if (k >= NV && k < 2 * NV) v_store(sv, vdst, k - NV);
                   });
        const int k_old = kc;
        kc = kn;
        kn = kw;
        kw = k_old;
        __syncthreads();
        MV_PSTAMP(7);
    };
    int t = 0;
    // (whether a block needs the mask is uniform over the wave; once one does, all later ones do)
    for (; t + 1 < ntiles && 2 * t + 2 < n_full; ++t) tile_step(t, std::false_type{});
    for (; t + 1 < ntiles; ++t) tile_step(t, std::true_type{});
    // the last tile: its second block has no successor
    block_step(s0, s1, &kL[kc][0] + 32 * KROW, &vL[t & 1][0], nullptr, t * KT + 32, std::true_type{}, std::false_type{},
               std::true_type{}, 1000, 1, no_hook);
    block_step(s1, s0, &kL[kc][0], &vL[t & 1][0] + 32 * VROW, nullptr, 0, std::false_type{}, std::false_type{},
               std::false_type{}, 1000, 4, no_hook);

#ifdef MV_PFX_STAMP
    __syncthreads();
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < 8 * 16 * 8; i += NT) g_pfx_stamp[i] = stampL[i];
    if (threadIdx.x == 0 && blockIdx.x < 1024) g_pfx_wg[blockIdx.x * 3 + 1] = wall_clock64();
#endif
    // oacc[db][r] = O^T[d = 32 db + (r & 3) + 8 (r >> 2) + 4 hi][this lane's query]
    float l = lrun0 + lrun1;
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    // A query's 8 output columns 8 k .. 8 k + 7 are split between its two lanes (4 each).  One
    // v_permlane32_swap per register pair of two neighbouring column groups leaves the lower lane
    // all 8 columns of group k and the upper lane those of group k + 1: 16-byte stores, half as many.
    uint16_t* op = a.out + ((int64_t)b * a.S + min(qi, a.S - 1)) * a.out_stride + (int64_t)head * D + 8 * hi;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int rq = 0; rq < 4; rq += 2) {
            uint32_t ax = pack_bf16x2(oacc[db][4 * rq] * inv, oacc[db][4 * rq + 1] * inv);
            uint32_t ay = pack_bf16x2(oacc[db][4 * rq + 2] * inv, oacc[db][4 * rq + 3] * inv);
            uint32_t bx = pack_bf16x2(oacc[db][4 * rq + 4] * inv, oacc[db][4 * rq + 5] * inv);
            uint32_t by = pack_bf16x2(oacc[db][4 * rq + 6] * inv, oacc[db][4 * rq + 7] * inv);
            const auto rx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
            const auto ry = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
            if (qi < a.S) *reinterpret_cast<uint4*>(op + db * 32 + rq * 8) = make_uint4(rx[0], ry[0], rx[1], ry[1]);
        }
}

}  // namespace mv

#ifdef MV_PFX_STAMP
extern "C" int mv_debug_read_pfx_stamps(void* dst, size_t bytes) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(mv::g_pfx_stamp), bytes) == hipSuccess ? 0 : -3;
}
extern "C" int mv_debug_read_pfx_workgroups(void* dst, size_t bytes) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(mv::g_pfx_wg), bytes) == hipSuccess ? 0 : -3;
}
#endif

extern "C" {

int mv_rope_store_prefill(void* qkv, int64_t row_stride, const void* rot_cos, const void* rot_sin,
                          int rot_max_pos, const int32_t* cache_seqlens, const int32_t* block_table,
                          int bt_stride, void* k_cache, void* v_cache, int bs, int seq, int hq,
                          int hkv, int d, int page, int num_pages, mv_stream_t stream) {
    using namespace mv;
    if (bs == 0 || seq == 0) return MV_OK;
    if (!qkv || !cache_seqlens || !block_table || !k_cache || !v_cache) return MV_ERR_INVALID_ARG;
    if (bs < 0 || seq < 0 || hq <= 0 || hkv <= 0 || page <= 0 || num_pages <= 0 || bt_stride <= 0)
        return MV_ERR_INVALID_ARG;
    if ((rot_cos == nullptr) != (rot_sin == nullptr)) return MV_ERR_INVALID_ARG;
    if (rot_cos && rot_max_pos <= 0) return MV_ERR_INVALID_ARG;
    if ((d % 16) != 0 || d > 256) return MV_ERR_UNSUPPORTED;
    if (page & (page - 1)) return MV_ERR_UNSUPPORTED;
    if (!aligned16(qkv) || (row_stride % 8) != 0 || row_stride < (int64_t)(hq + 2 * hkv) * d ||
        !aligned16(k_cache) || !aligned16(v_cache))
        return MV_ERR_INVALID_ARG;
    if (rot_cos && (!aligned16(rot_cos) || !aligned16(rot_sin))) return MV_ERR_INVALID_ARG;
    RopeStoreArgs a;
    a.qkv = (uint16_t*)qkv;
    a.row_stride = row_stride;
    a.cos = (const uint16_t*)rot_cos;
    a.sin = (const uint16_t*)rot_sin;
    a.rot_max_pos = rot_max_pos;
    a.start = cache_seqlens;
    a.block_table = block_table;
    a.bt_stride = bt_stride;
    a.k_cache = (uint16_t*)k_cache;
    a.v_cache = (uint16_t*)v_cache;
    a.B = bs;
    a.S = seq;
    a.hq = hq;
    a.hkv = hkv;
    a.d = d;
    a.page = page;
    a.page_shift = __builtin_ctz((unsigned)page);
    a.num_pages = num_pages;
    const int64_t total = (int64_t)bs * seq * (hq + 2 * hkv) * (d / 16);
    int64_t blocks = (total + 255) / 256;
    const int64_t cap = (int64_t)num_cus() * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(rope_store_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch();
}

int mv_rope_prefill(const void* x, int64_t x_stride, void* y, int64_t y_stride, const void* rot_cos,
                    const void* rot_sin, int rot_max_pos, int64_t tokens, int seq, int heads, int d,
                    mv_stream_t stream) {
    using namespace mv;
    if (tokens == 0) return MV_OK;
    if (!x || !y || !rot_cos || !rot_sin || tokens < 0 || seq <= 0 || heads <= 0 || rot_max_pos <= 0)
        return MV_ERR_INVALID_ARG;
    if ((d % 16) != 0 || d > 256) return MV_ERR_UNSUPPORTED;
    if (!aligned16(x) || !aligned16(y) || !aligned16(rot_cos) || !aligned16(rot_sin) ||
        (x_stride % 8) != 0 || (y_stride % 8) != 0 || x_stride < (int64_t)heads * d ||
        y_stride < (int64_t)heads * d)
        return MV_ERR_INVALID_ARG;
    const int64_t total = tokens * heads * (d / 16);
    int64_t blocks = (total + 255) / 256;
    const int64_t cap = (int64_t)num_cus() * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(rope_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)x, x_stride, (uint16_t*)y, y_stride, (const uint16_t*)rot_cos,
                       (const uint16_t*)rot_sin, rot_max_pos, tokens, seq, heads, d);
    return check_launch();
}

int mv_paged_attn_prefill_form(int bs, int seq, int hq, int hkv) {
    if (bs <= 0 || seq <= 0 || hq <= 0 || hkv <= 0 || hq % hkv) return 0;
    return mv::pf_eight_waves(bs, seq, hkv, hq / hkv) ? 8 : 4;
}

int mv_paged_attn_prefill(const void* q, int64_t q_stride, const void* k_cache, const void* v_cache,
                          const int32_t* cache_seqlens, const int32_t* block_table, int bt_stride,
                          void* out, int64_t out_stride, int bs, int seq, int hq, int hkv, int d,
                          int page, int num_pages, float scale, mv_stream_t stream) {
    using namespace mv;
    if (bs == 0 || seq == 0) return MV_OK;
    if (!q || !k_cache || !v_cache || !cache_seqlens || !block_table || !out) return MV_ERR_INVALID_ARG;
    if (bs < 0 || seq < 0 || hq <= 0 || hkv <= 0 || page <= 0 || num_pages <= 0 || bt_stride <= 0)
        return MV_ERR_INVALID_ARG;
    if (d != 64 && d != 128) return MV_ERR_UNSUPPORTED;
    if (hq % hkv) return MV_ERR_UNSUPPORTED;
    const int g = hq / hkv;
    if (g < 1 || g > 8) return MV_ERR_UNSUPPORTED;
    if ((page & (page - 1)) || page < 64) return MV_ERR_UNSUPPORTED;  // a 64-key tile sits in one page
    if (!aligned16(q) || !aligned16(k_cache) || !aligned16(v_cache) || !aligned16(out) ||
        (q_stride % 8) != 0 || (out_stride % 4) != 0 || q_stride < (int64_t)hq * d ||
        out_stride < (int64_t)hq * d)
        return MV_ERR_INVALID_ARG;
    PfArgs a;
    a.q = (const uint16_t*)q;
    a.q_stride = q_stride;
    a.k_cache = (const uint16_t*)k_cache;
    a.v_cache = (const uint16_t*)v_cache;
    a.start = cache_seqlens;
    a.block_table = block_table;
    a.bt_stride = bt_stride;
    a.out = (uint16_t*)out;
    a.out_stride = out_stride;
    a.B = bs;
    a.S = seq;
    a.hq = hq;
    a.hkv = hkv;
    a.g = g;
    const int gp = pf_group_slots(g);
    const int qt8 = 256 / gp;
    const bool wide = pf_eight_waves(bs, seq, hkv, g);
    a.qt = wide ? qt8 : 128 / gp;
    const bool eight = a.qt == qt8;
    a.qt_shift = __builtin_ctz((unsigned)a.qt);
    a.nqb = (seq + a.qt - 1) / a.qt;
    a.page = page;
    a.page_shift = __builtin_ctz((unsigned)page);
    a.num_pages = num_pages;
    a.scale_log2 = scale * 1.4426950408889634f;
    a.raise = g_pf_raise;
    const int64_t grid = (int64_t)a.nqb * hkv * bs;
    if (grid > 0x7fffffffLL) return MV_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    // the pipelined form keeps a page's byte size in 32 bits and stores 16 bytes at a time
    const bool pipe = eight && g_pf_pipe && (int64_t)page * hkv * d * 2 < (1LL << 32) && (out_stride % 8) == 0;
    if (d == 128 && pipe)
        hipLaunchKernelGGL((paged_attn_prefill_pipe_kernel<128>), dim3((unsigned)grid), dim3(512), 0, st, a);
    else if (pipe)
        hipLaunchKernelGGL((paged_attn_prefill_pipe_kernel<64>), dim3((unsigned)grid), dim3(512), 0, st, a);
    else if (d == 128 && eight)
        hipLaunchKernelGGL((paged_attn_prefill_kernel<128, 8>), dim3((unsigned)grid), dim3(512), 0, st, a);
    else if (d == 128)
        hipLaunchKernelGGL((paged_attn_prefill_kernel<128, 4>), dim3((unsigned)grid), dim3(256), 0, st, a);
    else if (eight)
        hipLaunchKernelGGL((paged_attn_prefill_kernel<64, 8>), dim3((unsigned)grid), dim3(512), 0, st, a);
    else
        hipLaunchKernelGGL((paged_attn_prefill_kernel<64, 4>), dim3((unsigned)grid), dim3(256), 0, st, a);
    return check_launch();
}

}  // extern "C"
