// Microbenchmark: the rate at which the access pattern of the batched decode attention can be read,
// with no arithmetic behind it - to tell what the kernel loses to its pattern from what it loses to
// its own structure.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/kv_stream tools/micro/kv_stream.hip && tools/micro/kv_stream
// Layout as the KV pages: rows of Hkv x D bf16 (2 KiB for 8 kv heads of 128), 256 rows a page, the pages
// of a sequence scattered by a shuffled block table, K and V in two pools.  A workgroup streams the
// rows of one (sequence, kv head): 16 lanes per row (16 bytes each), four rows per wave instruction, kAC
// instructions per stage and pool, two stages in flight - the attention kernel's own shape - and
// xors what it reads into one register.  Variants: waves per workgroup, splits of a sequence,
// stages in flight, and the same bytes read as one contiguous stream (the projections' pattern).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <string>
#include <vector>

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));            \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

struct Args {
    const uint4* k;       // [pages][256][hkv][D/8] 16-byte vectors
    const uint4* v;
    const int* table;     // [bs][pages_per_seq]
    unsigned* sink;
    int bs, hkv, ctx, pages_per_seq, nsplit;
    int contiguous;       // 1: ignore the pattern, read the same number of bytes as one linear stream
};

template <int W, int kAC>
__global__ __launch_bounds__(W * 64) void stream_kernel(Args a) {
    constexpr int LPR = 16, RPI = 4, NS = W * RPI, CR = NS * kAC;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int lr = lane % LPR, tg = lane / LPR;
    const int split = blockIdx.x % a.nsplit, bh = blockIdx.x / a.nsplit;
    const int kvh = bh % a.hkv, b = bh / a.hkv;
    const int* bt = a.table + (int64_t)b * a.pages_per_seq;
    const int64_t row_vecs = (int64_t)a.hkv * LPR;   // 16-byte vectors per row
    const int seq_chunks = (a.ctx + CR - 1) / CR;
    const int nchunk = seq_chunks > split ? (seq_chunks - split + a.nsplit - 1) / a.nsplit : 0;
    uint4 acc = make_uint4(0, 0, 0, 0);
    auto eat = [&](const uint4& x) {
        acc.x ^= x.x; acc.y ^= x.y; acc.z ^= x.z; acc.w ^= x.w;
    };
    if (a.contiguous) {
        // this workgroup's share of the bytes, back to back: ctx/nsplit rows x 256 B x 2 pools
        const int64_t vecs = (int64_t)nchunk * CR * LPR;           // per pool
        const int64_t base = (int64_t)blockIdx.x * vecs;
        for (int64_t i = threadIdx.x; i < vecs; i += (int64_t)W * 64 * 4) {
            uint4 t[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t o = std::min(i + (int64_t)j * W * 64, vecs - 1);
                t[j] = a.k[base + o];
                t[4 + j] = a.v[base + o];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) eat(t[j]);
        }
    } else {
        auto row_of = [&](int c, int i) { return (((c * a.nsplit + split) * kAC + i) * W + wid) * RPI + tg; };
        auto load = [&](uint4(&kb)[kAC], uint4(&vb)[kAC], int c) {
#pragma unroll
            for (int i = 0; i < kAC; ++i) {
                const int t = std::min(row_of(c, i), a.ctx - 1);
                const int pg = bt[t >> 8];
                const int64_t off = ((int64_t)pg * 256 + (t & 255)) * row_vecs + kvh * LPR + lr;
                kb[i] = a.k[off];
                vb[i] = a.v[off];
            }
        };
        uint4 kA[kAC], vA[kAC], kB[kAC], vB[kAC];
        load(kA, vA, 0);
        load(kB, vB, std::min(1, std::max(nchunk - 1, 0)));
        int c = 0;
        for (; c + 3 < nchunk; c += 2) {
#pragma unroll
            for (int i = 0; i < kAC; ++i) { eat(kA[i]); eat(vA[i]); }
            __builtin_amdgcn_sched_barrier(0);
            load(kA, vA, c + 2);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < kAC; ++i) { eat(kB[i]); eat(vB[i]); }
            __builtin_amdgcn_sched_barrier(0);
            load(kB, vB, c + 3);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < kAC; ++i) { eat(kA[i]); eat(vA[i]); eat(kB[i]); eat(vB[i]); }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) a.sink[0] = 1;   // keeps the loads alive
}

// average duration of a launch, the pools visited in turn `reps` times between two events
template <int W, int kAC>
static double run(const std::vector<Args>& as, int reps) {
    const int grid = as[0].bs * as[0].hkv * as[0].nsplit;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (const Args& a : as) hipLaunchKernelGGL((stream_kernel<W, kAC>), dim3(grid), dim3(W * 64), 0, 0, a);
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i)
        for (const Args& a : as) hipLaunchKernelGGL((stream_kernel<W, kAC>), dim3(grid), dim3(W * 64), 0, 0, a);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
    return ms * 1e3 / (reps * as.size());
}

int main(int argc, char** argv) {
    const int bs = argc > 1 ? atoi(argv[1]) : 32, hkv = 8, D = 128, ctx = argc > 2 ? atoi(argv[2]) : 4096;
    const int layers = 6;   // distinct pools visited in turn, so that no launch finds its rows in a cache
    const int pps = (ctx + 255) / 256 + 1, pages = bs * pps;
    const size_t pool_vecs = (size_t)pages * 256 * hkv * (D / 8);
    std::vector<uint4*> k(layers), v(layers);
    for (int l = 0; l < layers; ++l) {
        CHECK(hipMalloc(&k[l], pool_vecs * 16));
        CHECK(hipMalloc(&v[l], pool_vecs * 16));
        CHECK(hipMemset(k[l], l + 1, pool_vecs * 16));
        CHECK(hipMemset(v[l], l + 7, pool_vecs * 16));
    }
    std::vector<int> perm(pages);
    std::iota(perm.begin(), perm.end(), 0);
    std::mt19937 rng(2);
    std::shuffle(perm.begin(), perm.end(), rng);
    int* table;
    unsigned* sink;
    CHECK(hipMalloc(&table, pages * sizeof(int)));
    CHECK(hipMemcpy(table, perm.data(), pages * sizeof(int), hipMemcpyHostToDevice));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(sink, 0, 64));
    const double mb = (double)bs * hkv * ctx * D * 2 * 2 / 1e6;
    printf("bs %d, %d kv heads x %d, ctx %d: %.1f MB per launch, %d pools in turn\n", bs, hkv, D, ctx, mb, layers);
    auto report = [&](const char* what, auto kernel_run, int nsplit, int contiguous) {
        std::vector<Args> as;
        for (int l = 0; l < layers; ++l) as.push_back(Args{k[l], v[l], table, sink, bs, hkv, ctx, pps, nsplit, contiguous});
        const double us = kernel_run(as, 4);
        printf("  %-58s %8.2f us  %7.1f GB/s\n", what, us, mb / us * 1e3);
    };
    if (argc > 3 && std::string(argv[3]) == "calib") {
        // Calibration of byte counters (rocprofv3 --pmc FETCH_SIZE -- kv_stream BS CTX calib SPLITS):
        // launches of ONE variant - the attention kernel's own shape, 4 waves, 4 rows x 4 per stage -
        // each of which reads exactly `mb` MB (every row of every sequence and kv head once, K and V),
        // from pools visited in turn so that no launch finds its rows in a cache.
        const int nsplit = argc > 4 ? atoi(argv[4]) : 1;
        std::vector<Args> as;
        for (int l = 0; l < layers; ++l) as.push_back(Args{k[l], v[l], table, sink, bs, hkv, ctx, pps, nsplit, 0});
        const double us = run<4, 4>(as, 4);
        printf("calib: stream_kernel<4,4> %d splits, %d launches, %.0f bytes per launch, %.2f us\n", nsplit,
               (int)(5 * as.size()), mb * 1e6, us);
        return 0;
    }
    for (int pass = 0; pass < 2; ++pass) {
        printf(" pass %d\n", pass);
        report("paged, 4 waves, 4 rows x 4 per stage, 1 split", run<4, 4>, 1, 0);
        report("paged, 4 waves, 4 x 4, 2 splits", run<4, 4>, 2, 0);
        report("paged, 8 waves, 4 x 4, 1 split", run<8, 4>, 1, 0);
        report("paged, 4 waves, 4 x 8 per stage, 1 split", run<4, 8>, 1, 0);
        report("paged, 8 waves, 4 x 8, 1 split", run<8, 8>, 1, 0);
        report("paged, 16 waves, 4 x 4, 1 split", run<16, 4>, 1, 0);
        report("contiguous, 4 waves", run<4, 4>, 1, 1);
        report("contiguous, 8 waves", run<8, 4>, 1, 1);
        report("contiguous, 16 waves", run<16, 4>, 1, 1);
    }
    return 0;
}
