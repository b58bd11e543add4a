// One-shot all-reduce of small [rows, n] bf16 messages among the GPUs of one node, fused with the
// add + RMSNorm that follows it in a tensor-parallel decoder block.
//
// Why not a ring: the messages are 8-128 KiB (one hidden-state row per sequence), so the cost is
// latency, and xGMI is point to point - every GPU has a direct link to every other.  One launch
// per rank does everything:
//   push   each rank writes its partial row into slot [its rank] of EVERY rank's exchange buffer
//          (remote stores over the direct links, all links at once), then raises a flag there;
//   wait   it polls the flags the others raised in ITS buffer (local memory);
//   sum    it adds the `world` partials of its own buffer in rank order in fp32 - the same order
//          on every rank, so all ranks end with the same bits - rounds the sum to bf16 once, and
//          goes on with residual add and RMSNorm on the row it already holds in registers.
// One workgroup per row; rows synchronise independently (no grid barrier).
//
// The exchange buffers are uncached device memory shared through HIP IPC handles (one process per
// GPU); flags carry a per-row epoch that only grows, and data slots alternate with the epoch's
// parity, so nothing is ever reset and a captured graph can be replayed any number of times:
//   * rank r writes slot e&1 of peer p for epoch e only after it finished epoch e-1, in which it
//     saw p's flag e-1, which p raised after finishing epoch e-2 - p's last read of that slot;
//   * the epoch lives in device memory next to the buffer (one counter per row), not in a kernel
//     argument, so replays advance it like eager launches do.
// All ranks must issue the same sequence of calls with the same row counts (tensor parallelism
// does).  A peer that never arrives is given up on after a timeout: the kernel sets the
// communicator's status word and finishes (with a wrong sum), so a dead rank cannot hang the GPU.
// The status word lives in host memory mapped into the device (written with a system-scope store), so
// the host reads it with a plain load right after the wait it already does for the step's tokens -
// no extra synchronisation - and turns it into an error before the tokens are used.
#include "common.h"

#include <stdlib.h>
#include <string.h>

namespace mv {

constexpr int kArMaxWorld = 8;
constexpr int kArMaxRows = 64;
constexpr int kArThreads = 256;
constexpr int kArChunks = 4;               // 16-byte chunks of a row a thread keeps: n <= 8192
constexpr int kArMaxN = 8 * kArThreads * kArChunks;
constexpr int kArFlagStride = 32;          // uints between two rows' flags (128 bytes)
constexpr size_t kArFlagBytes = (size_t)kArMaxRows * kArFlagStride * sizeof(unsigned int);

struct ArArgs {
    unsigned char* peers[kArMaxWorld];  // every rank's exchange buffer as mapped here (peers[rank] = own)
    int rank, world;
    int max_rows, max_n;                // capacity the buffers were sized for
    unsigned int* epoch;                // [kArMaxRows] this rank's per-row epoch (plain device memory)
    int* status;                        // host-mapped word, set non-zero when a wait timed out
    unsigned long long timeout_ticks;   // of the constant 100 MHz counter
    // this rank's partial: bf16 rows, or fp32 k-slabs [ksplit][rows][n] of the projection that made it
    const uint16_t* in;
    int64_t ld_in;
    const float* slabs;
    int ksplit, rows;
    int n;
    // what follows the sum (all optional): s = t + residual; y = bf16(s); normed = rmsnorm(s) * ln_w
    const uint16_t* residual;
    int64_t ld_res;
    uint16_t* y;
    int64_t ldy;
    const uint16_t* ln_w;
    float eps;
    uint16_t* normed;
    int64_t ld_normed;
    int var_from_rounded;
    int fences;                         // full system-scope fences around the exchange (see the kernel)
};

// 16 bytes to / from an exchange buffer at system scope: two 8-byte relaxed atomics, which the
// compiler emits as sc0 sc1 accesses - stores write through to the destination, loads are served from
// it - so that neither side depends on a cache being written back or invalidated.
__device__ __forceinline__ void st16_system(void* p, uint4 v) {
    unsigned long long* q = reinterpret_cast<unsigned long long*>(p);
    __hip_atomic_store(q, ((unsigned long long)v.y << 32) | v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(q + 1, ((unsigned long long)v.w << 32) | v.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ uint4 ld16_system(const void* p) {
    const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
    const unsigned long long lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
}

__device__ __forceinline__ size_t ar_data_off(const ArArgs& a, int slot, int src, int row) {
    return kArFlagBytes + (((size_t)slot * a.world + src) * a.max_rows + row) * (size_t)a.max_n * 2;
}

__global__ __launch_bounds__(kArThreads) void allreduce_addnorm_kernel(ArArgs a) {
    __shared__ float red[kArThreads / 64];
    __shared__ unsigned int s_epoch;
    const int row = blockIdx.x;
    const int tid = threadIdx.x;
    const int nvec = a.n >> 3;

    if (tid == 0) s_epoch = a.epoch[row] + 1u;

    // ---- this rank's partial row, rounded to bf16 as the projection's output would be ----------
    uint4 mine[kArChunks];
    uint4 r_in[kArChunks], w_in[kArChunks];
#pragma unroll
    for (int j = 0; j < kArChunks; ++j) {
        const int c = min(tid + j * kArThreads, nvec - 1);   // chunks past the row end re-read the last
        const int n0 = c * 8;
        if (a.slabs) {
            // the first four slabs are requested before any is added (added in order either way)
            f32x4_t s_lo[4], s_hi[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks >= a.ksplit) break;   // uniform
                const float* p = a.slabs + ((int64_t)ks * a.rows + row) * a.n + n0;
                s_lo[ks] = *reinterpret_cast<const f32x4_t*>(p);
                s_hi[ks] = *reinterpret_cast<const f32x4_t*>(p + 4);
            }
            f32x4_t lo = (f32x4_t){0.f, 0.f, 0.f, 0.f}, hi = lo;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks < a.ksplit) {
                    lo += s_lo[ks];
                    hi += s_hi[ks];
                }
            }
            for (int ks = 4; ks < a.ksplit; ++ks) {
                const float* p = a.slabs + ((int64_t)ks * a.rows + row) * a.n + n0;
                lo += *reinterpret_cast<const f32x4_t*>(p);
                hi += *reinterpret_cast<const f32x4_t*>(p + 4);
            }
            mine[j] = make_uint4(pack_bf16x2(lo[0], lo[1]), pack_bf16x2(lo[2], lo[3]),
                                 pack_bf16x2(hi[0], hi[1]), pack_bf16x2(hi[2], hi[3]));
        } else {
            mine[j] = ld16(a.in + (int64_t)row * a.ld_in + n0);
        }
        r_in[j] = a.residual ? ld16(a.residual + (int64_t)row * a.ld_res + n0) : make_uint4(0, 0, 0, 0);
        w_in[j] = a.ln_w ? ld16(a.ln_w + n0) : make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    const unsigned int e = s_epoch;
    const int slot = (int)(e & 1u);

    // ---- push: my row into slot [rank] of every rank's buffer (my own included) -----------------
    for (int p = 0; p < a.world; ++p) {
        unsigned char* dst = a.peers[p] + ar_data_off(a, slot, a.rank, row);
#pragma unroll
        for (int j = 0; j < kArChunks; ++j) {
            const int c = tid + j * kArThreads;
            if (c < nvec) st16_system(dst + (size_t)c * 16, mine[j]);
        }
    }
    // Every thread's stores are performed system-wide before the flags go up.  The stores are
    // write-through (system scope) to uncached memory, so "performed" is what the wait for their
    // acknowledgements gives; a release FENCE would in addition write back every dirty line of this
    // XCD's L2 (KV rows, projection outputs: none of it the peers' business) and costs 1.5-2 us per
    // reduction.  `fences` (MVLLM_ALLREDUCE_FENCES=1) keeps the full fences for a platform where
    // the exchange memory turns out to be cached.
    if (a.fences)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid < a.world) {
        unsigned int* f = reinterpret_cast<unsigned int*>(a.peers[tid]) + row * kArFlagStride + a.rank;
        __hip_atomic_store(f, e, a.fences ? __ATOMIC_RELEASE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        // ---- wait: thread q watches the flag rank q raises here ----------------------------------
        const unsigned int* mf = reinterpret_cast<const unsigned int*>(a.peers[a.rank]) + row * kArFlagStride + tid;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (;;) {
            const unsigned int v = __hip_atomic_load(mf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((int)(v - e) >= 0) break;
            if (__builtin_amdgcn_s_memrealtime() - t0 > a.timeout_ticks) {
                __hip_atomic_store(a.status, 1 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    // the partials are read with system-scope loads (never from a cache): no invalidate needed
    if (a.fences)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    else
        asm volatile("" ::: "memory");

    // ---- sum in rank order, then the epilogue of a row-parallel projection + the next norm ------
    const unsigned char* own = a.peers[a.rank];
    uint4 keep[kArChunks];
    float ss = 0.f;
    // all partials of this thread's chunks are requested before the first is used: one round trip
    // to the buffer, not one per rank (chunk rows and ranks that do not exist are skipped: uniform)
    uint4 part[kArChunks][kArMaxWorld];
#pragma unroll
    for (int j = 0; j < kArChunks; ++j) {
        if (j * kArThreads >= nvec) break;
        const int c = min(tid + j * kArThreads, nvec - 1);
#pragma unroll
        for (int q = 0; q < kArMaxWorld; ++q) {
            if (q >= a.world) break;
            part[j][q] = ld16_system(own + ar_data_off(a, slot, q, row) + (size_t)c * 16);
        }
    }
#pragma unroll
    for (int j = 0; j < kArChunks; ++j) {
        const int c = tid + j * kArThreads;
        if (c < nvec) {
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < kArMaxWorld; ++q) {
                if (q < a.world) {   // uniform
                    const uint4 v = part[j][q];
                    const uint32_t v4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        acc[2 * k] += bf16_lo(v4[k]);
                        acc[2 * k + 1] += bf16_hi(v4[k]);
                    }
                }
            }
            const uint32_t r4[4] = {r_in[j].x, r_in[j].y, r_in[j].z, r_in[j].w};
            uint32_t o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float t0 = round_bf16(acc[2 * k]), t1 = round_bf16(acc[2 * k + 1]);
                float s0 = t0, s1 = t1;
                if (a.residual) {
                    s0 = t0 + bf16_lo(r4[k]);
                    s1 = t1 + bf16_hi(r4[k]);
                }
                o[k] = pack_bf16x2(s0, s1);
                if (a.var_from_rounded) {
                    s0 = bf16_lo(o[k]);
                    s1 = bf16_hi(o[k]);
                }
                ss += s0 * s0;
                ss += s1 * s1;
            }
            keep[j] = make_uint4(o[0], o[1], o[2], o[3]);
            if (a.y) st16(a.y + (int64_t)row * a.ldy + c * 8, keep[j]);
        }
    }
    if (tid == 0) a.epoch[row] = e;
    if (!a.ln_w) return;   // uniform
    // same chunk assignment and summation order as the stand-alone norm kernels
    const float total = block_sum(ss, red);
    const float rrms = 1.0f / sqrtf(total / (float)a.n + a.eps);
#pragma unroll
    for (int j = 0; j < kArChunks; ++j) {
        const int c = tid + j * kArThreads;
        if (c < nvec) {
            const uint32_t s4[4] = {keep[j].x, keep[j].y, keep[j].z, keep[j].w};
            const uint32_t w4[4] = {w_in[j].x, w_in[j].y, w_in[j].z, w_in[j].w};
            uint32_t o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                o[k] = pack_bf16x2((bf16_lo(s4[k]) * rrms) * bf16_lo(w4[k]),
                                   (bf16_hi(s4[k]) * rrms) * bf16_hi(w4[k]));
            st16(a.normed + (int64_t)row * a.ld_normed + c * 8, make_uint4(o[0], o[1], o[2], o[3]));
        }
    }
}

// ---- all-gather of [rows, n_local] bf16 slices into [rows, world * n_local] rows on every rank ------
// For the LM head cut by vocabulary rows over the ranks: every rank computes the logits of its slice
// and needs the whole row (greedy pick, sampler).  Same exchange as above - push into every rank's
// buffer, raise a per-row epoch flag there, wait for the flags raised here - but nothing is summed:
// rank r's slice lands at columns [r * n_local, (r + 1) * n_local) of slot e & 1 of every buffer, and
// each rank then copies the finished row out of its own buffer into ordinary memory (`out`), which is
// what the kernels behind it read.  One workgroup per row.
//   buffer: flags [kArMaxRows][kArFlagStride] | data [2][max_rows][world * max_n] bf16
struct AgArgs {
    unsigned char* peers[kArMaxWorld];
    int rank, world, max_rows, max_n;
    unsigned int* epoch;
    int* status;
    unsigned long long timeout_ticks;
    const uint16_t* in;
    int64_t ld_in;
    uint16_t* out;
    int64_t ld_out;
    int n_local;
};
constexpr int kAgThreads = 512;

// 16 bytes to a peer's buffer at system scope in one store (the flags order it; nothing reads it
// before they are up, so it need not be atomic)
__device__ __forceinline__ void st16_system_wide(void* p, uint4 v) {
    const u32x4_t d = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(d) : "memory");
}

__global__ __launch_bounds__(kAgThreads) void allgather_kernel(AgArgs a) {
    __shared__ unsigned int s_epoch;
    const int row = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) s_epoch = a.epoch[row] + 1u;
    __syncthreads();
    const unsigned int e = s_epoch;
    const size_t row_bytes = (size_t)a.world * a.max_n * 2;
    const size_t base = kArFlagBytes + ((size_t)(e & 1u) * a.max_rows + row) * row_bytes;
    const int nvec = a.n_local >> 3;
    // ---- push my slice into every rank's row (my own included) ---------------------------------
    for (int c = tid; c < nvec; c += kAgThreads) {
        const uint4 v = ld16(a.in + (int64_t)row * a.ld_in + c * 8);
        for (int p = 0; p < a.world; ++p)
            st16_system_wide(a.peers[p] + base + ((size_t)a.rank * a.n_local + c * 8) * 2, v);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every store performed before the flags go up
    __syncthreads();
    if (tid < a.world) {
        unsigned int* f = reinterpret_cast<unsigned int*>(a.peers[tid]) + row * kArFlagStride + a.rank;
        __hip_atomic_store(f, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned int* mf = reinterpret_cast<const unsigned int*>(a.peers[a.rank]) + row * kArFlagStride + tid;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (;;) {
            const unsigned int v = __hip_atomic_load(mf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((int)(v - e) >= 0) break;
            if (__builtin_amdgcn_s_memrealtime() - t0 > a.timeout_ticks) {
                __hip_atomic_store(a.status, 1 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    // ---- the finished row out of my own buffer (system-scope loads: never from a cache) ---------
    const unsigned char* own = a.peers[a.rank] + base;
    const int total = a.world * nvec;
    constexpr int kAhead = 4;   // loads in flight per thread
    for (int c0 = tid; c0 < total; c0 += kAgThreads * kAhead) {
        uint4 v[kAhead];
#pragma unroll
        for (int j = 0; j < kAhead; ++j) {
            const int c = min(c0 + j * kAgThreads, total - 1);
            v[j] = ld16_system(own + (size_t)c * 16);
        }
#pragma unroll
        for (int j = 0; j < kAhead; ++j) {
            const int c = c0 + j * kAgThreads;
            if (c < total) st16(a.out + (int64_t)row * a.ld_out + (int64_t)c * 8, v[j]);
        }
    }
    if (tid == 0) a.epoch[row] = e;
}

struct ArComm {
    int rank = 0, world = 1, device = 0;
    int max_rows = 0, max_n = 0;
    bool gather = false;   // an all-gather communicator (mv_allgather_*): max_n = columns of ONE rank
    size_t bytes = 0;
    unsigned char* local = nullptr;
    unsigned char* peers[kArMaxWorld] = {};
    bool opened[kArMaxWorld] = {};
    hipIpcMemHandle_t handle;
    unsigned int* epoch = nullptr;
    int* status = nullptr;        // device address of the status word
    int* status_host = nullptr;   // the same word as the host sees it (pinned, mapped, coherent)
    unsigned long long timeout_ticks = 0;
    bool connected = false;
    bool fences = false;   // MVLLM_ALLREDUCE_FENCES=1
};

static int ar_fail(hipError_t e) {
    set_last_hip_error(e);
    return MV_ERR_LAUNCH;
}

}  // namespace mv

extern "C" {

size_t mv_allreduce_handle_bytes(void) { return sizeof(hipIpcMemHandle_t); }

static int comm_init(int rank, int world, int max_rows, int max_n, bool gather, void** comm_out);

int mv_allreduce_init(int rank, int world, int max_rows, int max_n, void** comm_out) {
    using namespace mv;
    if (!comm_out || world < 1 || world > kArMaxWorld || rank < 0 || rank >= world) return MV_ERR_INVALID_ARG;
    if (max_rows < 1 || max_rows > kArMaxRows || max_n < 8 || (max_n % 8) != 0 || max_n > kArMaxN)
        return MV_ERR_UNSUPPORTED;
    return comm_init(rank, world, max_rows, max_n, false, comm_out);
}

int mv_allgather_init(int rank, int world, int max_rows, int n_local, void** comm_out) {
    using namespace mv;
    if (!comm_out || world < 1 || world > kArMaxWorld || rank < 0 || rank >= world) return MV_ERR_INVALID_ARG;
    if (max_rows < 1 || max_rows > kArMaxRows || n_local < 8 || (n_local % 8) != 0 || n_local > (1 << 20))
        return MV_ERR_UNSUPPORTED;
    return comm_init(rank, world, max_rows, n_local, true, comm_out);
}

static int comm_init(int rank, int world, int max_rows, int max_n, bool gather, void** comm_out) {
    using namespace mv;
    ArComm* c = new ArComm();
    c->rank = rank;
    c->world = world;
    c->max_rows = max_rows;
    c->max_n = max_n;
    c->gather = gather;
    hipError_t e = hipGetDevice(&c->device);
    if (e != hipSuccess) { delete c; return ar_fail(e); }
    // (the same size either way: 2 slots x max_rows rows of world x max_n columns)
    c->bytes = kArFlagBytes + (size_t)2 * world * max_rows * max_n * 2;
    // Uncached device memory: stores go straight to memory and loads come from it, on this GPU and
    // on the peers that map it (what a flag protocol between agents needs).
    void* p = nullptr;
    e = hipExtMallocWithFlags(&p, c->bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        e = hipExtMallocWithFlags(&p, c->bytes, hipDeviceMallocFinegrained);
    }
    if (e != hipSuccess) { delete c; return ar_fail(e); }
    c->local = (unsigned char*)p;
    c->peers[rank] = c->local;
    void *ep = nullptr, *sp = nullptr, *sp_dev = nullptr;
    if ((e = hipMalloc(&ep, kArMaxRows * sizeof(unsigned int))) != hipSuccess ||
        (e = hipHostMalloc(&sp, 256, hipHostMallocMapped | hipHostMallocCoherent)) != hipSuccess ||
        (e = hipHostGetDevicePointer(&sp_dev, sp, 0)) != hipSuccess ||
        (e = hipMemset(c->local, 0, c->bytes)) != hipSuccess ||
        (e = hipMemset(ep, 0, kArMaxRows * sizeof(unsigned int))) != hipSuccess ||
        (e = hipDeviceSynchronize()) != hipSuccess) {
        if (ep) (void)hipFree(ep);
        if (sp) (void)hipHostFree(sp);
        (void)hipFree(c->local);
        delete c;
        return ar_fail(e);
    }
    memset(sp, 0, 256);
    c->epoch = (unsigned int*)ep;
    c->status_host = (int*)sp;
    c->status = (int*)sp_dev;
    memset(&c->handle, 0, sizeof(c->handle));
    if (world > 1) {
        e = hipIpcGetMemHandle(&c->handle, c->local);
        if (e != hipSuccess) {
            (void)hipFree(ep); (void)hipHostFree(sp); (void)hipFree(c->local);
            delete c;
            return ar_fail(e);
        }
    }
    // how long a rank waits for its peers before it gives up (a dead peer must not hang the GPU)
    double ms = 10000.0;
    if (const char* s = getenv("MVLLM_ALLREDUCE_TIMEOUT_MS")) {
        const double v = atof(s);
        if (v > 0.0) ms = v;
    }
    c->timeout_ticks = (unsigned long long)(ms * 1.0e5);   // 100 MHz
    if (const char* f = getenv("MVLLM_ALLREDUCE_FENCES")) c->fences = atoi(f) != 0;
    c->connected = (world == 1);
    *comm_out = c;
    return MV_OK;
}

int mv_allreduce_get_handle(void* comm, void* handle_out) {
    if (!comm || !handle_out) return MV_ERR_INVALID_ARG;
    memcpy(handle_out, &((mv::ArComm*)comm)->handle, sizeof(hipIpcMemHandle_t));
    return MV_OK;
}

int mv_allreduce_connect(void* comm, const void* handles) {
    using namespace mv;
    if (!comm || !handles) return MV_ERR_INVALID_ARG;
    ArComm* c = (ArComm*)comm;
    if (c->connected) return MV_OK;
    const hipIpcMemHandle_t* h = (const hipIpcMemHandle_t*)handles;
    for (int p = 0; p < c->world; ++p) {
        if (p == c->rank) continue;
        void* mapped = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&mapped, h[p], hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess || !mapped) return ar_fail(e != hipSuccess ? e : hipErrorInvalidValue);
        c->peers[p] = (unsigned char*)mapped;
        c->opened[p] = true;
    }
    c->connected = true;
    return MV_OK;
}

int mv_allreduce_set_timeout_ms(void* comm, double ms) {
    if (!comm || !(ms > 0.0)) return MV_ERR_INVALID_ARG;
    ((mv::ArComm*)comm)->timeout_ticks = (unsigned long long)(ms * 1.0e5);   // 100 MHz
    return MV_OK;
}

int mv_allreduce_status(void* comm, int* status_out) {
    if (!comm || !status_out) return MV_ERR_INVALID_ARG;
    // a plain load of host memory: it tells about the step calls the caller has already waited for
    *status_out = *(volatile int*)((mv::ArComm*)comm)->status_host;
    return MV_OK;
}

int mv_allreduce_destroy(void* comm) {
    using namespace mv;
    if (!comm) return MV_OK;
    ArComm* c = (ArComm*)comm;
    (void)hipDeviceSynchronize();
    for (int p = 0; p < c->world; ++p)
        if (c->opened[p]) (void)hipIpcCloseMemHandle(c->peers[p]);
    if (c->epoch) (void)hipFree(c->epoch);
    if (c->status_host) (void)hipHostFree(c->status_host);
    if (c->local) (void)hipFree(c->local);
    delete c;
    return MV_OK;
}

int mv_allreduce_addnorm(void* comm, const void* in, int64_t ld_in, const float* slabs, int ksplit,
                         const void* residual, int64_t ld_res, void* y, int64_t ldy, const void* ln_w,
                         float eps, void* normed, int64_t ld_normed, int norm_flags, int rows, int n,
                         mv_stream_t stream) {
    using namespace mv;
    if (!comm || rows <= 0 || n <= 0) return MV_ERR_INVALID_ARG;
    ArComm* c = (ArComm*)comm;
    if (!c->connected || c->gather) return MV_ERR_INVALID_ARG;
    if ((in == nullptr) == (slabs == nullptr)) return MV_ERR_INVALID_ARG;   // exactly one source
    if (slabs && ksplit < 1) return MV_ERR_INVALID_ARG;
    if (rows > c->max_rows || n > c->max_n || (n % 8) != 0) return MV_ERR_UNSUPPORTED;
    if ((ln_w == nullptr) != (normed == nullptr)) return MV_ERR_INVALID_ARG;
    if (!y && !normed) return MV_ERR_INVALID_ARG;
    if (in && (!aligned16(in) || (ld_in % 8) != 0 || ld_in < n)) return MV_ERR_INVALID_ARG;
    if (slabs && !aligned16(slabs)) return MV_ERR_INVALID_ARG;
    if (residual && (!aligned16(residual) || (ld_res % 8) != 0 || ld_res < n)) return MV_ERR_INVALID_ARG;
    if (y && (!aligned16(y) || (ldy % 8) != 0 || ldy < n)) return MV_ERR_INVALID_ARG;
    if (normed && (!aligned16(normed) || !aligned16(ln_w) || (ld_normed % 8) != 0 || ld_normed < n))
        return MV_ERR_INVALID_ARG;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev != c->device) return MV_ERR_INVALID_ARG;

    ArArgs a = {};
    for (int p = 0; p < c->world; ++p) a.peers[p] = c->peers[p];
    a.rank = c->rank;
    a.world = c->world;
    a.max_rows = c->max_rows;
    a.max_n = c->max_n;
    a.epoch = c->epoch;
    a.status = c->status;
    a.timeout_ticks = c->timeout_ticks;
    a.fences = c->fences ? 1 : 0;
    a.in = (const uint16_t*)in;
    a.ld_in = ld_in;
    a.slabs = slabs;
    a.ksplit = ksplit;
    a.rows = rows;
    a.n = n;
    a.residual = (const uint16_t*)residual;
    a.ld_res = ld_res;
    a.y = (uint16_t*)y;
    a.ldy = ldy;
    a.ln_w = (const uint16_t*)ln_w;
    a.eps = eps;
    a.normed = (uint16_t*)normed;
    a.ld_normed = ld_normed;
    a.var_from_rounded = (norm_flags & MV_NORM_VAR_FROM_ROUNDED) ? 1 : 0;
    hipLaunchKernelGGL(allreduce_addnorm_kernel, dim3(rows), dim3(kArThreads), 0, (hipStream_t)stream, a);
    return check_launch();
}

int mv_allgather_rows(void* comm, const void* in, int64_t ld_in, void* out, int64_t ld_out, int rows,
                      int n_local, mv_stream_t stream) {
    using namespace mv;
    if (!comm || !in || !out || rows <= 0 || n_local <= 0) return MV_ERR_INVALID_ARG;
    ArComm* c = (ArComm*)comm;
    if (!c->connected || !c->gather) return MV_ERR_INVALID_ARG;
    if (rows > c->max_rows || n_local > c->max_n || (n_local % 8) != 0) return MV_ERR_UNSUPPORTED;
    if (!aligned16(in) || !aligned16(out) || (ld_in % 8) != 0 || (ld_out % 8) != 0 || ld_in < n_local ||
        ld_out < (int64_t)c->world * n_local)
        return MV_ERR_INVALID_ARG;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev != c->device) return MV_ERR_INVALID_ARG;
    AgArgs a = {};
    for (int p = 0; p < c->world; ++p) a.peers[p] = c->peers[p];
    a.rank = c->rank;
    a.world = c->world;
    a.max_rows = c->max_rows;
    a.max_n = c->max_n;
    a.epoch = c->epoch;
    a.status = c->status;
    a.timeout_ticks = c->timeout_ticks;
    a.in = (const uint16_t*)in;
    a.ld_in = ld_in;
    a.out = (uint16_t*)out;
    a.ld_out = ld_out;
    a.n_local = n_local;
    hipLaunchKernelGGL(allgather_kernel, dim3(rows), dim3(kAgThreads), 0, (hipStream_t)stream, a);
    return check_launch();
}

int mv_allreduce_step(void* comm, const void* in, int64_t ld_in, void* out, int64_t ld_out, int rows,
                      int n, mv_stream_t stream) {
    return mv_allreduce_addnorm(comm, in, ld_in, nullptr, 0, nullptr, 0, out, ld_out, nullptr, 0.f, nullptr,
                                0, 0, rows, n, stream);
}

}  // extern "C"
