// Microbenchmark: what a barrier among the 32 workgroups of ONE XCD costs against a barrier among all
// 256 workgroups of the chip (one workgroup per CU, flags in device memory).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/xcd_barrier tools/micro/xcd_barrier.hip && /tmp/xcd_barrier
// Three flag forms each: sc0 sc1 stores and loads (system scope), plain stores with sc1 loads (the line
// stays in the writer's L2; an sc1 load passes L1), sc1 stores with sc1 loads (agent scope).
// Every spin is bounded: a barrier that does not complete within kSpinMax polls is given up on and the
// kernel reports it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kSpinMax = 1 << 22;

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}
__device__ __forceinline__ void st_plain(unsigned* p, unsigned v) {
    asm volatile("global_store_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void st_sc1(unsigned* p, unsigned v) {
    asm volatile("global_store_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ unsigned ld_sc1(const unsigned* p) {
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void st_global(unsigned* p, unsigned v) {
    asm volatile("global_store_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ unsigned ld_global(const unsigned* p) {
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// flags: [256] one per workgroup (64 bytes apart); slot claim counters per XCD
struct Args {
    unsigned* flags;      // [256 * 16]
    unsigned* claim;      // [8 * 16]
    unsigned* xcd_of;     // [256] out
    unsigned* failed;     // out
    long long* ticks;     // [256] out: s_memtime ticks for the rounds
    int rounds;
    int local;            // 1 = barrier among the XCD's workgroups, 0 = among all
    int mode;             // 0 = sc0 sc1 stores and loads, 1 = plain stores + sc1 loads, 2 = sc1 stores + sc1 loads
};

__global__ __launch_bounds__(256) void barrier_kernel(Args a) {
    __shared__ unsigned s_slot, s_xcd;
    const int wg = blockIdx.x;
    if (threadIdx.x == 0) {
        s_xcd = xcc_id();
        s_slot = atomicAdd(&a.claim[s_xcd * 16], 1u);   // rank inside the XCD
        a.xcd_of[wg] = s_xcd * 64 + s_slot;
    }
    __syncthreads();
    const unsigned xcd = s_xcd, slot = s_slot;
    // my flag lives at (xcd * 32 + slot)
    unsigned* mine = a.flags + (xcd * 32 + slot) * 16;
    long long t0 = 0, t1 = 0;
    bool ok = true;
    for (int r = 1; r <= a.rounds && ok; ++r) {
        if (r == 2 && threadIdx.x == 0) t0 = __builtin_readcyclecounter();
        __syncthreads();
        if (threadIdx.x == 0) {
            if (a.mode == 0) st_global(mine, (unsigned)r);
            else if (a.mode == 1) st_plain(mine, (unsigned)r);
            else st_sc1(mine, (unsigned)r);
        }
        // lanes poll the flags of the group: 32 (local) or 256 (global)
        const int n = a.local ? 32 : 256;
        if (threadIdx.x < n) {
            const unsigned* f = a.local ? a.flags + (xcd * 32 + threadIdx.x) * 16 : a.flags + threadIdx.x * 16;
            int spins = 0;
            while (true) {
                const unsigned v = a.mode == 0 ? ld_global(f) : ld_sc1(f);
                if (v >= (unsigned)r) break;
                if (++spins > kSpinMax) { ok = false; atomicAdd(a.failed, 1u); break; }
            }
        }
        ok = __syncthreads_and(ok);
    }
    if (threadIdx.x == 0) {
        t1 = __builtin_readcyclecounter();
        a.ticks[wg] = t1 - t0;
    }
}

int main() {
    const int rounds = 2001;
    unsigned *flags, *claim, *xcd_of, *failed;
    long long* ticks;
    hipMalloc(&flags, 256 * 16 * 4);
    hipMalloc(&claim, 8 * 16 * 4);
    hipMalloc(&xcd_of, 256 * 4);
    hipMalloc(&failed, 4);
    hipMalloc(&ticks, 256 * 8);
    for (int cfg = 0; cfg < 6; ++cfg) {
        const int local = cfg & 1, mode = cfg >> 1;
        hipMemset(flags, 0, 256 * 16 * 4);
        hipMemset(claim, 0, 8 * 16 * 4);
        hipMemset(failed, 0, 4);
        Args a{flags, claim, xcd_of, failed, ticks, rounds, local, mode};
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(barrier_kernel, dim3(256), dim3(256), 0, 0, a);
        hipEventRecord(e1);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned> x(256);
        unsigned f;
        hipMemcpy(x.data(), xcd_of, 256 * 4, hipMemcpyDeviceToHost);
        hipMemcpy(&f, failed, 4, hipMemcpyDeviceToHost);
        int per[8] = {0};
        for (int i = 0; i < 256; ++i) per[(x[i] / 64) & 7]++;
        printf("%s, %s: %d rounds in %.3f ms = %.3f us per barrier (failed polls %u); workgroups per XCD:",
               local ? "XCD-local" : "chip-wide",
               mode == 0 ? "sc0 sc1 stores and loads" : mode == 1 ? "plain stores, sc1 loads" : "sc1 stores, sc1 loads", rounds, ms, ms * 1e3 / rounds, f);
        for (int i = 0; i < 8; ++i) printf(" %d", per[i]);
        printf("\n");
    }
    return 0;
}
