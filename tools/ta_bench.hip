// tools/ta_bench.hip -- what ONE wave64 `global_load_dwordx4` costs the CU's texture addresser / L1 as a function of how its 64 lanes'
// 16-byte records lie in memory.  k_persist's stale-row fetches are such loads (tp_persist.h: pk_walk_pass) and a patch whose rows are all
// stale spends 6-7 us of a grad-iter on 768 x 15 of them (profiles/r06_meninas_timeline.json): about one lane per clock.  Which layouts
// would the hardware coalesce?
//
// 256 workgroups x 768 threads (one per CU, as k_persist), every thread N x 8 loads; patterns (g lanes share a block):
//   contiguous      lane i -> base + 16 i                                   (the best case: 1 KB per wave-load)
//   scattered       every lane its own 128-byte line
//   pair32          2 lanes: 32 contiguous bytes of one line
//   quad64          4 lanes: 64 contiguous bytes
//   quad_stride32   4 lanes: one 128-byte line, 16 bytes every 32
//   oct128          8 lanes: one whole line
//   quad_same       4 lanes: the same 16 bytes
//   quad_2lines     4 lanes: 2 + 2 in two adjacent lines (32 contiguous bytes each)
// over regions of 16 KB (L1 hits), 96 KB (L2 hits) and 1.5 MB (past the L2) per workgroup.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/ta_bench.hip -o /tmp/tab && /tmp/tab
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define THREADS 768
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

struct args { const char* table; unsigned region; int pattern, rounds; unsigned long long* out; unsigned* sink; };
// MOD: 0 plain, 1 sc0, 2 sc1, 3 nt, 4 sc0 sc1, 5 sc1 nt -- eight loads and the wait for them in ONE asm statement (the compiler does not count
// loads it did not issue: their registers must not be touched before the data has landed)
#define LD8(BITS) asm volatile( \
    "global_load_dwordx4 %0, %8, off " BITS "\n global_load_dwordx4 %1, %9, off " BITS "\n global_load_dwordx4 %2, %10, off " BITS "\n global_load_dwordx4 %3, %11, off " BITS "\n" \
    "global_load_dwordx4 %4, %12, off " BITS "\n global_load_dwordx4 %5, %13, off " BITS "\n global_load_dwordx4 %6, %14, off " BITS "\n global_load_dwordx4 %7, %15, off " BITS "\n" \
    "s_waitcnt vmcnt(0)" \
    : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]) \
    : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]) : "memory")
template <int MOD> __device__ __forceinline__ void ld8(const char* const p[8], u4 v[8]) {
    if (MOD == 0) LD8("");
    else if (MOD == 1) LD8("sc0");
    else if (MOD == 2) LD8("sc1");
    else if (MOD == 3) LD8("nt");
    else if (MOD == 4) LD8("sc0 sc1");
    else LD8("sc1 nt");
}
__device__ __forceinline__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int MOD>
__global__ __launch_bounds__(THREADS) void k_ta(args A) {
    extern __shared__ char smem[];
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const unsigned lines = A.region >> 7;   // 128-byte lines of the workgroup's region
    unsigned off;
    switch (A.pattern) {
        case 0: off = (tid * 16u) % A.region; break;
        case 1: off = (hash32(tid * 2654435761u + 17u) % lines) * 128u + (hash32(tid) & 7u) * 16u; break;
        case 2: off = (hash32((wave * 64u + (lane >> 1)) * 2654435761u + 17u) % lines) * 128u + (hash32(wave * 64 + (lane >> 1)) & 3u) * 32u + (lane & 1u) * 16u; break;
        case 3: off = (hash32((wave * 64u + (lane >> 2)) * 2654435761u + 17u) % lines) * 128u + (hash32(wave * 64 + (lane >> 2)) & 1u) * 64u + (lane & 3u) * 16u; break;
        case 4: off = (hash32((wave * 64u + (lane >> 2)) * 2654435761u + 17u) % lines) * 128u + (lane & 3u) * 32u; break;
        case 5: off = (hash32((wave * 64u + (lane >> 3)) * 2654435761u + 17u) % lines) * 128u + (lane & 7u) * 16u; break;
        case 6: off = (hash32((wave * 64u + (lane >> 2)) * 2654435761u + 17u) % lines) * 128u + (hash32(wave * 64 + (lane >> 2)) & 7u) * 16u; break;
        default: off = ((hash32((wave * 64u + (lane >> 2)) * 2654435761u + 17u) % (lines - 1)) + ((lane >> 1) & 1u)) * 128u + (lane & 1u) * 16u; break;
    }
    const char* base = A.table + (size_t)blockIdx.x * A.region;
    u4 acc = {0, 0, 0, 0};
    // warm the caches with this thread's own addresses
    for (int r = 0; r < 4; r++) {
        const u4 v = *reinterpret_cast<const u4*>(base + (off + (unsigned)r * 128u * 37u) % A.region);
        acc ^= v;
    }
    __syncthreads();
    const unsigned long long t0 = wall_clock64(), c0 = __builtin_readcyclecounter();
    unsigned o = off;
    for (int r = 0; r < A.rounds; r++) {
        u4 v[8];
        const char* p[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            p[u] = base + o;
            o += 128u * 37u;
            o = o >= A.region ? o - A.region : o;
        }
        ld8<MOD>(p, v);
#pragma unroll
        for (int u = 0; u < 8; u++) acc ^= v[u];
    }
    __syncthreads();
    const unsigned long long t1 = wall_clock64(), c1 = __builtin_readcyclecounter();
    if (tid == 0) { A.out[2 * blockIdx.x] = t1 - t0; A.out[2 * blockIdx.x + 1] = c1 - c0; }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) A.sink[0] = 1;
}

int main() {
    args A{};
    const size_t maxregion = 1536u * 1024u;
    char* table;
    hipMalloc(&table, maxregion * 256);
    hipMemset(table, 1, maxregion * 256);
    hipMalloc(&A.out, 512 * 8);
    hipMalloc(&A.sink, 4);
    A.table = table;
    const int lds = 100 * 1024;
    const char* names[] = {"contiguous", "scattered", "pair32", "quad64", "quad_stride32", "oct128", "quad_same", "quad_2lines"};
    const char* mods[] = {"plain", "sc0", "sc1", "nt", "sc0 sc1", "sc1 nt"};
    const unsigned regions[] = {16u * 1024u, 96u * 1024u, 1536u * 1024u};
    typedef void (*kern)(args);
    kern ks[] = {k_ta<0>, k_ta<1>, k_ta<2>, k_ta<3>, k_ta<4>, k_ta<5>};
    for (auto k : ks) hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    printf("%-14s %-8s %9s | %12s %12s %10s\n", "pattern", "load", "region", "ns/wave-load", "clk/wave-load", "GB/s/CU");
    for (unsigned region : regions)
        for (int p = 0; p < 8; p++)
            for (int m = 0; m < 6; m++) {
                if (m > 0 && p != 1 && p != 3) continue;   // (the cache-policy bits: on the scattered and the quad pattern only)
                A.region = region; A.pattern = p; A.rounds = 64;
                for (int rep = 0; rep < 2; rep++) {
                    hipLaunchKernelGGL(ks[m], dim3(256), dim3(THREADS), lds, 0, A);
                    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
                }
                std::vector<unsigned long long> t(512);
                hipMemcpy(t.data(), A.out, 512 * 8, hipMemcpyDeviceToHost);
                double ticks = 0, clks = 0;
                for (int b = 0; b < 256; b++) { ticks += (double)t[2 * b]; clks += (double)t[2 * b + 1]; }
                ticks /= 256; clks /= 256;
                const double loads = (double)(THREADS / 64) * A.rounds * 8;   // wave-loads per CU
                printf("%-14s %-8s %7u K | %12.2f %12.1f %10.1f\n", names[p], mods[m], region >> 10, ticks * 10.0 / loads, clks / loads, 1024.0 * loads / (ticks * 10.0));
                fflush(stdout);
            }
    return 0;
}
