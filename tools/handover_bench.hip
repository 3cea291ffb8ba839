// tools/handover_bench.hip -- what ONE position hand-over between the patches of k_persist costs, on its own.
//
// 256 workgroups in lockstep, laid out like the patches of a plan (a 16 x 16 grid of patches, workgroup b on XCD b mod 8, every XCD one
// compact 4 x 8 region -- tp_persist.hip: patch_of_block).  Each owns 6 "vertices" and reads 13 of its 8 neighbours'.  A turn: poll the
// foreign granules {tag : 32, value : 32} for this turn's tag (agent-scope loads), wait `work` ticks of the 100 MHz clock (the grad-iter's
// own chain), post the own granules with the next tag (one sc1 store each).  period - work = the hand-over.  Knobs, each a k_persist design
// question (round 5):
//   wide      one 16-byte load per vertex instead of two 8-byte ones
//   pad       16-byte units between two vertices' granule pairs (1: packed, 8: a 128-byte line each, 16: 256 bytes)
//   depth     requests per granule in flight
//   presleep  s_sleep before the first poll of a turn (x 64 clocks)
//   local     loads of granules whose owner runs on the SAME XCD (HW_REG_XCC_ID, exchanged at start) go to that XCD's L2 (sc0) instead of
//             past it (sc1)
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/handover_bench.hip -o /tmp/hb && /tmp/hb
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned int gu32;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
#define RLX_WG __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP

#define NV 6
#define NF 13

struct args {
    unsigned long long* lbox;  // local == 2: a second mailbox of the same shape, written with PLAIN stores and polled with sc1 loads by readers on the owner's XCD
    unsigned long long* box;   // [2][256 * NV][pad * 2]
    unsigned* xcc;             // [256] XCC id of every patch's workgroup; [256]: arrival counter
    int turns, work, wide, pad, depth, presleep, local;
    unsigned long long* out;   // [256] ticks
    int* stats;                // [0] neighbour pairs on one XCD, [1] all pairs
};

__device__ __forceinline__ int patch_of_block(int b) { return (b & 7) * 32 + (b >> 3); }
// patch -> (px, py) in the 16 x 16 grid: XCD region r = patch / 32 is 4 wide x 8 tall
__device__ __forceinline__ void place(int p, int& px, int& py) { const int r = p >> 5, i = p & 31; px = (r & 3) * 4 + (i & 3); py = (r >> 2) * 8 + (i >> 2); }
__device__ __forceinline__ int patch_at(int px, int py) { return ((py >> 3) * 4 + (px >> 2)) * 32 + (py & 7) * 4 + (px & 3); }

// local == 2 (round 6, the round-5 review's item 2a): sc1 loads of a granule the owner wrote with a PLAIN store -- if an agent-scope load is served by
// the XCD's L2 and the plain store leaves the line there, a same-XCD hand-over never goes to the memory side
__device__ __forceinline__ void poll_plain_sc1(gu64* g, uint32_t tag, uint32_t& va, uint32_t& vb, unsigned long long limit_ticks, bool& timed_out) {
    const unsigned long long t0 = wall_clock64();
    for (;;) {
        const unsigned long long a = __hip_atomic_load(g, RLX_AGENT), b = __hip_atomic_load(g + 1, RLX_AGENT);
        if ((uint32_t)(a >> 32) == tag && (uint32_t)(b >> 32) == tag) { va = (uint32_t)a; vb = (uint32_t)b; return; }
        if (wall_clock64() - t0 > limit_ticks) { timed_out = true; va = vb = 0; return; }   // (the store never became visible: fall back is the caller's)
        __builtin_amdgcn_s_sleep(1);
    }
}
template <int DEPTH, int WIDE>
__device__ __forceinline__ void poll(gu64* g, uint32_t tag, bool local, uint32_t& va, uint32_t& vb) {
    unsigned long long ra[DEPTH], rb[DEPTH];
    // sc1: agent scope, past this XCD's L2 (what the compiler emits for an agent-scope atomic load); sc0: past the CU's L1 only
    for (;;) {
        unsigned long long a, b;
        if (WIDE) {
            typedef unsigned int u4 __attribute__((ext_vector_type(4)));
            u4 r;
            if (local) asm volatile("global_load_dwordx4 %0, %1, off sc0\ns_waitcnt vmcnt(0)" : "=v"(r) : "v"(g) : "memory");
            else asm volatile("global_load_dwordx4 %0, %1, off sc1\ns_waitcnt vmcnt(0)" : "=v"(r) : "v"(g) : "memory");
            a = (unsigned long long)r.x | ((unsigned long long)r.y << 32); b = (unsigned long long)r.z | ((unsigned long long)r.w << 32);
        } else if (DEPTH == 1) {
            if (local) asm volatile("global_load_dwordx2 %0, %2, off sc0\nglobal_load_dwordx2 %1, %2, off offset:8 sc0\ns_waitcnt vmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(g) : "memory");
            else { a = __hip_atomic_load(g, RLX_AGENT); b = __hip_atomic_load(g + 1, RLX_AGENT); }
        } else {
            // (the compiler's agent-scope atomic loads, DEPTH rounds in flight)
            for (int d = 0; d < DEPTH - 1; d++) { ra[d] = __hip_atomic_load(g, RLX_AGENT); rb[d] = __hip_atomic_load(g + 1, RLX_AGENT); }
            for (;;) {
                ra[DEPTH - 1] = __hip_atomic_load(g, RLX_AGENT); rb[DEPTH - 1] = __hip_atomic_load(g + 1, RLX_AGENT);
                a = ra[0]; b = rb[0];
                if ((uint32_t)(a >> 32) == tag && (uint32_t)(b >> 32) == tag) break;
                for (int d = 0; d < DEPTH - 1; d++) { ra[d] = ra[d + 1]; rb[d] = rb[d + 1]; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        if ((uint32_t)(a >> 32) == tag && (uint32_t)(b >> 32) == tag) { va = (uint32_t)a; vb = (uint32_t)b; return; }
        __builtin_amdgcn_s_sleep(1);
    }
}

template <int DEPTH, int WIDE>
__global__ __launch_bounds__(64) void k_bench(args A) {
    extern __shared__ char smem[];
    const int tid = threadIdx.x, b = blockIdx.x, p = patch_of_block(b);
    int px, py;
    place(p, px, py);
    // the 13 foreign vertices: two from each edge neighbour, one from each diagonal one, one more from the east
    const int dx[NF] = {1, 1, -1, -1, 0, 0, 0, 0, 1, 1, -1, -1, 1}, dy[NF] = {0, 0, 0, 0, 1, 1, -1, -1, 1, -1, 1, -1, 0}, kk[NF] = {0, 1, 2, 3, 4, 5, 0, 1, 2, 3, 4, 5, 2};
    unsigned my_xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(my_xcc));
    my_xcc &= 15u;
    if (tid == 0) {
        __hip_atomic_store((gu32*)A.xcc + p, my_xcc | 0x100u, RLX_AGENT);
        __hip_atomic_fetch_add((gu32*)A.xcc + 256, 1u, RLX_AGENT);
        while (__hip_atomic_load((gu32*)A.xcc + 256, RLX_AGENT) < gridDim.x) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
    int q = -1, k = 0;
    bool local = false;
    if (tid < NF) {
        const int qx = px + dx[tid], qy = py + dy[tid];
        if (qx >= 0 && qx < 16 && qy >= 0 && qy < 16) {
            q = patch_at(qx, qy); k = kk[tid];
            const unsigned oxcc = __hip_atomic_load((gu32*)A.xcc + q, RLX_AGENT) & 15u;
            if (oxcc == my_xcc) atomicAdd(&A.stats[0], 1);
            atomicAdd(&A.stats[1], 1);
            local = A.local && oxcc == my_xcc;
        }
    }
    gu64* box = (gu64*)A.box;
    const size_t stride = (size_t)256 * NV * A.pad * 2;
    const unsigned long long t_start = wall_clock64();
    uint32_t acc = 0;
    for (int e = 1; e <= A.turns; e++) {
        const uint32_t tag = 0x80000000u | (uint32_t)e;
        if (e > 1 && q >= 0) {
            for (int z = 0; z < A.presleep; z++) __builtin_amdgcn_s_sleep(1);
            uint32_t va, vb;
            const size_t at = (size_t)(e & 1) * stride + ((size_t)q * NV + k) * A.pad * 2;
            if (A.local == 2 && local) {
                bool timed_out = false;
                poll_plain_sc1((gu64*)A.lbox + at, tag, va, vb, 2000ull, timed_out);   // (20 us: then the global mailbox)
                if (timed_out) { atomicAdd(&A.stats[2], 1); poll<DEPTH, WIDE>(box + at, tag, false, va, vb); }
            } else poll<DEPTH, WIDE>(box + at, tag, A.local == 1 && local, va, vb);
            acc += va + vb;
        }
        __syncthreads();
        // the grad-iter's own chain
        if (A.work) { const unsigned long long t0 = wall_clock64(); while (wall_clock64() - t0 < (unsigned long long)A.work) __builtin_amdgcn_s_sleep(1); }
        if (tid < 2 * NV) {
            const unsigned long long w = ((unsigned long long)(0x80000000u | (uint32_t)(e + 1)) << 32) | (uint32_t)(acc + e);
            const size_t at = (size_t)((e + 1) & 1) * stride + ((size_t)p * NV + (tid >> 1)) * A.pad * 2 + (tid & 1);
            if (A.local == 2) A.lbox[at] = w;   // (a plain store: it stays in this XCD's L2)
            __hip_atomic_store(box + at, w, RLX_AGENT);
        }
    }
    if (tid == 0) A.out[b] = wall_clock64() - t_start;
}

template <int DEPTH, int WIDE>
static double run(args A, int lds) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_bench<DEPTH, WIDE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipMemset(A.box, 0, (size_t)2 * 256 * NV * A.pad * 2 * 8);
    hipMemset(A.xcc, 0, 257 * 4);
    hipMemset(A.stats, 0, 12);
    hipMemset(A.lbox, 0, (size_t)2 * 256 * NV * A.pad * 2 * 8);
    hipLaunchKernelGGL((k_bench<DEPTH, WIDE>), dim3(256), dim3(64), lds, 0, A);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); exit(1); }
    std::vector<unsigned long long> t(256);
    hipMemcpy(t.data(), A.out, 256 * 8, hipMemcpyDeviceToHost);
    unsigned long long mx = 0;
    for (auto v : t) mx = v > mx ? v : mx;
    return (double)mx / 100.0 / A.turns;   // us per turn
}

int main(int argc, char** argv) {
    args A{};
    const int maxpad = 64;
    hipMalloc(&A.box, (size_t)2 * 256 * NV * maxpad * 2 * 8);
    hipMalloc(&A.xcc, 257 * 4);
    hipMalloc(&A.out, 256 * 8);
    hipMalloc(&A.stats, 12);
    hipMalloc(&A.lbox, (size_t)2 * 256 * NV * maxpad * 2 * 8);
    A.turns = 20000;
    const int lds = 100 * 1024;   // one workgroup per CU
    struct cfg { int work, wide, pad, depth, presleep, local; };
    const cfg C[] = {
        {0, 0, 1, 1, 0, 0}, {380, 0, 1, 1, 0, 0},                    // as k_persist polls today: bare and with a 3.8 us chain
        {380, 1, 1, 1, 0, 0},                                         // one 16-byte load per vertex
        {380, 0, 8, 1, 0, 0}, {380, 0, 16, 1, 0, 0}, {380, 1, 16, 1, 0, 0}, {380, 1, 64, 1, 0, 0},   // a line / 256 bytes / 1 KB per vertex
        {380, 0, 1, 2, 0, 0}, {380, 0, 1, 4, 0, 0},                  // requests in flight
        {380, 0, 1, 1, 2, 0}, {380, 0, 1, 1, 4, 0}, {380, 0, 1, 1, 8, 0}, {380, 1, 1, 1, 4, 0},     // wait before the first poll
        // (local == 1, round 5: sc0 loads of the global mailbox for same-XCD granules -- served by the CU's own L1, they never see the post: the run hangs; left out)
        {380, 0, 1, 1, 0, 2}, {0, 0, 1, 1, 0, 2}, {380, 0, 16, 1, 0, 2},   // same-XCD granules: plain store + sc1 loads of an XCD-local copy
        {380, 0, 1, 1, 0, 0},
    };
    printf("work wide pad depth presleep local | period us | hand-over us\n");
    for (const cfg& c : C) {
        A.work = c.work; A.wide = c.wide; A.pad = c.pad; A.depth = c.depth; A.presleep = c.presleep; A.local = c.local;
        double us;
        if (c.wide) us = run<1, 1>(A, lds);
        else if (c.depth == 1) us = run<1, 0>(A, lds);
        else if (c.depth == 2) us = run<2, 0>(A, lds);
        else us = run<4, 0>(A, lds);
        int st[3];
        hipMemcpy(st, A.stats, 12, hipMemcpyDeviceToHost);
        printf("%4d %4d %3d %5d %8d %5d | %9.3f | %6.3f   (neighbour reads on the owner's XCD: %d of %d; local polls that timed out: %d)\n", c.work, c.wide, c.pad, c.depth, c.presleep, c.local, us,
               us - c.work / 100.0, st[0], st[1], st[2]);
        fflush(stdout);
    }
    return 0;
}
