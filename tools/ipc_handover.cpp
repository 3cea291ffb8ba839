// Row e3: the price of a device-side hand-over BETWEEN TWO PROCESSES through a buffer exported with
// hipIpcGetMemHandle -- what a band split of one direction over two GPUs would pay per grad-iter at the seam.  Two processes,
//   ipc_handover a <file>     allocates the mailbox, writes its IPC handle to <file>, plays "ping"
//   ipc_handover b <file>     opens the handle, plays "pong"
// each with ONE workgroup resident: A stores tag i (one 8-byte granule {tag, payload}, system scope), B polls it and answers with
// its own granule, N rounds inside one launch; the round trip is timed with the wall clock on the device.  On a one-GPU box both
// processes share device 0: a proxy for the latency of the mechanism (same HBM, no xGMI hop), a LOWER bound for two GPUs.
// Every wait is bounded (2 s): a peer that never shows up ends the kernel with a failure code instead of hanging the GPU.
//   hipcc --offload-arch=gfx950 -O2 -o ipc_handover tools/ipc_handover.cpp
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
typedef __attribute__((address_space(1))) unsigned long long gu64;
#define SYS __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM

// box[0]: A's granule, box[8]: B's granule (separate 64-byte lines); payload words ride in the low half
__global__ void k_pingpong(unsigned long long* box, int me, int rounds, int payload_granules, unsigned long long* out) {
    gu64* mine = (gu64*)(box + (me ? 64 : 0));
    gu64* theirs = (gu64*)(box + (me ? 0 : 64));
    const int lane = threadIdx.x;
    const unsigned long long limit = 200000000ull;   // 2 s of the 100 MHz clock
    unsigned long long t0 = 0, t1 = 0;
    int failed = 0;
    for (int i = 1; i <= rounds && !failed; i++) {
        if (i == 17 && lane == 0) t0 = wall_clock64();   // (the first rounds include the peer's start-up)
        if (me == 0) {
            if (lane < payload_granules) __hip_atomic_store(mine + lane, ((unsigned long long)i << 32) | (unsigned)lane, SYS);
            if (lane < payload_granules) {
                const unsigned long long s = wall_clock64();
                while ((unsigned)(__hip_atomic_load(theirs + lane, SYS) >> 32) < (unsigned)i)
                    if (wall_clock64() - s > limit) { failed = 1; break; }
            }
        } else {
            if (lane < payload_granules) {
                const unsigned long long s = wall_clock64();
                while ((unsigned)(__hip_atomic_load(theirs + lane, SYS) >> 32) < (unsigned)i)
                    if (wall_clock64() - s > limit) { failed = 1; break; }
            }
            failed = __any(failed);
            if (!failed && lane < payload_granules) __hip_atomic_store(mine + lane, ((unsigned long long)i << 32) | (unsigned)lane, SYS);
        }
        failed = __any(failed);
    }
    if (lane == 0) { t1 = wall_clock64(); out[0] = failed ? 0ull : t1 - t0; out[1] = (unsigned long long)failed; }
}

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: ipc_handover a|b <file> [rounds]\n"); return 2; }
    const int me = argv[1][0] == 'b';
    const int rounds = argc > 3 ? atoi(argv[3]) : 20000;
    CHECK(hipSetDevice(0));
    unsigned long long* box = nullptr;
    if (!me) {
        CHECK(hipMalloc((void**)&box, 4096));
        CHECK(hipMemset(box, 0, 4096));
        CHECK(hipDeviceSynchronize());
        hipIpcMemHandle_t h;
        CHECK(hipIpcGetMemHandle(&h, box));
        const std::string tmp = std::string(argv[2]) + ".tmp";
        FILE* f = fopen(tmp.c_str(), "wb");
        if (!f || fwrite(&h, sizeof h, 1, f) != 1) { fprintf(stderr, "cannot write %s\n", tmp.c_str()); return 2; }
        fclose(f);
        rename(tmp.c_str(), argv[2]);
    } else {
        hipIpcMemHandle_t h;
        FILE* f = nullptr;
        for (int tries = 0; tries < 2000 && !(f = fopen(argv[2], "rb")); tries++) std::this_thread::sleep_for(std::chrono::milliseconds(5));
        if (!f || fread(&h, sizeof h, 1, f) != 1) { fprintf(stderr, "no handle in %s\n", argv[2]); return 2; }
        fclose(f);
        CHECK(hipIpcOpenMemHandle((void**)&box, h, hipIpcMemLazyEnablePeerAccess));
    }
    unsigned long long* out = nullptr;
    CHECK(hipHostMalloc((void**)&out, 64, hipHostMallocDefault));
    for (int granules : {1, 2, 16, 64}) {   // 8 B, 16 B (one vertex position), 128 B, 512 B per hand-over
        out[0] = out[1] = 0;
        // (both sides restart their tags: the box is cleared by A between the runs, after both kernels have ended -- the
        // file system is the barrier: B waits for A's "go" file of this run)
        const std::string go = std::string(argv[2]) + ".go" + std::to_string(granules);
        if (!me) {
            CHECK(hipMemset(box, 0, 4096));
            CHECK(hipDeviceSynchronize());
            FILE* g = fopen(go.c_str(), "wb"); if (g) fclose(g);
        } else {
            FILE* g = nullptr;
            for (int tries = 0; tries < 4000 && !(g = fopen(go.c_str(), "rb")); tries++) std::this_thread::sleep_for(std::chrono::milliseconds(1));
            if (!g) { fprintf(stderr, "peer never started run %d\n", granules); return 2; }
            fclose(g);
        }
        hipLaunchKernelGGL(k_pingpong, dim3(1), dim3(64), 0, 0, box, me, rounds, granules, out);
        CHECK(hipDeviceSynchronize());
        if (out[1]) { printf("{\"role\": \"%s\", \"granules\": %d, \"failed\": true}\n", me ? "b" : "a", granules); continue; }
        const double rt_us = (double)out[0] / 100.0 / (double)(rounds - 16);
        printf("{\"role\": \"%s\", \"granules_of_8_bytes\": %d, \"rounds\": %d, \"round_trip_us\": %.3f, \"one_way_us\": %.3f}\n",
               me ? "b" : "a", granules, rounds - 16, rt_us, rt_us / 2);
        fflush(stdout);
        std::this_thread::sleep_for(std::chrono::milliseconds(200));   // (the slower side ends before the box is cleared)
    }
    return 0;
}
