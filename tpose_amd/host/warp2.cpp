// warp2 -- the two-way consistent hierarchical warp on TWO GPUs, one image (one direction) per GPU, vertex buffers
// exchanged with RCCL over xGMI.  C++ host code over the C ABI (include/tpose_hip.h); no Python, no torch.
//
//   rank 0:  warp2 -rank 0 -idfile F -ia A.ppm -ib B.ppm -ta A.tri -tb B.tri [-levelframes N] [-device D]
//   rank 1:  warp2 -rank 1 -idfile F ...                       (same arguments; one process per GPU)
//
// Rank 0 holds raster B and descends T(A) against it; rank 1 holds raster A and descends T(B) (the reference runs the
// two directions one after the other on one GPU, software/warp/main.cpp:214-283).  Schedule "mutual" of warp_core.hpp:
// per level and phase the two descents run concurrently; between the phases each rank sends its descended mesh
// {NT, NP, triangles, points, originpoints} (tens of KB, latency-bound) and receives the peer's -- ONE grouped
// ncclSend / ncclRecv pair -- re-seeds its own points through the peer's reverse warp
// (source/triangulation.hpp:492-520), descends again and appends its level to <tri>.warp.  The output is byte-identical
// to `warp -schedule mutual` on one GPU.
//
//   -transport rccl   (default) ncclSend / ncclRecv between the two GPUs; the unique id travels through -idfile, tagged with
//                     -nonce N (the same number for both ranks, e.g. the launcher's pid) so that a file left by an earlier
//                     run is never taken for this run's id
//   -transport fifo   two named pipes <idfile>.0to1 / <idfile>.1to0 -- for the CPU test of the schedule (built with
//                     -DWARP2_NO_RCCL against the oracle-backed C ABI) and for one-GPU boxes, where RCCL refuses
//                     two ranks on the same device
//   -bands 2 -band b   every direction is split over TWO processes / GPUs: band b of a direction runs half of
//                     the patches of each descent (tp_band_attach) and hands vertex positions over to its band-mate on the
//                     device, through mailboxes mapped into each other's process (hipIpcGetMemHandle; the handles travel
//                     through the pipes <idfile>.mate<rank>.*).  Four processes: -rank {0,1} x -band {0,1}; band b of one
//                     direction exchanges meshes with band b of the other (its own -idfile: <idfile>.b<band>), so both
//                     bands hold the same meshes at every step.  Band 0 writes <tri>.warp, band 1 <tri>.warp.band1 (the
//                     same bytes).  The convergence test of a descent runs on every band's host over all bands' per-frame
//                     energies (they travel like the positions).  -bandpatches N: patches per band (default: one per compute unit; two bands sharing ONE
//                     device -- the one-GPU test box -- must not ask for more than the device has between them)
//   -fixedframes      every descent runs exactly -levelframes frames (no convergence test)
//   -selftest         one rank sends a mesh-sized buffer to itself through RCCL and checks it (exercises the library,
//                     the stream and the device buffers on a one-GPU box)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fcntl.h>
#include <iostream>
#include <string>
#include <sys/file.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>

#include "tpose/io.hpp"
#include "tpose/triangulation.hpp"
#include "image_io.hpp"
#include "warp_core.hpp"

#ifndef WARP2_NO_RCCL
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#endif

using namespace tpose;

namespace {
long long g_nonce = 0;   // -nonce: names this run in the id file

[[noreturn]] void die(const std::string& what) {
    std::cerr << "warp2: " << what << std::endl;
    std::exit(1);
}

// send `out` to the peer and receive its message
struct transport {
    virtual ~transport() {}
    virtual std::vector<int32_t> exchange(const std::vector<int32_t>& out) = 0;
};

// ---- named pipes ---------------------------------------------------------------------------------------------
struct fifo_transport : transport {
    int rank, wr = -1, rd = -1;
    fifo_transport(int rank_, const std::string& base) : rank(rank_) {
        const std::string p01 = base + ".0to1", p10 = base + ".1to0";
        mkfifo(p01.c_str(), 0600);  // either rank may come first
        mkfifo(p10.c_str(), 0600);
        if (rank == 0) { wr = open(p01.c_str(), O_WRONLY); rd = open(p10.c_str(), O_RDONLY); }
        else { rd = open(p01.c_str(), O_RDONLY); wr = open(p10.c_str(), O_WRONLY); }
        if (wr < 0 || rd < 0) die("cannot open the pipes " + base + ".*");
    }
    ~fifo_transport() override { if (wr >= 0) close(wr); if (rd >= 0) close(rd); }
    void put(const void* p, size_t n) { const char* c = (const char*)p; while (n) { ssize_t k = write(wr, c, n); if (k <= 0) die("pipe write"); c += k; n -= (size_t)k; } }
    void get(void* p, size_t n) { char* c = (char*)p; while (n) { ssize_t k = read(rd, c, n); if (k <= 0) die("pipe read"); c += k; n -= (size_t)k; } }
    std::vector<int32_t> exchange(const std::vector<int32_t>& out) override {
        uint64_t n_out = out.size(), n_in = 0;
        std::vector<int32_t> in;
        if (rank == 0) {  // rank 0 talks first: messages may exceed the pipe buffer
            put(&n_out, sizeof n_out); put(out.data(), n_out * sizeof(int32_t));
            get(&n_in, sizeof n_in); in.resize(n_in); get(in.data(), n_in * sizeof(int32_t));
        } else {
            get(&n_in, sizeof n_in); in.resize(n_in); get(in.data(), n_in * sizeof(int32_t));
            put(&n_out, sizeof n_out); put(out.data(), n_out * sizeof(int32_t));
        }
        return in;
    }
};

#ifndef WARP2_NO_RCCL
#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) die(std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)
#define NCCLCHECK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) die(std::string(#x) + ": " + ncclGetErrorString(r_)); } while (0)

// ---- RCCL: one grouped send/recv per message, device buffers on this rank's GPU -------------------------------
struct rccl_transport : transport {
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    int rank, nranks, peer;
    int32_t *dsend = nullptr, *drecv = nullptr;
    size_t cap = 0;
    rccl_transport(int rank_, int nranks_, int device, const std::string& idfile) : rank(rank_), nranks(nranks_), peer(nranks_ == 1 ? 0 : 1 - rank_) {
        HIPCHECK(hipSetDevice(device));
        ncclUniqueId id;
        // the id file carries a nonce both ranks were started with (-nonce, default 0): a file left by an earlier run is not
        // this run's id (ncclCommInitRank with a stale id never returns)
        if (rank == 0) {
            std::remove(idfile.c_str());
            NCCLCHECK(ncclGetUniqueId(&id));
            const std::string tmp = idfile + ".tmp";
            FILE* f = std::fopen(tmp.c_str(), "wb");
            if (!f || std::fwrite(&g_nonce, sizeof g_nonce, 1, f) != 1 || std::fwrite(&id, sizeof id, 1, f) != 1) die("cannot write " + tmp);
            std::fclose(f);
            if (std::rename(tmp.c_str(), idfile.c_str()) != 0) die("cannot publish " + idfile);
        } else {
            bool got = false;
            for (int tries = 0; tries < 3000 && !got; tries++) {
                if (FILE* f = std::fopen(idfile.c_str(), "rb")) {
                    long long n = 0;
                    got = std::fread(&n, sizeof n, 1, f) == 1 && n == g_nonce && std::fread(&id, sizeof id, 1, f) == 1;
                    std::fclose(f);
                }
                if (!got) std::this_thread::sleep_for(std::chrono::milliseconds(10));
            }
            if (!got) die("no RCCL id of this run (nonce) in " + idfile);
        }
        NCCLCHECK(ncclCommInitRank(&comm, nranks, id, rank));
        if (rank == 0) std::remove(idfile.c_str());   // (every rank has read it: the communicator exists)
        HIPCHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    }
    ~rccl_transport() override {
        if (comm) ncclCommDestroy(comm);
        if (dsend) (void)hipFree(dsend);
        if (drecv) (void)hipFree(drecv);
        if (stream) (void)hipStreamDestroy(stream);
    }
    void reserve(size_t n) {
        if (n <= cap) return;
        if (dsend) HIPCHECK(hipFree(dsend));
        if (drecv) HIPCHECK(hipFree(drecv));
        cap = n + n / 2 + 1024;
        HIPCHECK(hipMalloc((void**)&dsend, cap * sizeof(int32_t)));
        HIPCHECK(hipMalloc((void**)&drecv, cap * sizeof(int32_t)));
    }
    void swap_device(size_t n_send, size_t n_recv) {  // ONE grouped send / recv
        NCCLCHECK(ncclGroupStart());
        NCCLCHECK(ncclSend(dsend, n_send, ncclInt32, peer, comm, stream));
        NCCLCHECK(ncclRecv(drecv, n_recv, ncclInt32, peer, comm, stream));
        NCCLCHECK(ncclGroupEnd());
        HIPCHECK(hipStreamSynchronize(stream));
    }
    std::vector<int32_t> exchange(const std::vector<int32_t>& out) override {
        // sizes first (two int32 words), then the payload
        reserve(out.size() + 2);
        int32_t n_out[2] = {(int32_t)out.size(), 0}, n_in[2] = {0, 0};
        HIPCHECK(hipMemcpyAsync(dsend, n_out, sizeof n_out, hipMemcpyHostToDevice, stream));
        swap_device(2, 2);
        HIPCHECK(hipMemcpy(n_in, drecv, sizeof n_in, hipMemcpyDeviceToHost));
        reserve((size_t)n_in[0] + 2);
        HIPCHECK(hipMemcpyAsync(dsend, out.data(), out.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream));
        swap_device(out.size(), (size_t)n_in[0]);
        std::vector<int32_t> in((size_t)n_in[0]);
        HIPCHECK(hipMemcpy(in.data(), drecv, in.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
        return in;
    }
};
#endif

}  // namespace

int main(int argc, char** argv) {
    std::string ia, ib, ta, tb, idfile, how = "rccl";
    long levelframes = 1L << 40;
    int device = -1, rank = -1, bands = 1, band = 0, bandpatches = 0;
    bool quiet = false, selftest = false, fixed = false, coarse = false;
    for (int a = 1; a < argc; a++) {
        const std::string k = argv[a];
        auto val = [&]() -> const char* { if (a + 1 >= argc) die("missing value for " + k); return argv[++a]; };
        if (k == "-ia") ia = val();
        else if (k == "-ib") ib = val();
        else if (k == "-ta") ta = val();
        else if (k == "-tb") tb = val();
        else if (k == "-rank") rank = std::atoi(val());
        else if (k == "-idfile") idfile = val();
        else if (k == "-nonce") g_nonce = std::atoll(val());
        else if (k == "-transport") how = val();
        else if (k == "-levelframes") levelframes = std::atol(val());
        else if (k == "-device") device = std::atoi(val());
        else if (k == "-quiet") quiet = true;
        else if (k == "-bands") bands = std::atoi(val());
        else if (k == "-band") band = std::atoi(val());
        else if (k == "-bandpatches") bandpatches = std::atoi(val());
        else if (k == "-fixedframes") fixed = true;
        else if (k == "-coarse") coarse = true;
        else if (k == "-selftest") selftest = true;
        else die("unknown option " + k);
    }
    if (selftest) {
#ifndef WARP2_NO_RCCL
        if (idfile.empty()) die("-selftest needs -idfile");
        rccl_transport t(0, 1, device < 0 ? 0 : device, idfile);
        std::vector<int32_t> msg(20000);
        for (size_t i = 0; i < msg.size(); i++) msg[i] = (int32_t)(i * 2654435761u);
        const std::vector<int32_t> back = t.exchange(msg);
        if (back != msg) die("RCCL self exchange returned different data");
        int ver = 0;
        ncclGetVersion(&ver);
        std::cout << "RCCL self exchange OK (" << msg.size() * 4 << " bytes, library version " << ver << ")" << std::endl;
        // what one hand-over costs when nothing else is in the way: grouped send + recv of a seam-sized message that is
        // already on the device, stream synchronised (a lower bound for a second GPU across xGMI)
        for (size_t words : {(size_t)16, (size_t)16384, (size_t)475000}) {
            t.reserve(words);
            for (int k = 0; k < 20; k++) t.swap_device(words, words);
            const auto t0 = std::chrono::steady_clock::now();
            const int reps = 200;
            for (int k = 0; k < reps; k++) t.swap_device(words, words);
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
            std::cout << "RCCL grouped send+recv to self, " << words * 4 << " bytes: " << us << " us per exchange" << std::endl;
        }
        return 0;
#else
        die("built without RCCL");
#endif
    }
    if (rank != 0 && rank != 1) die("-rank 0 or -rank 1");
    if (ia.empty() || ib.empty() || ta.empty() || tb.empty() || idfile.empty()) die("need -ia -ib -ta -tb -idfile");
    if (bands != 1 && bands != 2) die("-bands 1 or 2");
    if (band < 0 || band >= bands) die("-band out of range");
    if (bands == 2) idfile += ".b" + std::to_string(band);   // band b of one direction talks to band b of the other
    if (device < 0) device = rank * bands + band;  // one process per GPU
    Raster A, B;
    if (!load_raster(ia, A) || !load_raster(ib, B)) die("failed to load the images");
    if (A.w != B.w || A.h != B.h) die("images do not have the same dimension");
    io::verbose = !quiet;

    transport* link = nullptr;
    if (how == "fifo") link = new fifo_transport(rank, idfile);
#ifndef WARP2_NO_RCCL
    else if (how == "rccl") link = new rccl_transport(rank, 2, device, idfile);
#endif
    else die("unknown transport " + how);

    RATIO = (float)A.w / (float)A.h;
    tpose::init(A.w, A.h, device);
    tpose::flavour = TP_WARP;
    // one image per GPU: the raster this rank's direction sweeps (rank 0: T(A) against B)
    if (rank == 0) tpose::image(TP_IMAGE_B, B.rgba.data(), (size_t)B.w * 4);
    else tpose::image(TP_IMAGE_A, A.rgba.data(), (size_t)A.w * 4);

    warpcore::direction mine;
    mine.warpA = rank == 0;
    const std::string my_tri = rank == 0 ? ta : tb;
    io::read(&mine.tr, my_tri);
    void* boxes[2] = {nullptr, nullptr};
    if (bands == 2) {
        // the two bands of this direction: one mailbox each, mapped into the other's process
        const std::string mate_base = idfile.substr(0, idfile.size() - 3) + ".mate" + std::to_string(rank);
        fifo_transport mate(band, mate_base);
        const int cap_points = 1 << 15, cap_tris = 1 << 16;   // (what the mailboxes and their rings are sized for)
        const size_t bytes = tp_band_mailbox_bytes(cap_points, cap_tris);
#ifndef WARP2_NO_RCCL
        HIPCHECK(hipSetDevice(device));
        // fine-grained device memory: the mate's GPU writes positions into it while this band's kernel polls it (tpose_hip.h:
        // tp_band_attach, MEMORY TYPE).  -coarse keeps the round-3 allocation (plain hipMalloc) for A/B timing on one device.
        hipIpcMemHandle_t h;
        bool fine = !coarse;
        if (fine && tp_band_mailbox_alloc(tpose::ctx, bytes, &boxes[band]) != TP_OK) die(std::string("tp_band_mailbox_alloc: ") + tp_last_error(tpose::ctx));
        if (fine && hipIpcGetMemHandle(&h, boxes[band]) != hipSuccess) {
            (void)hipGetLastError();
            std::cerr << "warp2: fine-grained memory cannot be shared between processes on this system; falling back to hipMalloc "
                         "(bands on DIFFERENT devices will time out in every launch)" << std::endl;
            tp_band_mailbox_free(tpose::ctx, boxes[band]); boxes[band] = nullptr; fine = false;
        }
        if (!fine) {
            HIPCHECK(hipMalloc(&boxes[band], bytes));
            HIPCHECK(hipMemset(boxes[band], 0, bytes));
            HIPCHECK(hipDeviceSynchronize());
            HIPCHECK(hipIpcGetMemHandle(&h, boxes[band]));
        }
        if (!quiet) std::cout << "band " << band << " of rank " << rank << ": mailbox in " << (fine ? "fine-grained" : "coarse-grained") << " device memory" << std::endl;
        std::vector<int32_t> out(sizeof h / 4);
        std::memcpy(out.data(), &h, sizeof h);
        const std::vector<int32_t> in = mate.exchange(out);
        if (in.size() != out.size()) die("band-mate sent no mailbox handle");
        std::memcpy(&h, in.data(), sizeof h);
        HIPCHECK(hipIpcOpenMemHandle(&boxes[1 - band], h, hipIpcMemLazyEnablePeerAccess));
#else
        static char dummy[2][128];   // (the CPU stand-in of the C ABI runs every band's descents whole)
        boxes[0] = dummy[0]; boxes[1] = dummy[1];
        (void)mate.exchange(std::vector<int32_t>(1, band));
#endif
        if (tp_band_attach(tpose::ctx, band, 2, boxes, bytes, cap_points, cap_tris, bandpatches) != TP_OK) die(std::string("tp_band_attach: ") + tp_last_error(tpose::ctx));
        // the census of resident workgroups and the first plan, while no band spins in a launch yet (two bands may share a device)
        tpose::warpA = mine.warpA;
        tpose::upload(&mine.tr);
        tp_params p0;
        tp_default_params(TP_WARP, &p0);
        p0.image_slot = mine.warpA ? TP_IMAGE_B : TP_IMAGE_A;
        {   // (one process at a time: a census shares its device with nobody -- the four may have been given the same one)
            const std::string lock_name = idfile.substr(0, idfile.size() - 3) + ".census";
            const int lock = open(lock_name.c_str(), O_CREAT | O_RDWR, 0600);
            if (lock >= 0) flock(lock, LOCK_EX);
            const int rc = tp_prepare(tpose::ctx, &p0);
            if (rc == TP_OK) tp_synchronize(tpose::ctx);
            if (lock >= 0) { flock(lock, LOCK_UN); close(lock); }
            if (rc != TP_OK) die(std::string("tp_prepare: ") + tp_last_error(tpose::ctx));
        }
        (void)mate.exchange(std::vector<int32_t>(1, band));   // both are ready
    }
    auto descend = [&](warpcore::direction& d) { return fixed ? warpcore::descend_fixed(d, levelframes) : warpcore::descend(d, levelframes); };
    const std::string out_name = my_tri + ".warp" + (band ? ".band1" : "");
    long frames = 0;
    int level = 0;
    while (true) {
        frames += descend(mine);                                             // phase 1
        triangulation peer;
        warpcore::unpack(link->exchange(warpcore::pack(mine.tr)), peer);    // hand-over
        warpcore::reseed(mine, peer);
        frames += descend(mine);                                             // phase 2
        io::write(&mine.tr, out_name);
        level++;
        const int32_t more = io::read(&mine.tr, my_tri, true) ? 1 : 0;
        const std::vector<int32_t> theirs = link->exchange(std::vector<int32_t>(1, more));
        if (!more || theirs.empty() || !theirs[0]) break;  // stacks may differ in depth: stop together
    }
    std::cout << "rank " << rank << " frames " << frames << " levels " << level << std::endl;
    if (bands == 2) {
        int64_t launches = 0, failures = 0, patches = 0;
        int64_t census = 0;
        tp_get_info(tpose::ctx, 5, &launches); tp_get_info(tpose::ctx, 9, &failures); tp_get_info(tpose::ctx, 2, &patches); tp_get_info(tpose::ctx, 7, &census);
        std::cout << "band " << band << " of rank " << rank << ": persistent launches " << launches << ", patches of the plan " << patches
                  << ", launches given up " << failures << " (census " << census << ")" << std::endl;
    }
    delete link;
    tpose::quit();
    return 0;
}
