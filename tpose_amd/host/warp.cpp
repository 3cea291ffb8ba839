// warp -- headless counterpart of the reference's software/warp program: hierarchical two-view warp of
// a pair of stacked triangulations (software/warp/main.cpp:214-283) through the tpose:: host mirror.
//
//   warp -ia A.ppm -ib B.ppm -ta A.tri -tb B.tri [-schedule as_written|two_way]
//        [-maxframes N] [-levelframes N] [-fixedframes] [-device D] [-quiet]
//
// Per frame: doreset (count), doenergy against the OTHER image with the stored triangle colours,
// doshift; when the relative energy change falls below 1e-6 the optimised triangulation's reverse
// warp seeds the other one and the direction flips.
//   as_written: exactly one direction per level (the reference's `NWARPA < 1 && NWARPB < 1` test is
//               never true after the first convergence), then both triangulations are written to
//               <tri>.warp and the next finer level is read, warped on read.
//   two_way:    both directions per level (the README's description; `||` instead of `&&`).
//   mutual:     the two-way form with independent directions inside a phase (warp_core.hpp) -- what warp2 runs on two
//               GPUs at once; here its four descents per level run one after the other.  Same bytes.
//               -fixedframes: every descent runs exactly -levelframes frames, without the convergence test.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>

#include "tpose/io.hpp"
#include "tpose/triangulation.hpp"
#include "image_io.hpp"
#include "warp_core.hpp"

using namespace tpose;

// -schedule mutual on one GPU
static int run_mutual(const std::string& ta, const std::string& tb, long levelframes, bool fixed) {
    auto descend = [&](warpcore::direction& d) { return fixed ? warpcore::descend_fixed(d, levelframes) : warpcore::descend(d, levelframes); };
    warpcore::direction A, B;
    A.warpA = true; B.warpA = false;
    io::read(&A.tr, ta);
    io::read(&B.tr, tb);
    long frames = 0;
    int level = 0;
    while (true) {
        frames += descend(A);   // phase 1
        frames += descend(B);
        triangulation peerA, peerB;                      // the meshes handed over
        warpcore::unpack(warpcore::pack(A.tr), peerA);
        warpcore::unpack(warpcore::pack(B.tr), peerB);
        warpcore::reseed(A, peerB);
        warpcore::reseed(B, peerA);
        frames += descend(A);   // phase 2
        frames += descend(B);
        io::write(&A.tr, ta + ".warp");
        io::write(&B.tr, tb + ".warp");
        level++;
        const bool moreA = io::read(&A.tr, ta, true), moreB = io::read(&B.tr, tb, true);
        if (!moreA || !moreB) break;
    }
    std::cout << "frames " << frames << " levels " << level << std::endl;
    return 0;
}

int main(int argc, char** argv) {
    std::string ia, ib, ta, tb, schedule = "as_written";
    long maxframes = 1L << 40, levelframes = 1L << 40;
    int device = 0;
    bool quiet = false, fixed = false;
    for (int a = 1; a < argc; a++) {
        const std::string k = argv[a];
        auto val = [&]() -> const char* { if (a + 1 >= argc) { std::cerr << "missing value for " << k << "\n"; std::exit(2); } return argv[++a]; };
        if (k == "-ia") ia = val();
        else if (k == "-fixedframes") fixed = true;
        else if (k == "-ib") ib = val();
        else if (k == "-ta") ta = val();
        else if (k == "-tb") tb = val();
        else if (k == "-schedule") schedule = val();
        else if (k == "-maxframes") maxframes = std::atol(val());
        else if (k == "-levelframes") levelframes = std::atol(val());
        else if (k == "-device") device = std::atoi(val());
        else if (k == "-quiet") quiet = true;
        else { std::cerr << "unknown option " << k << "\n"; return 2; }
    }
    if (ia.empty() || ib.empty()) { std::cout << "Please specify two input images with -ia, -ib." << std::endl; return 0; }
    if (ta.empty() || tb.empty()) { std::cout << "Please specify two input triangulations with -ta, -tb." << std::endl; return 0; }
    const bool two_way = schedule == "two_way";
    Raster A, B;
    if (!load_raster(ia, A) || !load_raster(ib, B)) { std::cout << "Failed to load image." << std::endl; return 0; }
    if (A.w != B.w || A.h != B.h) { std::cout << "Images don't have the same dimension" << std::endl; return 0; }
    io::verbose = !quiet;

    RATIO = (float)A.w / (float)A.h;
    tpose::init(A.w, A.h, device);
    tpose::flavour = TP_WARP;
    tpose::image(TP_IMAGE_A, A.rgba.data(), (size_t)A.w * 4);
    tpose::image(TP_IMAGE_B, B.rgba.data(), (size_t)B.w * 4);

    if (schedule == "mutual") {
        const int rc = run_mutual(ta, tb, levelframes, fixed);
        tpose::quit();
        return rc;
    }
    triangulation trA, trB;
    io::read(&trA, ta);
    io::read(&trB, tb);
    tpose::warpA = true;
    triangulation* tr = &trA;
    tpose::upload(tr);
    tpose::doreset();

    int nwarpa = 0, nwarpb = 0, level = 0;
    long frame = 0, inlevel = 0;
    while (frame < maxframes) {
        // the reference's frames (doreset + doenergy + doshift, four read-backs, geterr) up to the one that passes the test or
        // exhausts the level's budget -- run and tested on the library's side of the boundary, read back once
        const long n = tpose::descend(tr, 1E-6, std::min(levelframes - inlevel, maxframes - frame));
        frame += n; inlevel += n;
        if (std::fabs(tpose::relerr) < 1E-6 || inlevel >= levelframes) {
            inlevel = 0;
            if (tpose::warpA) {
                trB.points = trB.originpoints;
                trA.reversewarp(trB.points);
                nwarpa++;
            } else {
                trA.points = trA.originpoints;
                trB.reversewarp(trA.points);
                nwarpb++;
            }
            tpose::warpA = !tpose::warpA;
            tr = tpose::warpA ? &trA : &trB;
            const bool more = two_way ? (nwarpa < 1 || nwarpb < 1) : (nwarpa < 1 && nwarpb < 1);
            if (more) { tpose::upload(tr); continue; }
            nwarpa = nwarpb = 0;
            io::write(&trA, ta + ".warp");
            io::write(&trB, tb + ".warp");
            level++;
            if (!io::read(&trA, ta, true)) break;
            if (!io::read(&trB, tb, true)) break;
            tpose::upload(tr);
        }
    }
    std::cout << "frames " << frame << " levels " << level << std::endl;
    tpose::quit();
    return 0;
}
