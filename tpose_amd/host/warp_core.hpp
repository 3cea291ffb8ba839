// warp_core.hpp -- what the warp harnesses share: one descent of one direction of the two-view warp, and the
// "mutual" two-way schedule the two-GPU driver (warp2.cpp) runs concurrently.
//
// Reference: software/warp/main.cpp:214-283.  There, one direction descends per frame until the relative energy
// change of consecutive frames falls below 1e-6 (:231, tested EVERY frame), then the other triangulation is re-seeded
// from the reverse warp (source/triangulation.hpp:492-520) and the direction flips (:235-254).  The README
// (README.md:49-53) describes the two-way consistent form: the initial condition of a warping is the reverse warping
// of the other image's triangulation.  "mutual" is that form with the two directions independent inside a phase, so
// that they can run on two GPUs at once; per hierarchy level
//     phase 1   T(A) descends against image B, T(B) against image A, each from its level state
//     exchange  each side takes the other's descended mesh {triangles, points, originpoints}
//     re-seed   own points = own originpoints pulled through the other's REVERSE warp (two-way consistency)
//     phase 2   both descend again from the seeds
//     write     one record each to <tri>.warp; the next finer level is read warped-on-read (source/io.hpp:139)
// On one GPU (warp -schedule mutual) the four descents of a level run one after the other; on two GPUs
// (warp2) the two of a phase run concurrently and the exchange goes over RCCL.  Same arithmetic, same bytes.
#pragma once

#include <cstdint>
#include <cstring>
#include <vector>

#include "tpose/io.hpp"
#include "tpose/triangulation.hpp"

namespace warpcore {

struct direction {
    tpose::triangulation tr;
    bool warpA = true;     // true: T(A) against image B
    float toterr = 1.0f;   // the reference's global of that name, one per direction (source/triangulation.hpp:648)
};

// descend until converged (geterr < 1e-6 on consecutive frames, software/warp/main.cpp:231) or `levelframes` frames
inline long descend(direction& d, long levelframes) {
    tpose::warpA = d.warpA;
    tpose::toterr = d.toterr;
    tpose::upload(&d.tr);
    // doreset + doenergy + doshift of the reference's frame and its test, frame after frame on the library's side
    const long frames = tpose::descend(&d.tr, 1E-6, levelframes);
    d.toterr = tpose::toterr;
    return frames;
}

// exactly `frames` frames, no convergence test (-fixedframes: a fixed budget per descent, e.g. for descents split over
// bands of GPUs, whose launches carry no per-frame energies): upload, one enqueue, one read-back
inline long descend_fixed(direction& d, long frames) {
    tpose::warpA = d.warpA;
    tpose::upload(&d.tr);
    tpose::doframes(frames);
    tpose::retrieve(&d.tr);
    return frames;
}

// the mesh a side hands over: {NT, NP, triangles ivec4[NT], points vec2[NP], originpoints vec2[NP]}
inline std::vector<int32_t> pack(const tpose::triangulation& t) {
    std::vector<int32_t> b(2 + 4 * (size_t)t.NT + 4 * (size_t)t.NP);
    b[0] = t.NT; b[1] = t.NP;
    std::memcpy(&b[2], (const void*)t.triangles.data(), sizeof(int32_t) * 4 * (size_t)t.NT);
    std::memcpy(&b[2 + 4 * (size_t)t.NT], (const void*)t.points.data(), sizeof(float) * 2 * (size_t)t.NP);
    std::memcpy(&b[2 + 4 * (size_t)t.NT + 2 * (size_t)t.NP], (const void*)t.originpoints.data(), sizeof(float) * 2 * (size_t)t.NP);
    return b;
}
inline void unpack(const std::vector<int32_t>& b, tpose::triangulation& t) {
    t.NT = b[0]; t.NP = b[1];
    t.triangles.resize(t.NT); t.points.resize(t.NP); t.originpoints.resize(t.NP);
    t.halfedges.assign(3 * (size_t)t.NT, -1);  // reversewarp never looks at them
    std::memcpy((void*)t.triangles.data(), &b[2], sizeof(int32_t) * 4 * (size_t)t.NT);
    std::memcpy((void*)t.points.data(), &b[2 + 4 * (size_t)t.NT], sizeof(float) * 2 * (size_t)t.NP);
    std::memcpy((void*)t.originpoints.data(), &b[2 + 4 * (size_t)t.NT + 2 * (size_t)t.NP], sizeof(float) * 2 * (size_t)t.NP);
}

// own points = own originpoints pulled through the peer's reverse warp
inline void reseed(direction& mine, tpose::triangulation& peer) {
    mine.tr.points = mine.tr.originpoints;
    peer.reversewarp(mine.tr.points);
}

}  // namespace warpcore
