// tpose/triangulation.hpp -- host mirror of the reference's `tpose::triangulation` API, with the GPU
// glue re-targeted from TinyEngine/OpenGL SSBOs to the C ABI of include/tpose_hip.h.
//
// Kept verbatim from the reference (weigert/t-pose, source/triangulation.hpp:24-95): the struct's
// members (NT, triangles, halfedges, colors, NP, points, originpoints, in/out streams, MAXT) and the
// method names/meaning (angle, hlength, boundary, eraset, erasep, prune, flip, collapse, split,
// optimize, warp, reversewarp); and the free functions upload / geterr / gettoterr / maxerrid with the
// globals terr, perr, cn, col, toterr, newerr, relerr, maxerr (source/triangulation.hpp:576-719).
// The implementations below are this repository's own.
//
// Conventions: half-edge h = 3t + k runs from vertex k of triangle t to vertex
// (k+1)%3; halfedges[h] is the twin's id or -1; angle(h) is the interior angle at the third vertex.
#pragma once

#include <cmath>
#include <cstring>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <functional>
#include <vector>

#include "tpose.hpp"
#include "utility.hpp"
#include "vec.hpp"
#include "../tpose_hip.h"

namespace tpose {

const float PI = 3.14159265f;

struct triangulation {
    static inline size_t MAXT = (2 << 18);  // capacity of every GPU buffer: 13*NT <= MAXT

    int NT;                        // number of triangles
    std::vector<ivec4> triangles;  // x,y,z = vertex ids (w unused)
    std::vector<int> halfedges;    // 3 per triangle: twin id or -1
    std::vector<ivec4> colors;     // per-triangle colour, 0..255 (w = 1)

    int NP;                           // number of points
    std::vector<vec2> points;         // current positions
    std::vector<vec2> originpoints;   // positions before warping

    std::ofstream out;  // stacked .tri output (io::write appends one record per call)
    std::ifstream in;   // stacked .tri input  (io::read consumes one record per call)

    // two triangles over the four domain corners (-R,-1), (-R,1), (R,-1), (R,1)
    triangulation() {
        points = {vec2(-RATIO, -1), vec2(-RATIO, 1), vec2(RATIO, -1), vec2(RATIO, 1)};
        triangles = {ivec4(0, 1, 2, 0), ivec4(2, 1, 3, 0)};
        halfedges = {-1, 3, -1, 1, -1, -1};
        NP = 4;
        NT = 2;
        colors.assign(MAXT, ivec4(0, 0, 0, 0));
        colors[0] = colors[1] = ivec4(0, 0, 0, 1);
        originpoints = points;
    }
    ~triangulation() {
        if (out.is_open()) out.close();
        if (in.is_open()) in.close();
    }

    // --- queries ---------------------------------------------------------------------------
    int org(int h) const { return triangles[h / 3][h % 3]; }            // origin vertex of half-edge h
    int dst(int h) const { return triangles[h / 3][(h + 1) % 3]; }      // destination vertex
    int apex(int h) const { return triangles[h / 3][(h + 2) % 3]; }     // vertex opposite h

    float angle(int h) {  // interior angle at the vertex opposite half-edge h
        const vec2 u = points[org(h)] - points[apex(h)], w = points[dst(h)] - points[apex(h)];
        if (length(u) == 0) return 0;
        if (length(w) == 0) return 0;
        return std::acos(dot(u, w) / length(u) / length(w));
    }
    float hlength(int h) { return length(points[dst(h)] - points[org(h)]); }

    // Per-frame sweeps (software/triangulate/main.cpp:316-346 runs `angle(h) > 0.8 PI` for every half-edge and
    // `collapse(shortest half-edge)` for every triangle, every frame).  Both tests are almost always far from their
    // bounds; these two filters say "cannot fire" from products alone -- no square root, division or arc cosine -- and
    // leave every close call to the test itself, so the schedule's decisions are the same.
    // false: angle(h) <= 0.8 PI for certain.  cos(0.8 PI) = -0.809; a cosine above -0.7885 (its square below 0.95 x
    // 0.6545 = 0.6218 of |u|^2 |w|^2) is an angle below 2.4793 < 2.5133, float rounding (1e-6) notwithstanding;
    // zero-length sides and NaN fall through to the test.
    bool maybe_wider_than_08pi(int h) const {
        const vec2 a = points[apex(h)];
        const vec2 u = points[org(h)] - a, w = points[dst(h)] - a;
        const float d = dot(u, w);
        if (d >= 0) return false;
        return !(d * d < 0.6218f * (dot(u, u) * dot(w, w)));
    }
    // both filters for the three half-edges of triangle t at once (no h / 3, h % 3 per call): bit k = maybe_wider_than_08pi(3 t + k),
    // bit 3 = maybe_collapsible(t)
    int sweep_candidates(int t) const {
        const ivec4 v = triangles[t];
        const vec2 a = points[v.x], b = points[v.y], c = points[v.z];
        const vec2 ab = b - a, bc = c - b, ca = a - c;
        const float lab = dot(ab, ab), lbc = dot(bc, bc), lca = dot(ca, ca);
        // half-edge k runs vertex k -> k + 1; angle(3 t + k) is the angle at the third vertex: k = 0 at c, 1 at a, 2 at b
        const float dc = -dot(ca, bc), da = -dot(ab, ca), db = -dot(bc, ab);   // (u . w with u, w pointing away from the apex)
        int m = 0;
        if (!(dc >= 0) && !(dc * dc < 0.6218f * (lca * lbc))) m |= 1;
        if (!(da >= 0) && !(da * da < 0.6218f * (lab * lca))) m |= 2;
        if (!(db >= 0) && !(db * db < 0.6218f * (lbc * lab))) m |= 4;
        const float lim = 0.0101f * 0.0101f;
        if (!(lab > lim && lbc > lim && lca > lim)) m |= 8;
        return m;
    }
    // false: every side of t is longer than 0.0101, so collapse() of any of its half-edges returns false (bound 0.01)
    bool maybe_collapsible(int t) const {
        const vec2 a = points[triangles[t].x], b = points[triangles[t].y], c = points[triangles[t].z];
        const float lim = 0.0101f * 0.0101f;
        return !(dot(b - a, b - a) > lim && dot(c - b, c - b) > lim && dot(a - c, a - c) > lim);
    }

    static bool boundary(vec2 p) { return p.x <= -RATIO || p.y <= -1 || p.x >= RATIO || p.y >= 1; }
    int boundary(int t) {  // how many vertices of t lie on (or beyond) the domain boundary
        return (int)boundary(points[triangles[t].x]) + (int)boundary(points[triangles[t].y]) +
               (int)boundary(points[triangles[t].z]);
    }

    // --- direct modifiers ------------------------------------------------------------------
    // remove triangle t (vertices stay); optionally detach its neighbours' twin links first
    bool eraset(int t, bool adjusth = true) {
        if (t >= (int)triangles.size()) return false;
        if (adjusth)
            for (int k = 0; k < 3; k++) {
                const int twin = halfedges[3 * t + k];
                if (twin >= 0) halfedges[twin] = -1;
            }
        triangles.erase(triangles.begin() + t);
        halfedges.erase(halfedges.begin() + 3 * t, halfedges.begin() + 3 * t + 3);
        NT--;
        for (auto& h : halfedges)  // ids behind the hole slide down by one triangle
            if (h >= 3 * (t + 1)) h -= 3;
        return true;
    }
    bool erasep(int p) {
        if (p >= (int)points.size()) return false;
        points.erase(points.begin() + p);
        for (auto& t : triangles)
            for (int k = 0; k < 3; k++)
                if (t[k] >= p) t[k]--;
        NP--;
        return true;
    }

    // --- topological alterations -----------------------------------------------------------
    bool prune(int t) {  // drop a flat triangle sitting on the hull
        if (halfedges[3 * t] >= 0 && halfedges[3 * t + 1] >= 0 && halfedges[3 * t + 2] >= 0) return false;
        for (int k = 0; k < 3; k++)
            if (angle(3 * t + k) > 0 && angle(3 * t + k) < PI) return false;
        return eraset(t);
    }

    // flip the edge shared by ha and its twin if the quad is strictly convex and the two opposite
    // angles sum to at least minangle (PI = Delaunay criterion)
    bool flip(int ha, float minangle = PI) {
        if (ha < 0) return false;
        const int hb = halfedges[ha];
        if (hb < 0) return false;
        const int ta = ha / 3, tb = hb / 3;
        const int ja = ha % 3, jb = hb % 3;

        auto ccw = [](vec2 P, vec2 Q, vec2 S) { return (S.y - P.y) * (Q.x - P.x) > (Q.y - P.y) * (S.x - P.x); };
        const vec2 A = points[org(ha)], B = points[org(hb)], C = points[apex(ha)], D = points[apex(hb)];
        if (ccw(A, C, D) == ccw(B, C, D) || ccw(A, B, C) == ccw(A, B, D)) return false;  // diagonals must cross

        if (minangle <= 0.0f) {
            // (the schedule's flip set: no bound on the angles' sum, only "neither angle is zero".  acos(x) of a float is 0 for x == 1 and at
            // least acos(1 - 2^-24) = 3.5e-4 below it, and NaN -- which passes `<= 1E-8` as false -- above: no arc cosine needed to decide)
            auto zero = [&](int h) {
                const vec2 u = points[org(h)] - points[apex(h)], w = points[dst(h)] - points[apex(h)];
                if (length(u) == 0 || length(w) == 0) return true;
                return dot(u, w) / length(u) / length(w) == 1.0f;
            };
            if (zero(ha) || zero(hb)) return false;
        } else {
            const float aa = angle(ha), ab = angle(hb);
            if (aa + ab < minangle) return false;
            if (aa <= 1E-8 || ab <= 1E-8) return false;
        }

        // the four outer half-edges around the quad, and the two triangles' labels, before the flip
        const int a1 = halfedges[3 * ta + (ja + 1) % 3], a2 = halfedges[3 * ta + (ja + 2) % 3];
        const int b1 = halfedges[3 * tb + (jb + 1) % 3], b2 = halfedges[3 * tb + (jb + 2) % 3];
        const ivec4 va = triangles[ta], vb = triangles[tb];

        // the shared edge keeps its two ids (slot ja of ta, slot jb of tb); the other slots rotate
        halfedges[3 * ta + (ja + 1) % 3] = a2;
        halfedges[3 * ta + (ja + 2) % 3] = b1;
        halfedges[3 * tb + (jb + 1) % 3] = b2;
        halfedges[3 * tb + (jb + 2) % 3] = a1;
        if (a1 >= 0) halfedges[a1] = 3 * tb + (jb + 2) % 3;
        if (a2 >= 0) halfedges[a2] = 3 * ta + (ja + 1) % 3;
        if (b1 >= 0) halfedges[b1] = 3 * ta + (ja + 2) % 3;
        if (b2 >= 0) halfedges[b2] = 3 * tb + (jb + 1) % 3;

        triangles[ta][ja] = vb[(jb + 2) % 3];
        triangles[ta][(ja + 1) % 3] = va[(ja + 2) % 3];
        triangles[ta][(ja + 2) % 3] = vb[(jb + 1) % 3];
        triangles[tb][jb] = va[(ja + 2) % 3];
        triangles[tb][(jb + 1) % 3] = vb[(jb + 2) % 3];
        triangles[tb][(jb + 2) % 3] = va[(ja + 1) % 3];
        return true;
    }

    // collapse the (short) edge of half-edge ha into one new vertex appended at the end
    bool collapse(int ha) {
        if (ha < 0) return false;
        int ta = ha / 3;
        const int ia = org(ha);
        int ib = dst(ha);
        if (length(points[ia] - points[ib]) > 0.01) return false;

        const bool ba = boundary(points[ia]), bb = boundary(points[ib]);
        const vec2 merged = (ba == bb) ? 0.5f * (points[ia] + points[ib]) : (ba ? points[ia] : points[ib]);
        const int in = (int)points.size();
        points.push_back(merged);
        NP++;

        // sew the two surviving sides of each dying triangle together
        auto sew = [&](int h) {
            const int t = h / 3, j = h % 3;
            const int s1 = halfedges[3 * t + (j + 1) % 3], s2 = halfedges[3 * t + (j + 2) % 3];
            if (s1 >= 0) halfedges[s1] = s2;
            if (s2 >= 0) halfedges[s2] = s1;
        };
        sew(ha);
        const int hb = halfedges[ha];
        if (hb >= 0) {
            int tb = hb / 3;
            sew(hb);
            eraset(ta, false);
            if (ta < tb) tb--;
            eraset(tb, false);
        } else {
            eraset(ta, false);
        }

        for (auto& t : triangles)
            for (int k = 0; k < 3; k++)
                if (t[k] == ia || t[k] == ib) t[k] = in;
        erasep(ia);
        if (ia < ib) ib--;
        erasep(ib);
        return true;
    }

    // 1 -> 3 split at the centroid: t keeps (x, y, new); (y, z, new) and (z, x, new) are appended
    bool split(int t) {
        const ivec4 v = triangles[t];
        const int pn = (int)points.size();
        points.push_back((points[v.x] + points[v.y] + points[v.z]) / 3.0f);

        const int ox = halfedges[3 * t], oy = halfedges[3 * t + 1], oz = halfedges[3 * t + 2];
        const int tb = (int)triangles.size(), tc = tb + 1;
        triangles.push_back(ivec4(v.y, v.z, pn, 0));
        triangles.push_back(ivec4(v.z, v.x, pn, 0));
        triangles[t].z = pn;

        halfedges[3 * t + 1] = 3 * tb + 2;
        halfedges[3 * t + 2] = 3 * tc + 1;
        const int add[6] = {oy, 3 * tc + 2, 3 * t + 1, oz, 3 * t + 2, 3 * tb + 1};
        halfedges.insert(halfedges.end(), add, add + 6);
        if (ox >= 0) halfedges[ox] = 3 * t;
        if (oy >= 0) halfedges[oy] = 3 * tb;
        if (oz >= 0) halfedges[oz] = 3 * tc;

        NT += 2;
        NP += 1;
        return true;
    }

    // prune hull slivers, Delaunay-flip across each triangle's widest angle, collapse each triangle's
    // shortest edge -- with the reference's selection quirk (slot 2 is only tried after slot 1 won)
    bool optimize() {
        for (size_t t = 0; t < (size_t)NT; t++)
            if (boundary((int)t) == 3) prune((int)t);
        for (size_t t = 0; t < (size_t)NT; t++) {
            int h = 3 * (int)t;
            float widest = angle(h);
            if (angle(h + 1) > widest) widest = angle(++h);
            if (angle(h + 1) > widest) widest = angle(++h);
            flip(h);
        }
        for (size_t t = 0; t < triangles.size(); t++) {
            int h = 3 * (int)t;
            float shortest = hlength(h);
            if (hlength(h + 1) < shortest) shortest = hlength(++h);
            if (hlength(h + 1) < shortest) shortest = hlength(++h);
            collapse(h);
        }
        return true;
    }

    // --- warping -----------------------------------------------------------------------------
    // carry each non-boundary point from the `from` embedding to the `to` embedding through the first
    // triangle (index order) that strictly contains it
    void transfer(std::vector<vec2>& pts, std::vector<vec2>& from, std::vector<vec2>& to) {
        if (triangles.empty() || points.empty() || originpoints.empty()) return;
        for (auto& p : pts) {
            if (boundary(p)) continue;
            for (auto& t : triangles) {
                if (!intriangle(p, t, from)) continue;
                p = cartesian(barycentric(p, t, from), t, to);
                break;
            }
        }
    }
    void warp(std::vector<vec2>& pts) { transfer(pts, originpoints, points); }         // origin -> current
    void reversewarp(std::vector<vec2>& pts) { transfer(pts, points, originpoints); }  // current -> origin
};

// ================================================================================================
// GPU glue: the reference's tpose::init/quit/upload and host mirrors (source/triangulation.hpp:576-643)
// over one tp_context.  `flavour` selects which program's shaders are "bound" (software/triangulate
// vs software/warp); `warpA` is the warp program's uniform (warp/shader/triangle.fs:49-50).
// ================================================================================================
inline tp_context* ctx = nullptr;
inline int flavour = TP_TRIANGULATE;
inline bool warpA = true;

inline int* terr = nullptr;    // tenergy mirror, int[MAXT]
inline int* perr = nullptr;    // penergy mirror (dead in the reference: always zero)
inline int* cn = nullptr;      // colnum mirror
inline ivec4* col = nullptr;   // colour staging for upload

inline void check(int rc, const char* what) {
    if (rc != TP_OK) {
        std::fprintf(stderr, "tpose: %s failed: %s\n", what, tp_last_error(ctx));
        std::exit(1);
    }
}

// tpose::init() + window/texture setup: one context for a width x height raster on `device`
inline void init(int width, int height, int device = 0) {
    check(tp_create(device, width, height, &ctx), "init");
    terr = new int[triangulation::MAXT]();
    perr = new int[triangulation::MAXT]();
    cn = new int[triangulation::MAXT]();
    col = new ivec4[triangulation::MAXT];
}
inline void quit() {
    tp_destroy(ctx);
    ctx = nullptr;
    delete[] terr; delete[] perr; delete[] cn; delete[] col;
    terr = perr = cn = nullptr; col = nullptr;
}
// Texture tex(IMG): RGBA8, row 0 on top
inline void image(int slot, const uint8_t* rgba, size_t stride_bytes) {
    check(tp_set_image(ctx, slot, rgba, stride_bytes), "image");
}
inline void upload(triangulation* tr, bool uploadcolor = true) {
    check(tp_set_ratio(ctx, RATIO), "upload(RATIO)");
    check(tp_upload(ctx, &tr->points[0].x, tr->NP, &tr->triangles[0].x, tr->NT,
                    uploadcolor ? &tr->colors[0].x : nullptr), "upload");
}
// the draw / dispatch lambdas of the two programs
inline int swept_slot() { return flavour == TP_WARP ? (warpA ? TP_IMAGE_B : TP_IMAGE_A) : TP_IMAGE_A; }
inline void computecolors() { check(tp_accumulate(ctx, flavour, swept_slot()), "computecolors"); }
inline void doreset() { computecolors(); }
inline void doenergy() { check(tp_energy(ctx, flavour), "doenergy"); }
inline void doshift() { check(tp_shift(ctx, flavour == TP_WARP ? 0.00003f : 0.00005f), "doshift"); }
// computecolors + doenergy + doshift as ONE fused launch sequence without host round trips (tp_iterate): the
// same buffers afterwards -- tenergy / colnum / colacc of the sweep at the positions BEFORE the step, points after
inline void doframe() {
    tp_params p;
    tp_default_params(flavour, &p);
    p.image_slot = swept_slot();
    check(tp_iterate(ctx, &p, 1), "doframe");
}
// n frames without the convergence test (a fixed budget per level): one enqueue
inline void doframes(long n) {
    tp_params p;
    tp_default_params(flavour, &p);
    p.image_slot = swept_slot();
    check(tp_iterate(ctx, &p, (int)n), "doframes");
}
// frames until geterr < threshold (or maxframes of them), the test applied on the library's side of the boundary
// (tp_iterate_until: no read-back per frame); afterwards everything stands as after the reference's loop
//     do { doframe(); retrieve(tr); } while (geterr(tr) >= threshold)
// -- the buffers of the last frame in terr / perr / cn, its points in tr, toterr / newerr / relerr as geterr left them.

// the four Buffer::retrieve calls of every frame.  base_only: just the entries the host looks at before the next retrieve --
// those of the NT base variants (id < NT: geterr, maxerrid, the flip ranking, the export) and the two behind them, which a
// split of the same frame brings below the new NT (gettoterr then sums them, as the reference does: software/triangulate/
// main.cpp:348); the other displaced variants only feed gradient.cs on the device
inline size_t base_entries(const triangulation* tr) { return std::min((size_t)13 * tr->NT, (size_t)tr->NT + 2); }
inline void retrieve(triangulation* tr, bool base_only = false) {
    const int what[4] = {TP_BUF_TENERGY, TP_BUF_PENERGY, TP_BUF_COLNUM, TP_BUF_POINTS};
    void* const dst[4] = {terr, perr, cn, &tr->points[0].x};
    const size_t V = base_only ? base_entries(tr) : (size_t)13 * tr->NT;
    const size_t count[4] = {V, V, V, (size_t)2 * tr->NP};
    check(tp_retrieve_many(ctx, 4, what, dst, count), "retrieve");  // one wait for the four buffers
}
inline void retrieve_energy(triangulation* tr, bool base_only = false) {
    check(tp_retrieve(ctx, TP_BUF_TENERGY, terr, base_only ? base_entries(tr) : (size_t)13 * tr->NT), "retrieve(tenergy)");
}
// the base energies n triangles WOULD have at the device's current positions (vertex triples into the uploaded points; tp_evaluate_triangles):
// what "flip, upload, computecolors, doenergy, retrieve" yields for the flipped pairs, without making them real.  false: not available
// for this raster (beyond 4096 columns or rows) -- the caller takes the upload path
// variants: empty (base variants), or one per triangle (0..12: the entry variant * NT + t of `tenergy` if the triple were triangle t)
inline bool evaluate(const std::vector<int>& vertices, std::vector<int>& energy, const std::vector<int>& variants = std::vector<int>()) {
    const int n = (int)(vertices.size() / 3);
    energy.assign((size_t)n, 0);
    if (n == 0) return true;
    const int rc = tp_evaluate_triangles(ctx, swept_slot(), n, vertices.data(), variants.empty() ? nullptr : variants.data(), energy.data(), nullptr);
    if (rc == TP_ERR_STATE) return false;
    check(rc, "evaluate");
    return true;
}
// `terr` as "upload(tr); computecolors(); doenergy(); retrieve_energy(tr, base_only = true)" would leave it -- the entries [0, NT + 2): the base
// energies of tr's triangles and the two entries behind them (variant-major: entry e >= NT is variant e / NT of triangle e % NT) -- from
// tp_evaluate_triangles at the device's positions, WITHOUT the upload: the device keeps the mesh it has.  false: not available.
inline bool evaluate_mesh_energy(const triangulation* tr) {
    const int NT = tr->NT;
    const size_t entries = base_entries(tr);
    std::vector<int> vs, va, e;
    vs.reserve(entries * 3); va.reserve(entries);
    for (size_t k = 0; k < entries; k++) {
        const int t = (int)(k % (size_t)NT), i = (int)(k / (size_t)NT);
        for (int j = 0; j < 3; j++) vs.push_back(tr->triangles[t][j]);
        va.push_back(i);
    }
    if (!evaluate(vs, e, va)) return false;
    for (size_t k = 0; k < entries; k++) terr[k] = e[k];
    return true;
}
// tcolaccbuf->retrieve(tr.NT, &tr.colors[0])
inline void retrieve_colors(triangulation* tr) {
    check(tp_retrieve(ctx, TP_BUF_COLACC, &tr->colors[0].x, (size_t)4 * tr->NT), "retrieve(colacc)");
}

// the `draw` lambda (mode 2 of triangle.fs; software/triangulate/main.cpp:157-174) without a window:
// a flat-shaded RGBA8 picture of the current triangulation, average colours of the last sweep.
// software/view draws the STORED colours at mix(points, originpoints, s) (view/shader/triangle.vs):
// draw_stored(tr, s, ...) after upload(tr).
inline void draw(uint8_t* rgba, size_t stride_bytes) {
    check(tp_render(ctx, TP_RENDER_AVERAGE, nullptr, rgba, stride_bytes), "draw");
}
inline void draw_stored(triangulation* tr, float s, uint8_t* rgba, size_t stride_bytes) {
    std::vector<vec2> mixed(tr->points.size());
    for (size_t i = 0; i < mixed.size(); i++) {  // glsl mix(a, b, s) = a*(1-s) + b*s, per component
        mixed[i].x = tr->points[i].x * (1.0f - s) + tr->originpoints[i].x * s;
        mixed[i].y = tr->points[i].y * (1.0f - s) + tr->originpoints[i].y * s;
    }
    check(tp_render(ctx, TP_RENDER_STORED, &mixed[0].x, rgba, stride_bytes), "draw_stored");
}

// error bookkeeping (float32, ascending t -- the summation order is part of the contract)
inline float toterr = 1.0f;
inline float newerr;
inline float relerr;
inline float maxerr;

inline void sum_energy(triangulation* tr) {
    maxerr = 0.0f;
    newerr = 0.0f;
    for (int i = 0; i < tr->NT; i++) {
        float err = 0.0f;
        err += terr[i];
        if (std::sqrt(err) >= maxerr) maxerr = std::sqrt(err);
        newerr += err;
    }
    relerr = (toterr - newerr) / toterr;
    toterr = newerr;
}
inline float geterr(triangulation* tr) { sum_energy(tr); return std::fabs(relerr); }
inline long descend(triangulation* tr, double threshold, long maxframes) {
    tp_params p;
    tp_default_params(flavour, &p);
    p.image_slot = swept_slot();
    int frames = 0;
    float rel = 0.0f;
    check(tp_iterate_until(ctx, &p, (int)(maxframes > 0x3fffffff ? 0x3fffffff : maxframes), threshold, &toterr, &frames, &rel), "descend");
    newerr = toterr; relerr = rel;
    retrieve(tr);
    return frames;
}
// Frames WITHOUT a read-back each (tp_iterate_frames, round 6): up to maxframes frames of { computecolors; doenergy; doshift } run in chunks on
// the device; after every frame, in order, `terr[0, NT)` and `tr->points` hold what the reference's four retrieves would have left (the
// frame's base energies; the positions after its shift) and fn(k) decides: TP_FRAME_GO_ON, TP_FRAME_STOP_REPLAY (the device is brought to
// the state the frame left -- all buffers readable, as after doframe()), or TP_FRAME_STOP (the caller uploads next).  Returns the frames run.
// `cn` and `perr` are NOT refreshed per frame (the reference's loop looks at them at an export only, which follows a replayed frame).
template <class F>
inline long frames(triangulation* tr, long maxframes, F&& fn) {
    struct pack { triangulation* tr; F* fn; } pk = {tr, &fn};
    tp_params p;
    tp_default_params(flavour, &p);
    p.image_slot = swept_slot();
    int n = 0;
    check(tp_iterate_frames(ctx, &p, (int)(maxframes > 0x3fffffff ? 0x3fffffff : maxframes),
                            [](void* user, int k, const int32_t* ten, const float* pts) -> int {
                                pack* q = static_cast<pack*>(user);
                                std::memcpy(terr, ten, sizeof(int) * (size_t)q->tr->NT);
                                std::memcpy(&q->tr->points[0].x, pts, sizeof(float) * 2 * (size_t)q->tr->NP);
                                return (*q->fn)(k);
                            }, &pk, &n), "frames");
    return n;
}
inline float gettoterr(triangulation* tr) { sum_energy(tr); return std::fabs(toterr); }
inline int maxerrid(triangulation* tr) {
    maxerr = 0;
    int worst = -1;
    for (int i = 0; i < tr->NT; i++) {
        float err = 0.0f;
        err += std::abs(terr[i]);
        if (std::sqrt(err) > maxerr) { maxerr = std::sqrt(err); worst = i; }
    }
    return worst;
}

}  // namespace tpose
