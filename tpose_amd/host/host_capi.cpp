// host_capi.cpp -- C wrapper over the C++ host mirror (include/tpose/triangulation.hpp, io.hpp) so
// that the Python multi-GPU drivers can use tpose::triangulation / tpose::io without re-implementing
// them: stacked .tri I/O, warp / reversewarp, topology ops.  Host-only (no HIP): libtpose_host.so.
#include <cstring>
#include <string>
#include <vector>

#include "tpose/io.hpp"
#include "tpose/triangulation.hpp"

using tpose::triangulation;

extern "C" {

void tph_set_ratio(float r) { tpose::RATIO = r; }
float tph_get_ratio() { return tpose::RATIO; }
void tph_set_verbose(int v) { tpose::io::verbose = v != 0; }

void* tph_new() { return new triangulation(); }
void tph_free(void* h) { delete static_cast<triangulation*>(h); }
int tph_nt(void* h) { return static_cast<triangulation*>(h)->NT; }
int tph_np(void* h) { return static_cast<triangulation*>(h)->NP; }
int32_t* tph_triangles(void* h) { return &static_cast<triangulation*>(h)->triangles[0].x; }
int32_t* tph_halfedges(void* h) { return static_cast<triangulation*>(h)->halfedges.data(); }
int32_t* tph_colors(void* h) { return &static_cast<triangulation*>(h)->colors[0].x; }
float* tph_points(void* h) { return &static_cast<triangulation*>(h)->points[0].x; }
float* tph_originpoints(void* h) { return &static_cast<triangulation*>(h)->originpoints[0].x; }

// replace the mesh (triangles ivec4[NT], points / originpoints vec2[NP]); half-edges and colours optional
void tph_assign(void* h, int NT, int NP, const int32_t* tris, const float* pts, const float* origin,
                const int32_t* halfedges, const int32_t* colors) {
    triangulation* t = static_cast<triangulation*>(h);
    t->NT = NT; t->NP = NP;
    t->triangles.resize(NT); std::memcpy(&t->triangles[0].x, tris, sizeof(int32_t) * 4 * NT);
    t->points.resize(NP); std::memcpy(&t->points[0].x, pts, sizeof(float) * 2 * NP);
    t->originpoints.resize(NP); std::memcpy(&t->originpoints[0].x, origin, sizeof(float) * 2 * NP);
    t->halfedges.assign(3 * (size_t)NT, -1);
    if (halfedges) std::memcpy(t->halfedges.data(), halfedges, sizeof(int32_t) * 3 * NT);
    t->colors.assign(NT, tpose::ivec4(0, 0, 0, 1));
    if (colors) std::memcpy(&t->colors[0].x, colors, sizeof(int32_t) * 4 * NT);
}
void tph_set_points(void* h, const float* pts) {
    triangulation* t = static_cast<triangulation*>(h);
    std::memcpy(&t->points[0].x, pts, sizeof(float) * 2 * t->NP);
}
void tph_points_from_origin(void* h) { triangulation* t = static_cast<triangulation*>(h); t->points = t->originpoints; }
void tph_origin_from_points(void* h) { triangulation* t = static_cast<triangulation*>(h); t->originpoints = t->points; }

int tph_read(void* h, const char* file, int dowarp) { return tpose::io::read(static_cast<triangulation*>(h), file, dowarp != 0) ? 1 : 0; }
void tph_write(void* h, const char* file) { tpose::io::write(static_cast<triangulation*>(h), file); }

static void run_warp(triangulation* t, float* pts, int n, bool reverse) {
    std::vector<tpose::vec2> v(n);
    std::memcpy(&v[0].x, pts, sizeof(float) * 2 * n);
    if (reverse) t->reversewarp(v); else t->warp(v);
    std::memcpy(pts, &v[0].x, sizeof(float) * 2 * n);
}
void tph_warp(void* h, float* pts, int n) { run_warp(static_cast<triangulation*>(h), pts, n, false); }
void tph_reversewarp(void* h, float* pts, int n) { run_warp(static_cast<triangulation*>(h), pts, n, true); }

int tph_flip(void* h, int he, float minangle) { return static_cast<triangulation*>(h)->flip(he, minangle); }
int tph_split(void* h, int t) { return static_cast<triangulation*>(h)->split(t); }
int tph_collapse(void* h, int he) { return static_cast<triangulation*>(h)->collapse(he); }
int tph_prune(void* h, int t) { return static_cast<triangulation*>(h)->prune(t); }

}  // extern "C"
