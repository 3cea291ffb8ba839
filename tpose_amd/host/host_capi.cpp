// host_capi.cpp -- C wrapper over the C++ host mirror (include/tpose/triangulation.hpp, io.hpp) so
// that the Python multi-GPU drivers can use tpose::triangulation / tpose::io without re-implementing
// them: stacked .tri I/O, warp / reversewarp, topology ops.  Host-only (no HIP): libtpose_host.so.
#include <cstring>
#include <string>
#include <vector>

#include "tpose/io.hpp"
#include "tpose/multiview.hpp"
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


// ---- two-view geometry (include/tpose/multiview.hpp); matches as float[2N] arrays, F as float[9] row-major
static void to_vec(const float* p, int n, std::vector<tpose::vec2>& v) { v.resize(n); for (int i = 0; i < n; i++) v[i] = tpose::vec2(p[2 * i], p[2 * i + 1]); }
static void put_F(const tpose::mview::Matrix3f& F, float* out) { for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) out[3 * r + c] = F(r, c); }
static tpose::mview::Matrix3f get_F(const float* in) { tpose::mview::Matrix3f F; for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) F(r, c) = in[3 * r + c]; return F; }

// method 0: F_8Point, 1: F_Sampson, 2: F_RANSAC (threshold 0.001), 3: F_LMEDS (the reference's alias, 0.0025)
void tph_fundamental(int method, const float* a, const float* b, int n, float* F9) {
    std::vector<tpose::vec2> A, B;
    to_vec(a, n, A); to_vec(b, n, B);
    using namespace tpose::mview;
    put_F(method == 0 ? F_8Point(A, B) : method == 1 ? F_Sampson(A, B) : method == 2 ? F_RANSAC(A, B) : F_LMEDS(A, B), F9);
}
double tph_mean_sampson(const float* F9, const float* a, const float* b, int n) {
    std::vector<tpose::vec2> A, B;
    to_vec(a, n, A); to_vec(b, n, B);
    return tpose::mview::mean_sampson(get_F(F9), A, B);
}
// optimal correction of every match, in place
void tph_correct_matches(const float* F9, float* a, float* b, int n) {
    const tpose::mview::Matrix3f F = get_F(F9);
    for (int i = 0; i < n; i++) {
        tpose::vec2 A(a[2 * i], a[2 * i + 1]), B(b[2 * i], b[2 * i + 1]);
        tpose::mview::triangulate(F, A, B);
        a[2 * i] = A.x; a[2 * i + 1] = A.y; b[2 * i] = B.x; b[2 * i + 1] = B.y;
    }
}
// 3D points float[4N] for pose candidate `check`
void tph_triangulate(const float* F9, const float* K9, const float* a, const float* b, int n, int check, float* X4) {
    std::vector<tpose::vec2> A, B;
    to_vec(a, n, A); to_vec(b, n, B);
    tpose::mview::check = check;
    const auto pts = tpose::mview::triangulate(get_F(F9), get_F(K9), A, B);
    for (int i = 0; i < n; i++) { X4[4 * i] = pts[i].x; X4[4 * i + 1] = pts[i].y; X4[4 * i + 2] = pts[i].z; X4[4 * i + 3] = pts[i].w; }
}
int tph_realroots(const double* coeff, int ncoeff, double* out) {
    const std::vector<double> r = tpose::mview::realroots(std::vector<double>(coeff, coeff + ncoeff));
    for (size_t i = 0; i < r.size(); i++) out[i] = r[i];
    return (int)r.size();
}
void tph_epole(const float* F9, int right, float* e2) { const tpose::vec2 e = tpose::mview::epole(get_F(F9), right != 0); e2[0] = e.x; e2[1] = e.y; }
int tph_readmatches(const char* file, float* a, float* b, int cap) {
    std::vector<tpose::vec2> A, B;
    if (!tpose::io::readmatches(file, A, B)) return -1;
    const int n = (int)A.size() < cap ? (int)A.size() : cap;
    for (int i = 0; i < n; i++) { a[2 * i] = A[i].x; a[2 * i + 1] = A[i].y; b[2 * i] = B[i].x; b[2 * i + 1] = B[i].y; }
    return (int)A.size();
}

}  // extern "C"
