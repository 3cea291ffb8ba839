// tpose/multiview.hpp -- two-view geometry of the tpose host mirror ("next" row f-3, BASELINE config 5):
// fundamental matrix from warped-vertex correspondences and two-view triangulation.
//
// Same names, argument meaning and results-up-to-rounding as source/multiview.hpp of the reference
// (weigert/t-pose): normalize (:62-87), epole / eline (:91-121), F_8Point (:130-183), F_Sampson
// (:187-242, weighted overload :244-300), F_RANSAC / F_LMEDS (:304-358), HDLT (:369-379), GetPose
// (:390-412), triangulate (:416-520 per match, :522-627 for a set).
//
// PARITY UNPINNED.  The reference's arithmetic lives in Eigen (JacobiSVD, EigenSolver) and OpenCV
// (findFundamentalMat, RANSAC with an unseeded RNG) -- un-vendored, un-pinned, absent from this image.
// This file restates the published algorithms on a small dense-algebra core of its own (cyclic
// Jacobi in double precision); the results agree with an SVD-based evaluation to rounding, and the
// tests compare by Sampson error on the same matches, not by matrix entries.
// Differences kept on purpose are marked "reference:" below.
#pragma once

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdint>
#include <iostream>
#include <vector>

#include "tpose.hpp"
#include "vec.hpp"

namespace tpose {
namespace mview {

// ---------------------------------------------------------------------------------------------
// dense 3x3 / 3x4 / 4-vector types, row-major, float storage like Eigen::Matrix3f
// ---------------------------------------------------------------------------------------------
struct Matrix3f {
    float m[3][3];
    Matrix3f() { for (auto& r : m) for (auto& v : r) v = 0; }
    static Matrix3f Identity() { Matrix3f I; I.m[0][0] = I.m[1][1] = I.m[2][2] = 1; return I; }
    float& operator()(int r, int c) { return m[r][c]; }
    float operator()(int r, int c) const { return m[r][c]; }
    Matrix3f transpose() const { Matrix3f t; for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) t.m[r][c] = m[c][r]; return t; }
};
inline Matrix3f operator*(const Matrix3f& a, const Matrix3f& b) {
    Matrix3f o;
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) {
        double s = 0; for (int k = 0; k < 3; k++) s += (double)a.m[r][k] * b.m[k][c];
        o.m[r][c] = (float)s;
    }
    return o;
}
inline Matrix3f operator/(const Matrix3f& a, float s) { Matrix3f o; for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) o.m[r][c] = a.m[r][c] / s; return o; }
struct Vector3f { float v[3]; float& operator()(int i) { return v[i]; } float operator()(int i) const { return v[i]; } };
struct Vector4f { float v[4]; float& operator()(int i) { return v[i]; } float operator()(int i) const { return v[i]; } };
struct Matrix34f { float m[3][4]; float& operator()(int r, int c) { return m[r][c]; } float operator()(int r, int c) const { return m[r][c]; } };
inline Vector3f operator*(const Matrix3f& a, const Vector3f& x) {
    Vector3f o;
    for (int r = 0; r < 3; r++) o.v[r] = (float)((double)a.m[r][0] * x.v[0] + (double)a.m[r][1] * x.v[1] + (double)a.m[r][2] * x.v[2]);
    return o;
}
inline Matrix34f operator*(const Matrix3f& a, const Matrix34f& b) {
    Matrix34f o;
    for (int r = 0; r < 3; r++) for (int c = 0; c < 4; c++) {
        double s = 0; for (int k = 0; k < 3; k++) s += (double)a.m[r][k] * b.m[k][c];
        o.m[r][c] = (float)s;
    }
    return o;
}
inline Vector3f operator*(const Matrix34f& a, const Vector4f& x) {
    Vector3f o;
    for (int r = 0; r < 3; r++) { double s = 0; for (int k = 0; k < 4; k++) s += (double)a.m[r][k] * x.v[k]; o.v[r] = (float)s; }
    return o;
}
inline std::ostream& operator<<(std::ostream& os, const Matrix3f& a) {
    for (int r = 0; r < 3; r++) os << a.m[r][0] << " " << a.m[r][1] << " " << a.m[r][2] << (r < 2 ? "\n" : "");
    return os;
}

namespace detail {

// Symmetric eigen-decomposition by cyclic Jacobi rotations (double).  A: n x n row-major, destroyed;
// V: eigenvectors in columns.  Returns eigenvalues unsorted in w.
inline void jacobi_eigen(std::vector<double>& A, int n, std::vector<double>& V, std::vector<double>& w) {
    V.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n; i++) V[(size_t)i * n + i] = 1.0;
    for (int sweep = 0; sweep < 64; sweep++) {
        double off = 0, diag = 0;
        for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) (i == j ? diag : off) += A[(size_t)i * n + j] * A[(size_t)i * n + j];
        if (off <= 1e-60 || off <= 1e-32 * diag) break;
        for (int p = 0; p < n - 1; p++)
            for (int q = p + 1; q < n; q++) {
                const double apq = A[(size_t)p * n + q];
                if (apq == 0.0) continue;
                const double theta = (A[(size_t)q * n + q] - A[(size_t)p * n + p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; k++) {  // A <- A J
                    const double akp = A[(size_t)k * n + p], akq = A[(size_t)k * n + q];
                    A[(size_t)k * n + p] = c * akp - s * akq;
                    A[(size_t)k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; k++) {  // A <- J^T A
                    const double apk = A[(size_t)p * n + k], aqk = A[(size_t)q * n + k];
                    A[(size_t)p * n + k] = c * apk - s * aqk;
                    A[(size_t)q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; k++) {
                    const double vkp = V[(size_t)k * n + p], vkq = V[(size_t)k * n + q];
                    V[(size_t)k * n + p] = c * vkp - s * vkq;
                    V[(size_t)k * n + q] = s * vkp + c * vkq;
                }
            }
    }
    w.resize(n);
    for (int i = 0; i < n; i++) w[i] = A[(size_t)i * n + i];
}

// right singular vector of the smallest singular value of the rows x n matrix M (row-major doubles):
// what JacobiSVD(M, ComputeFullV).matrixV().col(n-1) holds, up to sign
inline std::vector<double> null_vector(const std::vector<double>& M, int rows, int n) {
    std::vector<double> G((size_t)n * n, 0.0), V, w;
    for (int r = 0; r < rows; r++)
        for (int i = 0; i < n; i++) {
            const double a = M[(size_t)r * n + i];
            if (a == 0.0) continue;
            for (int j = 0; j < n; j++) G[(size_t)i * n + j] += a * M[(size_t)r * n + j];
        }
    jacobi_eigen(G, n, V, w);
    int best = 0;
    for (int i = 1; i < n; i++) if (w[i] < w[best]) best = i;
    std::vector<double> v(n);
    for (int i = 0; i < n; i++) v[i] = V[(size_t)i * n + best];
    return v;
}

// full SVD of a 3x3: F = U diag(S) V^T, S descending, U and V orthogonal (double)
struct SVD3 { double U[3][3], S[3], V[3][3]; };
inline SVD3 svd3(const Matrix3f& F) {
    std::vector<double> G(9, 0.0), V, w;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++) G[(size_t)i * 3 + j] += (double)F.m[k][i] * F.m[k][j];
    jacobi_eigen(G, 3, V, w);
    int ord[3] = {0, 1, 2};
    std::sort(ord, ord + 3, [&](int a, int b) { return w[a] > w[b]; });
    SVD3 o;
    for (int c = 0; c < 3; c++) {
        o.S[c] = std::sqrt(std::max(w[ord[c]], 0.0));
        for (int r = 0; r < 3; r++) o.V[r][c] = V[(size_t)r * 3 + ord[c]];
    }
    // U columns: F v / sigma; columns of (near-)zero singular values completed orthogonally
    const double tol = 1e-12 * std::max(o.S[0], 1e-300);
    int have = 0;
    for (int c = 0; c < 3; c++) {
        if (o.S[c] > tol) {
            for (int r = 0; r < 3; r++) {
                double s = 0; for (int k = 0; k < 3; k++) s += (double)F.m[r][k] * o.V[k][c];
                o.U[r][c] = s / o.S[c];
            }
            have = c + 1;
        }
    }
    auto cross = [&](int a, int b, int dst) {
        o.U[0][dst] = o.U[1][a] * o.U[2][b] - o.U[2][a] * o.U[1][b];
        o.U[1][dst] = o.U[2][a] * o.U[0][b] - o.U[0][a] * o.U[2][b];
        o.U[2][dst] = o.U[0][a] * o.U[1][b] - o.U[1][a] * o.U[0][b];
    };
    if (have == 0) { for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) o.U[r][c] = r == c; }
    else if (have == 1) {  // any unit vector orthogonal to column 0, then the cross product
        int k = std::fabs(o.U[0][0]) < std::fabs(o.U[1][0]) ? (std::fabs(o.U[0][0]) < std::fabs(o.U[2][0]) ? 0 : 2)
                                                            : (std::fabs(o.U[1][0]) < std::fabs(o.U[2][0]) ? 1 : 2);
        double e[3] = {0, 0, 0}; e[k] = 1;
        double d = o.U[k][0], nrm = 0;
        for (int r = 0; r < 3; r++) { o.U[r][1] = e[r] - d * o.U[r][0]; nrm += o.U[r][1] * o.U[r][1]; }
        nrm = std::sqrt(nrm);
        for (int r = 0; r < 3; r++) o.U[r][1] /= nrm;
        cross(0, 1, 2);
    } else if (have == 2) cross(0, 1, 2);
    return o;
}

inline Matrix3f rank2(const Matrix3f& F) {  // zero the smallest singular value
    const SVD3 s = svd3(F);
    Matrix3f o;
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++)
        o.m[r][c] = (float)(s.U[r][0] * s.S[0] * s.V[c][0] + s.U[r][1] * s.S[1] * s.V[c][1]);
    return o;
}

inline Matrix3f unflatten(const std::vector<double>& f) {
    Matrix3f F;
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) F.m[r][c] = (float)f[(size_t)3 * r + c];
    return F;
}

inline Matrix3f inverse(const Matrix3f& a) {
    const double m00 = a.m[0][0], m01 = a.m[0][1], m02 = a.m[0][2], m10 = a.m[1][0], m11 = a.m[1][1], m12 = a.m[1][2],
                 m20 = a.m[2][0], m21 = a.m[2][1], m22 = a.m[2][2];
    const double det = m00 * (m11 * m22 - m12 * m21) - m01 * (m10 * m22 - m12 * m20) + m02 * (m10 * m21 - m11 * m20);
    Matrix3f o;
    o.m[0][0] = (float)((m11 * m22 - m12 * m21) / det); o.m[0][1] = (float)((m02 * m21 - m01 * m22) / det); o.m[0][2] = (float)((m01 * m12 - m02 * m11) / det);
    o.m[1][0] = (float)((m12 * m20 - m10 * m22) / det); o.m[1][1] = (float)((m00 * m22 - m02 * m20) / det); o.m[1][2] = (float)((m02 * m10 - m00 * m12) / det);
    o.m[2][0] = (float)((m10 * m21 - m11 * m20) / det); o.m[2][1] = (float)((m01 * m20 - m00 * m21) / det); o.m[2][2] = (float)((m00 * m11 - m01 * m10) / det);
    return o;
}

}  // namespace detail

// ---------------------------------------------------------------------------------------------
// camera intrinsics (source/multiview.hpp:33-49: the RealSense numbers of the reference's tests)
// ---------------------------------------------------------------------------------------------
inline int check = 3;
inline float px = 488.421f / 960.0f;
inline float py = 268.8f / 960.0f;
inline float fx = 673.101f / 960.0f;
inline float fy = 673.328f / 960.0f;

inline Matrix3f Camera() {
    Matrix3f K;
    K(0, 0) = 1.0f / fx; K(0, 2) = px;
    K(1, 1) = 1.0f / fy; K(1, 2) = py;
    K(2, 2) = 1;
    return K;
}

// ---------------------------------------------------------------------------------------------
// normalisation, epipoles, epipolar lines
// ---------------------------------------------------------------------------------------------
// Hartley normalisation IN PLACE: centroid to the origin, mean distance sqrt(2); returns H with
// p_normalised = H p  (source/multiview.hpp:62-87)
inline Matrix3f normalize(std::vector<vec2>& points) {
    vec2 c(0);
    for (auto& p : points) c += p;
    c = c / (float)points.size();
    float dist = 0.0f;
    for (auto& p : points) { p -= c; dist += length(p); }
    dist /= (float)points.size();
    const float scale = (float)(std::sqrt(2.0) / dist);
    for (auto& p : points) p = p * scale;
    Matrix3f H;
    H(0, 0) = scale; H(0, 2) = -c.x * scale;
    H(1, 1) = scale; H(1, 2) = -c.y * scale;
    H(2, 2) = 1;
    return H;
}

// right (F e = 0) or left (e^T F = 0) epipole, dehomogenised  (:91-107)
inline vec2 epole(const Matrix3f& F, bool right = true) {
    const detail::SVD3 s = detail::svd3(F);
    if (right) return vec2((float)(s.V[0][2] / s.V[2][2]), (float)(s.V[1][2] / s.V[2][2]));
    return vec2((float)(s.U[0][2] / s.U[2][2]), (float)(s.U[1][2] / s.U[2][2]));
}

// epipolar line F p, scaled so that its third coefficient is 1  (:109-114)
inline vec3 eline(const Matrix3f& F, vec2 p) {
    Vector3f x{{p.x, p.y, 1.0f}};
    const Vector3f l = F * x;
    return vec3(l(0) / l(2), l(1) / l(2), 1.0f);
}
// F^T p  (:116-121)
inline vec3 eline(vec2 p, const Matrix3f& F) { return eline(F.transpose(), p); }

// first-order geometric (Sampson) distance of a match to F, squared: (b^T F a)^2 / (|Fa|_xy^2 + |F^T b|_xy^2)
inline double sampson2(const Matrix3f& F, vec2 a, vec2 b) {
    const double ax = a.x, ay = a.y, bx = b.x, by = b.y;
    const double l0 = F(0, 0) * ax + F(0, 1) * ay + F(0, 2), l1 = F(1, 0) * ax + F(1, 1) * ay + F(1, 2), l2 = F(2, 0) * ax + F(2, 1) * ay + F(2, 2);
    const double r0 = F(0, 0) * bx + F(1, 0) * by + F(2, 0), r1 = F(0, 1) * bx + F(1, 1) * by + F(2, 1);
    const double e = bx * l0 + by * l1 + l2;
    return e * e / (l0 * l0 + l1 * l1 + r0 * r0 + r1 * r1);
}
inline double mean_sampson(const Matrix3f& F, const std::vector<vec2>& A, const std::vector<vec2>& B) {
    double s = 0;
    for (size_t n = 0; n < A.size(); n++) s += sampson2(F, A[n], B[n]);
    return A.empty() ? 0.0 : s / (double)A.size();
}

// ---------------------------------------------------------------------------------------------
// fundamental matrix estimation:  pB^T F pA = 0
// ---------------------------------------------------------------------------------------------
namespace detail {
inline void fill_rows(std::vector<double>& M, const std::vector<vec2>& pA, const std::vector<vec2>& pB, const double* w) {
    const size_t N = pA.size();
    M.assign(N * 9, 0.0);
    for (size_t n = 0; n < N; n++) {
        const double s = w ? w[n] : 1.0, ax = pA[n].x, ay = pA[n].y, bx = pB[n].x, by = pB[n].y;
        double* r = &M[n * 9];
        r[0] = s * ax * bx; r[1] = s * ay * bx; r[2] = s * bx;
        r[3] = s * ax * by; r[4] = s * ay * by; r[5] = s * by;
        r[6] = s * ax;      r[7] = s * ay;      r[8] = s;
    }
}
}  // namespace detail

// normalised 8-point algorithm  (:130-183)
inline Matrix3f F_8Point(std::vector<vec2> pA, std::vector<vec2> pB) {
    Matrix3f F = Matrix3f::Identity();
    if (pA.size() != pB.size()) { std::cout << "Error: Matched sets have different size" << std::endl; return F; }
    if (pA.size() == 0) { std::cout << "Error: Matched sets are empty" << std::endl; return F; }
    const Matrix3f HA = normalize(pA), HB = normalize(pB);
    std::vector<double> M;
    detail::fill_rows(M, pA, pB, nullptr);
    F = detail::rank2(detail::unflatten(detail::null_vector(M, (int)pA.size(), 9)));
    F = HB.transpose() * F * HA;
    return F / F(2, 2);
}

// iteratively re-weighted 8-point: 100 rounds of weights 1 / (|F^T b|_xy^2 + |F a|_xy^2) on the lines
// scaled to third coefficient 1, as written in the reference  (:187-242; `w`: extra per-match weights,
// :244-300).  reference: the first round evaluates the UN-normalised initial guess on the normalised
// points (F_8Point returns image coordinates, the points are normalised afterwards) -- kept.
inline Matrix3f F_Sampson(std::vector<vec2> pA, std::vector<vec2> pB, const std::vector<float>* w = nullptr) {
    Matrix3f F = F_8Point(pA, pB);
    const size_t N = pA.size();
    if (N == 0 || pB.size() != N) return F;
    const Matrix3f HA = normalize(pA), HB = normalize(pB);
    std::vector<double> W(N), M;
    const size_t MAXITER = 100;
    for (size_t k = 0; k < MAXITER; k++) {
        for (size_t n = 0; n < N; n++) {
            const vec3 L = eline(F.transpose(), pB[n]);
            const vec3 R = eline(F, pA[n]);
            W[n] = 1.0f / (L.x * L.x + L.y * L.y + R.x * R.x + R.y * R.y);
            if (w) W[n] *= (*w)[n];
        }
        detail::fill_rows(M, pA, pB, W.data());
        F = detail::rank2(detail::unflatten(detail::null_vector(M, (int)N, 9)));
    }
    F = HB.transpose() * F * HA;
    return F / F(2, 2);
}
inline Matrix3f F_Sampson(std::vector<vec2> pA, std::vector<vec2> pB, std::vector<float> w) { return F_Sampson(pA, pB, &w); }

// RANSAC over 8-match samples with a Sampson-distance inlier test, then a refit on the inliers.
// reference: cv::findFundamentalMat(FM_RANSAC, threshold, 0.99) on the matches that do not touch the
// domain boundary (:332-358; F_LMEDS calls FM_RANSAC as well, with threshold 0.0025, :304-330).  OpenCV's
// sampler is unseeded, so the reference's own output is not reproducible; this one is (splitmix64).
inline Matrix3f F_RANSAC(const std::vector<vec2>& A, const std::vector<vec2>& B, double threshold = 0.001, double confidence = 0.99,
                         uint64_t seed = 1) {
    std::vector<vec2> pA, pB;
    for (size_t i = 0; i < A.size() && i < B.size(); i++) {
        if (A[i].x <= -tpose::RATIO || A[i].x >= tpose::RATIO || A[i].y <= -1 || A[i].y >= 1) continue;  // no boundary points
        if (B[i].x <= -tpose::RATIO || B[i].x >= tpose::RATIO || B[i].y <= -1 || B[i].y >= 1) continue;
        pA.push_back(A[i]); pB.push_back(B[i]);
    }
    const size_t N = pA.size();
    if (N < 8) return Matrix3f::Identity();
    auto next = [&seed]() { uint64_t z = (seed += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); };
    const double t2 = threshold * threshold;
    std::vector<char> best(N, 0);
    size_t best_count = 0;
    long trials = 2000;
    for (long it = 0; it < trials; it++) {
        size_t idx[8];
        for (int k = 0; k < 8; k++) {
            bool dup;
            do { idx[k] = (size_t)(next() % N); dup = false; for (int j = 0; j < k; j++) dup |= idx[j] == idx[k]; } while (dup);
        }
        std::vector<vec2> sa(8), sb(8);
        for (int k = 0; k < 8; k++) { sa[k] = pA[idx[k]]; sb[k] = pB[idx[k]]; }
        const Matrix3f F = F_8Point(sa, sb);
        if (!std::isfinite(F(0, 0))) continue;
        size_t count = 0;
        std::vector<char> in(N, 0);
        for (size_t n = 0; n < N; n++) if (sampson2(F, pA[n], pB[n]) <= t2) { in[n] = 1; count++; }
        if (count > best_count) {
            best_count = count; best.swap(in);
            const double ratio = (double)count / (double)N;  // adaptive trial count for the requested confidence
            const double miss = 1.0 - std::pow(ratio, 8);
            if (miss < 1e-12) trials = it + 1;
            else trials = std::min(trials, std::max(it + 1, (long)std::ceil(std::log(1.0 - confidence) / std::log(miss))));
        }
    }
    if (best_count < 8) return F_8Point(pA, pB);
    std::vector<vec2> ia, ib;
    for (size_t n = 0; n < N; n++) if (best[n]) { ia.push_back(pA[n]); ib.push_back(pB[n]); }
    return F_8Point(ia, ib);
}
inline Matrix3f F_LMEDS(const std::vector<vec2>& A, const std::vector<vec2>& B) { return F_RANSAC(A, B, 0.0025, 0.99); }

// ---------------------------------------------------------------------------------------------
// polynomial helpers (source/utility.hpp:85-135)
// ---------------------------------------------------------------------------------------------
inline double horner(double x, const std::vector<double>& a) {
    double r = a[a.size() - 1];
    for (int i = (int)a.size() - 2; i >= 0; i--) r = r * x + a[i];
    return r;
}

// real roots of a[0] + a[1] x + ... + a[K] x^K, refined by 25 Newton steps like the reference.
// reference: eigenvalues of the companion matrix (Eigen::EigenSolver), kept when imag() == 0; here
// Durand-Kerner in complex doubles, kept when |imag| is at rounding level; ascending order.
inline std::vector<double> realroots(std::vector<double> a) {
    while (a.size() > 1 && a.back() == 0.0) a.pop_back();
    const int K = (int)a.size() - 1;
    std::vector<double> out;
    if (K < 1) return out;
    typedef std::complex<double> cd;
    std::vector<cd> z(K);
    double bound = 0;
    for (int k = 0; k < K; k++) bound = std::max(bound, std::fabs(a[k] / a[K]));
    bound = 1.0 + bound;
    for (int k = 0; k < K; k++) z[k] = std::polar(bound * 0.5 + 0.1, 2.0 * M_PI * k / K + 0.4);
    auto eval = [&](cd x) { cd r = a[K]; for (int i = K - 1; i >= 0; i--) r = r * x + a[i]; return r; };
    for (int it = 0; it < 500; it++) {
        double move = 0;
        for (int k = 0; k < K; k++) {
            cd den = a[K];
            for (int j = 0; j < K; j++) if (j != k) den *= (z[k] - z[j]);
            if (std::abs(den) == 0.0) { z[k] += cd(1e-8, 1e-8); continue; }
            const cd d = eval(z[k]) / den;
            z[k] -= d;
            move = std::max(move, std::abs(d));
        }
        if (move < 1e-15 * bound) break;
    }
    std::vector<double> deriv(K);
    for (int k = 1; k <= K; k++) deriv[k - 1] = a[k] * (double)k;
    for (int k = 0; k < K; k++) {
        if (std::fabs(z[k].imag()) > 1e-7 * std::max(1.0, std::abs(z[k]))) continue;
        double r = z[k].real();
        for (int n = 0; n < 25; n++) {
            const double d = horner(r, deriv);
            if (d == 0.0) break;
            r -= horner(r, a) / d;
        }
        out.push_back(r);
    }
    std::sort(out.begin(), out.end());
    return out;
}

// ---------------------------------------------------------------------------------------------
// triangulation
// ---------------------------------------------------------------------------------------------
// homogeneous DLT: X with xA ~ PA X, xB ~ PB X  (:369-379)
inline Vector4f HDLT(const Matrix34f& PA, const Matrix34f& PB, const Vector3f& xA, const Vector3f& xB) {
    std::vector<double> H(16);
    for (int c = 0; c < 4; c++) {
        H[0 * 4 + c] = (double)xA(0) * PA(2, c) - PA(0, c);
        H[1 * 4 + c] = (double)xA(1) * PA(2, c) - PA(1, c);
        H[2 * 4 + c] = (double)xB(0) * PB(2, c) - PB(0, c);
        H[3 * 4 + c] = (double)xB(1) * PB(2, c) - PB(1, c);
    }
    const std::vector<double> v = detail::null_vector(H, 4, 4);
    return Vector4f{{(float)v[0], (float)v[1], (float)v[2], (float)v[3]}};
}

struct Pose { Matrix3f R1, R2; Vector3f t; };

// the two rotations and the translation direction (sign open) of an essential matrix  (:390-412)
inline Pose GetPose(const Matrix3f& Essential) {
    const detail::SVD3 s = detail::svd3(Essential);
    Matrix3f U, V, W;
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { U(r, c) = (float)s.U[r][c]; V(r, c) = (float)s.V[r][c]; }
    W(0, 1) = -1; W(1, 0) = 1; W(2, 2) = 1;
    Pose pose;
    pose.t = Vector3f{{U(0, 2), U(1, 2), U(2, 2)}};
    pose.R1 = U * W * V.transpose();
    pose.R2 = U * W.transpose() * V.transpose();
    return pose;
}

// optimal two-view correction (Hartley & Sturm): moves A and B the least (sum of squared
// distances) so that B^T F A = 0 exactly  (:416-520).
// Two deliberate deviations from the reference, both listed in DESIGN.md (parity notes): its candidate loop evaluates
// the cost at the loop INDEX (`S(r)` instead of `S(R[r])`, :496) and starts from an eigenvalue order this build cannot
// reproduce; here the cost is evaluated at the real roots R[r] themselves.  Like the reference, the asymptotic
// candidate t -> infinity of the method (cost 1/m^2 + c^2 / (a^2 + n^2 c^2)) is NOT examined, and nothing is
// corrected when the polynomial has no real root.
inline void triangulate(Matrix3f F, vec2& A, vec2& B) {
    Matrix3f TA = Matrix3f::Identity(), TB = Matrix3f::Identity();
    TA(0, 2) = -A.x; TA(1, 2) = -A.y;
    TB(0, 2) = -B.x; TB(1, 2) = -B.y;
    const Matrix3f TAi = detail::inverse(TA), TBi = detail::inverse(TB);
    F = TBi.transpose() * F * TAi;  // both points at the origin
    const detail::SVD3 s = detail::svd3(F);
    double eA[3] = {s.V[0][2], s.V[1][2], s.V[2][2]}, eB[3] = {s.U[0][2], s.U[1][2], s.U[2][2]};
    const double nA = std::sqrt(eA[0] * eA[0] + eA[1] * eA[1]), nB = std::sqrt(eB[0] * eB[0] + eB[1] * eB[1]);
    for (int k = 0; k < 3; k++) { eA[k] /= nA; eB[k] /= nB; }
    Matrix3f RA, RB;
    RA(0, 0) = (float)eA[0]; RA(0, 1) = (float)eA[1]; RA(1, 0) = (float)-eA[1]; RA(1, 1) = (float)eA[0]; RA(2, 2) = 1;
    RB(0, 0) = (float)eB[0]; RB(0, 1) = (float)eB[1]; RB(1, 0) = (float)-eB[1]; RB(1, 1) = (float)eB[0]; RB(2, 2) = 1;
    F = RB * F * RA.transpose();
    const double m = eA[2], n = eB[2], a = F(1, 1), b = F(1, 2), c = F(2, 1), d = F(2, 2);
    auto S = [&](double t) {
        return t * t / (1.0 + m * m * t * t) + (c * t + d) * (c * t + d) / ((a * t + b) * (a * t + b) + n * n * (c * t + d) * (c * t + d));
    };
    // gradient polynomial, degree 6 (:469-475)
    const double m2 = m * m, m4 = m2 * m2, n2 = n * n, n4 = n2 * n2;
    const double a0 = b * b * c * d - a * b * d * d;
    const double a1 = b * b * b * b + (b * b * c * c - a * a * d * d) + 2.0 * b * b * d * d * n2 + d * d * d * d * n4;
    const double a2 = (a * b * c * c - a * a * c * d) + 4.0 * a * b * b * b + 2.0 * (b * b * c * d - a * b * d * d) * m2 +
                      4.0 * (a * b * d * d + b * b * c * d) * n2 + 4.0 * c * d * d * d * n4;
    const double a3 = 6.0 * a * a * b * b + 2.0 * (b * b * c * c - a * a * d * d) * m2 + 2.0 * a * a * d * d * n2 + 8.0 * a * b * c * d * n2 +
                      2.0 * b * b * c * c * n2 + 6.0 * c * c * d * d * n4;
    const double a4 = (b * b * c * d - a * b * d * d) * m4 + 4.0 * a * a * a * b + 2.0 * (a * b * c * c - a * a * c * d) * m2 +
                      4.0 * (a * a * c * d + a * b * c * c) * n2 + 4.0 * c * c * c * d * n4;
    const double a5 = a * a * a * a + (b * b * c * c - a * a * d * d) * m4 + 2.0 * a * a * c * c * n2 + c * c * c * c * n4;
    const double a6 = (a * b * c * c - a * a * c * d) * m4;
    const std::vector<double> R = realroots({a0, a1, a2, a3, a4, a5, a6});
    if (R.empty()) return;
    double t = R[0], minerr = S(R[0]);
    for (size_t r = 1; r < R.size(); r++) {
        const double cur = S(R[r]);
        if (cur < minerr) { t = R[r]; minerr = cur; }
    }
    // closest points on the two epipolar lines of parameter t
    const double LA[3] = {t * m, 1.0, -t}, LB[3] = {-n * (c * t + d), a * t + b, c * t + d};
    Vector3f XA{{(float)(-LA[0] * LA[2]), (float)(-LA[1] * LA[2]), (float)(LA[0] * LA[0] + LA[1] * LA[1])}};
    Vector3f XB{{(float)(-LB[0] * LB[2]), (float)(-LB[1] * LB[2]), (float)(LB[0] * LB[0] + LB[1] * LB[1])}};
    XA = TAi * (RA.transpose() * XA);
    XB = TBi * (RB.transpose() * XB);
    A = vec2(XA(0) / XA(2), XA(1) / XA(2));
    B = vec2(XB(0) / XB(2), XB(1) / XB(2));
}

struct vec4 { float x, y, z, w; };

// corrected matches -> 3D points with cameras PA = K [I | 0], PB = K [R | t] for the pose candidate
// `check` (0: R1,+t  1: R1,-t  2: R2,+t  3: R2,-t)  (:522-627)
inline std::vector<vec4> triangulate(const Matrix3f& F, const Matrix3f& K, std::vector<vec2> A, std::vector<vec2> B) {
    std::vector<vec4> points;
    const size_t N = A.size();
    for (size_t n = 0; n < N; n++) triangulate(F, A[n], B[n]);
    const Matrix3f E = K.transpose() * F * K;
    const Pose P = GetPose(E);
    Matrix34f PA{}, PB{};
    for (int r = 0; r < 3; r++) for (int c = 0; c < 4; c++) PA(r, c) = r == c ? 1.0f : 0.0f;
    const Matrix3f& R = check < 2 ? P.R1 : P.R2;
    const float sg = (check & 1) ? -1.0f : 1.0f;
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) PB(r, c) = R(r, c); PB(r, 3) = sg * P.t(r); }
    const Matrix34f KA = K * PA, KB = K * PB;
    for (size_t n = 0; n < N; n++) {
        const Vector3f XA{{A[n].x, A[n].y, 1}}, XB{{B[n].x, B[n].y, 1}};
        Vector4f X = HDLT(KA, KB, XA, XB);
        points.push_back(vec4{X(0) / X(3), X(1) / X(3), X(2) / X(3), 1.0f});
    }
    return points;
}

}  // namespace mview
}  // namespace tpose
