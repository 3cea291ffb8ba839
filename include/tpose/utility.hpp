// tpose/utility.hpp -- barycentric helpers of the tpose host mirror.
// Same contracts as source/utility.hpp:26-83 of the reference (barycentric / intriangle / cartesian);
// the polynomial-root and Eigen<->glm helpers of that file feed only multiview.hpp and are outside the
// hot path.  The reference evaluates these through glm::mat3 (determinant, inverse, mat*vec); glm is an
// un-vendored dependency, so its published cofactor formulas are restated here in float32 -- results
// agree with the reference to float rounding, not necessarily to the last ulp ("parity unpinned").
#pragma once

#include <cmath>
#include <vector>

#include "vec.hpp"

namespace tpose {

// Barycentric coordinates s of p in triangle t over vertex set v:  sum s = 1, sum s_k v_k = p.
// Degenerate triangles (|det| < 1e-8) yield (1,1,1), which intriangle() rejects.
inline vec3 barycentric(vec2 p, ivec4 t, const std::vector<vec2>& v) {
    // column-major R = [ (1, ax, ay) | (1, bx, by) | (1, cx, cy) ];  R s = (1, px, py)
    const float m00 = 1, m01 = v[t.x].x, m02 = v[t.x].y;
    const float m10 = 1, m11 = v[t.y].x, m12 = v[t.y].y;
    const float m20 = 1, m21 = v[t.z].x, m22 = v[t.z].y;
    const float det = m00 * (m11 * m22 - m21 * m12) - m10 * (m01 * m22 - m21 * m02) + m20 * (m01 * m12 - m11 * m02);
    if (std::fabs(det) < 1E-8) return vec3(1, 1, 1);
    const float ood = 1.0f / det;
    // inverse by cofactors, I[col][row]
    const float i00 = +(m11 * m22 - m21 * m12) * ood, i10 = -(m10 * m22 - m20 * m12) * ood, i20 = +(m10 * m21 - m20 * m11) * ood;
    const float i01 = -(m01 * m22 - m21 * m02) * ood, i11 = +(m00 * m22 - m20 * m02) * ood, i21 = -(m00 * m21 - m20 * m01) * ood;
    const float i02 = +(m01 * m12 - m11 * m02) * ood, i12 = -(m00 * m12 - m10 * m02) * ood, i22 = +(m00 * m11 - m10 * m01) * ood;
    const float vx = 1, vy = p.x, vz = p.y;
    return vec3(i00 * vx + i10 * vy + i20 * vz, i01 * vx + i11 * vy + i21 * vz, i02 * vx + i12 * vy + i22 * vz);
}

// strictly inside: every barycentric coordinate in the open interval (0, 1)
inline bool intriangle(vec2 p, ivec4 t, const std::vector<vec2>& v) {
    if (length(v[t.x] - v[t.y]) == 0) return false;
    if (length(v[t.y] - v[t.z]) == 0) return false;
    if (length(v[t.z] - v[t.x]) == 0) return false;
    const vec3 s = barycentric(p, t, v);
    if (s.x <= 0 || s.x >= 1) return false;
    if (s.y <= 0 || s.y >= 1) return false;
    if (s.z <= 0 || s.z >= 1) return false;
    return true;
}

inline vec2 cartesian(vec3 s, ivec4 t, const std::vector<vec2>& v) {
    return s.x * v[t.x] + s.y * v[t.y] + s.z * v[t.z];
}

}  // namespace tpose
