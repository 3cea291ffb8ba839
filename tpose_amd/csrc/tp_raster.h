// tp_raster.h -- per-lane geometry of the t-pose hot path: vertex stage, edge setup and the exact
// row-span walker used by the HIP kernels (tp_kernels.hip).
//
// What it replaces in the reference (weigert/t-pose):
//   software/triangulate/shader/triangle.vs:45-84  (warp: software/warp/shader/triangle.vs:48-87)
//       -- the 13-variant vertex stage;
//   the OpenGL rasteriser between triangle.vs and triangle.fs -- here an explicit, exact rule:
//       pixel-centre sampling, vertices snapped to 1/256 px, integer edge functions, top-left rule,
//       orientation agnostic (the reference disables culling, software/triangulate/main.cpp:56).
//
// Nothing tests pixels one by one.  The picture pass (k_render) walks the rows of a triangle and gets the covered
// column span [lo, hi) of each row in O(1) from three 32.32 fixed-point edge walkers whose floor is provably exact
// (tp_make_walker).  The cost function never visits the inside of a triangle at all: a variant's pixel moments are
// the signed sum of three LINE sums ("Edge-centric form" below), a line sum reads one record of the per-image row
// prefix table per row ("Per-image row prefix table"), and a line's crossing column of every row comes from ONE
// 24.40 fixed-point set-up per line and iteration ("Whole-line walkers").
//
// Everything here is __host__ __device__ so that tests can run the same integer logic on the CPU
// (tests/emul) -- the shipped library never executes it on the host.
#pragma once

#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define TP_HD __host__ __device__ __forceinline__
#else
#define TP_HD inline
#endif

// snapped coordinates are clamped to [-2^22, 2^23]: every edge delta fits in 24 bits
#define TP_COORD_MIN (-4194304.0f)
#define TP_COORD_MAX (8388608.0f)
#define TP_MAX_RASTER 16384
#define TP_NVARIANTS 13

// ---------------------------------------------------------------------------------------------
// float helpers: each is exactly one IEEE-754 binary32 operation (no contraction)
// ---------------------------------------------------------------------------------------------
TP_HD float tp_fadd(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __fadd_rn(a, b);
#else
    volatile float r = a + b; return r;
#endif
}
TP_HD float tp_fsub(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __fsub_rn(a, b);
#else
    volatile float r = a - b; return r;
#endif
}
TP_HD float tp_fmul(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __fmul_rn(a, b);
#else
    volatile float r = a * b; return r;
#endif
}
TP_HD float tp_fdiv(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __fdiv_rn(a, b);
#else
    volatile float r = a / b; return r;
#endif
}

// shift.cs:45  `/ 256 / 256`: two divisions by a power of two are one multiplication by 2^-16 -- exact, unless the result
// were subnormal, which rate * gradient / 65536 never is for a non-zero integer gradient (>= 1e-5 * 2^-16)
TP_HD float tp_shift_scale(float v) { return tp_fmul(v, 1.52587890625e-05f); }

// dp law, triangle.vs:60-62 (triangulate: 4, 3000) / warp triangle.vs:63-65 (9, 1000)
inline float tp_reference_dp(int flavour, int NT) {
    volatile float k = flavour ? 9.0f : 4.0f, m = flavour ? 1000.0f : 3000.0f;
    volatile float a = k * (float)NT;
    volatile float b = a / m;
    volatile float c = 1.0f + b;
    volatile float d = 0.05f / c;
    return d;
}

TP_HD int32_t tp_min(int32_t a, int32_t b) { return a < b ? a : b; }
TP_HD int32_t tp_max(int32_t a, int32_t b) { return a > b ? a : b; }
// v clamped to [0, hi] (hi >= 0, the same for every lane): v_med3_i32 with an inline constant and a scalar operand
TP_HD int32_t tp_clamp0(int32_t v, int32_t hi) {
#if defined(__HIP_DEVICE_COMPILE__)
    int32_t d;
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(d) : "v"(v), "s"(hi));
    return d;
#else
    return v < 0 ? 0 : (v > hi ? hi : v);
#endif
}

TP_HD int32_t tp_snap256(float f) {
    float v = tp_fadd(tp_fmul(f, 256.0f), 0.5f);
    v = fmaxf(v, TP_COORD_MIN);  // NaN -> lower bound
    v = fminf(v, TP_COORD_MAX);
    return (int32_t)floorf(v);
}

struct tp_view {
    float dp, ratio, halfW, halfH;  // halfW = 0.5f * W (exact)
    int W, H;
};

// Vertex stage of a vertex displaced by (Dx, Dy) t-pose units (triangle.vs:66-84): displacement BEFORE x /= RATIO.
TP_HD void tp_vertex_stage_d(float px, float py, float Dx, float Dy, const tp_view& vw, int32_t& X, int32_t& Y) {
    float tx = tp_fadd(px, Dx);
    float ty = tp_fadd(py, Dy);
    float nx = tx;
    if (vw.ratio != 1.0f) {   // (x / 1 is x: square rasters skip the division -- a branch, the same for every lane, not a select behind the division)
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("");
#endif
        nx = tp_fdiv(tx, vw.ratio);
    }
    float fx = tp_fmul(tp_fadd(nx, 1.0f), vw.halfW);
    float fy = tp_fmul(tp_fsub(1.0f, ty), vw.halfH);
    X = tp_snap256(fx);
    Y = tp_snap256(fy);
}
// the displacement of move m (0: none, 1..4: +dp x, -dp x, +dp y, -dp y -- triangle.vs:66-78)
TP_HD float tp_move_dx(int m, float dp) { return m == 1 ? dp : m == 2 ? -dp : 0.0f; }
TP_HD float tp_move_dy(int m, float dp) { return m == 3 ? dp : m == 4 ? -dp : 0.0f; }
// Vertex stage for model-vertex `slot` of variant `i` (TDIV)
TP_HD void tp_vertex_stage(float px, float py, int i, int slot, const tp_view& vw, int32_t& X,
                           int32_t& Y) {
    const int m = (i > 0 && ((i - 1) >> 2) == slot) ? ((i - 1) & 3) + 1 : 0;
    tp_vertex_stage_d(px, py, tp_move_dx(m, vw.dp), tp_move_dy(m, vw.dp), vw, X, Y);
}

TP_HD int32_t tp_floor_shr8(int32_t v) { return v >> 8; }  // arithmetic shift == floor(v/256)

// pixel (inclusive) range whose centres 256c+128 lie within [vmin, vmax]
TP_HD int32_t tp_first_centre(int32_t vmin) { return (vmin - 128 + 255) >> 8; }
TP_HD int32_t tp_last_centre(int32_t vmax) { return (vmax - 128) >> 8; }

// ---------------------------------------------------------------------------------------------
// Edge walker.  For one edge with d = |a| (1 <= d < 2^24) the column bound of row r is
// floor(N_r / d) with N_{r+1} = N_r + step, integers, |N| < 2^41, |step| < 2^24.  It is tracked as
// x_r = x_0 + r*s in 32.32 fixed point, built WITHOUT integer division:
//     inv = 1/d (double),  t = N_0*inv,  x_0 = floor(t)*2^32 + trunc(frac(t)*2^32) + BIAS,
//     ts = step*inv,       s   = floor(ts)*2^32 + trunc(frac(ts)*2^32),   BIAS = floor(2^31*inv).
// Error budget in units of 2^-32, with u = 2^32/d >= 256 the spacing of the possible exact
// fractional parts of N_r/d:  t and ts carry relative error <= 2^-50 (|t| <= 2^41/d -> < 2^-9 u),
// each truncation loses < 1 unit, so after at most 32 rows  x_r - BIAS  lies in
// (N_r/d - 34 units - 2^-8 u,  N_r/d + 2^-8 u]  -- within (-u/4, +u/4) of the exact value.  Adding
// BIAS = u/2 puts x_r strictly between N_r/d and N_r/d + u whenever N_r/d is a multiple of 1/d,
// hence  floor(x_r) == floor(N_r/d)  for every row: the span ends are EXACT.
// Quotients beyond +-2^30 are clamped (per-row drift < 2^24, so they stay far outside any raster
// for the whole <= 32-row window and int32 never overflows).
// ---------------------------------------------------------------------------------------------
#define TP_WALK_MAXROWS 32

struct tp_walker {
    int64_t x, s;
};

TP_HD double tp_rcp_exact(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(d);  // v_rcp_f64 (reduced precision) + two Newton steps
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    return r;
#else
    return 1.0 / d;
#endif
}

// floor(t)*2^32 + trunc(frac(t)*2^32) for |t| <= 2^30
TP_HD int64_t tp_fix32(double t) {
    const double q = floor(t);
    const double f = (t - q) * 4294967296.0;  // exact subtraction, in [0, 2^32)
    const int32_t qi = (int32_t)q;
    uint32_t fi = (f >= 4294967295.0) ? 4294967295u : (uint32_t)f;
    return (int64_t)(((uint64_t)(uint32_t)qi << 32) | fi);
}

TP_HD tp_walker tp_make_walker(int64_t N0, int32_t step, int32_t d) {
    const double inv = tp_rcp_exact((double)d);
    double t = (double)N0 * inv;
    t = fmin(fmax(t, -1073741824.0), 1073741824.0);
    tp_walker w;
    w.x = tp_fix32(t) + (int64_t)(uint32_t)(2147483648.0 * inv);  // d = 1: bias 2^31 = half a unit step
    w.s = tp_fix32((double)step * inv);
    return w;
}

TP_HD int32_t tp_walker_value(const tp_walker& w) { return (int32_t)(w.x >> 32); }

// ---------------------------------------------------------------------------------------------
// Span setup for one variant inside a window of rows [win_r0, win_r1] (inclusive, absolute).
// Canonical slots: A is a left edge (lower bound), B a right edge (exclusive upper bound), C the
// third edge (either kind, or neutral when horizontal).
// ---------------------------------------------------------------------------------------------
struct tp_span {
    tp_walker A, B;    // a left edge (inclusive lower bound) and a right edge (exclusive upper bound)
    tp_walker Cl, Ch;  // the third edge, filed under the side it bounds; the other one is inert
    int32_t r0, r1;    // absolute row range (inclusive); empty when r0 > r1
};

#define TP_INERT_LO (-((int64_t)1 << 62))  // walker value -2^30: never the max
#define TP_INERT_HI (((int64_t)1 << 62))   // walker value +2^30: never the min

// straight-line (single exit, no early returns: keeps everything in registers on the GPU)
TP_HD void tp_setup_span(const int32_t X[3], const int32_t Y[3], int32_t win_r0, int32_t win_r1, tp_span& sp) {
    const int64_t area2 = (int64_t)(X[1] - X[0]) * (Y[2] - Y[0]) - (int64_t)(Y[1] - Y[0]) * (X[2] - X[0]);
    const int32_t sg = area2 > 0 ? 1 : -1;

    const int32_t ymin = tp_min(Y[0], tp_min(Y[1], Y[2]));
    const int32_t ymax = tp_max(Y[0], tp_max(Y[1], Y[2]));

    const int32_t a0 = -(Y[1] - Y[0]) * sg, b0 = (X[1] - X[0]) * sg;
    const int32_t a1 = -(Y[2] - Y[1]) * sg, b1 = (X[2] - X[1]) * sg;
    const int32_t a2 = -(Y[0] - Y[2]) * sg, b2 = (X[0] - X[2]) * sg;
    // horizontal edge with the interior above it: a centre exactly on it is excluded
    const int bottom_flat = ((a0 == 0) & (b0 < 0)) | ((a1 == 0) & (b1 < 0)) | ((a2 == 0) & (b2 < 0));
    const int32_t r0 = tp_max(win_r0, tp_first_centre(ymin));
    const int32_t r1 = tp_min(win_r1, tp_last_centre(ymax - bottom_flat));
    const int64_t cy = 256LL * r0 + 128;

    // per edge: E(c,r) >= 0 (with the tie rule)  <=>  a*256*c + K >= 0,  Kf = floor(K/256) steps by b
    //   a > 0 (left edge):   c >= ceil(-Kf / a)  =  floor((-Kf + a - 1) / a)
    //   a < 0 (right edge):  c <= floor(Kf / d)   ->  exclusive bound floor((Kf + d) / d),  d = -a
    //   a == 0: horizontal, already folded into r0 / r1
    tp_walker w0, w1, w2;
#define TP_EDGE(W, A, B, XE, YE)                                                                 \
    {                                                                                            \
        const int32_t left = (A) > 0;                                                            \
        const int64_t K = (int64_t)(A) * (128 - (XE)) + (int64_t)(B) * (cy - (YE)) + left - 1;   \
        const int64_t Kf = K >> 8;                                                               \
        const int32_t d = (A) == 0 ? 1 : ((A) > 0 ? (A) : -(A));                                 \
        W = tp_make_walker(left ? (-Kf + d - 1) : (Kf + d), left ? -(B) : (B), d);               \
    }
    TP_EDGE(w0, a0, b0, X[0], Y[0])
    TP_EDGE(w1, a1, b1, X[1], Y[1])
    TP_EDGE(w2, a2, b2, X[2], Y[2])
#undef TP_EDGE
    // canonical slots: A = first left edge, B = first right edge, C = the remaining one
    const int l0 = a0 > 0, l1 = a1 > 0, l2 = a2 > 0;
    const int u0 = a0 < 0, u1 = a1 < 0, u2 = a2 < 0;
    const int ia = l0 ? 0 : l1 ? 1 : 2;
    const int ib = u0 ? 0 : u1 ? 1 : 2;
    const int ic = 3 - ia - ib;
    sp.A = ia == 0 ? w0 : ia == 1 ? w1 : w2;
    sp.B = ib == 0 ? w0 : ib == 1 ? w1 : w2;
    const tp_walker wc = ic == 0 ? w0 : ic == 1 ? w1 : w2;
    const int cl = ic == 0 ? l0 : ic == 1 ? l1 : l2;
    const int cu = ic == 0 ? u0 : ic == 1 ? u1 : u2;
    sp.Cl.x = cl ? wc.x : TP_INERT_LO; sp.Cl.s = cl ? wc.s : 0;
    sp.Ch.x = cu ? wc.x : TP_INERT_HI; sp.Ch.s = cu ? wc.s : 0;
    // degenerate: zero area, or (never for area != 0) no left / no right edge
    const int ok = (area2 != 0) & ((l0 | l1 | l2) != 0) & ((u0 | u1 | u2) != 0) & (ia != ib);
    sp.r0 = ok ? r0 : 0;
    sp.r1 = ok ? r1 : -1;
}

// column span of the current row, clipped to [clip_lo, clip_hi); hi >= lo always (hi == lo: empty);
// then advance one row
TP_HD void tp_span_row(tp_span& sp, int32_t clip_lo, int32_t clip_hi, int32_t& lo, int32_t& hi) {
    lo = tp_max(tp_max(tp_walker_value(sp.A), tp_walker_value(sp.Cl)), clip_lo);
    hi = tp_min(tp_min(tp_walker_value(sp.B), tp_walker_value(sp.Ch)), clip_hi);
    hi = tp_max(hi, lo);
    sp.A.x += sp.A.s; sp.B.x += sp.B.s; sp.Cl.x += sp.Cl.s; sp.Ch.x += sp.Ch.s;
}

// conservative pixel bounding box of all 13 variants of a triangle (inclusive; may be empty)
struct tp_bbox { int32_t c0, c1, r0, r1; };

TP_HD tp_bbox tp_triangle_bbox(const float p[3][2], const tp_view& vw) {
    int32_t xmin = INT32_MAX, xmax = INT32_MIN, ymin = INT32_MAX, ymax = INT32_MIN;
#pragma unroll
    for (int s = 0; s < 3; s++) {
#pragma unroll
        for (int k = 0; k < 5; k++) {  // k = 0: unmoved, 1..4: the four displacements of slot s
            int32_t X, Y;
            tp_vertex_stage(p[s][0], p[s][1], k == 0 ? 0 : 4 * s + k, s, vw, X, Y);
            xmin = tp_min(xmin, X); xmax = tp_max(xmax, X);
            ymin = tp_min(ymin, Y); ymax = tp_max(ymax, Y);
        }
    }
    tp_bbox bb;
    bb.c0 = tp_max(0, tp_first_centre(xmin)); bb.c1 = tp_min(vw.W - 1, tp_last_centre(xmax));
    bb.r0 = tp_max(0, tp_first_centre(ymin)); bb.r1 = tp_min(vw.H - 1, tp_last_centre(ymax));
    return bb;
}

// ---------------------------------------------------------------------------------------------
// Moments -> energy (replaces the mode-1 pass, triangle.fs:37-43 / warp :46-53).
// For an integer reference colour a:  sum_i (|I_i - a|^2 >> 1) = (S - n_oddS) / 2 with
// S = Q - 2 a.sumI + n |a|^2 and |I_i - a|^2 odd  <=>  (r+g+b) + |a|^2 odd.
// ---------------------------------------------------------------------------------------------
struct tp_moments { int64_t n, nodd, sr, sg, sb, q; };

TP_HD int32_t tp_wrap32(int64_t v) { return (int32_t)(uint32_t)(uint64_t)v; }

TP_HD int64_t tp_energy64(const tp_moments& m, int64_t ar, int64_t ag, int64_t ab) {
    const int64_t a2 = ar * ar + ag * ag + ab * ab;
    const int64_t S = m.q - 2 * (ar * m.sr + ag * m.sg + ab * m.sb) + m.n * a2;
    const int64_t nodd = (a2 & 1) ? (m.n - m.nodd) : m.nodd;
    return (S - nodd) / 2;
}

// triangulate flavour: a = ca.rgb / cn with the int32 (wrapped) sums of the reference SSBO;
// energy 0 when cn == 0 (triangle.fs:40)
// x / n as C truncates it (n > 0), for the three channel sums of a variant: on the device ONE 32-bit reciprocal serves all three (the
// compiler's expansion of an int32 division builds its own each time: ~35 instructions, four of them quarter-rate multiplications, on
// the chain between the walk and the step).  z = floor(2^32 / n) to within 2 after one Newton step on the float estimate; the quotient
// estimate mulhi(|x|, z) is then at most 2 short, which the two remainder steps make good -- the usual unsigned expansion, shared.
struct tp_rcp32 { uint32_t n, z; };
TP_HD tp_rcp32 tp_rcp32_of(int32_t n) {
    tp_rcp32 r; r.n = (uint32_t)n; r.z = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t z = (uint32_t)(__builtin_amdgcn_rcpf((float)r.n) * 4294966784.0f);   // (0x4f7ffffe: never above 2^32 / n)
    z += __umulhi(z, (0u - r.n) * z);
    r.z = z;
#endif
    return r;
}
TP_HD int32_t tp_div_by(int32_t x, const tp_rcp32& r) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t ax = x < 0 ? 0u - (uint32_t)x : (uint32_t)x;
    uint32_t q = __umulhi(ax, r.z), rem = ax - q * r.n;
    if (rem >= r.n) { q++; rem -= r.n; }
    if (rem >= r.n) q++;
    return x < 0 ? (int32_t)(0u - q) : (int32_t)q;
#else
    return x / (int32_t)r.n;
#endif
}
TP_HD int64_t tp_energy_triangulate(const tp_moments& m) {
    const int32_t n32 = tp_wrap32(m.n);
    if (n32 <= 0) return 0;
    const tp_rcp32 r = tp_rcp32_of(n32);
    return tp_energy64(m, tp_div_by(tp_wrap32(m.sr), r), tp_div_by(tp_wrap32(m.sg), r), tp_div_by(tp_wrap32(m.sb), r));
}

// =============================================================================================
// Edge-centric form.  The covered span of a triangle row is [f(left edge), f(right edge)) where,
// for ANY edge line through snapped points T (top) and B (bottom), dy = Yb - Yt > 0,
//     f(r) = the first column c whose centre satisfies (256c + 128 - Xt) * dy >= dx * (256r + 128 - Yt)
// -- left edges include a centre lying exactly on the line, right edges exclude it, and both rules
// give this same f (that is what makes the rasterisation watertight).  Rows belong to an edge when
// Yt <= 256r + 128 < Yb; the non-horizontal edges on either side of a triangle partition its rows.
// With P_r(x) the full-row prefix sum of any pixel moment,
//     moment(triangle) = sum_{right edges} W(e) - sum_{left edges} W(e),   W(e) = sum_{rows of e} P_r(clamp(f(r), 0, W))
// so the 13 variants of all triangles only need W for each distinct edge line: per undirected edge
// the base line + 4 displacements of either endpoint.  A line is walked once, with ONE prefix
// lookup per row, and both triangles sharing it reuse the result.
// =============================================================================================
// signs with which the three edge sums enter a variant's moments: +1 right edge, -1 left edge, 0
// horizontal or degenerate.  Edge k runs from vertex k to vertex (k+1)%3.
TP_HD void tp_variant_coeffs(const int32_t X[3], const int32_t Y[3], int32_t c[3]) {
    const int64_t area2 = (int64_t)(X[1] - X[0]) * (Y[2] - Y[0]) - (int64_t)(Y[1] - Y[0]) * (X[2] - X[0]);
    const int32_t sg = area2 > 0 ? 1 : (area2 < 0 ? -1 : 0);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int j = k == 2 ? 0 : k + 1;
        const int32_t a = -(Y[j] - Y[k]) * sg;  // > 0: interior to the right of the edge (left edge)
        c[k] = (a < 0) - (a > 0);
    }
}

// which of an edge's nine lines a variant uses for its edge k (origin = vertex slot k, destination =
// slot (k+1)%3): 0 base, 1..4 canonical first endpoint displaced by move m, 5..8 the second.
// `flipped`: the half-edge runs from the edge's second endpoint to its first.
TP_HD int tp_edge_version(int variant, int k, int flipped) {
    if (variant == 0) return 0;
    const int s = (variant - 1) >> 2, m = ((variant - 1) & 3) + 1;
    if (s == k) return flipped ? 4 + m : m;                       // origin displaced
    if (s == (k == 2 ? 0 : k + 1)) return flipped ? m : 4 + m;   // destination displaced
    return 0;
}

// =============================================================================================
// Whole-line walkers (round 2).  A line is set up ONCE per iteration as x(r) = x(ra) + (r - ra) * s in 24.40
// fixed point, valid for every row of the line inside the raster (up to TP_MAX_RASTER rows); a lane of k_lines steps
// it by a fixed number of rows at a time (integer arithmetic: the same values).
//
// Error budget, in units of 2^-40 with u = 2^40 / d the spacing of the attainable fractions of N_r / d
// (d = dy < 2^23.6 because snapped coordinates lie in [-2^22, 2^23]):  t = N_0 / d and ts = dx / d carry a
// relative error <= 2^-50 (|t| <= 2^15 + 2: the crossing column of a row BETWEEN the line's endpoints;
// |ts| < 2^16 whenever the line has two rows, i.e. d > 256), each truncation loses < 1 unit, so after R <=
// 16383 steps   x_r - BIAS  lies in  (N_r/d - (R + 2) - 2^-8 u,  N_r/d + 2^-8 u].  With BIAS = u / 2 the value
// therefore stays strictly inside [N_r/d, N_r/d + u) as long as  R + 2 + 2^-7 u < u / 2,  which holds for every
// d < 2^23.6 (u > 86 000 > 2 * 16385 / (1 - 2^-6)).  Hence floor(x_r) == floor(N_r / d): EXACT.
// Single-row lines store s = 0 (the slope of a nearly horizontal line does not fit the format and is
// never used); empty lines have ra > rb.
// =============================================================================================
#define TP_LINE_FRAC 40

struct tp_line {
    int64_t x, s;    // 24.40: crossing column of row ra (+ bias), step per row
    int32_t ra, rb;  // absolute rows (inclusive) inside the raster; empty when ra > rb
};

// floor(t) * 2^40 + trunc(frac(t) * 2^40) for |t| < 2^22
TP_HD int64_t tp_fix40(double t) {
    const double q = floor(t);
    const double f = (t - q) * 1099511627776.0;  // exact subtraction; [0, 2^40)
    const double fh = floor(f * (1.0 / 4294967296.0));
    const double fl = f - fh * 4294967296.0;     // exact: [0, 2^32)
    const int32_t qi = (int32_t)q;
    const uint32_t hi = (uint32_t)fh > 255u ? 255u : (uint32_t)fh;
    const uint32_t lo = fl >= 4294967295.0 ? 4294967295u : (uint32_t)fl;
    return (int64_t)((uint64_t)(int64_t)qi << 40) + (int64_t)(((uint64_t)hi << 32) | lo);
}

TP_HD void tp_setup_line(int32_t Xa, int32_t Ya, int32_t Xb, int32_t Yb, int32_t H, tp_line& ln) {
    const bool swap = Ya > Yb;
    const int32_t Xt = swap ? Xb : Xa, Yt = swap ? Yb : Ya, Xq = swap ? Xa : Xb, Yq = swap ? Ya : Yb;
    const int32_t dy = Yq - Yt, dx = Xq - Xt;
    const int32_t ra = tp_max(0, tp_first_centre(Yt));          // Yt <= 256 r + 128
    const int32_t rb = tp_min(H - 1, tp_first_centre(Yq) - 1);  // 256 r + 128 < Yq
    const bool live = dy > 0 && ra <= rb;
    const int32_t d = dy > 0 ? dy : 1;
    const int64_t N1 = (int64_t)dx * (256LL * ra + 128 - Yt) + (int64_t)(Xt - 128) * dy;
    const int64_t Nc = -((-N1) >> 8);  // ceil(N1 / 256)
    const double inv = tp_rcp_exact((double)d);
    double t = (double)(Nc + d - 1) * inv;   // floor((Nc + d - 1) / d) = ceil(Nc / d): the first column on or right of the line
    t = fmin(fmax(t, -4194304.0), 4194304.0);  // never reached for a live line (|t| <= 2^15 + 2)
    const int64_t bias = (int64_t)(549755813888.0 * inv);  // 2^39 / d = u / 2
    ln.x = live ? tp_fix40(t) + bias : 0;
    ln.s = (live && rb > ra) ? tp_fix40((double)dx * inv) : 0;
    ln.ra = live ? ra : 1;
    ln.rb = live ? rb : 0;
}

// exact crossing column of row r (ra <= r <= rb), clamped to [0, W] like the walk clamps it
TP_HD int32_t tp_line_col(const tp_line& ln, int32_t r, int32_t W) {
    const int32_t x = (int32_t)((ln.x + (int64_t)(r - ln.ra) * ln.s) >> TP_LINE_FRAC);
    return x < 0 ? 0 : (x > W ? W : x);
}


// =============================================================================================
// Per-image row prefix table (round 2b).  The raster does not change between iterations, so the full-row prefix
// sums P_r(c) the line sums need are tabulated once per image.  To keep the table small (8 bytes per pixel) a row
// is cut into groups of four pixels; the 32-byte record of group g holds
//   words 0..3  the moments of the pixels x < 4g:  sum r, sum g, sum b, sum (r^2 + g^2 + b^2)
//               (W <= 16384: 255 W < 2^22, 3 * 255^2 W < 2^32)
//   words 4..6  the r, g, b bytes of the group's four pixels (pixel i in byte i; zero beyond the raster)
//   word 7      n_odd of the pixels x < 4g
// and P_r(c) = record(c >> 2) + the first (c & 3) pixels of the group: one mask, three ANDs, six 4-byte dot products
// and a population count.
// n_odd counts the pixels with r + g + b odd (parity of the sum = parity of r xor g xor b).
// =============================================================================================
#define TP_PFX_WORDS 8
TP_HD int32_t tp_prefix_groups(int32_t W) { return (W >> 2) + 1; }                   // c = 0..W  ->  g = 0..W >> 2
TP_HD int32_t tp_prefix_pitch(int32_t W) { return (tp_prefix_groups(W) + 3) & ~3; }  // records per row: whole 128-byte lines

TP_HD uint32_t tp_udot4(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_udot4(a, b, c, false);
#else
    for (int i = 0; i < 4; i++) c += ((a >> (8 * i)) & 0xffu) * ((b >> (8 * i)) & 0xffu);
    return c;
#endif
}
TP_HD uint32_t tp_popc(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_popcount(v);
#else
    uint32_t n = 0; for (; v; v &= v - 1) n++; return n;
#endif
}

// running moments m = {n_odd, r, g, b, q} of the pixels before a group, and the group's pixels -> record
TP_HD void tp_prefix_pack(const uint32_t m[5], const uint32_t px[4], int npx, uint32_t rec[TP_PFX_WORDS]) {
    rec[0] = m[1]; rec[1] = m[2]; rec[2] = m[3]; rec[3] = m[4];
    uint32_t R = 0, G = 0, B = 0;
    for (int i = 0; i < npx; i++) {
        R |= (px[i] & 0xffu) << (8 * i); G |= ((px[i] >> 8) & 0xffu) << (8 * i); B |= ((px[i] >> 16) & 0xffu) << (8 * i);
    }
    rec[4] = R; rec[5] = G; rec[6] = B; rec[7] = m[0];
}
// the moments of the pixels x < c of the row whose record for group c >> 2 is `rec`
TP_HD void tp_prefix_eval(const uint32_t rec[TP_PFX_WORDS], int32_t c, uint32_t& nodd, uint32_t& r, uint32_t& g, uint32_t& b, uint32_t& q) {
    const uint32_t keep = (1u << ((c << 3) & 31)) - 1u;  // the bytes of the first c & 3 pixels (v_bfm_b32)
    const uint32_t R = rec[4] & keep, G = rec[5] & keep, B = rec[6] & keep;
    nodd = rec[7] + tp_popc((R ^ G ^ B) & 0x01010101u);
    r = tp_udot4(R, 0x01010101u, rec[0]);
    g = tp_udot4(G, 0x01010101u, rec[1]);
    b = tp_udot4(B, 0x01010101u, rec[2]);
    q = tp_udot4(R, R, tp_udot4(G, G, tp_udot4(B, B, rec[3])));
}


// =============================================================================================
// Pixel records (round 3): the row prefix sums in 16 bytes per PIXEL COLUMN, for rasters up to TP_PX_MAXW columns -- what
// the persistent kernel (tp_persist.hip) reads.  Record c of a row holds the moments of the pixels x < c in fields
// with FOUR BITS OF HEADROOM each, so that up to 16 records can be added as two 64-bit integers without a carry from
// one field into the next, and unpacked once:
//   low  64 bits   sum r (20 bits: 255 * 4096 < 2^20) in a 24-bit slot | sum g << 24 | n_odd bits 0..11 << 48 (16-bit slot)
//   high 64 bits   sum b in a 24-bit slot | sum (r^2 + g^2 + b^2) (30 bits: 195075 * 4096 < 2^30) << 24 (34-bit slot)
//                  | n_odd bit 12 << 58 (6-bit slot)
// One look-up is one 16-byte load and two 64-bit additions; a lane of the walk unpacks its six sums once per grad-iter.
// (The 32-byte records per four pixels above cost ~14 instructions per look-up to evaluate and two loads; on this part
// a wave64 integer instruction takes four cycles of its SIMD, and the walk was bound by exactly that.)
// =============================================================================================
#define TP_PX_MAXW 4096
#define TP_PX_MAXSUM 16   /* records that may be added before unpacking */
TP_HD int32_t tp_px_pitch(int32_t W) { return (W + 1 + 7) & ~7; }   // records per row (c = 0..W): whole 128-byte lines
// The same records a second time, TILED: a 128-byte line holds 4 rows x 2 columns instead of 8 columns of one row.  What walks a line
// WITHOUT keeping its records (tp_persist.h: pk_walk_rows_tiled -- every record fetched again every grad-iter) reads this copy: the
// versions of an edge cross a row within a column or two of each other and neighbouring rows a column or two further on, so a steep line's
// 4 rows x 9 versions sit in 2-4 lines here against 4-8 in the row-major table, and lines are what such a walk is bound by.
TP_HD uint32_t tp_px_tiled_rows(uint32_t H) { return (H + 3u) & ~3u; }
TP_HD uint32_t tp_px_tiled_row_part(uint32_t row, uint32_t pitch) { return (row >> 2) * (pitch << 6) + ((row & 3u) << 5); }   // (pitch a multiple of 8)
TP_HD uint32_t tp_px_tiled_col_part(uint32_t col) { return ((col & ~1u) << 6) + ((col & 1u) << 4); }

// running moments m = {n_odd, r, g, b, q} of the pixels before column c -> record {low, high}
TP_HD void tp_px_pack(const uint32_t m[5], uint64_t rec[2]) {
    rec[0] = (uint64_t)m[1] | ((uint64_t)m[2] << 24) | ((uint64_t)(m[0] & 0xfffu) << 48);
    rec[1] = (uint64_t)m[3] | ((uint64_t)m[4] << 24) | ((uint64_t)(m[0] >> 12) << 58);
}
// the sum of up to TP_PX_MAXSUM records -> the sums of their moments
TP_HD void tp_px_unpack(uint64_t lo, uint64_t hi, uint32_t& nodd, uint32_t& r, uint32_t& g, uint32_t& b, uint64_t& q) {
    r = (uint32_t)lo & 0xffffffu;
    g = (uint32_t)(lo >> 24) & 0xffffffu;
    nodd = (uint32_t)(lo >> 48) + ((uint32_t)(hi >> 58) << 12);
    b = (uint32_t)hi & 0xffffffu;
    q = (hi >> 24) & 0x3ffffffffull;
}
