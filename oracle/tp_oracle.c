/*
 * tp_oracle.c -- CPU ORACLE for the t-pose hot path.  TEST INFRASTRUCTURE ONLY.
 * PARITY UNPINNED (see tp_oracle.h): restates the reference GLSL; no reference golden vectors
 * exist for this path and the GL pipeline cannot run here.
 *
 * Build: gcc -O3 -march=native -ffp-contract=off -fopenmp -shared -fPIC (see oracle/Makefile).
 * -ffp-contract=off matters: every float op below is one IEEE-754 binary32 operation, which is
 * what the HIP path reproduces bit-for-bit.
 */
#include "tp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------
 * dp law.  triangulate: software/triangulate/shader/triangle.vs:60-62
 *            dp = 0.05f; dp /= (1.0f + 4.0f*float(KTriangles)/3000.0f);
 *          warp:        software/warp/shader/triangle.vs:63-65   (9.0f, 1000.0f)
 * ---------------------------------------------------------------------------------------- */
float tpo_dp(int flavour, int NT) {
    volatile float k = (flavour == TPO_WARP) ? 9.0f : 4.0f;
    volatile float m = (flavour == TPO_WARP) ? 1000.0f : 3000.0f;
    volatile float a = k * (float)NT;
    volatile float b = a / m;
    volatile float c = 1.0f + b;
    volatile float dp = 0.05f / c;
    return dp;
}

/* ------------------------------------------------------------------------------------------
 * Vertex stage, triangle.vs:45-84.  Variant i = TDIV, slot = which of the three one-hot
 * model vertices this is.  i = 4*slot+1..4*slot+4 displaces that slot by (+dp,0),(-dp,0),
 * (0,+dp),(0,-dp) in t-pose units BEFORE the x /= RATIO (:66-81).
 *
 * Build-defined part (GL leaves it to the implementation): the viewport transform maps NDC to a
 * W x H raster with x right / y DOWN (image row 0 on top, matching uv = (.5(1+x), .5(1-y)),
 * triangle.vs:84), then positions are snapped to 1/256 pixel, round-half-up, and clamped to
 * [-2^22, 2^23] so that all edge deltas fit in 24 bits.
 * ---------------------------------------------------------------------------------------- */
static inline int32_t snap256(float f) {
    float v = f * 256.0f + 0.5f;
    v = fmaxf(v, -4194304.0f); /* NaN -> lower bound */
    v = fminf(v, 8388608.0f);
    return (int32_t)floorf(v);
}

static inline void vertex_stage(float px, float py, int i, int slot, float dp, float ratio, int W,
                                int H, int32_t* X, int32_t* Y) {
    float Dx = 0.0f, Dy = 0.0f;
    if (i > 0 && (i - 1) / 4 == slot) {
        switch ((i - 1) % 4) {
            case 0: Dx = dp; break;
            case 1: Dx = -dp; break;
            case 2: Dy = dp; break;
            default: Dy = -dp; break;
        }
    }
    float tx = px + Dx;
    float ty = py + Dy;
    float nx = tx / ratio;
    float fx = (nx + 1.0f) * (0.5f * (float)W);
    float fy = (1.0f - ty) * (0.5f * (float)H);
    *X = snap256(fx);
    *Y = snap256(fy);
}

void tpo_variant_vertices(const float* points, const int32_t* tris, int t, int i, float dp,
                          float ratio, int W, int H, int32_t out[6]) {
    for (int s = 0; s < 3; s++) {
        int v = tris[4 * t + s];
        vertex_stage(points[2 * v], points[2 * v + 1], i, s, dp, ratio, W, H, &out[2 * s],
                     &out[2 * s + 1]);
    }
}

/* ------------------------------------------------------------------------------------------
 * Coverage rule (build-defined; GL: "fragment generated when the pixel centre is inside").
 * Integer edge functions on the snapped vertices, evaluated at the pixel centre
 * (256c+128, 256r+128); orientation agnostic (culling is disabled in the reference,
 * software/triangulate/main.cpp:56); ties by the top-left rule.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int64_t a[3], b[3];
    int32_t ox[3], oy[3];
    int tl[3];
    int empty;
    int cmin, cmax, rmin, rmax; /* conservative pixel bbox (inclusive) */
} tri_setup;

static inline int64_t floordiv256(int64_t v) { return v >> 8; }

static void setup_tri(const int32_t xy[6], int W, int H, tri_setup* s) {
    int64_t X0 = xy[0], Y0 = xy[1], X1 = xy[2], Y1 = xy[3], X2 = xy[4], Y2 = xy[5];
    int64_t area2 = (X1 - X0) * (Y2 - Y0) - (Y1 - Y0) * (X2 - X0);
    s->empty = (area2 == 0);
    int64_t sg = area2 > 0 ? 1 : -1;
    const int64_t vx[3] = {X0, X1, X2}, vy[3] = {Y0, Y1, Y2};
    for (int e = 0; e < 3; e++) {
        int j = (e + 1) % 3;
        s->a[e] = -(vy[j] - vy[e]) * sg;
        s->b[e] = (vx[j] - vx[e]) * sg;
        s->ox[e] = (int32_t)vx[e];
        s->oy[e] = (int32_t)vy[e];
        s->tl[e] = (s->a[e] > 0) || (s->a[e] == 0 && s->b[e] > 0);
    }
    int64_t xmin = X0, xmax = X0, ymin = Y0, ymax = Y0;
    if (X1 < xmin) xmin = X1; if (X1 > xmax) xmax = X1;
    if (X2 < xmin) xmin = X2; if (X2 > xmax) xmax = X2;
    if (Y1 < ymin) ymin = Y1; if (Y1 > ymax) ymax = Y1;
    if (Y2 < ymin) ymin = Y2; if (Y2 > ymax) ymax = Y2;
    /* pixel centres 256c+128 within [xmin, xmax] */
    int64_t cmin = floordiv256(xmin - 128 + 255), cmax = floordiv256(xmax - 128);
    int64_t rmin = floordiv256(ymin - 128 + 255), rmax = floordiv256(ymax - 128);
    if (cmin < 0) cmin = 0; if (rmin < 0) rmin = 0;
    if (cmax > W - 1) cmax = W - 1; if (rmax > H - 1) rmax = H - 1;
    s->cmin = (int)cmin; s->cmax = (int)cmax; s->rmin = (int)rmin; s->rmax = (int)rmax;
    if (cmin > cmax || rmin > rmax) s->empty = 1;
}

static inline int inside_edge(int64_t E, int tl) { return E > 0 || (E == 0 && tl); }

int tpo_covered(const int32_t xy[6], int c, int r) {
    tri_setup s;
    setup_tri(xy, 1 << 30, 1 << 30, &s);
    int64_t X0 = xy[0], Y0 = xy[1], X1 = xy[2], Y1 = xy[3], X2 = xy[4], Y2 = xy[5];
    if ((X1 - X0) * (Y2 - Y0) - (Y1 - Y0) * (X2 - X0) == 0) return 0;
    int64_t px = 256 * (int64_t)c + 128, py = 256 * (int64_t)r + 128;
    for (int e = 0; e < 3; e++) {
        int64_t E = s.a[e] * (px - s.ox[e]) + s.b[e] * (py - s.oy[e]);
        if (!inside_edge(E, s.tl[e])) return 0;
    }
    return 1;
}

/* per-fragment visitor over one variant: calls F(c, r) for every covered pixel.  Row-start
 * evaluation + per-pixel increment of the three edge functions (same integers as tpo_covered). */
#define FOR_EACH_FRAGMENT(S, ...)                                                            \
    do {                                                                                     \
        if (!(S).empty) {                                                                    \
            for (int r = (S).rmin; r <= (S).rmax; r++) {                                     \
                int64_t py = 256 * (int64_t)r + 128, px0 = 256 * (int64_t)(S).cmin + 128;    \
                int64_t E0 = (S).a[0] * (px0 - (S).ox[0]) + (S).b[0] * (py - (S).oy[0]);     \
                int64_t E1 = (S).a[1] * (px0 - (S).ox[1]) + (S).b[1] * (py - (S).oy[1]);     \
                int64_t E2 = (S).a[2] * (px0 - (S).ox[2]) + (S).b[2] * (py - (S).oy[2]);     \
                for (int c = (S).cmin; c <= (S).cmax; c++) {                                 \
                    if (inside_edge(E0, (S).tl[0]) && inside_edge(E1, (S).tl[1]) &&          \
                        inside_edge(E2, (S).tl[2])) {                                        \
                        __VA_ARGS__                                                          \
                    }                                                                        \
                    E0 += 256 * (S).a[0]; E1 += 256 * (S).a[1]; E2 += 256 * (S).a[2];        \
                }                                                                            \
            }                                                                                \
        }                                                                                    \
    } while (0)

static inline const uint8_t* texel(const tpo_raster* img, int c, int r) {
    return img->rgba + (size_t)r * img->stride + 4 * (size_t)c;
}

/* wrapping int32 add, as the GL integer atomics behave */
static inline int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }

/* ------------------------------------------------------------------------------------------
 * mode 0, triangle.fs:27-35:  cn[id] += 1;  ca[id].rgb += int(255*texture(...).rgb)
 * (the vertex stage zeroes cn/ca first, triangle.vs:88-91).  texture() of an RGBA8 texel v is
 * v/255.0f; int(255.0f*(v/255.0f)) == v for all 256 values (checked in tests).
 * ---------------------------------------------------------------------------------------- */
void tpo_accumulate_literal(const tpo_raster* img, const float* points, const int32_t* tris,
                            int NT, float dp, float ratio, int count_only, int32_t* cn,
                            int32_t* ca, int nthreads) {
    const int V = 13 * NT;
    (void)nthreads;
#pragma omp parallel for schedule(dynamic, 16) num_threads(nthreads > 0 ? nthreads : 1)
    for (int id = 0; id < V; id++) {
        int i = id / NT, t = id % NT;
        int32_t xy[6];
        tri_setup S;
        tpo_variant_vertices(points, tris, t, i, dp, ratio, img->W, img->H, xy);
        setup_tri(xy, img->W, img->H, &S);
        int32_t n = 0, sr = 0, sg = 0, sb = 0;
        if (count_only) {
            FOR_EACH_FRAGMENT(S, { n = wadd(n, 1); (void)c; });
        } else {
            FOR_EACH_FRAGMENT(S, {
                const uint8_t* p = texel(img, c, r);
                float fr = (float)p[0] / 255.0f, fg = (float)p[1] / 255.0f,
                      fb = (float)p[2] / 255.0f;
                n = wadd(n, 1);
                sr = wadd(sr, (int32_t)(255.0f * fr));
                sg = wadd(sg, (int32_t)(255.0f * fg));
                sb = wadd(sb, (int32_t)(255.0f * fb));
            });
        }
        cn[id] = n;
        if (!count_only) {
            ca[4 * id + 0] = sr; ca[4 * id + 1] = sg; ca[4 * id + 2] = sb; ca[4 * id + 3] = 0;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * mode 1.  triangulate triangle.fs:37-43:
 *     vec3 d = vec3(0);
 *     if(cn[id] > 0) d = 255*texture(..).rgb - vec3(ca[id].rgb/cn[id]);   (integer division)
 *     atomicAdd(ten[id], int(0.5*dot(d, d)));
 * warp triangle.fs:46-53:  d = 255*texture(other image).rgb - vec3(ca[id].rgb)
 * ---------------------------------------------------------------------------------------- */
void tpo_energy_literal(const tpo_raster* img, const float* points, const int32_t* tris, int NT,
                        float dp, float ratio, int flavour, const int32_t* cn, const int32_t* ca,
                        int32_t* ten, int nthreads) {
    const int V = 13 * NT;
    (void)nthreads;
#pragma omp parallel for schedule(dynamic, 16) num_threads(nthreads > 0 ? nthreads : 1)
    for (int id = 0; id < V; id++) {
        int i = id / NT, t = id % NT;
        int32_t xy[6];
        tri_setup S;
        tpo_variant_vertices(points, tris, t, i, dp, ratio, img->W, img->H, xy);
        setup_tri(xy, img->W, img->H, &S);
        float ar = 0, ag = 0, ab = 0;
        int use = 1;
        if (flavour == TPO_TRIANGULATE) {
            if (cn[id] > 0) {
                ar = (float)(ca[4 * id + 0] / cn[id]);
                ag = (float)(ca[4 * id + 1] / cn[id]);
                ab = (float)(ca[4 * id + 2] / cn[id]);
            } else {
                use = 0;
            }
        } else {
            ar = (float)ca[4 * id + 0]; ag = (float)ca[4 * id + 1]; ab = (float)ca[4 * id + 2];
        }
        int32_t e = 0;
        FOR_EACH_FRAGMENT(S, {
            float dx = 0, dy = 0, dz = 0;
            if (use) {
                const uint8_t* p = texel(img, c, r);
                dx = 255.0f * ((float)p[0] / 255.0f) - ar;
                dy = 255.0f * ((float)p[1] / 255.0f) - ag;
                dz = 255.0f * ((float)p[2] / 255.0f) - ab;
            }
            float dot = dx * dx + dy * dy + dz * dz;
            e = wadd(e, (int32_t)(0.5f * dot));
        });
        ten[id] = e;
    }
}

/* ------------------------------------------------------------------------------------------
 * Single-sweep moment formulation.  One pass over the covered pixels of every variant,
 * six int64 moments; tpo_finalize turns them into the reference's two-pass results.
 * ---------------------------------------------------------------------------------------- */
void tpo_moments(const tpo_raster* img, const float* points, const int32_t* tris, int NT, float dp,
                 float ratio, int64_t* mom) {
    const int V = 13 * NT;
    for (int id = 0; id < V; id++) {
        int i = id / NT, t = id % NT;
        int32_t xy[6];
        tri_setup S;
        tpo_variant_vertices(points, tris, t, i, dp, ratio, img->W, img->H, xy);
        setup_tri(xy, img->W, img->H, &S);
        int64_t n = 0, no = 0, sr = 0, sg = 0, sb = 0, q = 0;
        FOR_EACH_FRAGMENT(S, {
            const uint8_t* p = texel(img, c, r);
            int64_t R = p[0], G = p[1], B = p[2];
            n++; no += (R + G + B) & 1; sr += R; sg += G; sb += B; q += R * R + G * G + B * B;
        });
        int64_t* m = mom + 6 * (size_t)id;
        m[0] = n; m[1] = no; m[2] = sr; m[3] = sg; m[4] = sb; m[5] = q;
    }
}

void tpo_finalize(const int64_t* mom, int NT, int flavour, const int32_t* colors, int32_t* ten,
                  int32_t* cn, int32_t* ca, int64_t* ten64) {
    const int V = 13 * NT;
    for (int id = 0; id < V; id++) {
        int t = id % NT;
        const int64_t* m = mom + 6 * (size_t)id;
        int64_t n = m[0], no = m[1], sr = m[2], sg = m[3], sb = m[4], q = m[5];
        int32_t n32 = (int32_t)(uint32_t)(uint64_t)n;
        int64_t ar = 0, ag = 0, ab = 0, E = 0;
        int use = 1;
        if (flavour == TPO_TRIANGULATE) {
            /* ca is an int32 SSBO in the reference: the average uses the wrapped sums */
            int32_t r32 = (int32_t)(uint32_t)(uint64_t)sr, g32 = (int32_t)(uint32_t)(uint64_t)sg,
                    b32 = (int32_t)(uint32_t)(uint64_t)sb;
            if (n32 > 0) { ar = r32 / n32; ag = g32 / n32; ab = b32 / n32; } else use = 0;
            if (ca) { ca[4 * id] = r32; ca[4 * id + 1] = g32; ca[4 * id + 2] = b32; ca[4 * id + 3] = 0; }
        } else {
            ar = colors[4 * t]; ag = colors[4 * t + 1]; ab = colors[4 * t + 2];
        }
        if (use) {
            int64_t a2 = ar * ar + ag * ag + ab * ab;
            int64_t S = q - 2 * (ar * sr + ag * sg + ab * sb) + n * a2; /* sum of |I-a|^2 */
            int64_t nodd = (a2 & 1) ? (n - no) : no;                    /* #fragments with odd |I-a|^2 */
            E = (S - nodd) / 2;                                         /* sum of (|I-a|^2 >> 1) */
        }
        if (ten) ten[id] = (int32_t)(uint32_t)(uint64_t)E;
        if (ten64) ten64[id] = E;
        if (cn) cn[id] = n32;
    }
}

/* ------------------------------------------------------------------------------------------
 * gradient.cs:19-36 (triangulate) / :19-51 (warp): un-normalised central differences scattered
 * with int atomics into gr[NP] (ivec2).  gr is zeroed by the mode-1 vertex stage (triangle.vs:93-98).
 * ---------------------------------------------------------------------------------------- */
void tpo_gradient(const int32_t* ten, const int32_t* tris, int NT, int NP, int32_t* gr) {
    memset(gr, 0, sizeof(int32_t) * 2 * (size_t)NP);
    for (int t = 0; t < NT; t++) {
        for (int s = 0; s < 3; s++) {
            int v = tris[4 * t + s];
            int32_t gx = (int32_t)((uint32_t)ten[(4 * s + 1) * NT + t] - (uint32_t)ten[(4 * s + 2) * NT + t]);
            int32_t gy = (int32_t)((uint32_t)ten[(4 * s + 3) * NT + t] - (uint32_t)ten[(4 * s + 4) * NT + t]);
            gr[2 * v] = wadd(gr[2 * v], gx);
            gr[2 * v + 1] = wadd(gr[2 * v + 1], gy);
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * shift.cs:16-47.  index < 4 never moves; clamp-to-domain zeroes that gradient component
 * BEFORE the step; p -= rate * vec2(gr) / 256 / 256 in float32 (rate = 0.00005 / 0.00003).
 * ---------------------------------------------------------------------------------------- */
void tpo_shift(float* points, int NP, const int32_t* gr, float ratio, float rate) {
    for (int i = 4; i < NP; i++) {
        float gx = (float)gr[2 * i], gy = (float)gr[2 * i + 1];
        float x = points[2 * i], y = points[2 * i + 1];
        if (x <= -ratio) { x = -ratio; gx = 0.0f; }
        else if (x >= ratio) { x = ratio; gx = 0.0f; }
        if (y <= -1.0f) { y = -1.0f; gy = 0.0f; }
        else if (y >= 1.0f) { y = 1.0f; gy = 0.0f; }
        float sx = rate * gx; sx = sx / 256.0f; sx = sx / 256.0f;
        float sy = rate * gy; sy = sy / 256.0f; sy = sy / 256.0f;
        points[2 * i] = x - sx;
        points[2 * i + 1] = y - sy;
    }
}

static void replicate_colors(const int32_t* colors, int NT, int32_t* ca) {
    /* tpose::upload, source/triangulation.hpp:633-641: col[i*NT + k] = colors[k], i = 0..12 */
    for (int i = 0; i < 13; i++)
        for (int k = 0; k < NT; k++)
            memcpy(ca + 4 * ((size_t)i * NT + k), colors + 4 * (size_t)k, 4 * sizeof(int32_t));
}

void tpo_iterate_literal(const tpo_raster* img, float* points, int NP, const int32_t* tris, int NT,
                         int flavour, const int32_t* colors, float dp, float ratio, float rate,
                         int iters, int32_t* ten, int32_t* cn, int32_t* ca, int32_t* gr,
                         int nthreads) {
    if (flavour == TPO_WARP) replicate_colors(colors, NT, ca);
    for (int k = 0; k < iters; k++) {
        tpo_accumulate_literal(img, points, tris, NT, dp, ratio, flavour == TPO_WARP, cn, ca,
                               nthreads);
        tpo_energy_literal(img, points, tris, NT, dp, ratio, flavour, cn, ca, ten, nthreads);
        tpo_gradient(ten, tris, NT, NP, gr);
        tpo_shift(points, NP, gr, ratio, rate);
    }
}

void tpo_iterate_moments(const tpo_raster* img, float* points, int NP, const int32_t* tris, int NT,
                         int flavour, const int32_t* colors, float dp, float ratio, float rate,
                         int iters, int32_t* ten, int32_t* cn, int32_t* ca, int32_t* gr) {
    int64_t* mom = (int64_t*)malloc(sizeof(int64_t) * 6 * 13 * (size_t)NT);
    if (flavour == TPO_WARP) replicate_colors(colors, NT, ca);
    for (int k = 0; k < iters; k++) {
        tpo_moments(img, points, tris, NT, dp, ratio, mom);
        tpo_finalize(mom, NT, flavour, colors, ten, cn, flavour == TPO_WARP ? NULL : ca, NULL);
        tpo_gradient(ten, tris, NT, NP, gr);
        tpo_shift(points, NP, gr, ratio, rate);
    }
    free(mom);
}

/* ------------------------------------------------------------------------------------------
 * source/triangulation.hpp:653-719.  state = {toterr, newerr, relerr, maxerr}; toterr starts 1.
 * ---------------------------------------------------------------------------------------- */
static void sum_err(const int32_t* terr, int NT, float st[4]) {
    st[3] = 0.0f; st[1] = 0.0f;
    for (int i = 0; i < NT; i++) {
        float err = 0.0f;
        err += (float)terr[i];
        if (sqrtf(err) >= st[3]) st[3] = sqrtf(err);
        st[1] += err;
    }
    st[2] = (st[0] - st[1]) / st[0];
    st[0] = st[1];
}

float tpo_geterr(const int32_t* terr, int NT, float st[4]) { sum_err(terr, NT, st); return fabsf(st[2]); }
float tpo_gettoterr(const int32_t* terr, int NT, float st[4]) { sum_err(terr, NT, st); return fabsf(st[0]); }

int tpo_maxerrid(const int32_t* terr, int NT, float st[4]) {
    st[3] = 0.0f;
    int tta = -1;
    for (int i = 0; i < NT; i++) {
        float err = 0.0f;
        err += (float)abs(terr[i]);
        if (sqrtf(err) > st[3]) { st[3] = sqrtf(err); tta = i; }
    }
    return tta;
}
