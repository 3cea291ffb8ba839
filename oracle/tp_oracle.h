/*
 * tp_oracle.h -- CPU ORACLE for the t-pose hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * PARITY UNPINNED: the reference's implementation of this path is GLSL executed by an
 * OpenGL 4.6 driver through the un-vendored TinyEngine; it cannot be built or run here,
 * and the reference holds no golden vectors for it.  This file is a
 * plain-C restatement of the reference shaders' arithmetic; where OpenGL leaves behaviour
 * implementation-defined (rasteriser coverage, sub-pixel snapping) the rule is fixed here
 * and documented in DESIGN.md.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call into this
 * library; the product (libtpose_hip.so) never links or loads it.
 *
 * Reference files restated (relative to the reference tree):
 *   software/triangulate/shader/triangle.vs:45-98   vertex stage, 13 variants
 *   software/triangulate/shader/triangle.fs:27-43   mode 0 accumulate, mode 1 energy
 *   software/triangulate/shader/gradient.cs:19-36   per-vertex gradient scatter
 *   software/triangulate/shader/shift.cs:16-47      clamped gradient step
 *   software/warp/shader/triangle.vs:48-100, triangle.fs:40-53, gradient.cs:19-51, shift.cs:16-47
 *   source/triangulation.hpp:628-719                upload / geterr / gettoterr / maxerrid
 */
#ifndef TP_ORACLE_H
#define TP_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { TPO_TRIANGULATE = 0, TPO_WARP = 1 };

/* raster description: RGBA8 plane, row 0 = top, stride in bytes */
typedef struct {
    const uint8_t* rgba;
    size_t stride;
    int W, H;
} tpo_raster;

/* dp law (triangle.vs:60-62 / warp triangle.vs:63-65), float32 arithmetic */
float tpo_dp(int flavour, int NT);

/* vertex stage: snapped 24.8 fixed-point raster coordinates (x right, y down) of the three
 * vertices of variant i (0..12) of triangle t.  out = {X0,Y0,X1,Y1,X2,Y2}. */
void tpo_variant_vertices(const float* points, const int32_t* tris, int t, int i, float dp,
                          float ratio, int W, int H, int32_t out[6]);

/* coverage of pixel (c,r) (centre sampled) by the snapped triangle; top-left rule,
 * orientation agnostic.  Literal per-pixel evaluation of the three edge functions. */
int tpo_covered(const int32_t xy[6], int c, int r);

/* LITERAL reference form: per-fragment loops, 13 variants, int32 accumulators (wrapping).
 * mode 0 (triangle.fs:27-35 / warp :40-44).  cn[13NT], ca[4*13NT] (ivec4, w untouched=0).
 * count_only != 0 -> warp flavour (cn only, ca untouched). */
void tpo_accumulate_literal(const tpo_raster* img, const float* points, const int32_t* tris,
                            int NT, float dp, float ratio, int count_only, int32_t* cn,
                            int32_t* ca, int nthreads);

/* mode 1 (triangle.fs:37-43 / warp :46-53).  ten[13NT].  flavour selects the reference colour:
 * TRIANGULATE: ca/cn of the variant (integer division), 0 energy if cn==0; WARP: ca as stored. */
void tpo_energy_literal(const tpo_raster* img, const float* points, const int32_t* tris, int NT,
                        float dp, float ratio, int flavour, const int32_t* cn, const int32_t* ca,
                        int32_t* ten, int nthreads);

/* SINGLE-SWEEP moment formulation (SURVEY.md section 7 obs. 3): per variant
 * mom[6] = {n, n_odd, sum r, sum g, sum b, sum (r^2+g^2+b^2)} as int64; one pass over pixels. */
void tpo_moments(const tpo_raster* img, const float* points, const int32_t* tris, int NT, float dp,
                 float ratio, int64_t* mom);

/* moments -> reference-layout outputs.  colors: stored colours ivec4[NT] (warp) or NULL.
 * ten[13NT], cn[13NT], ca[4*13NT] (ca written for TRIANGULATE only), ten64 optional. */
void tpo_finalize(const int64_t* mom, int NT, int flavour, const int32_t* colors, int32_t* ten,
                  int32_t* cn, int32_t* ca, int64_t* ten64);

/* gradient.cs: gr[2*NP] (ivec2), zeroed then scattered; int32 wrapping */
void tpo_gradient(const int32_t* ten, const int32_t* tris, int NT, int NP, int32_t* gr);

/* shift.cs: in-place on points[2*NP]; vertices 0..3 never move */
void tpo_shift(float* points, int NP, const int32_t* gr, float ratio, float rate);

/* one or more full grad-iters in literal reference form: accumulate (mode 0) -> energy (mode 1)
 * -> gradient -> shift.  On return ten/cn/ca/gr hold the LAST iteration's (pre-shift) buffers.
 * colors: ivec4[NT] stored colours for WARP (replicated x13 like tpose::upload), NULL otherwise. */
void tpo_iterate_literal(const tpo_raster* img, float* points, int NP, const int32_t* tris, int NT,
                         int flavour, const int32_t* colors, float dp, float ratio, float rate,
                         int iters, int32_t* ten, int32_t* cn, int32_t* ca, int32_t* gr,
                         int nthreads);

/* same, through the moment formulation */
void tpo_iterate_moments(const tpo_raster* img, float* points, int NP, const int32_t* tris, int NT,
                         int flavour, const int32_t* colors, float dp, float ratio, float rate,
                         int iters, int32_t* ten, int32_t* cn, int32_t* ca, int32_t* gr);

/* triangulation.hpp:653-719 error bookkeeping.  state = {toterr, newerr, relerr, maxerr} */
float tpo_geterr(const int32_t* terr, int NT, float state[4]);
float tpo_gettoterr(const int32_t* terr, int NT, float state[4]);
int tpo_maxerrid(const int32_t* terr, int NT, float state[4]);

#ifdef __cplusplus
}
#endif
#endif
