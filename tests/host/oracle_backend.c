/* oracle_backend.c -- TEST INFRASTRUCTURE: the subset of the C ABI (include/tpose_hip.h) that the
 * headless harnesses use, implemented on the CPU oracle (oracle/tp_oracle.c, literal reference form).
 * Linking a harness against this instead of libtpose_hip.so gives a schedule-level reference run:
 * tests compare the .tri files the two builds write, byte for byte.  Never shipped, never linked by
 * the product. */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/tpose_hip.h"
#include "../../oracle/tp_oracle.h"

struct tp_context {
    int W, H;
    float ratio, dp_override;
    uint8_t* img[2];
    float* points; int NP;
    int32_t* tris; int NT;
    int32_t *ca, *cn, *ten, *gr;
    int acc_flavour, acc_slot;
};

static const char* g_err = "";

int tp_abi_version(void) { return TP_ABI_VERSION; }
const char* tp_last_error(const tp_context* c) { (void)c; return g_err; }

int tp_create(int device, int width, int height, tp_context** out) {
    (void)device;
    tp_context* c = (tp_context*)calloc(1, sizeof *c);
    c->W = width; c->H = height; c->ratio = (float)width / (float)height;
    *out = c;
    return TP_OK;
}
int tp_destroy(tp_context* c) {
    if (!c) return TP_OK;
    free(c->img[0]); free(c->img[1]); free(c->points); free(c->tris); free(c->ca); free(c->cn); free(c->ten); free(c->gr);
    free(c);
    return TP_OK;
}
int tp_set_ratio(tp_context* c, float r) { c->ratio = r; return TP_OK; }
int tp_set_image(tp_context* c, int slot, const uint8_t* rgba, size_t stride) {
    free(c->img[slot]);
    c->img[slot] = (uint8_t*)malloc((size_t)c->W * c->H * 4);
    for (int y = 0; y < c->H; y++) memcpy(c->img[slot] + (size_t)y * c->W * 4, rgba + (size_t)y * stride, (size_t)c->W * 4);
    return TP_OK;
}
int tp_upload(tp_context* c, const float* points, int NP, const int32_t* tris, int NT, const int32_t* colors) {
    if (NT != c->NT) {
        int32_t* ca = (int32_t*)calloc((size_t)13 * NT * 4, 4);
        if (c->ca && c->NT) {  /* colacc persists across uploads in the reference (one big SSBO) */
            /* variant-major layout changes with NT: the stale contents are never read before the
             * next mode-0 pass in triangulate, and warp always uploads colours */
        }
        free(c->ca); c->ca = ca;
        free(c->cn); c->cn = (int32_t*)calloc((size_t)13 * NT, 4);
        free(c->ten); c->ten = (int32_t*)calloc((size_t)13 * NT, 4);
    }
    if (NP != c->NP) { free(c->gr); c->gr = (int32_t*)calloc((size_t)2 * NP, 4); }
    free(c->points); c->points = (float*)malloc(sizeof(float) * 2 * NP); memcpy(c->points, points, sizeof(float) * 2 * NP);
    free(c->tris); c->tris = (int32_t*)malloc(sizeof(int32_t) * 4 * NT); memcpy(c->tris, tris, sizeof(int32_t) * 4 * NT);
    c->NP = NP; c->NT = NT;
    if (colors)
        for (int i = 0; i < 13; i++) memcpy(c->ca + (size_t)4 * i * NT, colors, sizeof(int32_t) * 4 * NT);
    return TP_OK;
}
static float the_dp(tp_context* c, int flavour) { return c->dp_override > 0 ? c->dp_override : tpo_dp(flavour, c->NT); }
int tp_accumulate(tp_context* c, int flavour, int slot) {
    tpo_raster r = {c->img[slot], (size_t)c->W * 4, c->W, c->H};
    tpo_accumulate_literal(&r, c->points, c->tris, c->NT, the_dp(c, flavour), c->ratio, flavour == TP_WARP, c->cn, c->ca, 1);
    c->acc_flavour = flavour; c->acc_slot = slot;
    return TP_OK;
}
int tp_energy(tp_context* c, int flavour) {
    tpo_raster r = {c->img[c->acc_slot], (size_t)c->W * 4, c->W, c->H};
    tpo_energy_literal(&r, c->points, c->tris, c->NT, the_dp(c, flavour), c->ratio, flavour, c->cn, c->ca, c->ten, 1);
    return TP_OK;
}
int tp_shift(tp_context* c, float rate) {
    tpo_gradient(c->ten, c->tris, c->NT, c->NP, c->gr);
    tpo_shift(c->points, c->NP, c->gr, c->ratio, rate);
    return TP_OK;
}
int tp_retrieve(tp_context* c, int what, void* dst, size_t count) {
    switch (what) {
        case TP_BUF_TENERGY: memcpy(dst, c->ten, count * 4); break;
        case TP_BUF_COLNUM: memcpy(dst, c->cn, count * 4); break;
        case TP_BUF_COLACC: memcpy(dst, c->ca, count * 4); break;
        case TP_BUF_POINTS: memcpy(dst, c->points, count * 4); break;
        case TP_BUF_GRADIENT: memcpy(dst, c->gr, count * 4); break;
        case TP_BUF_PENERGY: memset(dst, 0, count * 4); break;
        default: return TP_ERR_INVALID;
    }
    return TP_OK;
}

/* display pass (triangle.fs mode 2 / software/view): per-pixel ownership by the oracle's coverage test */
int tp_render(tp_context* c, int source, const float* points, uint8_t* dst, size_t stride) {
    const float* pts = points ? points : c->points;
    for (int y = 0; y < c->H; y++)
        for (int x = 0; x < c->W; x++) {
            uint8_t* p = dst + (size_t)y * stride + 4 * (size_t)x;
            p[0] = p[1] = p[2] = 0; p[3] = 255;
        }
    for (int t = 0; t < c->NT; t++) {
        int32_t xy[6];
        tpo_variant_vertices(pts, c->tris, t, 0, 0.0f, c->ratio, c->W, c->H, xy);
        uint8_t col[3];
        const int32_t* a = c->ca + (size_t)4 * t;
        if (source == TP_RENDER_AVERAGE) {
            if (c->cn[t] == 0) continue;
            for (int k = 0; k < 3; k++) {
                float f = ((float)a[k] / (float)c->cn[t]) / 255.0f;
                f = f < 0 ? 0 : f > 1 ? 1 : f;
                col[k] = (uint8_t)(int)(f * 255.0f + 0.5f);
            }
        } else
            for (int k = 0; k < 3; k++) col[k] = (uint8_t)(a[k] < 0 ? 0 : a[k] > 255 ? 255 : a[k]);
        int ymin = xy[1], ymax = xy[1], xmin = xy[0], xmax = xy[0];
        for (int k = 1; k < 3; k++) {
            if (xy[2 * k] < xmin) xmin = xy[2 * k]; if (xy[2 * k] > xmax) xmax = xy[2 * k];
            if (xy[2 * k + 1] < ymin) ymin = xy[2 * k + 1]; if (xy[2 * k + 1] > ymax) ymax = xy[2 * k + 1];
        }
        int r0 = (ymin >> 8) - 1, r1 = (ymax >> 8) + 1, c0 = (xmin >> 8) - 1, c1 = (xmax >> 8) + 1;
        if (r0 < 0) r0 = 0; if (r1 > c->H - 1) r1 = c->H - 1; if (c0 < 0) c0 = 0; if (c1 > c->W - 1) c1 = c->W - 1;
        for (int r = r0; r <= r1; r++)
            for (int x = c0; x <= c1; x++)
                if (tpo_covered(xy, x, r)) memcpy(dst + (size_t)r * stride + 4 * (size_t)x, col, 3);
    }
    return TP_OK;
}

int tp_retrieve_many(tp_context* c, int n, const int* what, void* const* dst, const size_t* count) {
    for (int k = 0; k < n; k++) {
        int rc = tp_retrieve(c, what[k], dst[k], count[k]);
        if (rc) return rc;
    }
    return TP_OK;
}

void tp_default_params(int flavour, tp_params* p) {
    p->flavour = flavour;
    p->image_slot = flavour == TP_WARP ? TP_IMAGE_B : TP_IMAGE_A;
    p->rate = flavour == TP_WARP ? 0.00003f : 0.00005f;
    p->dp = 0.0f;
}

/* fused grad-iters = the three piecewise steps in the reference's order */
int tp_iterate(tp_context* c, const tp_params* p, int n_iters) {
    for (int k = 0; k < n_iters; k++) {
        tp_accumulate(c, p->flavour, p->image_slot);
        tp_energy(c, p->flavour);
        tp_shift(c, p->rate);
    }
    return TP_OK;
}

/* the reference's loop up to its convergence test, frame by frame (software/warp/main.cpp:220-231) */
int tp_iterate_until(tp_context* c, const tp_params* p, int max_frames, double threshold, float* toterr, int* frames, float* relerr) {
    float tot = *toterr, rel = 0.0f;
    int done = 0;
    while (done < max_frames) {
        tp_iterate(c, p, 1);
        done++;
        float newerr = 0.0f;
        for (int i = 0; i < c->NT; i++) { float err = 0.0f; err += (float)c->ten[i]; newerr += err; }
        rel = (tot - newerr) / tot;
        tot = newerr;
        if ((double)fabsf(rel) < threshold) break;
    }
    *toterr = tot; *frames = done;
    if (relerr) *relerr = rel;
    return TP_OK;
}

/* the frame loop with the host in it: frame by frame, the caller's verdict after each (the frame has just run: everything stands as it left it) */
int tp_iterate_frames(tp_context* c, const tp_params* p, int max_frames, tp_frame_fn fn, void* user, int* frames) {
    int done = 0;
    while (done < max_frames) {
        tp_iterate(c, p, 1);
        const int verdict = fn(user, done, c->ten, c->points);
        done++;
        if (verdict != TP_FRAME_GO_ON) break;
    }
    *frames = done;
    return TP_OK;
}

/* band split (tp_band_attach): the stand-in runs every band's descents whole -- the same results, nothing shared */
size_t tp_band_mailbox_bytes(int points, int triangles) { (void)triangles; return (size_t)(points > 0 ? points : 0) * 64 + 64; }
int tp_band_attach(tp_context* c, int band, int n_bands, void* const* mailboxes, size_t bytes_each, int points, int triangles, int patches_per_band) {
    (void)c; (void)band; (void)n_bands; (void)mailboxes; (void)bytes_each; (void)points; (void)triangles; (void)patches_per_band;
    return TP_OK;
}
int tp_band_mailbox_alloc(tp_context* c, size_t bytes, void** out) { (void)c; if (!out) return TP_ERR_INVALID; *out = calloc(bytes ? bytes : 1, 1); return *out ? TP_OK : TP_ERR_CAPACITY; }
int tp_band_mailbox_free(tp_context* c, void* box) { (void)c; free(box); return TP_OK; }
int tp_band_mailbox_export(tp_context* c, void* box, void* handle) { (void)c; (void)box; if (handle) memset(handle, 0, TP_MAILBOX_HANDLE_BYTES); return TP_OK; }
int tp_band_mailbox_import(tp_context* c, const void* handle, void** box) { (void)c; (void)handle; if (box) *box = NULL; return TP_ERR_STATE; }
int tp_band_mailbox_close(tp_context* c, void* box) { (void)c; (void)box; return TP_OK; }
/* hypothetical triangles: the oracle's single-sweep moments of a mesh made of just them, variant 0 (triangulate flavour) */
int tp_evaluate_triangles(tp_context* c, int slot, int n, const int32_t* vertices, const int32_t* variants, int32_t* energy, int32_t* count) {
    if (n <= 0) return TP_OK;
    int32_t* tris = (int32_t*)calloc((size_t)4 * n, 4);
    for (int k = 0; k < n; k++) { tris[4 * k] = vertices[3 * k]; tris[4 * k + 1] = vertices[3 * k + 1]; tris[4 * k + 2] = vertices[3 * k + 2]; }
    int64_t* mom = (int64_t*)calloc((size_t)13 * n * 6, 8);
    int32_t* ten = (int32_t*)calloc((size_t)13 * n, 4);
    int32_t* cn = (int32_t*)calloc((size_t)13 * n, 4);
    int32_t* ca = (int32_t*)calloc((size_t)13 * n * 4, 4);
    tpo_raster r = {c->img[slot], (size_t)c->W * 4, c->W, c->H};
    tpo_moments(&r, c->points, tris, n, the_dp(c, TP_TRIANGULATE), c->ratio, mom);   /* (the dp of the context's sweeps: at the UPLOADED NT) */
    tpo_finalize(mom, n, TPO_TRIANGULATE, NULL, ten, cn, ca, NULL);
    for (int k = 0; k < n; k++) {
        const int i = variants ? variants[k] : 0;
        energy[k] = ten[(size_t)i * n + k];
        if (count) count[k] = cn[(size_t)i * n + k];
    }
    free(tris); free(mom); free(ten); free(cn); free(ca);
    return TP_OK;
}
int tp_prepare(tp_context* c, const tp_params* p) { (void)c; (void)p; return TP_OK; }
int tp_get_info(tp_context* c, int what, int64_t* value) { (void)c; (void)what; if (value) *value = 0; return TP_OK; }
int tp_synchronize(tp_context* c) { (void)c; return TP_OK; }
