// tp_eval.hip -- tp_evaluate_triangles: the energy a triangle WOULD have, for triangles that are not in the uploaded mesh.
//
// The reference's convergence step evaluates its flip set by making it real (software/triangulate/main.cpp:233-306): flip every chosen
// edge on the host, upload the whole topology, computecolors + doenergy over every triangle and every variant, read the energies back --
// and look at two numbers per flipped edge.  The energy of a triangle depends on its own pixels only (triangle.fs:27-43: its mean colour
// from mode 0, the squared distances to it from mode 1), so those two numbers are the base energies of two hypothetical triangles at the
// current positions: this entry computes exactly them, from the same pixel-record table and with the same exact arithmetic as the
// grad-iter kernels (whole-line walkers, one record per line and row, signed sum of three line sums -> moments -> energy), without an
// upload and without touching any buffer of the context.  One 64-lane wave per triangle: the lanes take the rows of each of its three
// edge lines in residue classes, as the lane-items of k_persist do.
#include "tp_context.h"
#include "tp_persist.h"

namespace {

// variants: null (all base variants), or per triangle 0..12 -- variant i > 0 displaces vertex (i - 1) / 4 of the triple by move (i - 1) % 4 + 1
// of vw.dp, as the vertex stage of the sweep does (triangle.vs:66-78)
__global__ __launch_bounds__(64) void k_eval_triangles(tp_view vw, const float2* points, int NP, const int32_t* tri3, const int32_t* variants, int n,
                                                       const char* table, int px_pitch, int32_t* energy, int32_t* count) {
    const int t = blockIdx.x, lane = threadIdx.x;
    if (t >= n) return;
    const int variant = variants ? variants[t] : 0;
    int32_t X[3], Y[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        int v = tri3[3 * (size_t)t + k];
        v = v < 0 ? 0 : (v >= NP ? NP - 1 : v);   // (the entry has checked them; nothing is read out of bounds whatever comes)
        const float2 p = points[v];
        tp_vertex_stage(p.x, p.y, variant, k, vw, X[k], Y[k]);
    }
    unsigned long long S[3][PK_SUM_WORDS];
    int dir[3];
    const uint32_t magic = pk_magic(64);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int j = k == 2 ? 0 : k + 1;
        tp_line ln;
        tp_setup_line(X[k], Y[k], X[j], Y[j], vw.H, ln);
        pk_walker wk;
        wk.x = ln.x; wk.s = ln.s; wk.ra = ln.ra; wk.rb = ln.rb;
        dir[k] = (Y[j] > Y[k]) - (Y[j] < Y[k]);
        pk_acc a;
        a.xs = 0; a.nodd = 0; a.r = 0; a.g = 0; a.b = 0; a.q = 0;
        pk_rows r = pk_lane_rows(wk, lane, 64, magic, px_pitch);
        pk_walk_rows<4>(r, table, vw.W, a);
        // the wave's sums of this line (a whole line's sums fit the fields: tp_persist.h, pk_fold_words)
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            a.xs += (uint32_t)__shfl_xor((int)a.xs, d); a.nodd += (uint32_t)__shfl_xor((int)a.nodd, d);
            a.r += (uint64_t)__shfl_xor((long long)a.r, d); a.g += (uint64_t)__shfl_xor((long long)a.g, d);
            a.b += (uint64_t)__shfl_xor((long long)a.b, d); a.q += (uint64_t)__shfl_xor((long long)a.q, d);
        }
        pk_fold_words(a, S[k]);
    }
    if (lane == 0) {
        // signs as pk_signed_moments takes them: how each edge runs down the raster, times whatever makes the pixel count non-negative
        const int64_t cnt = (int64_t)dir[0] * (int64_t)(uint32_t)S[0][0] + (int64_t)dir[1] * (int64_t)(uint32_t)S[1][0] + (int64_t)dir[2] * (int64_t)(uint32_t)S[2][0];
        const int sg = cnt < 0 ? -1 : 1;
        const tp_moments mm = pk_moments3(dir[0] * sg, S[0], dir[1] * sg, S[1], dir[2] * sg, S[2]);
        energy[t] = tp_wrap32(tp_energy_triangulate(mm));
        if (count) count[t] = tp_wrap32(mm.n);
    }
}

// tp_selftest_variant: the packed form of a variant's moments and energy (tp_persist.h: pk_signed_packed, pk_energy_var -- what P6 of
// k_persist runs, float reciprocal and all) beside the general 64-bit form, on line sums the caller made up
__global__ void k_selftest_variant(const unsigned long long* sums, const int32_t* meta, int n, int32_t* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned long long S[3 * PK_SUM_STRIDE];
    int32_t dir[3];
    for (int k = 0; k < 3; k++) {
        for (int q = 0; q < PK_SUM_WORDS; q++) S[k * PK_SUM_STRIDE + q] = sums[((size_t)i * 3 + k) * PK_SUM_WORDS + q];
        dir[k] = meta[8 * (size_t)i + k];
    }
    pk_view V;
    V.sums = S; V.ldir = dir;
    const int flips = meta[8 * (size_t)i + 3], flavour = meta[8 * (size_t)i + 4];
    const pk_i4 col = {meta[8 * (size_t)i + 5], meta[8 * (size_t)i + 6], meta[8 * (size_t)i + 7], 0};
    const pk_var v = pk_signed_packed(V, 0, 1, 2, flips);
    const tp_moments mm = pk_signed_moments(V, 0, 1, 2, flips);
    int32_t* o = out + 10 * (size_t)i;
    o[0] = (int32_t)v.n; o[1] = (int32_t)v.nodd; o[2] = (int32_t)v.r; o[3] = (int32_t)v.g; o[4] = (int32_t)v.b; o[5] = (int32_t)(uint32_t)v.q; o[6] = (int32_t)(uint32_t)(v.q >> 32);
    o[7] = pk_energy_var(v, flavour, col);
    o[8] = pk_energy(mm, flavour, col);
    o[9] = mm.n == (int64_t)v.n && mm.nodd == (int64_t)v.nodd && mm.sr == (int64_t)v.r && mm.sg == (int64_t)v.g && mm.sb == (int64_t)v.b && mm.q == (int64_t)v.q;
}

}  // namespace

using namespace tpctx;

extern "C" int tp_selftest_variant(tp_context* c, const uint64_t* sums, const int32_t* meta, int n, int32_t* out) {
    api_guard api_lock;
    if (!c || !sums || !meta || !out || n < 0) return TP_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    unsigned long long* ds = nullptr; int32_t *dm = nullptr, *dout = nullptr;
    hipError_t e = dev_alloc(&ds, (size_t)n * 12);   // (one exit: the buffers are freed on every path)
    if (e == hipSuccess) e = dev_alloc(&dm, (size_t)n * 8);
    if (e == hipSuccess) e = dev_alloc(&dout, (size_t)n * 10);
    if (e == hipSuccess) e = hipMemcpy(ds, sums, sizeof(uint64_t) * 12 * (size_t)n, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dm, meta, sizeof(int32_t) * 8 * (size_t)n, hipMemcpyHostToDevice);
    if (e == hipSuccess && n > 0) {
        hipLaunchKernelGGL(k_selftest_variant, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, ds, dm, n, dout);
        e = hipStreamSynchronize(c->stream);
    }
    if (e == hipSuccess) e = hipMemcpy(out, dout, sizeof(int32_t) * 10 * (size_t)n, hipMemcpyDeviceToHost);
    hipFree(ds); hipFree(dm); hipFree(dout);
    if (e != hipSuccess) return fail(c, TP_ERR_HIP, "selftest_variant: %s", hipGetErrorString(e));
    return TP_OK;
}

extern "C" int tp_evaluate_triangles(tp_context* c, int slot, int n, const int32_t* vertices, const int32_t* variants, int32_t* energy, int32_t* count) {
    api_guard api_lock;
    if (!c) return TP_ERR_INVALID;
    if (n < 0 || (n > 0 && (!vertices || !energy))) return fail(c, TP_ERR_INVALID, "evaluate_triangles: bad arguments");
    if (slot != TP_IMAGE_A && slot != TP_IMAGE_B) return fail(c, TP_ERR_INVALID, "bad image slot %d", slot);
    if (!c->uploaded) return fail(c, TP_ERR_STATE, "evaluate_triangles before upload");
    if (!c->px_pitch || !c->px[slot]) return fail(c, TP_ERR_STATE, "evaluate_triangles: no pixel-record table of image %d (no image yet, or a raster beyond %d columns or rows)", slot, TP_PX_MAXW);
    for (int k = 0; k < 3 * n; k++)
        if (vertices[k] < 0 || vertices[k] >= c->NP) return fail(c, TP_ERR_INVALID, "evaluate_triangles: vertex %d of triangle %d is %d (NP=%d)", k % 3, k / 3, vertices[k], c->NP);
    for (int k = 0; variants && k < n; k++)
        if (variants[k] < 0 || variants[k] > 12) return fail(c, TP_ERR_INVALID, "evaluate_triangles: variant %d of triangle %d", variants[k], k);
    if (n == 0) return TP_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    if (int rc = settle_persistent(c)) return rc;   // (the positions are those behind everything called so far)
    // one pinned block {vertices, variants in | energies, counts out}, grown as needed: the kernel reads its list and writes its answers
    // across the link itself (a few words per wave) -- one launch and one wait per call, no copies
    const size_t words = (size_t)6 * n;
    if (words > c->eval_cap) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (c->eval_host) hipHostFree(c->eval_host);
        c->eval_host = nullptr; c->eval_cap = 0;
        const size_t cap = words + words / 2 + 1024;
        HIP_TRY(c, hipHostMalloc((void**)&c->eval_host, cap * sizeof(int32_t), hipHostMallocDefault));
        c->eval_cap = cap;
    }
    memcpy(c->eval_host, vertices, sizeof(int32_t) * 3 * (size_t)n);
    if (variants) memcpy(c->eval_host + 3 * (size_t)n, variants, sizeof(int32_t) * (size_t)n);
    tp_view vw;
    vw.dp = resolve_dp(c, TP_TRIANGULATE, c->dp_override);   // (the dp of the context's sweeps: tp_set_dp, or the reference's law at the uploaded NT)
    vw.ratio = c->ratio; vw.halfW = 0.5f * (float)c->W; vw.halfH = 0.5f * (float)c->H; vw.W = c->W; vw.H = c->H;
    int32_t* out = c->eval_host + 4 * (size_t)n;
    hipLaunchKernelGGL(k_eval_triangles, dim3((unsigned)n), dim3(64), 0, c->stream, vw, (const float2*)c->points, c->NP, (const int32_t*)c->eval_host,
                       variants ? (const int32_t*)(c->eval_host + 3 * (size_t)n) : (const int32_t*)nullptr, n,
                       reinterpret_cast<const char*>(c->px[slot]), c->px_pitch, out, out + n);
    c->tail_is_finish = false;
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, wait_stream(c->stream));
    memcpy(energy, c->eval_host + 4 * (size_t)n, sizeof(int32_t) * (size_t)n);
    if (count) memcpy(count, c->eval_host + 5 * (size_t)n, sizeof(int32_t) * (size_t)n);
    return TP_OK;
}
