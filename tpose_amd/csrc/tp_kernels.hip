// tp_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the t-pose hot path.  wave64 only.
//
// One grad-iter of the reference = two instanced draws of 13*NT triangles (mode 0: 4 same-address
// int atomics per fragment, mode 1: one) + gradient.cs + shift.cs
// (software/triangulate/main.cpp:121-155).  Here the work is organised around EDGE LINES
// (tp_raster.h, "edge-centric form"): a variant's pixel moments are the signed sum of three line
// sums W(e) = sum over the line's rows of the FULL-ROW prefix sum of the pixel moments at the line's
// crossing column.  The raster does not change between iterations (the reference uploads its texture
// once, software/triangulate/main.cpp:74), so its row prefix sums are a per-image table built by
// tp_set_image (k_prefix, 8 bytes per pixel): an iteration touches one table record per (line, row)
// -- work proportional to the total edge length, not to the raster area.
//
//   k_prefix      per image: 32-byte records per four pixels -- the packed moments of the pixels left of the group and
//                 the group's r, g, b bytes (tp_raster.h, "Per-image row prefix table")
//   k_lines       THE hot kernel (tp_iterate, tp_accumulate): per edge line (nine per undirected edge) the vertex stage of
//                 both endpoints, the line's whole-line 24.40 walker, one table record per row -> line sums `wline`
//   k_finalize    per variant: signed sum of its three line sums -> exact moments -> `colnum`, `colacc`,
//                 `tenergy` (reference layout)
//   k_shift       gradient.cs gathered per vertex + shift.cs
//   k_update      k_finalize + k_shift in one launch, organised by vertex (tp_iterate)
#include "tp_kernels.h"
#include <hip/hip_ext.h>

// debug flavour of the library only (tools/kernel_timeline.py): thread 0 of the first 4096 workgroups of a kernel
// stamps the 100 MHz wall clock at its phase boundaries; region 0 k_lines, 2 k_update
#ifdef TPOSE_DEBUG
#define TP_STAMP(region, k) do { if (threadIdx.x == 0 && blockIdx.x < 4096) L.dbg[((region) * 4096 + blockIdx.x) * 8 + (k)] = wall_clock64(); } while (0)
#else
#define TP_STAMP(region, k) do { } while (0)
#endif

// ------------------------------------------------------------------------------------------------
// Per-image row prefix table (built once per tp_set_image): layout and arithmetic in tp_raster.h.
// ------------------------------------------------------------------------------------------------
static_assert(TP_MAX_RASTER <= 16384, "packing of the prefix table records");

__device__ __forceinline__ void px_moments5(uint32_t rgba, uint32_t m[5]) {
    const uint32_t r = rgba & 0xffu, g = (rgba >> 8) & 0xffu, b = (rgba >> 16) & 0xffu;
    m[0] += (r + g + b) & 1u; m[1] += r; m[2] += g; m[3] += b; m[4] += r * r + g * g + b * b;
}

// one 256-thread workgroup per row; thread t owns the groups [t C, (t + 1) C), C = ceil(groups / 256)
__global__ __launch_bounds__(256) void k_prefix(const uint8_t* img, int pitch, int W, int prefix_pitch, uint4* P) {
    __shared__ uint32_t wave_total[4][5];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NG = tp_prefix_groups(W);
    const int C = (NG + 255) / 256;
    const int g0 = min(NG, tid * C), g1 = min(NG, g0 + C);
    const uint32_t* src = reinterpret_cast<const uint32_t*>(img + (size_t)row * pitch);
    uint32_t own[5] = {0, 0, 0, 0, 0};
    for (int c = 4 * g0; c < min(W, 4 * g1); c++) px_moments5(src[c], own);
    uint32_t inc[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        uint32_t v = own[k];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)v, o);
            if (lane >= o) v += up;
        }
        inc[k] = v;
        if (lane == 63) wave_total[wave][k] = v;
    }
    __syncthreads();
    uint32_t run[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        uint32_t before = 0;
        for (int w = 0; w < wave; w++) before += wave_total[w][k];
        run[k] = before + inc[k] - own[k];
    }
    uint4* dst = P + (size_t)row * prefix_pitch * 2;
    for (int g = g0; g < g1; g++) {
        uint32_t px[4] = {0, 0, 0, 0}, rec[TP_PFX_WORDS];
        const int npx = min(4, W - 4 * g);
        for (int i = 0; i < npx; i++) px[i] = src[4 * g + i];
        tp_prefix_pack(run, px, npx, rec);
        dst[2 * g] = make_uint4(rec[0], rec[1], rec[2], rec[3]);
        dst[2 * g + 1] = make_uint4(rec[4], rec[5], rec[6], rec[7]);
        for (int i = 0; i < npx; i++) px_moments5(px[i], run);
    }
}
void tp_launch_prefix_table(const uint8_t* img, int pitch, int W, int H, int prefix_pitch, uint4* P, hipStream_t s) {
    hipLaunchKernelGGL(k_prefix, dim3((unsigned)H), dim3(256), 0, s, img, pitch, W, prefix_pitch, P);
}

// the same sums in pixel records (tp_raster.h, "Pixel records": 16 bytes per pixel column; rasters up to 4096 columns) --
// what the persistent kernel reads.  One 256-thread workgroup per row; thread t owns the columns [t C, (t + 1) C)
__global__ __launch_bounds__(256) void k_prefix_px(const uint8_t* img, int pitch, int W, int px_pitch, uint4* P, uint4* Pt) {
    __shared__ uint32_t wave_total[4][5];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NC = W + 1;   // columns c = 0..W
    const int C = (NC + 255) / 256;
    const int c0 = min(NC, tid * C), c1 = min(NC, c0 + C);
    const uint32_t* src = reinterpret_cast<const uint32_t*>(img + (size_t)row * pitch);
    uint32_t own[5] = {0, 0, 0, 0, 0};
    for (int c = c0; c < min(W, c1); c++) px_moments5(src[c], own);
    uint32_t inc[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        uint32_t v = own[k];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)v, o);
            if (lane >= o) v += up;
        }
        inc[k] = v;
        if (lane == 63) wave_total[wave][k] = v;
    }
    __syncthreads();
    uint32_t run[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        uint32_t before = 0;
        for (int w = 0; w < wave; w++) before += wave_total[w][k];
        run[k] = before + inc[k] - own[k];
    }
    uint4* dst = P + (size_t)row * px_pitch;
    for (int c = c0; c < c1; c++) {
        uint64_t rec[2];
        tp_px_pack(run, rec);
        const uint4 v = make_uint4((uint32_t)rec[0], (uint32_t)(rec[0] >> 32), (uint32_t)rec[1], (uint32_t)(rec[1] >> 32));
        dst[c] = v;
        if (Pt) Pt[(tp_px_tiled_row_part((uint32_t)row, (uint32_t)px_pitch) + tp_px_tiled_col_part((uint32_t)c)) >> 4] = v;   // (the tiled copy: tp_raster.h)
        if (c < W) px_moments5(src[c], run);
    }
}
void tp_launch_px_table(const uint8_t* img, int pitch, int W, int H, int px_pitch, uint4* P, uint4* P_tiled, hipStream_t s) {
    hipLaunchKernelGGL(k_prefix_px, dim3((unsigned)H), dim3(256), 0, s, img, pitch, W, px_pitch, P, P_tiled);
}

// ------------------------------------------------------------------------------------------------
// Line sums.  W(line) = sum over the line's rows of the row prefix at the crossing column: six values
// {sum x, n_odd, sum r, sum g, sum b, q}.
// ------------------------------------------------------------------------------------------------
struct line_acc {
    uint32_t xs, nodd;     // <= rows * W < 2^28
    uint64_t r, g, b, q;
};
// LINE_BATCH records at the crossing columns c[]: every record is evaluated on its own (no chain of dependent
// instructions from record to record: a chained version with a third fewer instructions ran 12 % slower), the channel
// sums of a few records fit 32 bits (8 x 2^22) and are widened once per trip
template <int N>
__device__ __forceinline__ void acc_records(line_acc& a, const int (&c)[N], const uint4 (&d0)[N], const uint4 (&d1)[N]) {
    static_assert(N <= 8, "32-bit partial sums");
    uint32_t xs = 0, nodd = 0, r = 0, g = 0, b = 0;
#pragma unroll
    for (int u = 0; u < N; u++) {
        const uint32_t rec[TP_PFX_WORDS] = {d0[u].x, d0[u].y, d0[u].z, d0[u].w, d1[u].x, d1[u].y, d1[u].z, d1[u].w};
        uint32_t no, ru, gu, bu, qu;
        tp_prefix_eval(rec, c[u], no, ru, gu, bu, qu);
        xs += (uint32_t)c[u]; nodd += no; r += ru; g += gu; b += bu;
        a.q += qu;
    }
    a.xs += xs; a.nodd += nodd; a.r += r; a.g += g; a.b += b;
}
#ifndef LINE_BATCH
#define LINE_BATCH 4  // table records requested together by one lane (65 VGPRs: every workgroup of a launch is resident at once)
#endif

// k_lines.  A workgroup takes `eb` edges (three: 27 lines; or, for coarse meshes whose lines have hundreds of rows,
// eb == 0: ONE line).  Wave 0 sets every line up ONCE, lane per line, and parks the walkers in LDS.  Then
// thread (l, c) -- line l = tid mod LP, chunk c = tid / LP of TL -- takes the rows rmin + c, rmin + c + TL, ... of the
// band the nine lines of its edge cover: the nine lines of an edge sit in adjacent lanes ON THE SAME ROW, and their
// crossing columns lie within a few pixels of each other (the moves are small), so their table entries share a
// cache line or two.  Chunks meet in LDS.
struct lds_line { int64_t x, s; int32_t ra, rb; };
#ifndef TP_LINES_LP_SHIFT
#define TP_LINES_LP_SHIFT 5  // fine meshes: 32 lanes per chunk = the 27 lines of three edges (6: 64 lanes, seven edges -- fewer, larger workgroups: slower)
#endif
#define LINES_EB (((1 << TP_LINES_LP_SHIFT) / TP_NLINES))
#ifndef TP_LINES_FINE_MAX
#define TP_LINES_FINE_MAX 16  // chunks per line up to which three edges share a workgroup
#endif
#ifndef TP_LINES_WAVES_PER_EU
#define TP_LINES_WAVES_PER_EU 4
#endif

__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(TP_LINES_WAVES_PER_EU))) void k_lines(tp_launch L, int eb, int lp_shift) {
    __shared__ lds_line s_ln[64];
    __shared__ int s_rmin[LINES_EB];
    __shared__ unsigned long long S[64][TP_W_WORDS];
    const int tid = threadIdx.x;
    const int TL = (int)blockDim.x >> lp_shift;
    const int tl_log = 31 - __clz(TL);
#ifndef TP_NO_XCD_MAP
    // workgroup b runs on XCD b mod 8 (each with its own L2): give every XCD one contiguous run of edges -- neighbouring
    // edges read neighbouring table lines
    const int per = (int)gridDim.x >> 3;  // (the grid is padded to a multiple of 8)
    const int blk = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
    if ((eb ? blk * eb * TP_NLINES : blk) >= L.NE * TP_NLINES) return;  // (uniform per workgroup)
#else
    const int blk = blockIdx.x;
#endif
    const int nl = eb ? eb * TP_NLINES : 1;           // lines of this workgroup (eb == 0: ONE line, coarse meshes)
    const int line0 = eb ? blk * eb * TP_NLINES : blk;  // ... the first of them
    TP_STAMP(0, 0);
    if (tid < 64) {  // (one wave: LDS keeps its program order)
        const int j = tid / TP_NLINES;  // edge of the workgroup
        const int e = (line0 + tid) / TP_NLINES, q = line0 + tid - e * TP_NLINES;
        const bool on = tid < nl && e < L.NE;
        if (tid < LINES_EB) s_rmin[tid] = 0x3fffffff;
        tp_line ln; ln.x = 0; ln.s = 0; ln.ra = 1; ln.rb = 0;
        if (on) {
            const int2 uv = L.edge_uv[e];
            const float4 ep = reinterpret_cast<const float4*>(L.epos)[e];
            // line q: endpoint u displaced by move mu, endpoint v by move mv
            const int mu = (q >= 1 && q <= 4) ? q : 0, mv = q >= 5 ? q - 4 : 0;
            int32_t Xa, Ya, Xb, Yb;
            tp_vertex_stage(ep.x, ep.y, mu, 0, L.vw, Xa, Ya);
            tp_vertex_stage(ep.z, ep.w, mv, 0, L.vw, Xb, Yb);
            tp_setup_line(Xa, Ya, Xb, Yb, L.vw.H, ln);
            // one edge per vertex publishes its snapped positions (k_finalize reads them)
            if (mv == 0 && ((uv.x >> 30) & 1)) L.vpos[(size_t)(uv.x & 0x3fffffff) * 5 + mu] = make_int2(Xa, Ya);
            if (mu == 0 && ((uv.y >> 30) & 1)) L.vpos[(size_t)(uv.y & 0x3fffffff) * 5 + mv] = make_int2(Xb, Yb);
            if (ln.ra <= ln.rb) atomicMin(&s_rmin[j], ln.ra);  // (after the initialisation: same wave, LDS keeps program order)
        }
        s_ln[tid].x = ln.x; s_ln[tid].s = ln.s; s_ln[tid].ra = ln.ra; s_ln[tid].rb = ln.rb;
#pragma unroll
        for (int k = 0; k < TP_W_WORDS; k++) S[tid][k] = 0ull;
    }
    __syncthreads();
    TP_STAMP(0, 1);
    const int l = tid & ((1 << lp_shift) - 1), c = tid >> lp_shift;
    const int j = l / TP_NLINES;
    const bool on = l < nl && line0 + l < L.NE * TP_NLINES;
    line_acc a = {0, 0, 0, 0, 0, 0};
    if (on) {
        const lds_line ln = s_ln[l];
        // this thread's rows of the line: first, first + TL, ... <= rb, where first is the first row >= ra on the
        // thread's residue (rmin + c) mod TL of the band
        const int base = s_rmin[j] + c;
        const int first = base + ((ln.ra - base + TL - 1) & -TL);  // (ra >= rmin: the numerator is > -TL; TL is a power of two)
        int n = ln.rb >= first ? ((ln.rb - first) >> tl_log) + 1 : 0;
        // the walker of the line stepped by TL rows, and the table row pointer likewise
        int64_t x = ln.x + (int64_t)(first - ln.ra) * ln.s;
        const int64_t xs = (int64_t)((uint64_t)ln.s * (uint64_t)TL);   // (unsigned: a steep two-row line may wrap; its step is never used)
        // (byte offsets into the table: uint32 arithmetic, 16384 rows x 4100 records x 32 bytes < 2^32)
        const char* table = reinterpret_cast<const char*>(L.prefix);
        uint32_t row = (uint32_t)(n > 0 ? first : 0) * (uint32_t)L.prefix_pitch * 32u;
        const uint32_t rs = (uint32_t)TL * (uint32_t)L.prefix_pitch * 32u;
        const int W = L.vw.W;
        for (; n > 0; n -= LINE_BATCH) {
            uint4 d0[LINE_BATCH], d1[LINE_BATCH];
            int col[LINE_BATCH];
#pragma unroll
            for (int u = 0; u < LINE_BATCH; u++) {
                d0[u] = make_uint4(0, 0, 0, 0); d1[u] = d0[u]; col[u] = 0;
                if (u < n) {
                    const int xc = (int)(x >> TP_LINE_FRAC);
                    col[u] = xc < 0 ? 0 : (xc > W ? W : xc);
                    const uint4* rec = reinterpret_cast<const uint4*>(table + (row + (((uint32_t)col[u] & ~3u) << 3)));
                    d0[u] = rec[0]; d1[u] = rec[1];
                    x = (int64_t)((uint64_t)x + (uint64_t)xs); row += rs;
                }
            }
            acc_records(a, col, d0, d1);
        }
    }
    TP_STAMP(0, 2);
    int64_t* w = L.wline + ((size_t)line0 + l) * TP_W_WORDS;
    if (TL == 1) {
        if (on) { w[0] = a.xs; w[1] = a.nodd; w[2] = (int64_t)a.r; w[3] = (int64_t)a.g; w[4] = (int64_t)a.b; w[5] = (int64_t)a.q; }
    } else {
        if (on && (a.xs | a.nodd | a.r | a.q)) {
            unsigned long long* sq = S[l];
            atomicAdd(&sq[0], (unsigned long long)a.xs); atomicAdd(&sq[1], (unsigned long long)a.nodd);
            atomicAdd(&sq[2], (unsigned long long)a.r); atomicAdd(&sq[3], (unsigned long long)a.g);
            atomicAdd(&sq[4], (unsigned long long)a.b); atomicAdd(&sq[5], (unsigned long long)a.q);
        }
        __syncthreads();
        if (on && c == 0) {
#pragma unroll
            for (int k = 0; k < TP_W_WORDS; k++) w[k] = (int64_t)S[l][k];
        }
    }
    TP_STAMP(0, 3);
}
void tp_launch_lines(const tp_launch& L, hipStream_t s, hipEvent_t start, hipEvent_t stop) {
    // lanes_per_line: chunks per line (1 .. 1024).  Up to 16 chunks: three edges per workgroup, 32 lanes per chunk;
    // beyond (coarse meshes, lines of hundreds of rows): one LINE per workgroup, a lane per chunk
    const int tl = L.lanes_per_line;
    const int eb = tl <= TP_LINES_FINE_MAX ? LINES_EB : 0, lp_shift = tl <= TP_LINES_FINE_MAX ? TP_LINES_LP_SHIFT : 0;
    const unsigned blocks = eb ? (unsigned)((L.NE + eb - 1) / eb) : (unsigned)(L.NE * TP_NLINES);
    hipExtLaunchKernelGGL(k_lines, dim3((blocks + 7u) & ~7u), dim3((unsigned)(tl << lp_shift)), 0, s, start, stop, 0, L, eb, lp_shift);
}

__device__ __forceinline__ void line_sum(const tp_launch& L, int e, int ver, int64_t w[TP_W_WORDS]) {
    const int64_t* src = L.wline + ((size_t)e * TP_NLINES + ver) * TP_W_WORDS;
#pragma unroll
    for (int q = 0; q < TP_W_WORDS; q++) w[q] = src[q];
}

// per-variant moments = signed sum of three line sums
__device__ __forceinline__ tp_moments variant_moments(const tp_launch& L, int t, int i) {
    const int4 tri = L.tris[t];
    const int vid[3] = {tri.x, tri.y, tri.z};
    const int ms = i > 0 ? (i - 1) >> 2 : 3, mm = i > 0 ? ((i - 1) & 3) + 1 : 0;
    int32_t X[3], Y[3], c[3];
    int64_t w[3][TP_W_WORDS];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int he = L.he_edge[3 * t + k];
        line_sum(L, he >> 1, tp_edge_version(i, k, he & 1), w[k]);
    }
#pragma unroll
    for (int s = 0; s < 3; s++) {
        const int2 q = L.vpos[(size_t)vid[s] * 5 + (s == ms ? mm : 0)];
        X[s] = q.x; Y[s] = q.y;
    }
    tp_variant_coeffs(X, Y, c);
    int64_t m[TP_W_WORDS];
#pragma unroll
    for (int q = 0; q < TP_W_WORDS; q++) m[q] = (int64_t)c[0] * w[0][q] + (int64_t)c[1] * w[1][q] + (int64_t)c[2] * w[2][q];
    tp_moments res = {m[0], m[1], m[2], m[3], m[4], m[5]};
    return res;
}

__device__ __forceinline__ int32_t emit_variant(const tp_launch& L, int flavour, int t, int i, const tp_moments& m,
                                                bool write_moments, bool store = true) {
    const int id = i * L.NT + t;
    int64_t E;
    if (flavour == 0) {
        E = tp_energy_triangulate(m);
        if (store) L.ca[id] = make_int4(tp_wrap32(m.sr), tp_wrap32(m.sg), tp_wrap32(m.sb), 0);
    } else {
        const int4 col = L.ca[id];  // stored colour, replicated x13 by upload
        E = tp_energy64(m, col.x, col.y, col.z);
    }
    const int32_t e32 = tp_wrap32(E);
    if (store) {
        L.ten[id] = e32;
        L.cn[id] = tp_wrap32(m.n);
        if (L.mirror_ten && id < L.mirror_n) { L.mirror_ten[id] = e32; L.mirror_cn[id] = tp_wrap32(m.n); }
    }
    if (write_moments) {
        int64_t* o = L.moments + (size_t)id * 6;
        o[0] = m.n; o[1] = m.nodd; o[2] = m.sr; o[3] = m.sg; o[4] = m.sb; o[5] = m.q;
    }
    return e32;
}

// k_finalize (tp_energy): thread per (triangle, variant); id = i*NT + t in the outputs
__global__ __launch_bounds__(256) void k_finalize(tp_launch L, int flavour, int write_moments) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= L.NT * TP_NVARIANTS) return;
    const int t = gid / TP_NVARIANTS, i = gid - t * TP_NVARIANTS;
    emit_variant(L, flavour, t, i, variant_moments(L, t, i), write_moments != 0);
}
void tp_launch_finalize(const tp_launch& L, int flavour, bool write_moments, hipStream_t s) {
    const int n = L.NT * TP_NVARIANTS;
    hipLaunchKernelGGL(k_finalize, dim3((n + 255) / 256), dim3(256), 0, s, L, flavour, write_moments ? 1 : 0);
}

#define UPD_THREADS 64
#define UPD_CHUNK 7  // generic path: incident triangles per pass, 7 x 9 = 63 lanes
#define UPD_FAN 8    // fast path: up to eight incident triangles and eight incident edges

// file vertex v's position p with every edge it ends (epos[edge][side]); `lane`, `nlanes`: the lanes sharing the work.
// Which side is decided by the edge's own endpoint list (an edge of a triangle soup can name the same vertex twice).
__device__ __forceinline__ void publish_to_edge(const tp_launch& L, int e, int v, float2 p) {
    const int2 uv = L.edge_uv[e];
    if ((uv.x & 0x3fffffff) == v) L.epos[(size_t)e * 2] = p;
    if ((uv.y & 0x3fffffff) == v) L.epos[(size_t)e * 2 + 1] = p;
}
__device__ __forceinline__ void publish_position(const tp_launch& L, int v, float2 p, int lane, int nlanes) {
    const int r0 = L.vref[(size_t)v * 64];
    if (r0 != -2) {  // fast layout: slot 4 b names incident edge b
        for (int b = lane; b < UPD_FAN; b += nlanes) {
            const int r = L.vref[(size_t)v * 64 + 4 * b];
            if (r >= 0) publish_to_edge(L, r >> 4, v, p);
        }
    } else {         // more than eight incident triangles: through the adjacency
        const int k0 = L.vtx_off[v], deg = L.vtx_off[v + 1] - k0;
        for (int a = lane; a < deg; a += nlanes) {
            const int h = L.vtx_adj[k0 + a], t = h / 3, sl = h - 3 * t;
            publish_to_edge(L, L.he_edge[3 * t + sl] >> 1, v, p);                      // the edge leaving the vertex
            publish_to_edge(L, L.he_edge[3 * t + (sl == 0 ? 2 : sl - 1)] >> 1, v, p);  // the edge arriving at it
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_shift (tp_shift): gradient.cs gathered per vertex (no atomics) + shift.cs
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_shift(tp_launch L, float rate) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= L.NP) return;
    uint32_t gx = 0, gy = 0;  // int32 wrapping sums, like the reference's int atomics
    const int NT = L.NT;
    for (int k = L.vtx_off[gid]; k < L.vtx_off[gid + 1]; k++) {
        const int h = L.vtx_adj[k], t = h / 3, s = h - 3 * t;
        const int32_t* e = L.ten + t;
        gx += (uint32_t)e[(4 * s + 1) * NT] - (uint32_t)e[(4 * s + 2) * NT];
        gy += (uint32_t)e[(4 * s + 3) * NT] - (uint32_t)e[(4 * s + 4) * NT];
    }
    L.gr[gid] = make_int2((int)gx, (int)gy);
    if (gid < 4) {  // shift.cs:20 -- the four corners never move
        if (L.mirror_pts) L.mirror_pts[gid] = L.points[gid];
        return;
    }

    float tgx = (float)(int)gx, tgy = (float)(int)gy;
    float2 p = L.points[gid];
    const float R = L.vw.ratio;
    if (p.x <= -R) { p.x = -R; tgx = 0.0f; } else if (p.x >= R) { p.x = R; tgx = 0.0f; }
    if (p.y <= -1.0f) { p.y = -1.0f; tgy = 0.0f; } else if (p.y >= 1.0f) { p.y = 1.0f; tgy = 0.0f; }
    // p -= rate * tgr / 256 / 256  (shift.cs:45), one rounding per operation
    p.x = tp_fsub(p.x, tp_shift_scale(tp_fmul(rate, tgx)));
    p.y = tp_fsub(p.y, tp_shift_scale(tp_fmul(rate, tgy)));
    L.points[gid] = p;
    if (L.mirror_pts) L.mirror_pts[gid] = p;
    publish_position(L, gid, p, 0, 1);
}
void tp_launch_shift(const tp_launch& L, float rate, hipStream_t s) {
    hipLaunchKernelGGL(k_shift, dim3((L.NP + 255) / 256), dim3(256), 0, s, L, rate);
}

// ------------------------------------------------------------------------------------------------
// k_update: k_finalize + k_shift in ONE launch (used by tp_iterate), organised by VERTEX: one wave per vertex.
// A vertex's gradient needs the 4 displaced variants of each incident (triangle, slot); those use, per incident
// triangle, nine line sums: the opposite edge's base line and the four displaced versions of either edge at the
// vertex.  Lane (a, l) = (incident triangle a of a chunk of seven, line l of nine) forms one line sum and parks it
// in LDS; lane (a, m) then combines three of them into the moments of variant (t, 4s + m), writes `colnum`,
// `colacc`, `tenergy` (reference layout) and keeps the energy; central differences are a lane-pair subtraction and
// the vertex's gradient (wrapping int32, like the reference's atomics -- integer sums commute) a wave reduction.
// Lane 0 takes the shift.cs step.  No atomics, no arrival counters.  The base variants (i = 0) do not enter any
// gradient: extra workgroups behind the vertices write their outputs, one thread per triangle.
// ------------------------------------------------------------------------------------------------

// line l of nine for the incident (triangle, slot) h = 3t + s of a vertex, packed edge << 4 | version:
// l = 0: the opposite edge's base line; 1..4: the edge leaving the vertex, vertex displaced by move l;
// 5..8: the edge arriving at the vertex, vertex displaced by move l - 4
__device__ __forceinline__ int vertex_line_ref(const tp_launch& L, int h, int l) {
    const int t = h / 3, s = h - 3 * t;
    const int m = l == 0 ? 0 : ((l - 1) & 3) + 1;
    const int k = l == 0 ? (s == 2 ? 0 : s + 1) : l <= 4 ? s : (s == 0 ? 2 : s - 1);
    const int he = L.he_edge[3 * t + k];
    return ((he >> 1) << 4) | tp_edge_version(l == 0 ? 0 : 4 * s + m, k, he & 1);
}

// Per upload: what the lanes of k_update's wave for vertex v work on (instead of three dependent loads per iteration).
// Fast layout, for a vertex with at most UPD_FAN incident triangles and incident edges:
//   vref[v][4 b + m - 1], b < 8, m = 1..4   incident edge b with the vertex displaced by move m   (edge << 4 | version)
//   vref[v][32 + a], a < 8                    the edge opposite the vertex in incident triangle a, base version
//   vvar[v][a]                                3t + s | slot of the edge leaving the vertex << 20 | slot of the edge arriving << 24
// Otherwise vref[v][0] = -2 and k_update derives everything itself, seven triangles at a time.
__global__ void k_vertex_refs(tp_launch L, int* vref, int* vvar) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= L.NP) return;
    const int k0 = L.vtx_off[v], deg = L.vtx_off[v + 1] - k0;
    // gather everything first; the rows are written once, at the end (no exits from inside the loops)
    int edges[UPD_FAN], flips[UPD_FAN], ne = 0;  // incident edges and whether the vertex is their second endpoint
    int comb[UPD_FAN], opp[UPD_FAN];
    bool generic = deg > UPD_FAN;
    for (int a = 0; a < UPD_FAN; a++) { comb[a] = -1; opp[a] = -1; }
    for (int a = 0; a < deg && !generic; a++) {
        const int h = L.vtx_adj[k0 + a], t = h / 3, s = h - 3 * t;
        const int he_out = L.he_edge[3 * t + s], he_in = L.he_edge[3 * t + (s == 0 ? 2 : s - 1)];
        int slot[2] = {0, 0};
        for (int w = 0; w < 2; w++) {
            const int he = w == 0 ? he_out : he_in;
            // leaving edge: the vertex is its origin -- second endpoint when the half-edge is flipped; arriving: the other way
            const int second = w == 0 ? (he & 1) : !(he & 1);
            int b = 0;
            while (b < ne && edges[b] != (he >> 1)) b++;
            if (b == ne) {
                const int2 uv = L.edge_uv[he >> 1];
                // a ninth incident edge, or an edge of a triangle soup that names this vertex twice (the fast layout files a
                // position with ONE end of an edge)
                if (ne == UPD_FAN || ((uv.x ^ uv.y) & 0x3fffffff) == 0) generic = true;
                else { edges[ne] = he >> 1; flips[ne] = second; ne++; }
            }
            slot[w] = b;
        }
        comb[a] = h | (slot[0] << 20) | (slot[1] << 24);
        opp[a] = ((L.he_edge[3 * t + (s == 2 ? 0 : s + 1)] >> 1) << 4) | 0;
    }
    int* r = vref + (size_t)v * 64;
    int* c = vvar + (size_t)v * 8;
    for (int k = 0; k < 64; k++) r[k] = -1;
    for (int k = 0; k < 8; k++) c[k] = -1;
    if (generic) { r[0] = -2; return; }
    for (int a = 0; a < UPD_FAN; a++) { c[a] = comb[a]; r[32 + a] = opp[a]; }
    for (int b = 0; b < ne; b++)
        for (int m = 1; m <= 4; m++) r[4 * b + m - 1] = (edges[b] << 4) | (flips[b] ? 4 + m : m);
}
// per upload, after k_vertex_refs: every vertex files its position with its edges
__global__ void k_publish_positions(tp_launch L) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < L.NP) publish_position(L, v, L.points[v], 0, 1);
}
__global__ __launch_bounds__(256) void k_copy_list(tp_copy_list G) {
    for (int k = 0; k < G.n; k++) {
        const uint32_t* src = G.src[k];
        uint32_t* dst = G.dst[k];
        for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < G.words[k]; i += gridDim.x * 256u) dst[i] = src[i];
    }
}
// tp_iterate_until: geterr's float32 sum of a frame's base energies, ascending t (source/triangulation.hpp:653-674), for the C frames of a
// chunk.  A frame's sum is ONE chain of NT dependent additions in the reference's order (float addition does not reassociate), but the
// frames do not depend on each other: one wave per frame here instead of 3 MB across the link and eight chains at a time on the host.
// The wave fetches 512 energies at a time, coalesced (the next 512 are on their way while these are added), and every lane runs the same
// chain over them, element after element out of the lanes' registers.  Written straight into the pinned buffer the host tests from.
__global__ __launch_bounds__(64) void k_frame_sums(const int32_t* ering, int C, int NT, float* out) {
    const int j = blockIdx.x, lane = threadIdx.x;
    if (j >= C) return;
    const int32_t* e = ering + (size_t)j * NT;
    auto fetch = [&](int base, float v[8]) {
#pragma unroll
        for (int k = 0; k < 8; k++) { const int i = base + 64 * k + lane; v[k] = i < NT ? (float)e[i] : 0.0f; }
    };
    float newerr = 0.0f, cur[8], nxt[8];
    fetch(0, cur);
    for (int base = 0; base < NT; base += 512) {
        if (base + 512 < NT) fetch(base + 512, nxt);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int left = NT - (base + 64 * k);   // (the same for every lane)
            if (left >= 64) {
#pragma unroll
                for (int l = 0; l < 64; l++) { float err = 0.0f; err += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cur[k]), l)); newerr += err; }
            } else {
                for (int l = 0; l < left; l++) { float err = 0.0f; err += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cur[k]), l)); newerr += err; }
            }
        }
#pragma unroll
        for (int k = 0; k < 8; k++) cur[k] = nxt[k];
    }
    if (lane == 0) out[j] = newerr;
}
void tp_launch_frame_sums(const int32_t* ering, int C, int NT, float* out, hipStream_t s) {
    hipLaunchKernelGGL(k_frame_sums, dim3((unsigned)C), dim3(64), 0, s, ering, C, NT, out);
}
void tp_launch_copy_list(const tp_copy_list& G, hipStream_t s) {
    uint32_t most = 0;
    for (int k = 0; k < G.n; k++) most = G.words[k] > most ? G.words[k] : most;
    if (!most) return;
    const unsigned blocks = (most + 1023u) / 1024u;   // four words per thread of the longest array
    hipLaunchKernelGGL(k_copy_list, dim3(blocks > 1024u ? 1024u : blocks), dim3(256), 0, s, G);
}
void tp_launch_vertex_refs(const tp_launch& L, int* vref, int* vvar, hipStream_t s) {
    hipLaunchKernelGGL(k_vertex_refs, dim3((unsigned)((L.NP + 63) / 64)), dim3(64), 0, s, L, vref, vvar);
    hipLaunchKernelGGL(k_publish_positions, dim3((unsigned)((L.NP + 63) / 64)), dim3(64), 0, s, L);
}

__global__ __launch_bounds__(UPD_THREADS) void k_update(tp_launch L, int flavour, float rate) {
    __shared__ int64_t S[64][TP_W_WORDS];  // line sums of the wave: [lane] (fast path), [a][l] (generic), [a][k] (base variants)
    const int lane = threadIdx.x;
    TP_STAMP(2, 0);
    // one variant: signed sum of three parked line sums -> outputs; returns the energy
    auto variant_at = [&](int h, int m, const int32_t X[3], const int32_t Y[3], const int64_t* Sout, const int64_t* Sin,
                          const int64_t* Sopp) -> int32_t {
        const int t = h / 3, s = h - 3 * t;
        int32_t c[3];
        tp_variant_coeffs(X, Y, c);
        const int kn = s == 2 ? 0 : s + 1, kp = s == 0 ? 2 : s - 1;
        const int cs = s == 0 ? c[0] : s == 1 ? c[1] : c[2];
        const int cn = kn == 0 ? c[0] : kn == 1 ? c[1] : c[2];
        const int cp = kp == 0 ? c[0] : kp == 1 ? c[1] : c[2];
        int64_t mo[TP_W_WORDS];
#pragma unroll
        for (int q = 0; q < TP_W_WORDS; q++) mo[q] = (int64_t)cs * Sout[q] + (int64_t)cp * Sin[q] + (int64_t)cn * Sopp[q];
        const tp_moments mm = {mo[0], mo[1], mo[2], mo[3], mo[4], mo[5]};
        return emit_variant(L, flavour, t, 4 * s + m, mm, false);
    };
    auto variant = [&](int h, int m, const int64_t* Sout, const int64_t* Sin, const int64_t* Sopp) -> int32_t {
        const int t = h / 3, s = h - 3 * t;
        const int4 tri = L.tris[t];
        const int vid[3] = {tri.x, tri.y, tri.z};
        int32_t X[3], Y[3];
#pragma unroll
        for (int ss = 0; ss < 3; ss++) {
            const int2 q = L.vpos[(size_t)vid[ss] * 5 + (ss == s ? m : 0)];
            X[ss] = q.x; Y[ss] = q.y;
        }
        return variant_at(h, m, X, Y, Sout, Sin, Sopp);
    };
    if ((int)blockIdx.x >= L.NP) {
        // base variants: 21 triangles per workgroup, lane (a, k) sums the base line of edge k, lane a combines
        const int tb = ((int)blockIdx.x - L.NP) * 21;
        {
            const int a = lane / 3, k = lane - 3 * a, t = tb + a;
            if (a < 21 && t < L.NT) {
                const int he = L.he_edge[3 * t + k];
                int64_t w[TP_W_WORDS];
                line_sum(L, he >> 1, 0, w);
#pragma unroll
                for (int q = 0; q < TP_W_WORDS; q++) S[lane][q] = w[q];
            }
        }
        __syncthreads();
        const int t = tb + lane;
        if (lane < 21 && t < L.NT) {
            const int4 tri = L.tris[t];
            const int vid[3] = {tri.x, tri.y, tri.z};
            int32_t X[3], Y[3], c[3];
#pragma unroll
            for (int ss = 0; ss < 3; ss++) {
                const int2 q = L.vpos[(size_t)vid[ss] * 5];
                X[ss] = q.x; Y[ss] = q.y;
            }
            tp_variant_coeffs(X, Y, c);
            int64_t mo[TP_W_WORDS];
#pragma unroll
            for (int q = 0; q < TP_W_WORDS; q++)
                mo[q] = (int64_t)c[0] * S[3 * lane][q] + (int64_t)c[1] * S[3 * lane + 1][q] + (int64_t)c[2] * S[3 * lane + 2][q];
            const tp_moments mm = {mo[0], mo[1], mo[2], mo[3], mo[4], mo[5]};
            emit_variant(L, flavour, t, 0, mm, false);
        }
    } else {
        const int v = blockIdx.x;
        const int ref = L.vref[(size_t)v * 64 + lane];
        const int comb = lane < 4 * UPD_FAN ? L.vvar[(size_t)v * 8 + (lane >> 2)] : -1;
        float2 p = make_float2(0.0f, 0.0f);
        if (lane == 0) p = L.points[v];
        uint32_t gx = 0, gy = 0;
        const int generic = __shfl(ref, 0) == -2;
        int deg = 1;
        if (!generic) {
            // fast path: lane 4 b + m - 1 sums the line of incident edge b displaced by move m, lane 32 + a the base line
            // opposite the vertex in incident triangle a; then lane 4 a + m - 1 forms variant (t_a, 4 s_a + m).
            // The variant's three snapped vertices are requested first: they fly beside the records.
            int32_t VX[3] = {0, 0, 0}, VY[3] = {0, 0, 0};
            if (comb >= 0) {
                const int h = comb & 0xfffff, t = h / 3, s = h - 3 * t, m = (lane & 3) + 1;
                const int4 tri = L.tris[t];
                const int vid[3] = {tri.x, tri.y, tri.z};
#pragma unroll
                for (int ss = 0; ss < 3; ss++) {
                    const int2 q = L.vpos[(size_t)vid[ss] * 5 + (ss == s ? m : 0)];
                    VX[ss] = q.x; VY[ss] = q.y;
                }
            }
            if (ref >= 0) {
                int64_t w[TP_W_WORDS];
                line_sum(L, ref >> 4, ref & 15, w);
#pragma unroll
                for (int q = 0; q < TP_W_WORDS; q++) S[lane][q] = w[q];
            }
            __syncthreads();
            TP_STAMP(2, 1);
            int32_t e = 0;
            if (comb >= 0) {
                const int m = (lane & 3) + 1, a = lane >> 2;
                e = variant_at(comb & 0xfffff, m, VX, VY, S[4 * ((comb >> 20) & 15) + m - 1], S[4 * ((comb >> 24) & 15) + m - 1], S[32 + a]);
            }
            // central differences: lanes 4a+0/1 hold E(+dx)/E(-dx), 4a+2/3 E(+dy)/E(-dy)
            const uint32_t d = (uint32_t)e - (uint32_t)__shfl_xor(e, 1);
            if ((lane & 3) == 0) gx += d;
            if ((lane & 3) == 2) gy += d;
            deg = __any(comb >= 0) ? 1 : 0;
            TP_STAMP(2, 2);
        } else {
            // generic path (a vertex of more than eight triangles): seven incident triangles per pass, lane (a, l) sums
            // line l of nine of triangle a, lane (a, m) forms its variant
            const int k0 = L.vtx_off[v];
            deg = L.vtx_off[v + 1] - k0;
            for (int base = 0; base < deg; base += UPD_CHUNK) {
                {
                    const int a = lane / TP_NLINES, l = lane - a * TP_NLINES;
                    if (a < UPD_CHUNK && base + a < deg) {
                        const int r = vertex_line_ref(L, L.vtx_adj[k0 + base + a], l);
                        int64_t w[TP_W_WORDS];
                        line_sum(L, r >> 4, r & 15, w);
#pragma unroll
                        for (int q = 0; q < TP_W_WORDS; q++) S[lane][q] = w[q];
                    }
                }
                __syncthreads();
                int32_t e = 0;
                {
                    const int a = lane >> 2, m = (lane & 3) + 1;
                    if (a < UPD_CHUNK && base + a < deg)
                        e = variant(L.vtx_adj[k0 + base + a], m, S[9 * a + m], S[9 * a + 4 + m], S[9 * a]);
                }
                const uint32_t d = (uint32_t)e - (uint32_t)__shfl_xor(e, 1);
                if ((lane & 3) == 0) gx += d;
                if ((lane & 3) == 2) gy += d;
                __syncthreads();  // S is rewritten by the next pass
            }
        }
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { gx += (uint32_t)__shfl_xor((int)gx, o); gy += (uint32_t)__shfl_xor((int)gy, o); }
        float2 newp = make_float2(0.0f, 0.0f);
        int moved = 0;
        if (lane == 0) {
            if (deg > 0) L.gr[v] = make_int2((int)gx, (int)gy);  // vertices no triangle uses: the gradient is never touched
            if (v >= 4) {  // shift.cs:20 -- the four corners never move
                const float R = L.vw.ratio;
                float tgx = (float)(int)gx, tgy = (float)(int)gy;
                float x = p.x, y = p.y;
                if (x <= -R) { x = -R; tgx = 0.0f; } else if (x >= R) { x = R; tgx = 0.0f; }
                if (y <= -1.0f) { y = -1.0f; tgy = 0.0f; } else if (y >= 1.0f) { y = 1.0f; tgy = 0.0f; }
                if (deg > 0) {  // (unused vertices are only clamped: shift.cs:25-43 runs for every i in [4, NPoints))
                    x = tp_fsub(x, tp_shift_scale(tp_fmul(rate, tgx)));
                    y = tp_fsub(y, tp_shift_scale(tp_fmul(rate, tgy)));
                }
                L.points[v] = make_float2(x, y);
                newp = make_float2(x, y); moved = 1;
            }
            if (L.mirror_pts) L.mirror_pts[v] = moved ? newp : p;
        }
        // the new position goes to every edge the vertex ends (k_lines reads endpoints by edge)
        if (__shfl(moved, 0)) {
            const float2 q = make_float2(__shfl(newp.x, 0), __shfl(newp.y, 0));
            if (!generic) {
                // lane 4 b holds (incident edge b, version 1..4: the vertex is the edge's first endpoint, 5..8: its second):
                // no table to look the side up in (publish_position's dependent load at the very end of the kernel)
                if ((lane & 3) == 0 && lane < 4 * UPD_FAN && ref >= 0) L.epos[(size_t)(ref >> 4) * 2 + ((ref & 15) > 4 ? 1 : 0)] = q;
            } else
                publish_position(L, v, q, lane, UPD_THREADS);
        }
    }
    TP_STAMP(2, 3);
}
void tp_launch_update(const tp_launch& L, int flavour, float rate, hipStream_t s) {
    const int nblocks = L.NP + (L.NT + 20) / 21;  // a wave per vertex, then the base variants (21 triangles per wave)
    hipLaunchKernelGGL(k_update, dim3((unsigned)nblocks), dim3(UPD_THREADS), 0, s, L, flavour, rate);
}

// after a persistent launch (tp_persist.hip): the positions it left in `points_out` become `points` (vertices no triangle
// uses are not owned by any patch and keep theirs), and every vertex files its position with its edges (k_lines reads
// endpoints by edge)
// A launch that gave up (status[0] raised: its workgroups were not all resident, tp_context.hip) leaves everything as it was.
// file_edges: every vertex also files its position with its edges (`epos`, what k_lines reads) -- only when the two-kernel path runs
// next; otherwise that is left to k_publish_positions, launched when it does (the dependent table reads cost 4 us of every call)
// status[3]: a ticket counter -- the LAST block to finish counts the launch as completed and mirrors the words into pinned memory,
// so that a host spinning on the mirror sees it only when every block's stores are out
__global__ void k_persist_finish(tp_launch L, const float2* points_out, unsigned* status, unsigned* host_status, int file_edges) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int per = file_edges ? 8 : 1;   // lanes per vertex: with edges one incident edge each (the table reads are dependent ones)
    const int v = gid / per, lane = gid - v * per;
    const unsigned gave_up = status ? status[0] : 0u;
    if (gave_up == 0u && v < L.NP) {
        float2 p = L.points[v];
        if (L.vtx_off[v + 1] > L.vtx_off[v]) p = points_out[v];
        else if (v >= 4) {   // shift.cs:25-43 runs for every i in [4, NPoints): a vertex no triangle uses is still clamped
            const float R = L.vw.ratio;
            p.x = p.x <= -R ? -R : (p.x >= R ? R : p.x);
            p.y = p.y <= -1.0f ? -1.0f : (p.y >= 1.0f ? 1.0f : p.y);
        }
        // (every lane has read the old position before any lane of the vertex writes the new one: the lanes of a vertex sit in one wave)
        if (lane == 0) L.points[v] = p;
        if (file_edges) publish_position(L, v, p, lane, 8);
    }
    if (status) {
        // (no release here: whatever reads the positions is work on this stream, ordered behind the end of the kernel; the pinned word only
        // spares the host a question to the runtime -- and a fence in every lane is a cache write-back per lane)
        __syncthreads();
        if (threadIdx.x == 0 && atomicAdd(&status[3], 1u) == gridDim.x - 1u) {
            status[3] = 0u;
            const unsigned done = status[2] + (gave_up ? 0u : 1u);   // (launches complete one after the other)
            status[2] = done;
            if (host_status) { host_status[0] = gave_up; __threadfence_system(); host_status[2] = done; }   // what the host looks at: no copy
        }
    }
}
void tp_launch_persist_finish(const tp_launch& L, const float2* points_out, unsigned* status, unsigned* host_status, int file_edges, hipStream_t s) {
    const int threads = (file_edges ? 8 : 1) * L.NP;
    hipLaunchKernelGGL(k_persist_finish, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, L, points_out, status, host_status, file_edges);
}
void tp_launch_publish_positions(const tp_launch& L, hipStream_t s) {
    hipLaunchKernelGGL(k_publish_positions, dim3((unsigned)((L.NP + 63) / 64)), dim3(64), 0, s, L);
}

// tpose::upload colour replication (source/triangulation.hpp:633-641): col[i*NT + k] = colors[k]
__global__ void k_replicate_colors(tp_launch L) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= L.NT * TP_NVARIANTS) return;
    L.ca[gid] = L.colors[gid % L.NT];
}
void tp_launch_replicate_colors(const tp_launch& L, hipStream_t s) {
    const int n = L.NT * TP_NVARIANTS;
    hipLaunchKernelGGL(k_replicate_colors, dim3((n + 255) / 256), dim3(256), 0, s, L);
}

// device-side self-test of the edge walker (tp_selftest_walker): 32 row values per (N0, step, d)
__global__ void k_selftest_walker(const int64_t* N0, const int32_t* step, const int32_t* d, int n, int32_t* out) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n) return;
    tp_walker w = tp_make_walker(N0[gid], step[gid], d[gid]);
    for (int r = 0; r < 32; r++) { out[(size_t)gid * 32 + r] = tp_walker_value(w); w.x += w.s; }
}
void tp_launch_selftest_walker(const int64_t* N0, const int32_t* step, const int32_t* d, int n, int32_t* out, hipStream_t s) {
    hipLaunchKernelGGL(k_selftest_walker, dim3((n + 255) / 256), dim3(256), 0, s, N0, step, d, n, out);
}

// ------------------------------------------------------------------------------------------------
// k_render (tp_render): flat-shaded picture, one 64-thread block per triangle, one 32-row window per
// thread; spans from the same exact walkers as the sweep, so pixels are owned exactly once
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_render(tp_launch L, const float2* pts, int source, uchar4* out, int out_pitch_px) {
    const int t = blockIdx.x;
    const int4 tri = L.tris[t];
    const int vid[3] = {tri.x, tri.y, tri.z};
    int32_t X[3], Y[3];
#pragma unroll
    for (int s = 0; s < 3; s++) {
        const float2 p = pts[vid[s]];
        tp_vertex_stage(p.x, p.y, 0, s, L.vw, X[s], Y[s]);
    }
    uchar4 col = make_uchar4(0, 0, 0, 255);
    if (source == 0) {  // triangle.fs:48  vec3(ca.rgb) / cn / 255 -> RGBA8 (round to nearest)
        const int4 a = L.ca[t];
        const int n = L.cn[t];
        if (n == 0) return;
        const float r = tp_fdiv(tp_fdiv((float)a.x, (float)n), 255.0f), g = tp_fdiv(tp_fdiv((float)a.y, (float)n), 255.0f),
                    b = tp_fdiv(tp_fdiv((float)a.z, (float)n), 255.0f);
        col.x = (unsigned char)floorf(fminf(fmaxf(r, 0.0f), 1.0f) * 255.0f + 0.5f);
        col.y = (unsigned char)floorf(fminf(fmaxf(g, 0.0f), 1.0f) * 255.0f + 0.5f);
        col.z = (unsigned char)floorf(fminf(fmaxf(b, 0.0f), 1.0f) * 255.0f + 0.5f);
    } else {
        const int4 a = L.colors[t];
        col.x = (unsigned char)min(max(a.x, 0), 255); col.y = (unsigned char)min(max(a.y, 0), 255);
        col.z = (unsigned char)min(max(a.z, 0), 255);
    }
    const int ymin = min(Y[0], min(Y[1], Y[2])), ymax = max(Y[0], max(Y[1], Y[2]));
    const int rtop = max(tp_first_centre(ymin), 0), rbot = min(tp_last_centre(ymax), L.vw.H - 1);
    for (int w0 = rtop + 32 * (int)threadIdx.x; w0 <= rbot; w0 += 32 * 64) {
        tp_span sp;
        tp_setup_span(X, Y, w0, min(w0 + 31, rbot), sp);
        for (int r = sp.r0; r <= sp.r1; r++) {
            int32_t lo, hi;
            tp_span_row(sp, 0, L.vw.W, lo, hi);
            uchar4* row = out + (size_t)r * out_pitch_px;
            for (int c = lo; c < hi; c++) row[c] = col;
        }
    }
}
void tp_launch_render(const tp_launch& L, const float2* pts, int source, void* out, int out_pitch_px, hipStream_t s) {
    hipLaunchKernelGGL(k_render, dim3(L.NT), dim3(64), 0, s, L, pts, source, (uchar4*)out, out_pitch_px);
}


// device-side self-test of the whole-line walker (tp_selftest_line): for the line (Xa,Ya)-(Xb,Yb) on a raster of
// H rows, out[0..1] = (ra, rb) and out[2 + k] = the (unclamped) crossing column of row ra + k (k < rows), derived
// exactly as the line walk derives it
__global__ void k_selftest_line(const int4* ends, const int* Hs, int n, int rows, int32_t* out) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n) return;
    const int4 e = ends[gid];
    tp_line ln;
    tp_setup_line(e.x, e.y, e.z, e.w, Hs[gid], ln);
    int32_t* o = out + (size_t)gid * (rows + 2);
    o[0] = ln.ra; o[1] = ln.rb;
    for (int k = 0; k < rows; k++) o[2 + k] = 0;
    if (ln.ra > ln.rb) return;
    const int last = min(ln.rb, ln.ra + rows - 1);
    for (int r = ln.ra; r <= last; r++) o[2 + r - ln.ra] = (int32_t)((ln.x + (int64_t)(r - ln.ra) * ln.s) >> TP_LINE_FRAC);
}
void tp_launch_selftest_line(const int4* ends, const int* H, int n, int rows, int32_t* out, hipStream_t s) {
    hipLaunchKernelGGL(k_selftest_line, dim3((n + 255) / 256), dim3(256), 0, s, ends, H, n, rows, out);
}
