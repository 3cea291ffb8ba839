// tp_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the t-pose hot path.  wave64 only.
//
// One grad-iter of the reference = two instanced draws of 13*NT triangles (mode 0: 4 same-address
// int atomics per fragment, mode 1: one) + gradient.cs + shift.cs
// (software/triangulate/main.cpp:121-155).  Here the work is organised around EDGE LINES
// (tp_raster.h, "edge-centric form"): a variant's pixel moments are the signed sum of three line
// sums W(e) = sum over the line's rows of the row-prefix sum at the line's crossing column, and the
// 13 variants of all triangles share 9 lines per undirected edge.
//
//   k_bin         per edge: snapped endpoint positions for the five vertex moves (-> vpos), bounding
//                 box of its nine lines -> the tiles the band of lines can cross -> per-tile work lists
//                 and a contiguous run of 32-byte records per edge
//   k_accumulate  THE hot kernel: three resident workgroups per CU walk 128x32-pixel tiles; a tile's RGBA8
//                 pixels are read once (32 B per lane, prefetched during the previous tile's walk), turned
//                 into per-row prefix sums of the pixel moments in LDS (12-byte packed entries, no bank
//                 conflicts), and every (edge line, tile) pair is walked by ONE lane: per row one exact
//                 crossing column from a 32.32 edge walker and ONE LDS entry.  No atomics, no per-fragment
//                 work.
//   k_reduce      per line: sum its per-tile records -> W(e)
//   k_update      per variant: signed sum of three W(e) -> exact moments -> `colnum`, `colacc`,
//                 `tenergy` (reference layout); central differences; per-vertex arrival atomics;
//                 shift.cs step.  (k_finalize + k_shift: the same as two launches, piecewise API.)
#include "tp_kernels.h"
#include <hip/hip_ext.h>

#define TW TP_TILE_W
#define TH TP_TILE_H
#define ROWLEN (TW + 1)  // exclusive prefix has TW+1 entries per row
#define ACC_THREADS 512

static_assert(TW == 128, "prefix build assumes 8 lanes x 16 pixels per row");
static_assert(TH == 32 && TH <= TP_WALK_MAXROWS, "tile height");

// LDS: the prefix table + the static-table rows bounding the tile's 32 rows (33 x 5 int64)
#define T2_LDS_WORDS ((TH + 1) * TP_T2_WORDS)
#define ACC_GRID 768
#define WALK_ROWS 4   // rows per unrolled trip of the line walk
#define ENTRY_WORDS 3  // 12-byte prefix entries: three workgroups per CU (3 x 50.9 KB of the 160 KB)
size_t tp_accumulate_lds_bytes() { return (size_t)TH * ROWLEN * ENTRY_WORDS * sizeof(uint32_t) + T2_LDS_WORDS * sizeof(int64_t); }

__device__ __forceinline__ int tile_col_of(int x, int tiles_x) { return min(x / TW, tiles_x - 1); }

// ------------------------------------------------------------------------------------------------
// static per-image table (built once per tp_set_image)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void px_moments5(uint32_t rgba, uint32_t m[5]) {
    const uint32_t r = rgba & 0xffu, g = (rgba >> 8) & 0xffu, b = (rgba >> 16) & 0xffu;
    m[0] += (r + g + b) & 1u; m[1] += r; m[2] += g; m[3] += b; m[4] += r * r + g * g + b * b;
}

// seg[r][tc][5]: moments of row r inside tile column tc
__global__ void k_static_seg(const uint8_t* img, int pitch, int W, int H, int tiles_x, uint32_t* seg) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= H * tiles_x) return;
    const int r = gid / tiles_x, tc = gid - r * tiles_x;
    const uint32_t* row = reinterpret_cast<const uint32_t*>(img + (size_t)r * pitch);
    uint32_t m[5] = {0, 0, 0, 0, 0};
    const int c1 = min((tc + 1) * TW, W);
    for (int c = tc * TW; c < c1; c++) px_moments5(row[c], m);
    for (int k = 0; k < 5; k++) seg[(size_t)gid * 5 + k] = m[k];
}
// column prefix over rows, stored shifted by one tile column: t2[r][tc+1] = sum_{r' < r} seg[r'][tc]
__global__ void k_static_cols(const uint32_t* seg, int H, int tiles_x, int64_t* t2) {
    const int tc = blockIdx.x * blockDim.x + threadIdx.x;
    if (tc >= tiles_x) return;
    int64_t acc[5] = {0, 0, 0, 0, 0};
    for (int r = 0; r <= H; r++) {
        int64_t* o = t2 + ((size_t)r * (tiles_x + 1) + tc + 1) * TP_T2_WORDS;
        for (int k = 0; k < 5; k++) o[k] = acc[k];
        if (r < H) for (int k = 0; k < 5; k++) acc[k] += seg[((size_t)r * tiles_x + tc) * 5 + k];
    }
}
// prefix over tile columns in place: t2[r][tc] = moments of rows < r, columns < tc*TW
__global__ void k_static_rows(int H, int tiles_x, int64_t* t2) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > H) return;
    int64_t run[5] = {0, 0, 0, 0, 0};
    int64_t* row = t2 + (size_t)r * (tiles_x + 1) * TP_T2_WORDS;
    for (int k = 0; k < 5; k++) row[k] = 0;
    for (int tc = 1; tc <= tiles_x; tc++)
        for (int k = 0; k < 5; k++) { run[k] += row[tc * TP_T2_WORDS + k]; row[tc * TP_T2_WORDS + k] = run[k]; }
}
void tp_launch_static_table(const uint8_t* img, int pitch, int W, int H, int tiles_x, uint32_t* seg, int64_t* t2, hipStream_t s) {
    hipLaunchKernelGGL(k_static_seg, dim3((H * tiles_x + 255) / 256), dim3(256), 0, s, img, pitch, W, H, tiles_x, seg);
    hipLaunchKernelGGL(k_static_cols, dim3((tiles_x + 63) / 64), dim3(64), 0, s, seg, H, tiles_x, t2);
    hipLaunchKernelGGL(k_static_rows, dim3((H + 1 + 255) / 256), dim3(256), 0, s, H, tiles_x, t2);
}

// ------------------------------------------------------------------------------------------------
// k_bin: BIN_EDGES edges per workgroup, 8 lanes per edge (lanes 0/1 transform the two endpoints)
// ------------------------------------------------------------------------------------------------
#define BIN_EDGES 32   // edges per workgroup, 8 lanes each (lanes 0/1 transform the endpoints)
#define BIN_LOG 5      // log2(BIN_EDGES)
#define BIN_THREADS (BIN_EDGES * 8)

__global__ __launch_bounds__(BIN_THREADS) void k_bin(tp_launch L, int epb) {  // epb <= BIN_EDGES edges per workgroup
    __shared__ int s_excl[BIN_EDGES + 1];  // exclusive scan of rectangle sizes
    __shared__ int s_rect[BIN_EDGES][4];   // tx0, ty0, ntx, #tiles
    __shared__ int s_geom[BIN_EDGES][6];   // base endpoints (Xa, Ya, Xb, Yb) and the moves' reach (dX, dY), 1/256 px
    __shared__ int s_uv[BIN_EDGES][2];
    __shared__ int2 s_pos[BIN_EDGES][2][5];
    __shared__ int s_kept[BIN_EDGES], s_rank[BIN_EDGES], s_first[BIN_EDGES + 1];  // tiles some line can cross: count, arrivals, scan
    __shared__ uint32_t s_base;
    const int tid = threadIdx.x;
    const uint32_t rebin_word = L.state->rebin_req;  // consumed late: the loads below do not wait for it
    int dbgk = 0;
#define TPB_STAMP() do { if ((L.debug & 16) && tid == 0 && blockIdx.x < 512 && dbgk < 16) L.dbg[blockIdx.x * 16 + dbgk++] = wall_clock64(); } while (0)
    TPB_STAMP();

    const int j = tid >> 3, q = tid & 7;
    const int e = j < epb ? blockIdx.x * epb + j : L.NE;  // lanes beyond epb edges idle (small meshes: more workgroups)
    int32_t xmin = INT32_MAX, xmax = INT32_MIN, ymin = INT32_MAX, ymax = INT32_MIN, dX = 0, dY = 0;
    if (e < L.NE && q < 2) {
        const int2 uv = L.edge_uv[e];
        const int vraw = q == 0 ? uv.x : uv.y;
        const int v = vraw & 0x3fffffff;
        const bool publish = (vraw >> 30) & 1;  // one edge per vertex writes vpos (k_update reads it)
        const float2 p = L.points[v];
        int32_t bx = 0, by = 0;
#pragma unroll
        for (int m = 0; m < 5; m++) {  // vertex stage for the five moves (several edges write the same values)
            int32_t X, Y;
            tp_vertex_stage(p.x, p.y, m, 0, L.vw, X, Y);
            if (publish) L.vpos[(size_t)v * 5 + m] = make_int2(X, Y);
            s_pos[j][q][m] = make_int2(X, Y);
            if (m == 0) { bx = X; by = Y; s_geom[j][2 * q] = X; s_geom[j][2 * q + 1] = Y; s_uv[j][q] = v; }
            xmin = min(xmin, X); xmax = max(xmax, X); ymin = min(ymin, Y); ymax = max(ymax, Y);
            dX = max(dX, abs(X - bx)); dY = max(dY, abs(Y - by));
        }
    }
    {  // combine the two endpoint lanes
        xmin = min(xmin, __shfl_xor(xmin, 1)); xmax = max(xmax, __shfl_xor(xmax, 1));
        ymin = min(ymin, __shfl_xor(ymin, 1)); ymax = max(ymax, __shfl_xor(ymax, 1));
        dX = max(dX, __shfl_xor(dX, 1)); dY = max(dY, __shfl_xor(dY, 1));
    }
    TPB_STAMP();
    const bool rebin = rebin_word != 0;  // lists still valid otherwise (tp_set_margin)
    if (!rebin) return;
    if (L.margin_px >= 2)  // only the margin vote of k_update reads it
        for (int v = blockIdx.x * BIN_THREADS + tid; v < L.NP; v += gridDim.x * BIN_THREADS) L.points_binned[v] = L.points[v];
    if (q == 0) {
        int tx0 = 0, ty0 = 0, ntx = 1, cnt = 0;
        if (e < L.NE) {
            const int m = L.margin_px;
            // rows whose centre lies in [ymin, ymax), crossing columns in [first_centre(xmin), first_centre(xmax)]
            const int r0 = max(tp_first_centre(ymin) - m, 0), r1 = min(tp_first_centre(ymax) - 1 + m, L.vw.H - 1);
            const int c0 = min(max(tp_first_centre(xmin) - m, 0), L.vw.W), c1 = min(max(tp_first_centre(xmax) + m, 0), L.vw.W);
            if (r0 <= r1) {
                tx0 = tile_col_of(c0, L.tiles_x); ty0 = r0 / TH;
                ntx = tile_col_of(c1, L.tiles_x) - tx0 + 1;
                cnt = ntx * (r1 / TH - ty0 + 1);
            }
        }
        s_rect[j][0] = tx0; s_rect[j][1] = ty0; s_rect[j][2] = ntx; s_rect[j][3] = cnt;
        s_geom[j][4] = dX + 256 * L.margin_px; s_geom[j][5] = dY + 256 * L.margin_px;
    }
    __syncthreads();
    if (tid < BIN_EDGES) {  // wave 0: inclusive scan of the rectangle sizes by shuffles
        int inc = s_rect[tid][3];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(inc, o);
            if (tid >= o) inc += v;
        }
        s_excl[tid + 1] = inc;
        if (tid == 0) s_excl[0] = 0;
        s_kept[tid] = 0; s_rank[tid] = 0;
    }
    __syncthreads();
    TPB_STAMP();
    const int total = s_excl[BIN_EDGES];

    // (edge, tile of its rectangle) -> which edge, which tile, and whether any of the nine lines can cross it.
    // Every sample (row, crossing column) of the nine lines lies within the base segment (+) box(dX, dY) (+)
    // [0, 1 px) in x: a tile strictly on one side of that band is never crossed.  Tiles of the first / last
    // tile column also receive the clamped columns: kept.
    auto pair_of = [&](int p, int& lo, int& tile) -> bool {
        int hi = BIN_EDGES;  // largest jj with s_excl[jj] <= p
        lo = 0;
#pragma unroll
        for (int it = 0; it < BIN_LOG; it++) {
            const int mid = (lo + hi) >> 1;
            if (s_excl[mid] <= p) lo = mid; else hi = mid;
        }
        const int k = p - s_excl[lo], ntx = s_rect[lo][2];
        const int ky = k / ntx, kx = k - ky * ntx;
        const int txx = s_rect[lo][0] + kx, tyy = s_rect[lo][1] + ky;
        tile = tyy * L.tiles_x + txx;
        const int64_t a = (int64_t)s_geom[lo][3] - s_geom[lo][1], b = -((int64_t)s_geom[lo][2] - s_geom[lo][0]);
        const int64_t slack = (a < 0 ? -a : a) * ((int64_t)s_geom[lo][4] + 256) + (b < 0 ? -b : b) * (int64_t)s_geom[lo][5];
        const int64_t x0 = 256LL * (txx * TW) + 128, x1 = 256LL * min(txx * TW + TW - 1, L.vw.W - 1) + 128;
        const int64_t y0 = 256LL * (tyy * TH) + 128, y1 = 256LL * min(tyy * TH + TH - 1, L.vw.H - 1) + 128;
        const int64_t ex0 = a * (x0 - s_geom[lo][0]), ex1 = a * (x1 - s_geom[lo][0]);
        const int64_t ey0 = b * (y0 - s_geom[lo][1]), ey1 = b * (y1 - s_geom[lo][1]);
        const int64_t emin = min(ex0, ex1) + min(ey0, ey1), emax = max(ex0, ex1) + max(ey0, ey1);
        const bool edge_col = txx == 0 || txx == L.tiles_x - 1;
        return edge_col || !(emin > slack || emax < -slack);
    };

    // ---- pass A: how many tiles every edge keeps (records are allocated for those only, contiguously per edge)
    const bool one_pass = total <= BIN_THREADS;  // the common case: the pair stays in registers for pass B
    int lo1 = 0, tile1 = -1;
    bool keep1 = false;
    for (int pass0 = 0; pass0 < total; pass0 += BIN_THREADS) {
        const int p = pass0 + tid;
        if (p < total) {
            keep1 = pair_of(p, lo1, tile1);
            if (keep1) atomicAdd(&s_kept[lo1], 1);
        }
    }
    __syncthreads();
    if (tid < BIN_EDGES) {
        int inc = s_kept[tid];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(inc, o);
            if (tid >= o) inc += v;
        }
        s_first[tid + 1] = inc;
        if (tid == 0) s_first[0] = 0;
        if (tid == BIN_EDGES - 1) {
            // record ids: every block owns a slice of the lower half of the record buffer (no global
            // atomic on the common path); a block with long edges draws from the shared upper half
            const uint32_t half = (uint32_t)L.visit_cap / 2, slice = half / gridDim.x;
            uint32_t base = blockIdx.x * slice;
            if ((uint32_t)inc > slice) {
                base = half + atomicAdd(&L.state->visit_total, (uint32_t)inc);
                if (base + (uint32_t)inc > (uint32_t)L.visit_cap) atomicOr(&L.state->flags, TP_FLAG_VISIT_OVERFLOW);
            }
            s_base = base;
        }
    }
    __syncthreads();
    const uint32_t base = s_base;

    // ---- pass B: the block groups its kept pairs by tile in an LDS hash table so that each distinct tile
    //      costs ONE returning global atomic per block; record slot = the edge's first + its arrival rank
    __shared__ int h_key[BIN_THREADS], h_cnt[BIN_THREADS], h_base[BIN_THREADS];
    for (int pass0 = 0; pass0 < total; pass0 += BIN_THREADS) {
        h_key[tid] = -1; h_cnt[tid] = 0;
        __syncthreads();
        const int p = pass0 + tid;
        int tile = -1, lo = 0, hslot = 0, rank = 0, visit = 0;
        if (p < total) {
            bool keep;
            if (one_pass) { keep = keep1; lo = lo1; tile = tile1; }
            else keep = pair_of(p, lo, tile);
            if (keep) {
                visit = (int)base + s_first[lo] + atomicAdd(&s_rank[lo], 1);
                hslot = (tile * 40503) & (BIN_THREADS - 1);
                while (true) {  // open addressing; at most BIN_THREADS distinct keys for as many slots
                    const int old = atomicCAS(&h_key[hslot], -1, tile);
                    if (old == -1 || old == tile) break;
                    hslot = (hslot + 1) & (BIN_THREADS - 1);
                }
                rank = atomicAdd(&h_cnt[hslot], 1);
            } else
                tile = -1;
        }
        __syncthreads();
        if (h_key[tid] >= 0) h_base[tid] = atomicAdd(&L.tilecount[h_key[tid]], h_cnt[tid]);
        __syncthreads();
        if (tile >= 0) {
            const int slot = h_base[hslot] + rank;
            if (slot < L.list_cap) {
                tp_list_entry en;
                en.visit = visit;
                en.edge = blockIdx.x * epb + lo;
                en.u = s_uv[lo][0]; en.v = s_uv[lo][1];
#pragma unroll
                for (int m = 0; m < 5; m++) { en.a[m] = s_pos[lo][0][m]; en.b[m] = s_pos[lo][1][m]; }
                L.tilelist[(size_t)tile * L.list_cap + slot] = en;
            } else
                atomicOr(&L.state->flags, TP_FLAG_LIST_OVERFLOW);
        }
        __syncthreads();
    }
    TPB_STAMP();
    TPB_STAMP();
    if (q == 0 && e < L.NE) {
        const long long first = (long long)base + s_first[j];
        const bool fits = first + s_kept[j] <= (long long)L.visit_cap;  // overflow is flagged; k_reduce must stay in bounds
        L.edge_visit[e] = make_int2(fits ? (int)first : 0, fits ? s_kept[j] : 0);
    }
}

void tp_launch_bin(const tp_launch& L, hipStream_t s) {
    // coarse meshes on large rasters (long edges, up to thousands of tiles each): fewer edges per workgroup
    const long long tiles = (long long)L.tiles_x * L.tiles_y;
    long long epb = 8LL * L.NE / (tiles > 0 ? tiles : 1);
    epb = epb < 1 ? 1 : epb > BIN_EDGES ? BIN_EDGES : epb;
    hipLaunchKernelGGL(k_bin, dim3((unsigned)((L.NE + epb - 1) / epb)), dim3(BIN_THREADS), 0, s, L, (int)epb);
}

// LDS prefix entry (12 bytes), per row exclusive prefix over the tile's 128 columns, 16-bit fields packed in pairs:
//   x = sum r | sum g << 16,   y = sum b | n_odd << 16,   z = sum r^2+g^2+b^2
// (128 pixels: sum of a channel <= 32640 < 2^16, n_odd <= 128, q < 2^25: nothing carries between fields, so the
// prefix build adds and scans whole words).  The walk adds the 16-bit halves into 32-bit accumulators (SDWA
// word selects, one instruction per field per row).
struct pix3 { uint32_t x, y, z; };

__device__ __forceinline__ pix3 pixel_moments(uint32_t rgba) {
    const uint32_t m = rgba & 0x00ffffffu;
    const uint32_t s = __builtin_amdgcn_udot4(m, 0x00010101u, 0u, false);  // r + g + b
    pix3 o;
    o.x = __builtin_amdgcn_perm(0u, m, 0x0c010c00u);                        // r | g << 16
    o.y = (m >> 16) | ((s & 1u) << 16);                                     // b | odd << 16
    o.z = __builtin_amdgcn_udot4(m, m, 0u, false);                          // r^2 + g^2 + b^2
    return o;
}
__device__ __forceinline__ pix3 operator+(pix3 a, pix3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }

// ------------------------------------------------------------------------------------------------
// k_accumulate
// ------------------------------------------------------------------------------------------------
// Phase-1 lane mapping: wave w owns tile rows P1_RL*w ..; lane = seg*P1_RL + rl walks the P1_PX pixels
// [P1_PX seg, P1_PX (seg+1)) of row P1_RL*w + rl sequentially, and the segments of a row are combined
// by a log2(P1_SEGS)-step scan at lane distance P1_RL.  Consecutive lanes belong to consecutive ROWS,
// whose LDS rows are 1548 B = 12 B (mod 128) apart, so the lanes of a 12-byte store group hit different banks.
#define P1_PX 8                    // pixels per lane
#define P1_SEGS (TW / P1_PX)       // lanes per tile row
#define P1_RL (64 / P1_SEGS)       // tile rows per wave
#define P1_WAVES (TH / P1_RL)
static_assert(P1_WAVES * 64 <= ACC_THREADS, "phase-1 roles");

__device__ __forceinline__ uint32_t scan_segments(uint32_t v, int seg) {
#pragma unroll
    for (int d = 1; d < P1_SEGS; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)v, P1_RL * d);
        v += seg >= d ? o : 0u;
    }
    return v;
}

__global__ __launch_bounds__(ACC_THREADS, 6) void k_accumulate(tp_launch L) {  // 6 waves per SIMD: 3 workgroups per CU
    extern __shared__ __attribute__((aligned(16))) uint32_t P[];  // [TH][ROWLEN][3], then int64 T2s[TH+1][5]
    int64_t* T2s = reinterpret_cast<int64_t*>(P + TH * ROWLEN * ENTRY_WORDS);

    const int tid = threadIdx.x;
    const int ntiles = L.tiles_x * L.tiles_y;
    const int lane = tid & 63, wave = tid >> 6;
    const int rl = lane % P1_RL, seg = lane / P1_RL, prow = wave * P1_RL + rl;  // phase-1 role
    if (blockIdx.x == 0 && tid == 0) L.state->rebin_req = 0;  // consumed by the k_bin that ran before us

    // the block walks tiles blockIdx.x, +gridDim.x, ...; the pixels of the next tile are fetched into
    // registers while the lines of the current one are walked
    uint4 px[P1_PX / 4];
    auto fetch = [&](int tile) {
        const int tx = tile % L.tiles_x, ty = tile / L.tiles_x;
        const uint8_t* src = L.img + (size_t)(ty * TH + prow) * L.pitch + (size_t)(tx * TW + seg * P1_PX) * 4;
#pragma unroll
        for (int k = 0; k < P1_PX / 4; k++) px[k] = reinterpret_cast<const uint4*>(src)[k];
    };
    int tile = blockIdx.x;
    if (tile < ntiles && wave < P1_WAVES) fetch(tile);
    int dbgk = 0;
#define TP_STAMP() do { if ((L.debug & 8) && tid == 0 && dbgk < 16) L.dbg[blockIdx.x * 16 + dbgk++] = wall_clock64(); } while (0)
    TP_STAMP();

    for (; tile < ntiles; tile += gridDim.x) {
        const int tx = tile % L.tiles_x, ty = tile / L.tiles_x;
        int nlist = L.tilecount[tile];
        if (nlist > L.list_cap) nlist = L.list_cap;
        // work unit = (edge, line, 1/split of the tile's rows): `split` adjacent lanes share a line.  The
        // walk is VALU-bound and every part repeats the line's set-up, so lines are only split while
        // all parts still fit two waves (short lists, e.g. the two-triangle start state)
        const int nlines = nlist * TP_NLINES;
        const int lsplit = (L.debug & 4) ? 0 : (L.debug & 32) ? (nlines * 4 <= ACC_THREADS ? 2 : nlines * 2 <= ACC_THREADS ? 1 : 0) : nlines * 4 <= 128 ? 2 : nlines * 2 <= 128 ? 1 : 0;  // log2(split)
        const int split = 1 << lsplit;
        const int nitems = nlines << lsplit;
        const tp_list_entry* list = L.tilelist + (size_t)tile * L.list_cap;
        int item = tid;
        // this lane's first work item: record slot and the two endpoints of its line, fetched now
        // so that nothing after the barrier waits on global memory
        const tp_list_entry* e0 = list + (item < nitems ? (item >> lsplit) / TP_NLINES : 0);
        const int ver0 = (item >> lsplit) % TP_NLINES;
        int visit = e0->visit;
        int2 A = e0->a[(ver0 >= 1 && ver0 <= 4) ? ver0 : 0], B = e0->b[ver0 >= 5 ? ver0 - 4 : 0];
        const bool stale = L.margin_px >= 2;  // lists reused across iterations: positions come from vpos
        const int next = tile + gridDim.x;
        // static-table rows for this tile's row boundaries -> LDS
        if (tid < T2_LDS_WORDS && nlist > 0) {
            const int k = tid, rr = k / TP_T2_WORDS, ww = k - rr * TP_T2_WORDS;
            const int rabs = min(ty * TH + rr, L.vw.H);
            T2s[k] = L.t2[((size_t)rabs * (L.tiles_x + 1) + tx) * TP_T2_WORDS + ww];
        }

        // ---- phase 1: pixels -> row prefix sums in LDS --------------------------------------
        if (wave < P1_WAVES && nlist > 0 && !(L.debug & 1)) {
            pix3 loc[P1_PX];  // exclusive prefix inside the lane's segment
            pix3 run = {0, 0, 0};
#pragma unroll
            for (int k = 0; k < P1_PX; k++) {
                const uint32_t w = k % 4 == 0 ? px[k / 4].x : k % 4 == 1 ? px[k / 4].y : k % 4 == 2 ? px[k / 4].z : px[k / 4].w;
                loc[k] = run;
                run = run + pixel_moments(w);
            }
            pix3 ex;  // everything left of the segment
            ex.x = scan_segments(run.x, seg) - run.x;
            ex.y = scan_segments(run.y, seg) - run.y;
            ex.z = scan_segments(run.z, seg) - run.z;
            uint32_t* row = P + (prow * ROWLEN + seg * P1_PX) * ENTRY_WORDS;
#pragma unroll
            for (int k = 0; k < P1_PX; k++) {
                const pix3 e = ex + loc[k];
                row[3 * k] = e.x; row[3 * k + 1] = e.y; row[3 * k + 2] = e.z;
            }
            if (seg == P1_SEGS - 1) {
                const pix3 e = ex + run;
                row[3 * P1_PX] = e.x; row[3 * P1_PX + 1] = e.y; row[3 * P1_PX + 2] = e.z;
            }
        }
        if (next < ntiles && wave < P1_WAVES) fetch(next);  // in flight during phase 2
        TP_STAMP();
        __syncthreads();
        TP_STAMP();

        // ---- phase 2: one lane per (edge line, tile) -------------------------------------------
        const int row0 = ty * TH;
        const int row1 = min(row0 + TH - 1, L.vw.H - 1);
        const int col0 = tx * TW;
        const int W = L.vw.W;

        if (!(L.debug & 2))
        for (; item < nitems; item += ACC_THREADS) {
            const int part = item & (split - 1), line = item >> lsplit;
            const int en = line / TP_NLINES, ver = line - en * TP_NLINES;
            const int mu = (ver >= 1 && ver <= 4) ? ver : 0, mv = ver >= 5 ? ver - 4 : 0;
            if (item != tid) {  // rare: more than 512 lines in this tile
                const tp_list_entry* ee = list + en;
                visit = ee->visit;
                A = stale ? L.vpos[(size_t)ee->u * 5 + mu] : ee->a[mu];
                B = stale ? L.vpos[(size_t)ee->v * 5 + mv] : ee->b[mv];
            } else if (stale) {
                const tp_list_entry* ee = list + en;
                A = L.vpos[(size_t)ee->u * 5 + mu]; B = L.vpos[(size_t)ee->v * 5 + mv];
            }
            tp_edge_walk ew;
            const int pr = TH >> lsplit;  // rows per part
            tp_setup_edge(A.x, A.y, B.x, B.y, row0 + pr * part, min(row0 + pr * part + pr - 1, row1), ew);
#define TP_STAMP_AT(slot) do { if ((L.debug & 8) && tid == 0 && item == tid) L.dbg[blockIdx.x * 16 + (slot) + (tile == (int)blockIdx.x ? 0 : 3)] = wall_clock64(); } while (0)
            TP_STAMP_AT(9);
            // eight rows per trip, fully unrolled and predicated so that the eight prefix reads are in
            // flight together: rows outside the line's rows, or whose crossing column falls into another
            // tile column, read the all-zero entry P[r][0] and are not counted
            uint32_t ar = 0, ag = 0, ab = 0, ao = 0, aq = 0;  // 32 rows: channel sums < 2^20, q < 2^30
            uint32_t sx = 0, nin = 0;
            int32_t first = INT32_MAX;
            const int rbase = row0 + pr * part;
            const uint32_t nvalid = (uint32_t)max(ew.rb - ew.ra + 1, 0);
            // columns of this tile column: [col0, col0 + TW), the last one also takes the clamp value W
            const uint32_t lim = tx == L.tiles_x - 1 ? (uint32_t)(W - col0 + 1) : (uint32_t)TW;
            int64_t xw = ew.w.x - (int64_t)(ew.ra - rbase) * ew.w.s;  // walker moved back to row rbase
            int rowi = (rbase - row0) * ROWLEN * ENTRY_WORDS;  // 32-bit LDS word index of the trip's first row
            for (int c0 = 0; c0 < pr; c0 += WALK_ROWS, rowi += WALK_ROWS * ROWLEN * ENTRY_WORDS) {
                const int koff = ew.ra - rbase - c0;  // chunk-relative index of the first valid row
                if (!__any((int)nvalid + koff > 0 && koff < WALK_ROWS)) { xw += WALK_ROWS * ew.w.s; continue; }
                pix3 ent[WALK_ROWS];
                uint32_t inmask = 0;
#pragma unroll
                for (int k = 0; k < WALK_ROWS; k++) {
                    const int32_t x = min(max((int32_t)(xw >> 32), 0), W);
                    xw += ew.w.s;
                    const uint32_t xl = (uint32_t)(x - col0);
                    const bool in = xl < lim && (uint32_t)(k - koff) < nvalid;
                    const int ei = rowi + k * ROWLEN * ENTRY_WORDS + (in ? (int)(xl * ENTRY_WORDS) : 0);
                    ent[k].x = P[ei]; ent[k].y = P[ei + 1]; ent[k].z = P[ei + 2];
                    sx += in ? (uint32_t)x : 0u;
                    inmask |= in ? (1u << k) : 0u;
                }
#pragma unroll
                for (int k = 0; k < WALK_ROWS; k++) {
                    ar += ent[k].x & 0xffffu; ag += ent[k].x >> 16;
                    ab += ent[k].y & 0xffffu; ao += ent[k].y >> 16;
                    aq += ent[k].z;
                }
                if (inmask) {
                    nin += (uint32_t)__builtin_popcount(inmask);
                    first = min(first, rbase + c0 + (int)__builtin_ctz(inmask));
                }
            }
            TP_STAMP_AT(10);
            // combine the parts (adjacent lanes; a line's lanes are always active together)
            for (int o = 1; o < split; o <<= 1) {
                sx += (uint32_t)__shfl_xor((int)sx, o);
                nin += (uint32_t)__shfl_xor((int)nin, o);
                first = min(first, __shfl_xor(first, o));
                ar += (uint32_t)__shfl_xor((int)ar, o); ag += (uint32_t)__shfl_xor((int)ag, o);
                ab += (uint32_t)__shfl_xor((int)ab, o); ao += (uint32_t)__shfl_xor((int)ao, o);
                aq += (uint32_t)__shfl_xor((int)aq, o);
            }
            if (part != 0) continue;
            // rows that count are contiguous (the line is monotone): add everything left of this tile
            // column for them from the static table
            int64_t st[TP_T2_WORDS] = {0, 0, 0, 0, 0};
            if (nin) {
                const int64_t* t0 = T2s + (first - row0) * TP_T2_WORDS;
                const int64_t* t1 = T2s + (first - row0 + (int)nin) * TP_T2_WORDS;
#pragma unroll
                for (int k = 0; k < TP_T2_WORDS; k++) st[k] = t1[k] - t0[k];
            }
            TP_STAMP_AT(11);
            if (visit < L.visit_cap) {  // 32-byte record: two 16-byte stores
                uint4* out = reinterpret_cast<uint4*>(L.visits + ((size_t)visit * TP_NLINES + ver) * TP_REC_DWORDS);
                const uint64_t q = (uint64_t)((int64_t)aq + st[4]);
                out[0] = make_uint4(sx, (uint32_t)((int64_t)ao + st[0]), (uint32_t)((int64_t)ar + st[1]), (uint32_t)((int64_t)ag + st[2]));
                out[1] = make_uint4((uint32_t)((int64_t)ab + st[3]), 0u, (uint32_t)q, (uint32_t)(q >> 32));
            }
        }
        TP_STAMP();
        if (next < ntiles || (L.debug & 8)) __syncthreads();  // the table is rebuilt for the next tile
        TP_STAMP();
    }
}

static int accumulate_grid(const tp_launch& L) {
    // every workgroup resident at once (3 per CU by LDS and registers); each walks its tiles with prefetch
    const int ntiles = L.tiles_x * L.tiles_y;
    return ntiles < ACC_GRID ? ntiles : ACC_GRID;
}

hipError_t tp_kernels_init() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_accumulate),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)tp_accumulate_lds_bytes());
}

void tp_launch_accumulate(const tp_launch& L, hipStream_t s) {
    hipLaunchKernelGGL(k_accumulate, dim3(accumulate_grid(L)), dim3(ACC_THREADS), tp_accumulate_lds_bytes(), s, L);
}

// same launch with the dispatch's own begin/end timestamps recorded into two events
void tp_launch_accumulate_timed(const tp_launch& L, hipStream_t s, hipEvent_t start, hipEvent_t stop) {
    hipExtLaunchKernelGGL(k_accumulate, dim3(accumulate_grid(L)), dim3(ACC_THREADS), tp_accumulate_lds_bytes(), s,
                          start, stop, 0, L);
}

// ------------------------------------------------------------------------------------------------
// k_reduce: W(line) = sum of its per-tile records; one thread per (edge, line, word)
// ------------------------------------------------------------------------------------------------
template <int P>  // P adjacent lanes share one sum (coarse meshes: hundreds of records per line)
__global__ __launch_bounds__(256) void k_reduce(tp_launch L) {
    const int tidg = blockIdx.x * blockDim.x + threadIdx.x;
    const int gid = tidg / P, part = tidg % P;
    if (L.margin_px < 2) {
        // work lists are rebuilt every iteration: k_accumulate has consumed them, re-arm them here
        // (with a margin, k_update's vote decides)
        for (int k = tidg; k < L.tiles_x * L.tiles_y; k += gridDim.x * blockDim.x) L.tilecount[k] = 0;
        if (tidg == 0) { L.state->visit_total = 0; L.state->rebin_req = 1; L.state->rebin_count++; }
    }
    const int per_edge = TP_NLINES * TP_W_WORDS;  // 54 consecutive int64 per edge in wline
    if (gid >= L.NE * per_edge) return;
    const int e = gid / per_edge, lw = gid - e * per_edge, line = lw / TP_W_WORDS, w = lw - line * TP_W_WORDS;
    const int2 ev = L.edge_visit[e];  // first record and number of records (tiles a line of this edge can cross)
    // records are 8 dwords per line: fields 0..4 are u32, the q field a u64 at dwords 6..7 (its low half is
    // loaded like a u32 field, the high half by a second load that only the q threads issue)
    const int stride = TP_NLINES * TP_REC_DWORDS;
    const uint32_t* src = L.visits + ((size_t)ev.x * TP_NLINES + line) * TP_REC_DWORDS + (w < 5 ? w : 6);
    const bool wide = w == 5;
    uint64_t acc = 0;
    uint32_t acch = 0;
    int k = part;  // lane `part` of the P sharing this sum takes records part, part + P, ...
    for (; k + 15 * P < ev.y; k += 16 * P) {  // long edges: sixteen loads in flight per trip
        uint32_t v[16], h[16];
#pragma unroll
        for (int u = 0; u < 16; u++) { v[u] = src[(size_t)(k + u * P) * stride]; h[u] = wide ? src[(size_t)(k + u * P) * stride + 1] : 0u; }
#pragma unroll
        for (int u = 0; u < 16; u++) { acc += v[u]; acch += h[u]; }
    }
    while (k < ev.y) {  // the usual few records: up to eight loads in flight
        uint32_t v[8], h[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const bool on = k + u * P < ev.y;
            v[u] = on ? src[(size_t)(k + u * P) * stride] : 0u;
            h[u] = (on && wide) ? src[(size_t)(k + u * P) * stride + 1] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) { acc += v[u]; acch += h[u]; }
        k += 8 * P;
    }
    acc += (uint64_t)acch << 32;
#pragma unroll
    for (int o = 1; o < P; o <<= 1)  // the P lanes are adjacent and always active together
        acc += ((uint64_t)(uint32_t)__shfl_xor((int)(acc >> 32), o) << 32 | (uint32_t)__shfl_xor((int)(uint32_t)acc, o)) ;
    if (part != 0) return;
    L.wline[gid] = (int64_t)acc;
}
void tp_launch_reduce(const tp_launch& L, hipStream_t s) {
    const int n = L.NE * TP_NLINES * TP_W_WORDS;
    // records per line grow with tiles per edge: several lanes per sum when there are more tiles than edges
    const long long tiles = (long long)L.tiles_x * L.tiles_y;
    if (tiles < L.NE) hipLaunchKernelGGL(k_reduce<1>, dim3((n + 255) / 256), dim3(256), 0, s, L);
    else if (tiles < 8LL * L.NE) hipLaunchKernelGGL(k_reduce<4>, dim3((n * 4 + 255) / 256), dim3(256), 0, s, L);
    else hipLaunchKernelGGL(k_reduce<16>, dim3((n * 16 + 255) / 256), dim3(256), 0, s, L);
}

// ------------------------------------------------------------------------------------------------
// per-variant moments = signed sum of three line sums
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ tp_moments variant_moments(const tp_launch& L, int t, int i) {
    const int4 tri = L.tris[t];
    const int vid[3] = {tri.x, tri.y, tri.z};
    const int ms = i > 0 ? (i - 1) >> 2 : 3, mm = i > 0 ? ((i - 1) & 3) + 1 : 0;
    int32_t X[3], Y[3], c[3];
#pragma unroll
    for (int s = 0; s < 3; s++) {
        const int2 q = L.vpos[(size_t)vid[s] * 5 + (s == ms ? mm : 0)];
        X[s] = q.x; Y[s] = q.y;
    }
    tp_variant_coeffs(X, Y, c);
    int64_t m[TP_W_WORDS] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int he = L.he_edge[3 * t + k];
        const int64_t* w = L.wline + ((size_t)(he >> 1) * TP_NLINES + tp_edge_version(i, k, he & 1)) * TP_W_WORDS;
#pragma unroll
        for (int q = 0; q < TP_W_WORDS; q++) m[q] += (int64_t)c[k] * w[q];
    }
    tp_moments r = {m[0], m[1], m[2], m[3], m[4], m[5]};
    return r;
}

__device__ __forceinline__ int32_t emit_variant(const tp_launch& L, int flavour, int t, int i, const tp_moments& m,
                                                bool write_moments) {
    const int id = i * L.NT + t;
    int64_t E;
    if (flavour == 0) {
        E = tp_energy_triangulate(m);
        L.ca[id] = make_int4(tp_wrap32(m.sr), tp_wrap32(m.sg), tp_wrap32(m.sb), 0);
    } else {
        const int4 col = L.ca[id];  // stored colour, replicated x13 by upload
        E = tp_energy64(m, col.x, col.y, col.z);
    }
    const int32_t e32 = tp_wrap32(E);
    L.ten[id] = e32;
    L.cn[id] = tp_wrap32(m.n);
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
    if (gid < 4) return;  // shift.cs:20 -- the four corners never move

    float tgx = (float)(int)gx, tgy = (float)(int)gy;
    float2 p = L.points[gid];
    const float R = L.vw.ratio;
    if (p.x <= -R) { p.x = -R; tgx = 0.0f; } else if (p.x >= R) { p.x = R; tgx = 0.0f; }
    if (p.y <= -1.0f) { p.y = -1.0f; tgy = 0.0f; } else if (p.y >= 1.0f) { p.y = 1.0f; tgy = 0.0f; }
    // p -= rate * tgr / 256 / 256  (shift.cs:45), one rounding per operation
    p.x = tp_fsub(p.x, tp_fdiv(tp_fdiv(tp_fmul(rate, tgx), 256.0f), 256.0f));
    p.y = tp_fsub(p.y, tp_fdiv(tp_fdiv(tp_fmul(rate, tgy), 256.0f), 256.0f));
    L.points[gid] = p;
}
void tp_launch_shift(const tp_launch& L, float rate, hipStream_t s) {
    hipLaunchKernelGGL(k_shift, dim3((L.NP + 255) / 256), dim3(256), 0, s, L, rate);
}

// ------------------------------------------------------------------------------------------------
// k_update: k_finalize + k_shift in ONE launch (used by tp_iterate).  One thread per variant; the
// four displacements of a vertex slot sit in adjacent lanes, so the central differences are two
// shuffles.  The quad leader adds them to its vertex with one returning 64-bit atomic per component
// -- (difference << 32) + 1 -- so the thread that completes the vertex's arrival count already
// holds the whole (wrapping int32) gradient component and takes the shift.cs step for it.  x and y
// never interact in shift.cs, so they settle independently; integer sums commute, so the result
// does not depend on arrival order.  The last block to finish knows whether any vertex left its
// work-list margin and re-arms the lists for the next k_bin.
// ------------------------------------------------------------------------------------------------
#define UPD_THREADS 64  // small workgroups: 13 NT threads are only ~600 waves, spread them over all CUs
__global__ __launch_bounds__(UPD_THREADS) void k_update(tp_launch L, int flavour, float rate) {
    __shared__ int s_last;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    // a work list overflowed in this or an earlier iteration: the line sums are incomplete.  Do not
    // step -- the host grows the lists and replays from the last good iteration (check_flags)
    if (L.state->flags) return;
    if (gid == 0) L.state->iters_done++;

    // threads [0, 12 NT): quads (t, s, k); threads [12 NT, 13 NT): the base variants
    const int NT = L.NT;
    const bool live = gid < 13 * NT;
    int t = 0, i = 0;
    if (gid < 12 * NT) { t = gid / 12; i = gid - 12 * t + 1; }
    else if (live) { t = gid - 12 * NT; i = 0; }
    const bool leader = live && i > 0 && ((i - 1) & 3) == 0;
    // the quad leader's vertex data does not depend on the energies: fetch it early
    int v = 0, deg = 0;
    float2 p = make_float2(0.0f, 0.0f), pb = p;
    if (leader) {
        const int s = (i - 1) >> 2;
        const int4 tri = L.tris[t];
        v = s == 0 ? tri.x : s == 1 ? tri.y : tri.z;
        deg = L.vtx_off[v + 1] - L.vtx_off[v];
        p = L.points[v];
        if (L.margin_px >= 2) pb = L.points_binned[v];
    }
    int32_t e = 0;
    if (live) e = emit_variant(L, flavour, t, i, variant_moments(L, t, i), false);
    // central differences inside the quad: lanes 4q+0/1 hold E(+dx)/E(-dx), 4q+2/3 E(+dy)/E(-dy)
    const uint32_t e1 = (uint32_t)__shfl_xor(e, 1);
    const uint32_t gx = (uint32_t)e - e1;                     // valid on even lanes of the quad
    const uint32_t gy = (uint32_t)__shfl_down((int)gx, 2);    // lane 4q+0 fetches lane 4q+2's value
    int need = 0;
    if (leader) {
        const float R = L.vw.ratio;
        const float lim = (float)(L.margin_px - 1);
        // both components settle with one returning atomic each, issued back to back
        const unsigned long long ox = atomicAdd(&L.gacc[2 * v], ((unsigned long long)gx << 32) + 1ull);
        const unsigned long long oy = atomicAdd(&L.gacc[2 * v + 1], ((unsigned long long)gy << 32) + 1ull);
        if ((int)(ox & 0xffffffffull) == deg - 1) {
            const uint32_t tot = (uint32_t)(ox >> 32) + gx;
            L.gacc[2 * v] = 0ull;
            reinterpret_cast<int*>(L.gr)[2 * v] = (int)tot;
            if (v >= 4) {
                float x = p.x, tg = (float)(int)tot;
                if (x <= -R) { x = -R; tg = 0.0f; } else if (x >= R) { x = R; tg = 0.0f; }
                x = tp_fsub(x, tp_fdiv(tp_fdiv(tp_fmul(rate, tg), 256.0f), 256.0f));
                reinterpret_cast<float*>(L.points)[2 * v] = x;
                need |= !(fabsf(x - pb.x) * (L.vw.halfW / R) <= lim);
            }
        }
        if ((int)(oy & 0xffffffffull) == deg - 1) {
            const uint32_t tot = (uint32_t)(oy >> 32) + gy;
            L.gacc[2 * v + 1] = 0ull;
            reinterpret_cast<int*>(L.gr)[2 * v + 1] = (int)tot;
            if (v >= 4) {
                float y = p.y, tg = (float)(int)tot;
                if (y <= -1.0f) { y = -1.0f; tg = 0.0f; } else if (y >= 1.0f) { y = 1.0f; tg = 0.0f; }
                y = tp_fsub(y, tp_fdiv(tp_fdiv(tp_fmul(rate, tg), 256.0f), 256.0f));
                reinterpret_cast<float*>(L.points)[2 * v + 1] = y;
                need |= !(fabsf(y - pb.y) * L.vw.halfH <= lim);
            }
        }
    }
    // vertices no triangle uses get no arrival, but shift.cs still clamps them to the domain (shift.cs:25-43
    // runs for every i in [4, NPoints); their gradient is never touched)
    if (gid >= 4 && gid < L.NP && L.vtx_off[gid + 1] == L.vtx_off[gid]) {
        float2 q = L.points[gid];
        const float R = L.vw.ratio;
        q.x = q.x <= -R ? -R : (q.x >= R ? R : q.x);
        q.y = q.y <= -1.0f ? -1.0f : (q.y >= 1.0f ? 1.0f : q.y);
        L.points[gid] = q;
    }
    if (L.margin_px < 2) return;  // no margin: k_reduce re-arms the lists every iteration
    need = __syncthreads_or(need);
    if (threadIdx.x == 0) {
        const uint32_t old = atomicAdd(&L.state->arrive, 1u + (need ? 0x10000u : 0u));
        const uint32_t now = old + 1u + (need ? 0x10000u : 0u);
        s_last = ((now & 0xffffu) == gridDim.x) ? ((now >> 16) ? 2 : 1) : 0;
    }
    __syncthreads();
    if (s_last) {
        if (s_last == 2)
            for (int k = threadIdx.x; k < L.tiles_x * L.tiles_y; k += blockDim.x) L.tilecount[k] = 0;
        if (threadIdx.x == 0) {
            L.state->arrive = 0;
            if (s_last == 2) { L.state->visit_total = 0; L.state->rebin_req = 1; L.state->rebin_count++; }
        }
    }
}
void tp_launch_update(const tp_launch& L, int flavour, float rate, hipStream_t s) {
    const int n = 13 * L.NT > L.NP ? 13 * L.NT : L.NP;  // one thread per variant, and at least one per vertex
    hipLaunchKernelGGL(k_update, dim3((n + UPD_THREADS - 1) / UPD_THREADS), dim3(UPD_THREADS), 0, s, L, flavour, rate);
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

// ------------------------------------------------------------------------------------------------
// launch-overhead probes (debug entry tp_debug_null_launch; not part of the product path)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(ACC_THREADS) void k_probe(const uint4* src, uint4* dst, int mode, int n16) {
    extern __shared__ __attribute__((aligned(16))) uint4 PP[];
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (mode == 1) {  // 8 MB of record-like stores
        dst[gid] = make_uint4(gid, 1, 2, 3);
        dst[gid + gridDim.x * blockDim.x] = make_uint4(gid, 4, 5, 6);
    } else if (mode == 2) {  // read 16 MB, no compute
        uint4 a = make_uint4(0, 0, 0, 0);
        for (int k = gid; k < n16; k += gridDim.x * blockDim.x) { const uint4 v = src[k]; a.x ^= v.x; a.y ^= v.y; a.z ^= v.z; a.w ^= v.w; }
        if ((a.x ^ a.y ^ a.z ^ a.w) == 0x12345678u) dst[gid] = a;
    } else if (mode == 5) {  // where do the waves of a workgroup land?  HW_ID: wave, SIMD, CU, SH, SE (+ XCC_ID)
        if ((threadIdx.x & 63) == 0) {
            const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);   // HW_REG_HW_ID, all 32 bits
            const uint32_t xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);  // HW_REG_XCC_ID
            reinterpret_cast<uint2*>(dst)[blockIdx.x * 16 + (threadIdx.x >> 6)] = make_uint2(hw, xcc);
        }
        __builtin_amdgcn_s_sleep(100);
    } else if (mode == 4) {
        PP[threadIdx.x] = make_uint4(gid, 0, 0, 0);
        __syncthreads();
        if (PP[(threadIdx.x + 1) & (ACC_THREADS - 1)].x == 0xffffffffu) dst[gid] = PP[0];
    }
}
void tp_launch_probe(const void* src, void* dst, int mode, int n16, int blocks, int threads, size_t lds, hipStream_t s,
                     hipEvent_t start, hipEvent_t stop) {
    hipExtLaunchKernelGGL(k_probe, dim3(blocks), dim3(threads), lds, s, start, stop, 0, (const uint4*)src, (uint4*)dst, mode, n16);
}
