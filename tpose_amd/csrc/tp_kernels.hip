// tp_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the t-pose hot path.  wave64 only.
//
// One grad-iter of the reference = two instanced draws of 13*NT triangles (mode 0: 4 same-address
// int atomics per fragment, mode 1: one) + gradient.cs + shift.cs
// (software/triangulate/main.cpp:121-155).  Here the work is organised around EDGE LINES
// (tp_raster.h, "edge-centric form"): a variant's pixel moments are the signed sum of three line
// sums W(e) = sum over the line's rows of the row-prefix sum at the line's crossing column, and the
// 13 variants of all triangles share 9 lines per undirected edge.  Three kernels per grad-iter:
//
//   k_bin         per edge (16 lanes): vertex stage of both endpoints for the five moves, the nine lines set
//                 up ONCE as whole-line 24.40 walkers (line table), then the tiles the band of lines can touch,
//                 enumerated tile row by tile row -> per-tile work lists of (edge, record slot)
//   k_accumulate  THE hot kernel: one 256-thread workgroup per 128x16-pixel tile (six resident per CU, the
//                 dispatcher balances the rest); the tile's RGBA8 pixels are read once (32 B per lane), turned
//                 into per-row prefix sums of the pixel moments in LDS (12-byte packed entries, DPP row scan,
//                 conflict-free padded layout), and every (edge line, tile) pair is walked by one to four lanes:
//                 per row one exact crossing column from the line's walker and ONE LDS entry.  No atomics, no
//                 per-fragment work.
//   k_update      per variant: signed sum of the records of its three lines -> exact moments -> `colnum`,
//                 `colacc`, `tenergy` (reference layout); central differences; per-vertex arrival atomics;
//                 shift.cs step; re-arms the work lists.  (k_finalize + k_shift: the same as two launches,
//                 piecewise API.)
#include "tp_kernels.h"
#include <hip/hip_ext.h>

#define TW TP_TILE_W
#define TH TP_TILE_H
#define ACC_THREADS (16 * TH)  // 16 lanes (8 pixels each) per tile row

static_assert(TW == 128, "prefix build: 16 lanes x 8 pixels per row, 16-bit channel sums");
static_assert(TH % 4 == 0 && TH <= TP_WALK_MAXROWS && ACC_THREADS % 64 == 0, "tile height");

// LDS prefix table: entry x of a row (x = 0..128, exclusive prefix over the tile's columns) lives at word
// 3 x + (x >> 3): one pad word after every eight entries, so that the sixteen lanes of a row -- each storing
// eight consecutive entries -- start 25 words apart and a 12-byte store group (8 lanes) touches 24 distinct banks.
#define ROW_WORDS 404  // 3 * 129 + 16, rounded up to a multiple of 4
#define T2_LDS_WORDS ((TH + 1) * TP_T2_WORDS)
size_t tp_accumulate_lds_bytes() { return (size_t)TH * ROW_WORDS * sizeof(uint32_t) + T2_LDS_WORDS * sizeof(int64_t); }

// DPP moves inside a row of 16 lanes; lanes without a source read 0
template <int CTRL>
__device__ __forceinline__ uint32_t dpp(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false); }
#define DPP_ROW_SHR(n) (0x110 + (n))
#define DPP_ROW_ROR(n) (0x120 + (n))
__device__ __forceinline__ uint32_t row_scan16(uint32_t v) {  // inclusive scan over the 16 lanes of a row
    v += dpp<DPP_ROW_SHR(1)>(v); v += dpp<DPP_ROW_SHR(2)>(v); v += dpp<DPP_ROW_SHR(4)>(v); v += dpp<DPP_ROW_SHR(8)>(v);
    return v;
}
__device__ __forceinline__ int row_max16(int v) {  // maximum over the 16 lanes of a row, in every lane
    v = max(v, (int)dpp<DPP_ROW_ROR(8)>((uint32_t)v)); v = max(v, (int)dpp<DPP_ROW_ROR(4)>((uint32_t)v));
    v = max(v, (int)dpp<DPP_ROW_ROR(2)>((uint32_t)v)); v = max(v, (int)dpp<DPP_ROW_ROR(1)>((uint32_t)v));
    return v;
}

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
// k_bin: BIN_EDGES edges per workgroup, 16 lanes per edge
// ------------------------------------------------------------------------------------------------
#define BIN_EDGES 16
#define BIN_THREADS (BIN_EDGES * 16)

__global__ __launch_bounds__(BIN_THREADS) void k_bin(tp_launch L, int epb) {  // epb <= BIN_EDGES edges per workgroup
    __shared__ int s_cnt[BIN_EDGES];     // tiles kept per edge
    __shared__ int s_first[BIN_EDGES];   // exclusive scan
    __shared__ uint32_t s_base;
    const int tid = threadIdx.x;
    const uint32_t rebin_word = L.state->rebin_req;  // consumed late: the loads below do not wait for it
    const int j = tid >> 4, q = tid & 15;
    const int e = j < epb ? blockIdx.x * epb + j : L.NE;  // lanes beyond epb edges idle (coarse meshes: more workgroups)
    tp_band band = {0, 0, 0, 0, 0, 0};
    int dX = 0, dY = 0;
    if (e < L.NE) {
        const int2 uv = L.edge_uv[e];
        const int u = uv.x & 0x3fffffff, v = uv.y & 0x3fffffff;
        const float2 pu = L.points[u], pv = L.points[v];
        tp_vertex_stage(pu.x, pu.y, 0, 0, L.vw, band.Xa, band.Ya);
        tp_vertex_stage(pv.x, pv.y, 0, 0, L.vw, band.Xb, band.Yb);
        if (q < TP_NLINES) {  // lane q sets up line q: endpoint u displaced by move mu, endpoint v by move mv
            const int mu = (q >= 1 && q <= 4) ? q : 0, mv = q >= 5 ? q - 4 : 0;
            int32_t Xa, Ya, Xb, Yb;
            tp_vertex_stage(pu.x, pu.y, mu, 0, L.vw, Xa, Ya);
            tp_vertex_stage(pv.x, pv.y, mv, 0, L.vw, Xb, Yb);
            // one edge per vertex publishes its snapped positions (k_update reads them)
            if (mv == 0 && ((uv.x >> 30) & 1)) L.vpos[(size_t)u * 5 + mu] = make_int2(Xa, Ya);
            if (mu == 0 && ((uv.y >> 30) & 1)) L.vpos[(size_t)v * 5 + mv] = make_int2(Xb, Yb);
            tp_line ln;
            tp_setup_line(Xa, Ya, Xb, Yb, L.vw.H, ln);
            L.line_xs[(size_t)e * TP_NLINES + q] = make_longlong2(ln.x, ln.s);
            L.line_rows[(size_t)e * TP_NLINES + q] = make_int2(ln.ra, ln.rb);
            dX = max(abs(Xa - band.Xa), abs(Xb - band.Xb));
            dY = max(abs(Ya - band.Ya), abs(Yb - band.Yb));
        }
    }
    if (rebin_word == 0) return;  // lists still valid (tp_set_margin)
    if (L.margin_px >= 2)  // only the margin vote of k_update reads it
        for (int v = blockIdx.x * BIN_THREADS + tid; v < L.NP; v += gridDim.x * BIN_THREADS) L.points_binned[v] = L.points[v];
    band.dX = row_max16(dX) + 256 * L.margin_px;
    band.dY = row_max16(dY) + 256 * L.margin_px;

    // ---- pass A: tiles per edge, tile row by tile row (lane q takes tile rows ty0 + q, + 16, ...)
    int ty0 = 0, ty1 = -1;
    if (e < L.NE) {
        int32_t r0, r1;
        tp_band_rows(band, L.vw.H, r0, r1);
        if (r0 <= r1) { ty0 = r0 / TH; ty1 = r1 / TH; }
    }
    int cnt = 0;
    for (int ty = ty0 + q; ty <= ty1; ty += 16) {
        int32_t tx0, tx1;
        const int row0 = ty * TH;
        if (tp_band_cols(band, row0, min(row0 + TH - 1, L.vw.H - 1), L.vw.W, TW, L.tiles_x, tx0, tx1)) cnt += tx1 - tx0 + 1;
    }
    const int inc = (int)row_scan16((uint32_t)cnt);
    if (q == 15) s_cnt[j] = inc;
    __syncthreads();
    if (tid < 64) {  // wave 0: scan of the per-edge counts, record slots for the block
        int v = tid < BIN_EDGES ? s_cnt[tid] : 0;
        const int incl = (int)row_scan16((uint32_t)v);  // BIN_EDGES == 16: one row
        if (tid < BIN_EDGES) s_first[tid] = incl - v;
        if (tid == BIN_EDGES - 1) {
            // record slots: every block owns a slice of the lower half of the record buffer (no global
            // atomic on the common path); a block with long edges draws from the shared upper half
            const uint32_t half = (uint32_t)L.visit_cap / 2, slice = half / gridDim.x;
            uint32_t base = blockIdx.x * slice;
            if ((uint32_t)incl > slice) {
                base = half + atomicAdd(&L.state->visit_total, (uint32_t)incl);
                if (base + (uint32_t)incl > (uint32_t)L.visit_cap) atomicOr(&L.state->flags, TP_FLAG_VISIT_OVERFLOW);
            }
            s_base = base;
        }
    }
    __syncthreads();
    // ---- pass B: one returning atomic per (edge, tile) -> list slot; the entry names the edge and its record slot
    const long long first = (long long)s_base + s_first[j];
    int visit = (int)first + (inc - cnt);
    for (int ty = ty0 + q; ty <= ty1; ty += 16) {
        int32_t tx0, tx1;
        const int row0 = ty * TH;
        if (!tp_band_cols(band, row0, min(row0 + TH - 1, L.vw.H - 1), L.vw.W, TW, L.tiles_x, tx0, tx1)) continue;
        for (int tx = tx0; tx <= tx1; tx++, visit++) {
            const int tile = ty * L.tiles_x + tx;
            const int slot = atomicAdd(&L.tilecount[tile], 1);
            if (slot < L.list_cap) L.tilelist[(size_t)tile * L.list_cap + slot] = make_int2(e, visit);
            else atomicOr(&L.state->flags, TP_FLAG_LIST_OVERFLOW);
        }
    }
    if (q == 15 && e < L.NE) {
        const bool fits = first + inc <= (long long)L.visit_cap;  // overflow is flagged; readers must stay in bounds
        L.edge_visit[e] = make_int2(fits ? (int)first : 0, fits ? inc : 0);
    }
}

void tp_launch_bin(const tp_launch& L, hipStream_t s) {
    // coarse meshes on large rasters (long edges, hundreds of tiles each): fewer edges per workgroup
    const long long tiles = (long long)L.tiles_x * L.tiles_y;
    long long epb = 16LL * L.NE / (tiles > 0 ? tiles : 1);
    epb = epb < 1 ? 1 : epb > BIN_EDGES ? BIN_EDGES : epb;
    hipLaunchKernelGGL(k_bin, dim3((unsigned)((L.NE + epb - 1) / epb)), dim3(BIN_THREADS), 0, s, L, (int)epb);
}

// LDS prefix entry (12 bytes), per row exclusive prefix over the tile's 128 columns, 16-bit fields packed in pairs:
//   x = sum r | sum g << 16,   y = sum b | n_odd << 16,   z = sum r^2+g^2+b^2
// (128 pixels: sum of a channel <= 32640 < 2^16, n_odd <= 128, q < 2^25: nothing carries between fields, so the
// prefix build adds and scans whole words).
struct pix3 { uint32_t x, y, z; };

__device__ __forceinline__ pix3 pixel_moments(uint32_t rgba) {
    const uint32_t m = rgba & 0x00ffffffu;
    const uint32_t s = __builtin_amdgcn_udot4(rgba, 0x00010101u, 0u, false);  // r + g + b
    pix3 o;
    o.x = __builtin_amdgcn_perm(0u, rgba, 0x0c010c00u);          // r | g << 16
    o.y = __builtin_amdgcn_perm(s & 1u, rgba, 0x0c040c02u);      // b | odd << 16
    o.z = __builtin_amdgcn_udot4(rgba, m, 0u, false);            // r^2 + g^2 + b^2
    return o;
}
__device__ __forceinline__ pix3 operator+(pix3 a, pix3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }

// ------------------------------------------------------------------------------------------------
// k_accumulate: one workgroup per tile.  Lane = (tile row, 8-pixel segment) for the prefix build, then
// lane = (edge line of the tile's work list, 1/split of the tile's rows) for the walk.
// ------------------------------------------------------------------------------------------------
#define WALK_ROWS 4  // rows per unrolled trip of the line walk

#ifdef TPOSE_DEBUG
#define TP_STAMP(k) do { if (tid == 0 && blockIdx.x < 4096) L.dbg[blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)
#else
#define TP_STAMP(k) do { } while (0)
#endif

__global__ __launch_bounds__(ACC_THREADS, 6) void k_accumulate(tp_launch L) {  // 6 workgroups per CU (LDS: 6 x 26.5 KB)
    extern __shared__ __attribute__((aligned(16))) uint32_t P[];  // [TH][ROW_WORDS], then int64 T2s[TH+1][5]
    int64_t* T2s = reinterpret_cast<int64_t*>(P + TH * ROW_WORDS);

    const int tid = threadIdx.x;
    // XCD-aware tile order: workgroup b runs on XCD b % 8; give every XCD one contiguous band of tile rows so
    // that the lines, list entries and raster rows a tile shares with its neighbours stay in that XCD's L2
    const int ntiles = L.tiles_x * L.tiles_y;
    const int chunk = (ntiles + 7) >> 3;
    const int tile = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
    if (blockIdx.x == 0 && tid == 0) L.state->rebin_req = 0;  // consumed by the k_bin that ran before us
    if ((int)(blockIdx.x >> 3) >= chunk || tile >= ntiles) return;
    const int tx = tile % L.tiles_x, ty = tile / L.tiles_x;
    TP_STAMP(0);

    // the tile's pixels: lane = (row, segment of 8 pixels), 32 bytes per lane
    const int prow = tid >> 4, seg = tid & 15;
    uint4 px[2];
    {
        const uint4* src = reinterpret_cast<const uint4*>(L.img + (size_t)(ty * TH + prow) * L.pitch + (size_t)(tx * TW + seg * 8) * 4);
        px[0] = src[0]; px[1] = src[1];
    }
    int nlist = L.tilecount[tile];
    if (nlist > L.list_cap) nlist = L.list_cap;
    const int2* list = L.tilelist + (size_t)tile * L.list_cap;
    // work unit = (edge, line, 1/split of the tile's rows): `split` adjacent lanes share a line while all parts fit
    // the workgroup (the set-up of a part is a handful of instructions)
    const int nlines = nlist * TP_NLINES;
    const int lsplit = nlines * 4 <= ACC_THREADS ? 2 : nlines * 2 <= ACC_THREADS ? 1 : 0;  // log2(split)
    const int split = 1 << lsplit;
    const int nitems = nlines << lsplit;
    // this lane's first work item is requested now: nothing after the barrier waits on global memory twice
    int item = tid;
    int2 ent = make_int2(0, 0);
    longlong2 lxs = make_longlong2(0, 0);
    int2 lrows = make_int2(1, 0);
    if (item < nitems) {
        const int line = item >> lsplit, en = line / TP_NLINES, ver = line - en * TP_NLINES;
        ent = list[en];
        lxs = L.line_xs[(size_t)ent.x * TP_NLINES + ver];
        lrows = L.line_rows[(size_t)ent.x * TP_NLINES + ver];
    }
    if (nlist == 0) return;  // nothing crosses this tile (uniform)
    // static-table rows for this tile's row boundaries -> LDS
    if (tid < T2_LDS_WORDS) {
        const int rr = tid / TP_T2_WORDS, ww = tid - rr * TP_T2_WORDS;
        const int rabs = min(ty * TH + rr, L.vw.H);
        T2s[tid] = L.t2[((size_t)rabs * (L.tiles_x + 1) + tx) * TP_T2_WORDS + ww];
    }

    // ---- phase 1: pixels -> row prefix sums in LDS ----------------------------------------------
    {
        pix3 loc[8];  // exclusive prefix inside the lane's segment
        pix3 run = {0, 0, 0};
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t w = k % 4 == 0 ? px[k / 4].x : k % 4 == 1 ? px[k / 4].y : k % 4 == 2 ? px[k / 4].z : px[k / 4].w;
            loc[k] = run;
            run = run + pixel_moments(w);
        }
        pix3 ex;  // everything left of the segment: the 16 lanes of a DPP row are the 16 segments of a tile row
        ex.x = row_scan16(run.x) - run.x;
        ex.y = row_scan16(run.y) - run.y;
        ex.z = row_scan16(run.z) - run.z;
        uint32_t* row = P + prow * ROW_WORDS + seg * 25;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const pix3 e = ex + loc[k];
            row[3 * k] = e.x; row[3 * k + 1] = e.y; row[3 * k + 2] = e.z;
        }
        if (seg == 15) {  // entry 128: the whole row
            const pix3 e = ex + run;
            row[25] = e.x; row[26] = e.y; row[27] = e.z;
        }
    }
    TP_STAMP(1);
    __syncthreads();
    TP_STAMP(2);

    // ---- phase 2: the lines ------------------------------------------------------------------------
    const int row0 = ty * TH;
    const int col0 = tx * TW;
    const int W = L.vw.W;
    // columns of this tile column: [col0, col0 + TW), the last one also takes the clamp value W
    const uint32_t lim = tx == L.tiles_x - 1 ? (uint32_t)(W - col0 + 1) : (uint32_t)TW;
    const int pr = TH >> lsplit;  // rows per part

    for (; item < nitems; item += ACC_THREADS) {
        const int part = item & (split - 1), line = item >> lsplit;
        const int en = line / TP_NLINES, ver = line - en * TP_NLINES;
        if (item != tid) {  // rare: more items than lanes
            ent = list[en];
            lxs = L.line_xs[(size_t)ent.x * TP_NLINES + ver];
            lrows = L.line_rows[(size_t)ent.x * TP_NLINES + ver];
        }
        const int j0 = part * pr;                       // first tile row of this part
        const int koff = lrows.x - row0 - j0;           // part-relative index of the line's first row
        const uint32_t nvalid = (uint32_t)max(lrows.y - lrows.x + 1, 0);
        tp_line ln; ln.x = lxs.x; ln.s = lxs.y; ln.ra = lrows.x; ln.rb = lrows.y;
        tp_walker wk = tp_line_at(ln, row0 + j0);       // exact 32.32 walker for this tile's rows
        uint32_t ar = 0, ag = 0, ab = 0, ao = 0, aq = 0;  // <= 16 rows: channel sums < 2^20, q < 2^29
        uint32_t sx = 0, inmask = 0;
        const uint32_t* Pp = P + j0 * ROW_WORDS;
        for (int c0 = 0; c0 < pr; c0 += WALK_ROWS, Pp += WALK_ROWS * ROW_WORDS) {
            // rows outside the line's rows, or whose crossing column falls into another tile column, read the
            // all-zero entry 0 of the row and are not counted; a trip no lane of the wave needs is skipped
            if (!__any((int)nvalid + koff - c0 > 0 && koff - c0 < WALK_ROWS)) { wk.x += WALK_ROWS * wk.s; continue; }
            pix3 entv[WALK_ROWS];
#pragma unroll
            for (int k = 0; k < WALK_ROWS; k++) {
                const int32_t x = min(max((int32_t)(wk.x >> 32), 0), W);
                wk.x += wk.s;
                const uint32_t xl = (uint32_t)(x - col0);
                const bool in = xl < lim && (uint32_t)(c0 + k - koff) < nvalid;
                const uint32_t xs = in ? xl : 0u;
                const uint32_t* ep = Pp + k * ROW_WORDS + xs * 3 + (xs >> 3);
                entv[k].x = ep[0]; entv[k].y = ep[1]; entv[k].z = ep[2];
                sx += xs;
                inmask |= in ? (1u << (c0 + k)) : 0u;
            }
#pragma unroll
            for (int k = 0; k < WALK_ROWS; k++) {
                ar += entv[k].x & 0xffffu; ag += entv[k].x >> 16;
                ab += entv[k].y & 0xffffu; ao += entv[k].y >> 16;
                aq += entv[k].z;
            }
        }
        uint32_t nin = (uint32_t)__builtin_popcount(inmask);
        int32_t first = inmask ? row0 + j0 + (int)__builtin_ctz(inmask) : INT32_MAX;
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
        if (ent.y < L.visit_cap) {  // 32-byte record: two 16-byte stores
            uint4* out = reinterpret_cast<uint4*>(L.visits + ((size_t)ent.y * TP_NLINES + ver) * TP_REC_DWORDS);
            const uint64_t q = (uint64_t)((int64_t)aq + st[4]);
            out[0] = make_uint4(sx + nin * (uint32_t)col0, (uint32_t)((int64_t)ao + st[0]), (uint32_t)((int64_t)ar + st[1]),
                                (uint32_t)((int64_t)ag + st[2]));
            out[1] = make_uint4((uint32_t)((int64_t)ab + st[3]), 0u, (uint32_t)q, (uint32_t)(q >> 32));
        }
    }
    TP_STAMP(3);
}

static int accumulate_grid(const tp_launch& L) {
    const int ntiles = L.tiles_x * L.tiles_y;
    return ((ntiles + 7) >> 3) * 8;
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
// per-variant moments = signed sum of three line sums; a line sum = sum of the line's per-tile records.
// G adjacent lanes share a variant (coarse meshes: hundreds of records per line): lane `part` takes records
// part, part + G, ... and the partial moments are combined with shuffles -- every lane returns the full moments.
// ------------------------------------------------------------------------------------------------
template <int G>
__device__ __forceinline__ tp_moments variant_moments(const tp_launch& L, int t, int i, int part) {
    const int4 tri = L.tris[t];
    const int vid[3] = {tri.x, tri.y, tri.z};
    const int ms = i > 0 ? (i - 1) >> 2 : 3, mm = i > 0 ? ((i - 1) & 3) + 1 : 0;
    int32_t X[3], Y[3], c[3];
    int2 ev[3];
    const uint4* rec[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {  // the records do not depend on the coefficients: request everything first
        const int he = L.he_edge[3 * t + k];
        ev[k] = L.edge_visit[he >> 1];  // first record slot, number of slots
        rec[k] = reinterpret_cast<const uint4*>(L.visits) + ((size_t)ev[k].x * TP_NLINES + tp_edge_version(i, k, he & 1)) * 2;
    }
#pragma unroll
    for (int s = 0; s < 3; s++) {
        const int2 q = L.vpos[(size_t)vid[s] * 5 + (s == ms ? mm : 0)];
        X[s] = q.x; Y[s] = q.y;
    }
    tp_variant_coeffs(X, Y, c);
    int64_t m[TP_W_WORDS] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        uint64_t a[TP_W_WORDS] = {0, 0, 0, 0, 0, 0};
        const uint4* r = rec[k];
        const int n = ev[k].y;
        int j = part;
        for (; j + 3 * G < n; j += 4 * G) {  // four records (eight 16-byte loads) in flight
            uint4 lo[4], hi[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { lo[u] = r[(size_t)(j + u * G) * (TP_NLINES * 2)]; hi[u] = r[(size_t)(j + u * G) * (TP_NLINES * 2) + 1]; }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                a[0] += lo[u].x; a[1] += lo[u].y; a[2] += lo[u].z; a[3] += lo[u].w; a[4] += hi[u].x;
                a[5] += (uint64_t)hi[u].z | ((uint64_t)hi[u].w << 32);
            }
        }
        for (; j < n; j += G) {
            const uint4 lo = r[(size_t)j * (TP_NLINES * 2)], hi = r[(size_t)j * (TP_NLINES * 2) + 1];
            a[0] += lo.x; a[1] += lo.y; a[2] += lo.z; a[3] += lo.w; a[4] += hi.x;
            a[5] += (uint64_t)hi.z | ((uint64_t)hi.w << 32);
        }
#pragma unroll
        for (int q = 0; q < TP_W_WORDS; q++) m[q] += (int64_t)c[k] * (int64_t)a[q];
    }
#pragma unroll
    for (int o = 1; o < G; o <<= 1)  // the G lanes are adjacent and always active together
#pragma unroll
        for (int q = 0; q < TP_W_WORDS; q++) {
            const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)m[q], o), hi = (uint32_t)__shfl_xor((int)(uint32_t)((uint64_t)m[q] >> 32), o);
            m[q] += (int64_t)(((uint64_t)hi << 32) | lo);
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

// how many lanes share a variant: records per line grow with tiles per edge
static int lanes_per_variant(const tp_launch& L) {
    const long long tiles = (long long)L.tiles_x * L.tiles_y;
    return tiles < 2LL * L.NE ? 1 : tiles < 16LL * L.NE ? 4 : 16;
}

// k_finalize (tp_energy): G lanes per (triangle, variant); id = i*NT + t in the outputs
template <int G>
__global__ __launch_bounds__(256) void k_finalize(tp_launch L, int flavour, int write_moments) {
    const int gid = (blockIdx.x * blockDim.x + threadIdx.x) / G, part = threadIdx.x % G;
    if (gid >= L.NT * TP_NVARIANTS) return;  // whole groups leave together
    const int t = gid / TP_NVARIANTS, i = gid - t * TP_NVARIANTS;
    const tp_moments m = variant_moments<G>(L, t, i, part);
    if (part == 0) emit_variant(L, flavour, t, i, m, write_moments != 0);
}
void tp_launch_finalize(const tp_launch& L, int flavour, bool write_moments, hipStream_t s) {
    const int G = lanes_per_variant(L);
    const long long n = (long long)L.NT * TP_NVARIANTS * G;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (G == 1) hipLaunchKernelGGL(k_finalize<1>, grid, block, 0, s, L, flavour, write_moments ? 1 : 0);
    else if (G == 4) hipLaunchKernelGGL(k_finalize<4>, grid, block, 0, s, L, flavour, write_moments ? 1 : 0);
    else hipLaunchKernelGGL(k_finalize<16>, grid, block, 0, s, L, flavour, write_moments ? 1 : 0);
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
// k_update: k_finalize + k_shift in ONE launch (used by tp_iterate).  G lanes per variant; the
// four displacements of a vertex slot sit in adjacent lane groups, so the central differences are two
// shuffles.  The quad leader adds them to its vertex with one returning 64-bit atomic per component
// -- (difference << 32) + 1 -- so the thread that completes the vertex's arrival count already
// holds the whole (wrapping int32) gradient component and takes the shift.cs step for it.  x and y
// never interact in shift.cs, so they settle independently; integer sums commute, so the result
// does not depend on arrival order.  Without a margin every launch re-arms the work lists for the next
// k_bin; with one, the last block to finish knows whether any vertex left its margin.
// ------------------------------------------------------------------------------------------------
#define UPD_THREADS 64  // small workgroups: 13 NT threads are only ~600 waves, spread them over all CUs
template <int G>
__global__ __launch_bounds__(UPD_THREADS) void k_update(tp_launch L, int flavour, float rate) {
    __shared__ int s_last;
    const int tidg = blockIdx.x * blockDim.x + threadIdx.x;
    const int gid = tidg / G, part = threadIdx.x % G;
    // a work list overflowed in this or an earlier iteration: the line sums are incomplete.  Do not
    // step -- the host grows the lists and replays from the last good iteration (check_flags)
    if (L.state->flags) return;
    if (tidg == 0) L.state->iters_done++;
    if (L.margin_px < 2) {
        // work lists are rebuilt every iteration: k_accumulate has consumed them, re-arm them here
        for (int k = tidg; k < L.tiles_x * L.tiles_y; k += gridDim.x * blockDim.x) L.tilecount[k] = 0;
        if (tidg == 0) { L.state->visit_total = 0; L.state->rebin_req = 1; L.state->rebin_count++; }
    }

    // variants [0, 12 NT): quads (t, s, k); variants [12 NT, 13 NT): the base variants
    const int NT = L.NT;
    const bool live = gid < 13 * NT;
    int t = 0, i = 0;
    if (gid < 12 * NT) { t = gid / 12; i = gid - 12 * t + 1; }
    else if (live) { t = gid - 12 * NT; i = 0; }
    const bool leader = live && part == 0 && i > 0 && ((i - 1) & 3) == 0;
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
    if (live) {
        const tp_moments m = variant_moments<G>(L, t, i, part);
        if (part == 0) e = emit_variant(L, flavour, t, i, m, false);
    }
    // central differences inside the quad: groups 4q+0/1 hold E(+dx)/E(-dx), 4q+2/3 E(+dy)/E(-dy)
    const uint32_t e1 = (uint32_t)__shfl_xor(e, G);
    const uint32_t gx = (uint32_t)e - e1;                         // valid on even groups of the quad
    const uint32_t gy = (uint32_t)__shfl_down((int)gx, 2 * G);    // group 4q+0 fetches group 4q+2's value
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
    if (tidg >= 4 && tidg < L.NP && L.vtx_off[tidg + 1] == L.vtx_off[tidg]) {
        float2 q = L.points[tidg];
        const float R = L.vw.ratio;
        q.x = q.x <= -R ? -R : (q.x >= R ? R : q.x);
        q.y = q.y <= -1.0f ? -1.0f : (q.y >= 1.0f ? 1.0f : q.y);
        L.points[tidg] = q;
    }
    if (L.margin_px < 2) return;
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
    const int G = lanes_per_variant(L);
    long long n = 13LL * L.NT * G;  // G lanes per variant, and at least one thread per vertex
    if (n < L.NP) n = L.NP;
    const dim3 grid((unsigned)((n + UPD_THREADS - 1) / UPD_THREADS)), block(UPD_THREADS);
    if (G == 1) hipLaunchKernelGGL(k_update<1>, grid, block, 0, s, L, flavour, rate);
    else if (G == 4) hipLaunchKernelGGL(k_update<4>, grid, block, 0, s, L, flavour, rate);
    else hipLaunchKernelGGL(k_update<16>, grid, block, 0, s, L, flavour, rate);
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
// H rows, out[0..1] = (ra, rb) and out[2 + k] = the crossing column of row ra + k (k < rows) derived tile by tile
// exactly as k_accumulate does (tp_line_at at the tile's first row, then one step per row)
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
    for (int row0 = ln.ra / TH * TH; row0 <= last; row0 += TH) {
        tp_walker w = tp_line_at(ln, row0);
        for (int j = 0; j < TH; j++) {
            const int r = row0 + j;
            if (r >= ln.ra && r <= last) o[2 + r - ln.ra] = tp_walker_value(w);
            w.x += w.s;
        }
    }
}
void tp_launch_selftest_line(const int4* ends, const int* H, int n, int rows, int32_t* out, hipStream_t s) {
    hipLaunchKernelGGL(k_selftest_line, dim3((n + 255) / 256), dim3(256), 0, s, ends, H, n, rows, out);
}
