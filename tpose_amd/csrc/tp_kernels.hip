// tp_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the t-pose hot path.  wave64 only.
//
// One grad-iter of the reference = two instanced draws of 13*NT triangles (mode 0: 4 same-address
// int atomics per fragment, mode 1: one) + gradient.cs + shift.cs
// (software/triangulate/main.cpp:121-155).  Here the work is organised around EDGE LINES
// (tp_raster.h, "edge-centric form"): a variant's pixel moments are the signed sum of three line
// sums W(e) = sum over the line's rows of the row-prefix sum at the line's crossing column, and the
// 13 variants of all triangles share 9 lines per undirected edge.  Three kernels per grad-iter:
//
//   k_bin         per edge (16 lanes, nine of them one line each): vertex stage of both endpoints for the five
//                 moves, the nine lines set up ONCE as whole-line 24.40 walkers (line table) with the static part
//                 of their sums (everything left of the tile column, from the per-image table), then the tiles the
//                 band of lines can touch, tile row by tile row, with an exact per-line liveness test -> per-tile
//                 work lists of the LIVE (line, record) pairs
//   k_accumulate  THE hot kernel: one 256-thread workgroup per 128x16-pixel tile (six resident per CU, the
//                 dispatcher balances the rest); the tile's RGBA8 pixels are read once (32 B per lane), turned
//                 into per-row prefix sums of the pixel moments in LDS (12-byte packed entries; DPP row scan), and
//                 every live (line, tile) pair is walked by one to four lanes: per row one exact crossing column from the line's walker and ONE LDS entry.  No
//                 atomics, no per-fragment work.
//   k_update      per variant: signed sum of its three lines (static part + tile records) -> exact moments ->
//                 `colnum`, `colacc`, `tenergy` (reference layout); central differences; per-vertex arrival
//                 atomics; shift.cs step; re-arms the work lists.  (k_finalize + k_shift: the same as two
//                 launches, piecewise API.)
#include "tp_kernels.h"
#include <hip/hip_ext.h>

#define TW TP_TILE_W
#define TH TP_TILE_H
#define ACC_THREADS (16 * TH)  // 16 lanes (8 pixels each) per tile row

static_assert(TW == 128, "prefix build: 16 lanes x 8 pixels per row, 16-bit channel sums");
static_assert(TH % 4 == 0 && TH <= TP_WALK_MAXROWS && ACC_THREADS % 64 == 0, "tile height");

// debug flavour of the library only (tools/kernel_timeline.py): thread 0 of the first 4096 workgroups of a kernel
// stamps the 100 MHz wall clock at its phase boundaries; region 0 k_bin, 1 k_accumulate, 2 k_update
#ifdef TPOSE_DEBUG
#define TP_STAMP(region, k) do { if (threadIdx.x == 0 && blockIdx.x < 4096) L.dbg[((region) * 4096 + blockIdx.x) * 8 + (k)] = wall_clock64(); } while (0)
#else
#define TP_STAMP(region, k) do { } while (0)
#endif

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
// static per-image data (built once per tp_set_image)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void px_moments5(uint32_t rgba, uint32_t m[5]) {
    const uint32_t r = rgba & 0xffu, g = (rgba >> 8) & 0xffu, b = (rgba >> 16) & 0xffu;
    m[0] += (r + g + b) & 1u; m[1] += r; m[2] += g; m[3] += b; m[4] += r * r + g * g + b * b;
}

// The sweep never looks at alpha (neither does the reference: triangle.fs uses .rgb only), so the context's own
// padded copy of the raster keeps (r + g + b) & 1 there: the parity every pixel contributes to n_odd.
__global__ void k_static_alpha(uint8_t* img, int pitch, int Wp, int Hp) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= Wp * Hp) return;
    uint32_t* p = reinterpret_cast<uint32_t*>(img + (size_t)(gid / Wp) * pitch) + gid % Wp;
    const uint32_t w = *p & 0x00ffffffu;
    *p = w | ((((w & 0xffu) + ((w >> 8) & 0xffu) + (w >> 16)) & 1u) << 24);
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
// (one thread per (tile column, word); the rows are a serial chain of H steps)
__global__ void k_static_cols(const uint32_t* seg, int H, int tiles_x, int64_t* t2) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= tiles_x * 5) return;
    const int tc = gid / 5, k = gid - tc * 5;
    int64_t acc = 0;
    for (int r = 0; r <= H; r++) {
        t2[((size_t)r * (tiles_x + 1) + tc + 1) * TP_T2_WORDS + k] = acc;
        if (r < H) acc += seg[((size_t)r * tiles_x + tc) * 5 + k];
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
void tp_launch_static_table(uint8_t* img, int pitch, int W, int H, int Hp, int tiles_x, uint32_t* seg, int64_t* t2, hipStream_t s) {
    const int Wp = tiles_x * TW;
    hipLaunchKernelGGL(k_static_alpha, dim3((unsigned)(((size_t)Wp * Hp + 255) / 256)), dim3(256), 0, s, img, pitch, Wp, Hp);
    hipLaunchKernelGGL(k_static_seg, dim3((H * tiles_x + 255) / 256), dim3(256), 0, s, img, pitch, W, H, tiles_x, seg);
    hipLaunchKernelGGL(k_static_cols, dim3((tiles_x * 5 + 63) / 64), dim3(64), 0, s, seg, H, tiles_x, t2);
    hipLaunchKernelGGL(k_static_rows, dim3((H + 1 + 255) / 256), dim3(256), 0, s, H, tiles_x, t2);
}

// ------------------------------------------------------------------------------------------------
// k_bin: sixteen edges per workgroup, a row of 16 lanes each.
//   phase 0  lane q < 9 of a row owns line q of the edge: vertex stage, the whole-line walker (line table), and per
//            tile row of the line the range of tile columns it crosses there (exact; LDS)
//   pass A   the tiles the band of the nine lines can touch, tile row by tile row (lane q takes tile rows
//            ty0 + q, + 16, ...): counted, scanned -> consecutive visit ids per edge, then entered in an LDS table
//   pass B   ONE LANE PER VISIT: which of the nine lines are live there (nine range look-ups), one returning atomic
//            reserves list slots for the live ones -- every visit's atomic is in flight at once -- then the entries
//   static   the static part of every line's sums (everything left of the tile column, per run of rows inside one
//            tile column a difference of the cumulative per-image table): its loads fly beside the atomics
// ------------------------------------------------------------------------------------------------
#define BIN_THREADS 256
#define BIN_EDGES 16
#define BIN_VISITS 1024  // visits per pass of the LDS table (more: further passes)
#define BIN_HASH_LOG 11
#define BIN_HASH (1 << BIN_HASH_LOG)
#define BIN_TROWS 8      // tile rows per line with precomputed column ranges (longer lines: tested per visit)

__global__ __launch_bounds__(BIN_THREADS) void k_bin(tp_launch L, int epb) {  // epb: edges per workgroup, 16 or (coarse meshes) 1
    __shared__ int s_cnt[BIN_EDGES];     // visits per edge
    __shared__ int s_first[BIN_EDGES];   // exclusive scan
    __shared__ int s_total;
    __shared__ uint32_t s_base;
    __shared__ int64_t s_lx[BIN_EDGES][TP_NLINES][2];
    __shared__ int s_lr[BIN_EDGES][TP_NLINES][2];
    __shared__ uint16_t s_rng[BIN_EDGES][TP_NLINES][BIN_TROWS];  // first | last << 8 tile column of the line in tile row ty_line0 + k
    __shared__ int s_vis[BIN_VISITS];    // (edge of the block << 27) | tile
    __shared__ int h_key[BIN_HASH], h_cnt[BIN_HASH], h_base[BIN_HASH];  // the block's visits grouped by tile
    __shared__ unsigned long long s_st[TP_NLINES][TP_T2_WORDS];         // coarse meshes: static sums gathered from 16 chunks per line
    const int tid = threadIdx.x;
    const uint32_t rebin_word = L.state->rebin_req;  // consumed late: the loads below do not wait for it
    if (blockIdx.x == 0 && tid == 0) L.state->sweep++;  // records of this sweep carry its number (single writer)
    const int j = tid >> 4, q = tid & 15;
    const int e = j < epb ? blockIdx.x * epb + j : L.NE;  // coarse meshes: one edge per workgroup, the other rows help
    tp_band band = {0, 0, 0, 0, 0, 0};
    tp_line ln; ln.x = 0; ln.s = 0; ln.ra = 1; ln.rb = 0;
    const bool owner = e < L.NE && q < TP_NLINES;
    TP_STAMP(0, 0);
    {   // phase 0
        int dX = 0, dY = 0;
        if (e < L.NE) {
            const int2 uv = L.edge_uv[e];  // (only for the vpos stores below: nothing waits for it)
            const int u = uv.x & 0x3fffffff, v = uv.y & 0x3fffffff;
            // the endpoints' positions, filed per edge by whoever moved the vertex (k_update, k_shift, upload): one
            // coalesced load instead of ids -> positions
            const float4 ep = reinterpret_cast<const float4*>(L.epos)[e];
            const float2 pu = make_float2(ep.x, ep.y), pv = make_float2(ep.z, ep.w);
            tp_vertex_stage(pu.x, pu.y, 0, 0, L.vw, band.Xa, band.Ya);
            tp_vertex_stage(pv.x, pv.y, 0, 0, L.vw, band.Xb, band.Yb);
            if (q < TP_NLINES) {  // line q: endpoint u displaced by move mu, endpoint v by move mv
                const int mu = (q >= 1 && q <= 4) ? q : 0, mv = q >= 5 ? q - 4 : 0;
                int32_t Xa, Ya, Xb, Yb;
                tp_vertex_stage(pu.x, pu.y, mu, 0, L.vw, Xa, Ya);
                tp_vertex_stage(pv.x, pv.y, mv, 0, L.vw, Xb, Yb);
                tp_setup_line(Xa, Ya, Xb, Yb, L.vw.H, ln);
                dX = max(abs(Xa - band.Xa), abs(Xb - band.Xb));
                dY = max(abs(Ya - band.Ya), abs(Yb - band.Yb));
                // one edge per vertex publishes its snapped positions (k_update reads them)
                if (mv == 0 && ((uv.x >> 30) & 1)) L.vpos[(size_t)u * 5 + mu] = make_int2(Xa, Ya);
                if (mu == 0 && ((uv.y >> 30) & 1)) L.vpos[(size_t)v * 5 + mv] = make_int2(Xb, Yb);
                s_lx[j][q][0] = ln.x; s_lx[j][q][1] = ln.s; s_lr[j][q][0] = ln.ra; s_lr[j][q][1] = ln.rb;
            }
        }
        band.dX = row_max16(dX);
        band.dY = row_max16(dY);
    }
    TP_STAMP(0, 1);
    // Static part of the line sums, first half: the runs of rows inside one tile column (almost always one or two)
    // and the loads of the cumulative per-image table for them.  Called once per thread, right after the thread's
    // list atomic was issued: the arithmetic and the loads run beside it.
    int64_t st[TP_T2_WORDS] = {0, 0, 0, 0, 0};
    int64_t sa[2][TP_T2_WORDS], sb[2][TP_T2_WORDS];
    int nruns = 0;
    bool static_done = false;
    auto static_first_half = [&]() {
    if (owner) {
        const int64_t* t2 = L.t2;
        const int tx1 = L.tiles_x + 1;
        int rtc[2] = {0, 0}, rra[2] = {0, 0}, rrb[2] = {0, 0};
        tp_line_column_runs(ln, L.vw.W, TW, L.tiles_x, [&](int32_t tc, int32_t ra, int32_t rb) {
            if (tc == 0) return;  // nothing is left of the first tile column
            if (nruns == 0) { rtc[0] = tc; rra[0] = ra; rrb[0] = rb; }
            else if (nruns == 1) { rtc[1] = tc; rra[1] = ra; rrb[1] = rb; }
            else {  // a third run and beyond (long or nearly horizontal lines): summed on the spot
                const int64_t* a = t2 + ((size_t)ra * tx1 + tc) * TP_T2_WORDS;
                const int64_t* b = t2 + ((size_t)(rb + 1) * tx1 + tc) * TP_T2_WORDS;
#pragma unroll
                for (int k = 0; k < TP_T2_WORDS; k++) st[k] += b[k] - a[k];
            }
            nruns++;
        });
#pragma unroll
        for (int r = 0; r < 2; r++)
            if (r < nruns) {
                const int64_t* a = t2 + ((size_t)rra[r] * tx1 + rtc[r]) * TP_T2_WORDS;
                const int64_t* b = t2 + ((size_t)(rrb[r] + 1) * tx1 + rtc[r]) * TP_T2_WORDS;
#pragma unroll
                for (int k = 0; k < TP_T2_WORDS; k++) { sa[r][k] = a[k]; sb[r][k] = b[k]; }
            }
    }
    };
    if (rebin_word != 0) {  // (uniform) profiling replays the sweep over unchanged lists (tp_profile_accumulate)
        // the line's tile-column range in each of its first BIN_TROWS tile rows: the crossing column is monotone in the
        // row, so the two end rows of the overlap bound it (exactly the test of tp_line_live)
        if (owner && ln.ra <= ln.rb) {
            const int t0 = ln.ra / TH;
            for (int k = 0; k < BIN_TROWS && (t0 + k) * TH <= ln.rb; k++) {
                const int rlo = max(ln.ra, (t0 + k) * TH), rhi = min(ln.rb, (t0 + k) * TH + TH - 1);
                const int ca = min(tp_line_col(ln, rlo, L.vw.W) / TW, L.tiles_x - 1), cb = min(tp_line_col(ln, rhi, L.vw.W) / TW, L.tiles_x - 1);
                s_rng[j][q][k] = (uint16_t)(min(ca, cb) | (max(ca, cb) << 8));
            }
        }

        // ---- pass A: visits (tiles of the band), lane q takes the tile rows ty0 + q, + 16, ...
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
        __syncthreads();  // (also publishes the lines and ranges of phase 0)
        if (tid < 64) {  // wave 0: scan of the per-edge counts, visit ids for the block
            const int v = tid < BIN_EDGES ? s_cnt[tid] : 0;
            const int incl = (int)row_scan16((uint32_t)v);
            if (tid < BIN_EDGES) s_first[tid] = incl - v;
            if (tid == BIN_EDGES - 1) {
                // visit ids: every block owns a slice of the lower half of the record buffer (no global
                // atomic on the common path); a block with long edges draws from the shared upper half
                const uint32_t half = (uint32_t)L.visit_cap / 2, slice = half / gridDim.x;
                uint32_t base = blockIdx.x * slice;
                if ((uint32_t)incl > slice) {
                    base = half + atomicAdd(&L.state->visit_total, (uint32_t)incl);
                    if (base + (uint32_t)incl > (uint32_t)L.visit_cap) atomicOr(&L.state->flags, TP_FLAG_VISIT_OVERFLOW);
                }
                s_base = base;
                s_total = incl;
            }
        }
        __syncthreads();
        TP_STAMP(0, 2);
        const int total = s_total;
        const uint32_t base = s_base;
        const bool fits = (long long)base + total <= (long long)L.visit_cap;  // overflow is flagged; everyone stays in bounds
        if (q == 15 && e < L.NE) L.edge_visit[e] = make_int2(fits ? (int)base + s_first[j] : 0, fits ? inc : 0);

        for (int v0 = 0; fits && v0 < total; v0 += BIN_VISITS) {
            // the visits [v0, v0 + BIN_VISITS) of the block -> LDS table, in the order they were counted
            int k = s_first[j] + (inc - cnt) - v0;
            for (int ty = ty0 + q; ty <= ty1; ty += 16) {
                int32_t tx0, tx1;
                const int row0 = ty * TH;
                if (!tp_band_cols(band, row0, min(row0 + TH - 1, L.vw.H - 1), L.vw.W, TW, L.tiles_x, tx0, tx1)) continue;
                for (int tx = tx0; tx <= tx1; tx++, k++)
                    if (k >= 0 && k < BIN_VISITS) s_vis[k] = (j << 27) | (ty * L.tiles_x + tx);
            }
            for (int i = tid; i < BIN_HASH; i += BIN_THREADS) { h_key[i] = -1; h_cnt[i] = 0; }
            __syncthreads();
            TP_STAMP(0, 3);
            // ---- pass B: a lane per visit.  The block first groups its visits by tile in an LDS hash table, so that
            // every distinct tile costs ONE returning global atomic per block (visits of neighbouring edges share tiles:
            // the atomics are fewer and far less contended), then every visit writes its entries.
            const int nv = min(total - v0, BIN_VISITS);
            uint32_t vmask[BIN_VISITS / BIN_THREADS];
            int vslot[BIN_VISITS / BIN_THREADS], vrank[BIN_VISITS / BIN_THREADS];
#pragma unroll
            for (int r = 0; r < BIN_VISITS / BIN_THREADS; r++) {
                const int t = tid + r * BIN_THREADS;
                uint32_t mask = 0;
                vslot[r] = 0; vrank[r] = 0;
                if (t < nv) {
                    const int w = s_vis[t], jj = w >> 27, tile = w & 0x7ffffff;
                    const int ty = tile / L.tiles_x, tx = tile - ty * L.tiles_x;
#pragma unroll
                    for (int l = 0; l < TP_NLINES; l++) {
                        const int ra = s_lr[jj][l][0], rb = s_lr[jj][l][1];
                        const int k = ty - ra / TH;  // which of the line's tile rows
                        bool live = ra <= rb && k >= 0 && ty * TH <= rb;
                        if (live) {
                            if (k < BIN_TROWS) {
                                const uint32_t rg = s_rng[jj][l][k];
                                live = tx >= (int)(rg & 0xffu) && tx <= (int)(rg >> 8);
                            } else {  // a long line (coarse mesh): tested here
                                tp_line ll; ll.x = s_lx[jj][l][0]; ll.s = s_lx[jj][l][1]; ll.ra = ra; ll.rb = rb;
                                const int col0 = tx * TW;
                                live = tp_line_live(ll, ty * TH, min(ty * TH + TH - 1, L.vw.H - 1), col0,
                                                    tx == L.tiles_x - 1 ? L.vw.W - col0 + 1 : TW, L.vw.W);
                            }
                        }
                        mask |= live ? 1u << l : 0u;
                    }
                    if (mask) {
                        int hs = (int)(((uint32_t)tile * 2654435761u) >> (32 - BIN_HASH_LOG));
                        while (true) {  // open addressing: at most BIN_VISITS distinct keys for twice as many slots
                            const int old = atomicCAS(&h_key[hs], -1, tile);
                            if (old == -1 || old == tile) break;
                            hs = (hs + 1) & (BIN_HASH - 1);
                        }
                        vslot[r] = hs;
                        vrank[r] = atomicAdd(&h_cnt[hs], (int)__builtin_popcount(mask));
                    }
                }
                vmask[r] = mask;
            }
            __syncthreads();
            for (int i = tid; i < BIN_HASH; i += BIN_THREADS)
                if (h_key[i] >= 0) h_base[i] = atomicAdd(&L.tilecount[(size_t)h_key[i] * TP_COUNT_STRIDE], h_cnt[i]);
            if (!static_done && epb != 1) { static_first_half(); static_done = true; }  // arithmetic and loads beside the atomics in flight
            __syncthreads();
#pragma unroll
            for (int r = 0; r < BIN_VISITS / BIN_THREADS; r++) {
                const int t = tid + r * BIN_THREADS;
                const uint32_t mask = vmask[r];
                if (mask == 0) continue;
                const int w = s_vis[t], jj = w >> 27, tile = w & 0x7ffffff;
                const int visit = (int)base + v0 + t;
                int pos = h_base[vslot[r]] + vrank[r];
                uint4* dst = reinterpret_cast<uint4*>(L.tilelist + (size_t)tile * L.list_cap * 2);
#pragma unroll
                for (int l = 0; l < TP_NLINES; l++)
                    if ((mask >> l) & 1u) {
                        if (pos < L.list_cap) {  // the entry is self-contained: the line's walker, its rows, where its record goes
                            const uint64_t x = (uint64_t)s_lx[jj][l][0], sl = (uint64_t)s_lx[jj][l][1];
                            dst[2 * pos] = make_uint4((uint32_t)x, (uint32_t)(x >> 32), (uint32_t)sl, (uint32_t)(sl >> 32));
                            dst[2 * pos + 1] = make_uint4((uint32_t)s_lr[jj][l][0], (uint32_t)s_lr[jj][l][1], (uint32_t)(visit * TP_NLINES + l), 0u);
                        } else
                            atomicOr(&L.state->flags, TP_FLAG_LIST_OVERFLOW);
                        pos++;
                    }
            }
            if (v0 + BIN_VISITS < total) __syncthreads();  // the table is rewritten by the next pass
        }
    }
    // ---- static part of the line sums, second half
    if (epb == 1) {
        // Coarse meshes: a line crosses many tile columns (a run and a bisection per column).  The workgroup has one
        // edge: 16 lanes per line take a sixteenth of the line's rows each -- a sub-line with the same walker -- and
        // their partial sums meet in LDS.
        if (tid < TP_NLINES * TP_T2_WORDS) s_st[tid / TP_T2_WORDS][tid % TP_T2_WORDS] = 0ull;
        __syncthreads();  // (also: the lines of phase 0 are in LDS)
        const int l = tid >> 4, c = tid & 15;
        if (l < TP_NLINES && blockIdx.x < (unsigned)L.NE) {
            tp_line whole; whole.x = s_lx[0][l][0]; whole.s = s_lx[0][l][1]; whole.ra = s_lr[0][l][0]; whole.rb = s_lr[0][l][1];
            const int rows = whole.rb - whole.ra + 1;
            if (rows > 0) {
                tp_line sub = whole;
                sub.ra = whole.ra + (int)(((long long)rows * c) >> 4);
                sub.rb = whole.ra + (int)(((long long)rows * (c + 1)) >> 4) - 1;
                sub.x = whole.x + (int64_t)(sub.ra - whole.ra) * whole.s;
                int64_t part[TP_T2_WORDS] = {0, 0, 0, 0, 0};
                const int64_t* t2 = L.t2;
                const int tx1 = L.tiles_x + 1;
                tp_line_column_runs(sub, L.vw.W, TW, L.tiles_x, [&](int32_t tc, int32_t ra, int32_t rb) {
                    if (tc == 0) return;
                    const int64_t* a = t2 + ((size_t)ra * tx1 + tc) * TP_T2_WORDS;
                    const int64_t* b = t2 + ((size_t)(rb + 1) * tx1 + tc) * TP_T2_WORDS;
#pragma unroll
                    for (int k = 0; k < TP_T2_WORDS; k++) part[k] += b[k] - a[k];
                });
#pragma unroll
                for (int k = 0; k < TP_T2_WORDS; k++) atomicAdd(&s_st[l][k], (unsigned long long)part[k]);
            }
        }
        __syncthreads();
        if (owner) {
            const size_t li = (size_t)e * TP_NLINES + q;
#pragma unroll
            for (int k = 0; k < TP_T2_WORDS; k++) L.line_static[li * TP_T2_WORDS + k] = (int64_t)s_st[q][k];
        }
        TP_STAMP(0, 4);
        return;
    }
    if (!static_done) static_first_half();
    if (owner) {
#pragma unroll
        for (int r = 0; r < 2; r++)
            if (r < nruns)
#pragma unroll
                for (int k = 0; k < TP_T2_WORDS; k++) st[k] += sb[r][k] - sa[r][k];
        const size_t li = (size_t)e * TP_NLINES + q;
#pragma unroll
        for (int k = 0; k < TP_T2_WORDS; k++) L.line_static[li * TP_T2_WORDS + k] = st[k];
    }
    TP_STAMP(0, 4);
}

void tp_launch_bin(const tp_launch& L, hipStream_t s) {
    // coarse meshes on large rasters (long edges, hundreds of tiles each): one edge per workgroup
    const int epb = tp_coarse_mesh(L) ? 1 : BIN_EDGES;
    hipLaunchKernelGGL(k_bin, dim3((unsigned)((L.NE + epb - 1) / epb)), dim3(BIN_THREADS), 0, s, L, epb);
}

// LDS prefix entry (12 bytes), per row exclusive prefix over the tile's 128 columns, 16-bit fields packed in pairs:
//   x = sum r | sum g << 16,   y = sum b | n_odd << 16,   z = sum (r^2+g^2+b^2) + n_odd
// (128 pixels: sum of a channel <= 32640 < 2^16, n_odd <= 128, z < 2^25: nothing carries between fields, so whole
// words are added).  The alpha byte of the context's raster copy holds the pixel's parity f = (r+g+b) & 1, which
// makes a pixel's three words ONE instruction each: two byte permutes and dot4(w, w) = r^2+g^2+b^2 + f.
struct pix3 { uint32_t x, y, z; };

__device__ __forceinline__ pix3 pixel_moments(uint32_t w) {
    pix3 o;
    o.x = __builtin_amdgcn_perm(0u, w, 0x0c010c00u);       // r | g << 16
    o.y = __builtin_amdgcn_perm(0u, w, 0x0c030c02u);       // b | f << 16
    o.z = __builtin_amdgcn_udot4(w, w, 0u, false);         // r^2 + g^2 + b^2 + f
    return o;
}
__device__ __forceinline__ pix3 operator+(pix3 a, pix3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }

// ------------------------------------------------------------------------------------------------
// k_accumulate: one workgroup per tile.  Lane = (tile row, 8-pixel segment) for the prefix build, then
// lane = (live edge line of the tile's work list, 1/split of the tile's rows) for the walk.
// ------------------------------------------------------------------------------------------------
#define WALK_ROWS 4   // rows per unrolled trip of the line walk
// LDS prefix table: entry x of a row (x = 0..128, exclusive prefix over the tile's columns) lives at word
// 3 x + TP_SEG_PAD (x >> 3): pad words after every eight entries, so that the sixteen lanes of a tile row -- each
// storing eight consecutive entries -- start SEG_WORDS apart and one store instruction spreads over the banks
// (without padding the lanes are 24 words apart and only four bank groups are hit: 4-way conflicts on every store).
#ifndef TP_SEG_PAD
#define TP_SEG_PAD 1
#endif
#define SEG_WORDS (24 + TP_SEG_PAD)
#define ROW_WORDS (16 * SEG_WORDS + 3 + (TP_SEG_PAD == 1 ? 1 : 0))  // pad 1: 404 (20 mod 32); pad 2: 419 (3 mod 32); pad 0: 387
size_t tp_accumulate_lds_bytes() { return (size_t)(TH * ROW_WORDS + 1) * sizeof(uint32_t); }


__global__ __launch_bounds__(ACC_THREADS, 6) void k_accumulate(tp_launch L) {  // 6 workgroups per CU
    extern __shared__ __attribute__((aligned(16))) uint32_t P[];  // [TH][ROW_WORDS]

    const int tid = threadIdx.x;
    const uint32_t sweep = L.state->sweep;  // stamped into the records
    // Every workgroup is resident (six per CU) and takes tiles slot, slot + nslots, ... of its XCD: workgroup b runs on
    // XCD b % 8, and every XCD owns one contiguous band of tile rows, so that the raster rows, list entries and records a
    // tile shares with its neighbours stay in that XCD's L2.  The loads of a workgroup's NEXT tile are issued before it
    // works on the current one: the second round of tiles never waits for memory.
    const int ntiles = L.tiles_x * L.tiles_y;
    const int chunk = (ntiles + 7) >> 3;            // tiles per XCD
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
    const int tile_end = min((xcd + 1) * chunk, ntiles);
    if (blockIdx.x == 0 && tid == 0) L.state->rebin_req = 0;  // consumed by the k_bin that ran before us
    const int prow = tid >> 4, seg = tid & 15;  // the 16 lanes of a DPP row are the 16 segments of a tile row
    const int W = L.vw.W;

    struct fetched { uint4 px[2]; int nlist; };
    auto fetch = [&](int tile, fetched& f) {
        const int tx = tile % L.tiles_x, ty = tile / L.tiles_x;
        const uint4* src = reinterpret_cast<const uint4*>(L.img + (size_t)(ty * TH + prow) * L.pitch + (size_t)(tx * TW + seg * 8) * 4);
        f.px[0] = src[0]; f.px[1] = src[1];
        f.nlist = min(L.tilecount[(size_t)tile * TP_COUNT_STRIDE], L.list_cap);
    };
    int tile = xcd * chunk + slot;
    if (tile >= tile_end) return;
    fetched cur, nxt;
    fetch(tile, cur);
    TP_STAMP(1, 0);

    for (; tile < tile_end; tile += nslots, cur = nxt) {
        const int next = tile + nslots;
        if (next < tile_end) fetch(next, nxt);  // in flight while this tile is processed
        const int tx = tile % L.tiles_x, ty = tile / L.tiles_x;
        const int nlist = cur.nlist;
        const uint4* list = reinterpret_cast<const uint4*>(L.tilelist + (size_t)tile * L.list_cap * 2);
        // work unit = (live line, 1/split of the tile's rows): `split` adjacent lanes share a line while all parts fit
        // the workgroup
        const int lsplit = nlist * 4 <= ACC_THREADS ? 2 : nlist * 2 <= ACC_THREADS ? 1 : 0;  // log2(split)
        const int split = 1 << lsplit;
        const int nitems = nlist << lsplit;
        int item = tid;
        uint4 e0 = make_uint4(0, 0, 0, 0), e1 = make_uint4(1, 0, 0, 0);  // this lane's first work item, requested now
        if (item < nitems) { e0 = list[2 * (item >> lsplit)]; e1 = list[2 * (item >> lsplit) + 1]; }

        // ---- phase 1: pixels -> row prefix sums in LDS (running sums seeded by the static table: no scan) -------
        if (nlist > 0) {
#if defined(TPOSE_ABLATE) && (TPOSE_ABLATE & 1)  // timing experiments only (tools/build_variants.py): no prefix build
            if ((cur.px[0].x ^ cur.px[1].w) == 0x12345u) P[tid] = 1;
#else
            pix3 run;
            uint32_t* row = P + prow * ROW_WORDS + seg * SEG_WORDS;
            {   // everything left of the lane's segment: a DPP scan of the segment totals over the row's 16 lanes
                pix3 tot = {0, 0, 0};
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const uint32_t w = k % 4 == 0 ? cur.px[k / 4].x : k % 4 == 1 ? cur.px[k / 4].y : k % 4 == 2 ? cur.px[k / 4].z : cur.px[k / 4].w;
                    tot = tot + pixel_moments(w);
                }
                run.x = row_scan16(tot.x) - tot.x; run.y = row_scan16(tot.y) - tot.y; run.z = row_scan16(tot.z) - tot.z;
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const uint32_t w = k % 4 == 0 ? cur.px[k / 4].x : k % 4 == 1 ? cur.px[k / 4].y : k % 4 == 2 ? cur.px[k / 4].z : cur.px[k / 4].w;
                row[3 * k] = run.x; row[3 * k + 1] = run.y; row[3 * k + 2] = run.z;
                run = run + pixel_moments(w);
            }
            if (seg == 15) { row[SEG_WORDS] = run.x; row[SEG_WORDS + 1] = run.y; row[SEG_WORDS + 2] = run.z; }  // entry 128: the whole row
#endif
        }
        TP_STAMP(1, 1);
        __syncthreads();
        TP_STAMP(1, 2);

        // ---- phase 2: the lines ------------------------------------------------------------------------
        const int row0 = ty * TH;
        const int col0 = tx * TW;
        // columns of this tile column: [col0, col0 + TW), the last one also takes the clamp value W
        const uint32_t lim = tx == L.tiles_x - 1 ? (uint32_t)(W - col0 + 1) : (uint32_t)TW;
        const int pr = TH >> lsplit;  // rows per part

#if defined(TPOSE_ABLATE) && (TPOSE_ABLATE & 2)  // timing experiments only: no walk
        if ((e0.x ^ e1.x) == 0x1234567 && e1.z < (uint32_t)L.visit_cap * TP_NLINES) L.visits[(size_t)e1.z * TP_REC_DWORDS] = sweep;
        if (false)
#endif
        for (; item < nitems; item += ACC_THREADS) {
            const int part = item & (split - 1);
            if (item != tid) { e0 = list[2 * (item >> lsplit)]; e1 = list[2 * (item >> lsplit) + 1]; }  // rare: more items than lanes
            tp_line ln;
            ln.x = (int64_t)((uint64_t)e0.x | ((uint64_t)e0.y << 32)); ln.s = (int64_t)((uint64_t)e0.z | ((uint64_t)e0.w << 32));
            ln.ra = (int)e1.x; ln.rb = (int)e1.y;
            const uint32_t rec = e1.z;
            const int j0 = part * pr;                       // first tile row of this part
            const int koff = ln.ra - row0 - j0;             // part-relative index of the line's first row
            const uint32_t nvalid = (uint32_t)max(ln.rb - ln.ra + 1, 0);
            tp_walker wk = tp_line_at(ln, row0 + j0);       // exact 32.32 walker for this tile's rows
            // packed sums of two rows never carry (2 x 32640 < 2^16); unpacked into 32-bit sums per pair of rows
            uint32_t ar = 0, ag = 0, ab = 0, ao = 0, aq = 0, sx = 0;
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
                    const uint32_t* ep = Pp + k * ROW_WORDS + (in ? __umul24(xl, 3u) + (TP_SEG_PAD ? (xl >> 3) * TP_SEG_PAD : 0u) : 0u);
                    entv[k].x = ep[0]; entv[k].y = ep[1]; entv[k].z = ep[2];
                    sx += in ? (uint32_t)x : 0u;
                }
#pragma unroll
                for (int k = 0; k < WALK_ROWS; k += 2) {
                    const uint32_t px2 = entv[k].x + entv[k + 1].x, py2 = entv[k].y + entv[k + 1].y;
                    ar += px2 & 0xffffu; ag += px2 >> 16;
                    ab += py2 & 0xffffu; ao += py2 >> 16;
                    aq += entv[k].z + entv[k + 1].z;
                }
            }
            // combine the parts (adjacent lanes; a line's lanes are always active together)
            for (int o = 1; o < split; o <<= 1) {
                sx += (uint32_t)__shfl_xor((int)sx, o);
                ar += (uint32_t)__shfl_xor((int)ar, o); ag += (uint32_t)__shfl_xor((int)ag, o);
                ab += (uint32_t)__shfl_xor((int)ab, o); ao += (uint32_t)__shfl_xor((int)ao, o);
                aq += (uint32_t)__shfl_xor((int)aq, o);
            }
            if (part != 0) continue;
            if (rec < (uint32_t)L.visit_cap * TP_NLINES) {  // 32-byte record
                uint4* out = reinterpret_cast<uint4*>(L.visits + (size_t)rec * TP_REC_DWORDS);
                out[0] = make_uint4(sx, ao, ar, ag); out[1] = make_uint4(ab, aq, sweep, 0u);
            }
        }
        TP_STAMP(1, 3);
        if (next < tile_end) __syncthreads();  // the table is rebuilt for the next tile
    }
}

static int accumulate_grid(const tp_launch& L) {
    // every workgroup resident: 6 per CU on 256 CUs, a multiple of 8 (XCDs); fewer when there are fewer tiles
    const int ntiles = L.tiles_x * L.tiles_y;
    const int per_xcd = (ntiles + 7) >> 3;
    return 8 * (per_xcd < 192 ? per_xcd : 192);
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
// Line sums.  W(line) = its static part (k_bin) + the tile-local records of the tiles it is live in
// (k_accumulate): six values {sum x, n_odd, sum r, sum g, sum b, q}.  Fine meshes sum the handful of records
// where they are needed; coarse meshes on large rasters (hundreds of tiles per edge) run k_linesum first --
// one wave per line -- and everything downstream reads `wline`.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void add_record(uint64_t a[TP_W_WORDS], bool live, const uint4 r0, const uint4 r1) {
    a[0] += live ? r0.x : 0u; a[1] += live ? r0.y : 0u;
    a[2] += live ? r0.z : 0u; a[3] += live ? r0.w : 0u;
    a[4] += live ? r1.x : 0u;
    a[5] += live ? r1.y - r0.y : 0u;  // q: the record holds q + n_odd
}

// visits first + j0, first + j0 + stride, ... of an edge, for line version `ver`: eight visits per trip, all of
// their loads in flight together (an ordinary edge has about six visits: one round trip).  A record counts when
// it carries the number of the current sweep: the line was live in that tile.
__device__ __forceinline__ void sum_records(const tp_launch& L, uint32_t sweep, int first, int n, int ver, int j0, int stride,
                                            uint64_t a[TP_W_WORDS]) {
    for (int j = j0; j < n; j += 8 * stride) {
        uint4 r[8][2];
        bool on[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int jj = j + u * stride;
            on[u] = jj < n;
            const size_t visit = (size_t)first + (on[u] ? jj : j);
            const uint4* rec = reinterpret_cast<const uint4*>(L.visits + (visit * TP_NLINES + ver) * TP_REC_DWORDS);
            r[u][0] = rec[0]; r[u][1] = rec[1];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) add_record(a, on[u] && r[u][1].z == sweep, r[u][0], r[u][1]);
    }
}

__device__ __forceinline__ void line_sum(const tp_launch& L, uint32_t sweep, int e, int ver, int64_t w[TP_W_WORDS]) {
    const size_t line = (size_t)e * TP_NLINES + ver;
    if (L.wline) {  // coarse meshes: summed by k_linesum
#pragma unroll
        for (int q = 0; q < TP_W_WORDS; q++) w[q] = L.wline[line * TP_W_WORDS + q];
        return;
    }
    const int2 ev = L.edge_visit[e];  // first visit, number of visits
    int64_t st[TP_T2_WORDS];
#pragma unroll
    for (int q = 0; q < TP_T2_WORDS; q++) st[q] = L.line_static[line * TP_T2_WORDS + q];
    uint64_t a[TP_W_WORDS] = {0, 0, 0, 0, 0, 0};
    sum_records(L, sweep, ev.x, ev.y, ver, 0, 1, a);
    w[0] = (int64_t)a[0];
#pragma unroll
    for (int q = 1; q < TP_W_WORDS; q++) w[q] = (int64_t)a[q] + st[q - 1];
}

// coarse meshes: one wave per line
__global__ __launch_bounds__(64) void k_linesum(tp_launch L) {
    const int line = blockIdx.x, lane = threadIdx.x;
    const int e = line / TP_NLINES, ver = line - e * TP_NLINES;
    const int2 ev = L.edge_visit[e];
    uint64_t a[TP_W_WORDS] = {0, 0, 0, 0, 0, 0};
    sum_records(L, L.state->sweep, ev.x, ev.y, ver, lane, 64, a);
#pragma unroll
    for (int o = 1; o < 64; o <<= 1)
#pragma unroll
        for (int q = 0; q < TP_W_WORDS; q++) {
            const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)a[q], o), hi = (uint32_t)__shfl_xor((int)(uint32_t)(a[q] >> 32), o);
            a[q] += ((uint64_t)hi << 32) | lo;
        }
    if (lane < TP_W_WORDS) {
        uint64_t v = lane == 0 ? a[0] : lane == 1 ? a[1] : lane == 2 ? a[2] : lane == 3 ? a[3] : lane == 4 ? a[4] : a[5];
        if (lane > 0) v += (uint64_t)L.line_static[(size_t)line * TP_T2_WORDS + lane - 1];
        L.wline[(size_t)line * TP_W_WORDS + lane] = (int64_t)v;
    }
}
bool tp_coarse_mesh(const tp_launch& L) { return (long long)L.tiles_x * L.tiles_y > 4LL * L.NE; }
void tp_launch_linesum(const tp_launch& L, hipStream_t s) {
    hipLaunchKernelGGL(k_linesum, dim3((unsigned)(L.NE * TP_NLINES)), dim3(64), 0, s, L);
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
        line_sum(L, L.state->sweep, he >> 1, tp_edge_version(i, k, he & 1), w[k]);
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
// Every launch re-arms the work lists for the next k_bin.
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
    int* r = vref + (size_t)v * 64;
    int* c = vvar + (size_t)v * 8;
    for (int k = 0; k < 64; k++) r[k] = -1;
    for (int k = 0; k < 8; k++) c[k] = -1;
    if (deg > UPD_FAN) { r[0] = -2; return; }
    int edges[UPD_FAN], flips[UPD_FAN], ne = 0;  // incident edges and whether the vertex is their second endpoint
    for (int a = 0; a < deg; a++) {
        const int h = L.vtx_adj[k0 + a], t = h / 3, s = h - 3 * t;
        const int he_out = L.he_edge[3 * t + s], he_in = L.he_edge[3 * t + (s == 0 ? 2 : s - 1)];
        int slot[2];
        for (int w = 0; w < 2; w++) {
            const int he = w == 0 ? he_out : he_in;
            // leaving edge: the vertex is its origin -- second endpoint when the half-edge is flipped; arriving: the other way
            const int second = w == 0 ? (he & 1) : !(he & 1);
            int b = 0;
            while (b < ne && edges[b] != (he >> 1)) b++;
            if (b == ne) {
                if (ne == UPD_FAN) { for (int k = 0; k < 64; k++) r[k] = -1; r[0] = -2; return; }
                edges[ne] = he >> 1; flips[ne] = second; ne++;
            }
            slot[w] = b;
        }
        c[a] = h | (slot[0] << 20) | (slot[1] << 24);
        r[32 + a] = ((L.he_edge[3 * t + (s == 2 ? 0 : s + 1)] >> 1) << 4) | 0;
    }
    for (int b = 0; b < ne; b++)
        for (int m = 1; m <= 4; m++) r[4 * b + m - 1] = (edges[b] << 4) | (flips[b] ? 4 + m : m);
}
// per upload, after k_vertex_refs: every vertex files its position with its edges
__global__ void k_publish_positions(tp_launch L) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < L.NP) publish_position(L, v, L.points[v], 0, 1);
}
void tp_launch_vertex_refs(const tp_launch& L, int* vref, int* vvar, hipStream_t s) {
    hipLaunchKernelGGL(k_vertex_refs, dim3((unsigned)((L.NP + 63) / 64)), dim3(64), 0, s, L, vref, vvar);
    hipLaunchKernelGGL(k_publish_positions, dim3((unsigned)((L.NP + 63) / 64)), dim3(64), 0, s, L);
}

__global__ __launch_bounds__(UPD_THREADS) void k_update(tp_launch L, int flavour, float rate) {
    __shared__ int64_t S[64][TP_W_WORDS];  // line sums of the wave: [lane] (fast path), [a][l] (generic), [a][k] (base variants)
    const int lane = threadIdx.x;
    const int tidg = blockIdx.x * UPD_THREADS + lane;
    // A work list overflowed in this or an earlier iteration: the line sums are incomplete.  Do not step -- the host
    // grows the lists and replays from the last good iteration (check_flags).  The flag word is requested here and
    // looked at only where something would be written, so that the loads below do not queue behind it.
    const uint32_t flags = L.state->flags;
    const uint32_t sweep = L.state->sweep;
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
        return emit_variant(L, flavour, t, 4 * s + m, mm, false, flags == 0);
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
                line_sum(L, sweep, he >> 1, 0, w);
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
            emit_variant(L, flavour, t, 0, mm, false, flags == 0);
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
                line_sum(L, sweep, ref >> 4, ref & 15, w);
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
                        line_sum(L, sweep, r >> 4, r & 15, w);
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
        if (lane == 0 && flags == 0) {
            if (deg > 0) L.gr[v] = make_int2((int)gx, (int)gy);  // vertices no triangle uses: the gradient is never touched
            if (v >= 4) {  // shift.cs:20 -- the four corners never move
                const float R = L.vw.ratio;
                float tgx = (float)(int)gx, tgy = (float)(int)gy;
                float x = p.x, y = p.y;
                if (x <= -R) { x = -R; tgx = 0.0f; } else if (x >= R) { x = R; tgx = 0.0f; }
                if (y <= -1.0f) { y = -1.0f; tgy = 0.0f; } else if (y >= 1.0f) { y = 1.0f; tgy = 0.0f; }
                if (deg > 0) {  // (unused vertices are only clamped: shift.cs:25-43 runs for every i in [4, NPoints))
                    x = tp_fsub(x, tp_fdiv(tp_fdiv(tp_fmul(rate, tgx), 256.0f), 256.0f));
                    y = tp_fsub(y, tp_fdiv(tp_fdiv(tp_fmul(rate, tgy), 256.0f), 256.0f));
                }
                L.points[v] = make_float2(x, y);
                newp = make_float2(x, y); moved = 1;
            }
        }
        // the new position goes to every edge the vertex ends (k_bin reads endpoints by edge)
        if (__shfl(moved, 0)) publish_position(L, v, make_float2(__shfl(newp.x, 0), __shfl(newp.y, 0)), lane, UPD_THREADS);
    }
    TP_STAMP(2, 3);
    if (flags) return;  // (uniform) nothing was stepped; the host repairs and replays
    if (tidg == 0) L.state->iters_done++;
    // the work lists are rebuilt every iteration: k_accumulate has consumed them, re-arm them here
    for (int k = tidg; k < L.tiles_x * L.tiles_y; k += gridDim.x * UPD_THREADS) L.tilecount[(size_t)k * TP_COUNT_STRIDE] = 0;
    if (tidg == 0) { L.state->visit_total = 0; L.state->rebin_req = 1; L.state->rebin_count++; }
}
void tp_launch_update(const tp_launch& L, int flavour, float rate, hipStream_t s) {
    const int nblocks = L.NP + (L.NT + 20) / 21;  // a wave per vertex, then the base variants (21 triangles per wave)
    hipLaunchKernelGGL(k_update, dim3((unsigned)nblocks), dim3(UPD_THREADS), 0, s, L, flavour, rate);
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
