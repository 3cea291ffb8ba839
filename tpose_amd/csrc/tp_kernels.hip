// tp_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the t-pose hot path.  wave64 only.
//
// One grad-iter of the reference = two instanced draws of 13*NT triangles (mode 0: 4 same-address
// int atomics per fragment, mode 1: one) + gradient.cs + shift.cs
// (software/triangulate/main.cpp:121-155).  Here:
//
//   k_bin         triangles -> per-tile work lists (conservative bbox of the 13 variants)
//   k_accumulate  THE hot kernel: one workgroup per 128x32-pixel tile.  The tile's RGBA8 pixels are
//                 read once, coalesced (16 B per lane), turned into per-row prefix sums of the five
//                 pixel moments in LDS (DPP wave scans), then every (variant, tile) pair is walked
//                 by ONE lane: per row an exact column span [lo,hi) from three 32.32 edge walkers
//                 and two LDS lookups.  No atomics, no per-fragment work; per-pair partial moments
//                 go out as 24-byte records.
//   k_finalize    per variant: sum its partials -> exact moments -> `colnum`, `colacc`, `tenergy`
//                 in the reference layout (replaces triangle.fs mode 0/1).
//   k_shift       per vertex: gather the central differences of its incident triangles
//                 (gradient.cs) and take the clamped step (shift.cs).
#include "tp_kernels.h"
#include <hip/hip_ext.h>

#define TW TP_TILE_W
#define TH TP_TILE_H
#define ROWLEN (TW + 1)  // exclusive prefix has TW+1 entries per row

static_assert(TW == 128, "prefix build assumes 32 lanes x 4 pixels per row");
#define ACC_THREADS 512
#define ACC_ROWS_PER_PASS (ACC_THREADS / 32)
static_assert(TH % ACC_ROWS_PER_PASS == 0 && TH <= TP_WALK_MAXROWS, "tile height");

size_t tp_accumulate_lds_bytes() { return (size_t)TH * ROWLEN * sizeof(uint4); }

// ------------------------------------------------------------------------------------------------
// k_bin
// ------------------------------------------------------------------------------------------------
#define BIN_TRIS 64  // triangles per 256-thread block

// Can any variant of the triangle cover a pixel of the rectangle [c0,c1] x [r0,r1]?  Conservative:
// every variant lies inside base (+) box(dX, dY), so it suffices that for one base edge the whole
// rectangle sits more than dX|a| + dY|b| outside.  Integer arithmetic on the snapped vertices.
__device__ __forceinline__ bool may_touch(const int32_t X[3], const int32_t Y[3], int64_t dX, int64_t dY,
                                          int c0, int c1, int r0, int r1) {
    const int64_t area2 = (int64_t)(X[1] - X[0]) * (Y[2] - Y[0]) - (int64_t)(Y[1] - Y[0]) * (X[2] - X[0]);
    if (area2 == 0) return true;
    const int64_t sg = area2 > 0 ? 1 : -1;
    bool touch = true;
#pragma unroll
    for (int e = 0; e < 3; e++) {
        const int j = e == 2 ? 0 : e + 1;
        const int64_t a = -(int64_t)(Y[j] - Y[e]) * sg, b = (int64_t)(X[j] - X[e]) * sg;
        const int64_t xs = 256LL * (a > 0 ? c1 : c0) + 128, ys = 256LL * (b > 0 ? r1 : r0) + 128;
        const int64_t emax = a * (xs - X[e]) + b * (ys - Y[e]);
        const int64_t slack = dX * (a < 0 ? -a : a) + dY * (b < 0 ? -b : b);
        touch = touch && (emax >= -slack);
    }
    return touch;
}

__global__ __launch_bounds__(256) void k_bin(tp_launch L) {
    __shared__ int s_excl[BIN_TRIS + 1];             // exclusive scan of rectangle sizes
    __shared__ int s_rect[BIN_TRIS][4];              // tx0, ty0, ntx, #tiles
    __shared__ int s_X[BIN_TRIS][3], s_Y[BIN_TRIS][3];
    __shared__ int s_infl[BIN_TRIS][2];              // dX, dY: how far a displaced vertex strays (1/256 px)
    __shared__ unsigned long long s_keep[BIN_TRIS];  // tiles of the rectangle some variant can reach
    __shared__ uint32_t s_base;
    const bool rebin = L.state->rebin_req != 0;      // lists still valid otherwise (tp_set_margin)
    const int tid = threadIdx.x;
    if (rebin)
        for (int v = blockIdx.x * 256 + tid; v < L.NP; v += gridDim.x * 256) L.points_binned[v] = L.points[v];

    // ---- vertex stage of all 13 variants, once per triangle per iteration: 4 lanes per triangle,
    //      lane q < 3 transforms vertex slot q (unmoved + its four displacements)
    const int j = tid >> 2, q = tid & 3;
    const int t = blockIdx.x * BIN_TRIS + j;
    int32_t xmin = INT32_MAX, xmax = INT32_MIN, ymin = INT32_MAX, ymax = INT32_MIN, dX = 0, dY = 0;
    if (t < L.NT && q < 3) {
        const int4 tri = L.tris[t];
        const float2 p = L.points[q == 0 ? tri.x : q == 1 ? tri.y : tri.z];
        int2* vs = L.vsnap + (size_t)t * TP_VSNAP_STRIDE;
        int32_t bx, by;
        tp_vertex_stage(p.x, p.y, 0, q, L.vw, bx, by);
        vs[q] = make_int2(bx, by);
        s_X[j][q] = bx; s_Y[j][q] = by;
        xmin = xmax = bx; ymin = ymax = by;
#pragma unroll
        for (int k = 1; k <= 4; k++) {
            int32_t mx, my;
            tp_vertex_stage(p.x, p.y, 4 * q + k, q, L.vw, mx, my);
            vs[2 + 4 * q + k] = make_int2(mx, my);
            xmin = min(xmin, mx); xmax = max(xmax, mx); ymin = min(ymin, my); ymax = max(ymax, my);
            dX = max(dX, abs(mx - bx)); dY = max(dY, abs(my - by));
        }
    }
#pragma unroll
    for (int o = 1; o <= 2; o <<= 1) {  // reduce over the triangle's four lanes
        xmin = min(xmin, __shfl_xor(xmin, o)); xmax = max(xmax, __shfl_xor(xmax, o));
        ymin = min(ymin, __shfl_xor(ymin, o)); ymax = max(ymax, __shfl_xor(ymax, o));
        dX = max(dX, __shfl_xor(dX, o)); dY = max(dY, __shfl_xor(dY, o));
    }
    if (!rebin) return;
    if (q == 0) {
        int tx0 = 0, ty0 = 0, ntx = 1, cnt = 0;
        if (t < L.NT) {
            const int m = L.margin_px;
            const int c0 = max(tp_first_centre(xmin) - m, 0), c1 = min(tp_last_centre(xmax) + m, L.vw.W - 1);
            const int r0 = max(tp_first_centre(ymin) - m, 0), r1 = min(tp_last_centre(ymax) + m, L.vw.H - 1);
            if (c0 <= c1 && r0 <= r1) {
                tx0 = c0 / TW; ty0 = r0 / TH;
                ntx = c1 / TW - tx0 + 1;
                cnt = ntx * (r1 / TH - ty0 + 1);
            }
        }
        s_rect[j][0] = tx0; s_rect[j][1] = ty0; s_rect[j][2] = ntx; s_rect[j][3] = cnt;
        s_infl[j][0] = dX + 256 * L.margin_px; s_infl[j][1] = dY + 256 * L.margin_px;
        s_keep[j] = 0ull;
    }
    __syncthreads();
    if (tid < BIN_TRIS) {  // wave 0: inclusive scan of the rectangle sizes by shuffles
        int inc = s_rect[tid][3];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(inc, o);
            if (tid >= o) inc += v;
        }
        s_excl[tid + 1] = inc;
        if (tid == 0) s_excl[0] = 0;
        if (tid == BIN_TRIS - 1) {
            uint32_t base = 0;
            if (inc) base = atomicAdd(&L.state->pair_total, (uint32_t)inc);
            if (base + (uint32_t)inc > (uint32_t)L.pair_cap) atomicOr(&L.state->flags, TP_FLAG_PAIR_OVERFLOW);
            s_base = base;
        }
    }
    __syncthreads();
    const int total = s_excl[BIN_TRIS];
    const uint32_t base = s_base;
    // ---- one thread per (triangle, tile of its rectangle): keep it if some variant can reach it
    for (int p = tid; p < total; p += 256) {
        int lo = 0, hi = BIN_TRIS;  // largest jj with s_excl[jj] <= p
#pragma unroll
        for (int it = 0; it < 6; it++) {
            const int mid = (lo + hi) >> 1;
            if (s_excl[mid] <= p) lo = mid; else hi = mid;
        }
        const int k = p - s_excl[lo], ntx = s_rect[lo][2];
        const int ky = k / ntx, kx = k - ky * ntx;
        const int txx = s_rect[lo][0] + kx, tyy = s_rect[lo][1] + ky;
        const int32_t X[3] = {s_X[lo][0], s_X[lo][1], s_X[lo][2]}, Y[3] = {s_Y[lo][0], s_Y[lo][1], s_Y[lo][2]};
        const bool dense = s_rect[lo][3] > 64;  // huge triangles: no culling, no mask
        if (!dense && !may_touch(X, Y, s_infl[lo][0], s_infl[lo][1], txx * TW, min(txx * TW + TW - 1, L.vw.W - 1),
                                 tyy * TH, min(tyy * TH + TH - 1, L.vw.H - 1)))
            continue;
        if (!dense) atomicOr(&s_keep[lo], 1ull << k);
        const int tile = tyy * L.tiles_x + txx;
        const int slot = atomicAdd(&L.tilecount[tile], 1);
        if (slot < L.list_cap) {
            tp_list_entry e;
            e.pair = (int)base + p;
            e.tri = blockIdx.x * BIN_TRIS + lo;
            L.tilelist[(size_t)tile * L.list_cap + slot] = e;
        } else
            atomicOr(&L.state->flags, TP_FLAG_LIST_OVERFLOW);
    }
    __syncthreads();
    if (q == 0 && t < L.NT) {
        const int cnt = s_rect[j][3];
        L.tri_pair[t] = make_int2((int)base + s_excl[j], cnt);
        L.tri_mask[t] = cnt > 64 ? ~0ull : s_keep[j];
    }
}

void tp_launch_bin(const tp_launch& L, hipStream_t s) {
    hipLaunchKernelGGL(k_bin, dim3((L.NT + BIN_TRIS - 1) / BIN_TRIS), dim3(256), 0, s, L);
}

// ------------------------------------------------------------------------------------------------
// DPP inclusive scan over each 32-lane half of the wave (all 64 lanes must be active)
// ------------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_add(uint32_t v) {
    return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ uint32_t scan32_inclusive(uint32_t v) {
    v = dpp_add<0x111, 0xf>(v);  // row_shr:1
    v = dpp_add<0x112, 0xf>(v);  // row_shr:2
    v = dpp_add<0x114, 0xf>(v);  // row_shr:4
    v = dpp_add<0x118, 0xf>(v);  // row_shr:8
    v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    return v;
}

// LDS prefix entry (uint4), per row exclusive prefix over the tile's 128 columns:
//   x = sum r,  y = sum g,  z = sum b | n_odd << 20,  w = (sum r^2+g^2+b^2) << 2
// Read as two u64 {x,y} and {z,w}: sums of entries over up to 32 rows never carry between the
// fields that matter -- sum b < 2^20 (4096*255), n_odd <= 4096 spills at most into bit 32, which
// the << 2 on q keeps free -- so a lane accumulates whole entries with 64-bit adds and unpacks once.
struct pix4 { uint32_t x, y, z, w; };

__device__ __forceinline__ pix4 pixel_moments(uint32_t rgba) {
    const uint32_t m = rgba & 0x00ffffffu;
    const uint32_t r = m & 0xffu, g = (m >> 8) & 0xffu, b = m >> 16;
    pix4 o;
    o.x = r; o.y = g;
    o.z = b | (((r + g + b) & 1u) << 20);
    o.w = (r * r + g * g + b * b) << 2;
    return o;
}
__device__ __forceinline__ pix4 operator+(pix4 a, pix4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
__device__ __forceinline__ uint4 as_uint4(pix4 a) { return make_uint4(a.x, a.y, a.z, a.w); }

// ------------------------------------------------------------------------------------------------
// k_accumulate
// ------------------------------------------------------------------------------------------------
// Phase-1 lane mapping (waves 0..3 only): wave w owns tile rows 8w..8w+7; lane = seg*8 + rl walks the
// 16 pixels [16 seg, 16 seg + 16) of row 8w + rl sequentially, and the eight segments of a row are
// combined by a 3-step scan at lane distance 8.  Consecutive lanes belong to consecutive ROWS, whose
// LDS rows are 2064 B = 16 B (mod 128) apart, so every ds_write_b128 lane group hits 8 distinct
// 16-byte slots: the prefix table is written without bank conflicts.
#define P1_WAVES 4
#define P1_PX 16

__device__ __forceinline__ uint32_t scan8_stride8(uint32_t v, int seg) {
#pragma unroll
    for (int d = 1; d < 8; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)v, 8 * d);
        v += seg >= d ? o : 0u;
    }
    return v;
}

__global__ __launch_bounds__(ACC_THREADS) void k_accumulate(tp_launch L) {
    extern __shared__ __attribute__((aligned(16))) uint4 P[];  // [TH][ROWLEN]

    const int tid = threadIdx.x;
    const int ntiles = L.tiles_x * L.tiles_y;
    const int lane = tid & 63, wave = tid >> 6;
    const int rl = lane & 7, seg = lane >> 3, prow = wave * 8 + rl;  // phase-1 role (wave < P1_WAVES)
    if (blockIdx.x == 0 && tid == 0) L.state->rebin_req = 0;  // consumed by the k_bin that ran before us

    // the block walks tiles blockIdx.x, +gridDim.x, ...; the pixels of the next tile are fetched into
    // registers while the spans of the current one are walked
    uint4 px[P1_PX / 4];
    auto fetch = [&](int tile) {
        const int tx = tile % L.tiles_x, ty = tile / L.tiles_x;
        const uint8_t* src = L.img + (size_t)(ty * TH + prow) * L.pitch + (size_t)(tx * TW + seg * P1_PX) * 4;
#pragma unroll
        for (int k = 0; k < P1_PX / 4; k++) px[k] = reinterpret_cast<const uint4*>(src)[k];
    };
    int tile = blockIdx.x;
    if (tile < ntiles && wave < P1_WAVES) fetch(tile);

    for (; tile < ntiles; tile += gridDim.x) {
        const int tx = tile % L.tiles_x, ty = tile / L.tiles_x;
        int nlist = L.tilecount[tile];
        if (nlist > L.list_cap) nlist = L.list_cap;
        const int nitems = nlist * TP_NVARIANTS;
        const tp_list_entry* list = L.tilelist + (size_t)tile * L.list_cap;
        int item = tid;
        tp_list_entry ent = list[item < nitems ? item / TP_NVARIANTS : 0];
        const int next = tile + gridDim.x;

        // ---- phase 1: pixels -> row prefix sums in LDS --------------------------------------
        if (wave < P1_WAVES && nlist > 0 && !(L.debug & 1)) {
            pix4 loc[P1_PX];  // exclusive prefix inside the lane's 16-pixel segment
            pix4 run = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < P1_PX; k++) {
                const uint32_t w = k % 4 == 0 ? px[k / 4].x : k % 4 == 1 ? px[k / 4].y : k % 4 == 2 ? px[k / 4].z : px[k / 4].w;
                loc[k] = run;
                run = run + pixel_moments(w);
            }
            pix4 ex;  // everything left of the segment
            ex.x = scan8_stride8(run.x, seg) - run.x;
            ex.y = scan8_stride8(run.y, seg) - run.y;
            ex.z = scan8_stride8(run.z, seg) - run.z;
            ex.w = scan8_stride8(run.w, seg) - run.w;
            uint4* row = P + prow * ROWLEN + seg * P1_PX;
#pragma unroll
            for (int k = 0; k < P1_PX; k++) row[k] = as_uint4(ex + loc[k]);
            if (seg == 7) row[P1_PX] = as_uint4(ex + run);
        }
        if (next < ntiles && wave < P1_WAVES) fetch(next);  // in flight during phase 2
        __syncthreads();

        // ---- phase 2: one lane per (triangle, variant) of this tile ----------------------------
        const int row0 = ty * TH;
        const int row1 = min(row0 + TH - 1, L.vw.H - 1);
        const int col0 = tx * TW;
        const int colE = min(col0 + TW, L.vw.W);

        if (!(L.debug & 2))
        for (; item < nitems; item += ACC_THREADS) {
            const int e = item / TP_NVARIANTS, v = item - e * TP_NVARIANTS;
            if (item != tid) ent = list[e];
            // snapped vertices of this variant: two base vertices + (v > 0) the displaced one
            const int2* vs = L.vsnap + (size_t)ent.tri * TP_VSNAP_STRIDE;
            const int ms = v > 0 ? (v - 1) >> 2 : 3;
            const int2 q0 = vs[ms == 0 ? 2 + v : 0], q1 = vs[ms == 1 ? 2 + v : 1], q2 = vs[ms == 2 ? 2 + v : 2];
            const int32_t X[3] = {q0.x, q1.x, q2.x}, Y[3] = {q0.y, q1.y, q2.y};
            tp_span sp;
            tp_setup_span(X, Y, row0, row1, sp);
            if (L.debug & 4) sp.r1 = sp.r0 - 1 + (int)(sp.A.x & 1);
            // one row per trip, branch-free: an empty row has hi == lo and its two reads cancel.
            // Whole entries are accumulated with 64-bit adds, hi-side and lo-side apart.
            uint64_t bxy = 0, bzw = 0, axy = 0, azw = 0;
            uint32_t n = 0;
            const ulonglong2* rowp = reinterpret_cast<const ulonglong2*>(P) + (sp.r0 - row0) * ROWLEN - col0;
            for (int r = sp.r0; r <= sp.r1; ++r, rowp += ROWLEN) {
                int32_t lo, hi;
                tp_span_row(sp, col0, colE, lo, hi);
                const ulonglong2 a = rowp[lo], b = rowp[hi];
                n += (uint32_t)(hi - lo);
                bxy += b.x; bzw += b.y; axy += a.x; azw += a.y;
            }
            const uint64_t dxy = bxy - axy, dzw = bzw - azw;
            const uint32_t sr = (uint32_t)dxy, sg = (uint32_t)(dxy >> 32);
            const uint32_t sb = (uint32_t)dzw & 0xfffffu, no = (uint32_t)(dzw >> 20) & 0x3fffu;
            const uint32_t q = (uint32_t)(dzw >> 34);
            if (ent.pair < L.pair_cap) {
                uint32_t* out = L.partials + ((size_t)ent.pair * TP_NVARIANTS + v) * TP_PARTIAL_WORDS;
                reinterpret_cast<uint2*>(out)[0] = make_uint2(n, no);
                reinterpret_cast<uint2*>(out)[1] = make_uint2(sr, sg);
                reinterpret_cast<uint2*>(out)[2] = make_uint2(sb, q);
            }
        }
        if (next < ntiles) __syncthreads();  // the table is rebuilt for the next tile
    }
}

static int accumulate_grid(const tp_launch& L) {
    // every workgroup resident at once (2 per CU by LDS); each walks its tiles with prefetch
    const int ntiles = L.tiles_x * L.tiles_y;
    return ntiles < 512 ? ntiles : 512;
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

__device__ __forceinline__ tp_moments sum_partials(const tp_launch& L, int2 pr, unsigned long long mask, int i) {
    tp_moments m = {0, 0, 0, 0, 0, 0};
    for (int k = 0; k < pr.y; k++) {
        if (pr.y <= 64 && !((mask >> k) & 1ull)) continue;  // tile culled by k_bin: no record
        const int pair = pr.x + k;
        if (pair >= L.pair_cap) break;
        const uint2* in = reinterpret_cast<const uint2*>(L.partials + ((size_t)pair * TP_NVARIANTS + i) * TP_PARTIAL_WORDS);
        const uint2 a = in[0], b = in[1], c = in[2];
        m.n += a.x; m.nodd += a.y; m.sr += b.x; m.sg += b.y; m.sb += c.x; m.q += c.y;
    }
    return m;
}

// ------------------------------------------------------------------------------------------------
// k_finalize: thread per (triangle, variant); id = i*NT + t in the outputs
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_finalize(tp_launch L, int flavour, int write_moments) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= L.NT * TP_NVARIANTS) return;
    const int t = gid / TP_NVARIANTS, i = gid - t * TP_NVARIANTS;
    const tp_moments m = sum_partials(L, L.tri_pair[t], L.tri_mask[t], i);
    const int id = i * L.NT + t;
    int64_t E;
    if (flavour == 0) {
        E = tp_energy_triangulate(m);
        L.ca[id] = make_int4(tp_wrap32(m.sr), tp_wrap32(m.sg), tp_wrap32(m.sb), 0);
    } else {
        const int4 col = L.ca[id];  // stored colour, replicated x13 by upload
        E = tp_energy64(m, col.x, col.y, col.z);
    }
    L.ten[id] = tp_wrap32(E);
    L.cn[id] = tp_wrap32(m.n);
    if (write_moments) {
        int64_t* o = L.moments + (size_t)id * 6;
        o[0] = m.n; o[1] = m.nodd; o[2] = m.sr; o[3] = m.sg; o[4] = m.sb; o[5] = m.q;
    }
}

void tp_launch_finalize(const tp_launch& L, int flavour, bool write_moments, hipStream_t s) {
    const int n = L.NT * TP_NVARIANTS;
    hipLaunchKernelGGL(k_finalize, dim3((n + 255) / 256), dim3(256), 0, s, L, flavour, write_moments ? 1 : 0);
}

// ------------------------------------------------------------------------------------------------
// k_shift: gradient.cs gathered per vertex (no atomics) + shift.cs; also re-arms the work lists
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
// k_update: k_finalize + k_shift in ONE launch (used by tp_iterate).  Thread (g, t): g = 0 handles
// the base variant of triangle t, g = 1..3 the four variants that displace vertex slot s = g-1.
// After writing the reference-layout outputs, a slot thread adds its central differences to its
// vertex with one returning 64-bit atomic per component -- (difference << 32) + 1 -- so the thread
// that completes the vertex's arrival count already holds the whole (wrapping int32) gradient
// component and takes the shift.cs step for it.  x and y never interact in shift.cs, so they are
// settled independently.  Integer sums commute: the result does not depend on arrival order.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int32_t emit_variant(const tp_launch& L, int flavour, int t, int i, const tp_moments& m) {
    const int id = i * L.NT + t;
    int64_t E;
    if (flavour == 0) {
        E = tp_energy_triangulate(m);
        L.ca[id] = make_int4(tp_wrap32(m.sr), tp_wrap32(m.sg), tp_wrap32(m.sb), 0);
    } else {
        const int4 col = L.ca[id];
        E = tp_energy64(m, col.x, col.y, col.z);
    }
    const int32_t e32 = tp_wrap32(E);
    L.ten[id] = e32;
    L.cn[id] = tp_wrap32(m.n);
    return e32;
}

// settle one gradient component; returns true (and the total) for the last arriver
__device__ __forceinline__ bool arrive(unsigned long long* slot, uint32_t contrib, int degree, uint32_t& total) {
    const unsigned long long old = atomicAdd(slot, ((unsigned long long)contrib << 32) + 1ull);
    if ((int)(old & 0xffffffffull) != degree - 1) return false;
    total = (uint32_t)(old >> 32) + contrib;
    *slot = 0ull;  // re-armed for the next launch (nobody else touches it any more)
    return true;
}

__global__ __launch_bounds__(256) void k_update(tp_launch L, int flavour, float rate) {
    __shared__ int s_last;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;

    // one thread per variant.  Threads [0, 12 NT): quads (t, s, k) = the four displacements of
    // vertex slot s, adjacent lanes; threads [12 NT, 13 NT): the base variants.
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
        pb = L.points_binned[v];
    }
    int32_t e = 0;
    if (live) e = emit_variant(L, flavour, t, i, sum_partials(L, L.tri_pair[t], L.tri_mask[t], i));
    // central differences inside the quad: lanes 4q+0/1 hold E(+dx)/E(-dx), 4q+2/3 E(+dy)/E(-dy)
    const uint32_t e1 = (uint32_t)__shfl_xor(e, 1);
    const uint32_t gx = (uint32_t)e - e1;                     // valid on even lanes of the quad
    const uint32_t gy = (uint32_t)__shfl_down((int)gx, 2);    // lane 4q+0 fetches lane 4q+2's value
    int need = L.margin_px < 2;                                // margin off: rebuild every iteration
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
    // grid-wide arrival: the last block knows whether ANY vertex left its margin and, if so,
    // re-arms the work lists so that the next k_bin rebuilds them
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
            if (s_last == 2) { L.state->pair_total = 0; L.state->rebin_req = 1; L.state->rebin_count++; }
        }
    }
}

void tp_launch_update(const tp_launch& L, int flavour, float rate, hipStream_t s) {
    const int n = 13 * L.NT;
    hipLaunchKernelGGL(k_update, dim3((n + 255) / 256), dim3(256), 0, s, L, flavour, rate);
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
