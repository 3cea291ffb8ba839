// tp_plan.h -- host side of the persistent grad-iter kernel (tp_persist.hip): the per-upload PLAN that cuts the mesh
// into patches, one per workgroup.
//
// What it replaces in the reference: nothing -- the reference issues two instanced draws and two dispatches per frame
// (software/triangulate/main.cpp:132-155) and lets the GL driver schedule them.  Here K grad-iters run inside ONE
// launch; a workgroup owns a compact patch of the mesh -- some VERTICES -- for the whole launch and computes everything
// their gradients need itself: for every corner (own vertex, incident triangle) the four displaced lines of the two edges
// at the vertex and the base line of the opposite edge (tp_raster.h, "Edge-centric form").  Only vertex positions travel
// between workgroups, from a vertex's owner to the owners of its neighbours: 16 bytes per vertex and grad-iter.  Line sums
// never leave a workgroup (a first version handed the sums of border edges over instead of recomputing them: each hop cost
// 2.4 us or more against 0.7 us for positions -- profiles/r03_*); the price is that the base line of a border edge is
// walked by up to three workgroups (+10 % lines at 3000 triangles) and its band of the table is read by each of them.
// Ownership is decided from the topology and the positions the plan is cut from (recursive coordinate bisection balanced
// by table look-ups) and stays fixed for a launch: there is nothing to re-bin while grad-iters run.  A long descent cuts a
// new plan between launches when the mesh has drifted (tp_context.hip: maybe_replan); how each patch cuts its LINES into
// lane chunks is decided by the workgroup itself, again and again (tp_persist.h: pk_recut_line).  Results cannot depend on
// either cut (integer sums commute; every float operation is per vertex).
//
// Plain C++ (no HIP): tests/emul compiles this header with g++ and replays the kernel's lane functions
// (tp_persist.h) phase by phase on the CPU.
#pragma once

#include <stdint.h>
#include <math.h>
#include <stdlib.h>
#include <algorithm>
#include <string>
#include <unordered_map>
#include <vector>

#define PK_NLINES 9
#define PK_SUM_WORDS 4          /* 64-bit words of a line's sums in LDS (tp_persist.h: pk_fold_words) */
#ifndef PK_SUM_STRIDE
#define PK_SUM_STRIDE 5         /* ... and the words from one line's sums to the next: at 4 (32 bytes) the lines a wave folds into fall on
                                   8 of the 32 LDS banks; at 5 they spread over all of them */
#endif
/* Threads per workgroup and lane-items per thread (round 5: 768 x 1, three waves per SIMD; rounds 3-4: 512 x 2).  The register file holds
   the same number of table records either way, but a wave walks at the LATENCY of its own instruction stream -- the row of a walk is one
   chain of dependent instructions (shift, clamp, mask, compare, branch, address, load) -- so what fills a SIMD is waves, not work per wave:
   two waves of 24 rows keep the VALU ~70 % busy, three of 14-16 rows ~85 % (profiles/r05_experiments.txt). */
#ifndef PK_THREADS
#define PK_THREADS 768
#endif
#ifndef PK_WG_PER_CU
#define PK_WG_PER_CU 1        /* workgroups (patches) per compute unit */
#endif
#ifndef PK_NI
#define PK_NI 1                /* lane-items of the walk a thread keeps table records for, in registers */
#endif
#define PK_CACHED (PK_THREADS * PK_NI)
#ifndef PK_ROWS_PER_LANE
#define PK_ROWS_PER_LANE 16  /* table records a lane of the walk keeps in registers: the most rows per lane */
#endif
/* the instantiations of the kernel: rows a lane keeps records for (fewer rows, fewer registers and less straight-line code) */
#define PK_RR0 (PK_ROWS_PER_LANE * 2 / 3)
#define PK_RR1 (PK_ROWS_PER_LANE * 5 / 6)
#define PK_RR2 (PK_ROWS_PER_LANE * 11 / 12)
/* Rows per lane beyond the registers (round 5): a plan whose patches need more than PK_ROWS_PER_LANE rows per lane -- an aged mesh: its lines have
   grown -- takes up to PK_LDS_ROWS more, and every thread keeps the records of those rows of its lane-item in LDS (tp_persist.h, pk_walk_lds_rows):
   a row more in the same walk costs a seventh of what a lane-item more does */
#define PK_LDS_ROWS 4
#define PK_ROWS_MAX (PK_ROWS_PER_LANE + PK_LDS_ROWS)
#define PK_ROWS_MID (PK_ROWS_PER_LANE + PK_LDS_ROWS / 2)   /* an instantiation between: the first plans beyond the registers need a row or two */
/* Round 6: EIGHT rows in LDS (24 per lane) for the plans that need them.  On a photograph the patches of standing vertices take the rows the
   fast ones shed and outgrow 768 slots x 20 rows between two plans; the lane-items without a slot they then walk cost such a patch 1.4-2.1 us of
   every grad-iter and made them the slowest of the grid (profiles/r06_persist_timeline_after4000.json).  Since a wave whose lanes all have at
   most 16 rows skips the LDS rows, and one whose lanes have at most 20 the second four (tp_persist.h: pk_lds_rows_idle), the extra rows cost only the
   waves that have them. */
#define PK_LDS_ROWS_BIG 8
#define PK_ROWS_BIG (PK_ROWS_PER_LANE + PK_LDS_ROWS_BIG)
#define PK_LDS_ROW_BYTES (PK_THREADS * PK_NI * 16)         /* per row: a record for every thread */
#define PK_LDS_COL_BYTES (PK_THREADS * PK_NI * 16)         /* ... and, once, the crossing columns of up to eight such rows for every thread (two 64-bit words of four) */
#define PK_MAX_SLOTS 1023    /* position slots of a workgroup (10-bit fields of the corner records) */
#define PK_MAX_TL 4095       /* chunks per line */
#ifndef PK_SLACK_ROWS
#define PK_SLACK_ROWS 3      /* rows a line may grow before its chunks are cut again */
#endif
#ifndef PK_STALE_COST
#define PK_STALE_COST 1.8f   /* what a table look-up costs MORE when its row's crossing column has changed since the last grad-iter (the record is fetched
                                again: a cache line through the CU's texture path; profiles/r06_ta_bench.txt, r06_meninas_timeline_2000_rows_only_plan.json: a
                                patch whose rows are all stale walks for 9 us, one whose rows stand for 1.7) -- the planner's weight of a row is
                                1 + this x P(stale).  Swept on four rasters (profiles/r06_experiments.txt): 1.5 / 2.5 / 4 / 6 -> meninas 8.3 / 8.0 / 8.3 /
                                9.4 us per grad-iter: beyond 2.5 the patches of standing vertices get more rows than their threads keep records for -- and again once hot
                                patches fetched their stale rows from shared cache lines (experiments 17-19): 1.2 / 1.8 / 2.5 / 3.5 -> meninas 6.9 / 7.0 / 7.35 / 8.1 in long
                                calls, 10.5 / 9.8 / 10.4 / 10.9 in calls of 20; 1.8 kept */
#endif
#ifndef PK_RECUT
#define PK_RECUT 64          /* grad-iters between two looks at the chunks of a patch's lines (a power of two) */
#endif

// per-workgroup header; every `off_*` indexes pk_plan::pool (int32 units)
struct pk_wg {
    int32_t n_own_v, n_slots;  // position slots: the patch's own vertices first, then the neighbours it reads
    int32_t n_edges;           // local edges: every edge one of the patch's corners uses
    int32_t n_lines;           // lines the patch walks every grad-iter = line-sum slots (per local edge: its needed versions, ascending)
    int32_t n_lines_all;       // ... plus the base lines only the LAST grad-iter of a call walks (outputs of the base variants)
    int32_t n_li, n_li_all;    // lane-items of the walk, (line, chunk), at the plan's positions; likewise (the workgroup cuts its
                               // lines into chunks itself, from the positions it has: tp_persist.h, pk_recut_line)
    int32_t n_corners;         // (own vertex, incident triangle): four lanes each, one per move
    int32_t n_base;            // triangles whose first vertex the patch owns: it writes their base variant's outputs
    int32_t li_cap;            // entries of the table of lane-items no thread keeps records for (beyond PK_CACHED, and the last grad-iter's)
    int32_t rows;              // rows a lane of the walk takes: chunks per line = the line's rows / this (<= PK_ROWS_PER_LANE)
    int32_t off_vid;           // [n_slots] global vertex id
    int32_t off_edges;         // [n_edges] slot_u | slot_v << 16
    int32_t off_lines;         // [n_lines_all] local edge | version << 16
    int32_t off_corners;       // [n_corners] {t, s | own << 2 | slot_a << 12 | slot_b << 22, out | in << 16, opp | flips << 16} -- flips: the half-edge
                               // leaving the vertex | arriving << 1 | opposite << 2 runs against its edge's first -> second endpoint
    int32_t off_base;          // [n_base] {t, own | slot_1 << 10 | slot_2 << 20, line of edge 0 | edge 1 << 16, line of edge 2 | flips of edges 0, 1, 2 << 16}
    int32_t lds_bytes;         // dynamic LDS of this workgroup (pk_lds_bytes)
    int32_t lds_rows;          // rows per lane whose records live in LDS: 0, or PK_LDS_ROWS when some patch of the plan takes more than PK_ROWS_PER_LANE (the same for every patch)
    int32_t hot;               // 1 (round 6): most of this patch's rows change their crossing column every grad-iter (the planner knows from the vertices'
                               // speeds) -- its slots are handed out line by line onto ADJACENT lanes and stale rows come from the tiled table, so that
                               // the lanes of a wave-load ask for consecutive rows of one line: one cache line for up to four of them
};
#ifndef PK_HOT_FRACTION
#define PK_HOT_FRACTION 0.1    /* share of a patch's rows expected stale per grad-iter from which the patch counts as hot.  Swept before the hot patches folded
                                  in rotated word order (0.25 ... 0.75: little difference) and after (0.03 / 0.07 / 0.1 / 0.15 / every patch hot: meninas' long calls 7.35 /
                                  7.3 / 7.3 / 7.35 / 7.35 us against 7.6-7.8 at 0.4; every patch hot costs the synthetic x0.10 raster 0.2 us, 0.03-0.15 nothing) */
#endif

struct pk_plan {
    bool ok = false;
    std::string why;           // when !ok: why the triangulation takes the two-kernel path instead
    int parts = 0;             // workgroups (grid size)
    int lds_bytes = 0;         // max over workgroups
    int rows_max = 0;          // most rows per lane of any patch
    int rows_cap = 0;          // the most a lane was allowed when this plan was cut (what the LDS left room for: the next cut of the same mesh starts there)
    std::vector<pk_wg> wg;
    std::vector<int32_t> pool;
    std::vector<int32_t> owner_v;           // (kept for tests and statistics)
    std::vector<float> patch_work, patch_rows;   // (likewise: what the planner weighed every patch with -- rows with and without the stale rows' cost)
    double work_max = 0.0, work_mean = 0.0; // weighted table look-ups per workgroup when the plan was cut (pk_edge_costs)
    int64_t lines_total = 0, foreign_total = 0;  // lines walked by all patches (9 NE if nothing were walked twice); foreign position slots
};

// dynamic LDS carve (bytes), identical on the device (tp_persist.h: pk_carve)
#if defined(__HIPCC__)
#define PK_HD __host__ __device__ inline
#else
#define PK_HD inline
#endif
PK_HD int pk_align16(int v) { return (v + 15) & ~15; }
// the instantiation that runs a plan whose patches take at most `rows` rows per lane: one row more than the plan's where
// there is one (a line that has grown by a chunk's worth of rows since the plan was cut still fits the records its lanes keep)
PK_HD int pk_rr_for(int rows) { return rows < PK_RR0 ? PK_RR0 : rows < PK_RR1 ? PK_RR1 : rows < PK_RR2 ? PK_RR2 : rows <= PK_ROWS_PER_LANE ? PK_ROWS_PER_LANE : rows <= PK_ROWS_MID ? PK_ROWS_MID : PK_ROWS_MAX; }
// chunks of a line of `rows` pixel rows when a lane takes `rpl` of them
PK_HD int pk_chunks(int rows, int rpl) {
    const int t = (rows + PK_SLACK_ROWS + rpl - 1) / rpl;
    return t < 1 ? 1 : t > PK_MAX_TL ? PK_MAX_TL : t;
}
inline int pk_lds_bytes(const pk_wg& w) {
    int b = 0;
    b += pk_align16(w.n_lines_all * 8 * PK_SUM_STRIDE);  // line sums
    b += pk_align16(w.n_lines_all * 24);            // walkers
    b += pk_align16(w.n_slots * 8);                 // positions
    b += pk_align16(w.n_own_v * 16);                // gradient: per own vertex and axis {corners counted : 32, sum of their central differences : 32}
    b += pk_align16(w.n_own_v * 4);                 // corners of every own vertex
    b += pk_align16(w.n_own_v * 8);                 // how far every own vertex has moved during the launch, per axis (for the planner)
    b += pk_align16(w.n_slots * 4);                 // vid
    b += pk_align16(w.n_edges * 4);                 // edges
    b += pk_align16(w.n_lines_all * 4);             // lines
    b += pk_align16((w.n_lines_all + 1) * 4);       // first lane-item WITHOUT a thread of its own of every line (and the total)
    b += 2 * pk_align16(w.n_lines_all * 4);         // chunks of every line; how many of them have a thread (slot) of their own
    b += 2 * pk_align16(PK_CACHED * 4);             // while the lines are cut again: the lane-item handed to a slot; the free slots
    b += pk_align16(w.li_cap * 12);                 // lane-items without a thread of their own
    b += pk_align16(w.n_corners * 16);              // corners
    b += pk_align16(w.n_base * 16);                 // base variants
    b += pk_align16(w.n_lines_all * 4);             // which way every line runs down the raster (this grad-iter's)
    b += 64;                                        // flags
    return b + w.lds_rows * PK_LDS_ROW_BYTES + (w.lds_rows ? PK_LDS_COL_BYTES : 0);   // records and crossing columns of the rows beyond the registers
}

namespace pk_detail {
struct rcb_vertex { int v; float x, y; double w; };

// recursive coordinate bisection of idx[lo, hi) into the parts [p0, p1), balanced by weight
inline void rcb(std::vector<rcb_vertex>& a, int lo, int hi, int p0, int p1, std::vector<int32_t>& owner) {
    if (lo >= hi) return;
    if (p1 - p0 <= 1) { for (int k = lo; k < hi; k++) owner[a[k].v] = p0; return; }
    if (hi - lo <= p1 - p0) {  // fewer vertices than parts: one each
        for (int k = lo; k < hi; k++) owner[a[k].v] = p0 + (k - lo);
        return;
    }
    float x0 = a[lo].x, x1 = a[lo].x, y0 = a[lo].y, y1 = a[lo].y;
    double total = 0.0;
    for (int k = lo; k < hi; k++) {
        x0 = std::min(x0, a[k].x); x1 = std::max(x1, a[k].x); y0 = std::min(y0, a[k].y); y1 = std::max(y1, a[k].y);
        total += a[k].w;
    }
    const bool byx = (x1 - x0) >= (y1 - y0);
    std::sort(a.begin() + lo, a.begin() + hi, [byx](const rcb_vertex& p, const rcb_vertex& q) {
        const float pa = byx ? p.x : p.y, qa = byx ? q.x : q.y;
        return pa < qa || (pa == qa && p.v < q.v);
    });
    const int pm = p0 + (p1 - p0) / 2;
    const double want = total * (double)(pm - p0) / (double)(p1 - p0);
    double run = 0.0;
    int cut = lo;
    while (cut < hi && run + a[cut].w * 0.5 < want) { run += a[cut].w; cut++; }
    cut = std::max(lo + 1, std::min(cut, hi - 1));  // (hi - lo > p1 - p0 >= 2: neither side stays empty)
    rcb(a, lo, cut, p0, pm, owner);
    rcb(a, cut, hi, pm, p1, owner);
}
}  // namespace pk_detail

// What the planner balances.  rows[e]: the table look-ups of ONE line of edge e per grad-iter -- its pixel rows at `points`, each weighted by
// 1 + PK_STALE_COST x P(its crossing column changes from one grad-iter to the next), where P = min(1, mean over the edge's endpoints of
// vspeed[v]) and vspeed[v] is how far the vertex moves per grad-iter in PIXELS (|dx| + |dy|, averaged over the last launch by the kernel
// itself: tp_persist.hip, `vspeed`; null: nothing is known, every row counts 1).  The reference's fixed-step descent (shift.cs:45) makes
// vertices in high-contrast regions jump a pixel or more per grad-iter for as long as it runs: their rows are fetched again every time and
// a patch made of such vertices took five times as long as its neighbours, who waited (profiles/r06_meninas_timeline_2000.json).
// wv[v]: a vertex's work -- per corner the four moves of its two edges (each edge at the vertex is shared by two corners) and the opposite
// base line (shared with the corner across the edge, if that one is in the same patch), plus the corner itself.  Returns the sum of wv.
inline float pk_stale_cost() {   // (TPOSE_STALE_COST: tuning runs only -- tools/photo_timing.py)
    static const float v = [] { const char* e = getenv("TPOSE_STALE_COST"); return e ? (float)atof(e) : PK_STALE_COST; }();
    return v;
}
inline double pk_vertex_work(int NP, int NT, const int32_t* tris, const float* points, int NE, const int32_t* edge_uv, const int32_t* he_edge, int H,
                             const float* vspeed, std::vector<float>& rows, std::vector<double>& wv, std::vector<int>& deg) {
    rows.assign((size_t)NE, 0.0f);
    for (int e = 0; e < NE; e++) {
        const int u = edge_uv[2 * (size_t)e] & 0x3fffffff, v = edge_uv[2 * (size_t)e + 1] & 0x3fffffff;
        const float ya = points[2 * (size_t)u + 1], yb = points[2 * (size_t)v + 1];
        float r = fabsf(ya - yb) * 0.5f * (float)H;
        if (!(r >= 0.0f)) r = 0.0f;                   // NaN positions: no rows
        r = std::min(r, (float)H) + 1.0f;
        if (vspeed) {
            float st = 0.5f * (vspeed[u] + vspeed[v]);
            if (!(st >= 0.0f)) st = 0.0f;
            r *= 1.0f + pk_stale_cost() * std::min(st, 1.0f);
        }
        rows[e] = r;
    }
    wv.assign((size_t)NP, 0.0);
    deg.assign((size_t)NP, 0);
    double total = 0.0;
    for (int t = 0; t < NT; t++)
        for (int s = 0; s < 3; s++) {
            const int v = tris[4 * (size_t)t + s];
            const int sn = s == 2 ? 0 : s + 1, sp = s == 0 ? 2 : s - 1;
            const double w = 2.0 * (rows[he_edge[3 * (size_t)t + s] >> 1] + rows[he_edge[3 * (size_t)t + sp] >> 1]) +
                             0.75 * rows[he_edge[3 * (size_t)t + sn] >> 1] + 40.0;
            deg[v]++; wv[v] += w; total += w;
        }
    return total;
}
// how far the heaviest patch of an assignment of vertices to patches lies above the mean, under the weights wv
inline double pk_imbalance(const std::vector<int32_t>& owner_v, const std::vector<double>& wv, int parts) {
    if (parts < 1) return 1.0;
    std::vector<double> load((size_t)parts, 0.0);
    double total = 0.0;
    for (size_t v = 0; v < owner_v.size() && v < wv.size(); v++)
        if (owner_v[v] >= 0 && owner_v[v] < parts) { load[(size_t)owner_v[v]] += wv[v]; total += wv[v]; }
    double most = 0.0;
    for (double l : load) most = std::max(most, l);
    return total > 0.0 ? most * (double)parts / total : 1.0;
}

// Build the plan.  tris: ivec4[NT]; points: vec2[NP] (upload-time positions); edge_uv: int[2 NE] endpoint ids (the low
// 30 bits; tp_upload keeps flags above); he_edge: int[3 NT] edge * 2 + direction; W, H: raster; dp_px: a hint, the size
// of the moves in pixels (lines get a few rows longer or shorter).  max_parts: workgroups that can be resident at once.
// base_every: the base lines of every triangle are walked in every grad-iter (not only in the last one of a call).
inline void pk_build_plan(int NP, int NT, const int32_t* tris, const float* points, int NE, const int32_t* edge_uv,
                          const int32_t* he_edge, int W, int H, float ratio, float dp_px, int max_parts, int lds_limit,
                          pk_plan& P, bool base_every = false, int rows_cap = PK_ROWS_BIG, const float* vspeed = nullptr) {
    P = pk_plan();
    P.rows_cap = rows_cap;
    if (NT < 1 || NE < 1 || max_parts < 1) { P.why = "empty triangulation"; return; }
    auto EU = [&](int e) { return edge_uv[2 * (size_t)e] & 0x3fffffff; };
    auto EV = [&](int e) { return edge_uv[2 * (size_t)e + 1] & 0x3fffffff; };
    for (int e = 0; e < NE; e++)
        if (EU(e) == EV(e)) { P.why = "an edge names one vertex twice"; return; }

    // rows of every edge at upload: table look-ups per line -- weighted by how often a row's record has to be fetched again (vspeed)
    std::vector<float> rows;
    std::vector<double> wv;
    std::vector<int> deg;
    const double total = pk_vertex_work(NP, NT, tris, points, NE, edge_uv, he_edge, H, vspeed, rows, wv, deg);

    // workgroups: at least ~4 look-ups per lane each, at most one per used vertex
    int used = 0;
    for (int v = 0; v < NP; v++) used += deg[v] > 0;
    int parts = (int)std::min<double>((double)max_parts, std::ceil(total / (double)(PK_CACHED * 4)));
    parts = std::max(1, std::min(parts, used));
    if (parts >= 16) parts &= ~7;  // whole runs of patches per XCD (tp_persist.hip maps workgroup b to XCD b mod 8)

    // 1. vertices -> patches
    std::vector<pk_detail::rcb_vertex> a;
    a.reserve((size_t)used);
    for (int v = 0; v < NP; v++) {
        if (!deg[v]) continue;
        float x = points[2 * (size_t)v] / ratio * 0.5f * (float)W, y = points[2 * (size_t)v + 1] * 0.5f * (float)H;
        if (!(x == x)) x = 0.0f;
        if (!(y == y)) y = 0.0f;
        a.push_back({v, x, y, wv[v]});
    }
    P.owner_v.assign((size_t)NP, -1);
    pk_detail::rcb(a, 0, (int)a.size(), 0, parts, P.owner_v);
    // 1b. even the patches out: a vertex on a patch border moves to the neighbouring patch whenever that narrows the gap
    // between the two (the heaviest patch sets the pace of every grad-iter; bisection alone leaves it ~15 % above the mean)
    if (parts > 1) {
        std::vector<double> load((size_t)parts, 0.0);
        std::vector<int> count((size_t)parts, 0);
        for (auto& q : a) { load[P.owner_v[q.v]] += q.w; count[P.owner_v[q.v]]++; }
        for (int pass = 0; pass < 12; pass++) {
            int moved = 0;
            for (int e = 0; e < NE; e++) {
                int u = EU(e), v = EV(e);
                int A = P.owner_v[u], B = P.owner_v[v];
                if (A == B) continue;
                if (load[A] < load[B]) { std::swap(u, v); std::swap(A, B); }   // u sits in the heavier patch A
                if (count[A] <= 1 || load[A] - load[B] <= wv[u]) continue;      // (moving u must leave A above B: no ping-pong)
                P.owner_v[u] = B;
                load[A] -= wv[u]; load[B] += wv[u]; count[A]--; count[B]++;
                moved++;
            }
            if (!moved) break;
        }
    }

    // which patches are hot: the share of their rows expected stale, from the weights with and without the speeds
    std::vector<double> wcold;
    static const bool no_hot = getenv("TPOSE_NO_HOT_LAYOUT") != nullptr;   // (A/B)
    if (vspeed && !no_hot) { std::vector<float> r0; std::vector<int> d0; pk_vertex_work(NP, NT, tris, points, NE, edge_uv, he_edge, H, nullptr, r0, wcold, d0); }
    std::vector<std::vector<int>> own_v((size_t)parts);
    for (auto& q : a) own_v[P.owner_v[q.v]].push_back(q.v);   // (a is in bisection order: neighbours stay neighbours)
    // vertex -> corners (t, s), in triangle order
    std::vector<int> voff((size_t)NP + 1, 0), vadj((size_t)3 * NT);
    for (int t = 0; t < NT; t++) for (int s = 0; s < 3; s++) voff[tris[4 * (size_t)t + s] + 1]++;
    for (int v = 0; v < NP; v++) voff[v + 1] += voff[v];
    { std::vector<int> cur(voff.begin(), voff.end() - 1);
      for (int t = 0; t < NT; t++) for (int s = 0; s < 3; s++) vadj[cur[tris[4 * (size_t)t + s]]++] = 3 * t + s; }

    // 2. per-patch tables.  Scratch indexed by global ids is stamped with the patch number instead of being cleared.
    P.parts = parts;
    P.wg.assign((size_t)parts, pk_wg());
    std::vector<int> vslot((size_t)NP, 0), vstamp((size_t)NP, -1), eloc((size_t)NE, 0), estamp((size_t)NE, -1);
    std::vector<int32_t> vid, edges, emask, eglob, lines, corners, first, base;
    P.pool.reserve((size_t)NT * 128 + 8192);
    for (int p = 0; p < parts; p++) {
        pk_wg& w = P.wg[p];
        vid.clear(); edges.clear(); emask.clear(); eglob.clear(); lines.clear(); corners.clear(); base.clear();
        auto slot = [&](int v) {
            if (vstamp[v] != p) { vstamp[v] = p; vslot[v] = (int)vid.size(); vid.push_back(v); }
            return vslot[v];
        };
        // which versions of edge e the patch needs: bit 0 the base line, bit 1 versions 1..4 (first endpoint displaced),
        // bit 2 versions 5..8 (second endpoint displaced); bit 3: the base line, for the outputs of a base variant only
        auto need = [&](int e, int bits) {
            if (estamp[e] != p) {
                estamp[e] = p; eloc[e] = (int)eglob.size();
                eglob.push_back(e); emask.push_back(0);
            }
            emask[eloc[e]] |= bits;
        };
        for (int v : own_v[p]) slot(v);
        w.n_own_v = (int)own_v[p].size();
        if (!wcold.empty()) {
            double a = 0.0, b = 0.0;
            for (int v : own_v[p]) { a += wv[v]; b += wcold[v]; }
            static const double hot_from = [] { const char* e = getenv("TPOSE_HOT_FRACTION"); return e ? atof(e) : (double)PK_HOT_FRACTION; }();   // (tuning runs only)
            w.hot = (b > 0.0 && (a / b - 1.0) / (double)pk_stale_cost() > hot_from) ? 1 : 0;
            P.patch_work.push_back((float)a); P.patch_rows.push_back((float)b);
        }
        for (int v : own_v[p])
            for (int j = voff[v]; j < voff[v + 1]; j++) {
                const int h = vadj[j], t = h / 3, s = h - 3 * t;
                const int sn = s == 2 ? 0 : s + 1, sp = s == 0 ? 2 : s - 1;
                const int he_out = he_edge[3 * (size_t)t + s], he_in = he_edge[3 * (size_t)t + sp], he_opp = he_edge[3 * (size_t)t + sn];
                // edge leaving the vertex: the vertex is its origin -- the edge's second endpoint when the half-edge is flipped
                // (tp_edge_version: flipped ? 4 + m : m); edge arriving: the vertex is its destination (flipped ? m : 4 + m)
                need(he_out >> 1, (he_out & 1) ? 4 : 2);
                need(he_in >> 1, (he_in & 1) ? 2 : 4);
                need(he_opp >> 1, 1);
                // the triangle's base variant is this patch's to write: in the last grad-iter of a call, or (tp_iterate_until: the
                // energy of every frame is wanted) in every grad-iter
                if (s == 0) { need(he_out >> 1, base_every ? 1 : 8); need(he_in >> 1, base_every ? 1 : 8); }
            }
        // local edges in order of first use; their lines (needed versions, ascending) and lane-items
        w.n_edges = (int)eglob.size();
        if (w.n_edges > 65535) { P.why = "a patch uses more than 65535 edges"; return; }
        first.assign((size_t)w.n_edges * PK_NLINES, -1);
        // rows per lane: the fewest (down to 4) that still give every lane-item its own thread, so that lanes can keep their
        // table records in registers from one grad-iter to the next (tp_persist.h)
        // chunks of line q of a local edge: its own rows -- a move in y makes the line dp longer or shorter, a move in x
        // leaves its rows alone -- over the rows per lane (an estimate for the choice of that number: the workgroup counts again)
        auto chunks = [&](int le, int q, int rpl) {
            const int e = eglob[le];
            const int mu = (q >= 1 && q <= 4) ? q : 0, mv = q >= 5 ? q - 4 : 0;
            float yu = points[2 * (size_t)EU(e) + 1] * 0.5f * (float)H, yv = points[2 * (size_t)EV(e) + 1] * 0.5f * (float)H;
            yu += mu == 3 ? dp_px : mu == 4 ? -dp_px : 0.0f;
            yv += mv == 3 ? dp_px : mv == 4 ? -dp_px : 0.0f;
            float rr = fabsf(yu - yv);
            if (!(rr >= 0.0f)) rr = 0.0f;   // NaN positions: no rows
            return pk_chunks((int)std::ceil(std::min(rr, (float)H)) + 1, rpl);
        };
        auto active = [&](int le, int q) { return (emask[le] & (q == 0 ? 1 : q <= 4 ? 2 : 4)) != 0; };
        int rpl = 4;
        for (; rpl < rows_cap; rpl++) {
            long n = 0;
            for (int le = 0; le < w.n_edges; le++)
                for (int q = 0; q < PK_NLINES; q++) if (active(le, q)) n += chunks(le, q, rpl);
            if (n <= PK_CACHED - PK_CACHED / 64) break;   // (a little room: lines grow and shrink while the descent runs)
            // (rows beyond the registers cost every patch of the plan 0.35 us per grad-iter, lane-items without a slot cost the patch that
            // has them twice that: the step from PK_ROWS_PER_LANE up is taken when the chunks do not fit at all)
            // (... and this count runs a row or two per line ahead of the workgroup's own: 3 % more chunks than slots still fit there)
            if (rpl == PK_ROWS_PER_LANE && n <= PK_CACHED + PK_CACHED / 32) break;
        }
        w.rows = rpl;
        int n_li = 0;
        for (int le = 0; le < w.n_edges; le++) {
            const int e = eglob[le];
            edges.push_back(slot(EU(e)) | (slot(EV(e)) << 16));
            for (int q = 0; q < PK_NLINES; q++) {
                if (!active(le, q)) continue;
                first[(size_t)le * PK_NLINES + q] = (int)lines.size();
                lines.push_back(le | (q << 16));
                n_li += chunks(le, q, rpl);   // (the chunks of a line are consecutive lane-items)
            }
        }
        w.n_lines = (int)lines.size();
        w.n_li = n_li;
        // base lines only the outputs of base variants need: walked by the last grad-iter of a call, after the others
        for (int le = 0; le < w.n_edges; le++) {
            if ((emask[le] & 9) != 8) continue;
            first[(size_t)le * PK_NLINES] = (int)lines.size();
            lines.push_back(le);
            n_li += chunks(le, 0, rpl);
        }
        w.n_lines_all = (int)lines.size();
        w.n_li_all = n_li;
        {   // (room for the lines to grow: what does not fit the table is looked up the slow way)
            const int beyond = n_li - std::min(w.n_li, PK_CACHED);
            w.li_cap = beyond + beyond / 8 + 64;
        }
        if (w.n_lines_all > 65535) { P.why = "a patch walks more than 65535 lines"; return; }
        for (int k = 0; k < w.n_own_v; k++) {
            const int v = own_v[p][k];
            for (int j = voff[v]; j < voff[v + 1]; j++) {
                const int h = vadj[j], t = h / 3, s = h - 3 * t;
                const int sn = s == 2 ? 0 : s + 1, sp = s == 0 ? 2 : s - 1;
                const int va = tris[4 * (size_t)t + sn], vb = tris[4 * (size_t)t + sp];
                const int he_out = he_edge[3 * (size_t)t + s], he_in = he_edge[3 * (size_t)t + sp], he_opp = he_edge[3 * (size_t)t + sn];
                const int so = first[(size_t)eloc[he_out >> 1] * PK_NLINES + ((he_out & 1) ? 5 : 1)];
                const int si = first[(size_t)eloc[he_in >> 1] * PK_NLINES + ((he_in & 1) ? 1 : 5)];
                const int sopp = first[(size_t)eloc[he_opp >> 1] * PK_NLINES];
                corners.push_back(t);
                corners.push_back(s | (k << 2) | (slot(va) << 12) | (slot(vb) << 22));
                corners.push_back(so | (si << 16));
                corners.push_back(sopp | ((he_out & 1) << 16) | ((he_in & 1) << 17) | ((he_opp & 1) << 18));
                if (s == 0) {
                    base.push_back(t);
                    base.push_back(k | (slot(va) << 10) | (slot(vb) << 20));
                    base.push_back(first[(size_t)eloc[he_out >> 1] * PK_NLINES] | (sopp << 16));
                    base.push_back(first[(size_t)eloc[he_in >> 1] * PK_NLINES] | ((he_out & 1) << 16) | ((he_opp & 1) << 17) | ((he_in & 1) << 18));
                }
            }
        }
        w.n_corners = (int)(corners.size() / 4);
        w.n_base = (int)(base.size() / 4);
        w.n_slots = (int)vid.size();
        if (w.n_slots > PK_MAX_SLOTS || w.n_own_v > 1023) { P.why = "a patch reads more than 1023 vertices"; return; }
        auto put = [&](const std::vector<int32_t>& src) {
            const int off = (int)P.pool.size();
            P.pool.insert(P.pool.end(), src.begin(), src.end());
            while (P.pool.size() & 3) P.pool.push_back(0);  // 16-byte aligned tables
            return off;
        };
        w.off_vid = put(vid); w.off_edges = put(edges); w.off_lines = put(lines); w.off_corners = put(corners);
        w.off_base = put(base);
        w.lds_bytes = pk_lds_bytes(w);
        P.lds_bytes = std::max(P.lds_bytes, w.lds_bytes);
        P.rows_max = std::max(P.rows_max, w.rows);
        P.lines_total += w.n_lines; P.foreign_total += w.n_slots - w.n_own_v;
        double work = 40.0 * w.n_corners;
        for (int l = 0; l < w.n_lines; l++) work += rows[eglob[lines[(size_t)l] & 0xffff]];
        P.work_max = std::max(P.work_max, work); P.work_mean += work / parts;
    }
    if (P.lds_bytes > lds_limit) { P.why = "a patch does not fit the LDS"; return; }
    if (P.rows_max > PK_ROWS_PER_LANE) {   // rows beyond the registers: in LDS, if the tables leave the room -- or the plan again with fewer rows per lane
        // (eight rows from 19 rows per lane on: lines grow between two plans, and the workgroup may take a row more per lane by itself up to what
        // its instantiation keeps -- tp_persist.hip, the cut)
        const int want = P.rows_max > PK_ROWS_MID ? PK_LDS_ROWS_BIG : PK_LDS_ROWS;
        int have = want;
        if (P.lds_bytes + have * PK_LDS_ROW_BYTES + PK_LDS_COL_BYTES > lds_limit) have = P.rows_max <= PK_ROWS_MAX ? PK_LDS_ROWS : 0;
        if (have && P.lds_bytes + have * PK_LDS_ROW_BYTES + PK_LDS_COL_BYTES > lds_limit) have = 0;
        if (!have || PK_ROWS_PER_LANE + have < P.rows_max) {
            const int cap = have ? PK_ROWS_PER_LANE + have : (rows_cap > PK_ROWS_MAX && P.lds_bytes + PK_LDS_ROWS * PK_LDS_ROW_BYTES + PK_LDS_COL_BYTES <= lds_limit ? PK_ROWS_MAX : PK_ROWS_PER_LANE);
            P = pk_plan();
            pk_build_plan(NP, NT, tris, points, NE, edge_uv, he_edge, W, H, ratio, dp_px, max_parts, lds_limit, P, base_every, cap, vspeed);
            return;
        }
        P.lds_bytes = 0;
        for (auto& w : P.wg) { w.lds_rows = have; w.lds_bytes = pk_lds_bytes(w); P.lds_bytes = std::max(P.lds_bytes, w.lds_bytes); }
    }
    P.ok = true;
}
