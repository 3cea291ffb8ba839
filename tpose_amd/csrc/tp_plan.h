// tp_plan.h -- host side of the persistent grad-iter kernel (tp_persist.hip): the per-upload PLAN that cuts the mesh
// into patches, one per workgroup.
//
// What it replaces in the reference: nothing -- the reference issues two instanced draws and two dispatches per frame
// (software/triangulate/main.cpp:132-155) and lets the GL driver schedule them.  Here K grad-iters run inside ONE
// launch; a workgroup owns a compact patch of the mesh (some vertices, some undirected edges) for the whole launch and
// only what crosses a patch border travels between workgroups:
//   * vertex positions, from the vertex's owner to the owners of its edges and of its neighbours;
//   * line sums (tp_raster.h, "Edge-centric form") of border edges, from the edge's owner to the owners of the vertices
//     whose gradient they enter.
// Ownership is decided ONCE per upload from the topology and the upload-time positions (recursive coordinate
// bisection balanced by table look-ups), never by where a vertex has drifted to: there is nothing to re-bin while
// the descent runs, and results cannot depend on the cut (integer sums commute; every float operation is per vertex).
//
// Plain C++ (no HIP): tests/emul compiles this header with g++ and replays the kernel's lane functions
// (tp_persist.h) phase by phase on the CPU against the oracle.
#pragma once

#include <stdint.h>
#include <math.h>
#include <algorithm>
#include <string>
#include <unordered_map>
#include <vector>

#define PK_NLINES 9
#define PK_GRANULES 5      /* 8-byte granules of a line sum in the mailbox (tp_persist.h) */
#define PK_THREADS 1024
#define PK_ROWS_PER_LANE 8   /* table records a lane of the walk requests together */
#define PK_MAX_SLOTS 1023    /* position slots of a workgroup (10-bit fields of the corner records) */
#define PK_MAX_SUMS 65535    /* line-sum slots of a workgroup (16-bit fields) */
#define PK_MAX_OWN_EDGES 255 /* 8-bit field of the item records */
#define PK_MAX_TL 4095       /* chunks per line (12-bit field) */

// per-workgroup header; every `off_*` indexes pk_plan::pool (int32 units)
struct pk_wg {
    int32_t n_own_v, n_slots;  // position slots: the patch's own vertices first, then the foreign ones it reads
    int32_t n_own_e;           // own undirected edges: nine lines each, line-sum slots [0, 9 n_own_e)
    int32_t n_items;           // walk items (edge, chunk): nine lanes each
    int32_t n_corners;         // (own vertex, incident triangle): four lanes each, one per move
    int32_t n_imp, n_exp;      // line sums read from / written to the mailbox
    int32_t n_sums;            // 9 n_own_e + n_imp
    int32_t off_vid;           // [n_slots] global vertex id
    int32_t off_edges;         // [n_own_e] {slot_u | slot_v << 16, global edge id}
    int32_t off_items;         // [n_items] {own edge | chunk << 8 | chunks << 20, magic = floor(2^32 / chunks) + 1}
    int32_t off_corners;       // [n_corners] {t, s | slot_a << 12 | slot_b << 22 | own << 2, out | in << 16, opp}
    int32_t off_imp;           // [n_imp] global line id (edge * 9 + version); destination slot 9 n_own_e + k
    int32_t off_exp;           // [n_exp] own line-sum slot (edge_local * 9 + version)
    int32_t lds_bytes;         // dynamic LDS of this workgroup (pk_lds_bytes)
    int32_t pad;
};

struct pk_plan {
    bool ok = false;
    std::string why;           // when !ok: why the triangulation takes the two-kernel path instead
    int parts = 0;             // workgroups (grid size)
    int lds_bytes = 0;         // max over workgroups
    std::vector<pk_wg> wg;
    std::vector<int32_t> pool;
    std::vector<int32_t> owner_v, owner_e;  // (kept for tests and statistics)
    double work_max = 0.0, work_mean = 0.0; // table look-ups per workgroup at upload
    int64_t imp_total = 0, exp_total = 0;
};

// dynamic LDS carve (bytes), identical on the device (tp_persist.h: pk_carve)
#if defined(__HIPCC__)
#define PK_HD __host__ __device__ inline
#else
#define PK_HD inline
#endif
PK_HD int pk_align16(int v) { return (v + 15) & ~15; }
inline int pk_lds_bytes(const pk_wg& w) {
    int b = 0;
    b += pk_align16(w.n_sums * 48);                 // line sums: six 64-bit words
    b += pk_align16(9 * w.n_own_e * 24);            // walkers
    b += pk_align16(w.n_slots * 8);                 // positions
    b += pk_align16((4 * w.n_own_v + w.n_slots) * 8);  // snapped positions: foreign slots unmoved only, own slots + 4 moves
    b += pk_align16(w.n_own_e * 8);                 // row band of an edge's nine lines
    b += pk_align16(w.n_own_v * 8);                 // gradient
    b += pk_align16(w.n_slots * 4);                 // vid
    b += pk_align16(w.n_own_e * 8);                 // edges
    b += pk_align16(w.n_items * 8);                 // items
    b += pk_align16(w.n_corners * 16);              // corners
    b += pk_align16(w.n_imp * 4);                   // imports
    b += pk_align16(w.n_exp * 4);                   // exports
    return b + 64;                                  // flags
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

// Build the plan.  tris: ivec4[NT]; points: vec2[NP] (upload-time positions); edge_uv: int[2 NE] endpoint ids (the low
// 30 bits; tp_upload keeps flags above); he_edge: int[3 NT] edge * 2 + direction; W, H: raster; dp_px: a hint, the size
// of the moves in pixels (lines get a few rows longer or shorter).  max_parts: workgroups that can be resident at once.
inline void pk_build_plan(int NP, int NT, const int32_t* tris, const float* points, int NE, const int32_t* edge_uv,
                          const int32_t* he_edge, int W, int H, float ratio, float dp_px, int max_parts, int lds_limit,
                          pk_plan& P) {
    P = pk_plan();
    if (NT < 1 || NE < 1 || max_parts < 1) { P.why = "empty triangulation"; return; }
    auto EU = [&](int e) { return edge_uv[2 * (size_t)e] & 0x3fffffff; };
    auto EV = [&](int e) { return edge_uv[2 * (size_t)e + 1] & 0x3fffffff; };
    for (int e = 0; e < NE; e++)
        if (EU(e) == EV(e)) { P.why = "an edge names one vertex twice"; return; }

    // rows of every edge at upload: table look-ups per line
    std::vector<float> rows((size_t)NE);
    std::vector<double> wv((size_t)NP, 0.0);
    std::vector<int> deg((size_t)NP, 0);
    double total = 0.0;
    for (int e = 0; e < NE; e++) {
        const float ya = points[2 * (size_t)EU(e) + 1], yb = points[2 * (size_t)EV(e) + 1];
        float r = fabsf(ya - yb) * 0.5f * (float)H;
        if (!(r >= 0.0f)) r = 0.0f;                   // NaN positions: no rows
        r = std::min(r, (float)H) + 1.0f;
        rows[e] = r;
        const double w = (double)PK_NLINES * r;
        wv[EU(e)] += 0.5 * w; wv[EV(e)] += 0.5 * w;
        total += w;
    }
    for (int t = 0; t < NT; t++)
        for (int s = 0; s < 3; s++) { const int v = tris[4 * (size_t)t + s]; deg[v]++; wv[v] += 40.0; total += 40.0; }

    // workgroups: about PK_THREADS * PK_ROWS_PER_LANE look-ups each, at most one per edge
    int parts = (int)std::min<double>((double)max_parts, std::ceil(total / (double)(PK_THREADS * PK_ROWS_PER_LANE)));
    parts = std::max(1, std::min(parts, NE));
    if (parts >= 16) parts &= ~7;  // whole runs of patches per XCD (tp_persist.hip maps workgroup b to XCD b mod 8)

    // 1. vertices -> patches
    std::vector<pk_detail::rcb_vertex> a;
    a.reserve((size_t)NP);
    for (int v = 0; v < NP; v++) {
        if (!deg[v]) continue;
        float x = points[2 * (size_t)v] / ratio * 0.5f * (float)W, y = points[2 * (size_t)v + 1] * 0.5f * (float)H;
        if (!(x == x)) x = 0.0f;
        if (!(y == y)) y = 0.0f;
        a.push_back({v, x, y, wv[v]});
    }
    P.owner_v.assign((size_t)NP, -1);
    pk_detail::rcb(a, 0, (int)a.size(), 0, parts, P.owner_v);

    // 2. edges -> the lighter of the two patches at their ends
    std::vector<double> load((size_t)parts, 0.0);
    for (int v = 0; v < NP; v++) if (deg[v]) load[P.owner_v[v]] += 40.0 * deg[v];
    P.owner_e.assign((size_t)NE, 0);
    for (int e = 0; e < NE; e++) {
        const int pu = P.owner_v[EU(e)], pv = P.owner_v[EV(e)];
        const int p = load[pu] <= load[pv] ? pu : pv;
        P.owner_e[e] = p;
        load[p] += (double)PK_NLINES * rows[e];
    }
    for (int p = 0; p < parts; p++) { P.work_max = std::max(P.work_max, load[p]); P.work_mean += load[p] / parts; }

    // 3. per-patch lists
    std::vector<std::vector<int>> own_v((size_t)parts), own_e((size_t)parts);
    for (auto& q : a) own_v[P.owner_v[q.v]].push_back(q.v);   // (a is in bisection order: neighbours stay neighbours)
    for (int e = 0; e < NE; e++) own_e[P.owner_e[e]].push_back(e);
    std::vector<int> elocal((size_t)NE);
    for (int p = 0; p < parts; p++) {
        if ((int)own_e[p].size() > PK_MAX_OWN_EDGES) { P.why = "a patch owns more than 255 edges"; return; }
        for (size_t k = 0; k < own_e[p].size(); k++) elocal[own_e[p][k]] = (int)k;
    }
    // vertex -> corners (t, s), in triangle order
    std::vector<int> voff((size_t)NP + 1, 0), vadj((size_t)3 * NT);
    for (int t = 0; t < NT; t++) for (int s = 0; s < 3; s++) voff[tris[4 * (size_t)t + s] + 1]++;
    for (int v = 0; v < NP; v++) voff[v + 1] += voff[v];
    { std::vector<int> cur(voff.begin(), voff.end() - 1);
      for (int t = 0; t < NT; t++) for (int s = 0; s < 3; s++) vadj[cur[tris[4 * (size_t)t + s]]++] = 3 * t + s; }

    P.parts = parts;
    P.wg.assign((size_t)parts, pk_wg());
    std::vector<std::vector<int32_t>> exports((size_t)parts);   // own line-sum slots other patches read
    std::vector<std::vector<char>> exported((size_t)parts);
    for (int p = 0; p < parts; p++) exported[p].assign(own_e[p].size() * PK_NLINES, 0);

    struct built { std::vector<int32_t> vid, edges, items, corners, imp; };
    std::vector<built> B((size_t)parts);
    for (int p = 0; p < parts; p++) {
        pk_wg& w = P.wg[p];
        built& b = B[p];
        std::unordered_map<int, int> slot_of;  // vertex -> position slot
        auto slot = [&](int v) {
            auto it = slot_of.find(v);
            if (it != slot_of.end()) return it->second;
            const int s = (int)b.vid.size();
            slot_of.emplace(v, s); b.vid.push_back(v);
            return s;
        };
        for (int v : own_v[p]) slot(v);
        w.n_own_v = (int)own_v[p].size();
        // own edges and their walk items
        w.n_own_e = (int)own_e[p].size();
        for (int e : own_e[p]) {
            const int su = slot(EU(e)), sv = slot(EV(e));
            b.edges.push_back(su | (sv << 16)); b.edges.push_back(e);
        }
        for (int le = 0; le < w.n_own_e; le++) {
            const float r = rows[own_e[p][le]] + dp_px;
            int tl = (int)std::ceil(r / (float)PK_ROWS_PER_LANE);
            tl = std::max(1, std::min(tl, PK_MAX_TL));
            const uint32_t magic = tl == 1 ? 0u : (uint32_t)(0x100000000ull / (uint64_t)tl) + 1u;
            for (int c = 0; c < tl; c++) { b.items.push_back(le | (c << 8) | (tl << 20)); b.items.push_back((int32_t)magic); }
        }
        w.n_items = (int)(b.items.size() / 2);
        // corners of own vertices; line sums they need
        std::unordered_map<int, int> imp_slot;  // global line id -> line-sum slot
        // slot of line `ver` of edge e, followed by the slots of the next n - 1 versions (a foreign edge's lines are
        // imported in the groups the corners use: the base line alone, the four moves of one endpoint together)
        auto line_slot = [&](int e, int ver, int n) {
            const int q = P.owner_e[e];
            if (q == p) return elocal[e] * PK_NLINES + ver;
            const int gl = e * PK_NLINES + ver;
            auto it = imp_slot.find(gl);
            if (it != imp_slot.end()) return it->second;
            const int s = w.n_own_e * PK_NLINES + (int)b.imp.size();
            for (int k = 0; k < n; k++) {
                imp_slot.emplace(gl + k, s + k); b.imp.push_back(gl + k);
                const int ls = elocal[e] * PK_NLINES + ver + k;
                if (!exported[q][ls]) { exported[q][ls] = 1; exports[q].push_back(ls); }
            }
            return s;
        };
        for (int k = 0; k < w.n_own_v; k++) {
            const int v = own_v[p][k];
            for (int j = voff[v]; j < voff[v + 1]; j++) {
                const int h = vadj[j], t = h / 3, s = h - 3 * t;
                const int sn = s == 2 ? 0 : s + 1, sp = s == 0 ? 2 : s - 1;
                const int va = tris[4 * (size_t)t + sn], vb = tris[4 * (size_t)t + sp];
                const int he_out = he_edge[3 * (size_t)t + s], he_in = he_edge[3 * (size_t)t + sp], he_opp = he_edge[3 * (size_t)t + sn];
                // edge leaving the vertex: the vertex is its origin (tp_edge_version: flipped ? 4 + m : m);
                // edge arriving: the vertex is its destination (flipped ? m : 4 + m); four consecutive slots, moves 1..4
                const int vo = (he_out & 1) ? 4 : 0, vi = (he_in & 1) ? 0 : 4;
                const int so = line_slot(he_out >> 1, vo + 1, 4), si = line_slot(he_in >> 1, vi + 1, 4);
                const int sopp = line_slot(he_opp >> 1, 0, 1);
                b.corners.push_back(t);
                b.corners.push_back(s | (k << 2) | (slot(va) << 12) | (slot(vb) << 22));
                b.corners.push_back(so | (si << 16));
                b.corners.push_back(sopp);
            }
        }
        w.n_corners = (int)(b.corners.size() / 4);
        w.n_slots = (int)b.vid.size();
        w.n_imp = (int)b.imp.size();
        w.n_sums = w.n_own_e * PK_NLINES + w.n_imp;
        if (w.n_slots > PK_MAX_SLOTS || w.n_own_v > 1023) { P.why = "a patch reads more than 1023 vertices"; return; }
        if (w.n_sums > PK_MAX_SUMS) { P.why = "a patch needs more than 65535 line sums"; return; }
    }
    // 4. lay the pool out
    for (int p = 0; p < parts; p++) {
        pk_wg& w = P.wg[p];
        built& b = B[p];
        w.n_exp = (int)exports[p].size();
        auto put = [&](const std::vector<int32_t>& src) {
            const int off = (int)P.pool.size();
            P.pool.insert(P.pool.end(), src.begin(), src.end());
            while (P.pool.size() & 3) P.pool.push_back(0);  // 16-byte aligned tables
            return off;
        };
        w.off_vid = put(b.vid); w.off_edges = put(b.edges); w.off_items = put(b.items);
        w.off_corners = put(b.corners); w.off_imp = put(b.imp); w.off_exp = put(exports[p]);
        w.lds_bytes = pk_lds_bytes(w);
        P.lds_bytes = std::max(P.lds_bytes, w.lds_bytes);
        P.imp_total += w.n_imp; P.exp_total += w.n_exp;
    }
    if (P.lds_bytes > lds_limit) { P.why = "a patch does not fit the LDS"; return; }
    P.ok = true;
}
