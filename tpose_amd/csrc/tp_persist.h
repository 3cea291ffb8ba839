// tp_persist.h -- what ONE LANE of the persistent grad-iter kernel (tp_persist.hip) does in each of its phases.
//
// K grad-iters of the reference's frame -- doenergy (mode-1 draw of 13 NT triangles), gradient.cs, shift.cs
// (software/triangulate/main.cpp:132-155, shader/gradient.cs:19-36, shader/shift.cs:16-47; the warp program likewise) --
// run inside one launch.  A workgroup owns a patch of the mesh (tp_plan.h) and per grad-iter
//   P0  reads the positions of the foreign vertices it uses from the position mailbox (tagged granules),
//   P1  snaps positions (vertex stage, triangle.vs:59-84) and sets its own edges' nine lines up (tp_setup_line),
//   P3  walks the lines over the per-image row prefix table: line sums into LDS,
//   P4  writes the line sums other patches use to the line mailbox,  P5 reads the ones it uses,
//   P6  forms the four displaced variants of every corner (own vertex, incident triangle): moments = signed sum of
//       three line sums, energy (triangle.fs:37-43), central differences (gradient.cs) into the vertex's gradient,
//   P7  takes the shift.cs step of its own vertices and posts the new positions.
// The buffers the reference reads back (`tenergy`, `colnum`, `colacc`, `gradient`) are not produced here: the LAST
// grad-iter of a tp_iterate call runs through k_lines + k_update (tp_kernels.hip), which write them.
//
// Everything here is __host__ __device__: tests/emul replays the phases on the CPU, workgroup by workgroup, with the
// mailboxes replaced by plain copies, and compares with the oracle.  The shipped library never runs it on the host.
#pragma once

#include "tp_raster.h"
#include "tp_plan.h"

struct pk_f2 { float x, y; };
struct pk_i2 { int32_t x, y; };
struct pk_i4 { int32_t x, y, z, w; };
struct pk_u4 { uint32_t x, y, z, w; };
struct pk_walker { int64_t x, s; int32_t ra, rb; };  // tp_line

// the workgroup's LDS, carved in the order of pk_lds_bytes (tp_plan.h)
struct pk_view {
    unsigned long long* sums;  // [n_sums][6] line sums {sum x, n_odd, sum r, sum g, sum b, q}
    pk_walker* wk;             // [9 n_own_e]
    pk_f2* pos;                // [n_slots]
    pk_i2* snap;               // own slot k: [5 k + move]; foreign slot s: [5 n_own_v + s - n_own_v] (unmoved)
    pk_i2* band;               // [n_own_e] first and last row of an edge's nine lines
    pk_i2* grad;               // [n_own_v]
    int32_t* vid;              // static tables, copied from the plan's pool at the start of the launch
    pk_i2* edges;
    pk_i2* items;
    pk_i4* corners;
    int32_t* imp;
    int32_t* exp_;
    int32_t* flags;
};

TP_HD void pk_carve(char* base, const pk_wg& w, pk_view& V) {
    char* p = base;
    V.sums = (unsigned long long*)p; p += pk_align16(w.n_sums * 48);
    V.wk = (pk_walker*)p; p += pk_align16(9 * w.n_own_e * 24);
    V.pos = (pk_f2*)p; p += pk_align16(w.n_slots * 8);
    V.snap = (pk_i2*)p; p += pk_align16((4 * w.n_own_v + w.n_slots) * 8);
    V.band = (pk_i2*)p; p += pk_align16(w.n_own_e * 8);
    V.grad = (pk_i2*)p; p += pk_align16(w.n_own_v * 8);
    V.vid = (int32_t*)p; p += pk_align16(w.n_slots * 4);
    V.edges = (pk_i2*)p; p += pk_align16(w.n_own_e * 8);
    V.items = (pk_i2*)p; p += pk_align16(w.n_items * 8);
    V.corners = (pk_i4*)p; p += pk_align16(w.n_corners * 16);
    V.imp = (int32_t*)p; p += pk_align16(w.n_imp * 4);
    V.exp_ = (int32_t*)p; p += pk_align16(w.n_exp * 4);
    V.flags = (int32_t*)p;
}

TP_HD int pk_snap_index(const pk_wg& w, int slot, int move) {
    return slot < w.n_own_v ? 5 * slot + move : 5 * w.n_own_v + (slot - w.n_own_v);
}

// P1a, lane j < 5 n_own_v + (n_slots - n_own_v): snapped raster position of one (slot, move)
TP_HD void pk_snap_lane(const pk_wg& w, const pk_view& V, const tp_view& vw, int j) {
    const int own5 = 5 * w.n_own_v;
    const int slot = j < own5 ? j / 5 : w.n_own_v + (j - own5), move = j < own5 ? j - 5 * slot : 0;
    const pk_f2 p = V.pos[slot];
    int32_t X, Y;
    tp_vertex_stage(p.x, p.y, move, 0, vw, X, Y);
    V.snap[j].x = X; V.snap[j].y = Y;
}

// P1b, lane l < 9 n_own_e: line l = (own edge l / 9, version l % 9) -- the walker of the whole line.  The caller
// folds ra / rb into the edge's band (LDS atomics on the device).
TP_HD void pk_setup_lane(const pk_view& V, const tp_view& vw, int l, pk_walker& out) {
    const int le = l / PK_NLINES, q = l - le * PK_NLINES;
    const int su = V.edges[le].x & 0xffff, sv = (V.edges[le].x >> 16) & 0xffff;
    const pk_f2 pu = V.pos[su], pv = V.pos[sv];
    // line q: endpoint u displaced by move mu, endpoint v by move mv (tp_kernels.hip: k_lines)
    const int mu = (q >= 1 && q <= 4) ? q : 0, mv = q >= 5 ? q - 4 : 0;
    int32_t Xa, Ya, Xb, Yb;
    tp_vertex_stage(pu.x, pu.y, mu, 0, vw, Xa, Ya);
    tp_vertex_stage(pv.x, pv.y, mv, 0, vw, Xb, Yb);
    tp_line ln;
    tp_setup_line(Xa, Ya, Xb, Yb, vw.H, ln);
    out.x = ln.x; out.s = ln.s; out.ra = ln.ra; out.rb = ln.rb;
}

// x / d for the item's chunk count d (magic = floor(2^32 / d) + 1, exact for x d < 2^32; d == 1: magic 0)
TP_HD uint32_t pk_div(uint32_t x, uint32_t magic) {
#if defined(__HIP_DEVICE_COMPILE__)
    return magic ? __umulhi(x, magic) : x;
#else
    return magic ? (uint32_t)(((uint64_t)x * magic) >> 32) : x;
#endif
}

struct pk_acc {
    uint32_t xs, nodd;  // <= rows * W < 2^29
    uint64_t r, g, b, q;
};

// P3, lane-item j < 9 n_items: item j / 9 = (own edge, chunk c of TL), version j % 9.  The lane takes the rows
// first, first + TL, ... of its line, where first is the first row >= ra on the residue (rmin + c) mod TL of the
// edge's band: the nine lines of an edge sit in adjacent lanes ON THE SAME ROWS, and their crossing columns lie within
// a few pixels of each other, so their table records share a cache line or two.  `table`: the image's row prefix
// table (tp_raster.h), `pitch` records per row.  Returns the line-sum slot, the partial sums in `a`.
template <int BATCH>
TP_HD int pk_walk_lane(const pk_view& V, const char* table, int pitch, int W, int j, pk_acc& a) {
    const int item = j / PK_NLINES, q = j - item * PK_NLINES;
    const pk_i2 it = V.items[item];
    const int le = it.x & 0xff, c = (it.x >> 8) & 0xfff, TL = (it.x >> 20) & 0xfff;
    const uint32_t magic = (uint32_t)it.y;
    const int l = le * PK_NLINES + q;
    const pk_walker ln = V.wk[l];
    a.xs = 0; a.nodd = 0; a.r = 0; a.g = 0; a.b = 0; a.q = 0;
    if (ln.ra > ln.rb) return l;
    const int base = V.band[le].x + c;                                            // ra >= rmin: ra - base > -TL
    const int first = base + (int)pk_div((uint32_t)(ln.ra - base + TL - 1), magic) * TL;
    int n = ln.rb >= first ? (int)pk_div((uint32_t)(ln.rb - first), magic) + 1 : 0;
    int64_t x = ln.x + (int64_t)(first - ln.ra) * ln.s;
    const int64_t xs = (int64_t)((uint64_t)ln.s * (uint64_t)TL);                  // (unsigned: a steep two-row line may wrap, unused then)
    // (byte offsets into the table fit 32 bits: 16384 rows x 4100 records x 32 bytes < 2^32)
    uint32_t row = (uint32_t)(n > 0 ? first : 0) * (uint32_t)pitch * 32u;
    const uint32_t rs = (uint32_t)TL * (uint32_t)pitch * 32u;
    for (; n > 0; n -= BATCH) {
        pk_u4 d0[BATCH], d1[BATCH];
        int col[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; u++) {
            d0[u].x = d0[u].y = d0[u].z = d0[u].w = 0; d1[u] = d0[u]; col[u] = 0;
            if (u < n) {
                const int xc = (int)(x >> TP_LINE_FRAC);
                col[u] = xc < 0 ? 0 : (xc > W ? W : xc);
                const pk_u4* rec = reinterpret_cast<const pk_u4*>(table + (row + (((uint32_t)col[u] & ~3u) << 3)));
                d0[u] = rec[0]; d1[u] = rec[1];
                x = (int64_t)((uint64_t)x + (uint64_t)xs); row += rs;
            }
        }
        uint32_t sx = 0, so = 0, sr = 0, sg = 0, sb = 0;
#pragma unroll
        for (int u = 0; u < BATCH; u++) {
            const uint32_t rec[TP_PFX_WORDS] = {d0[u].x, d0[u].y, d0[u].z, d0[u].w, d1[u].x, d1[u].y, d1[u].z, d1[u].w};
            uint32_t no, ru, gu, bu, qu;
            tp_prefix_eval(rec, col[u], no, ru, gu, bu, qu);   // (an all-zero record at column 0 adds nothing)
            sx += (uint32_t)col[u]; so += no; sr += ru; sg += gu; sb += bu;   // BATCH <= 8 records: 8 x 2^22 fits
            a.q += qu;
        }
        a.xs += sx; a.nodd += so; a.r += sr; a.g += sg; a.b += sb;
    }
    return l;
}

// ---------------------------------------------------------------------------------------------------------------
// Line mailbox.  A line sum travels as five 8-byte granules {tag : 16, payload : 48}, each written by ONE store and
// valid on its own (the tag names the grad-iter): xs, n_odd <= 16384 * 16384 = 2^28 (29 bits), r, g, b < 2^36,
// q < 2^46.
//   g0 = q                       g1 = r | xs[0:12] << 36        g2 = g | xs[12:24] << 36
//   g3 = b | nodd[0:12] << 36    g4 = xs[24:29] | nodd[12:29] << 5
// ---------------------------------------------------------------------------------------------------------------
TP_HD void pk_pack_line(const unsigned long long s[6], uint32_t tag, unsigned long long g[PK_GRANULES]) {
    const unsigned long long T = (unsigned long long)(tag & 0xffffu) << 48;
    const unsigned long long xs = s[0], no = s[1];
    g[0] = T | s[5];
    g[1] = T | s[2] | ((xs & 0xfffull) << 36);
    g[2] = T | s[3] | (((xs >> 12) & 0xfffull) << 36);
    g[3] = T | s[4] | ((no & 0xfffull) << 36);
    g[4] = T | ((xs >> 24) & 0x1full) | ((no >> 12) << 5);
}
TP_HD bool pk_granule_ok(unsigned long long g, uint32_t tag) { return (uint32_t)(g >> 48) == (tag & 0xffffu); }
TP_HD void pk_unpack_line(const unsigned long long g[PK_GRANULES], unsigned long long s[6]) {
    const unsigned long long M36 = (1ull << 36) - 1ull;
    s[5] = g[0] & ((1ull << 48) - 1ull);
    s[2] = g[1] & M36; s[3] = g[2] & M36; s[4] = g[3] & M36;
    s[0] = ((g[1] >> 36) & 0xfffull) | (((g[2] >> 36) & 0xfffull) << 12) | ((g[4] & 0x1full) << 24);
    s[1] = ((g[3] >> 36) & 0xfffull) | (((g[4] >> 5) & 0x1ffffull) << 12);
}
// tag of grad-iter `epoch` (1 .. 32767 between two resets of the mailboxes): never 0, differs between e and e - 2
TP_HD uint32_t pk_tag(uint32_t epoch) { return 0x8000u | (epoch & 0x7fffu); }

// P6, lane (corner k, move m = 1..4): the energy of variant (t, 4 s + m) of the corner's triangle -- the corner's vertex
// displaced by move m.  Moments = signed sum of three line sums (tp_raster.h, "Edge-centric form"); energy as
// k_update's emit_variant.  col: the stored colour of the variant (warp flavour; triangle.fs:49-50).
TP_HD int32_t pk_corner_lane(const pk_wg& w, const pk_view& V, int k, int m, int flavour, pk_i4 col) {
    const pk_i4 cr = V.corners[k];
    const int s = cr.y & 3, own = (cr.y >> 2) & 0x3ff, sa = (cr.y >> 12) & 0x3ff, sb = (cr.y >> 22) & 0x3ff;
    const int sn = s == 2 ? 0 : s + 1, sp = s == 0 ? 2 : s - 1;
    int32_t X[3], Y[3], c[3];
    const pk_i2 pv = V.snap[pk_snap_index(w, own, m)], pa = V.snap[pk_snap_index(w, sa, 0)], pb = V.snap[pk_snap_index(w, sb, 0)];
    X[s] = pv.x; Y[s] = pv.y; X[sn] = pa.x; Y[sn] = pa.y; X[sp] = pb.x; Y[sp] = pb.y;
    tp_variant_coeffs(X, Y, c);
    // edge s leaves the vertex, edge sp arrives at it, edge sn is opposite
    const int cs = s == 0 ? c[0] : s == 1 ? c[1] : c[2];
    const int cn = sn == 0 ? c[0] : sn == 1 ? c[1] : c[2];
    const int cp = sp == 0 ? c[0] : sp == 1 ? c[1] : c[2];
    const unsigned long long* Sout = V.sums + (size_t)((cr.z & 0xffff) + m - 1) * 6;
    const unsigned long long* Sin = V.sums + (size_t)(((cr.z >> 16) & 0xffff) + m - 1) * 6;
    const unsigned long long* Sopp = V.sums + (size_t)(cr.w & 0xffff) * 6;
    int64_t mo[6];
#pragma unroll
    for (int q = 0; q < 6; q++) mo[q] = (int64_t)cs * (int64_t)Sout[q] + (int64_t)cp * (int64_t)Sin[q] + (int64_t)cn * (int64_t)Sopp[q];
    const tp_moments mm = {mo[0], mo[1], mo[2], mo[3], mo[4], mo[5]};
    const int64_t E = flavour == 0 ? tp_energy_triangulate(mm) : tp_energy64(mm, col.x, col.y, col.z);
    return tp_wrap32(E);
}

// P7, own vertex k with gradient (gx, gy): the shift.cs step (shift.cs:16-47).  Vertices 0..3 never move.
TP_HD pk_f2 pk_vertex_lane(pk_f2 p, int32_t gx, int32_t gy, int vid, float ratio, float rate) {
    if (vid < 4) return p;
    float tgx = (float)gx, tgy = (float)gy;
    float x = p.x, y = p.y;
    if (x <= -ratio) { x = -ratio; tgx = 0.0f; } else if (x >= ratio) { x = ratio; tgx = 0.0f; }
    if (y <= -1.0f) { y = -1.0f; tgy = 0.0f; } else if (y >= 1.0f) { y = 1.0f; tgy = 0.0f; }
    x = tp_fsub(x, tp_fdiv(tp_fdiv(tp_fmul(rate, tgx), 256.0f), 256.0f));
    y = tp_fsub(y, tp_fdiv(tp_fdiv(tp_fmul(rate, tgy), 256.0f), 256.0f));
    pk_f2 r; r.x = x; r.y = y;
    return r;
}
