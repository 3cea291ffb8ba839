// tp_persist.h -- what ONE LANE of the persistent grad-iter kernel (tp_persist.hip) does in each of its phases.
//
// K grad-iters of the reference's frame -- doenergy (mode-1 draw of 13 NT triangles), gradient.cs, shift.cs
// (software/triangulate/main.cpp:132-155, shader/gradient.cs:19-36, shader/shift.cs:16-47; the warp program likewise) --
// run inside one launch.  A workgroup owns a patch of the mesh (tp_plan.h: some vertices) and per grad-iter
//   P0  reads the positions of the neighbouring vertices it uses from the position mailbox (tagged granules),
//   P1  snaps positions (vertex stage, triangle.vs:59-84) and sets up every line its corners use (tp_setup_line),
//   P3  walks the lines over the per-image row prefix table (pixel records): line sums into LDS,
//   P6  forms the four displaced variants of every corner (own vertex, incident triangle): moments = signed sum of
//       three line sums, energy (triangle.fs:37-43), central differences (gradient.cs) into the vertex's gradient,
//   P7  takes the shift.cs step of its own vertices and posts the new positions.
// The buffers the reference reads back (`tenergy`, `colnum`, `colacc`, `gradient`) are written by the LAST grad-iter of a
// tp_iterate call only: that one also walks the base lines the base variants need (tp_plan.h) and stores every
// variant's outputs in the reference's layout.
//
// Everything here is __host__ __device__: tests/emul replays the phases on the CPU, workgroup by workgroup, with the
// mailboxes replaced by plain copies, and tests/test_emul.py compares the result with the CPU restatement of the reference.
// The shipped library never runs it on the host.
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
    unsigned long long* sums;  // [n_lines_all][PK_SUM_STRIDE] line sums (PK_SUM_WORDS of them used) {sum x | n_odd << 32, sum r | sum g << 32, sum b, q} (pk_fold_words)
    pk_walker* wk;             // [n_lines_all]
    pk_f2* pos;                // [n_slots]
    unsigned long long* gacc;  // [n_own_v][2] per own vertex and axis: {corners that have added their central difference this grad-iter : 32 (low),
                               // the int32 wrapping sum of those differences : 32 (high)} -- ONE returning 64-bit LDS atomic per corner and axis
                               // adds (difference << 32) | 1, and the lane whose returned count completes the vertex (vdeg) holds the whole sum:
                               // it takes that axis' step right there, no workgroup barrier between the corners and the steps
    int32_t* vdeg;             // [n_own_v] corners of the vertex
    float* spd;                // [n_own_v][2] sum over the launch's grad-iters of |step| of the vertex, per axis (t-pose units): what the planner weighs rows by
    int32_t* vid;              // static tables, copied from the plan's pool at the start of the launch
    int32_t* edges;
    int32_t* lines;
    int32_t* cut;              // [n_lines_all + 1] first of every line's lane-items that have NO thread of their own (the uncached ones: chunks nc .. tl - 1;
                               // lines walked every grad-iter first, then the last grad-iter's base lines); the last entry: how many there are
    int32_t* tl;               // [n_lines_all] chunks of every line, as last cut
    int32_t* nc;               // [n_lines_all] ... of which the chunks [0, nc) belong to a thread (a slot) that keeps their records
    int32_t* st;               // [PK_CACHED] while the lines are cut again: line | chunk << 16 handed to a free slot, -1 none
    int32_t* freel;            // [PK_CACHED] while the lines are cut again: the free slots
    int32_t* li;               // [li_cap][3] the first li_cap uncached lane-items: {line | chunk << 16, chunks, magic}
    pk_i4* corners;
    pk_i4* base;
    int32_t* ldir;             // [n_lines_all] +1 / -1 / 0: the line's second endpoint lies below / above / level with its first (snapped rows)
    int32_t* flags;
    // rows PK_ROWS_PER_LANE .. PK_ROWS_MAX - 1 of every slot's lane-item, when the plan has them (pk_wg::lds_rows; pk_walk_lds_rows)
    char* lrec;                // [PK_LDS_ROWS][PK_CACHED] table records (a wave's 64 slots side by side: they arrive by global_load ... lds)
    uint16_t* lcol;            // [PK_CACHED][PK_LDS_ROWS] the crossing column each belongs to (a slot's four in one 64-bit word)
};

TP_HD void pk_carve(char* base, const pk_wg& w, pk_view& V) {
    char* p = base;
    V.sums = (unsigned long long*)p; p += pk_align16(w.n_lines_all * 8 * PK_SUM_STRIDE);
    V.wk = (pk_walker*)p; p += pk_align16(w.n_lines_all * 24);
    V.pos = (pk_f2*)p; p += pk_align16(w.n_slots * 8);
    V.gacc = (unsigned long long*)p; p += pk_align16(w.n_own_v * 16);
    V.vdeg = (int32_t*)p; p += pk_align16(w.n_own_v * 4);
    V.spd = (float*)p; p += pk_align16(w.n_own_v * 8);
    V.vid = (int32_t*)p; p += pk_align16(w.n_slots * 4);
    V.edges = (int32_t*)p; p += pk_align16(w.n_edges * 4);
    V.lines = (int32_t*)p; p += pk_align16(w.n_lines_all * 4);
    V.cut = (int32_t*)p; p += pk_align16((w.n_lines_all + 1) * 4);
    V.tl = (int32_t*)p; p += pk_align16(w.n_lines_all * 4);
    V.nc = (int32_t*)p; p += pk_align16(w.n_lines_all * 4);
    V.st = (int32_t*)p; p += pk_align16(PK_CACHED * 4);
    V.freel = (int32_t*)p; p += pk_align16(PK_CACHED * 4);
    V.li = (int32_t*)p; p += pk_align16(w.li_cap * 12);
    V.corners = (pk_i4*)p; p += pk_align16(w.n_corners * 16);
    V.base = (pk_i4*)p; p += pk_align16(w.n_base * 16);
    V.ldir = (int32_t*)p; p += pk_align16(w.n_lines_all * 4);
    V.flags = (int32_t*)p; p += 64;
    V.lrec = p; p += (size_t)w.lds_rows * 16 * PK_CACHED;
    V.lcol = (uint16_t*)p; p += w.lds_rows ? (size_t)16 * PK_CACHED : 0;   // (eight 16-bit columns per slot: two 64-bit words of four)
}

// P1b, lane l < n_lines: line l = (local edge, version) -- the walker of the whole line
// ... its endpoints displaced by (dxu, dyu) and (dxv, dyv) t-pose units (a thread's first line: worked out once per launch)
// Returns which way the line runs down the raster: +1 / -1 / 0 as its second endpoint lies below / above / level with its first.
TP_HD int pk_setup_moved(const pk_view& V, const tp_view& vw, int su, int sv, float dxu, float dyu, float dxv, float dyv, pk_walker& out) {
    const pk_f2 pu = V.pos[su], pv = V.pos[sv];
    int32_t Xa, Ya, Xb, Yb;
    tp_vertex_stage_d(pu.x, pu.y, dxu, dyu, vw, Xa, Ya);
    tp_vertex_stage_d(pv.x, pv.y, dxv, dyv, vw, Xb, Yb);
    tp_line ln;
    tp_setup_line(Xa, Ya, Xb, Yb, vw.H, ln);
    out.x = ln.x; out.s = ln.s; out.ra = ln.ra; out.rb = ln.rb;
    return (Yb > Ya) - (Yb < Ya);
}
TP_HD int pk_setup_ends(const pk_view& V, const tp_view& vw, int su, int sv, int q, pk_walker& out) {
    // line q: endpoint u displaced by move mu, endpoint v by move mv (tp_kernels.hip: k_lines)
    const int mu = (q >= 1 && q <= 4) ? q : 0, mv = q >= 5 ? q - 4 : 0;
    return pk_setup_moved(V, vw, su, sv, tp_move_dx(mu, vw.dp), tp_move_dy(mu, vw.dp), tp_move_dx(mv, vw.dp), tp_move_dy(mv, vw.dp), out);
}
TP_HD int pk_setup_lane(const pk_view& V, const tp_view& vw, int l, pk_walker& out) {
    const int le = V.lines[l] & 0xffff, q = V.lines[l] >> 16;
    return pk_setup_ends(V, vw, V.edges[le] & 0xffff, (V.edges[le] >> 16) & 0xffff, q, out);
}

// x / d for the item's chunk count d (magic = floor(2^32 / d) + 1, exact for x d < 2^32; d == 1: magic 0)
TP_HD uint32_t pk_div(uint32_t x, uint32_t magic) {
#if defined(__HIP_DEVICE_COMPILE__)
    return magic ? __umulhi(x, magic) : x;
#else
    return magic ? (uint32_t)(((uint64_t)x * magic) >> 32) : x;
#endif
}

// a * b for a, b < 2^24 (full rate on this part; 32-bit integer multiplications run at a quarter of it)
TP_HD uint32_t pk_mul24(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul24(a, b);
#else
    return a * b;
#endif
}
TP_HD uint32_t pk_magic(int d) { return d == 1 ? 0u : (uint32_t)(4294967296.0 / (double)d) + 1u; }

// ---- The chunks of a patch's lines, and who walks them.  A line of r rows is walked by ceil((r + slack) / rows-per-lane) lanes (its chunks);
// lines grow and shrink while the descent runs, so the workgroup counts again every PK_RECUT grad-iters (and when a lane finds more rows
// than it keeps records for), from the walkers of the grad-iter at hand -- on the device, with nothing from the host.  A line keeps its
// chunks while it still fits them and has not shrunk by a lane's worth.  How lines are cut never changes a sum.
//
// Round 5: SLOTS.  A thread's cached lane-item is a slot; rounds 3-4 numbered lane-items by a prefix sum over the lines, so one line that
// gained a chunk moved the lane-items of every line behind it, every thread behind it fetched its 14 records again, and a cut that changed
// anything cost ~10 us (which is why cutting more often, or on demand, never paid).  Now a line OWNS slots: a cut frees the slots of the
// lines whose chunk count changed (their rows change lanes: those records are gone anyway) and hands free slots to the chunks that want
// one -- nobody else's lane-item moves.  tl[l] = chunks of line l, nc[l] = how many of them (chunks 0 .. nc - 1) have a slot; the others,
// and the base lines only the last grad-iter of a call walks, are the UNCACHED lane-items, numbered by a prefix sum (`cut`) and walked
// without kept records.  Four passes with workgroup barriers between them:
//   A  one wave, lane i takes the lines [i B, (i + 1) B):  what every line wants now -> scratch {want, changed} (the line's sum slot)
//   B  every slot:  a slot whose line changed gives its lane-item up; every free slot files itself in `freel`
//   C  the wave again:  free slots to the chunks that want one (prefix sum over the lines), `st[slot]` = line | chunk << 16; tl, nc, cut
//   D  every slot:  a free slot that was handed a lane-item takes it (records to be fetched); the table of uncached lane-items
// A patch whose lines want more slots than there are takes a row more per lane (everything is cut afresh then).
TP_HD int pk_recut_line(const pk_view& V, int l, int rpl, bool first) {
    const pk_walker& k = V.wk[l];
    const int rows = k.ra > k.rb ? 0 : k.rb - k.ra + 1;
    const int fresh = pk_chunks(rows, rpl);
    if (first) return fresh;
    const int old = V.tl[l];
    return (rows + 1 <= old * rpl && old <= fresh + 1) ? old : fresh;
}
// pass A.  Returns the chunks the lines walked every grad-iter want together.
TP_HD int pk_cut_want(const pk_view& V, int n, int n_every, int lane, int lanes, int rpl, bool first, int& changed) {
    const int B = (n + lanes - 1) / lanes;
    int every = 0;
    for (int l = lane * B; l < n && l < (lane + 1) * B; l++) {
        const int t = pk_recut_line(V, l, rpl, first);
        const bool ch = first || t != V.tl[l];
        changed |= ch ? 1 : 0;
        V.sums[PK_SUM_STRIDE * (size_t)l] = (unsigned long long)(uint32_t)t | (ch ? 1ull << 32 : 0ull);
        every += l < n_every ? t : 0;
    }
    return every;
}
// ... when no line changed: the scratch is the lines' sum slots, which the walk adds to
TP_HD void pk_cut_forget(const pk_view& V, int n, int lane, int lanes) {
    const int B = (n + lanes - 1) / lanes;
    for (int l = lane * B; l < n && l < (lane + 1) * B; l++) V.sums[PK_SUM_STRIDE * (size_t)l] = 0ull;
}
TP_HD bool pk_cut_line_changed(const pk_view& V, int l) { return ((V.sums[PK_SUM_STRIDE * (size_t)l] >> 32) & 1ull) != 0ull; }
// A changed line KEEPS the slots of the chunks it still has (chunk c < what it wants now: the slot stays where it is, only its rows change
// lanes -- pk_slot_release) and asks for the others; an unchanged line asks for the chunks it has no slot for.
TP_HD int pk_cut_line_need(const pk_view& V, int l) {
    if (!pk_cut_line_changed(V, l)) return V.tl[l] - V.nc[l];
    const int want = (int)(uint32_t)V.sums[PK_SUM_STRIDE * (size_t)l];
    return want - (V.nc[l] < want ? V.nc[l] : want);
}
// pass C, first half: the slots the lane's lines ask for
TP_HD int pk_cut_need(const pk_view& V, int n, int n_every, int lane, int lanes) {
    const int B = (n + lanes - 1) / lanes;
    int need = 0;
    for (int l = lane * B; l < n && l < (lane + 1) * B && l < n_every; l++) need += pk_cut_line_need(V, l);
    return need;
}
// EVERYTHING afresh (a launch without a carry; a row more per lane): slots are handed out CHUNK-MAJOR -- the chunks 0 of all lines, then
// the chunks 1, ... -- so that the lanes of a wave walk the same chunk of consecutive lines: the versions of one edge, whose rows are the
// same and whose crossing columns lie within a few pixels of each other.  Their table records then share cache lines (a row's records: 8 to
// a 128-byte line), and a wave's load touches a third of the lines it touches when its lanes sit lane-items apart -- which is what a patch
// whose vertices move fast is bound by (every row stale: one L1 look-up per lane and row; profiles/r05_experiments.txt).  And the lanes of
// one LDS atomic still fold different lines.  Level c: how many of the lane's lines want a chunk c ...
TP_HD int pk_cut_level_count(const pk_view& V, int n_every, int lane, int lanes, int n, int c) {
    const int B = (n + lanes - 1) / lanes;
    int k = 0;
    for (int l = lane * B; l < n && l < (lane + 1) * B && l < n_every; l++) k += (int)(uint32_t)V.sums[PK_SUM_STRIDE * (size_t)l] > c ? 1 : 0;
    return k;
}
// ... and their slots: first + 0, 1, ... while there are slots (every slot is free: slot = place)
TP_HD void pk_cut_level_assign(const pk_view& V, int n_every, int lane, int lanes, int n, int c, int first, int n_slots) {
    const int B = (n + lanes - 1) / lanes;
    for (int l = lane * B; l < n && l < (lane + 1) * B && l < n_every; l++)
        if ((int)(uint32_t)V.sums[PK_SUM_STRIDE * (size_t)l] > c) {
            if (first < n_slots) { V.st[first] = l | (c << 16); V.nc[l] = c + 1; }
            first++;
        }
}
// ... and then tl, and the uncached chunks of the lane's lines (pk_cut_write numbers them)
TP_HD int pk_cut_fresh_done(const pk_view& V, int n, int lane, int lanes) {
    const int B = (n + lanes - 1) / lanes;
    int unc = 0;
    for (int l = lane * B; l < n && l < (lane + 1) * B; l++) {
        V.tl[l] = (int)(uint32_t)V.sums[PK_SUM_STRIDE * (size_t)l];
        V.sums[PK_SUM_STRIDE * (size_t)l] = 0ull;
        unc += V.tl[l] - V.nc[l];
    }
    return unc;
}
TP_HD void pk_cut_fresh_begin(const pk_view& V, int n, int lane, int lanes) {
    const int B = (n + lanes - 1) / lanes;
    for (int l = lane * B; l < n && l < (lane + 1) * B; l++) V.nc[l] = 0;
}
// ... second half: `base` = slots asked for by the lanes before this one, n_free = free slots filed in pass B.  Returns the lane's uncached chunks.
TP_HD int pk_cut_alloc(const pk_view& V, int n, int n_every, int lane, int lanes, int base, int n_free) {
    const int B = (n + lanes - 1) / lanes;
    int unc = 0;
    for (int l = lane * B; l < n && l < (lane + 1) * B; l++) {
        const int want = (int)(uint32_t)V.sums[PK_SUM_STRIDE * (size_t)l];
        if (pk_cut_line_changed(V, l)) { V.tl[l] = want; V.nc[l] = V.nc[l] < want ? V.nc[l] : want; }   // (the chunks below both counts kept their slots)
        V.sums[PK_SUM_STRIDE * (size_t)l] = 0ull;
        if (l < n_every) {
            const int need = V.tl[l] - V.nc[l];
            int k = n_free - base;
            k = k < 0 ? 0 : (k > need ? need : k);
            for (int i = 0; i < k; i++) V.st[V.freel[base + i]] = l | ((V.nc[l] + i) << 16);
            V.nc[l] += k;
            base += need;
        }
        unc += V.tl[l] - V.nc[l];
    }
    return unc;
}
// ... and the prefix sum of the uncached chunks: `offset` = those of the lanes before this one
TP_HD void pk_cut_write(const pk_view& V, int n, int lane, int lanes, int offset) {
    const int B = (n + lanes - 1) / lanes;
    for (int l = lane * B; l < n && l < (lane + 1) * B; l++) { V.cut[l] = offset; offset += V.tl[l] - V.nc[l]; }
    if (lane == lanes - 1) V.cut[n] = offset;   // (the last lane's lines are the last ones, or it has none and its offset is the total)
}
// after a cut, lane l < n_lines_all: the line's uncached lane-items into the table
TP_HD void pk_list_line(const pk_view& V, int l, int li_cap) {
    const int j0 = V.cut[l], j1 = V.cut[l + 1];
    if (j1 <= j0 || j0 >= li_cap) return;
    const int TL = V.tl[l], c0 = TL - (j1 - j0);
    const int32_t magic = (int32_t)pk_magic(TL);
    for (int j = j0; j < j1 && j < li_cap; j++) {
        int32_t* e = V.li + 3 * (size_t)j;
        e[0] = l | ((c0 + j - j0) << 16); e[1] = TL; e[2] = magic;
    }
}
// uncached lane-item j -> (line l, chunk c of TL): the line whose run of uncached lane-items holds j
TP_HD void pk_find_item(const pk_view& V, int n_lines_all, int j, int& l, int& c, int& TL) {
    int lo = 0, hi = n_lines_all;   // cut[lo] <= j < cut[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (V.cut[mid] <= j) lo = mid; else hi = mid;
    }
    l = lo; TL = V.tl[lo]; c = TL - (V.cut[lo + 1] - V.cut[lo]) + (j - V.cut[lo]);
}

struct pk_acc {
    uint32_t xs, nodd;  // <= rows * W < 2^29
    uint64_t r, g, b, q;
};

struct pk_rec { uint64_t lo, hi; };   // one pixel record (tp_raster.h, "Pixel records")

#if defined(PK_DBG_STALE) && defined(__HIPCC__)
// Counting flavour (tools/stale_counts.py; never in the product): per workgroup, summed over a launch's grad-iters -- {rows of cached lane-items whose record
// was fetched again, wave-loads that fetched them (rows x waves with at least one such lane), lane-items whose first row moved, rows walked}
static __device__ unsigned long long g_pk_cnt[512 * 4];
// ... and per vertex {rows fetched again, rows walked} of the lines displaced at the vertex (a base line: at its endpoint of the lower slot)
#define PK_DBG_VCNT 65536
static __device__ unsigned long long g_pk_vcnt[2 * PK_DBG_VCNT];
// ... and the crossing column every (workgroup, thread, row) had BEFORE its current one: how many of the rows fetched again go back to it (a second record
// per row would have had them) -- counted in g_pk_cnt[4 b + 1] in place of the wave-loads when PK_DBG_VICTIM is defined
static __device__ int g_pk_prevcol[256 * PK_THREADS * 16];
#endif
// a record of the table at byte offset `off`.  (-DTPOSE_DEBUG -DPK_DBG_BOUNDS flavour of the library only -- tools/hostile_repro.py: offsets beyond the table are counted, the
// first one is kept -- g_pk_fault = {table bytes, faults, offset, block | thread << 32} -- and the load is not made.)

#if defined(PK_DBG_BOUNDS) && defined(__HIPCC__)
static __device__ unsigned long long g_pk_fault[16];   // (one per translation unit; the kernel's and its reader are both in tp_persist.hip)
#endif
#if defined(PK_DBG_BOUNDS) && defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ pk_rec pk_load_rec(const char* table, uint32_t off) {
    if ((unsigned long long)off + 16ull > g_pk_fault[0]) {
        if (atomicAdd(&g_pk_fault[1], 1ull) == 0ull) { g_pk_fault[2] = off; g_pk_fault[3] = (unsigned long long)blockIdx.x | ((unsigned long long)threadIdx.x << 32); }
        pk_rec z; z.lo = 0; z.hi = 0; return z;
    }
    return *reinterpret_cast<const pk_rec*>(table + off);
}
#else
TP_HD pk_rec pk_load_rec(const char* table, uint32_t off) { return *reinterpret_cast<const pk_rec*>(table + off); }
#endif

// the sum of up to TP_PX_MAXSUM records into the partial sums of a lane
TP_HD void pk_add_unpacked(uint64_t lo, uint64_t hi, pk_acc& a) {
    uint32_t no, r, g, b; uint64_t q;
    tp_px_unpack(lo, hi, no, r, g, b, q);
    a.nodd += no; a.r += r; a.g += g; a.b += b; a.q += q;
}

// The rows of one lane: chunk c of TL takes the rows r = c (mod TL) of its line -- first, first + TL, ... (n of them).  The
// lines of an edge have the same chunks: adjacent lanes work ON THE SAME ROWS, and their crossing columns lie within a few
// pixels of each other, so what they fetch shares cache lines.  Residues are absolute (not counted from the line's first
// row): when an endpoint crosses a pixel row, one lane of the line gains or loses a row and the others keep theirs.
#ifndef PK_CONTIG_ALL
#define PK_CONTIG_ALL 0   /* 1 (experiment): chunk c of TL takes a BLOCK of consecutive rows instead of a residue class */
#endif
TP_HD int pk_row_step(int TL) { return PK_CONTIG_ALL ? 1 : TL; }   // image rows between two rows of a lane
struct pk_rows { int n; int64_t x, xs; uint32_t row, rs; };
TP_HD pk_rows pk_lane_rows(const pk_walker& ln, int c, int TL, uint32_t magic, int pitch, int* first_row = nullptr) {
    pk_rows r; r.n = 0; r.x = 0; r.xs = 0; r.row = 0; r.rs = 0;
    if (first_row) *first_row = 0;
    if (ln.ra > ln.rb) return r;
#if PK_CONTIG_ALL
    {
        const int rows = ln.rb - ln.ra + 1;
        const int B = (int)pk_div((uint32_t)(rows + TL - 1), magic);   // ceil(rows / TL) rows per chunk
        const int first = ln.ra + c * B;
        if (ln.rb < first) return r;
        r.n = ln.rb - first + 1 < B ? ln.rb - first + 1 : B;
        r.x = (int64_t)((uint64_t)ln.x + (uint64_t)ln.s * (uint64_t)(uint32_t)(c * B));
        r.xs = ln.s;
        r.row = pk_mul24((uint32_t)first, (uint32_t)pitch * 16u);
        r.rs = (uint32_t)pitch * 16u;
        if (first_row) *first_row = first;
        return r;
    }
#endif
    int d = c - (ln.ra - (int)pk_mul24(pk_div((uint32_t)ln.ra, magic), (uint32_t)TL));   // c - ra mod TL (rows and chunks < 2^13)
    d += d < 0 ? TL : 0;
    const int first = ln.ra + d;
    if (ln.rb < first) return r;
    r.n = (int)pk_div((uint32_t)(ln.rb - first), magic) + 1;
    // (d and TL are not negative: 64 x 32-bit products -- two quarter-rate multiplications each instead of three -- the same bits modulo 2^64)
    r.x = (int64_t)((uint64_t)ln.x + (uint64_t)ln.s * (uint64_t)(uint32_t)d);
    r.xs = (int64_t)((uint64_t)ln.s * (uint64_t)(uint32_t)TL);                    // (unsigned: a steep two-row line may wrap, unused then)
    // (byte offsets into the table fit 32 bits: 4096 rows x 4104 records x 16 bytes < 2^29; rows, chunks < 2^13 and a row's bytes < 2^17)
    r.row = pk_mul24((uint32_t)first, (uint32_t)pitch * 16u);
    r.rs = pk_mul24((uint32_t)TL, (uint32_t)pitch * 16u);
    if (first_row) *first_row = first;
    return r;
}
// crossing column of the current row, clamped to [0, W]; then one row on
TP_HD int pk_next_col(pk_rows& r, int W) {
    const int xc = (int32_t)((uint64_t)r.x >> 32) >> (TP_LINE_FRAC - 32);   // x >> 40: the high word, shifted arithmetically
    r.x = (int64_t)((uint64_t)r.x + (uint64_t)r.xs);
    return tp_clamp0(xc, W);
}

// rows [0, n) of a lane without a register cache, B records requested together
template <int B>
TP_HD void pk_walk_rows(pk_rows& r, const char* table, int W, pk_acc& a) {
    static_assert(B <= TP_PX_MAXSUM, "records added before unpacking");
    for (; r.n > 0; r.n -= B) {
        pk_rec d[B];
        uint32_t sx = 0;
#pragma unroll
        for (int u = 0; u < B; u++) {
            d[u].lo = 0; d[u].hi = 0;
            if (u < r.n) {
                const uint32_t col = (uint32_t)pk_next_col(r, W);
                sx += col;
#if defined(PK_DBG_BOUNDS) && defined(__HIP_DEVICE_COMPILE__)
                if ((unsigned long long)(r.row + (col << 4)) + 16ull > g_pk_fault[0] && atomicAdd(&g_pk_fault[14], 1ull) == 0ull)
                    g_pk_fault[15] = (unsigned long long)r.row | ((unsigned long long)(uint32_t)r.n << 32);
#endif
                d[u] = pk_load_rec(table, r.row + (col << 4));
                r.row += r.rs;
            }
        }
        uint64_t lo = 0, hi = 0;
#pragma unroll
        for (int u = 0; u < B; u++) { lo += d[u].lo; hi += d[u].hi; }
        a.xs += sx;
        pk_add_unpacked(lo, hi, a);
    }
}

// the same from the TILED copy of the table (tp_raster.h): rows first, first + TL, ...
template <int B>
TP_HD void pk_walk_rows_tiled(pk_rows& r, uint32_t row, uint32_t TL, uint32_t pitch, const char* tiled, int W, pk_acc& a) {
    static_assert(B <= TP_PX_MAXSUM, "records added before unpacking");
    for (; r.n > 0; r.n -= B) {
        pk_rec d[B];
        uint32_t sx = 0;
#pragma unroll
        for (int u = 0; u < B; u++) {
            d[u].lo = 0; d[u].hi = 0;
            if (u < r.n) {
                const uint32_t col = (uint32_t)pk_next_col(r, W);
                sx += col;
                d[u] = pk_load_rec(tiled, tp_px_tiled_row_part(row, pitch) + tp_px_tiled_col_part(col));
                row += TL;
            }
        }
        uint64_t lo = 0, hi = 0;
#pragma unroll
        for (int u = 0; u < B; u++) { lo += d[u].lo; hi += d[u].hi; }
        a.xs += sx;
        pk_add_unpacked(lo, hi, a);
    }
}

#ifndef PK_UNCACHED_BATCH
#define PK_UNCACHED_BATCH 4    /* records requested together by a lane-item without cached records (8 measured the same at 4096^2 in round 4, and its 32 registers in flight
                                  are what decides whether the kernel fits three waves per SIMD with 16 rows per lane) */
#endif
// P3, lane-item j >= PK_CACHED (a patch with more lane-items than its threads keep records for): (line l, chunk c of TL), nothing kept between
// grad-iters.  Returns the line-sum slot, the partial sums in `a`.
// tiled: the tiled copy of the table, or null (then `table`, row-major)
// part / parts: this call takes the rows [part B, (part + 1) B) of the lane-item, B = PK_UNCACHED_BATCH (the last part: all that is left) -- a
// lane-item of 14 rows walked by ONE lane is four memory latencies in a row; its four parts on four lanes are one (the kernel has idle lanes
// whenever a patch has few such lane-items: a mesh a hundred thousand grad-iters old, whose lines have grown beyond what the threads keep)
TP_HD int pk_walk_lane(const pk_view& V, const char* table, const char* tiled, int pitch, int W, int n_lines_all, int base, int li_cap, int j, pk_acc& a,
                       int part = 0, int parts = 1) {
    int l, c, TL;
    uint32_t magic;
#ifdef PK_EXP_NOLI
    if (false) {
#else
    if (j - base < li_cap) {
#endif
        const int32_t* e = V.li + 3 * (size_t)(j - base);
        l = e[0] & 0xffff; c = e[0] >> 16; TL = e[1]; magic = (uint32_t)e[2];
    } else {   // (the patch's lines have outgrown the table)
        pk_find_item(V, n_lines_all, j, l, c, TL);
        magic = pk_magic(TL);
    }
    a.xs = 0; a.nodd = 0; a.r = 0; a.g = 0; a.b = 0; a.q = 0;
    int first;
    pk_rows r = pk_lane_rows(V.wk[l], c, TL, magic, pitch, &first);
    if (parts > 1) {
        const int skip = part * PK_UNCACHED_BATCH;
        if (skip >= r.n) return l;   // (nothing left for this part)
        r.n -= skip; r.x = (int64_t)((uint64_t)r.x + (uint64_t)skip * (uint64_t)r.xs); r.row += (uint32_t)skip * r.rs; first += skip * pk_row_step(TL);
        if (part + 1 < parts && r.n > PK_UNCACHED_BATCH) r.n = PK_UNCACHED_BATCH;
    }
    if (tiled) pk_walk_rows_tiled<PK_UNCACHED_BATCH>(r, (uint32_t)first, (uint32_t)pk_row_step(TL), (uint32_t)pitch, tiled, W, a);
    else pk_walk_rows<PK_UNCACHED_BATCH>(r, table, W, a);
    return l;
}

// P3, lane-item j < PK_CACHED: one of thread (j mod PK_THREADS)'s own lane-items for the whole launch.  A lane walks the
// same rows of the same line every grad-iter and vertices move by a fraction of a pixel, so the record it needs for a row
// is usually the one it used a grad-iter ago: the records of its first R rows stay in registers (`rec`, with the crossing
// column they belong to in `col`; `row0` is the table offset of the first row -- when the line's first row changes,
// everything is fetched again) and a row is only loaded again when its crossing column has changed.  What is summed is
// always the record of the current crossing column -- the cache changes the traffic, never the values.
// Rows beyond the line's end count as column 0, whose record holds the moments of no pixels and adds nothing: no
// lane-dependent branches in either loop.
template <int R>
struct pk_lane_cache {
    int l, c, TL;           // the lane-item: line-sum slot, chunk, chunks (until the patch's lines are cut again)
    uint32_t magic;
    uint32_t row0;          // table offset of the first row the cached records belong to; ~0: nothing cached
    int32_t col[R];         // crossing column of rec[u]; -1: nothing cached
    pk_rec rec[R];
};
// A slot (a thread's cached lane-item) through a cut of the lines -- passes B and D above.
// where a slot stands in the free list when everything is cut afresh the OTHER way (PK_CHUNK_MAJOR 0: a line's chunks on lanes 64 / (slots / 64) apart)
#ifndef PK_PLACE_IDENTITY
#define PK_PLACE_IDENTITY 0   /* 1 (experiment): a line's chunks on ADJACENT lanes -- consecutive rows of one line in one wave-load */
#endif
#ifndef PK_STALE_TILED
#define PK_STALE_TILED 0      /* 1 (experiment): rows whose crossing column has changed are fetched from the tiled copy of the table */
#endif
TP_HD int pk_place_of_slot(int s) { return PK_PLACE_IDENTITY ? s : (s & 63) * (PK_CACHED / 64) + (s >> 6); }
// (A slot's records and columns are never touched here: a slot that changes hands gets a first row that matches nothing -- row0 = ~0 -- and
// the walk's own "the line's first row moved" path drops its columns and fetches; a slot without a lane-item has no rows, its columns
// read 0 and its records the table's all-zero record.  Loops over a slot's 16 rows in three more places cost the kernel 160 registers.)
template <int R>
TP_HD void pk_slot_clear(pk_lane_cache<R>& C) { C.l = 0; C.c = 0; C.TL = 0; C.magic = 0u; C.row0 = 0xffffffffu; }
// pass B: true when the slot is free after it (it had no lane-item, or its line is cut differently now)
// (afresh: every slot is given up; otherwise a slot whose chunk its line still has stays that chunk's, with the line's new count)
template <int R>
TP_HD bool pk_slot_release(pk_lane_cache<R>& C, const pk_view& V, bool afresh) {
    if (afresh) pk_slot_clear(C);
    else if (C.TL != 0 && pk_cut_line_changed(V, C.l)) {
        const int want = (int)(uint32_t)V.sums[PK_SUM_STRIDE * (size_t)C.l];
        if (C.c < want) { C.TL = want; C.magic = pk_magic(want); C.row0 = 0xffffffffu; }
        else pk_slot_clear(C);
    }
    return C.TL == 0;
}
// pass D
template <int R>
TP_HD void pk_slot_take(pk_lane_cache<R>& C, const pk_view& V, int s) {
    const int e = V.st[s];
    if (e < 0) return;
    V.st[s] = -1;
    C.l = e & 0xffff; C.c = e >> 16; C.TL = V.tl[C.l]; C.magic = pk_magic(C.TL);
    C.row0 = 0xffffffffu;
}
// The walk of a cached lane-item in two steps, so that a thread can run each step for all its lane-items before the next
// (the fetches of all of them are in flight while the sums begin).  RR <= R: how many of the R cached rows the lanes of this
// workgroup use (the plan's rows per lane, rounded up to one of the instantiated values: the loops are straight-line code,
// rows a lane does not have cost what the others cost).
// step 1: every row's crossing column, once; a row whose column has left its cached record is fetched right there (loads
// are issued, not waited for).  Returns the rows of the lane.  (Round 3 began with a scan that only compared, and a second
// walk that fetched when anything in the WAVE was stale -- which is nearly always: one pass is 0.3 us per grad-iter less.)
// rows RR .. RR + RL - 1 of slot s's lane-item: their records live in LDS (round 5), everything else as for the rows in registers -- what is
// summed is the record of the row's current crossing column, a row is fetched when that column has changed (`moved`: all of them).  t: the
// walk behind row RR - 1.  Stale records are requested straight into LDS (global_load ... lds: a wave's 64 slots side by side).
TP_HD void pk_lds_fetch(const char* src, char* wave_run, int lane) {
#if defined(__HIP_DEVICE_COMPILE__)
    (void)lane;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)wave_run, 16, 0, 0);
#else
    memcpy(wave_run + 16 * (size_t)lane, src, 16);
#endif
}
// (round 6) A plan's instantiation follows its LARGEST patch: on a photograph one patch in ten takes 17-20 rows per lane and every workgroup ran the
// rows beyond the registers -- 60 instructions of the pass, 16 of the sums, 0.35 us -- for lanes that have no such rows.  A WAVE none of whose lanes
// has more than RR rows skips them -- and one none of whose lanes has more than RR + 4 the second four of eight -- (device only: the CPU replay
// keeps checking every slot), and marks its slots' columns in LDS as belonging to no row (0xffff: no raster has that many columns), so that
// whatever the slot finds there when a lane-item of more rows arrives is fetched again.  h: which four of the rows
template <int RR>
TP_HD bool pk_lds_rows_idle(int n, int h = 0) {
#if defined(__HIP_DEVICE_COMPILE__)
    return !__any(n > RR + 4 * h);
#else
    (void)n; (void)h; return false;
#endif
}
template <int RR, int RL>
TP_HD void pk_walk_lds_rows(const pk_view& V, int s, int n, pk_rows& t, uint32_t live, bool moved, const char* table, int W) {
    static_assert(RL == 0 || RL == 2 || RL == 4 || RL == 8, "a slot's crossing columns in LDS are two 64-bit words of four");
    if (RL == 0) return;
    char* const run = V.lrec + 16 * (size_t)(s & ~63);
#pragma unroll
    for (int h = 0; h < (RL + 3) / 4; h++) {
        uint64_t* const word = reinterpret_cast<uint64_t*>(V.lcol) + 2 * (size_t)s + h;
        if (pk_lds_rows_idle<RR>(n, h)) { *word = ~0ull; continue; }   // (idle at h: idle beyond)
        uint64_t pc = *word;
#pragma unroll
        for (int v = 0; v < 4 && 4 * h + v < RL; v++) {
            const int u = 4 * h + v;
            const uint32_t on = 0u - ((live >> (RR + u)) & 1u);
            const uint32_t col = (uint32_t)pk_next_col(t, W) & on;
            if (moved || col != (uint32_t)((pc >> (16 * v)) & 0xffffu)) {
                pk_lds_fetch(table + (((t.row + (uint32_t)(RR + u) * t.rs) & on) + (col << 4)), run + 16 * (size_t)u * PK_CACHED, s & 63);
                pc = (pc & ~(0xffffull << (16 * v))) | ((uint64_t)col << (16 * v));
            }
        }
        *word = pc;
    }
}
template <int RR, int RL, int R>
TP_HD int pk_walk_pass(pk_lane_cache<R>& C, const pk_view& V, int s, int pitch, const char* table, int W, const char* tiled = nullptr, bool hot = false) {
    static_assert(RR <= R && RR + RL <= 32, "one bit per row");
    pk_rows t;
    int first = 0;
    if (C.TL == 0) { t.n = 0; t.x = 0; t.xs = 0; t.row = 0; t.rs = 0; }
    else t = pk_lane_rows(V.wk[C.l], C.c, C.TL, C.magic, pitch, &first);
    const uint32_t live = t.n >= 32 ? 0xffffffffu : ((1u << t.n) - 1u);   // bit u: row u exists
    const bool moved = t.row != C.row0;   // another first row: every record is another row's (an endpoint crossed a pixel row)
#if defined(__HIP_DEVICE_COMPILE__)
    if (__any(moved)) {   // (one answer per wave, and kept a branch: as selects the rare case would cost every row an instruction)
        asm volatile("" ::: "memory");
#else
    {
#endif
        if (moved) {
            C.row0 = t.row;
#pragma unroll
            for (int u = 0; u < RR; u++) C.col[u] = -1;
        }
    }
    const int n = t.n;
#if defined(PK_DBG_STALE) && defined(__HIP_DEVICE_COMPILE__)
    unsigned dbg_stale = 0, dbg_loads = 0, dbg_live = 0;
#endif
#pragma unroll
    for (int u = 0; u < RR; u++) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PK_NO_PRIO)
        // A wave that is behind its SIMD's other waves is issued first (round 5).  The hardware's order is oldest first: of the three waves
        // of a SIMD the youngest got what the others left and finished its pass 0.4 us behind them, alone on the SIMD, at the latency of
        // its own chain instead of the SIMD's issue rate (per-wave stamps).  The priority falls as a wave gets on with its rows, so the
        // three stay together: 4.37-4.40 -> 4.26-4.32 us per grad-iter, A/B on one box, four alternations.
        if (u == 0) __builtin_amdgcn_s_setprio(3);
        else if (u == RR / 3) __builtin_amdgcn_s_setprio(2);
        else if (u == 2 * (RR / 3)) __builtin_amdgcn_s_setprio(1);
#endif
        const uint32_t on = 0u - ((live >> u) & 1u);
        const int32_t col = pk_next_col(t, W) & (int32_t)on;
#if defined(PK_DBG_STALE) && defined(__HIP_DEVICE_COMPILE__)
        dbg_stale += col != C.col[u]; dbg_live += (col != C.col[u]) && on;
#if defined(PK_DBG_VICTIM)
        if (RR <= 16 && blockIdx.x < 256 && col != C.col[u]) {
            int* pc = &g_pk_prevcol[((size_t)blockIdx.x * PK_THREADS + threadIdx.x) * 16 + u];
            dbg_loads += (on && !moved && *pc == col && col != 0) ? 1u : 0u;   // (lane-local count here: summed by every lane below)
            *pc = moved ? -2 : C.col[u];
        }
#else
        dbg_loads += __any(col != C.col[u]) ? 1u : 0u;
#endif
#endif
        if (col != C.col[u]) {   // (a row beyond the line's end: the record of row 0, column 0)
#if !defined(PK_EXP_NOLOAD)   // timing experiments only (tools/build_variants.py); never defined in the product
#if defined(PK_DBG_BOUNDS) && defined(__HIP_DEVICE_COMPILE__)
            if ((unsigned long long)(((t.row + (uint32_t)u * t.rs) & on) + ((uint32_t)col << 4)) + 16ull > g_pk_fault[0] && atomicAdd(&g_pk_fault[4], 1ull) == 0ull) {
                g_pk_fault[5] = t.row; g_pk_fault[6] = (unsigned long long)u | ((unsigned long long)t.rs << 32); g_pk_fault[7] = (unsigned long long)(uint32_t)col | ((unsigned long long)(uint32_t)n << 32);
                g_pk_fault[8] = (unsigned long long)(uint32_t)C.l | ((unsigned long long)(uint32_t)C.c << 32); g_pk_fault[9] = (unsigned long long)(uint32_t)C.TL | ((unsigned long long)C.magic << 32);
                g_pk_fault[10] = (unsigned long long)(uint32_t)V.wk[C.l].ra | ((unsigned long long)(uint32_t)V.wk[C.l].rb << 32); g_pk_fault[11] = (unsigned long long)V.wk[C.l].x; g_pk_fault[12] = (unsigned long long)V.wk[C.l].s;
                g_pk_fault[13] = (unsigned long long)blockIdx.x | ((unsigned long long)threadIdx.x << 32);
            }
#endif
            if ((PK_STALE_TILED || hot) && tiled)   // (hot: one answer per workgroup -- a scalar branch)
                C.rec[u] = pk_load_rec(tiled, tp_px_tiled_row_part(((uint32_t)first + (uint32_t)u * (uint32_t)pk_row_step(C.TL)) & on, (uint32_t)pitch) + tp_px_tiled_col_part((uint32_t)col));
            else
                C.rec[u] = pk_load_rec(table, ((t.row + (uint32_t)u * t.rs) & on) + ((uint32_t)col << 4));
#endif
            C.col[u] = col;
        }
    }
    pk_walk_lds_rows<RR, RL>(V, s, n, t, live, moved, table, W);
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PK_NO_PRIO)
    __builtin_amdgcn_s_setprio(0);
#endif
#if defined(PK_DBG_STALE) && defined(__HIP_DEVICE_COMPILE__)
    atomicAdd(&g_pk_cnt[4 * blockIdx.x + 0], (unsigned long long)dbg_stale);
#if defined(PK_DBG_VICTIM)
    atomicAdd(&g_pk_cnt[4 * blockIdx.x + 1], (unsigned long long)dbg_loads);
#else
    if ((threadIdx.x & 63) == 0) atomicAdd(&g_pk_cnt[4 * blockIdx.x + 1], (unsigned long long)dbg_loads);
#endif
    atomicAdd(&g_pk_cnt[4 * blockIdx.x + 2], (unsigned long long)(moved && C.TL != 0));
    atomicAdd(&g_pk_cnt[4 * blockIdx.x + 3], (unsigned long long)(n < RR ? n : RR));
    if (C.TL != 0) {
        const int le = V.lines[C.l] & 0xffff, q = V.lines[C.l] >> 16;
        const int su = V.edges[le] & 0xffff, sv = (V.edges[le] >> 16) & 0xffff;
        const int v = V.vid[q >= 5 ? sv : q >= 1 ? su : (su < sv ? su : sv)];
        if (v >= 0 && v < PK_DBG_VCNT) { atomicAdd(&g_pk_vcnt[2 * v], (unsigned long long)dbg_live); atomicAdd(&g_pk_vcnt[2 * v + 1], (unsigned long long)(n < RR ? n : RR)); }
    }
#endif
    return n;
}
// step 1 in the FIRST grad-iter of a launch: nothing is cached yet, every row of every lane is fetched -- from the TILED copy of the table
// (4 rows x 2 columns per 128-byte line: the chunks of a line and the versions of an edge share lines there; the row-major table gives
// every record a line of its own, a million lines per launch at 2048^2 / 3000 and 15 us of a 20-step call).  No comparison, no branch; the
// address arithmetic of the tiled form costs this one grad-iter ~7 instructions per row and the others nothing.  (Rows kept in LDS: from
// the row-major table, as when the first row has moved.)
template <int RR, int RL, int R>
TP_HD int pk_walk_fill(pk_lane_cache<R>& C, const pk_view& V, int s, int pitch, const char* tiled, const char* table, int W) {
    static_assert(RR <= R && RR + RL <= 32, "one bit per row");
    pk_rows t;
    int first = 0;
    if (C.TL == 0) { t.n = 0; t.x = 0; t.xs = 0; t.row = 0; t.rs = 0; }
    else t = pk_lane_rows(V.wk[C.l], C.c, C.TL, C.magic, pitch, &first);
    const uint32_t live = t.n >= 32 ? 0xffffffffu : ((1u << t.n) - 1u);
    C.row0 = t.row;
    uint32_t row = (uint32_t)first;
    const int n = t.n;
#pragma unroll
    for (int u = 0; u < RR; u++) {
        const uint32_t on = 0u - ((live >> u) & 1u);
        const int32_t col = pk_next_col(t, W) & (int32_t)on;
        C.rec[u] = pk_load_rec(tiled, (tp_px_tiled_row_part(row, (uint32_t)pitch) & on) + tp_px_tiled_col_part((uint32_t)col));
        C.col[u] = col;
        row += (uint32_t)pk_row_step(C.TL);
    }
    pk_walk_lds_rows<RR, RL>(V, s, n, t, live, true, table, W);
    return n;
}
// step 2: the line's partial sums of this lane (the crossing columns are the cached ones by now: they are added up here,
// behind the fetches -- on the device behind a wait for the requests into LDS, if there are rows there).  n: the lane's rows, from step 1
template <int RR, int RL, int R>
TP_HD void pk_walk_sum(const pk_lane_cache<R>& C, int n, const pk_view& V, int s, int pitch, const char* table, int W, pk_acc& a) {
    static_assert(RL <= TP_PX_MAXSUM, "records added before unpacking");
    a.xs = 0; a.nodd = 0; a.r = 0; a.g = 0; a.b = 0; a.q = 0;
#pragma unroll
    for (int u = 0; u < RR; u++) a.xs += (uint32_t)C.col[u];
#pragma unroll
    for (int u0 = 0; u0 < RR; u0 += TP_PX_MAXSUM) {   // (so many records add up before a field carries into the next)
        uint64_t lo = 0, hi = 0;
#pragma unroll
        for (int u = u0; u < RR && u < u0 + TP_PX_MAXSUM; u++) { lo += C.rec[u].lo; hi += C.rec[u].hi; }
        pk_add_unpacked(lo, hi, a);
    }
    if (RL > 0 && !pk_lds_rows_idle<RR>(n)) {
        uint64_t lo = 0, hi = 0;
#pragma unroll
        for (int h = 0; h < (RL + 3) / 4; h++) {
            if (h && pk_lds_rows_idle<RR>(n, h)) continue;
            const uint64_t pc = reinterpret_cast<const uint64_t*>(V.lcol)[2 * (size_t)s + h];
#pragma unroll
            for (int v = 0; v < 4 && 4 * h + v < RL; v++) {
                const int u = 4 * h + v;
                a.xs += (uint32_t)((pc >> (16 * v)) & 0xffffu);
                const pk_rec d = *reinterpret_cast<const pk_rec*>(V.lrec + 16 * ((size_t)u * PK_CACHED + s));
                lo += d.lo; hi += d.hi;
            }
        }
        pk_add_unpacked(lo, hi, a);
    }
    if (n > RR + RL) {   // the line has grown beyond the rows this workgroup's lanes keep (its rows are worked out again here, in
                         // the rare case, so that nothing of step 1 but `n` stays in registers across the fetches)
        pk_rows r = pk_lane_rows(V.wk[C.l], C.c, C.TL, C.magic, pitch);
        r.n -= RR + RL; r.x = (int64_t)((uint64_t)r.x + (uint64_t)(RR + RL) * (uint64_t)r.xs); r.row += (uint32_t)(RR + RL) * r.rs;
        pk_walk_rows<4>(r, table, W, a);
    }
}
template <int RR, int RL, int R>
TP_HD void pk_walk_cached(pk_lane_cache<R>& C, const pk_view& V, int s, const char* table, int pitch, int W, pk_acc& a) {
    const int n = pk_walk_pass<RR, RL>(C, V, s, pitch, table, W);
    pk_walk_sum<RR, RL>(C, n, V, s, pitch, table, W, a);
}

// tag of grad-iter `epoch`: never 0 (a cleared mailbox matches nothing), and no two grad-iters of a context's life share one
// (epochs count on across uploads and launches; 2^31 of them)
TP_HD uint32_t pk_tag(uint32_t epoch) { return 0x80000000u | (epoch & 0x7fffffffu); }

// signed sum of three line sums: the exact pixel moments of a variant (tp_raster.h, "Edge-centric form")
// A line's sums in LDS: four 64-bit words {sum x | n_odd << 32, sum r | sum g << 32, sum b, q}.  The halves cannot carry into each
// other: over a whole line sum x and n_odd are at most rows x W <= 2^24, and sum r, sum g at most 255 x 2^24 < 2^32 (rasters of the
// persistent path have at most 4096 rows and columns) -- four LDS atomics per lane-item instead of six, twelve words instead of
// eighteen per variant.
TP_HD void pk_fold_words(const pk_acc& a, unsigned long long w[PK_SUM_WORDS]) {
    w[0] = (unsigned long long)a.xs | ((unsigned long long)a.nodd << 32);
    w[1] = a.r | (a.g << 32);
    w[2] = a.b;
    w[3] = a.q;
}
TP_HD tp_moments pk_moments3(int c0, const unsigned long long* S0, int c1, const unsigned long long* S1, int c2, const unsigned long long* S2) {
    // (measured: with the products replaced by mask / xor / subtract sequences a grad-iter takes 0.25 us MORE -- 64-bit
    // multiply-adds are not what this chain waits for, and the longer sequences cost registers the walk needs)
    int64_t mo[6];
    const unsigned long long* S[3] = {S0, S1, S2};
    const int c[3] = {c0, c1, c2};
#pragma unroll
    for (int q = 0; q < 6; q++) mo[q] = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const unsigned long long w0 = S[k][0], w1 = S[k][1], w2 = S[k][2], w3 = S[k][3];
        mo[0] += (int64_t)c[k] * (int64_t)(uint32_t)w0; mo[1] += (int64_t)c[k] * (int64_t)(w0 >> 32);
        mo[2] += (int64_t)c[k] * (int64_t)(uint32_t)w1; mo[3] += (int64_t)c[k] * (int64_t)(w1 >> 32);
        mo[4] += (int64_t)c[k] * (int64_t)w2; mo[5] += (int64_t)c[k] * (int64_t)w3;
    }
    const tp_moments mm = {mo[0], mo[1], mo[2], mo[3], mo[4], mo[5]};
    return mm;
}
// The signs with which three line sums enter a triangle's moments, WITHOUT its vertices (tp_raster.h: tp_variant_coeffs has them as
// c[k] = sg * sign(Y[k+1] - Y[k]), sg the orientation of the triangle).  sign(Y[k+1] - Y[k]) is how the triangle's edge k runs down
// the raster: the line's own direction (its set-up stored it: pk_view::ldir) -- reversed when the half-edge runs against the edge's
// first -> second endpoint (`flip`, topology: from the plan).  And sg is whatever makes the pixel count non-negative: the count of the
// covered pixels is the moment n, so M = sum_k sign_k W(e_k) has M_n = sg * n, and n > 0 fixes sg; n = 0: no pixel is covered, every
// moment is 0 whichever sign (a triangle without area: its lines cancel row by row).  Nothing of this needs a vertex position, so
// nothing between the positions and the walk computes signs any more (round 3 had three vertex stages and an orientation test per
// corner variant there, on as many waves as the line set-up).
TP_HD tp_moments pk_signed_moments(const pk_view& V, int l0, int l1, int l2, int flips) {
    const unsigned long long *S0 = V.sums + (size_t)l0 * PK_SUM_STRIDE, *S1 = V.sums + (size_t)l1 * PK_SUM_STRIDE, *S2 = V.sums + (size_t)l2 * PK_SUM_STRIDE;
    int c0 = V.ldir[l0], c1 = V.ldir[l1], c2 = V.ldir[l2];
    c0 = (flips & 1) ? -c0 : c0; c1 = (flips & 2) ? -c1 : c1; c2 = (flips & 4) ? -c2 : c2;
    const int64_t n = (int64_t)c0 * (int64_t)(uint32_t)S0[0] + (int64_t)c1 * (int64_t)(uint32_t)S1[0] + (int64_t)c2 * (int64_t)(uint32_t)S2[0];
    const int sg = n < 0 ? -1 : 1;
    return pk_moments3(c0 * sg, S0, c1 * sg, S1, c2 * sg, S2);
}
// P6, lane (corner, move): the variant's moments from the three line sums (slots of the edge leaving the vertex, arriving at it, opposite;
// flips: leaving | arriving << 1 | opposite << 2)
TP_HD tp_moments pk_corner_moments(const pk_view& V, int so, int si, int sopp, int flips) { return pk_signed_moments(V, so, si, sopp, flips); }
// energy of a variant as k_update's emit_variant forms it (triangle.fs:37-43; warp: against the stored colour, :46-53)
TP_HD int32_t pk_energy(const tp_moments& mm, int flavour, pk_i4 col) {
    return tp_wrap32(flavour == 0 ? tp_energy_triangulate(mm) : tp_energy64(mm, col.x, col.y, col.z));
}
// last grad-iter of a call, base variant k of the patch (a triangle whose first vertex it owns): moments from the three base lines
TP_HD tp_moments pk_base_moments(const pk_wg& w, const pk_view& V, int k, int& t) {
    const pk_i4 b = V.base[k];
    t = b.x;
    (void)w;
    return pk_signed_moments(V, b.z & 0xffff, (b.z >> 16) & 0xffff, b.w & 0xffff, (b.w >> 16) & 7);
}

// ---- P6 in packed words (round 5).  pk_signed_moments + pk_energy above are the GENERAL form: eighteen 64-bit products by -1 / 0 / +1 and
// 64-bit moments throughout -- on this part ~100 quarter-rate multiplications on the one wave per SIMD that forms corners, 0.9 us of every
// grad-iter between the walk and the step.  The same values come out of the line sums AS THEY LIE IN LDS:
//   * a line whose direction is 0 has no rows, its sums are 0, so it may take either sign: every line enters as +w or -w;
//   * sum_k (+-w_k) of the packed words {n | n_odd << 32, r | g << 32, b, q} is taken modulo 2^64 -- (w ^ s) + (s & 1) per negated word, the
//     +1s gathered into one addend -- and equals A_lo + 2^32 A_hi (mod 2^64) for the two field sums A_lo, A_hi, whatever borrows ran
//     through the middle;
//   * the orientation is the sign of the field n = the low half of word 0 as an int32 (|n| <= 3 x 2^24); negating all four words once more
//     leaves M_lo + 2^32 M_hi with both halves the moments of the covered pixels: >= 0 and < 2^32 (at most 2^24 pixels of 255), so the halves
//     ARE the fields.  b fits 32 bits for the same reason; only q needs 64.
// The energy then has a fast lane: fewer than 2^23 pixels (no int32 sum of the reference can have wrapped, so the channel averages are
// 0..255) and, warp flavour, a stored colour in 0..255 -- averages by one float reciprocal and a remainder fix (exact), the rest in 24- and
// 32-bit multiplications.  Anything else (a variant of 8 M pixels, a caller's colour outside a byte) takes the general form, lane by lane.
// tests/emul compares the two forms on random and extreme line sums.
struct pk_var { uint32_t n, nodd, r, g, b; uint64_t q; };
TP_HD pk_var pk_signed_packed(const pk_view& V, int l0, int l1, int l2, int flips) {
    const unsigned long long *S0 = V.sums + (size_t)l0 * PK_SUM_STRIDE, *S1 = V.sums + (size_t)l1 * PK_SUM_STRIDE, *S2 = V.sums + (size_t)l2 * PK_SUM_STRIDE;
    const int d0 = V.ldir[l0], d1 = V.ldir[l1], d2 = V.ldir[l2];
    const uint64_t s0 = ((d0 < 0) != ((flips & 1) != 0)) ? ~0ull : 0ull, s1 = ((d1 < 0) != ((flips & 2) != 0)) ? ~0ull : 0ull,
                   s2 = ((d2 < 0) != ((flips & 4) != 0)) ? ~0ull : 0ull;
    const uint64_t ones = (s0 & 1ull) + (s1 & 1ull) + (s2 & 1ull);
    uint64_t z[PK_SUM_WORDS];
#pragma unroll
    for (int q = 0; q < PK_SUM_WORDS; q++) z[q] = (S0[q] ^ s0) + (S1[q] ^ s1) + (S2[q] ^ s2) + ones;
    const uint64_t sg = (int32_t)(uint32_t)z[0] < 0 ? ~0ull : 0ull;
#pragma unroll
    for (int q = 0; q < PK_SUM_WORDS; q++) z[q] = (z[q] ^ sg) + (sg & 1ull);
    pk_var v;
    v.n = (uint32_t)z[0]; v.nodd = (uint32_t)(z[0] >> 32); v.r = (uint32_t)z[1]; v.g = (uint32_t)(z[1] >> 32); v.b = (uint32_t)z[2]; v.q = z[3];
    return v;
}
// floor(x / n) for x < 2^31, 0 < n < 2^23, x <= 255 n: the float quotient is within 1e-4 of the real one, so its truncation is at most one
// off either way and the remainder says which
TP_HD uint32_t pk_avg(uint32_t x, uint32_t n, float rn) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t a = (uint32_t)__fmul_rn((float)x, rn);
    const int32_t rem = (int32_t)(x - __umul24(a, n));
    a = rem < 0 ? a - 1u : ((uint32_t)rem >= n ? a + 1u : a);
    return a;
#else
    (void)rn;
    return x / n;
#endif
}
TP_HD int32_t pk_energy_var(const pk_var& v, int flavour, pk_i4 col) {
    const bool fast = v.n < (1u << 23) && (flavour == 0 || (((uint32_t)col.x | (uint32_t)col.y | (uint32_t)col.z) < 256u));
    if (fast) {
        uint32_t ar, ag, ab;
        if (flavour == 0) {
            if (v.n == 0u) return 0;   // triangle.fs:40
#if defined(__HIP_DEVICE_COMPILE__)
            const float rn = __builtin_amdgcn_rcpf((float)v.n);
#else
            const float rn = 0.0f;
#endif
            ar = pk_avg(v.r, v.n, rn); ag = pk_avg(v.g, v.n, rn); ab = pk_avg(v.b, v.n, rn);
        } else { ar = (uint32_t)col.x; ag = (uint32_t)col.y; ab = (uint32_t)col.z; }
        const uint32_t a2 = pk_mul24(ar, ar) + pk_mul24(ag, ag) + pk_mul24(ab, ab);
        const uint64_t dot = (uint64_t)ar * v.r + (uint64_t)ag * v.g + (uint64_t)ab * v.b;
        const uint64_t S = v.q + (uint64_t)v.n * a2 - 2ull * dot;                  // sum |I - a|^2 >= 0
        const uint32_t nodd = (a2 & 1u) ? v.n - v.nodd : v.nodd;                   // ... of which so many are odd
        return (int32_t)(uint32_t)((S - nodd) >> 1);
    }
    const tp_moments mm = {(int64_t)v.n, (int64_t)v.nodd, (int64_t)v.r, (int64_t)v.g, (int64_t)v.b, (int64_t)v.q};
    return pk_energy(mm, flavour, col);
}
TP_HD pk_var pk_base_var(const pk_view& V, int k, int& t) {
    const pk_i4 b = V.base[k];
    t = b.x;
    return pk_signed_packed(V, b.z & 0xffff, (b.z >> 16) & 0xffff, b.w & 0xffff, (b.w >> 16) & 7);
}

// P7, own vertex k with gradient (gx, gy): the shift.cs step (shift.cs:16-47).  Vertices 0..3 never move.
// one coordinate of it: clamped to [-bound, bound] (and its gradient dropped) BEFORE the step (shift.cs:25-45); bound = RATIO for x, 1 for y
TP_HD float pk_step_axis(float p, int32_t g, float bound, float rate) {
    float tg = (float)g;
    if (p <= -bound) { p = -bound; tg = 0.0f; } else if (p >= bound) { p = bound; tg = 0.0f; }
    return tp_fsub(p, tp_shift_scale(tp_fmul(rate, tg)));
}
TP_HD pk_f2 pk_vertex_lane(pk_f2 p, int32_t gx, int32_t gy, int vid, float ratio, float rate) {
    if (vid < 4) return p;
    pk_f2 r; r.x = pk_step_axis(p.x, gx, ratio, rate); r.y = pk_step_axis(p.y, gy, 1.0f, rate);
    return r;
}
// what a corner adds to its vertex's accumulator of one axis, and what the accumulator then says
TP_HD unsigned long long pk_gacc_word(uint32_t d) { return ((unsigned long long)d << 32) | 1ull; }
