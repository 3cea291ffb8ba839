// CPU replay of the persistent grad-iter kernel (tests only, never shipped): the plan of tp_plan.h and the lane
// functions of tp_persist.h, run workgroup by workgroup and phase by phase with the mailboxes as plain arrays.  What
// it checks (against the oracle, in tests/test_emul.py): that the patches, their position slots, line-sum slots,
// imports and exports describe the same grad-iter as k_lines + k_update -- everything except the device-side
// synchronisation itself.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <map>
#include <vector>
#include "../../tpose_amd/csrc/tp_persist.h"

namespace {
struct part_state {
    std::vector<char> lds;
    pk_view V;
    std::vector<pk_lane_cache<PK_ROWS_PER_LANE>> cache;   // one per cached lane-item, like the kernel's registers
    int n_li = 0, n_li_all = 0, rpl = 0;
};
int64_t* g_recuts = nullptr;   // optional: how often a patch cut its lines again
// optional walk statistics (tools/drift_stats.py): per grad-iter {lanes with a stale record, most lane-items of a patch,
// lanes with rows beyond the kept records, those rows, most such rows in one patch, most stale lanes in one patch}
int64_t* g_walk_stats = nullptr;
int g_walk_stats_iters = 0;
}  // namespace

extern "C" void emul_persist_walk_stats(int64_t* out, int iters, int64_t* recuts) { g_walk_stats = out; g_walk_stats_iters = iters; g_recuts = recuts; }
// pk_magic against the integer division it stands for: mismatches over every chunk count
extern "C" int emul_magic_check() {
    int bad = 0;
    for (int d = 1; d <= PK_MAX_TL; d++) bad += pk_magic(d) != (d == 1 ? 0u : (uint32_t)(0x100000000ull / (uint64_t)d) + 1u);
    return bad;
}

// The packed form of a variant's moments and energy (tp_persist.h: pk_signed_packed, pk_energy_var) against the general 64-bit form
// (pk_signed_moments, pk_energy) on made-up line sums: M = the moments of a covered pixel set (up to 2^24 pixels), three line sums with
// W_a + W_b - W_c = M under random signs, orientations, flips, level lines and orders; fields at their limits.  Returns mismatches.
extern "C" int emul_packed_check(uint64_t seed, int trials) {
    auto rnd = [&]() { seed += 0x9e3779b97f4a7c15ull; uint64_t z = seed; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); };
    std::vector<char> lds(4096, 0);
    pk_view V;
    V.sums = (unsigned long long*)lds.data();
    V.ldir = (int32_t*)(lds.data() + 2048);
    int bad = 0;
    for (int trial = 0; trial < trials; trial++) {
        // M: n pixels, every field somewhere between 0 and its bound (often AT the bound)
        const int kind = (int)(rnd() % 8);
        uint64_t n = kind == 0 ? 0 : kind == 1 ? (1ull << 24) : kind == 2 ? (1ull << 23) : kind == 3 ? (1ull << 23) - 1 : kind == 4 ? 1 + rnd() % 4000 : rnd() % ((1ull << 24) + 1);
        // ... of two colours (a real pixel set: sum |I - a|^2 >= 0 for every a), often black or white
        auto colour = [&](uint64_t c[3]) { const int k = (int)(rnd() % 4); for (int i = 0; i < 3; i++) c[i] = k == 0 ? 255 : k == 1 ? 0 : rnd() % 256; };
        uint64_t c1[3], c2[3];
        colour(c1); colour(c2);
        const uint64_t n1 = (rnd() & 1) ? n : (n ? rnd() % (n + 1) : 0), n2 = n - n1;
        const uint64_t o1 = (c1[0] + c1[1] + c1[2]) & 1, o2 = (c2[0] + c2[1] + c2[2]) & 1;
        uint64_t M[6] = {n, n1 * o1 + n2 * o2, n1 * c1[0] + n2 * c2[0], n1 * c1[1] + n2 * c2[1], n1 * c1[2] + n2 * c2[2],
                         n1 * (c1[0] * c1[0] + c1[1] * c1[1] + c1[2] * c1[2]) + n2 * (c2[0] * c2[0] + c2[1] * c2[1] + c2[2] * c2[2])};
        // X, Y >= 0 so that W_a = M + X, W_b = Y, W_c = X + Y stay inside a line sum's fields (n, n_odd <= 2^24; r, g < 2^32)
        uint64_t Wa[6], Wb[6], Wc[6];
        const uint64_t cap[6] = {1ull << 24, 1ull << 24, (1ull << 32) - 1, (1ull << 32) - 1, (1ull << 32), 1ull << 42};
        for (int f = 0; f < 6; f++) {
            const uint64_t room = cap[f] - M[f];
            const uint64_t X = room ? rnd() % (room / 2 + 1) : 0, Y = room ? rnd() % (room / 2 + 1) : 0;
            Wa[f] = M[f] + X; Wb[f] = Y; Wc[f] = X + Y;
        }
        // lines in a random order; the negative one by its direction or by its flip bit; the whole triangle maybe the other way round
        const uint64_t* W3[3] = {Wa, Wb, Wc};
        int sign[3] = {1, 1, -1};
        if (rnd() & 1) for (int k = 0; k < 3; k++) sign[k] = -sign[k];
        int perm[3] = {0, 1, 2};
        for (int k = 2; k > 0; k--) { const int j = (int)(rnd() % (k + 1)); const int t_ = perm[k]; perm[k] = perm[j]; perm[j] = t_; }
        int flips = 0, slot[3];
        for (int k = 0; k < 3; k++) {
            const int src = perm[k];
            slot[k] = k * 3 + (int)(rnd() % 3);   // (three different slots)
            unsigned long long* S = V.sums + (size_t)slot[k] * PK_SUM_STRIDE;
            const uint64_t* Wk = W3[src];
            const bool zero = Wk[0] == 0 && Wk[1] == 0 && Wk[2] == 0 && Wk[3] == 0 && Wk[4] == 0 && Wk[5] == 0;
            S[0] = Wk[0] | (Wk[1] << 32); S[1] = Wk[2] | (Wk[3] << 32); S[2] = Wk[4]; S[3] = Wk[5];
            const int fl = (int)(rnd() & 1);
            flips |= fl << k;
            V.ldir[slot[k]] = (zero && (rnd() & 1)) ? 0 : (fl ? -sign[src] : sign[src]);   // (a line without rows may be level)
        }
        for (int flavour = 0; flavour < 2; flavour++) {
            pk_i4 col = {(int32_t)(rnd() % 256), (int32_t)(rnd() % 256), (int32_t)(rnd() % 256), 0};
            if (flavour == 1 && rnd() % 4 == 0) col.y = (int32_t)rnd();   // (a caller's colour outside a byte: the general form)
            const pk_var v = pk_signed_packed(V, slot[0], slot[1], slot[2], flips);
            const tp_moments mm = pk_signed_moments(V, slot[0], slot[1], slot[2], flips);
            const bool same = mm.n == (int64_t)v.n && mm.nodd == (int64_t)v.nodd && mm.sr == (int64_t)v.r && mm.sg == (int64_t)v.g && mm.sb == (int64_t)v.b && mm.q == (int64_t)v.q &&
                              (uint64_t)mm.n == M[0] && (uint64_t)mm.q == M[5];
            bad += !same;
            bad += pk_energy_var(v, flavour, col) != pk_energy(mm, flavour, col);
        }
    }
    return bad;
}

// Band split (tp_band_attach): band `band` of `n_bands` replays the patches [band, band + 1) * parts / n_bands only; after every
// grad-iter `exchange` is handed the mailbox slot array its patches just posted into ({tag : 32, float : 32} granules, two per
// vertex) and the tag of the new positions, and brings in the other bands' posts (tests: torch.distributed all_gather).
typedef void (*emul_exchange_fn)(void* user, unsigned long long* slot_array, int NP, uint32_t tag);

// the packed line sums at their limit: a line of 4096 rows crossing column 4096 of an all-white raster, folded from 256
// lanes' partial sums of 16 rows each, read back through pk_moments3 -- mismatches against plain 64-bit sums
extern "C" int emul_fold_check() {
    unsigned long long words[3][PK_SUM_WORDS] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    int64_t want[6] = {0, 0, 0, 0, 0, 0};
    for (int lane = 0; lane < 256; lane++) {
        pk_acc a;
        a.xs = 16u * 4096u; a.nodd = 16u * 4096u; a.r = a.g = a.b = 16ull * 4096ull * 255ull; a.q = 16ull * 4096ull * 195075ull;
        unsigned long long wd[PK_SUM_WORDS];
        pk_fold_words(a, wd);
        for (int q = 0; q < PK_SUM_WORDS; q++) words[0][q] += wd[q];
        want[0] += a.xs; want[1] += a.nodd; want[2] += (int64_t)a.r; want[3] += (int64_t)a.g; want[4] += (int64_t)a.b; want[5] += (int64_t)a.q;
    }
    const tp_moments m = pk_moments3(1, words[0], 0, words[1], -1, words[2]);
    const tp_moments n = pk_moments3(-1, words[0], 1, words[1], 0, words[2]);
    int bad = 0;
    bad += m.n != want[0]; bad += m.nodd != want[1]; bad += m.sr != want[2]; bad += m.sg != want[3]; bad += m.sb != want[4]; bad += m.q != want[5];
    bad += n.n != -want[0]; bad += n.nodd != -want[1]; bad += n.sr != -want[2]; bad += n.sg != -want[3]; bad += n.sb != -want[4]; bad += n.q != -want[5];
    return bad;
}

// returns 0, or a negative code: -1 plan refused, -2 a position granule had the wrong tag, -3 the patches do not split evenly
static int emul_persist_impl(const uint8_t* img, size_t stride, int W, int H, float* points, int NP, const int32_t* tris,
                             int NT, const int32_t* ca, int flavour, float dp, float ratio, float rate, int iters,
                             int max_parts, int lds_limit, int64_t* stats, int32_t* ten, int32_t* cn, int32_t* ca_out, int32_t* gr,
                             int band, int n_bands, emul_exchange_fn exchange, void* user) {
    tp_view vw;
    vw.dp = dp; vw.ratio = ratio; vw.halfW = 0.5f * (float)W; vw.halfH = 0.5f * (float)H; vw.W = W; vw.H = H;
    // the per-image table in pixel records, as k_prefix_px builds it
    if (W > TP_PX_MAXW) return -1;
    const int pitch = tp_px_pitch(W);
    std::vector<uint64_t> T((size_t)H * pitch * 2, 0);
    // (... and its tiled copy; 0xCD where k_prefix_px writes nothing, so that a look-up outside the columns 0..W shows)
    std::vector<uint64_t> TT((size_t)tp_px_tiled_rows((uint32_t)H) * pitch * 2, 0xCDCDCDCDCDCDCDCDull);
    for (int r = 0; r < H; r++) {
        uint32_t run[5] = {0, 0, 0, 0, 0};
        for (int c = 0; c <= W; c++) {
            tp_px_pack(run, &T[((size_t)r * pitch + c) * 2]);
            tp_px_pack(run, &TT[(size_t)((tp_px_tiled_row_part((uint32_t)r, (uint32_t)pitch) + tp_px_tiled_col_part((uint32_t)c)) >> 3)]);
            if (c < W) {
                uint32_t px;
                memcpy(&px, img + (size_t)r * stride + 4 * (size_t)c, 4);
                const uint32_t R = px & 0xffu, G = (px >> 8) & 0xffu, B = (px >> 16) & 0xffu;
                run[0] += (R + G + B) & 1u; run[1] += R; run[2] += G; run[3] += B; run[4] += R * R + G * G + B * B;
            }
        }
    }
    // undirected edges in order of first appearance, as tp_upload numbers them
    std::map<std::pair<int, int>, int> eid;
    std::vector<int32_t> edge_uv, he_edge(3 * (size_t)NT);
    for (int t = 0; t < NT; t++)
        for (int k = 0; k < 3; k++) {
            const int o = tris[4 * t + k], d = tris[4 * t + (k + 1) % 3];
            const std::pair<int, int> key(o < d ? o : d, o < d ? d : o);
            auto it = eid.find(key);
            if (it == eid.end()) { it = eid.emplace(key, (int)(edge_uv.size() / 2)).first; edge_uv.push_back(key.first); edge_uv.push_back(key.second); }
            he_edge[3 * t + k] = it->second * 2 + (o != key.first ? 1 : 0);
        }
    const int NE = (int)(edge_uv.size() / 2);
    pk_plan P;
    pk_build_plan(NP, NT, tris, points, NE, edge_uv.data(), he_edge.data(), W, H, ratio, dp * 0.5f * (float)H, max_parts, lds_limit, P);
    if (stats) { stats[0] = P.ok; stats[1] = P.parts; stats[2] = P.lds_bytes; stats[3] = P.lines_total; stats[4] = P.foreign_total; stats[5] = NE;
                 stats[12] = P.rows_max; stats[13] = P.wg.empty() ? 0 : P.wg[0].lds_rows; }   // (rows per lane of the largest patch; of which in LDS)
    if (!P.ok) return -1;
    if (P.parts % n_bands) return -3;
    const int p_lo = band * (P.parts / n_bands), p_hi = (band + 1) * (P.parts / n_bands);

    std::vector<part_state> S((size_t)P.parts);
    for (int p = 0; p < P.parts; p++) {
        const pk_wg& w = P.wg[p];
        S[p].lds.assign((size_t)w.lds_bytes, (char)0xA5);   // (LDS holds anything when a kernel starts; the prologue clears what it must)
        pk_view& V = S[p].V;
        pk_carve(S[p].lds.data(), w, V);
        memcpy(V.vid, &P.pool[w.off_vid], sizeof(int32_t) * w.n_slots);
        memcpy(V.edges, &P.pool[w.off_edges], sizeof(int32_t) * w.n_edges);
        memcpy(V.lines, &P.pool[w.off_lines], sizeof(int32_t) * w.n_lines_all);
        memcpy(V.corners, &P.pool[w.off_corners], sizeof(int32_t) * 4 * w.n_corners);
        memcpy(V.base, &P.pool[w.off_base], sizeof(int32_t) * 4 * w.n_base);
        for (int s = 0; s < w.n_slots; s++) { V.pos[s].x = points[2 * V.vid[s]]; V.pos[s].y = points[2 * V.vid[s] + 1]; }
        S[p].cache.resize(PK_CACHED);
        memset((void*)S[p].cache.data(), 0xCD, sizeof(S[p].cache[0]) * S[p].cache.size());   // (registers hold anything when a kernel starts)
        for (auto& C : S[p].cache) pk_slot_clear(C);   // (... but no slot has a lane-item before the first cut: the kernel's prologue)
        for (int sl = 0; sl < PK_CACHED; sl++) V.st[sl] = -1;
    }
    std::vector<unsigned long long> posbox((size_t)2 * NP * 2, 0);
    const char* table = reinterpret_cast<const char*>(T.data());
    for (int it = 0; it < iters; it++) {
        const uint32_t e = 1 + it, tag = pk_tag(e), par = e & 1;
        const bool emit = it + 1 == iters && ten != nullptr;   // the last grad-iter writes the reference's buffers
        for (int p = p_lo; p < p_hi; p++) {
            const pk_wg& w = P.wg[p];
            pk_view& V = S[p].V;
            const bool recut = (it & (PK_RECUT - 1)) == 0;
            const int n_lines = emit ? w.n_lines_all : w.n_lines, n_setup = recut ? w.n_lines_all : n_lines;
            if (it > 0)  // P0
                for (int s = w.n_own_v; s < w.n_slots; s++) {
                    const unsigned long long* g = &posbox[((size_t)par * NP + V.vid[s]) * 2];
                    if ((uint32_t)(g[0] >> 32) != tag || (uint32_t)(g[1] >> 32) != tag) return -2;
                    const uint32_t bx = (uint32_t)g[0], by = (uint32_t)g[1];
                    memcpy(&V.pos[s].x, &bx, 4); memcpy(&V.pos[s].y, &by, 4);
                }
            // P1
            for (int l = 0; l < n_setup; l++) {
                pk_walker wkr;
                V.ldir[l] = pk_setup_lane(V, vw, l, wkr);
                V.wk[l] = wkr;
            }
            memset(V.sums, 0, sizeof(unsigned long long) * PK_SUM_STRIDE * (size_t)w.n_lines_all);
            if (recut) {   // the four passes of a cut (tp_persist.h, "SLOTS"): the wave's lanes / the slots one after the other
                const int RRk = w.lds_rows == PK_LDS_ROWS_BIG ? PK_ROWS_BIG : pk_rr_for(P.rows_max);   // tp_launch_persist's choice (tp_persist_host.hip)
                int changed = 0;
                int rpl = it == 0 ? w.rows : S[p].rpl;
                bool first = it == 0;
                for (;;) {
                    int every = 0;
                    changed = 0;
                    for (int lane = 0; lane < 64; lane++) every += pk_cut_want(V, w.n_lines_all, w.n_lines, lane, 64, rpl, first, changed);
                    if (every <= PK_CACHED || rpl >= RRk) break;
                    rpl++; first = true;
                }
                S[p].rpl = rpl;
                if (first) changed = 1;   // (a patch without lines still gets its empty tables written)
                if (!changed) for (int lane = 0; lane < 64; lane++) pk_cut_forget(V, w.n_lines_all, lane, 64);
                if (changed) {
                    int n_free = 0, unc[64], off = 0;
                    for (int sl = 0; sl < PK_CACHED; sl++)
                        if (pk_slot_release(S[p].cache[sl], V, first)) { if (!first) V.freel[n_free] = sl; n_free++; }
                    if (first && n_free != PK_CACHED) return -7;
                    if (first) {   // chunk-major, slot = place
                        for (int lane = 0; lane < 64; lane++) pk_cut_fresh_begin(V, w.n_lines_all, lane, 64);
                        int base = 0;
                        for (int c = 0;; c++) {
                            int k[64], total = 0;
                            for (int lane = 0; lane < 64; lane++) { k[lane] = pk_cut_level_count(V, w.n_lines, lane, 64, w.n_lines_all, c); total += k[lane]; }
                            if (total == 0) break;
                            int run = base;
                            for (int lane = 0; lane < 64; lane++) { pk_cut_level_assign(V, w.n_lines, lane, 64, w.n_lines_all, c, run, PK_CACHED); run += k[lane]; }
                            base += total;
                        }
                        for (int lane = 0; lane < 64; lane++) unc[lane] = pk_cut_fresh_done(V, w.n_lines_all, lane, 64);
                    } else {
                        int base = 0, need[64];
                        for (int lane = 0; lane < 64; lane++) need[lane] = pk_cut_need(V, w.n_lines_all, w.n_lines, lane, 64);
                        for (int lane = 0; lane < 64; lane++) { unc[lane] = pk_cut_alloc(V, w.n_lines_all, w.n_lines, lane, 64, base, n_free); base += need[lane]; }
                    }
                    for (int lane = 0; lane < 64; lane++) { pk_cut_write(V, w.n_lines_all, lane, 64, off); off += unc[lane]; }
                    for (int sl = 0; sl < PK_CACHED; sl++) pk_slot_take(S[p].cache[sl], V, sl);
                    for (int sl = 0; sl < PK_CACHED; sl++) if (V.st[sl] != -1) return -8;   // (a lane-item handed to a slot that was not free)
                    for (int l = 0; l < w.n_lines_all; l++) pk_list_line(V, l, w.li_cap);
                    // every chunk of every line walked every grad-iter belongs to exactly one slot, or to the uncached lane-items
                    std::vector<int> seen;
                    for (int l = 0; l < w.n_lines; l++) {
                        if (V.nc[l] < 0 || V.nc[l] > V.tl[l]) return -9;
                        seen.assign((size_t)V.tl[l], 0);
                        for (int sl = 0; sl < PK_CACHED; sl++) {
                            const auto& C = S[p].cache[sl];
                            if (C.TL != 0 && C.l == l) { if (C.TL != V.tl[l] || C.c < 0 || C.c >= V.nc[l] || seen[(size_t)C.c]) return -10; seen[(size_t)C.c] = 1; }
                        }
                        for (int c = 0; c < V.nc[l]; c++) if (!seen[(size_t)c]) return -11;
                    }
                }
                S[p].n_li = V.cut[w.n_lines]; S[p].n_li_all = V.cut[w.n_lines_all];   // (the UNCACHED lane-items)
                if (g_recuts) g_recuts[0] += changed;
            }
            for (size_t i = 0; i < (size_t)PK_SUM_STRIDE * (size_t)w.n_lines_all; i++) if (V.sums[i] != 0ull) return -12;   // (the cut's scratch left in a line's sums)
            const int n_li = emit ? S[p].n_li_all : S[p].n_li;
            for (int k = 0; k < w.n_own_v; k++) { V.gacc[2 * k] = 0ull; V.gacc[2 * k + 1] = 0ull; }
            // P3
            if (g_walk_stats && it < g_walk_stats_iters) {
                int64_t* ws = g_walk_stats + 6 * (size_t)it;
                const int RRk = w.lds_rows == PK_LDS_ROWS_BIG ? PK_ROWS_BIG : pk_rr_for(P.rows_max);   // tp_launch_persist's choice (tp_persist_host.hip)
                int64_t over_rows = 0, stale = 0;
                for (int j = 0; j < PK_CACHED; j++) {   // (what pk_walk_pass is about to find, without changing anything)
                    const auto& C = S[p].cache[j];
                    if (C.TL == 0) continue;
                    pk_rows r = pk_lane_rows(V.wk[C.l], C.c, C.TL, C.magic, pitch);
                    const int n = r.n;
                    bool st = r.row != C.row0;
                    for (int u = 0; u < PK_ROWS_PER_LANE; u++) {
                        const int32_t col = u < n ? pk_next_col(r, W) : 0;
                        st = st || col != C.col[u];
                    }
                    if (st) { ws[0]++; stale++; }
                    if (n > RRk) { ws[2]++; ws[3] += n - RRk; over_rows += n - RRk; }
                }
                if (over_rows > ws[4]) ws[4] = over_rows;
                if (stale > ws[5]) ws[5] = stale;
                if (S[p].n_li > ws[1]) ws[1] = S[p].n_li;
            }
            for (int j = 0; j < PK_CACHED; j++) {   // the cached slots (a slot without a lane-item walks nothing)
                pk_acc a;
                if (w.lds_rows == PK_LDS_ROWS_BIG) pk_walk_cached<PK_ROWS_PER_LANE, PK_LDS_ROWS_BIG>(S[p].cache[j], V, j, table, pitch, W, a);
                else if (w.lds_rows) pk_walk_cached<PK_ROWS_PER_LANE, PK_LDS_ROWS>(S[p].cache[j], V, j, table, pitch, W, a);
                else pk_walk_cached<PK_ROWS_PER_LANE, 0>(S[p].cache[j], V, j, table, pitch, W, a);
                if (S[p].cache[j].TL != 0) {   // what the slot keeps now -- in registers and in LDS -- is the record of every row's CURRENT crossing column
                    const auto& C = S[p].cache[j];
                    pk_rows r = pk_lane_rows(V.wk[C.l], C.c, C.TL, C.magic, pitch);
                    const int n = r.n;
                    if (C.row0 != r.row) return -13;
                    for (int u = 0; u < PK_ROWS_PER_LANE + w.lds_rows; u++) {
                        const uint32_t col = u < n ? (uint32_t)pk_next_col(r, W) : 0u;
                        const uint32_t off = (u < n ? r.row + (uint32_t)u * r.rs : 0u) + (col << 4);
                        const int v = u - PK_ROWS_PER_LANE;
                        const uint32_t have = v < 0 ? (uint32_t)C.col[u] : (uint32_t)V.lcol[(size_t)j * 8 + v];
                        const void* rec = v < 0 ? (const void*)&C.rec[u] : (const void*)(V.lrec + 16 * ((size_t)v * PK_CACHED + j));
                        if (have != col || memcmp(rec, table + off, 16) != 0) return -14;
                    }
                }
                unsigned long long* s = V.sums + (size_t)S[p].cache[j].l * PK_SUM_STRIDE;
                unsigned long long wd[PK_SUM_WORDS];
                pk_fold_words(a, wd);
                for (int q = 0; q < PK_SUM_WORDS; q++) s[q] += wd[q];
            }
            for (int j = 0; j < n_li; j++) {   // the lane-items without a slot
                // (every other lane-item in four parts on four "lanes", as the kernel walks them when a patch has few: the same sums)
                const int parts = (j & 1) ? 4 : 1;
                for (int part = 0; part < parts; part++) {
                    pk_acc a;
                    const int l = pk_walk_lane(V, table, reinterpret_cast<const char*>(TT.data()), pitch, W, w.n_lines_all, 0, w.li_cap, j, a, part, parts);
                    unsigned long long* s = V.sums + (size_t)l * PK_SUM_STRIDE;
                    unsigned long long wd[PK_SUM_WORDS];
                    pk_fold_words(a, wd);
                    for (int q = 0; q < PK_SUM_WORDS; q++) s[q] += wd[q];
                }
            }
            // P6
            for (int k = 0; k < w.n_corners; k++) {
                const pk_i4 cr = V.corners[k];
                const int t = cr.x, s = cr.y & 3, own = (cr.y >> 2) & 0x3ff;
                int32_t en[5] = {0, 0, 0, 0, 0};
                for (int m = 1; m <= 4; m++) {
                    pk_i4 col = {0, 0, 0, 0};
                    if (flavour == 1 && ca) { const int32_t* c4 = ca + 4 * ((size_t)(4 * s + m) * NT + t); col.x = c4[0]; col.y = c4[1]; col.z = c4[2]; }
                    const pk_var mv = pk_signed_packed(V, (cr.z & 0xffff) + m - 1, ((cr.z >> 16) & 0xffff) + m - 1, cr.w & 0xffff, (cr.w >> 16) & 7);
                    en[m] = pk_energy_var(mv, flavour, col);
                    {   // (the general form of both, which the packed one replaces)
                        const tp_moments mm = pk_corner_moments(V, (cr.z & 0xffff) + m - 1, ((cr.z >> 16) & 0xffff) + m - 1, cr.w & 0xffff, (cr.w >> 16) & 7);
                        if (mm.n != (int64_t)mv.n || mm.nodd != (int64_t)mv.nodd || mm.sr != (int64_t)mv.r || mm.sg != (int64_t)mv.g || mm.sb != (int64_t)mv.b ||
                            mm.q != (int64_t)mv.q || pk_energy(mm, flavour, col) != en[m]) return -5;
                    }
                    if (emit) {
                        const size_t id = (size_t)(4 * s + m) * NT + t;
                        if (flavour == 0) { ca_out[4 * id] = (int32_t)mv.r; ca_out[4 * id + 1] = (int32_t)mv.g; ca_out[4 * id + 2] = (int32_t)mv.b; ca_out[4 * id + 3] = 0; }
                        ten[id] = en[m]; cn[id] = (int32_t)mv.n;
                    }
                }
                V.gacc[2 * own] += pk_gacc_word((uint32_t)en[1] - (uint32_t)en[2]);       // (the kernel: one returning LDS atomic each)
                V.gacc[2 * own + 1] += pk_gacc_word((uint32_t)en[3] - (uint32_t)en[4]);
            }
            if (emit)
                for (int k = 0; k < w.n_base; k++) {
                    int t;
                    const pk_var mv = pk_base_var(V, k, t);
                    pk_i4 col = {0, 0, 0, 0};
                    if (flavour == 1 && ca) { col.x = ca[4 * t]; col.y = ca[4 * t + 1]; col.z = ca[4 * t + 2]; }
                    if (flavour == 0) { ca_out[4 * t] = (int32_t)mv.r; ca_out[4 * t + 1] = (int32_t)mv.g; ca_out[4 * t + 2] = (int32_t)mv.b; ca_out[4 * t + 3] = 0; }
                    ten[t] = pk_energy_var(mv, flavour, col); cn[t] = (int32_t)mv.n;
                }
            // P7: posts go to the OTHER parity of the mailbox, so workgroups replayed later in this sweep still read this
            // grad-iter's positions
            for (int k = 0; k < w.n_own_v; k++) {
                // (every corner of the vertex has been counted on both axes; the sums are the high words)
                int ncorn = 0;
                for (int q = 0; q < w.n_corners; q++) ncorn += ((V.corners[q].y >> 2) & 0x3ff) == k;
                if ((uint32_t)V.gacc[2 * k] != (uint32_t)ncorn || (uint32_t)V.gacc[2 * k + 1] != (uint32_t)ncorn) return -4;
                const int32_t gx_ = (int32_t)(uint32_t)(V.gacc[2 * k] >> 32), gy_ = (int32_t)(uint32_t)(V.gacc[2 * k + 1] >> 32);
                if (emit && gr) { gr[2 * V.vid[k]] = gx_; gr[2 * V.vid[k] + 1] = gy_; }
                pk_f2 np_ = V.pos[k];
                if (V.vid[k] >= 4) { np_.x = pk_step_axis(np_.x, gx_, ratio, rate); np_.y = pk_step_axis(np_.y, gy_, 1.0f, rate); }
                V.pos[k] = np_;
                uint32_t bx, by;
                memcpy(&bx, &np_.x, 4); memcpy(&by, &np_.y, 4);
                const uint32_t tn = pk_tag(e + 1);
                unsigned long long* g = &posbox[((size_t)((e + 1) & 1) * NP + V.vid[k]) * 2];
                g[0] = ((unsigned long long)tn << 32) | bx; g[1] = ((unsigned long long)tn << 32) | by;
            }
        }
        if (exchange) exchange(user, &posbox[((size_t)((e + 1) & 1) * NP) * 2], NP, pk_tag(e + 1));
    }
    for (int p = p_lo; p < p_hi; p++) {
        const pk_wg& w = P.wg[p];
        for (int k = 0; k < w.n_own_v; k++) { points[2 * S[p].V.vid[k]] = S[p].V.pos[k].x; points[2 * S[p].V.vid[k] + 1] = S[p].V.pos[k].y; }
    }
    if (n_bands > 1 && iters > 0) {   // the other bands' vertices: what they posted last (tp_launch_band_collect)
        const uint32_t tn = pk_tag(1 + (uint32_t)iters);
        const unsigned long long* slot = &posbox[((size_t)((1 + iters) & 1) * NP) * 2];
        for (int v = 0; v < NP; v++)
            if ((uint32_t)(slot[2 * v] >> 32) == tn && (uint32_t)(slot[2 * v + 1] >> 32) == tn) {
                const uint32_t bx = (uint32_t)slot[2 * v], by = (uint32_t)slot[2 * v + 1];
                memcpy(&points[2 * v], &bx, 4); memcpy(&points[2 * v + 1], &by, 4);
            }
    }
    return 0;
}
extern "C" int emul_persist(const uint8_t* img, size_t stride, int W, int H, float* points, int NP, const int32_t* tris,
                            int NT, const int32_t* ca, int flavour, float dp, float ratio, float rate, int iters,
                            int max_parts, int lds_limit, int64_t* stats, int32_t* ten, int32_t* cn, int32_t* ca_out, int32_t* gr) {
    return emul_persist_impl(img, stride, W, H, points, NP, tris, NT, ca, flavour, dp, ratio, rate, iters, max_parts, lds_limit, stats, ten, cn,
                             ca_out, gr, 0, 1, nullptr, nullptr);
}
extern "C" int emul_persist_band(const uint8_t* img, size_t stride, int W, int H, float* points, int NP, const int32_t* tris,
                                 int NT, const int32_t* ca, int flavour, float dp, float ratio, float rate, int iters,
                                 int max_parts, int lds_limit, int64_t* stats, int band, int n_bands, emul_exchange_fn exchange, void* user) {
    return emul_persist_impl(img, stride, W, H, points, NP, tris, NT, ca, flavour, dp, ratio, rate, iters, max_parts, lds_limit, stats, nullptr,
                             nullptr, nullptr, nullptr, band, n_bands, exchange, user);
}

// statistics of a plan, for tests of the cut itself: owner of every vertex and edge
extern "C" int emul_plan(const float* points, int NP, const int32_t* tris, int NT, int W, int H, float ratio, float dp_px,
                         int max_parts, int lds_limit, int32_t* owner_v, int32_t* owner_e, int64_t* stats) {
    std::map<std::pair<int, int>, int> eid;
    std::vector<int32_t> edge_uv, he_edge(3 * (size_t)NT);
    for (int t = 0; t < NT; t++)
        for (int k = 0; k < 3; k++) {
            const int o = tris[4 * t + k], d = tris[4 * t + (k + 1) % 3];
            const std::pair<int, int> key(o < d ? o : d, o < d ? d : o);
            auto it = eid.find(key);
            if (it == eid.end()) { it = eid.emplace(key, (int)(edge_uv.size() / 2)).first; edge_uv.push_back(key.first); edge_uv.push_back(key.second); }
            he_edge[3 * t + k] = it->second * 2 + (o != key.first ? 1 : 0);
        }
    const int NE = (int)(edge_uv.size() / 2);
    pk_plan P;
    pk_build_plan(NP, NT, tris, points, NE, edge_uv.data(), he_edge.data(), W, H, ratio, dp_px, max_parts, lds_limit, P);
    if (stats) {
        stats[0] = P.ok; stats[1] = P.parts; stats[2] = P.lds_bytes; stats[3] = P.lines_total; stats[4] = P.foreign_total; stats[5] = NE;
        stats[6] = (int64_t)P.work_max; stats[7] = (int64_t)P.work_mean;
        int ml = 0, mc = 0, ms = 0, mit = 0;
        for (auto& w : P.wg) { ml = tp_max(ml, w.n_lines); mc = tp_max(mc, w.n_corners); ms = tp_max(ms, w.n_slots); mit = tp_max(mit, w.n_li); }
        stats[8] = ml; stats[9] = mc; stats[10] = ms; stats[11] = mit;
        stats[12] = P.rows_max;   // rows per lane of the largest patch (above PK_ROWS_PER_LANE: the rows beyond the registers live in LDS)
    }
    if (!P.ok) return -1;
    if (owner_v) memcpy(owner_v, P.owner_v.data(), sizeof(int32_t) * NP);
    (void)owner_e;
    return 0;
}

// per-patch figures of a plan (tools/plan_stats.py): {own vertices, slots, edges, lines, lane-items, corners, rows per lane}
extern "C" int emul_plan_patches(const float* points, int NP, const int32_t* tris, int NT, int W, int H, float ratio, float dp_px,
                                 int max_parts, int lds_limit, int32_t* out, int cap) {
    std::map<std::pair<int, int>, int> eid;
    std::vector<int32_t> edge_uv, he_edge(3 * (size_t)NT);
    for (int t = 0; t < NT; t++)
        for (int k = 0; k < 3; k++) {
            const int o = tris[4 * t + k], d = tris[4 * t + (k + 1) % 3];
            const std::pair<int, int> key(o < d ? o : d, o < d ? d : o);
            auto it = eid.find(key);
            if (it == eid.end()) { it = eid.emplace(key, (int)(edge_uv.size() / 2)).first; edge_uv.push_back(key.first); edge_uv.push_back(key.second); }
            he_edge[3 * t + k] = it->second * 2 + (o != key.first ? 1 : 0);
        }
    pk_plan P;
    pk_build_plan(NP, NT, tris, points, (int)(edge_uv.size() / 2), edge_uv.data(), he_edge.data(), W, H, ratio, dp_px, max_parts, lds_limit, P);
    if (!P.ok) return -1;
    for (int p = 0; p < P.parts && p < cap; p++) {
        const pk_wg& w = P.wg[p];
        int32_t* o = out + 8 * p;
        o[0] = w.n_own_v; o[1] = w.n_slots; o[2] = w.n_edges; o[3] = w.n_lines; o[4] = w.n_li; o[5] = w.n_corners; o[6] = w.rows; o[7] = w.lds_bytes;
    }
    return P.parts;
}

// per-patch share of EARLY lines (tools/plan_stats.py): lines whose two endpoints are both the patch's own vertices -- what a workgroup can
// set up and walk before its neighbours' positions arrive.  out[4p ..] = {lines, early lines, rows of all lines, rows of the early ones}
extern "C" int emul_plan_early(const float* points, int NP, const int32_t* tris, int NT, int W, int H, float ratio, float dp_px,
                               int max_parts, int lds_limit, int32_t* out, int cap) {
    std::map<std::pair<int, int>, int> eid;
    std::vector<int32_t> edge_uv, he_edge(3 * (size_t)NT);
    for (int t = 0; t < NT; t++)
        for (int k = 0; k < 3; k++) {
            const int o = tris[4 * t + k], d = tris[4 * t + (k + 1) % 3];
            const std::pair<int, int> key(o < d ? o : d, o < d ? d : o);
            auto it = eid.find(key);
            if (it == eid.end()) { it = eid.emplace(key, (int)(edge_uv.size() / 2)).first; edge_uv.push_back(key.first); edge_uv.push_back(key.second); }
            he_edge[3 * t + k] = it->second * 2 + (o != key.first ? 1 : 0);
        }
    pk_plan P;
    pk_build_plan(NP, NT, tris, points, (int)(edge_uv.size() / 2), edge_uv.data(), he_edge.data(), W, H, ratio, dp_px, max_parts, lds_limit, P);
    if (!P.ok) return -1;
    for (int p = 0; p < P.parts && p < cap; p++) {
        const pk_wg& w = P.wg[p];
        int32_t* o = out + 4 * p;
        o[0] = w.n_lines; o[1] = 0; o[2] = 0; o[3] = 0;
        for (int l = 0; l < w.n_lines; l++) {
            const int ln = P.pool[w.off_lines + l], ed = P.pool[w.off_edges + (ln & 0xffff)];
            const int su = ed & 0xffff, sv = (ed >> 16) & 0xffff;
            const int vu = P.pool[w.off_vid + su], vv = P.pool[w.off_vid + sv];
            const int rows = (int)(fabsf(points[2 * vu + 1] - points[2 * vv + 1]) * 0.5f * (float)H) + 1;
            const bool early = su < w.n_own_v && sv < w.n_own_v;
            o[1] += early; o[2] += rows; o[3] += early ? rows : 0;
        }
    }
    return P.parts;
}
