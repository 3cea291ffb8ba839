// CPU emulation of the device data flow (tests only, never shipped): runs the SAME per-lane integer logic
// (tpose_amd/csrc/tp_raster.h) on the host -- the span walker of k_render window by window, and the table path of
// k_prefix / k_lines / k_finalize record by record -- so that it can be checked against the oracle without a GPU.
#include <stdint.h>
#include <string.h>
#include <map>
#include <vector>
#include "../../tpose_amd/csrc/tp_raster.h"

#define TW 128
#define TH 32

extern "C" int emul_moments(const uint8_t* img, size_t stride, int W, int H, const float* points,
                            const int32_t* tris, int NT, float dp, float ratio, int64_t* mom,
                            int64_t* npairs) {
    tp_view vw;
    vw.dp = dp; vw.ratio = ratio; vw.halfW = 0.5f * (float)W; vw.halfH = 0.5f * (float)H; vw.W = W; vw.H = H;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    memset(mom, 0, sizeof(int64_t) * 6 * 13 * (size_t)NT);
    std::vector<uint64_t> P((size_t)TH * (TW + 1) * 2);  // entry = {x | y<<32, z | w<<32} like the LDS table
    int64_t pairs = 0;
    for (int ty = 0; ty < tiles_y; ty++)
        for (int tx = 0; tx < tiles_x; tx++) {
            // prefix table (zero outside the raster, like the padded device plane)
            for (int r = 0; r < TH; r++) {
                uint32_t x = 0, y = 0, z = 0, w = 0;
                for (int c = 0; c <= TW; c++) {
                    uint64_t* e = &P[((size_t)r * (TW + 1) + c) * 2];
                    e[0] = (uint64_t)x | ((uint64_t)y << 32); e[1] = (uint64_t)z | ((uint64_t)w << 32);
                    const int ar = ty * TH + r, ac = tx * TW + c;
                    if (c < TW && ar < H && ac < W) {
                        const uint8_t* p = img + (size_t)ar * stride + 4 * (size_t)ac;
                        const uint32_t R = p[0], G = p[1], B = p[2];
                        x += R; y += G; z += B | (((R + G + B) & 1u) << 20); w += (R * R + G * G + B * B) << 2;
                    }
                }
            }
            const int row0 = ty * TH, row1 = tp_min(row0 + TH - 1, H - 1);
            const int col0 = tx * TW, colE = tp_min(col0 + TW, W);
            for (int t = 0; t < NT; t++) {
                float p[3][2];
                for (int s = 0; s < 3; s++) { p[s][0] = points[2 * tris[4 * t + s]]; p[s][1] = points[2 * tris[4 * t + s] + 1]; }
                tp_bbox bb = tp_triangle_bbox(p, vw);
                if (bb.c0 > bb.c1 || bb.r0 > bb.r1) continue;
                if (bb.c1 / TW < tx || bb.c0 / TW > tx || bb.r1 / TH < ty || bb.r0 / TH > ty) continue;
                pairs++;
                for (int v = 0; v < 13; v++) {
                    int32_t X[3], Y[3];
                    for (int s = 0; s < 3; s++) tp_vertex_stage(p[s][0], p[s][1], v, s, vw, X[s], Y[s]);
                    tp_span sp;
        tp_setup_span(X, Y, row0, row1, sp);
                    uint64_t bxy = 0, bzw = 0, axy = 0, azw = 0;
                    uint32_t n = 0;
                    for (int r = sp.r0; r <= sp.r1; r++) {
                        int32_t lo, hi;
                        tp_span_row(sp, col0, colE, lo, hi);
                        const uint64_t* a = &P[((size_t)(r - row0) * (TW + 1) + (lo - col0)) * 2];
                        const uint64_t* b = &P[((size_t)(r - row0) * (TW + 1) + (hi - col0)) * 2];
                        n += hi - lo; bxy += b[0]; bzw += b[1]; axy += a[0]; azw += a[1];
                    }
                    const uint64_t dxy = bxy - axy, dzw = bzw - azw;
                    const uint32_t sr = (uint32_t)dxy, sg = (uint32_t)(dxy >> 32);
                    const uint32_t sb = (uint32_t)dzw & 0xfffffu, no = (uint32_t)(dzw >> 20) & 0x3fffu;
                    const uint32_t q = (uint32_t)(dzw >> 34);
                    int64_t* m = mom + 6 * ((size_t)v * NT + t);
                    m[0] += n; m[1] += no; m[2] += sr; m[3] += sg; m[4] += sb; m[5] += q;
                }
            }
        }
    if (npairs) *npairs = pairs;
    return 0;
}

// finalize exactly as k_finalize does
extern "C" void emul_finalize(const int64_t* mom, int NT, int flavour, const int32_t* colors, int32_t* ten,
                              int32_t* cn, int32_t* ca) {
    for (int id = 0; id < 13 * NT; id++) {
        const int64_t* p = mom + 6 * (size_t)id;
        tp_moments m = {p[0], p[1], p[2], p[3], p[4], p[5]};
        int64_t E;
        if (flavour == 0) {
            E = tp_energy_triangulate(m);
            ca[4 * id] = tp_wrap32(m.sr); ca[4 * id + 1] = tp_wrap32(m.sg); ca[4 * id + 2] = tp_wrap32(m.sb); ca[4 * id + 3] = 0;
        } else {
            const int32_t* c = colors + 4 * (id % NT);
            E = tp_energy64(m, c[0], c[1], c[2]);
        }
        ten[id] = tp_wrap32(E); cn[id] = tp_wrap32(m.n);
    }
}

extern "C" float emul_reference_dp(int flavour, int NT) { return tp_reference_dp(flavour, NT); }

// walker values for rows 0..rows-1 (tests the exactness claim of tp_make_walker directly)
extern "C" void emul_walker(int64_t N0, int32_t step, int32_t d, int rows, int32_t* out) {
    tp_walker w = tp_make_walker(N0, step, d);
    for (int r = 0; r < rows; r++) { out[r] = tp_walker_value(w); w.x += w.s; }
}

// ---------------------------------------------------------------------------------------------
// The table path (k_prefix + k_lines + k_finalize): the per-image row prefix table in its packed 32-byte records
// (tp_prefix_pack), whole-line 24.40 walkers (tp_setup_line) stepped by `tl` rows at a time exactly as a lane of
// k_lines steps them, one record per (line, row) evaluated with tp_prefix_eval, and every variant as the signed sum
// of three line sums.
// ---------------------------------------------------------------------------------------------
extern "C" int emul_moments_table(const uint8_t* img, size_t stride, int W, int H, const float* points,
                                  const int32_t* tris, int NT, int NP, float dp, float ratio, int tl, int64_t* mom) {
    tp_view vw;
    vw.dp = dp; vw.ratio = ratio; vw.halfW = 0.5f * (float)W; vw.halfH = 0.5f * (float)H; vw.W = W; vw.H = H;
    const int NG = tp_prefix_groups(W), pitch = tp_prefix_pitch(W);
    std::vector<uint32_t> T((size_t)H * pitch * TP_PFX_WORDS, 0);
    for (int r = 0; r < H; r++) {
        uint32_t run[5] = {0, 0, 0, 0, 0};
        for (int g = 0; g < NG; g++) {
            uint32_t px[4] = {0, 0, 0, 0};
            const int npx = tp_min(4, W - 4 * g) < 0 ? 0 : tp_min(4, W - 4 * g);
            for (int i = 0; i < npx; i++) memcpy(&px[i], img + (size_t)r * stride + 4 * (size_t)(4 * g + i), 4);
            tp_prefix_pack(run, px, npx, &T[((size_t)r * pitch + g) * TP_PFX_WORDS]);
            for (int i = 0; i < npx; i++) {
                const uint32_t R = px[i] & 0xffu, G = (px[i] >> 8) & 0xffu, B = (px[i] >> 16) & 0xffu;
                run[0] += (R + G + B) & 1u; run[1] += R; run[2] += G; run[3] += B; run[4] += R * R + G * G + B * B;
            }
        }
    }
    std::vector<int32_t> vx((size_t)NP * 5), vy((size_t)NP * 5);
    for (int v = 0; v < NP; v++)
        for (int m = 0; m < 5; m++)
            tp_vertex_stage(points[2 * v], points[2 * v + 1], m, 0, vw, vx[v * 5 + m], vy[v * 5 + m]);
    std::map<std::pair<int, int>, int> eid;
    std::vector<std::pair<int, int>> edges;
    std::vector<int> he_edge(3 * (size_t)NT);
    for (int t = 0; t < NT; t++)
        for (int k = 0; k < 3; k++) {
            const int o = tris[4 * t + k], d = tris[4 * t + (k + 1) % 3];
            const std::pair<int, int> key(o < d ? o : d, o < d ? d : o);
            auto it = eid.find(key);
            if (it == eid.end()) { it = eid.emplace(key, (int)edges.size()).first; edges.push_back(key); }
            he_edge[3 * t + k] = it->second * 2 + (o != key.first ? 1 : 0);
        }
    std::vector<int64_t> Wt(edges.size() * 9 * 6, 0);
    for (size_t e = 0; e < edges.size(); e++) {
        const int u = edges[e].first, v = edges[e].second;
        tp_line ln[9];
        int rmin = 0x3fffffff;
        for (int ver = 0; ver < 9; ver++) {
            const int mu = ver >= 1 && ver <= 4 ? ver : 0, mv = ver >= 5 ? ver - 4 : 0;
            tp_setup_line(vx[u * 5 + mu], vy[u * 5 + mu], vx[v * 5 + mv], vy[v * 5 + mv], H, ln[ver]);
            if (ln[ver].ra <= ln[ver].rb) rmin = tp_min(rmin, ln[ver].ra);
        }
        for (int ver = 0; ver < 9; ver++) {
            int64_t* acc = &Wt[(e * 9 + ver) * 6];
            const tp_line& l = ln[ver];
            if (l.ra > l.rb) continue;
            for (int c = 0; c < tl; c++) {  // lane (line, chunk c) of k_lines
                const int base = rmin + c;
                const int first = base + ((l.ra - base + tl - 1) & -tl);
                int64_t x = l.x + (int64_t)(first - l.ra) * l.s;
                const int64_t xs = l.s * tl;
                for (int r = first; r <= l.rb; r += tl, x += xs) {
                    const int32_t xc = (int32_t)(x >> TP_LINE_FRAC);
                    const int32_t col = xc < 0 ? 0 : (xc > W ? W : xc);
                    if (col != tp_line_col(l, r, W)) return 1;  // stepping and direct evaluation agree
                    uint32_t nodd, R, G, B, Q;
                    tp_prefix_eval(&T[((size_t)r * pitch + (col >> 2)) * TP_PFX_WORDS], col, nodd, R, G, B, Q);
                    acc[0] += col; acc[1] += nodd; acc[2] += R; acc[3] += G; acc[4] += B; acc[5] += Q;
                }
            }
        }
    }
    for (int t = 0; t < NT; t++)
        for (int i = 0; i < 13; i++) {
            int32_t X[3], Y[3], c[3];
            for (int s = 0; s < 3; s++) {
                const int v = tris[4 * t + s];
                const int m = (i > 0 && ((i - 1) >> 2) == s) ? ((i - 1) & 3) + 1 : 0;
                X[s] = vx[v * 5 + m]; Y[s] = vy[v * 5 + m];
            }
            tp_variant_coeffs(X, Y, c);
            int64_t* m = mom + 6 * ((size_t)i * NT + t);
            for (int q = 0; q < 6; q++) m[q] = 0;
            for (int k = 0; k < 3; k++) {
                const int he = he_edge[3 * t + k];
                const int64_t* w = &Wt[((size_t)(he >> 1) * 9 + tp_edge_version(i, k, he & 1)) * 6];
                for (int q = 0; q < 6; q++) m[q] += c[k] * w[q];
            }
        }
    return 0;
}

// the packed prefix records against plain sums: every column c = 0..W of every row; returns the mismatches
extern "C" int emul_prefix_check(const uint8_t* img, size_t stride, int W, int H) {
    const int NG = tp_prefix_groups(W);
    int bad = 0;
    for (int r = 0; r < H; r++) {
        uint32_t run[5] = {0, 0, 0, 0, 0};
        uint64_t ref[5] = {0, 0, 0, 0, 0};
        int c = 0;
        for (int g = 0; g < NG; g++) {
            uint32_t px[4] = {0, 0, 0, 0}, rec[TP_PFX_WORDS];
            const int npx = tp_min(4, W - 4 * g) < 0 ? 0 : tp_min(4, W - 4 * g);
            for (int i = 0; i < npx; i++) memcpy(&px[i], img + (size_t)r * stride + 4 * (size_t)(4 * g + i), 4);
            tp_prefix_pack(run, px, npx, rec);
            for (int k = 0; k < 4 && c <= W; k++, c++) {
                if (c != 4 * g + k) return -1;
                uint32_t nodd, R, G, B, Q;
                tp_prefix_eval(rec, c, nodd, R, G, B, Q);
                if (nodd != ref[0] || R != ref[1] || G != ref[2] || B != ref[3] || Q != ref[4]) bad++;
                if (k < npx) {
                    const uint64_t r8 = px[k] & 0xffu, g8 = (px[k] >> 8) & 0xffu, b8 = (px[k] >> 16) & 0xffu;
                    ref[0] += (r8 + g8 + b8) & 1; ref[1] += r8; ref[2] += g8; ref[3] += b8; ref[4] += r8 * r8 + g8 * g8 + b8 * b8;
                }
            }
            for (int i = 0; i < npx; i++) {
                const uint32_t R = px[i] & 0xffu, G = (px[i] >> 8) & 0xffu, B = (px[i] >> 16) & 0xffu;
                run[0] += (R + G + B) & 1u; run[1] += R; run[2] += G; run[3] += B; run[4] += R * R + G * G + B * B;
            }
        }
        if (c != W + 1) return -2;
    }
    return bad;
}

// whole-line walker against exact integer arithmetic: rows ra..rb of the line through (Xa,Ya)-(Xb,Yb), evaluated
// directly (tp_line_col without the clamp) and stepped by `tl` rows as k_lines steps it.  Returns the mismatches.
extern "C" int emul_line_check(int32_t Xa, int32_t Ya, int32_t Xb, int32_t Yb, int32_t H, int tl, int32_t* first_bad) {
    tp_line ln;
    tp_setup_line(Xa, Ya, Xb, Yb, H, ln);
    const bool swap = Ya > Yb;
    const int64_t Xt = swap ? Xb : Xa, Yt = swap ? Yb : Ya, Xq = swap ? Xa : Xb, Yq = swap ? Ya : Yb;
    const int64_t dy = Yq - Yt, dx = Xq - Xt;
    int bad = 0;
    // expected row range
    int64_t era = (Yt - 128 + 255) >> 8, erb = ((Yq - 128 + 255) >> 8) - 1;
    if (era < 0) era = 0;
    if (erb > H - 1) erb = H - 1;
    if (dy <= 0 || era > erb) return ln.ra > ln.rb ? 0 : 1;
    if (ln.ra != era || ln.rb != erb) { if (first_bad) *first_bad = -1; return 1; }
    for (int c = 0; c < tl; c++) {
        int64_t x = ln.x + (int64_t)c * ln.s;
        const int64_t xs = ln.s * tl;
        for (int64_t r = era + c; r <= erb; r += tl, x += xs) {
            const int32_t got = (int32_t)(x >> TP_LINE_FRAC);
            const int32_t direct = (int32_t)((ln.x + (r - ln.ra) * ln.s) >> TP_LINE_FRAC);
            // first column c with (256 c + 128 - Xt) dy >= dx (256 r + 128 - Yt)
            const __int128 rhs = (__int128)dx * (256 * r + 128 - Yt) - (__int128)(128 - Xt) * dy;  // 256 dy c >= rhs
            const __int128 den = (__int128)256 * dy;
            __int128 q = rhs / den;
            if (q * den < rhs) q++;          // ceil for positive remainder
            while ((q - 1) * den >= rhs) q--;  // and for negative quotients (truncation toward zero)
            if ((__int128)got != q || got != direct) { if (!bad && first_bad) *first_bad = (int32_t)r; bad++; }
        }
    }
    return bad;
}


// the pixel records (16 bytes per pixel column, fields with headroom) against plain sums: every column c = 0..W of every
// row on its own, and sums of up to 16 records at pseudo-random columns; returns the mismatches
extern "C" int emul_px_check(const uint8_t* img, size_t stride, int W, int H) {
    int bad = 0;
    std::vector<uint64_t> recs(2 * (size_t)(W + 1));
    std::vector<uint64_t> ref(5 * (size_t)(W + 1));
    uint32_t seed = 12345u + (uint32_t)W * 31u + (uint32_t)H;
    for (int r = 0; r < H; r++) {
        uint32_t run[5] = {0, 0, 0, 0, 0};
        for (int c = 0; c <= W; c++) {
            tp_px_pack(run, &recs[2 * (size_t)c]);
            for (int k = 0; k < 5; k++) ref[5 * (size_t)c + k] = run[k];
            if (c < W) {
                uint32_t px;
                memcpy(&px, img + (size_t)r * stride + 4 * (size_t)c, 4);
                const uint32_t R = px & 0xffu, G = (px >> 8) & 0xffu, B = (px >> 16) & 0xffu;
                run[0] += (R + G + B) & 1u; run[1] += R; run[2] += G; run[3] += B; run[4] += R * R + G * G + B * B;
            }
        }
        for (int c = 0; c <= W; c++) {
            uint32_t nodd, R, G, B; uint64_t Q;
            tp_px_unpack(recs[2 * (size_t)c], recs[2 * (size_t)c + 1], nodd, R, G, B, Q);
            const uint64_t* e = &ref[5 * (size_t)c];
            if (nodd != e[0] || R != e[1] || G != e[2] || B != e[3] || Q != e[4]) bad++;
        }
        for (int trial = 0; trial < 8; trial++) {   // sums of 1..16 records, the widest columns included
            uint64_t lo = 0, hi = 0, e[5] = {0, 0, 0, 0, 0};
            const int n = 1 + (trial * 5) % TP_PX_MAXSUM;
            for (int k = 0; k < n; k++) {
                seed = seed * 1664525u + 1013904223u;
                const int c = (trial & 1) ? W - (int)((seed >> 8) % (uint32_t)(W < 4 ? W + 1 : 4)) : (int)((seed >> 8) % (uint32_t)(W + 1));
                lo += recs[2 * (size_t)c]; hi += recs[2 * (size_t)c + 1];
                for (int q = 0; q < 5; q++) e[q] += ref[5 * (size_t)c + q];
            }
            uint32_t nodd, R, G, B; uint64_t Q;
            tp_px_unpack(lo, hi, nodd, R, G, B, Q);
            if (nodd != e[0] || R != e[1] || G != e[2] || B != e[3] || Q != e[4]) bad++;
        }
    }
    return bad;
}
