// CPU emulation of the k_accumulate / k_finalize data flow (tests only, never shipped):
// runs the SAME per-lane integer logic (tpose_amd/csrc/tp_raster.h) tile by tile, with the LDS
// prefix table as a plain array.  Lets the span walker be checked against the oracle without a GPU.
#include <stdint.h>
#include <string.h>
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
// Edge-centric form: W for every distinct edge line (walked in 32-row windows like the tiles do,
// one full-row prefix lookup per row), then each variant = signed sum of three lines.
// ---------------------------------------------------------------------------------------------
#include <map>
extern "C" int emul_moments_edges(const uint8_t* img, size_t stride, int W, int H, const float* points,
                                  const int32_t* tris, int NT, int NP, float dp, float ratio, int64_t* mom) {
    tp_view vw;
    vw.dp = dp; vw.ratio = ratio; vw.halfW = 0.5f * (float)W; vw.halfH = 0.5f * (float)H; vw.W = W; vw.H = H;
    // full-row prefix sums of the six per-pixel quantities (index 0: pixel count -> P = x itself)
    std::vector<int64_t> P((size_t)H * (W + 1) * 5, 0);
    for (int r = 0; r < H; r++)
        for (int c = 0; c < W; c++) {
            const uint8_t* p = img + (size_t)r * stride + 4 * (size_t)c;
            const int64_t R = p[0], G = p[1], B = p[2];
            const int64_t v[5] = {(R + G + B) & 1, R, G, B, R * R + G * G + B * B};
            for (int k = 0; k < 5; k++) P[((size_t)r * (W + 1) + c + 1) * 5 + k] = P[((size_t)r * (W + 1) + c) * 5 + k] + v[k];
        }
    // per-vertex snapped positions for the five moves
    std::vector<int32_t> vx((size_t)NP * 5), vy((size_t)NP * 5);
    for (int v = 0; v < NP; v++)
        for (int m = 0; m < 5; m++)
            tp_vertex_stage(points[2 * v], points[2 * v + 1], m == 0 ? 0 : m, 0, vw, vx[v * 5 + m], vy[v * 5 + m]);
    // undirected edges
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
    // W[edge][ver][6]
    std::vector<int64_t> Wt(edges.size() * 9 * 6, 0);
    for (size_t e = 0; e < edges.size(); e++)
        for (int ver = 0; ver < 9; ver++) {
            const int u = edges[e].first, v = edges[e].second;
            const int mu = ver >= 1 && ver <= 4 ? ver : 0, mv = ver >= 5 ? ver - 4 : 0;
            int64_t* w = &Wt[(e * 9 + ver) * 6];
            for (int win = 0; win < H; win += 32) {
                tp_edge_walk ew;
                tp_setup_edge(vx[u * 5 + mu], vy[u * 5 + mu], vx[v * 5 + mv], vy[v * 5 + mv], win, tp_min(win + 31, H - 1), ew);
                for (int r = ew.ra; r <= ew.rb; r++) {
                    int32_t x = tp_walker_value(ew.w);
                    ew.w.x += ew.w.s;
                    x = x < 0 ? 0 : (x > W ? W : x);
                    w[0] += x;
                    for (int k = 0; k < 5; k++) w[1 + k] += P[((size_t)r * (W + 1) + x) * 5 + k];
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

// ---------------------------------------------------------------------------------------------
// Round-2 structure: whole-line 24.40 walkers (tp_setup_line), per-edge tile enumeration by tile row
// (tp_band_rows / tp_band_cols), and per (line, tile) the derived 32.32 walker (tp_line_at) with the
// kernel's own liveness tests.  A tile the band misses loses its contribution, so equality with the
// oracle also proves the enumeration conservative.  `margin` inflates the band like tp_set_margin.
// ---------------------------------------------------------------------------------------------
extern "C" int emul_moments_lines(const uint8_t* img, size_t stride, int W, int H, const float* points,
                                  const int32_t* tris, int NT, int NP, float dp, float ratio, int tile_w, int tile_h,
                                  int margin, int64_t* mom, int64_t* nvisits) {
    tp_view vw;
    vw.dp = dp; vw.ratio = ratio; vw.halfW = 0.5f * (float)W; vw.halfH = 0.5f * (float)H; vw.W = W; vw.H = H;
    const int tiles_x = (W + tile_w - 1) / tile_w, tiles_y = (H + tile_h - 1) / tile_h;
    std::vector<int64_t> P((size_t)H * (W + 1) * 5, 0);
    for (int r = 0; r < H; r++)
        for (int c = 0; c < W; c++) {
            const uint8_t* p = img + (size_t)r * stride + 4 * (size_t)c;
            const int64_t R = p[0], G = p[1], B = p[2];
            const int64_t v[5] = {(R + G + B) & 1, R, G, B, R * R + G * G + B * B};
            for (int k = 0; k < 5; k++) P[((size_t)r * (W + 1) + c + 1) * 5 + k] = P[((size_t)r * (W + 1) + c) * 5 + k] + v[k];
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
    int64_t visits = 0;
    // cumulative static table like the device's t2: T2[r][tc] = moments of rows < r, columns < tc * tile_w
    std::vector<int64_t> T2((size_t)(H + 1) * (tiles_x + 1) * 5, 0);
    for (int r = 0; r < H; r++)
        for (int tc = 0; tc <= tiles_x; tc++) {
            const int c = tp_min(tc * tile_w, W);
            for (int k = 0; k < 5; k++)
                T2[((size_t)(r + 1) * (tiles_x + 1) + tc) * 5 + k] = T2[((size_t)r * (tiles_x + 1) + tc) * 5 + k] + P[((size_t)r * (W + 1) + c) * 5 + k];
        }
    for (size_t e = 0; e < edges.size(); e++) {
        const int u = edges[e].first, v = edges[e].second;
        tp_line ln[9];
        tp_band b;
        b.Xa = vx[u * 5]; b.Ya = vy[u * 5]; b.Xb = vx[v * 5]; b.Yb = vy[v * 5];
        b.dX = 0; b.dY = 0;
        for (int m = 1; m < 5; m++) {
            b.dX = tp_max(b.dX, tp_max(abs(vx[u * 5 + m] - b.Xa), abs(vx[v * 5 + m] - b.Xb)));
            b.dY = tp_max(b.dY, tp_max(abs(vy[u * 5 + m] - b.Ya), abs(vy[v * 5 + m] - b.Yb)));
        }
        b.dX += 256 * margin; b.dY += 256 * margin;
        for (int ver = 0; ver < 9; ver++) {
            const int mu = ver >= 1 && ver <= 4 ? ver : 0, mv = ver >= 5 ? ver - 4 : 0;
            tp_setup_line(vx[u * 5 + mu], vy[u * 5 + mu], vx[v * 5 + mv], vy[v * 5 + mv], H, ln[ver]);
            // static part: per run of rows inside one tile column, a difference of the cumulative table
            int64_t* acc = &Wt[(e * 9 + ver) * 6];
            tp_line_column_runs(ln[ver], W, tile_w, tiles_x, [&](int32_t tc, int32_t ra, int32_t rb) {
                for (int k = 0; k < 5; k++)
                    acc[1 + k] += T2[((size_t)(rb + 1) * (tiles_x + 1) + tc) * 5 + k] - T2[((size_t)ra * (tiles_x + 1) + tc) * 5 + k];
            });
        }
        int32_t r0, r1;
        tp_band_rows(b, H, r0, r1);
        if (r0 > r1) continue;
        for (int ty = r0 / tile_h; ty <= r1 / tile_h; ty++) {
            int32_t tx0, tx1;
            const int row0 = ty * tile_h, row1 = tp_min(row0 + tile_h - 1, H - 1);
            if (!tp_band_cols(b, row0, row1, W, tile_w, tiles_x, tx0, tx1)) continue;
            for (int tx = tx0; tx <= tx1; tx++) {
                visits++;
                const int col0 = tx * tile_w;
                const int lim = tx == tiles_x - 1 ? W - col0 + 1 : tile_w;
                for (int ver = 0; ver < 9; ver++) {
                    if (!tp_line_live(ln[ver], row0, row1, col0, lim, W)) continue;  // k_bin lists live lines only
                    tp_walker w = tp_line_at(ln[ver], row0);
                    const int koff = ln[ver].ra - row0;
                    const uint32_t nvalid = (uint32_t)tp_max(ln[ver].rb - ln[ver].ra + 1, 0);
                    int64_t* acc = &Wt[(e * 9 + ver) * 6];
                    for (int j = 0; j < tile_h; j++) {
                        int32_t x = tp_walker_value(w);
                        w.x += w.s;
                        x = x < 0 ? 0 : (x > W ? W : x);
                        const uint32_t xl = (uint32_t)(x - col0);
                        const bool in = xl < (uint32_t)lim && (uint32_t)(j - koff) < nvalid;
                        if (!in) continue;
                        acc[0] += x;
                        // tile-local prefix = full-row prefix minus everything left of the tile column
                        for (int k = 0; k < 5; k++)
                            acc[1 + k] += P[((size_t)(row0 + j) * (W + 1) + x) * 5 + k] - P[((size_t)(row0 + j) * (W + 1) + col0) * 5 + k];
                    }
                }
            }
        }
    }
    if (nvisits) *nvisits = visits;
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

// whole-line walker against exact integer arithmetic: rows ra..rb of the line through (Xa,Ya)-(Xb,Yb),
// evaluated tile by tile (tile_h rows) exactly as the kernel does.  Returns the number of mismatches.
extern "C" int emul_line_check(int32_t Xa, int32_t Ya, int32_t Xb, int32_t Yb, int32_t H, int tile_h, int32_t* first_bad) {
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
    for (int row0 = (int)(era / tile_h) * tile_h; row0 <= erb; row0 += tile_h) {
        tp_walker w = tp_line_at(ln, row0);
        for (int j = 0; j < tile_h; j++) {
            const int64_t r = row0 + j;
            const int32_t got = tp_walker_value(w);
            w.x += w.s;
            if (r < era || r > erb) continue;
            // first column c with (256 c + 128 - Xt) dy >= dx (256 r + 128 - Yt)
            const __int128 rhs = (__int128)dx * (256 * r + 128 - Yt) - (__int128)(128 - Xt) * dy;  // 256 dy c >= rhs
            const __int128 den = (__int128)256 * dy;
            __int128 c = rhs / den;
            if (c * den < rhs) c++;          // ceil for positive remainder
            while ((c - 1) * den >= rhs) c--;  // and for negative quotients (truncation toward zero)
            if ((__int128)got != c) { if (!bad && first_bad) *first_bad = (int32_t)r; bad++; }
        }
    }
    return bad;
}
