// tp_kernels.h -- launch interface between the C ABI (tp_context.hip) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "tp_raster.h"

#define TP_TILE_W 128
#define TP_TILE_H 16
#define TP_NLINES 9        /* lines per undirected edge: base + 4 moves of either endpoint */
#define TP_W_WORDS 6       /* values per line sum: sum x, n_odd, sum r, sum g, sum b, q */
#define TP_REC_DWORDS 8    /* per-tile record of a line, 32 bytes: u32 sum x (absolute columns), then the TILE-LOCAL sums
                              n_odd, r, g, b, q + n_odd over its counted rows (<= 16 rows x 128 columns: < 2^29), the
                              number of the sweep that wrote it (a record is valid for that sweep only), 0 */
#define TP_T2_WORDS 5      /* int64 per static-table entry: n_odd, sum r, sum g, sum b, q */
#define TP_COUNT_STRIDE 32  /* ints between per-tile list counters: atomics on one memory line serialise, whatever the word */

// device-side flag bits (tp_device_state::flags)
#define TP_FLAG_LIST_OVERFLOW 1u
#define TP_FLAG_VISIT_OVERFLOW 2u

struct tp_device_state {
    uint32_t visit_total;  // (edge, tile) visits drawn from the shared half of the record buffer
    uint32_t flags;        // sticky overflow flags
    uint32_t rebin_req;    // 1: k_bin must rebuild the work lists (always, except for profiling replays of one sweep)
    uint32_t rebin_count;  // statistics: rebuilds so far
    uint32_t iters_done;   // fused iterations k_update completed (it does not step while a flag is up)
    uint32_t sweep;        // number of the current sweep: k_bin counts, k_accumulate stamps its records, readers compare
    unsigned long long pad;
};

struct tp_launch {
    // raster
    const uint8_t* img;  // padded RGBA8 plane; the alpha byte holds (r + g + b) & 1 (tp_set_image rewrites it)
    int pitch;           // bytes per padded row
    const int64_t* t2;   // static table of the swept image: [H+1][tiles_x+1][TP_T2_WORDS]
    tp_view vw;
    int tiles_x, tiles_y;
    // triangulation
    float2* points;
    const int4* tris;
    const int4* colors;     // stored colours ivec4[NT] (warp) -- may be null
    int NT, NP, NE;
    const int* vtx_off;     // CSR by origin vertex: half-edge ids 3t+s
    const int* vtx_adj;
    const int2* edge_uv;    // [NE] endpoints of every undirected edge, u <= v; bit 30: this edge publishes the vertex (vpos)
    const int* he_edge;     // [3 NT] edge id * 2 + (half-edge runs v -> u)
    const int* vref;        // [NP][64] per upload: the line (edge << 4 | version) each lane of k_update sums, -1 none; see k_vertex_refs
    const int* vvar;        // [NP][8]  per upload: per incident triangle 3t + s | out-edge slot << 20 | in-edge slot << 24, -1 none
    int2* vpos;             // [NP][5] snapped 24.8 position of every vertex: unmoved, +dx, -dx, +dy, -dy
    float2* epos;           // [NE][2] the positions of every edge's two endpoints (kept by whoever moves a vertex)
    int64_t* line_static;   // [NE][TP_NLINES][TP_T2_WORDS] static part of the line sums: everything left of the tile
                            // column in each of the line's rows (differences of t2 per column run)
    // work lists
    int* tilecount;           // [tiles][TP_COUNT_STRIDE] entries per tile (word 0 of a 128-byte line of its own)
    int4* tilelist;           // [tiles * list_cap][2] 32-byte entries, LIVE lines only: the line's 24.40 walker (x at row ra, step
                              // per row), its rows (ra, rb) inside the raster, its record = visit * 9 + version, 0
    int list_cap;
    int2* edge_visit;         // [NE] (first visit, #visits = tiles the band of the edge's lines can touch)
    uint32_t* visits;         // [visit_cap][TP_NLINES][TP_REC_DWORDS] per-tile line records; only live lines are written,
                              // the others keep the stamp of an older sweep
    int visit_cap;
    int64_t* wline;           // [NE][TP_NLINES][TP_W_WORDS] whole line sums -- only for coarse meshes (k_linesum), else null
    tp_device_state* state;
    // outputs (reference layout)
    int32_t* ten;
    int32_t* cn;
    int4* ca;
    int2* gr;
    int64_t* moments;          // optional int64[13NT][6]
    unsigned long long* gacc;  // [NP][2] fused-update accumulators: (gradient component << 32) | arrivals
#ifdef TPOSE_DEBUG
    unsigned long long* dbg;   // per-block phase timestamps (debug flavour of the library only)
#endif
};

void tp_launch_bin(const tp_launch& L, hipStream_t s);
void tp_launch_accumulate(const tp_launch& L, hipStream_t s);
void tp_launch_accumulate_timed(const tp_launch& L, hipStream_t s, hipEvent_t start, hipEvent_t stop);
bool tp_coarse_mesh(const tp_launch& L);   // hundreds of tiles per edge: line sums by k_linesum (needs L.wline)
void tp_launch_linesum(const tp_launch& L, hipStream_t s);
void tp_launch_finalize(const tp_launch& L, int flavour, bool write_moments, hipStream_t s);
void tp_launch_shift(const tp_launch& L, float rate, hipStream_t s);
void tp_launch_update(const tp_launch& L, int flavour, float rate, hipStream_t s);
void tp_launch_vertex_refs(const tp_launch& L, int* vref, int* vvar, hipStream_t s);  // once per upload
void tp_launch_replicate_colors(const tp_launch& L, hipStream_t s);
// static per-image table: t2[r][tc] = moments of all pixels in rows < r and columns < tc * TP_TILE_W
// also rewrites the alpha bytes of the padded plane (pixel parity)
void tp_launch_static_table(uint8_t* img, int pitch, int W, int H, int Hp, int tiles_x, uint32_t* seg_scratch,
                            int64_t* t2, hipStream_t s);
size_t tp_accumulate_lds_bytes();
hipError_t tp_kernels_init();  // per-device function attributes
void tp_launch_selftest_walker(const int64_t* N0, const int32_t* step, const int32_t* d, int n, int32_t* out, hipStream_t s);
void tp_launch_selftest_line(const int4* ends, const int* H, int n, int rows, int32_t* out, hipStream_t s);
void tp_launch_render(const tp_launch& L, const float2* pts, int source, void* out, int out_pitch_px, hipStream_t s);
