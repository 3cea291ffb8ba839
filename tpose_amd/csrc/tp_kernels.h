// tp_kernels.h -- launch interface between the C ABI (tp_context.hip) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "tp_raster.h"
#include "tp_plan.h"

#define TP_NLINES 9        /* lines per undirected edge: base + 4 moves of either endpoint */
#define TP_W_WORDS 6       /* values per line sum: sum x, n_odd, sum r, sum g, sum b, q */

struct tp_launch {
    // raster
    const uint8_t* img;   // RGBA8 plane (tp_render only)
    int pitch;            // bytes per row of img
    const uint4* prefix;  // row prefix table of the swept image: [H][prefix_pitch] 32-byte records (tp_raster.h), two uint4 each
    int prefix_pitch;     // records per row (tp_prefix_pitch)
    tp_view vw;
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
    int64_t* wline;         // [NE][TP_NLINES][TP_W_WORDS] whole line sums (k_lines)
    int lanes_per_line;     // k_lines: lanes sharing the rows of one line (power of two, 1..64)
    // outputs (reference layout)
    int32_t* ten;
    int32_t* cn;
    int4* ca;
    int2* gr;
    int64_t* moments;          // optional int64[13NT][6]
    // frame mirror (single frames of the schedules: tp_iterate(1) then tp_retrieve_many): the kernels that write `tenergy`, `colnum`
    // and `points` also write the first mirror_n entries / every point into pinned host memory -- the read-back needs no copy
    int32_t* mirror_ten; int32_t* mirror_cn; float2* mirror_pts; int mirror_n;
#ifdef TPOSE_DEBUG
    unsigned long long* dbg;   // per-block phase timestamps (debug flavour of the library only)
#endif
};

void tp_launch_lines(const tp_launch& L, hipStream_t s, hipEvent_t start = nullptr, hipEvent_t stop = nullptr);  // events: the dispatch's own timestamps
void tp_launch_finalize(const tp_launch& L, int flavour, bool write_moments, hipStream_t s);
void tp_launch_shift(const tp_launch& L, float rate, hipStream_t s);
void tp_launch_update(const tp_launch& L, int flavour, float rate, hipStream_t s);
void tp_launch_vertex_refs(const tp_launch& L, int* vref, int* vvar, hipStream_t s);  // once per upload
void tp_launch_replicate_colors(const tp_launch& L, hipStream_t s);
// per-image row prefix table (tp_raster.h, "Per-image row prefix table")
void tp_launch_prefix_table(const uint8_t* img, int pitch, int W, int H, int prefix_pitch, uint4* P, hipStream_t s);
void tp_launch_px_table(const uint8_t* img, int pitch, int W, int H, int px_pitch, uint4* P, uint4* P_tiled, hipStream_t s);  // rasters up to TP_PX_MAXW columns
void tp_launch_selftest_walker(const int64_t* N0, const int32_t* step, const int32_t* d, int n, int32_t* out, hipStream_t s);
void tp_launch_selftest_line(const int4* ends, const int* H, int n, int rows, int32_t* out, hipStream_t s);
// several arrays copied by ONE launch, either side of which may be pinned host memory (read or written by the device across
// the link): the read-backs of a frame and the tables of an upload -- one dispatch instead of one copy command per array
#define TP_COPY_MAX 8
struct tp_copy_list { const uint32_t* src[TP_COPY_MAX]; uint32_t* dst[TP_COPY_MAX]; uint32_t words[TP_COPY_MAX]; int n; };
void tp_launch_copy_list(const tp_copy_list& G, hipStream_t s);
void tp_launch_frame_sums(const int32_t* ering, int C, int NT, float* out, hipStream_t s);   // out: device-visible (pinned) float[C]
void tp_launch_render(const tp_launch& L, const float2* pts, int source, void* out, int out_pitch_px, hipStream_t s);

// ---- persistent grad-iter kernel (tp_persist.hip): K grad-iters per launch, one workgroup per patch of the mesh
#define PK_DBG_ITERS 64
#define PK_DBG_WITERS 32                                /* grad-iters with per-wave stamps (debug flavour) */
#define PK_DBG_WBASE ((size_t)512 * PK_DBG_ITERS * 16) /* ... which follow the per-workgroup stamps in the debug buffer */
#define PK_MAX_PEERS 3
struct pk_args {
    const pk_wg* wg;            // [parts] per-patch headers (tp_plan.h)
    const int32_t* pool;        // the patches' tables
    int parts;
    tp_view vw;
    const uint4* px;            // row prefix table of the swept image in pixel records (tp_raster.h): [H][px_pitch] 16 bytes
    const uint4* px_tiled;      // the same records tiled 4 rows x 2 columns per 128 bytes: what lane-items without kept records read
    int px_pitch;
    const float2* points;       // positions at the start of the launch
    float2* points_out;         // positions after n_iters grad-iters (vertices of at least one triangle only)
    const int4* ca;             // warp flavour: the stored colours, replicated x13 by upload (`colacc` as it stands)
    int emit;                   // the last grad-iter of this launch writes the reference's buffers:
    int32_t* ten; int32_t* cn; int4* ca_out; int2* gr;   // `tenergy`, `colnum`, `colacc` (triangulate), `gradient`
    int NT, NP, NE;
    int flavour;
    float rate;
    unsigned long long* posbox;   // position mailbox: [4][box_stride] slots of two 8-byte granules {tag : 32, float : 32} -- slots 0, 1: the
                                  // positions entering a grad-iter, by its parity; 2, 3: those a launch ends with, by the launch's number
    unsigned box_stride;          // vertices per slot array
    // band split (tp_band_attach): this launch runs the patches [part0, part0 + gridDim.x) of the plan; every position is also
    // posted to the other bands' mailboxes (peer memory: system-scope stores), polled from the own one
    int part0, n_peers;
    unsigned long long* peer_box[PK_MAX_PEERS];
    unsigned final_tag, final_slot;   // band split: tag of the positions this launch ends with, and which of the slots 2, 3 takes them (the
                                      // launch's number mod 2: a band cannot end launch n + 2 before every band has collected launch n, because its
                                      // launch n + 1 needs the others' launch n + 1, which they start after collecting n)
    int32_t* ering;               // tp_iterate_until: [n_iters][NT] energy of every triangle's base variant, frame by frame (or null)
    float2* pring;                // tp_iterate_until: [n_iters][NP] positions at the START of every frame (vertices of triangles)
    int32_t* peer_ering[PK_MAX_PEERS];   // band split: the other bands' rings -- every band's host applies the convergence test to
    float2* peer_pring[PK_MAX_PEERS];    // ALL triangles' energies, so what a band writes into its own rings it writes into theirs
    unsigned epoch;               // number of the launch's first grad-iter (tags carry 31 bits of it)
    int n_iters;                  // < 0: census of resident workgroups instead
    unsigned* status;             // [0] raised by a lane that gave up waiting, [1] census counter, [2] completed launches, [3] ticket counter
    // A launch that FINISHES ITSELF (plain tp_iterate, every vertex used by a triangle): `points_out` is the context's OTHER position buffer
    // -- the host swaps the two behind every launch -- and the last workgroup to finish counts the launch as completed and mirrors the
    // words into pinned memory; no k_persist_finish behind it (a launch + 5 us less per call).  A launch that gives up leaves `points` whole.
    unsigned* host_status;        // pinned mirror {gave up, -, launches completed}; null: k_persist_finish follows
    int inject_give_up;           // tests (TP_OPT_INJECT_GIVE_UP): one workgroup gives up before the last grad-iter
    // carry: [parts][carry_stride] words a launch leaves for the next one on the same plan -- {tag, grad-iters since the lines were cut, rows per
    // lane, lane-items, lane-items with the base lines, -, -, -}, the cut (first lane-item of every line), every thread's lane-item {line, chunk,
    // chunks}; null: none.  A workgroup whose words carry carry_tag starts from them (tp_persist.hip)
    int32_t* carry; int carry_stride; unsigned carry_tag; int carry_cut_cap;
    // what the planner weighs a vertex's rows by (tp_plan.h: pk_vertex_work): [NP][2] the mean |step| per grad-iter of every vertex over THIS
    // launch, per axis, in t-pose units -- written at the launch's end by the vertex's owner; null: not wanted
    float* vspeed;
#ifdef TPOSE_DEBUG
    unsigned long long* dbg;      // [parts][PK_DBG_ITERS][16] phase timestamps of the grad-iters dbg_first ...
    int dbg_first;
#endif
};
#ifdef PK_DBG_BOUNDS
int tp_persist_debug_faults(unsigned long long out[16]);   // debug flavour: table offsets beyond the table seen by k_persist
#endif
#ifdef PK_DBG_STALE
int tp_persist_debug_counts(unsigned long long* out, int reset);   // counting flavour: what k_persist counted, [512][4]
int tp_persist_debug_vcounts(unsigned long long* out, int n, int reset);   // ... and per vertex, [n][2]
#endif
int tp_persist_set_lds(int bytes);  // hipFuncSetAttribute(max dynamic LDS); returns the hipError_t
void tp_launch_persist(const pk_args& A, int grid, int rows, int lds_bytes, hipStream_t s);   // grid: workgroups = patches of this launch
// band split: the positions the launch ended with, from the own mailbox (every band posted there) into points_out; raises
// status[0] when they do not arrive
void tp_launch_band_collect(const tp_launch& L, const pk_args& A, float2* points_out, hipStream_t s);
void tp_launch_persist_finish(const tp_launch& L, const float2* points_out, unsigned* status, unsigned* host_status, int file_edges, hipStream_t s);
void tp_launch_publish_positions(const tp_launch& L, hipStream_t s);   // every vertex files its position with its edges (`epos`)
// (status: of the launch before, or null; host_status: pinned mirror of {gave up, -, launches completed})
