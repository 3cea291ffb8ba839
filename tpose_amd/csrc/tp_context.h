// tp_context.h -- the context behind the C ABI of include/tpose_hip.h and what its translation units share.
// Host side of the boundary that replaces tpose::init/quit/upload and the computecolors/doenergy/doshift lambdas of the
// reference (source/triangulation.hpp:576-643, software/triangulate/main.cpp:121-155, software/warp/main.cpp:140-178).
//   tp_context.hip       context, images, uploads, the piecewise API, the two-kernel path and its graphs, tp_iterate
//   tp_persist_host.hip  persistent launches: status words and replay, census, plans, tp_iterate_until
//   tp_replan.hip        re-planning while a descent runs (calling thread between chunks, worker thread after a call)
//   tp_bands.hip         band split of one descent over several GPUs: mailboxes, tp_band_attach
//   tp_readback.hip      tp_retrieve / tp_retrieve_many, the frame mirror, tp_render
#pragma once
#include "../../include/tpose_hip.h"
#include "tp_kernels.h"

#ifndef TP_LINES_ROWS
#define TP_LINES_ROWS 8  /* k_lines: rows per lane the number of groups per edge aims for */
#endif

#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#include <string>
#include <unordered_map>
#include <vector>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <thread>
#include <mutex>
#include <shared_mutex>

// Stream capture is process-wide state in the HIP runtime: while one host thread captures a graph, allocations, frees and
// synchronous copies issued by OTHER threads (other contexts) invalidate the capture, whatever the capture mode.  Every
// entry point therefore holds a shared lock, and a capture upgrades to the exclusive one: captures are rare (once per
// upload / parameter set), so contexts driven from different threads still run concurrently.
extern std::shared_mutex g_api_mutex;
extern thread_local int g_api_depth;
struct api_guard {  // outermost entry point of this thread takes the shared lock (entry points call each other)
    api_guard() { if (g_api_depth++ == 0) g_api_mutex.lock_shared(); }
    ~api_guard() { if (--g_api_depth == 0) g_api_mutex.unlock_shared(); }
};
struct capture_guard {  // inside an entry point: trade the shared lock for the exclusive one
    capture_guard() { g_api_mutex.unlock_shared(); g_api_mutex.lock(); }
    ~capture_guard() { g_api_mutex.unlock(); g_api_mutex.lock_shared(); }
};

struct graph_entry {
    hipGraphExec_t exec = nullptr;
    const void* points = nullptr;   // (the position buffer the captured kernels address: the context has two, swapped by launches that finish themselves)
    tp_params params{};
    int iters = 0;
    uint64_t generation = 0;
};

struct tp_context {
    int device = 0;
    int W = 0, H = 0;
    float ratio = 1.0f;
    hipStream_t stream = nullptr;
    std::string error;

    uint8_t* img[2] = {nullptr, nullptr};
    bool have_img[2] = {false, false};

    // triangulation
    int NT = 0, NP = 0, capT = 0, capP = 0;
    float2* points = nullptr;
    int4* tris = nullptr;
    int4* colors = nullptr;
    int* vtx_off = nullptr;
    int* vtx_adj = nullptr;
    int* vref = nullptr;   // per-upload reference tables of k_update
    int* vvar = nullptr;
    int NE = 0, capE = 0;
    int2* edge_uv = nullptr;
    int* he_edge = nullptr;
    int2* vpos = nullptr;
    float2* epos = nullptr;       // endpoint positions per edge
    int64_t* wline = nullptr;      // whole line sums [capE][9][6] (k_lines)
    int lanes_per_line = 1;        // k_lines: lanes per line, from the mean number of rows of an edge at upload
    uint4* prefix[2] = {nullptr, nullptr};   // per-image row prefix tables
    int prefix_pitch = 0;
    uint4* px[2] = {nullptr, nullptr};     // the same in pixel records (rasters up to TP_PX_MAXW columns): persistent kernel
    uint4* pxt[2] = {nullptr, nullptr};    // ... and tiled 4 rows x 2 columns per 128 bytes (tp_raster.h): lane-items that keep no records
    int px_pitch = 0;
    // outputs
    int32_t* ten = nullptr;
    int32_t* cn = nullptr;
    int4* ca = nullptr;
    int2* gr = nullptr;
    int64_t* moments = nullptr;

    bool uploaded = false, accumulated = false, energized = false, have_colors = false;
    int acc_slot = 0, acc_flavour = 0;
    float dp_override = 0.0f;  // <= 0: reference law
    int last_flavour = 0;
    uint64_t generation = 1;
    std::vector<graph_entry> graphs;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int32_t* eval_host = nullptr; size_t eval_cap = 0;   // tp_evaluate_triangles: {vertices, variants | energies, counts}, pinned (words)
    bool on_device = false;         // counted among its device's contexts (join_device / leave_device)
    hipEvent_t ev_tail = nullptr;   // a marker behind a single frame (tp_iterate), polled by the frame's read-back
    uint64_t tail_mark = ~0ull;     // the value of `mutations` when ev_tail was recorded as the LAST thing on the stream (a single frame)
    hipEvent_t ev_turn = nullptr;   // behind this context's last persistent launch, when other contexts share the device (tp_persist_host.hip: device turns)
    uint8_t* pinned = nullptr;   // host-pinned staging for readbacks (one synchronisation per batch)
    size_t pinned_bytes = 0;
    uint8_t* render_pic = nullptr;   // tp_render scratch (kept: the viewer renders every frame)
    float2* render_pts = nullptr;
    size_t render_pts_cap = 0;
    uint8_t* up_pinned = nullptr;  // host-pinned staging for tp_upload (copies ride the stream, no wait at the end)
    size_t up_pinned_bytes = 0;
    // persistent grad-iter kernel (tp_persist.hip): the plan of the current triangulation is built by the first
    // tp_iterate long enough to use it (not by tp_upload: schedules upload after every topology change)
    int persist_mode = TP_PERSIST_AUTO;
    int inject_give_up = 0;         // TP_OPT_INJECT_GIVE_UP: the n-th persistent launch from now gives up (tests of the replay)
    bool has_loose = false;         // a vertex no triangle uses (k_persist_finish clamps it; launches do not finish themselves then)
    int num_cus = 0;
    int census = 0;                 // 0 not taken, 1 every workgroup of a full grid is resident, -1 not: two-kernel path only
    int lds_attr = 0;               // dynamic LDS the kernel is currently allowed
    std::vector<float> h_points;    // host copies of the last upload (what the plan is cut from)
    std::vector<int32_t> h_tris, h_edge_uv, h_he_edge;
    pk_plan plan;
    uint64_t plan_generation = 0;   // generation the plan (ok or refused) belongs to
    pk_wg* d_wg = nullptr; int32_t* d_pool = nullptr;   // the plan the next launch reads (= plan_dev[plan_slot])
    size_t cap_wg = 0, cap_pool = 0;                    // (census only)
    // Re-planning while a long descent runs: vertices drift, lines grow, and the patches of the upload-time plan go out of
    // balance.  After every chunk of grad-iters the positions ride the stream into a pinned snapshot; before launching a
    // chunk the host waits for the snapshot of two chunks ago (never more than two chunks are in flight), and if a vertex
    // has moved more than PK_REPLAN_PX pixels since the current plan was cut, cuts a new one from it -- while the GPU runs
    // the chunk in between -- and uploads it into the other of two plan buffers.  Results do not depend on the cut.
    struct plan_buf { pk_wg* wg = nullptr; int32_t* pool = nullptr; size_t cap_wg = 0, cap_pool = 0; uint8_t* stage = nullptr; size_t cap_stage = 0; };
    plan_buf plan_dev[2];
    int plan_slot = 0;
    // What a persistent launch hands to the next one on the same plan (tp_persist.hip, "carry"): per workgroup the cut of its lines into lane
    // chunks, the lane-items of its threads and how long ago the lines were last cut -- a launch that finds them does not cut and search again
    // (3 us of a short call's first grad-iter).  Tagged: a new plan, image or dp gets a new tag, and words that carry another are ignored.
    int32_t* carry = nullptr; size_t cap_carry = 0;
    int carry_stride = 0;
    uint32_t carry_tag = 0, carry_seq = 0;
    float carry_dp = -1.0f; int carry_slot = -1;   // what the launches under the current tag were called with
    int64_t warm_launches = 0;       // (tp_get_info 11: persistent launches enqueued with a carry of the same tag behind them)
    bool carry_written = false;      // a launch with the current tag has been enqueued: the next one finds its carry
    std::vector<float> plan_points;           // positions the current plan was cut from
    // How far every vertex moves per grad-iter (round 6): the kernel itself averages |step| per vertex and axis over every launch (pk_args::vspeed)
    // and the snapshot carries it to the host beside the positions.  The planner weighs a vertex's rows by it (tp_plan.h: pk_vertex_work) -- a row
    // whose crossing column changes every grad-iter costs five times one that stands -- and a plan is cut again when the patches have gone out of
    // balance under the speeds of the day, not only when vertices have drifted.
    float* vspeed = nullptr; size_t cap_vspeed = 0;     // device, [NP][2], t-pose units per grad-iter
    float* snap_speed[2] = {nullptr, nullptr};          // pinned, beside snap_host
    std::vector<float> last_speed_px; uint64_t speed_generation = 0;   // the speeds the last plan of this triangulation was weighted with (pixels per grad-iter)
    double plan_balance = 1.0;                          // heaviest patch / mean patch of the current plan under the weights it was cut with
    double plan_heaviest_vertex = 0.0;                  // heaviest VERTEX / mean patch, likewise (a patch owns whole vertices: the floor of plan_balance)
    uint64_t probed_generation = 0;                     // the triangulation tp_prepare has probed the speeds of
    int64_t replans_balance = 0;                        // (tp_get_info 14: plans cut again because the patches were out of balance)
    float* snap_host[2] = {nullptr, nullptr};  // pinned: positions after a chunk
    size_t snap_cap = 0;
    hipEvent_t snap_ev[2] = {nullptr, nullptr};
    bool snap_pending[2] = {false, false};
    int snap_next = 0;
    int iters_since_snap = 0;
    int64_t replans = 0;
    int64_t iters_since_cut = 0;   // grad-iters enqueued since the current plan was cut
    // After the LAST chunk of a call a new plan is cut on a worker thread of the context (2.6 ms at 3000 triangles: a call of a
    // few grad-iters must not wait for it); a later call installs it when it finds it finished.  ONE thread for the life of the
    // context, started at the first such cut, working on its own copies of everything it reads.
    struct replan_worker {
        std::thread th;
        std::mutex m;
        std::condition_variable cv;
        bool stop = false, go = false, busy = false, done = false, superseded = false;
        pk_plan plan;
        std::vector<float> points, speed;   // (speed: pixels per grad-iter and vertex, or empty)
        std::vector<int32_t> tris, edge_uv, he_edge;
        int NP = 0, NT = 0, NE = 0, W = 0, H = 0, parts = 0;
        float ratio = 0.0f, dp = 0.0f;
        uint64_t generation = 0;
        bool base_every = false;
        int rows_cap = PK_ROWS_BIG;   // (tp_persist_host.hip: plan_rows_cap)
        double balance = 1.0;
    };
    std::unique_ptr<replan_worker> worker;
    bool plan_base_every = false;   // the current plan walks every triangle's base lines in every grad-iter (tp_iterate_until)
    int32_t* ering = nullptr; float2* pring = nullptr;   // tp_iterate_until: per-frame base energies / positions of a chunk
    size_t cap_ering = 0, cap_pring = 0;
    int32_t* ering_host = nullptr; size_t cap_ering_host = 0;   // pinned
    float2* pring_host = nullptr; size_t cap_pring_host = 0;    // pinned: tp_iterate_frames hands every frame's positions to the caller
    unsigned long long* posbox = nullptr;
    size_t cap_posbox = 0;   // (in vertices)
    // band split (tp_band_attach): this context runs band `band` of `n_bands` -- the patches [band, band + 1) * band_patches of a
    // plan of n_bands * band_patches -- and the mailbox is the caller's (one per band, mapped into every band's process)
    int band = 0, n_bands = 1, band_patches = 0;
    unsigned long long* band_box[PK_MAX_PEERS + 1] = {nullptr, nullptr, nullptr, nullptr};
    size_t band_cap = 0, band_cap_tris = 0;   // (vertices, triangles the mailboxes were sized for)
    uint64_t band_seq = 0;    // persistent launches since tp_band_attach: which of the two final slot arrays a launch ends in (the same on every band)
    uint64_t ring_seq = 0;    // ring chunks of tp_iterate_until since tp_band_attach: which half of the bands' rings a chunk writes (see ring_half)
    int ring_half = 0;
    bool box_finegrained = false;   // this band's mailbox was allocated by tp_band_mailbox_alloc as fine-grained memory
    float2* points_out = nullptr; size_t cap_points_out = 0;
    unsigned* d_status = nullptr;   // [0] a lane of a persistent launch gave up waiting, [1] census counter
    unsigned* h_status = nullptr;   // pinned mirror of [0] and [2], written by k_persist_finish: read after a wait, no copy
    // frame mirror: a single frame (tp_iterate(ctx, p, 1) on the two-kernel path) leaves the first frame_n entries of `tenergy` and
    // `colnum` and all points in pinned memory as well; tp_retrieve_many takes them from there while nothing has touched the
    // context since (`mutations` counts every call that may change what a retrieve returns)
    uint8_t* frame_mirror = nullptr; size_t frame_mirror_bytes = 0;
    int frame_n = 0; size_t frame_np = 0;
    uint64_t mutations = 0, ten_stamp = ~0ull, pts_stamp = ~0ull;   // (the mirror's energies / points are current while stamp == mutations)
    bool epos_stale = false;        // persistent launches moved the vertices and left the edges' endpoint copies (`epos`) behind: filed before k_lines runs
    bool tail_is_finish = false;    // the last thing enqueued on the stream is the small kernel behind a persistent launch: its pinned words say when the stream is done
    uint32_t epoch = 1;             // number of the next grad-iter of a persistent launch (mailbox tags)
    bool persist_unchecked = false; // persistent launches were enqueued since the status word was last read
    struct journal_entry { tp_params p; int iters; float2* before; };   // before: `points` when a launch that finishes itself was enqueued (else null)
    std::vector<journal_entry> journal;   // ... which ones (tp_iterate): replayed on the two-kernel path if a launch gave up
    unsigned done_base = 0;               // the device's count of completed persistent launches when the journal was last empty
    int64_t persist_failures = 0;
    int persist_streak = 0;                  // give-ups in a row (64 completed launches end a row): what the wait before the next try grows with
    int64_t completed_since_give_up = 0;
    int census_retries = 0;         // a census that timed out (the device was busy with somebody else's grid) is taken again, twice at most
    std::chrono::steady_clock::time_point persist_retry_at{};   // after a launch gave up: when persistent launches are tried again
    int64_t persist_launches = 0, persist_iters = 0;
    std::vector<uint64_t> hkeys;   // open-addressing table of tp_upload: undirected edge key -> id
    std::vector<int> hvals;
    std::vector<uint32_t> hstamp;
    uint32_t hgen = 0;
};

namespace tpctx {

int fail(tp_context* c, int code, const char* fmt, ...);

#define HIP_TRY(ctx, expr)                                                                         \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(ctx, TP_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),    \
                        __FILE__, __LINE__);                                                       \
    } while (0)

template <class T>
hipError_t dev_alloc(T** p, size_t n) {
    return hipMalloc(reinterpret_cast<void**>(p), (n ? n : 1) * sizeof(T));
}

template <class T>
int grow(tp_context* c, T** p, size_t* cap, size_t need) {
    if (need <= *cap && *p) return TP_OK;
    hipFree(*p); *p = nullptr; *cap = 0;
    const size_t n = need + need / 2 + 64;
    HIP_TRY(c, dev_alloc(p, n));
    *cap = n;
    return TP_OK;
}

#define PK_LDS_LIMIT (160 * 1024 / PK_WG_PER_CU - 512)  /* per workgroup (the kernel has a few static bytes of its own) */
#define PK_MIN_ITERS 4        /* shorter tp_iterate calls are not worth a plan (frame-by-frame schedules) */
#define PK_MAX_EPOCH 0x7f000000u   /* mailbox tags carry 31 bits of the grad-iter's number */
#ifndef PK_CHUNK
#define PK_CHUNK 512          /* grad-iters per launch of a long call: the granule of re-planning */
#endif
#ifndef PK_REPLAN_PX
#define PK_REPLAN_PX 2.0f     /* a vertex this far from where the plan saw it: cut a new plan */
#endif
#ifndef PK_REPLAN_BALANCE
#define PK_REPLAN_BALANCE 1.15   /* the heaviest patch this far above the mean under today's speeds (and 8 % worse than when the plan was cut): cut a new plan */
#endif
#ifndef PK_PROBE_ITERS
#define PK_PROBE_ITERS 8      /* grad-iters of tp_prepare's probe of the vertices' speeds */
#endif
#define PK_RING_FRAMES 256    /* frames of a chunk of tp_iterate_until a band's mailbox has rings for (two halves, used in turn) */

// tp_context.hip
void drop_graphs(tp_context* c);
tp_launch make_launch(const tp_context* c, int slot, float dp);
float resolve_dp(const tp_context* c, int flavour, float dp);
int check_slot(tp_context* c, int slot);
hipError_t wait_stream(hipStream_t s);
hipError_t wait_event(hipEvent_t ev);
int validate_params(tp_context* c, const tp_params* p, int n_iters);
int enqueue_iter(tp_context* c, const tp_params& p, float dp, bool mirror = false);
int enqueue_two_kernel(tp_context* c, const tp_params* p, float dp, int n);
// tp_persist_host.hip
int check_persist_status(tp_context* c);
int settle_persistent(tp_context* c);
int settle_epos(tp_context* c);   // before anything that reads `epos` (k_lines) is enqueued
hipError_t wait_context(tp_context* c);   // wait for the context's stream
int install_plan(tp_context* c, pk_plan& np, const float* points, int slot);
void drop_carry(tp_context* c);   // what the last launch left for the next is not to be used (a new plan, image or dp)
int plan_patches(const tp_context* c);
int build_plan(tp_context* c, const float* points, float dp, int slot, bool* ok, const float* speed_px = nullptr);
int plan_rows_cap(const tp_context* c);   // rows per lane the next cut of this triangulation starts from
int ensure_plan(tp_context* c, float dp, bool* use, bool base_every = false);
int enqueue_persistent(tp_context* c, const tp_params& p, float dp, int n, bool rings = false, bool probe = false, bool rings_emit = false);
int probe_speeds(tp_context* c, const tp_params& p, float dp);   // tp_prepare: a few grad-iters nobody keeps, for the planner
// tp_replan.hip
int take_replan(tp_context* c);
void join_device(tp_context* c);    // a context came to / left a device: persistent launches of the contexts of ONE device take turns
void leave_device(tp_context* c);
void stop_replan_worker(tp_context* c);
int maybe_replan(tp_context* c, float dp, bool more_chunks);
// tp_bands.hip
size_t band_slots_bytes(size_t cap);
size_t band_ering_bytes(size_t cap_tris);
bool banded_rings(const tp_context* c);
int32_t* band_ering(const tp_context* c, int b);
float2* band_pring(const tp_context* c, int b);
// tp_readback.hip
int frame_mirror_into(tp_context* c, tp_launch& L);

}  // namespace tpctx
