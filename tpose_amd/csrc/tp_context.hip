// tp_context.hip -- the C ABI of include/tpose_hip.h: context, device memory, launch sequencing,
// hipGraph-fused iteration.  Host side of the boundary that replaces tpose::init/quit/upload and
// the computecolors/doenergy/doshift lambdas of the reference (source/triangulation.hpp:576-643,
// software/triangulate/main.cpp:121-155, software/warp/main.cpp:140-178).
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
static std::shared_mutex g_api_mutex;
static thread_local int g_api_depth = 0;
struct api_guard {  // outermost entry point of this thread takes the shared lock (entry points call each other)
    api_guard() { if (g_api_depth++ == 0) g_api_mutex.lock_shared(); }
    ~api_guard() { if (--g_api_depth == 0) g_api_mutex.unlock_shared(); }
};
struct capture_guard {  // inside an entry point: trade the shared lock for the exclusive one
    capture_guard() { g_api_mutex.unlock_shared(); g_api_mutex.lock(); }
    ~capture_guard() { g_api_mutex.unlock(); g_api_mutex.lock_shared(); }
};

namespace {

thread_local std::string g_create_error;

struct graph_entry {
    hipGraphExec_t exec = nullptr;
    tp_params params{};
    int iters = 0;
    uint64_t generation = 0;
};

}  // namespace

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
    std::vector<float> plan_points;           // positions the current plan was cut from
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
        std::vector<float> points;
        std::vector<int32_t> tris, edge_uv, he_edge;
        int NP = 0, NT = 0, NE = 0, W = 0, H = 0, parts = 0;
        float ratio = 0.0f, dp = 0.0f;
        uint64_t generation = 0;
        bool base_every = false;
    };
    std::unique_ptr<replan_worker> worker;
    bool plan_base_every = false;   // the current plan walks every triangle's base lines in every grad-iter (tp_iterate_until)
    int32_t* ering = nullptr; float2* pring = nullptr;   // tp_iterate_until: per-frame base energies / positions of a chunk
    size_t cap_ering = 0, cap_pring = 0;
    int32_t* ering_host = nullptr; size_t cap_ering_host = 0;   // pinned
    unsigned long long* posbox = nullptr;
    size_t cap_posbox = 0;   // (in vertices)
    // band split (tp_band_attach): this context runs band `band` of `n_bands` -- the patches [band, band + 1) * band_patches of a
    // plan of n_bands * band_patches -- and the mailbox is the caller's (one per band, mapped into every band's process)
    int band = 0, n_bands = 1, band_patches = 0;
    unsigned long long* band_box[PK_MAX_PEERS + 1] = {nullptr, nullptr, nullptr, nullptr};
    size_t band_cap = 0, band_cap_tris = 0;   // (vertices, triangles the mailboxes were sized for)
    float2* points_out = nullptr; size_t cap_points_out = 0;
    unsigned* d_status = nullptr;   // [0] a lane of a persistent launch gave up waiting, [1] census counter
    unsigned* h_status = nullptr;   // pinned mirror of [0] and [2], written by k_persist_finish: read after a wait, no copy
    // frame mirror: a single frame (tp_iterate(ctx, p, 1) on the two-kernel path) leaves the first frame_n entries of `tenergy` and
    // `colnum` and all points in pinned memory as well; tp_retrieve_many takes them from there while nothing has touched the
    // context since (`mutations` counts every call that may change what a retrieve returns)
    uint8_t* frame_mirror = nullptr; size_t frame_mirror_bytes = 0;
    int frame_n = 0; size_t frame_np = 0;
    uint64_t mutations = 0, ten_stamp = ~0ull, pts_stamp = ~0ull;   // (the mirror's energies / points are current while stamp == mutations)
    uint32_t epoch = 1;             // number of the next grad-iter of a persistent launch (mailbox tags)
    bool persist_unchecked = false; // persistent launches were enqueued since the status word was last read
    struct journal_entry { tp_params p; int iters; };
    std::vector<journal_entry> journal;   // ... which ones (tp_iterate): replayed on the two-kernel path if a launch gave up
    unsigned done_base = 0;               // the device's count of completed persistent launches when the journal was last empty
    int64_t persist_failures = 0;
    int64_t persist_launches = 0, persist_iters = 0;
    std::vector<uint64_t> hkeys;   // open-addressing table of tp_upload: undirected edge key -> id
    std::vector<int> hvals;
    std::vector<uint32_t> hstamp;
    uint32_t hgen = 0;
};

namespace {

int fail(tp_context* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->error = buf; else g_create_error = buf;
    return code;
}

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

void drop_graphs(tp_context* c) {
    for (auto& g : c->graphs)
        if (g.exec) hipGraphExecDestroy(g.exec);
    c->graphs.clear();
}

void free_triangulation(tp_context* c) {
    hipFree(c->vref); hipFree(c->vvar); c->vref = nullptr; c->vvar = nullptr;
    hipFree(c->points); hipFree(c->tris); hipFree(c->colors); hipFree(c->vtx_off); hipFree(c->vtx_adj);
    hipFree(c->edge_uv); hipFree(c->he_edge); hipFree(c->vpos); hipFree(c->epos); hipFree(c->wline);
    hipFree(c->ten); hipFree(c->cn); hipFree(c->ca); hipFree(c->gr); hipFree(c->moments);
    c->points = nullptr; c->tris = nullptr; c->colors = nullptr; c->vtx_off = nullptr; c->vtx_adj = nullptr;
    c->edge_uv = nullptr; c->he_edge = nullptr; c->vpos = nullptr; c->epos = nullptr; c->wline = nullptr;
    c->ten = nullptr; c->cn = nullptr; c->ca = nullptr; c->gr = nullptr; c->moments = nullptr;
    c->capT = c->capP = c->capE = 0;
}

tp_launch make_launch(const tp_context* c, int slot, float dp) {
    tp_launch L{};
    L.img = c->img[slot];
    L.pitch = c->W * 4;
    L.prefix = c->prefix[slot]; L.prefix_pitch = c->prefix_pitch;
    L.vw.dp = dp; L.vw.ratio = c->ratio;
    L.vw.halfW = 0.5f * (float)c->W; L.vw.halfH = 0.5f * (float)c->H;
    L.vw.W = c->W; L.vw.H = c->H;
    L.points = c->points;
    L.tris = c->tris; L.colors = c->colors;
    L.NT = c->NT; L.NP = c->NP;
    L.vtx_off = c->vtx_off; L.vtx_adj = c->vtx_adj; L.vref = c->vref; L.vvar = c->vvar;
    L.edge_uv = c->edge_uv; L.he_edge = c->he_edge; L.vpos = c->vpos; L.epos = c->epos; L.NE = c->NE;
    L.wline = c->wline; L.lanes_per_line = c->lanes_per_line;
    L.ten = c->ten; L.cn = c->cn; L.ca = c->ca; L.gr = c->gr; L.moments = c->moments;
#ifdef TPOSE_DEBUG  // debug flavour of the library (tools/kernel_timeline.py): per-block phase timestamps
    static unsigned long long* dbgbuf = nullptr;
    if (!dbgbuf) { hipMalloc((void**)&dbgbuf, 3 * 4096 * 8 * sizeof(unsigned long long)); hipMemset(dbgbuf, 0, 3 * 4096 * 8 * 8); }
    L.dbg = dbgbuf;
#endif
    return L;
}

float resolve_dp(const tp_context* c, int flavour, float dp) {
    return dp > 0.0f ? dp : tp_reference_dp(flavour, c->NT);
}

int check_slot(tp_context* c, int slot) {
    if (slot != TP_IMAGE_A && slot != TP_IMAGE_B) return fail(c, TP_ERR_INVALID, "bad image slot %d", slot);
    if (!c->have_img[slot]) return fail(c, TP_ERR_STATE, "image slot %d was never set", slot);
    return TP_OK;
}

// Waiting for the stream: calls of a few grad-iters finish in tens of microseconds, less than it takes a blocked host
// thread to be woken (~20-30 us).  Poll for up to a quarter of a millisecond first, then block.
hipError_t wait_stream(hipStream_t s) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipStreamQuery(s);
        if (e != hipErrorNotReady) return e;
        if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(250)) break;
    }
    return hipStreamSynchronize(s);
}
hipError_t wait_event(hipEvent_t ev) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipEventQuery(ev);
        if (e != hipErrorNotReady) return e;
        if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(250)) break;
    }
    return hipEventSynchronize(ev);
}

// ---- persistent grad-iter kernel: status, census, plan ------------------------------------------------------------
#ifdef TPOSE_DEBUG  // debug flavour of the library (tools/persist_timeline.py): per-workgroup phase timestamps
static unsigned long long* g_persist_dbg = nullptr;
static const size_t PERSIST_DBG_WORDS = (size_t)512 * PK_DBG_ITERS * 16;
unsigned long long* persist_dbg_buffer(int parts, hipStream_t s) {
    if (!g_persist_dbg) { hipMalloc((void**)&g_persist_dbg, PERSIST_DBG_WORDS * 8); }
    hipMemsetAsync(g_persist_dbg, 0, PERSIST_DBG_WORDS * 8, s);
    return parts <= 512 ? g_persist_dbg : nullptr;
}
#endif
#define PK_LDS_LIMIT (160 * 1024 / PK_WG_PER_CU - 512)  /* per workgroup (the kernel has a few static bytes of its own) */
#define PK_MIN_ITERS 4        /* shorter tp_iterate calls are not worth a plan (frame-by-frame schedules) */
#define PK_MAX_EPOCH 0x7f000000u   /* mailbox tags carry 31 bits of the grad-iter's number */

int enqueue_two_kernel(tp_context* c, const tp_params* p, float dp, int n);

// After the stream was synchronised: did a lane of a persistent launch give up waiting?  That happens when the launch's
// workgroups were not all resident together -- another process or another context had a persistent launch of its own on
// the same GPU at that moment (the census only shows that a full grid fits an otherwise idle device).  A launch that
// gives up changes nothing: `points` is only written by the small kernel behind it, which does nothing once the status
// word is raised, and so do all later persistent launches.  So the grad-iters of the launches that did not complete are
// run again here, on the two-kernel path, and the context stops using persistent launches.
int check_persist_status(tp_context* c) {
    if (!c->persist_unchecked || !c->d_status) return TP_OK;
    c->persist_unchecked = false;
    // (every caller has waited for the stream: the mirror is what the last k_persist_finish left)
    const unsigned st[3] = {c->h_status[0], 0u, c->h_status[2]};
    const size_t completed = (size_t)(st[2] - c->done_base);
    c->done_base = st[2];
    if (st[0] == 0u) { c->journal.clear(); return TP_OK; }
    HIP_TRY(c, hipMemset(c->d_status, 0, sizeof(unsigned)));
    c->h_status[0] = 0u;
    c->census = -6;  // two kernels per grad-iter from now on in this context
    c->persist_failures++;
    c->mutations++;  // (what a retrieve returns is about to change)
    std::vector<tp_context::journal_entry> todo(c->journal.begin() + (completed < c->journal.size() ? completed : c->journal.size()), c->journal.end());
    c->journal.clear();
    for (auto& e : todo) {
        if (e.iters <= 0) continue;
        if (int rc = enqueue_two_kernel(c, &e.p, resolve_dp(c, e.p.flavour, e.p.dp), e.iters)) return rc;
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return TP_OK;
}

// before work that is NOT a persistent launch goes onto the stream: were the persistent launches ahead of it completed?  (A launch
// that gave up is run again at the next check -- and that must be before anything that continues from its result.)
int settle_persistent(tp_context* c) {
    if (!c->persist_unchecked) return TP_OK;
    HIP_TRY(c, wait_stream(c->stream));
    return check_persist_status(c);
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

// once per context: launch a full grid of the persistent kernel in census mode -- every workgroup arrives at a counter and
// waits for all the others.  If that times out, workgroups of such a grid are not resident together on this device
// (CU masking, another process) and hand-overs inside a launch would never complete: the context keeps to two kernels.
int take_census(tp_context* c) {
    if (c->census != 0) return TP_OK;
    c->census = -1;
    if (c->num_cus < 1) return TP_OK;
    if (!c->d_status) { HIP_TRY(c, dev_alloc(&c->d_status, 4)); }
    if (!c->h_status) { HIP_TRY(c, hipHostMalloc((void**)&c->h_status, 4 * sizeof(unsigned), hipHostMallocDefault)); memset(c->h_status, 0, 4 * sizeof(unsigned)); }
    HIP_TRY(c, hipMemsetAsync(c->d_status, 0, 4 * sizeof(unsigned), c->stream));
    if (tp_persist_set_lds(PK_LDS_LIMIT) != 0) { (void)hipGetLastError(); c->census = -2; return TP_OK; }
    c->lds_attr = PK_LDS_LIMIT;
    const int full = c->num_cus * PK_WG_PER_CU;   // the grid that must be resident at once
    std::vector<pk_wg> hw((size_t)full, pk_wg());
    if (int rc = grow(c, &c->d_wg, &c->cap_wg, hw.size())) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->d_wg, hw.data(), sizeof(pk_wg) * hw.size(), hipMemcpyHostToDevice, c->stream));
    pk_args A{};
    A.wg = c->d_wg; A.parts = full; A.n_iters = -1; A.status = c->d_status;
    tp_launch_persist(A, full, PK_ROWS_PER_LANE, PK_LDS_LIMIT, c->stream);
    if (hipGetLastError() != hipSuccess) { c->census = -3; return TP_OK; }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    unsigned st[2] = {1u, 0u};
    HIP_TRY(c, hipMemcpy(st, c->d_status, sizeof st, hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemset(c->d_status, 0, 4 * sizeof(unsigned)));
    c->done_base = 0;
    if (st[0] == 0u && st[1] == (unsigned)full) c->census = 1;
    else c->census = -4 - (int)(st[0] != 0u);
    return TP_OK;
}

#ifndef PK_CHUNK
#define PK_CHUNK 512          /* grad-iters per launch of a long call: the granule of re-planning */
#endif
#ifndef PK_REPLAN_PX
#define PK_REPLAN_PX 2.0f     /* a vertex this far from where the plan saw it: cut a new plan */
#endif


// send a plan that was cut from `points` to plan buffer `slot` (through that buffer's pinned staging area: the copy rides
// the stream and the host does not wait for it) and make it the context's plan
int install_plan(tp_context* c, pk_plan& np, const float* points, int slot) {
    tp_context::plan_buf& B = c->plan_dev[slot];
    if (int rc = grow(c, &B.wg, &B.cap_wg, np.wg.size())) return rc;
    if (int rc = grow(c, &B.pool, &B.cap_pool, np.pool.size())) return rc;
    const size_t b_wg = sizeof(pk_wg) * np.wg.size(), b_pool = sizeof(int32_t) * np.pool.size();
    static_assert(sizeof(pk_wg) % 4 == 0, "plans travel as 32-bit words");
    if (b_wg + b_pool > B.cap_stage) {
        // (the staging area may still feed a copy enqueued for an earlier plan in this buffer: wait before dropping it)
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (B.stage) hipHostFree(B.stage);
        B.stage = nullptr; B.cap_stage = 0;
        const size_t n = (b_wg + b_pool) * 3 / 2 + 4096;
        HIP_TRY(c, hipHostMalloc((void**)&B.stage, n, hipHostMallocDefault));
        B.cap_stage = n;
    }
    memcpy(B.stage, np.wg.data(), b_wg);
    memcpy(B.stage + b_wg, np.pool.data(), b_pool);
    // (ONE kernel reads the staging area across the link.  Two hipMemcpyAsync did this before; issued on an IDLE stream --
    // a plan cut on the side, installed at the start of a call -- they returned after 8 ms once in ~30 times.)
    tp_copy_list G{};
    G.src[0] = (const uint32_t*)B.stage; G.dst[0] = (uint32_t*)B.wg; G.words[0] = (uint32_t)(b_wg / 4);
    G.src[1] = (const uint32_t*)(B.stage + b_wg); G.dst[1] = (uint32_t*)B.pool; G.words[1] = (uint32_t)(b_pool / 4);
    G.n = 2;
    tp_launch_copy_list(G, c->stream);
    HIP_TRY(c, hipGetLastError());
    c->plan = std::move(np);
    c->plan_slot = slot;
    c->plan_points.assign(points, points + 2 * (size_t)c->NP);
    c->iters_since_cut = 0;
    return TP_OK;
}
int plan_patches(const tp_context* c) { return c->n_bands > 1 ? c->n_bands * c->band_patches : c->num_cus * PK_WG_PER_CU; }

// cut a plan from `points` and install it in plan buffer `slot`.  c->plan is replaced only when the new plan is usable.
int build_plan(tp_context* c, const float* points, float dp, int slot, bool* ok) {
    pk_plan np;
    pk_build_plan(c->NP, c->NT, c->h_tris.data(), points, c->NE, c->h_edge_uv.data(), c->h_he_edge.data(),
                  c->W, c->H, c->ratio, dp * 0.5f * (float)c->H, plan_patches(c), PK_LDS_LIMIT, np,
                  c->plan_base_every);
    // (a band split runs equal shares of the patches: a plan with fewer patches than asked for -- a tiny mesh -- is not split)
    if (np.ok && c->n_bands > 1 && np.parts != c->n_bands * c->band_patches) { np.ok = false; np.why = "fewer patches than the bands need"; }
    *ok = np.ok;
    if (!np.ok) { if (!c->plan.ok) c->plan = np; return TP_OK; }
    return install_plan(c, np, points, slot);
}

// the same cut on the context's worker thread, from a snapshot of the positions (maybe_replan); nothing of the context is
// touched until take_replan() finds the cut finished
void replan_worker_main(tp_context::replan_worker* w) {
    std::unique_lock<std::mutex> lk(w->m);
    for (;;) {
        w->cv.wait(lk, [w] { return w->go || w->stop; });
        if (w->stop) return;
        w->go = false;
        lk.unlock();
        pk_build_plan(w->NP, w->NT, w->tris.data(), w->points.data(), w->NE, w->edge_uv.data(), w->he_edge.data(), w->W, w->H, w->ratio,
                      w->dp * 0.5f * (float)w->H, w->parts, PK_LDS_LIMIT, w->plan, w->base_every);
        lk.lock();
        w->busy = false; w->done = true;
    }
}
void start_replan(tp_context* c, const float* points, float dp) {
    if (!c->worker) {
        c->worker.reset(new tp_context::replan_worker());
        c->worker->th = std::thread(replan_worker_main, c->worker.get());
    }
    tp_context::replan_worker* w = c->worker.get();
    std::lock_guard<std::mutex> lk(w->m);
    if (w->busy || w->done) return;   // (one cut at a time: the one under way is from positions nearly as new)
    w->points.assign(points, points + 2 * (size_t)c->NP);
    w->tris = c->h_tris; w->edge_uv = c->h_edge_uv; w->he_edge = c->h_he_edge;
    w->NP = c->NP; w->NT = c->NT; w->NE = c->NE; w->W = c->W; w->H = c->H; w->parts = plan_patches(c);
    w->ratio = c->ratio; w->dp = dp; w->generation = c->generation; w->base_every = c->plan_base_every;
    w->superseded = false; w->busy = true; w->go = true;
    w->cv.notify_one();
}
// a finished cut becomes the context's plan (for the launches enqueued from now on); one of another triangulation, of the other
// kind of plan, or overtaken by a cut on the calling thread is dropped
int take_replan(tp_context* c) {
    tp_context::replan_worker* w = c->worker.get();
    if (!w) return TP_OK;
    std::lock_guard<std::mutex> lk(w->m);
    if (!w->done) return TP_OK;
    w->done = false;
    if (w->superseded || w->generation != c->generation || c->plan_generation != c->generation || w->base_every != c->plan_base_every || !w->plan.ok || c->n_bands > 1) return TP_OK;
    if (int rc = install_plan(c, w->plan, w->points.data(), c->plan_slot ^ 1)) return rc;
    c->replans++;
    return TP_OK;
}
void stop_replan_worker(tp_context* c) {
    if (!c->worker) return;
    { std::lock_guard<std::mutex> lk(c->worker->m); c->worker->stop = true; }
    c->worker->cv.notify_one();
    if (c->worker->th.joinable()) c->worker->th.join();
    c->worker.reset();
}

// the plan of the current triangulation (built on first use after an upload); *use = whether tp_iterate may take the
// persistent path
int ensure_plan(tp_context* c, float dp, bool* use, bool base_every = false) {
    *use = false;
    // (bands keep ONE plan for tp_iterate and tp_iterate_until -- the one that walks every triangle's base lines in every grad-iter:
    // cutting a plan again allocates, and an allocation may wait for a device on which another band is already waiting for this one)
    if (c->n_bands > 1) base_every = true;
    if (c->persist_mode == TP_PERSIST_OFF || !c->px_pitch) return TP_OK;  // (rasters beyond 4096 columns or rows have no pixel-record table)
    if (int rc = take_census(c)) return rc;
    if (c->census != 1) return TP_OK;
    if (c->plan_generation == c->generation && base_every && !c->plan_base_every) {
        // the plan of this triangulation does not walk the base lines in every grad-iter yet: cut it again (from the
        // upload-time positions; a later re-plan follows the mesh) -- nothing in flight reads the plan buffers by then
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->plan_generation = 0;
    }
    if (c->plan_generation != c->generation) {
        c->plan_generation = c->generation;
        c->plan = pk_plan();
        c->plan_base_every = base_every;
        c->snap_pending[0] = c->snap_pending[1] = false;   // (uploads synchronise the stream: nothing is in flight)
        bool ok = false;
        if (int rc = build_plan(c, c->h_points.data(), dp, 0, &ok)) return rc;
        if (ok) {
            const size_t np = (size_t)c->NP;
            if (c->n_bands > 1 && np > c->band_cap) { c->plan.ok = false; c->plan.why = "more vertices than the bands' mailboxes hold"; *use = false; return TP_OK; }
            if (c->n_bands == 1 && (np > c->cap_posbox || !c->posbox)) {
                hipFree(c->posbox); c->posbox = nullptr; c->cap_posbox = 0;
                const size_t n = np + np / 2 + 64;
                HIP_TRY(c, dev_alloc(&c->posbox, n * 8)); c->cap_posbox = n;
                // a cleared mailbox matches no tag; afterwards tags never repeat (the epoch counts on across uploads), so
                // the granules of an earlier triangulation are never taken for this one's
                HIP_TRY(c, hipMemsetAsync(c->posbox, 0, c->cap_posbox * 8 * sizeof(unsigned long long), c->stream));
            }
            if (int rc = grow(c, &c->points_out, &c->cap_points_out, np)) return rc;
            if (np > c->snap_cap) {
                for (int k = 0; k < 2; k++) { if (c->snap_host[k]) hipHostFree(c->snap_host[k]); c->snap_host[k] = nullptr; }
                c->snap_cap = 0;
                const size_t n = np + np / 2 + 64;
                for (int k = 0; k < 2; k++) HIP_TRY(c, hipHostMalloc((void**)&c->snap_host[k], n * 2 * sizeof(float), hipHostMallocDefault));
                c->snap_cap = n;
            }
            for (int k = 0; k < 2; k++) if (!c->snap_ev[k]) HIP_TRY(c, hipEventCreateWithFlags(&c->snap_ev[k], hipEventDisableTiming));
        }
    }
    *use = c->plan.ok;
    return TP_OK;
}

// after a chunk has been enqueued: the snapshot taken after the chunk before it, if there is one -- a new plan for the
// chunks to come when the mesh has drifted
// more_chunks: the call has more chunks to enqueue behind the one just enqueued -- the cut is made right here, on the calling
// thread (the GPU runs that chunk meanwhile, and the next one starts on the new plan); otherwise on the context's worker
// thread, and a later call picks the plan up (a call of a few grad-iters never waits 2.6 ms for a cut)
int maybe_replan(tp_context* c, float dp, bool more_chunks) {
    const int k = c->snap_next;   // the older of the two snapshot slots: the one the next chunk's snapshot will overwrite
    if (!c->snap_pending[k]) return TP_OK;
    if (c->n_bands > 1) { c->snap_pending[k] = false; return TP_OK; }   // (bands keep the plan they all cut from the upload)
    HIP_TRY(c, hipEventSynchronize(c->snap_ev[k]));
    c->snap_pending[k] = false;
    const float* q = c->snap_host[k];
    const float* o = c->plan_points.data();
    const float sx = 0.5f * (float)c->W / c->ratio, sy = 0.5f * (float)c->H;
    float worst = 0.0f;
    for (size_t i = 0, n = 2 * (size_t)c->NP; i < n; i += 2) {
        const float dx = (q[i] - o[i]) * sx, dy = (q[i + 1] - o[i + 1]) * sy;
        const float d = (dx < 0 ? -dx : dx) > (dy < 0 ? -dy : dy) ? (dx < 0 ? -dx : dx) : (dy < 0 ? -dy : dy);
        if (d > worst) worst = d;   // (NaN never compares greater: a vertex gone to NaN does not trigger)
    }
    if (worst <= PK_REPLAN_PX) return TP_OK;
    if (!more_chunks) { start_replan(c, q, dp); return TP_OK; }
    if (c->worker) { std::lock_guard<std::mutex> lk(c->worker->m); c->worker->superseded = true; }
    bool ok = false;
    if (int rc = build_plan(c, q, dp, c->plan_slot ^ 1, &ok)) return rc;
    if (ok) c->replans++;
    return TP_OK;
}

// a band's mailbox: [4][cap] position slots of 16 bytes, then the rings of tp_iterate_until -- PK_RING_FRAMES frames of
// cap_tris base energies (int32) and of cap positions (float2)
#define PK_RING_FRAMES 256
static size_t band_slots_bytes(size_t cap) { return (cap * 64 + 255) & ~(size_t)255; }
static size_t band_ering_bytes(size_t cap_tris) { return ((size_t)PK_RING_FRAMES * cap_tris * 4 + 255) & ~(size_t)255; }
// band split: the rings of tp_iterate_until live in the bands' mailboxes (behind the position slots)
bool banded_rings(const tp_context* c) { return c->n_bands > 1 && c->band_cap_tris > 0; }
int32_t* band_ering(const tp_context* c, int b) { return (int32_t*)((char*)c->band_box[b] + band_slots_bytes(c->band_cap)); }
float2* band_pring(const tp_context* c, int b) { return (float2*)((char*)c->band_box[b] + band_slots_bytes(c->band_cap) + band_ering_bytes(c->band_cap_tris)); }

// n grad-iters of the persistent kernel -- the last one writes `tenergy`, `colnum`, `colacc`, `gradient` --, then
// `points_out` -> `points` / `epos`
int enqueue_persistent(tp_context* c, const tp_params& p, float dp, int n, bool rings = false) {
    while (n > 0) {
        if (int rc = take_replan(c)) return rc;   // (a plan cut on the side since an earlier call, if it is ready)
        // long calls go chunk by chunk (a chunk and a half rather than a short tail)
        const int k = n <= PK_CHUNK + PK_CHUNK / 2 ? n : PK_CHUNK;   // (rings: the caller's chunks are shorter than this)
        if (c->epoch + (uint32_t)k > PK_MAX_EPOCH) {
            if (c->n_bands > 1) return fail(c, TP_ERR_STATE, "band split: the mailbox tags are used up (2^31 grad-iters)");
            HIP_TRY(c, hipMemsetAsync(c->posbox, 0, c->cap_posbox * 8 * sizeof(unsigned long long), c->stream));
            c->epoch = 1;
        }
        pk_args A{};
        A.wg = c->plan_dev[c->plan_slot].wg; A.pool = c->plan_dev[c->plan_slot].pool; A.parts = c->plan.parts;
        A.vw.dp = dp; A.vw.ratio = c->ratio; A.vw.halfW = 0.5f * (float)c->W; A.vw.halfH = 0.5f * (float)c->H; A.vw.W = c->W; A.vw.H = c->H;
        A.px = c->px[p.image_slot]; A.px_pitch = c->px_pitch;
        A.points = c->points; A.points_out = c->points_out; A.ca = c->ca;
        A.NT = c->NT; A.NP = c->NP; A.NE = c->NE;
        A.flavour = p.flavour; A.rate = p.rate;
        const bool banded = c->n_bands > 1;
        const int grid = banded ? c->band_patches : c->plan.parts;
        A.posbox = banded ? c->band_box[c->band] : c->posbox;
        A.box_stride = (unsigned)(banded ? c->band_cap : c->cap_posbox);
        if (banded) {
            A.part0 = c->band * c->band_patches;
            for (int b = 0; b < c->n_bands; b++) if (b != c->band) A.peer_box[A.n_peers++] = c->band_box[b];
            A.final_tag = 0x80000000u | ((c->epoch + (uint32_t)k) & 0x7fffffffu);
            A.final_slot = (unsigned)(c->persist_launches & 1);
        }
        A.epoch = c->epoch; A.n_iters = k; A.status = c->d_status;
        A.emit = n == k && !rings; A.ten = c->ten; A.cn = c->cn; A.ca_out = c->ca; A.gr = c->gr;
        if (rings && !banded_rings(c)) { A.ering = c->ering; A.pring = c->pring; }
        if (rings && banded_rings(c)) {
            A.ering = band_ering(c, c->band); A.pring = band_pring(c, c->band);
            int n = 0;
            for (int b = 0; b < c->n_bands; b++) if (b != c->band) { A.peer_ering[n] = band_ering(c, b); A.peer_pring[n] = band_pring(c, b); n++; }
        }
#ifdef TPOSE_DEBUG
        A.dbg = persist_dbg_buffer(c->plan.parts, c->stream);
        { const char* f = getenv("TPOSE_DBG_FIRST"); A.dbg_first = f ? atoi(f) : 0; }
#endif
        tp_launch_persist(A, grid, c->plan.rows_max, c->plan.lds_bytes, c->stream);
        if (banded) tp_launch_band_collect(make_launch(c, p.image_slot, dp), A, c->points_out, c->stream);
        tp_launch_persist_finish(make_launch(c, p.image_slot, dp), c->points_out, c->d_status, c->h_status, c->stream);
        c->journal.push_back({p, rings ? 0 : k});   // (a chunk of tp_iterate_until is checked by its caller: nothing to replay)
        HIP_TRY(c, hipGetLastError());
        c->epoch += (uint32_t)k;
        c->persist_unchecked = true;
        c->persist_launches++; c->persist_iters += k;
        c->iters_since_snap += k; c->iters_since_cut += k;
        n -= k;
        if (c->iters_since_snap >= PK_CHUNK / 2) {   // the positions after this chunk, for a later maybe_replan
            const int sl = c->snap_next;
            HIP_TRY(c, hipMemcpyAsync(c->snap_host[sl], c->points, sizeof(float) * 2 * (size_t)c->NP, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipEventRecord(c->snap_ev[sl], c->stream));
            c->snap_pending[sl] = true;
            c->snap_next = sl ^ 1;
            c->iters_since_snap = 0;
        }
        // with this chunk on the stream (the GPU has work while the host cuts): does the mesh want a new plan?
        if (int rc = maybe_replan(c, dp, n > 0)) return rc;
    }
    return TP_OK;
}

// enqueue one grad-iter on the context stream (no sync)
// the frame mirror of this triangulation (pinned; grown when the triangulation outgrows it) into a launch
int frame_mirror_into(tp_context* c, tp_launch& L) {
    const int fn = (int)std::min<size_t>((size_t)13 * c->NT, (size_t)c->NT + 64);
    const size_t need = (size_t)8 * fn + (size_t)8 * c->NP;
    if (need > c->frame_mirror_bytes) {
        if (c->frame_mirror) hipHostFree(c->frame_mirror);
        c->frame_mirror = nullptr; c->frame_mirror_bytes = 0;
        HIP_TRY(c, hipHostMalloc((void**)&c->frame_mirror, need * 2, hipHostMallocDefault));
        c->frame_mirror_bytes = need * 2;
    }
    if (fn != c->frame_n || (size_t)c->NP != c->frame_np) { c->ten_stamp = c->pts_stamp = ~0ull; }   // (another layout: nothing in it is current)
    c->frame_n = fn; c->frame_np = (size_t)c->NP;
    L.mirror_n = fn;
    L.mirror_ten = (int32_t*)c->frame_mirror;
    L.mirror_cn = L.mirror_ten + fn;
    L.mirror_pts = (float2*)(L.mirror_cn + fn);
    return TP_OK;
}
int enqueue_iter(tp_context* c, const tp_params& p, float dp, bool mirror = false) {
    tp_launch L = make_launch(c, p.image_slot, dp);
    if (mirror) { if (int rc = frame_mirror_into(c, L)) return rc; }
    tp_launch_lines(L, c->stream);                       // vertex stage + the nine line sums of every edge
    tp_launch_update(L, p.flavour, p.rate, c->stream);  // variants + gradient + shift
    return TP_OK;
}

int enqueue_iters(tp_context* c, const tp_params* p, int n_iters);

}  // namespace

extern "C" {

int tp_abi_version(void) { return TP_ABI_VERSION; }

int tp_device_count(int* count) {
    api_guard api_lock;
    if (!count) return TP_ERR_INVALID;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(nullptr, TP_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *count = n;
    return TP_OK;
}

const char* tp_last_error(const tp_context* ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

int tp_create(int device, int width, int height, tp_context** out) {
    api_guard api_lock;
    if (!out) return fail(nullptr, TP_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (width < 1 || height < 1 || width > TP_MAX_RASTER || height > TP_MAX_RASTER)
        return fail(nullptr, TP_ERR_CAPACITY, "raster %dx%d outside 1..%d", width, height, TP_MAX_RASTER);
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(nullptr, TP_ERR_NO_DEVICE, "no HIP device available (the HIP path has no CPU fallback)");
    if (device < 0 || device >= n) return fail(nullptr, TP_ERR_NO_DEVICE, "device %d out of range (0..%d)", device, n - 1);
    HIP_TRY(nullptr, hipSetDevice(device));
    tp_context* c = new tp_context();
    c->device = device; c->W = width; c->H = height;
    c->prefix_pitch = tp_prefix_pitch(width);
    // (pixel records, the persistent kernel's table: rasters of at most 4096 columns AND rows -- a line's sums of r and g share
    // a 64-bit word in LDS, tp_persist.h: pk_fold_words)
    c->px_pitch = width <= TP_PX_MAXW && height <= TP_PX_MAXW ? tp_px_pitch(width) : 0;
    c->ratio = (float)width / (float)height;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->num_cus = prop.multiProcessorCount;
    }
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&c->ev0);
    if (e == hipSuccess) e = hipEventCreate(&c->ev1);
    if (e != hipSuccess) {
        int rc = fail(nullptr, TP_ERR_HIP, "context setup failed: %s", hipGetErrorString(e));
        tp_destroy(c);
        return rc;
    }
    *out = c;
    return TP_OK;
}

int tp_destroy(tp_context* c) {
    api_guard api_lock;
    if (!c) return TP_OK;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    stop_replan_worker(c);
    drop_graphs(c);
    free_triangulation(c);
    hipFree(c->img[0]); hipFree(c->img[1]); hipFree(c->prefix[0]); hipFree(c->prefix[1]); hipFree(c->px[0]); hipFree(c->px[1]);
    hipFree(c->render_pic); hipFree(c->render_pts);
    hipFree(c->d_wg); hipFree(c->d_pool); hipFree(c->posbox); hipFree(c->points_out); hipFree(c->d_status);
    if (c->h_status) hipHostFree(c->h_status);
    if (c->frame_mirror) hipHostFree(c->frame_mirror);
    hipFree(c->ering); hipFree(c->pring);
    if (c->ering_host) hipHostFree(c->ering_host);
    for (int k = 0; k < 2; k++) {
        hipFree(c->plan_dev[k].wg); hipFree(c->plan_dev[k].pool);
        if (c->plan_dev[k].stage) hipHostFree(c->plan_dev[k].stage);
        if (c->snap_host[k]) hipHostFree(c->snap_host[k]);
        if (c->snap_ev[k]) hipEventDestroy(c->snap_ev[k]);
    }
    if (c->pinned) hipHostFree(c->pinned);
    if (c->up_pinned) hipHostFree(c->up_pinned);
    if (c->ev0) hipEventDestroy(c->ev0);
    if (c->ev1) hipEventDestroy(c->ev1);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
    return TP_OK;
}

int tp_set_ratio(tp_context* c, float ratio) {
    api_guard api_lock;
    if (!c) return TP_ERR_INVALID;
    c->mutations++;   // (a retrieve no longer finds what the frame mirror holds)
    if (!(ratio > 0.0f)) return fail(c, TP_ERR_INVALID, "RATIO must be positive");
    if (ratio != c->ratio) { c->ratio = ratio; c->generation++; c->accumulated = c->energized = false; }
    return TP_OK;
}

int tp_set_dp(tp_context* c, float dp) {
    api_guard api_lock;
    if (!c) return TP_ERR_INVALID;
    c->mutations++;   // (a retrieve no longer finds what the frame mirror holds)
    c->dp_override = dp;  // piecewise calls only; tp_iterate takes dp from its params (part of the graph key)
    c->accumulated = c->energized = false;
    return TP_OK;
}

int tp_set_option(tp_context* c, int option, int64_t value) {
    api_guard api_lock;
    if (!c) return TP_ERR_INVALID;
    c->mutations++;   // (a retrieve no longer finds what the frame mirror holds)
    switch (option) {
        case TP_OPT_PERSISTENT:
            if (value != TP_PERSIST_OFF && value != TP_PERSIST_AUTO) return fail(c, TP_ERR_INVALID, "TP_OPT_PERSISTENT: bad value %lld", (long long)value);
            c->persist_mode = (int)value;
            return TP_OK;
        default: return fail(c, TP_ERR_INVALID, "unknown option %d", option);
    }
}

size_t tp_band_mailbox_bytes(int points, int triangles) {
    const size_t cap = (size_t)(points > 0 ? points : 0), ct = (size_t)(triangles > 0 ? triangles : 0);
    return band_slots_bytes(cap) + band_ering_bytes(ct) + (size_t)PK_RING_FRAMES * cap * 8 + 256;
}

int tp_band_attach(tp_context* c, int band, int n_bands, void* const* mailboxes, size_t bytes_each, int points, int triangles, int patches_per_band) {
    api_guard api_lock;
    if (!c) return TP_ERR_INVALID;
    c->mutations++;   // (a retrieve no longer finds what the frame mirror holds)
    HIP_TRY(c, hipSetDevice(c->device));
    if (int rc = tp_synchronize(c)) return rc;   // nothing in flight reads the mailbox or the plan
    if (n_bands <= 1) {
        c->band = 0; c->n_bands = 1; c->band_patches = 0; c->band_cap = 0; c->band_cap_tris = 0;
        for (auto& b : c->band_box) b = nullptr;
        c->plan_generation = 0;
        return TP_OK;
    }
    if (n_bands > PK_MAX_PEERS + 1 || band < 0 || band >= n_bands || !mailboxes) return fail(c, TP_ERR_INVALID, "band attach: band %d of %d (at most %d bands)", band, n_bands, PK_MAX_PEERS + 1);
    for (int b = 0; b < n_bands; b++) if (!mailboxes[b]) return fail(c, TP_ERR_INVALID, "band attach: mailbox %d is NULL", b);
    if (c->num_cus < 1) return fail(c, TP_ERR_STATE, "band attach: no compute units reported");
    const int ppb = patches_per_band > 0 ? patches_per_band : c->num_cus * PK_WG_PER_CU;
    if (ppb > c->num_cus * PK_WG_PER_CU) return fail(c, TP_ERR_CAPACITY, "band attach: %d patches per band on %d compute units", ppb, c->num_cus);
    if (points < 1 || triangles < 0 || tp_band_mailbox_bytes(points, triangles) > bytes_each)
        return fail(c, TP_ERR_INVALID, "band attach: %zu bytes for %d points, %d triangles", bytes_each, points, triangles);
    const size_t cap = (size_t)points, cap_tris = (size_t)triangles;
    c->band = band; c->n_bands = n_bands; c->band_patches = ppb;
    c->band_cap = cap; c->band_cap_tris = cap_tris;
    for (int b = 0; b < n_bands; b++) c->band_box[b] = (unsigned long long*)mailboxes[b];
    c->plan_generation = 0;   // the plan is cut again, into n_bands * ppb patches
    return TP_OK;
}

int tp_get_ratio(const tp_context* c, float* ratio) {
    if (!c || !ratio) return TP_ERR_INVALID;
    *ratio = c->ratio;
    return TP_OK;
}

static int set_image_common(tp_context* c, int slot, const void* src, size_t stride, hipMemcpyKind kind) {
    if (!c) return TP_ERR_INVALID;
    c->mutations++;
    if (slot != TP_IMAGE_A && slot != TP_IMAGE_B) return fail(c, TP_ERR_INVALID, "bad image slot %d", slot);
    if (!src) return fail(c, TP_ERR_INVALID, "image pointer is NULL");
    if (stride < (size_t)c->W * 4) return fail(c, TP_ERR_INVALID, "stride %zu < 4*width", stride);
    HIP_TRY(c, hipSetDevice(c->device));
    if (!c->img[slot]) HIP_TRY(c, dev_alloc(&c->img[slot], (size_t)c->W * c->H * 4));
    HIP_TRY(c, hipMemcpy2DAsync(c->img[slot], (size_t)c->W * 4, src, stride, (size_t)c->W * 4, c->H, kind, c->stream));
    // row prefix table of this image (32-byte records per four pixels: 8 bytes per pixel): what k_lines reads, grad-iter after grad-iter
    if (!c->prefix[slot]) HIP_TRY(c, dev_alloc(&c->prefix[slot], (size_t)c->H * c->prefix_pitch * 2));
    tp_launch_prefix_table(c->img[slot], c->W * 4, c->W, c->H, c->prefix_pitch, c->prefix[slot], c->stream);
    if (c->px_pitch) {  // pixel records of the same sums: what the persistent grad-iter kernel reads
        if (!c->px[slot]) HIP_TRY(c, dev_alloc(&c->px[slot], (size_t)c->H * c->px_pitch));
        tp_launch_px_table(c->img[slot], c->W * 4, c->W, c->H, c->px_pitch, c->px[slot], c->stream);
    }
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->have_img[slot] = true;
    c->accumulated = c->energized = false;
    return TP_OK;
}

int tp_set_image(tp_context* c, int slot, const uint8_t* rgba, size_t stride) {
    api_guard api_lock;
    return set_image_common(c, slot, rgba, stride, hipMemcpyHostToDevice);
}

int tp_set_image_device(tp_context* c, int slot, const void* dev, size_t stride) {
    api_guard api_lock;
    return set_image_common(c, slot, dev, stride, hipMemcpyDeviceToDevice);
}

int tp_upload(tp_context* c, const float* points, int NP, const int32_t* tris, int NT, const int32_t* colors) {
    api_guard api_lock;
    if (!c) return TP_ERR_INVALID;
    c->mutations++;   // (a retrieve no longer finds what the frame mirror holds)
    if (!points || !tris || NP < 1 || NT < 1) return fail(c, TP_ERR_INVALID, "upload: bad arguments (NP=%d NT=%d)", NP, NT);
    if ((size_t)13 * NT > (size_t)TP_MAXT) return fail(c, TP_ERR_CAPACITY, "13*NT = %d exceeds MAXT = %d", 13 * NT, TP_MAXT);
    if ((size_t)NP > (size_t)TP_MAXT) return fail(c, TP_ERR_CAPACITY, "NP = %d exceeds MAXT = %d", NP, TP_MAXT);
    for (int t = 0; t < NT; t++)
        for (int s = 0; s < 3; s++) {
            const int v = tris[4 * t + s];
            if (v < 0 || v >= NP) return fail(c, TP_ERR_INVALID, "triangle %d references vertex %d (NP=%d)", t, v, NP);
        }
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (int rc = check_persist_status(c)) return rc;

    if (NT > c->capT || NP > c->capP) {
        free_triangulation(c);
        const int capT = NT + NT / 2 + 64, capP = NP + NP / 2 + 64;
        HIP_TRY(c, dev_alloc(&c->points, capP));
        HIP_TRY(c, dev_alloc(&c->gr, capP));
        HIP_TRY(c, dev_alloc(&c->vtx_off, capP + 1));
        HIP_TRY(c, dev_alloc(&c->vref, (size_t)capP * 64));
        HIP_TRY(c, dev_alloc(&c->vvar, (size_t)capP * 8));
        HIP_TRY(c, dev_alloc(&c->tris, capT));
        HIP_TRY(c, dev_alloc(&c->colors, capT));
        HIP_TRY(c, dev_alloc(&c->vtx_adj, (size_t)3 * capT));
        HIP_TRY(c, dev_alloc(&c->he_edge, (size_t)3 * capT));
        HIP_TRY(c, dev_alloc(&c->vpos, (size_t)5 * capP));
        HIP_TRY(c, dev_alloc(&c->ten, (size_t)13 * capT));
        HIP_TRY(c, dev_alloc(&c->cn, (size_t)13 * capT));
        HIP_TRY(c, dev_alloc(&c->ca, (size_t)13 * capT));
        HIP_TRY(c, dev_alloc(&c->moments, (size_t)13 * capT * 6));
        HIP_TRY(c, hipMemset(c->ca, 0, sizeof(int4) * 13 * (size_t)capT));
        HIP_TRY(c, hipMemset(c->ten, 0, sizeof(int32_t) * 13 * (size_t)capT));
        HIP_TRY(c, hipMemset(c->cn, 0, sizeof(int32_t) * 13 * (size_t)capT));
        HIP_TRY(c, hipMemset(c->gr, 0, sizeof(int2) * (size_t)capP));
        c->capT = capT; c->capP = capP;
    }
    // undirected edges: every half-edge (o -> d) maps to the edge {min, max} and a direction bit.  Flat
    // open-addressing table kept in the context (uploads follow every topology update of the schedule).
    std::vector<int> he_edge((size_t)3 * NT);
    std::vector<int> edge_uv;  // 2 ints per edge
    edge_uv.reserve((size_t)6 * NT);
    {
        size_t cap = 1024;
        while (cap < (size_t)8 * NT) cap <<= 1;
        if (c->hkeys.size() != cap) { c->hkeys.assign(cap, 0); c->hvals.assign(cap, 0); c->hstamp.assign(cap, 0); c->hgen = 0; }
        if (++c->hgen == 0) { std::fill(c->hstamp.begin(), c->hstamp.end(), 0u); c->hgen = 1; }
        const uint32_t gen = c->hgen;
        const size_t hmask = cap - 1;
        for (int t = 0; t < NT; t++)
            for (int k = 0; k < 3; k++) {
                const int o = tris[4 * t + k], d = tris[4 * t + (k + 1) % 3];
                const int u = o < d ? o : d, v = o < d ? d : o;
                const uint64_t key = ((uint64_t)(uint32_t)u << 32) | (uint32_t)v;
                size_t slot = (size_t)((key * 0x9E3779B97F4A7C15ull) >> 20) & hmask;
                while (c->hstamp[slot] == gen && c->hkeys[slot] != key) slot = (slot + 1) & hmask;
                if (c->hstamp[slot] != gen) {
                    c->hstamp[slot] = gen; c->hkeys[slot] = key; c->hvals[slot] = (int)(edge_uv.size() / 2);
                    edge_uv.push_back(u); edge_uv.push_back(v);
                }
                he_edge[(size_t)3 * t + k] = c->hvals[slot] * 2 + (o != u ? 1 : 0);
            }
    }
    const int NE = (int)(edge_uv.size() / 2);
    // bit 30 of an endpoint id: this edge is the one that publishes the vertex's snapped positions (vpos) -- the
    // first edge that mentions the vertex; the others would only repeat the same stores
    {
        std::vector<char> owned((size_t)NP, 0);
        for (size_t k = 0; k < edge_uv.size(); k++) {
            const int v = edge_uv[k];
            if (!owned[v]) { owned[v] = 1; edge_uv[k] = v | (1 << 30); }
        }
    }
    if (NE > c->capE) {
        hipFree(c->edge_uv); hipFree(c->epos); hipFree(c->wline);
        c->edge_uv = nullptr; c->epos = nullptr; c->wline = nullptr;
        const int capE = NE + NE / 2 + 64;
        HIP_TRY(c, dev_alloc(&c->edge_uv, capE));
        HIP_TRY(c, dev_alloc(&c->epos, (size_t)2 * capE));
        HIP_TRY(c, dev_alloc(&c->wline, (size_t)capE * TP_NLINES * TP_W_WORDS));
        c->capE = capE;
    }
    c->NE = NE;
    // lanes per line of k_lines: about eight rows per lane at the mean height of an edge (a speed hint only)
    {
        double rows = 0.0;
        for (int e = 0; e < NE; e++) {
            const int u = edge_uv[(size_t)2 * e] & 0x3fffffff, v = edge_uv[(size_t)2 * e + 1] & 0x3fffffff;
            const double dy = (double)points[2 * (size_t)u + 1] - (double)points[2 * (size_t)v + 1];
            rows += dy < 0 ? -dy : dy;
        }
        rows = rows * 0.5 * (double)c->H / (double)(NE > 0 ? NE : 1);
        int lpl = 1;
        while (lpl < 1024 && rows > (double)TP_LINES_ROWS * lpl) lpl <<= 1;
        c->lanes_per_line = lpl;
    }

    // vertex -> outgoing half-edge ids (3t+s), the gather form of gradient.cs' scatter
    std::vector<int> off(NP + 1, 0), adj((size_t)3 * NT);
    for (int t = 0; t < NT; t++) for (int s = 0; s < 3; s++) off[tris[4 * t + s] + 1]++;
    for (int v = 0; v < NP; v++) off[v + 1] += off[v];
    {
        std::vector<int> cur(off.begin(), off.end() - 1);
        for (int t = 0; t < NT; t++) for (int s = 0; s < 3; s++) adj[cur[tris[4 * t + s]]++] = 3 * t + s;
    }
    // everything goes through one pinned staging buffer and rides the stream: no wait at the end (the next
    // upload waits for the stream before it touches the staging buffer again)
    {
        struct part { void* dst; const void* src; size_t bytes; };
        const part parts[] = {
            {c->edge_uv, edge_uv.data(), sizeof(int) * 2 * (size_t)NE},
            {c->he_edge, he_edge.data(), sizeof(int) * 3 * (size_t)NT},
            {c->points, points, sizeof(float) * 2 * (size_t)NP},
            {c->tris, tris, sizeof(int32_t) * 4 * (size_t)NT},
            {c->vtx_off, off.data(), sizeof(int) * (size_t)(NP + 1)},
            {c->vtx_adj, adj.data(), sizeof(int) * 3 * (size_t)NT},
            {c->colors, colors, colors ? sizeof(int32_t) * 4 * (size_t)NT : 0},
        };
        static_assert(sizeof(parts) / sizeof(parts[0]) <= TP_COPY_MAX, "one copy list");
        size_t total = 0;
        for (auto& pt : parts) total += (pt.bytes + 255) & ~(size_t)255;
        if (total > c->up_pinned_bytes) {
            if (c->up_pinned) hipHostFree(c->up_pinned);
            c->up_pinned = nullptr; c->up_pinned_bytes = 0;
            HIP_TRY(c, hipHostMalloc((void**)&c->up_pinned, total * 2, hipHostMallocDefault));
            c->up_pinned_bytes = total * 2;
        }
        size_t o = 0;
        tp_copy_list G{};   // (small uploads -- the schedules' -- are fetched from the staging buffer by one kernel)
        const bool one_launch = total <= ((size_t)4 << 20);
        for (auto& pt : parts) {
            if (!pt.bytes) continue;
            memcpy(c->up_pinned + o, pt.src, pt.bytes);
            if (one_launch) { G.src[G.n] = (const uint32_t*)(c->up_pinned + o); G.dst[G.n] = (uint32_t*)pt.dst; G.words[G.n] = (uint32_t)(pt.bytes / 4); G.n++; }
            else HIP_TRY(c, hipMemcpyAsync(pt.dst, c->up_pinned + o, pt.bytes, hipMemcpyHostToDevice, c->stream));
            o += (pt.bytes + 255) & ~(size_t)255;
        }
        if (one_launch) { tp_launch_copy_list(G, c->stream); HIP_TRY(c, hipGetLastError()); }
    }
    c->NT = NT; c->NP = NP;
    c->h_points.assign(points, points + 2 * (size_t)NP);
    c->h_tris.assign(tris, tris + 4 * (size_t)NT);
    c->h_edge_uv.swap(edge_uv); c->h_he_edge.swap(he_edge);
    c->have_colors = colors != nullptr;
    {   // reference tables of the fused update (device side: reads the arrays just copied)
        tp_launch L = make_launch(c, 0, 0.0f);
        tp_launch_vertex_refs(L, c->vref, c->vvar, c->stream);
        HIP_TRY(c, hipGetLastError());
    }
    if (colors) {
        tp_launch L = make_launch(c, 0, 0.0f);
        tp_launch_replicate_colors(L, c->stream);
        HIP_TRY(c, hipGetLastError());
    }
    c->generation++;  // captured graphs bake NT, NP, dp and buffer addresses
    c->uploaded = true; c->accumulated = c->energized = false;
    return TP_OK;
}

int tp_accumulate(tp_context* c, int flavour, int slot) {
    api_guard api_lock;
    if (!c) return TP_ERR_INVALID;
    c->mutations++;   // (a retrieve no longer finds what the frame mirror holds)
    if (flavour != TP_TRIANGULATE && flavour != TP_WARP) return fail(c, TP_ERR_INVALID, "bad flavour %d", flavour);
    if (!c->uploaded) return fail(c, TP_ERR_STATE, "accumulate before upload");
    if (int rc = check_slot(c, slot)) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    if (int rc = settle_persistent(c)) return rc;
    tp_launch L = make_launch(c, slot, resolve_dp(c, flavour, c->dp_override));
    tp_launch_lines(L, c->stream);
    HIP_TRY(c, hipGetLastError());
    c->acc_slot = slot; c->acc_flavour = flavour;
    c->accumulated = true; c->energized = false;
    return TP_OK;
}

int tp_energy(tp_context* c, int flavour) {
    api_guard api_lock;
    if (!c) return TP_ERR_INVALID;
    c->mutations++;   // (a retrieve no longer finds what the frame mirror holds)
    if (flavour != TP_TRIANGULATE && flavour != TP_WARP) return fail(c, TP_ERR_INVALID, "bad flavour %d", flavour);
    if (!c->accumulated) return fail(c, TP_ERR_STATE, "energy before accumulate");
    if (flavour != c->acc_flavour) return fail(c, TP_ERR_STATE, "energy flavour %d differs from the accumulate pass (%d)", flavour, c->acc_flavour);
    HIP_TRY(c, hipSetDevice(c->device));
    if (int rc = settle_persistent(c)) return rc;
    tp_launch L = make_launch(c, c->acc_slot, 0.0f);
    if (int rc = frame_mirror_into(c, L)) return rc;
    tp_launch_finalize(L, flavour, true, c->stream);
    HIP_TRY(c, hipGetLastError());
    c->ten_stamp = c->mutations;   // (the first entries of `tenergy` / `colnum` are in the frame mirror as well)
    c->energized = true; c->last_flavour = flavour;
    return TP_OK;
}

int tp_shift(tp_context* c, float rate) {
    api_guard api_lock;
    if (!c) return TP_ERR_INVALID;
    c->mutations++;   // (a retrieve no longer finds what the frame mirror holds)
    if (!c->energized) return fail(c, TP_ERR_STATE, "shift before energy");
    HIP_TRY(c, hipSetDevice(c->device));
    if (int rc = settle_persistent(c)) return rc;
    tp_launch L = make_launch(c, c->acc_slot, 0.0f);
    const bool energies_current = c->ten_stamp == c->mutations - 1;   // (tp_energy was the call before: the step leaves its energies alone)
    if (int rc = frame_mirror_into(c, L)) return rc;
    tp_launch_shift(L, rate, c->stream);
    HIP_TRY(c, hipGetLastError());
    c->pts_stamp = c->mutations;
    if (energies_current && c->ten_stamp != ~0ull) c->ten_stamp = c->mutations;
    c->accumulated = c->energized = false;  // geometry moved
    return TP_OK;
}

void tp_default_params(int flavour, tp_params* p) {
    if (!p) return;
    p->flavour = flavour;
    p->image_slot = flavour == TP_WARP ? TP_IMAGE_B : TP_IMAGE_A;
    p->rate = flavour == TP_WARP ? 0.00003f : 0.00005f;  // shift.cs:45 of each program
    p->dp = 0.0f;
}

static int validate_params(tp_context* c, const tp_params* p, int n_iters) {
    if (!p) return fail(c, TP_ERR_INVALID, "params is NULL");
    if (n_iters < 0) return fail(c, TP_ERR_INVALID, "n_iters < 0");
    if (p->flavour != TP_TRIANGULATE && p->flavour != TP_WARP) return fail(c, TP_ERR_INVALID, "bad flavour %d", p->flavour);
    if (!c->uploaded) return fail(c, TP_ERR_STATE, "iterate before upload");
    return check_slot(c, p->image_slot);
}

}  // extern "C"

namespace {
#ifndef TP_GRAPH_CHUNK
#define TP_GRAPH_CHUNK 16
#endif
const int CHUNK = TP_GRAPH_CHUNK;  // fused iterations per graph: hides the ~10 us replay floor; a remainder runs eagerly

// Capture whatever `body` enqueues on the context stream into an executable graph.  On any failure the
// capture is ended and the partial graph destroyed, so the stream never stays in capture mode.
template <class F>
int capture_graph(tp_context* c, F&& body, hipGraphExec_t* exec) {
    capture_guard capture_lock;
    hipGraph_t graph = nullptr;
    *exec = nullptr;
    hipError_t err = hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed);
    if (err != hipSuccess) return fail(c, TP_ERR_HIP, "hipStreamBeginCapture: %s", hipGetErrorString(err));
    body();
    hipError_t launch_err = hipGetLastError();  // launch errors raised while capturing
    err = hipStreamEndCapture(c->stream, &graph);
    if (err == hipSuccess && launch_err != hipSuccess) err = launch_err;
    if (err == hipSuccess) err = hipGraphInstantiate(exec, graph, nullptr, nullptr, 0);
    if (graph) hipGraphDestroy(graph);
    if (err != hipSuccess) {
        if (*exec) { hipGraphExecDestroy(*exec); *exec = nullptr; }
        return fail(c, TP_ERR_HIP, "graph capture: %s", hipGetErrorString(err));
    }
    return TP_OK;
}

// the graph of CHUNK fused grad-iters for these parameters (captured once per upload / parameter set)
int chunk_graph(tp_context* c, const tp_params* p, float dp, graph_entry** out) {
    for (auto& e : c->graphs)
        if (e.generation == c->generation && e.iters == CHUNK && memcmp(&e.params, p, sizeof *p) == 0) { *out = &e; return TP_OK; }
    graph_entry e;
    if (int rc = capture_graph(c, [&] { for (int k = 0; k < CHUNK; k++) enqueue_iter(c, *p, dp); }, &e.exec)) return rc;
    e.params = *p; e.iters = CHUNK; e.generation = c->generation;
    if (c->graphs.size() > 8) drop_graphs(c);
    c->graphs.push_back(e);
    *out = &c->graphs.back();
    return TP_OK;
}

// enqueue n fused grad-iters on the context stream and remember them until the flags were checked
int enqueue_iters(tp_context* c, const tp_params* p, int n_iters) {
    const float dp = resolve_dp(c, p->flavour, p->dp);

    int left = n_iters;
    if (left >= PK_MIN_ITERS) {
        // inside persistent launches; the last grad-iter also writes the buffers the reference reads back
        bool use = false;
        if (int rc = ensure_plan(c, dp, &use)) return rc;
        if (use) {
            if (int rc = enqueue_persistent(c, *p, dp, left)) return rc;
            left = 0;
        }
    }
    if (left > 0) { if (int rc = settle_persistent(c)) return rc; }
    if (n_iters == 1 && left == 1) {
        // a single frame (the schedules run them one by one and read back after each): its outputs also go to the frame mirror
        if (int rc = enqueue_iter(c, *p, dp, true)) return rc;
        HIP_TRY(c, hipGetLastError());
        c->ten_stamp = c->pts_stamp = c->mutations;
    } else if (int rc = enqueue_two_kernel(c, p, dp, left)) return rc;
    c->acc_slot = p->image_slot; c->last_flavour = p->flavour;
    c->accumulated = c->energized = false;
    return TP_OK;
}

// n grad-iters as k_lines + k_update each: whole chunks as graph replays, the rest eagerly
int enqueue_two_kernel(tp_context* c, const tp_params* p, float dp, int left) {
    if (left >= CHUNK) {
        graph_entry* g = nullptr;
        if (int rc = chunk_graph(c, p, dp, &g)) return rc;
        while (left >= CHUNK) {
            HIP_TRY(c, hipGraphLaunch(g->exec, c->stream));
            left -= CHUNK;
        }
    }
    for (int k = 0; k < left; k++) enqueue_iter(c, *p, dp);
    HIP_TRY(c, hipGetLastError());
    return TP_OK;
}
}  // namespace


extern "C" {

int tp_iterate(tp_context* c, const tp_params* p, int n_iters) {
    api_guard api_lock;
    if (!c) return TP_ERR_INVALID;
    if (int rc = validate_params(c, p, n_iters)) return rc;
    if (n_iters == 0) return TP_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    c->mutations++;
    return enqueue_iters(c, p, n_iters);
}

int tp_prepare(tp_context* c, const tp_params* p) {
    api_guard api_lock;
    if (!c) return TP_ERR_INVALID;
    if (int rc = validate_params(c, p, 0)) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    bool use = false;
    if (int rc = ensure_plan(c, resolve_dp(c, p->flavour, p->dp), &use)) return rc;
    if (use) return TP_OK;
    graph_entry* g = nullptr;
    return chunk_graph(c, p, resolve_dp(c, p->flavour, p->dp), &g);
}

int tp_iterate_until(tp_context* c, const tp_params* p, int max_frames, double threshold, float* toterr, int* frames, float* relerr) {
    api_guard api_lock;
    if (!c) return TP_ERR_INVALID;
    if (int rc = validate_params(c, p, max_frames)) return rc;
    if (!toterr || !frames) return fail(c, TP_ERR_INVALID, "iterate_until: toterr / frames is NULL");
    *frames = 0;
    if (relerr) *relerr = 0.0f;
    if (max_frames == 0) return TP_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    c->mutations++;
    if (int rc = settle_persistent(c)) return rc;
    const float dp = resolve_dp(c, p->flavour, p->dp);
    const int NT = c->NT;
    float tot = *toterr, rel = 0.0f;
    // geterr (source/triangulation.hpp:653-674) on the base energies of one frame: float32, ascending t
    auto frame_err = [&](const int32_t* terr) {
        float newerr = 0.0f;
        for (int i = 0; i < NT; i++) { float err = 0.0f; err += (float)terr[i]; newerr += err; }
        rel = (tot - newerr) / tot;
        tot = newerr;
        return (double)std::fabs(rel);   // (the reference compares the float with a double literal)
    };
    auto host_ring = [&](size_t ints) -> int {
        if (ints <= c->cap_ering_host && c->ering_host) return TP_OK;
        if (c->ering_host) hipHostFree(c->ering_host);
        c->ering_host = nullptr; c->cap_ering_host = 0;
        HIP_TRY(c, hipHostMalloc((void**)&c->ering_host, ints * sizeof(int32_t), hipHostMallocDefault));
        c->cap_ering_host = ints;
        return TP_OK;
    };
    bool use = false;
    if (max_frames >= PK_MIN_ITERS && (c->n_bands == 1 || banded_rings(c))) { if (int rc = ensure_plan(c, dp, &use, true)) return rc; }
    if (banded_rings(c) && (size_t)NT > c->band_cap_tris) use = false;   // (more triangles than the bands' rings were sized for: every band on its own)
    int done = 0, chunk = 32;
    bool converged = false;
    while (done < max_frames && !converged) {
        const int left = max_frames - done;
        if (!use || left < PK_MIN_ITERS) {
            // frame by frame on the two-kernel path: one frame, then the base energies come back
            if (int rc = host_ring((size_t)NT)) return rc;
            enqueue_iter(c, *p, dp);
            HIP_TRY(c, hipGetLastError());
            HIP_TRY(c, hipMemcpyAsync(c->ering_host, c->ten, sizeof(int32_t) * (size_t)NT, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, wait_stream(c->stream));
            done++;
            converged = frame_err(c->ering_host) < threshold;
            continue;
        }
        // a chunk of frames inside one persistent launch: every frame leaves its base energies and its starting positions
        const int C = left < chunk ? left : chunk;
        const bool shared = banded_rings(c);   // (band split: the rings are in the bands' mailboxes, every band writes into all of them)
        if (!shared) {
            if (int rc = grow(c, &c->ering, &c->cap_ering, (size_t)C * NT)) return rc;
            if (int rc = grow(c, &c->pring, &c->cap_pring, (size_t)C * c->NP)) return rc;
        }
        const int32_t* ering = shared ? band_ering(c, c->band) : c->ering;
        const float2* pring = shared ? band_pring(c, c->band) : c->pring;
        if (int rc = host_ring((size_t)256 * NT)) return rc;   // (for the longest chunk at once: freeing pinned memory waits for the device)
        if (int rc = enqueue_persistent(c, *p, dp, C, true)) return rc;
        HIP_TRY(c, hipMemcpyAsync(c->ering_host, ering, sizeof(int32_t) * (size_t)C * NT, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, wait_stream(c->stream));
        {
            const int64_t fails = c->persist_failures;
            if (int rc = check_persist_status(c)) return rc;
            if (c->persist_failures != fails) { use = false; continue; }   // the chunk gave up (nothing changed): frame by frame from here
        }
        int j = 0;
        for (; j < C; j++) {
            done++;
            if (frame_err(c->ering_host + (size_t)j * NT) < threshold) { converged = true; break; }
        }
        if (converged || done >= max_frames) {
            // back to the start of the last frame that counts, and that frame once more on the two-kernel path: it writes the
            // buffers the reference reads back (`tenergy`, `colnum`, `colacc`, `gradient`) and takes the step
            const int last = converged ? j : C - 1;
            tp_launch_persist_finish(make_launch(c, p->image_slot, dp), pring + (size_t)last * c->NP, nullptr, nullptr, c->stream);
            enqueue_iter(c, *p, dp);
            HIP_TRY(c, hipGetLastError());
            break;
        }
        if (chunk < 256) chunk *= 2;
    }
    if (!use || max_frames < PK_MIN_ITERS) { /* (the two-kernel frames left the buffers of the last frame in place) */ }
    c->acc_slot = p->image_slot; c->last_flavour = p->flavour;
    c->accumulated = c->energized = false;
    *toterr = tot; *frames = done;
    if (relerr) *relerr = rel;
    return TP_OK;
}

int tp_profile_iterate(tp_context* c, const tp_params* p, int n_iters, double* accumulate_us) {
    api_guard api_lock;
    if (!c || !accumulate_us) return TP_ERR_INVALID;
    c->mutations++;   // (a retrieve no longer finds what the frame mirror holds)
    if (int rc = validate_params(c, p, n_iters)) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    if (int rc = settle_persistent(c)) return rc;
    const float dp = resolve_dp(c, p->flavour, p->dp);
    // Eager launches enqueued back to back (no host sync in between); every k_lines dispatch
    // carries its own begin/end timestamps in an event pair.  (Timed launches record nothing when
    // captured into a hipGraph, so the fused path itself cannot be bracketed; the same kernel inside
    // a graph replay runs ~1-2 us shorter -- see profiles/.)
    std::vector<hipEvent_t> ev((size_t)2 * n_iters);
    for (auto& e : ev) HIP_TRY(c, hipEventCreate(&e));
    for (int k = 0; k < n_iters; k++) {
        tp_launch L = make_launch(c, p->image_slot, dp);
        tp_launch_lines(L, c->stream, ev[2 * k], ev[2 * k + 1]);
        tp_launch_update(L, p->flavour, p->rate, c->stream);
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    double total_ms = 0.0;
    for (int k = 0; k < n_iters; k++) {
        float ms = 0.0f;
        HIP_TRY(c, hipEventElapsedTime(&ms, ev[2 * k], ev[2 * k + 1]));
        total_ms += ms;
    }
    for (auto& e : ev) hipEventDestroy(e);
    *accumulate_us = n_iters ? total_ms * 1000.0 / n_iters : 0.0;
    c->accumulated = c->energized = false;
    return TP_OK;
}

int tp_profile_accumulate(tp_context* c, const tp_params* p, int launches, double* accumulate_us) {
    api_guard api_lock;
    if (!c || !accumulate_us) return TP_ERR_INVALID;
    c->mutations++;   // (a retrieve no longer finds what the frame mirror holds)
    if (int rc = validate_params(c, p, launches)) return rc;
    if (launches < 1) return fail(c, TP_ERR_INVALID, "launches < 1");
    if (int rc = tp_synchronize(c)) return rc;
    const float dp = resolve_dp(c, p->flavour, p->dp);
    tp_launch L = make_launch(c, p->image_slot, dp);
    hipGraphExec_t exec = nullptr;
    hipError_t err;
    if (int rc = capture_graph(c, [&] { for (int k = 0; k < launches; k++) tp_launch_lines(L, c->stream); }, &exec)) return rc;
    if (!c->ev0) { HIP_TRY(c, hipEventCreate(&c->ev0)); HIP_TRY(c, hipEventCreate(&c->ev1)); }
    float ms = 0.0f;
    err = hipGraphLaunch(exec, c->stream);  // warm-up replay
    if (err == hipSuccess) err = hipEventRecord(c->ev0, c->stream);
    if (err == hipSuccess) err = hipGraphLaunch(exec, c->stream);
    if (err == hipSuccess) err = hipEventRecord(c->ev1, c->stream);
    if (err == hipSuccess) err = hipStreamSynchronize(c->stream);
    if (err == hipSuccess) err = hipEventElapsedTime(&ms, c->ev0, c->ev1);
    hipGraphExecDestroy(exec);
    if (err != hipSuccess) return fail(c, TP_ERR_HIP, "profile_accumulate: %s", hipGetErrorString(err));
    *accumulate_us = (double)ms * 1000.0 / launches;
    c->accumulated = c->energized = false;
    return TP_OK;
}

int tp_timer_start(tp_context* c) {
    api_guard api_lock;
    if (!c) return TP_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
    return TP_OK;
}

int tp_timer_stop(tp_context* c, double* elapsed_us) {
    api_guard api_lock;
    if (!c || !elapsed_us) return TP_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
    HIP_TRY(c, wait_event(c->ev1));
    float ms = 0.0f;
    HIP_TRY(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    *elapsed_us = (double)ms * 1000.0;
    return check_persist_status(c);
}

int tp_synchronize(tp_context* c) {
    api_guard api_lock;
    if (!c) return TP_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, wait_stream(c->stream));
    return check_persist_status(c);
}

namespace {
// device source of a tp_buffer; bytes = 0 for the all-zero penergy
int buffer_source(tp_context* c, int what, size_t count, const void** src, size_t* bytes) {
    const size_t V = (size_t)13 * c->NT;
    size_t elem = 4, avail = 0;
    *src = nullptr;
    switch (what) {
        case TP_BUF_TENERGY: *src = c->ten; avail = V; break;
        case TP_BUF_COLNUM: *src = c->cn; avail = V; break;
        case TP_BUF_COLACC: *src = c->ca; avail = 4 * V; break;
        case TP_BUF_POINTS: *src = c->points; avail = 2 * (size_t)c->NP; break;
        case TP_BUF_GRADIENT: *src = c->gr; avail = 2 * (size_t)c->NP; break;
        case TP_BUF_PENERGY: *bytes = 0; return TP_OK;
        case TP_BUF_MOMENTS:
            if (!c->energized) return fail(c, TP_ERR_STATE, "moments are only kept by tp_energy (piecewise API)");
            *src = c->moments; avail = 6 * V; elem = 8; break;
        default: return fail(c, TP_ERR_INVALID, "retrieve: unknown buffer %d", what);
    }
    if (count > avail) return fail(c, TP_ERR_INVALID, "retrieve: count %zu > %zu available", count, avail);
    *bytes = count * elem;
    return TP_OK;
}
}  // namespace

int tp_retrieve(tp_context* c, int what, void* dst, size_t count) {
    api_guard api_lock;
    return tp_retrieve_many(c, 1, &what, &dst, &count);
}

int tp_retrieve_many(tp_context* c, int n, const int* what, void* const* dst, const size_t* count) {
    api_guard api_lock;
    if (!c) return TP_ERR_INVALID;
    if (n < 0 || (n && (!what || !dst || !count))) return fail(c, TP_ERR_INVALID, "retrieve: bad arguments");
    if (!c->uploaded) return fail(c, TP_ERR_STATE, "retrieve before upload");
    HIP_TRY(c, hipSetDevice(c->device));
    // the frame mirror: right behind a single frame the schedules' read-back (first entries of `tenergy` / `colnum`, the points)
    // is already in pinned memory -- wait for the stream and take it from there
    auto mirrored = [&]() {
        if (!c->frame_mirror) return false;
        for (int k = 0; k < n; k++) {
            const bool small = (what[k] == TP_BUF_TENERGY || what[k] == TP_BUF_COLNUM) && count[k] <= (size_t)c->frame_n && c->ten_stamp == c->mutations;
            const bool pts = what[k] == TP_BUF_POINTS && count[k] <= 2 * c->frame_np && c->frame_np == (size_t)c->NP && c->pts_stamp == c->mutations;
            if (!small && !pts && what[k] != TP_BUF_PENERGY) return false;
            if (!dst[k] && count[k]) return false;
        }
        return true;
    };
    if (mirrored()) {
        HIP_TRY(c, wait_stream(c->stream));
        if (int rc = check_persist_status(c)) return rc;
        if (mirrored()) {   // (still: nothing had to be run again)
            const int32_t* ten = (const int32_t*)c->frame_mirror;
            const int32_t* cn = ten + c->frame_n;
            const float* pts = (const float*)(cn + c->frame_n);
            for (int k = 0; k < n; k++) {
                if (what[k] == TP_BUF_TENERGY) memcpy(dst[k], ten, count[k] * 4);
                else if (what[k] == TP_BUF_COLNUM) memcpy(dst[k], cn, count[k] * 4);
                else if (what[k] == TP_BUF_POINTS) memcpy(dst[k], pts, count[k] * 4);
                else memset(dst[k], 0, count[k] * 4);
            }
            return TP_OK;
        }
    }
    std::vector<const void*> src(n);
    std::vector<size_t> bytes(n), off(n);
    size_t total = 0;
    for (int k = 0; k < n; k++) {
        if (!dst[k] && count[k]) return fail(c, TP_ERR_INVALID, "retrieve: dst is NULL");
        if (int rc = buffer_source(c, what[k], count[k], &src[k], &bytes[k])) return rc;
        off[k] = total;
        total += (bytes[k] + 255) & ~(size_t)255;
    }
    if (total > c->pinned_bytes) {
        if (c->pinned) hipHostFree(c->pinned);
        c->pinned = nullptr; c->pinned_bytes = 0;
        HIP_TRY(c, hipHostMalloc((void**)&c->pinned, total + total / 2, hipHostMallocDefault));
        c->pinned_bytes = total + total / 2;
    }
    // everything rides the context's stream behind the enqueued work: ONE wait for the whole batch.  Small batches (the
    // per-frame read-backs of the schedules) are written into the pinned buffer by one kernel instead of one copy command each
    if (n <= TP_COPY_MAX && total <= ((size_t)4 << 20)) {
        tp_copy_list G{};
        for (int k = 0; k < n; k++)
            if (bytes[k]) { G.src[G.n] = (const uint32_t*)src[k]; G.dst[G.n] = (uint32_t*)(c->pinned + off[k]); G.words[G.n] = (uint32_t)(bytes[k] / 4); G.n++; }
        tp_launch_copy_list(G, c->stream);
        HIP_TRY(c, hipGetLastError());
    } else {
        for (int k = 0; k < n; k++)
            if (bytes[k]) HIP_TRY(c, hipMemcpyAsync(c->pinned + off[k], src[k], bytes[k], hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(c, wait_stream(c->stream));
    if (int rc = check_persist_status(c)) return rc;
    for (int k = 0; k < n; k++) {
        if (bytes[k]) memcpy(dst[k], c->pinned + off[k], bytes[k]);
        else if (what[k] == TP_BUF_PENERGY) memset(dst[k], 0, count[k] * 4);
    }
    return TP_OK;
}

int tp_render(tp_context* c, int source, const float* points, uint8_t* dst, size_t stride) {
    api_guard api_lock;
    if (!c) return TP_ERR_INVALID;
    if (!dst) return fail(c, TP_ERR_INVALID, "render: dst is NULL");
    if (source != TP_RENDER_AVERAGE && source != TP_RENDER_STORED) return fail(c, TP_ERR_INVALID, "render: bad source %d", source);
    if (stride < (size_t)c->W * 4) return fail(c, TP_ERR_INVALID, "stride %zu < 4*width", stride);
    if (!c->uploaded) return fail(c, TP_ERR_STATE, "render before upload");
    if (source == TP_RENDER_STORED && !c->have_colors) return fail(c, TP_ERR_STATE, "render: no colours were uploaded");
    if (int rc = tp_synchronize(c)) return rc;  // settles (and, after an overflow, replays) fused iterations
    if (!c->render_pic) HIP_TRY(c, dev_alloc(&c->render_pic, (size_t)c->W * c->H * 4));
    uint8_t* pic = c->render_pic;
    float2* pts = nullptr;
    hipError_t e = hipMemsetD32Async((hipDeviceptr_t)pic, 0xff000000u, (size_t)c->W * c->H, c->stream);  // opaque black
    if (e == hipSuccess && points) {
        if (c->render_pts_cap < (size_t)c->NP) {
            hipFree(c->render_pts); c->render_pts = nullptr; c->render_pts_cap = 0;
            e = dev_alloc(&c->render_pts, (size_t)c->capP);
            if (e == hipSuccess) c->render_pts_cap = (size_t)c->capP;
        }
        pts = c->render_pts;
        if (e == hipSuccess) e = hipMemcpyAsync(pts, points, sizeof(float) * 2 * (size_t)c->NP, hipMemcpyHostToDevice, c->stream);
    }
    if (e == hipSuccess) {
        tp_launch L = make_launch(c, 0, 0.0f);
        tp_launch_render(L, pts ? pts : c->points, source, pic, c->W, c->stream);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy2D(dst, stride, pic, (size_t)c->W * 4, (size_t)c->W * 4, c->H, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail(c, TP_ERR_HIP, "render: %s", hipGetErrorString(e));
    return TP_OK;
}

int tp_get_stream(tp_context* c, void** s) {
    if (!c || !s) return TP_ERR_INVALID;
    *s = (void*)c->stream;
    return TP_OK;
}

int tp_selftest_walker(tp_context* c, const int64_t* N0, const int32_t* step, const int32_t* d, int n, int32_t* out) {
    api_guard api_lock;
    if (!c || !N0 || !step || !d || !out || n < 0) return TP_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    int64_t* dN = nullptr; int32_t *ds = nullptr, *dd = nullptr, *dout = nullptr;
    hipError_t e = dev_alloc(&dN, n);   // (one exit: the four buffers are freed on every path)
    if (e == hipSuccess) e = dev_alloc(&ds, n);
    if (e == hipSuccess) e = dev_alloc(&dd, n);
    if (e == hipSuccess) e = dev_alloc(&dout, (size_t)n * 32);
    if (e == hipSuccess) e = hipMemcpy(dN, N0, sizeof(int64_t) * n, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(ds, step, sizeof(int32_t) * n, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dd, d, sizeof(int32_t) * n, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        tp_launch_selftest_walker(dN, ds, dd, n, dout, c->stream);
        e = hipStreamSynchronize(c->stream);
    }
    if (e == hipSuccess) e = hipMemcpy(out, dout, sizeof(int32_t) * 32 * (size_t)n, hipMemcpyDeviceToHost);
    hipFree(dN); hipFree(ds); hipFree(dd); hipFree(dout);
    if (e != hipSuccess) return fail(c, TP_ERR_HIP, "selftest_walker: %s", hipGetErrorString(e));
    return TP_OK;
}

int tp_selftest_line(tp_context* c, const int32_t* ends, const int32_t* H, int n, int rows, int32_t* out) {
    api_guard api_lock;
    if (!c || !ends || !H || !out || n < 0 || rows < 1) return TP_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    int4* de = nullptr; int* dh = nullptr; int32_t* dout = nullptr;
    hipError_t e = dev_alloc(&de, n);
    if (e == hipSuccess) e = dev_alloc(&dh, n);
    if (e == hipSuccess) e = dev_alloc(&dout, (size_t)n * (rows + 2));
    if (e == hipSuccess) e = hipMemcpy(de, ends, sizeof(int4) * (size_t)n, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dh, H, sizeof(int) * (size_t)n, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        tp_launch_selftest_line(de, dh, n, rows, dout, c->stream);
        e = hipStreamSynchronize(c->stream);
    }
    if (e == hipSuccess) e = hipMemcpy(out, dout, sizeof(int32_t) * (size_t)n * (rows + 2), hipMemcpyDeviceToHost);
    hipFree(de); hipFree(dh); hipFree(dout);
    if (e != hipSuccess) return fail(c, TP_ERR_HIP, "selftest_line: %s", hipGetErrorString(e));
    return TP_OK;
}

#ifdef TPOSE_DEBUG
// debug flavour only (libtpose_hip_debug.so, tools/acc_timeline.py): the per-block phase timestamps
int tp_debug_dump(tp_context* c, unsigned long long* out, int n) {
    api_guard api_lock;
    tp_launch L = make_launch(c, 0, 0.0f);
    if (!L.dbg) return TP_ERR_STATE;
    hipStreamSynchronize(c->stream);
    hipMemcpy(out, L.dbg, sizeof(unsigned long long) * n, hipMemcpyDeviceToHost);
    return TP_OK;
}
#endif

#ifdef TPOSE_DEBUG
// debug flavour only (tools/persist_timeline.py): [workgroup][grad-iter < 64][8] phase timestamps of the last persistent launch
int tp_debug_dump_persist(tp_context* c, unsigned long long* out, int n) {
    api_guard api_lock;
    if (!g_persist_dbg) return TP_ERR_STATE;
    hipStreamSynchronize(c->stream);
    hipMemcpy(out, g_persist_dbg, sizeof(unsigned long long) * (size_t)n, hipMemcpyDeviceToHost);
    return TP_OK;
}
#endif

int tp_get_info(tp_context* c, int what, int64_t* value) {
    api_guard api_lock;
    if (!c || !value) return TP_ERR_INVALID;
    switch (what) {
        case 0: *value = c->prefix_pitch; return TP_OK;
        case 1: *value = c->lanes_per_line; return TP_OK;
        case 2: *value = (c->plan_generation == c->generation && c->plan.ok) ? c->plan.parts : 0; return TP_OK;
        case 3: *value = (c->plan_generation == c->generation && c->plan.ok) ? c->plan.lds_bytes : 0; return TP_OK;
        case 4: *value = (c->plan_generation == c->generation && c->plan.ok) ? c->plan.lines_total : 0; return TP_OK;
        case 5: *value = c->persist_launches; return TP_OK;
        case 6: *value = c->persist_iters; return TP_OK;
        case 7: *value = c->census; return TP_OK;
        case 8: *value = c->replans; return TP_OK;
        case 9: *value = c->persist_failures; return TP_OK;
        default: return fail(c, TP_ERR_INVALID, "unknown info %d", what);
    }
}

}  // extern "C"
