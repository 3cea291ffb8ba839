// tp_persist_host.hip -- host side of the persistent grad-iter kernel (tp_persist.hip): the status words of a launch and
// the replay of one that gave up, the census of resident workgroups, the plan of the current triangulation, the launches of a
// tp_iterate call, and tp_iterate_until (the reference's frame loop up to its convergence test: software/triangulate/main.cpp:
// 201-210, software/warp/main.cpp:226-231, source/triangulation.hpp:653-674).
#include "tp_context.h"
#include <atomic>

namespace tpctx {

// ---- persistent grad-iter kernel: status, census, plan ------------------------------------------------------------
#ifdef TPOSE_DEBUG  // debug flavour of the library (tools/persist_timeline.py): per-workgroup phase timestamps
static unsigned long long* g_persist_dbg = nullptr;
static const size_t PERSIST_DBG_WORDS = PK_DBG_WBASE + (size_t)512 * PK_DBG_WITERS * 16 * 16;   // (room for 16 waves per workgroup)
unsigned long long* persist_dbg_buffer(int parts, hipStream_t s) {
    if (!g_persist_dbg) { hipMalloc((void**)&g_persist_dbg, PERSIST_DBG_WORDS * 8); }
    hipMemsetAsync(g_persist_dbg, 0, PERSIST_DBG_WORDS * 8, s);
    return parts <= 512 ? g_persist_dbg : nullptr;
}
#endif

// After the stream was synchronised: did a lane of a persistent launch give up waiting?  That happens when the launch's
// workgroups were not all resident together -- another process or another context had a persistent launch of its own on
// the same GPU at that moment (the census only shows that a full grid fits an otherwise idle device).  A launch that
// gives up changes nothing: `points` is only written by the small kernel behind it, which does nothing once the status
// word is raised, and so do all later persistent launches.  So the grad-iters of the launches that did not complete are
// run again here, on the two-kernel path, and the context stops using persistent launches for a while.
int check_persist_status(tp_context* c) {
    if (!c->persist_unchecked || !c->d_status) return TP_OK;
    c->persist_unchecked = false;
    // (every caller has waited for the stream: the mirror is what the last k_persist_finish left)
    const unsigned st[3] = {c->h_status[0], 0u, c->h_status[2]};
    const size_t completed = (size_t)(st[2] - c->done_base);
    c->done_base = st[2];
    if (st[0] == 0u) {
        // (a run of launches that completed ends the streak: give-ups hours apart -- another process briefly on the device -- are each
        // the first of their own, not steps towards the longest wait)
        c->completed_since_give_up += (int64_t)c->journal.size();
        if (c->completed_since_give_up >= 64) c->persist_streak = 0;
        c->journal.clear();
        return TP_OK;
    }
    // (a workgroup that gives up tells the host at once; the others of its launch, and the launches behind it, are still on their way out)
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemset(c->d_status, 0, sizeof(unsigned)));
    HIP_TRY(c, hipMemset(c->d_status + 3, 0, sizeof(unsigned)));   // (the tickets of a launch that finishes itself and gave up on the way)
    c->h_status[0] = 0u;
    // two kernels per grad-iter from now on -- for a while: a collision with somebody else's launch is a transient thing, and the
    // two-kernel path is three times slower.  Persistent launches are tried again 0.2 s later -- 0.8, 3.2, then every 12.8 s while the give-ups
    // come in a row (a device the context really shares); 64 launches that complete start the count afresh.  Never for good.
    c->census = -6;
    c->persist_failures++;
    c->persist_streak++; c->completed_since_give_up = 0;
    c->persist_retry_at = std::chrono::steady_clock::now() + std::chrono::milliseconds(200ll << (2 * (c->persist_streak < 4 ? c->persist_streak - 1 : 3)));
    c->mutations++; c->tail_is_finish = false;  // (what a retrieve returns is about to change)
    // (snapshots taken behind launches that did nothing hold a half-written buffer: no plan is cut from them)
    c->snap_pending[0] = c->snap_pending[1] = false; c->iters_since_snap = 0;
    // (what the launches that gave up left for the next one is of positions the replay below discards, and its age counts grad-iters that never
    // took effect: a hint only -- tl / nc / cut / items stay self-consistent per workgroup -- but nothing is gained by keeping it)
    drop_carry(c);
    if (c->worker) { std::lock_guard<std::mutex> lk(c->worker->m); c->worker->superseded = true; }
    std::vector<tp_context::journal_entry> todo(c->journal.begin() + (completed < c->journal.size() ? completed : c->journal.size()), c->journal.end());
    c->journal.clear();
    // launches that finish themselves write the OTHER position buffer, and the host swaps the two behind each: the positions the first
    // launch that did not complete started from are whole in the buffer it read (later launches did nothing)
    for (auto& e : todo)
        if (e.before) { if (c->points != e.before) std::swap(c->points, c->points_out); break; }
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
    HIP_TRY(c, wait_context(c));
    return check_persist_status(c);
}

// the vertices' positions into the edges' endpoint copies, if persistent launches left them behind (their small finishing kernel
// skips that: 4 us of dependent table reads per call for something only the two-kernel path reads)
int settle_epos(tp_context* c) {
    if (!c->epos_stale) return TP_OK;
    c->epos_stale = false; c->tail_is_finish = false;
    tp_launch_publish_positions(make_launch(c, 0, 0.0f), c->stream);
    HIP_TRY(c, hipGetLastError());
    return TP_OK;
}

// Waiting for the context's stream.  Behind a persistent launch the last thing on the stream is its small finishing kernel, whose LAST
// block writes the count of completed launches into pinned memory: the host spins on that word (a PCIe write away from the kernel's end)
// instead of asking the runtime (several microseconds per answer).  Anything else enqueued since: the runtime's answer.
hipError_t wait_context(tp_context* c) {
    if (c->tail_is_finish && c->persist_unchecked && c->h_status) {
        const unsigned want = c->done_base + (unsigned)c->journal.size();
        volatile unsigned* hs = c->h_status;
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spin = 0;; spin++) {
            if (hs[2] == want || hs[0] != 0u) { std::atomic_thread_fence(std::memory_order_acquire); return hipSuccess; }
            if ((spin & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(400)) break;   // (a long call: block)
        }
    }
    return wait_stream(c->stream);
}

// once per context: launch a full grid of the persistent kernel in census mode -- every workgroup arrives at a counter and
// waits for all the others.  If that times out, workgroups of such a grid are not resident together on this device
// (CU masking, another process) and hand-overs inside a launch would never complete: the context keeps to two kernels.
int take_census(tp_context* c) {
    // (a census that timed out says the device was busy at that moment -- another process's grid -- not that a full grid never fits:
    // once more a second later, and once more four seconds after that; each try that fails costs its 30 ms)
    if ((c->census == -4 || c->census == -5) && c->census_retries < 2 && c->journal.empty() &&
        std::chrono::steady_clock::now() >= c->persist_retry_at) { c->census = 0; c->census_retries++; }
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
    else {
        c->census = -4 - (int)(st[0] != 0u);
        c->persist_retry_at = std::chrono::steady_clock::now() + std::chrono::milliseconds(c->census_retries == 0 ? 1000 : 4000);
    }
    return TP_OK;
}

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
    c->tail_is_finish = false;
    HIP_TRY(c, hipGetLastError());
    c->plan = std::move(np);
    c->plan_slot = slot;
    c->plan_points.assign(points, points + 2 * (size_t)c->NP);
    c->iters_since_cut = 0;
    {   // room for what the launches on this plan hand to each other, under a tag of its own
        int most = 0;
        for (const pk_wg& w : c->plan.wg) most = std::max(most, w.n_lines_all);
        const int cut_cap = (most + 1 + 3) & ~3;
        c->carry_stride = 8 + 3 * cut_cap + 3 * PK_CACHED;   // {header, chunks / slots / first uncached lane-item of every line, every slot's lane-item}
        const size_t words = (size_t)c->plan.parts * (size_t)c->carry_stride;
        if (words > c->cap_carry) {
            HIP_TRY(c, hipStreamSynchronize(c->stream));   // (an earlier launch may still be writing the old one)
            hipFree(c->carry); c->carry = nullptr; c->cap_carry = 0;
            const size_t n = words + words / 4 + 1024;
            HIP_TRY(c, dev_alloc(&c->carry, n));
            c->cap_carry = n;
        }
        // (cleared for every plan: the words of another plan's layout are anything -- a lane-item's chunk count where this layout keeps a
        // tag --, and no tag is 0)
        HIP_TRY(c, hipMemsetAsync(c->carry, 0, words * sizeof(int32_t), c->stream));
        drop_carry(c);
    }
    return TP_OK;
}
void drop_carry(tp_context* c) { c->carry_tag = ++c->carry_seq ? c->carry_seq : ++c->carry_seq; c->carry_written = false; }
int plan_patches(const tp_context* c) { return c->n_bands > 1 ? c->n_bands * c->band_patches : c->num_cus * PK_WG_PER_CU; }

// the rows per lane a cut of this triangulation may start from: what its last plan ended with (a plan that finds no room in LDS for the rows it
// took is cut again with fewer -- twice the planner's time, which at 4096^2 / 12 000 is milliseconds inside a call)
int plan_rows_cap(const tp_context* c) {
    return c->plan.ok && c->plan_generation == c->generation && c->plan.rows_cap > 0 ? c->plan.rows_cap : PK_ROWS_BIG;
}
// cut a plan from `points` and install it in plan buffer `slot`.  c->plan is replaced only when the new plan is usable.
int build_plan(tp_context* c, const float* points, float dp, int slot, bool* ok, const float* speed_px) {
    if (speed_px && speed_px != c->last_speed_px.data()) { c->last_speed_px.assign(speed_px, speed_px + c->NP); c->speed_generation = c->generation; }
    pk_plan np;
    pk_build_plan(c->NP, c->NT, c->h_tris.data(), points, c->NE, c->h_edge_uv.data(), c->h_he_edge.data(),
                  c->W, c->H, c->ratio, dp * 0.5f * (float)c->H, plan_patches(c), PK_LDS_LIMIT, np,
                  c->plan_base_every, plan_rows_cap(c), speed_px);
    if (np.ok) {
        std::vector<float> rows; std::vector<double> wv; std::vector<int> deg;
        pk_vertex_work(c->NP, c->NT, c->h_tris.data(), points, c->NE, c->h_edge_uv.data(), c->h_he_edge.data(), c->H, speed_px, rows, wv, deg);
        c->plan_balance = pk_imbalance(np.owner_v, wv, np.parts);
        double total = 0.0, most = 0.0;
        for (double x : wv) { total += x; most = std::max(most, x); }
        c->plan_heaviest_vertex = total > 0.0 ? most * (double)np.parts / total : 0.0;
    }
    // (a band split runs equal shares of the patches: a plan with fewer patches than asked for -- a tiny mesh -- is not split)
    if (np.ok && c->n_bands > 1 && np.parts != c->n_bands * c->band_patches) { np.ok = false; np.why = "fewer patches than the bands need"; }
    *ok = np.ok;
    if (!np.ok) { if (!c->plan.ok) c->plan = np; return TP_OK; }
    return install_plan(c, np, points, slot);
}

// the plan of the current triangulation (built on first use after an upload); *use = whether tp_iterate may take the
// persistent path
int ensure_plan(tp_context* c, float dp, bool* use, bool base_every) {
    *use = false;
    // (bands keep ONE plan for tp_iterate and tp_iterate_until -- the one that walks every triangle's base lines in every grad-iter:
    // cutting a plan again allocates, and an allocation may wait for a device on which another band is already waiting for this one)
    if (c->n_bands > 1) base_every = true;
    if (c->persist_mode == TP_PERSIST_OFF || !c->px_pitch) return TP_OK;  // (rasters beyond 4096 columns or rows have no pixel-record table)
    if (int rc = take_census(c)) return rc;
    if (c->census == -6 && c->n_bands == 1 && std::chrono::steady_clock::now() >= c->persist_retry_at)
        c->census = 1;   // (the census itself had passed: a launch gave up later, check_persist_status)
    if (c->census != 1) return TP_OK;
    bool recut_same_mesh = false;
    if (c->plan_generation == c->generation && base_every && !c->plan_base_every) {
        // the plan of this triangulation does not walk the base lines in every grad-iter yet: cut it again (from the
        // upload-time positions; a later re-plan follows the mesh) -- nothing in flight reads the plan buffers by then
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->plan_generation = 0;
        recut_same_mesh = true;
    }
    if (c->plan_generation != c->generation) {
        c->plan_generation = c->generation;
        c->plan = pk_plan();
        c->plan_base_every = base_every;
        c->snap_pending[0] = c->snap_pending[1] = false;   // (uploads synchronise the stream: nothing is in flight)
        bool ok = false;
        // (the same mesh cut again for the other kind of plan: what is known about its vertices' speeds -- tp_prepare's probe, a re-plan -- still holds)
        const bool keep_speed = recut_same_mesh && c->speed_generation == c->generation && c->last_speed_px.size() == (size_t)c->NP;
        if (!keep_speed) c->last_speed_px.clear();
        if (int rc = build_plan(c, c->h_points.data(), dp, 0, &ok, keep_speed ? c->last_speed_px.data() : nullptr)) return rc;
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
                for (int k = 0; k < 2; k++) { if (c->snap_speed[k]) hipHostFree(c->snap_speed[k]); c->snap_speed[k] = nullptr; }
                c->snap_cap = 0;
                const size_t n = np + np / 2 + 64;
                for (int k = 0; k < 2; k++) HIP_TRY(c, hipHostMalloc((void**)&c->snap_host[k], n * 2 * sizeof(float), hipHostMallocDefault));
                for (int k = 0; k < 2; k++) HIP_TRY(c, hipHostMalloc((void**)&c->snap_speed[k], n * 2 * sizeof(float), hipHostMallocDefault));
                c->snap_cap = n;
            }
            // (a new triangulation: nothing is known about how its vertices move)
            if (int rc = grow(c, &c->vspeed, &c->cap_vspeed, 2 * np)) return rc;
            HIP_TRY(c, hipMemsetAsync(c->vspeed, 0, 2 * np * sizeof(float), c->stream));
            for (int k = 0; k < 2; k++) if (!c->snap_ev[k]) HIP_TRY(c, hipEventCreateWithFlags(&c->snap_ev[k], hipEventDisableTiming));
        }
    }
    *use = c->plan.ok;
    return TP_OK;
}

// ---- device turns.  A persistent launch wants every compute unit; two of them from two contexts of one process (the two directions of a
// warp, a batch) may be handed out workgroup by workgroup to both at once, and then neither has its grid resident: both wait out their
// time limit, give up and are run again on the two-kernel path (45 ms for a call; seen as one pair in eight of tools/run_batch.py taking
// three times as long).  So while more than one context lives on a device, a persistent launch waits for the one before it, whoever
// enqueued it: an event behind every launch, a hipStreamWaitEvent in front of the next one of another context.  Bands are exempt (they are
// sized to be resident together), and so is the only context of its device (nothing recorded, nothing waited for).  Other processes' launches
// are still what the give-up path is for.
namespace {
struct device_turn { std::mutex m; hipEvent_t last = nullptr; const tp_context* owner = nullptr; int contexts = 0; };
device_turn g_turn[64];
device_turn* turn_of(const tp_context* c) { return c->device >= 0 && c->device < 64 ? &g_turn[c->device] : nullptr; }
}
void join_device(tp_context* c) {
    if (device_turn* T = turn_of(c)) { std::lock_guard<std::mutex> lk(T->m); T->contexts++; c->on_device = true; }
}
void leave_device(tp_context* c) {   // (the caller has waited for the context's stream: its event is complete)
    if (!c->on_device) return;
    c->on_device = false;
    if (device_turn* T = turn_of(c)) {
        std::lock_guard<std::mutex> lk(T->m);
        if (T->contexts > 0) T->contexts--;
        if (T->owner == c) { T->owner = nullptr; T->last = nullptr; }
    }
}

// n grad-iters of the persistent kernel -- the last one writes `tenergy`, `colnum`, `colacc`, `gradient` --, then
// `points_out` -> `points` / `epos`
int enqueue_persistent(tp_context* c, const tp_params& p, float dp, int n, bool rings, bool probe, bool rings_emit) {
    while (n > 0) {
        if (!probe) if (int rc = take_replan(c)) return rc;   // (a plan cut on the side since an earlier call, if it is ready)
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
        A.px = c->px[p.image_slot]; A.px_tiled = c->pxt[p.image_slot]; A.px_pitch = c->px_pitch;
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
            A.final_slot = (unsigned)(c->band_seq++ & 1);
        }
        A.epoch = c->epoch; A.n_iters = k; A.status = c->d_status;
        static const bool no_carry = getenv("TPOSE_NO_CARRY") != nullptr;   // (debugging: every launch cuts for itself)
        if (c->carry && c->carry_stride > 0 && !no_carry) {
            if (dp != c->carry_dp || p.image_slot != c->carry_slot) { drop_carry(c); c->carry_dp = dp; c->carry_slot = p.image_slot; }   // (another dp or image than the launch before)
            A.carry = c->carry; A.carry_stride = c->carry_stride; A.carry_tag = c->carry_tag; A.carry_cut_cap = (c->carry_stride - 8 - 3 * PK_CACHED) / 3;
            if (c->carry_written) c->warm_launches++;
            c->carry_written = true;
        }
        // (rings_emit: a chunk of tp_iterate_until that is the call's last -- its last frame writes the reference's buffers as a tp_iterate call's does)
        A.emit = n == k && !probe && (!rings || (rings_emit && c->n_bands == 1)); A.ten = c->ten; A.cn = c->cn; A.ca_out = c->ca; A.gr = c->gr;
        A.vspeed = banded ? nullptr : c->vspeed;   // (bands keep the plan they cut together)
        if (rings && !banded_rings(c)) { A.ering = c->ering; A.pring = c->pring; }
        if (rings && banded_rings(c)) {
            // (the half of the rings this chunk writes: the next chunk's frame 0 needs nothing from the other bands, so a band that is
            // a chunk ahead would otherwise store into a ring whose last chunk this band's host is still reading -- tp_iterate_until)
            const size_t eo = (size_t)c->ring_half * (PK_RING_FRAMES / 2) * (size_t)c->NT, po = (size_t)c->ring_half * (PK_RING_FRAMES / 2) * (size_t)c->NP;
            A.ering = band_ering(c, c->band) + eo; A.pring = band_pring(c, c->band) + po;
            int n = 0;
            for (int b = 0; b < c->n_bands; b++) if (b != c->band) { A.peer_ering[n] = band_ering(c, b) + eo; A.peer_pring[n] = band_pring(c, b) + po; n++; }
        }
#ifdef TPOSE_DEBUG
        A.dbg = persist_dbg_buffer(c->plan.parts, c->stream);
        { const char* f = getenv("TPOSE_DBG_FIRST"); A.dbg_first = f ? atoi(f) : 0; }
#endif
        // a plain launch over a mesh without unused vertices finishes itself: it writes the positions it ends with into the other position
        // buffer, its last workgroup counts it as completed, and the host swaps the buffers -- no small kernel behind it
        const bool self = !banded && !rings && !c->has_loose && c->points_out && c->cap_points_out >= (size_t)c->NP;
        if (probe && !self) return TP_OK;   // (a probe leaves `points` alone because it writes the OTHER buffer and nobody swaps the two)
        if (self) A.host_status = c->h_status;
        if (c->inject_give_up > 0 && --c->inject_give_up == 0) A.inject_give_up = 1;
        device_turn* T = banded ? nullptr : turn_of(c);
        std::unique_lock<std::mutex> turn;
        bool shared_device = false;
        if (T) {
            turn = std::unique_lock<std::mutex>(T->m);
            shared_device = T->contexts > 1;
            if (shared_device && T->last && T->owner != c) HIP_TRY(c, hipStreamWaitEvent(c->stream, T->last, 0));
        }
        // (a plan with eight LDS rows per lane runs the 24-row instantiation whatever its largest patch takes today: its lines may grow)
        tp_launch_persist(A, grid, !c->plan.wg.empty() && c->plan.wg[0].lds_rows == PK_LDS_ROWS_BIG ? PK_ROWS_BIG : c->plan.rows_max, c->plan.lds_bytes, c->stream);
        if (banded) tp_launch_band_collect(make_launch(c, p.image_slot, dp), A, c->points_out, c->stream);
        if (!self) tp_launch_persist_finish(make_launch(c, p.image_slot, dp), c->points_out, c->d_status, c->h_status, 0, c->stream);
        c->epos_stale = true; c->tail_is_finish = true;
        c->journal.push_back({p, rings || probe ? 0 : k, self && !probe ? c->points : nullptr});   // (a chunk of tp_iterate_until is checked by its caller: nothing to replay)
        if (self && !probe) std::swap(c->points, c->points_out);
        // A marker behind every launch (round 5): the runtime raises a completion signal for the LAST command of a stream only when somebody
        // asks -- a caller's hipStreamSynchronize / hipDeviceSynchronize behind a bare kernel submits a barrier packet of its own and waits
        // for the round trip (13.6 us behind a 20-step launch; 6.4 with the marker already queued: tools/host_call_cost.py).  0.6 us of
        // host time per launch.  It is also what another context's launch waits for when the device is shared (device turns, above).
        if (!c->ev_turn) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_turn, hipEventDisableTiming));
        HIP_TRY(c, hipEventRecord(c->ev_turn, c->stream));
        if (shared_device) { T->last = c->ev_turn; T->owner = c; }
        if (turn.owns_lock()) turn.unlock();
        HIP_TRY(c, hipGetLastError());
        c->epoch += (uint32_t)k;
        c->persist_unchecked = true;
        n -= k;
        if (probe) { drop_carry(c); continue; }   // (what the probe left is of positions nobody keeps)
        c->persist_launches++; c->persist_iters += k;
        c->iters_since_snap += k; c->iters_since_cut += k;
        if (c->iters_since_snap >= PK_CHUNK / 2) {   // the positions after this chunk, for a later maybe_replan
            const int sl = c->snap_next;
            c->tail_is_finish = false;
            HIP_TRY(c, hipMemcpyAsync(c->snap_host[sl], c->points, sizeof(float) * 2 * (size_t)c->NP, hipMemcpyDeviceToHost, c->stream));
            if (c->vspeed && c->snap_speed[sl]) HIP_TRY(c, hipMemcpyAsync(c->snap_speed[sl], c->vspeed, sizeof(float) * 2 * (size_t)c->NP, hipMemcpyDeviceToHost, c->stream));
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

// tp_prepare, behind the first plan of a triangulation: how fast do its vertices move?  PK_PROBE_ITERS grad-iters of the persistent kernel that
// nobody keeps -- the launch writes the positions it ends with into the context's OTHER position buffer and the host does not swap the two, no
// buffer of the reference is written -- tell the planner what the first launches of a descent cannot know yet: on a photograph a tenth of
// the patches hold the vertices that jump a pixel or more per grad-iter, walk five times as long as the others, and everybody waits for
// them (profiles/r06_meninas_timeline_first_launch.json: period 13.4 us, median patch 4.8).  The plan is cut again with the speeds measured.
int probe_speeds(tp_context* c, const tp_params& p, float dp) {
    static const bool off = getenv("TPOSE_NO_PROBE") != nullptr || getenv("TPOSE_NO_SPEED_PLAN") != nullptr;
    if (off || !c->plan.ok || c->n_bands > 1 || !c->vspeed || c->probed_generation == c->generation) return TP_OK;
    c->probed_generation = c->generation;
    if (int rc = enqueue_persistent(c, p, dp, PK_PROBE_ITERS, false, true)) return rc;
    HIP_TRY(c, wait_context(c));
    const int64_t before = c->persist_failures;
    if (int rc = check_persist_status(c)) return rc;
    if (c->persist_failures != before) return TP_OK;   // (the probe gave up: nothing was measured)
    std::vector<float> sp(2 * (size_t)c->NP), speed((size_t)c->NP);
    HIP_TRY(c, hipMemcpy(sp.data(), c->vspeed, sp.size() * sizeof(float), hipMemcpyDeviceToHost));
    const float sx = 0.5f * (float)c->W / c->ratio, sy = 0.5f * (float)c->H;
    for (size_t v = 0; v < speed.size(); v++) { const float q = sp[2 * v] * sx + sp[2 * v + 1] * sy; speed[v] = q >= 0.0f ? q : 0.0f; }
    std::vector<float> rows; std::vector<double> wv; std::vector<int> deg;
    pk_vertex_work(c->NP, c->NT, c->h_tris.data(), c->h_points.data(), c->NE, c->h_edge_uv.data(), c->h_he_edge.data(), c->H, speed.data(), rows, wv, deg);
    if (pk_imbalance(c->plan.owner_v, wv, c->plan.parts) <= PK_REPLAN_BALANCE) return TP_OK;   // (balanced as it is)
    bool ok = false;
    if (int rc = build_plan(c, c->h_points.data(), dp, c->plan_slot ^ 1, &ok, speed.data())) return rc;
    if (ok) c->replans_balance++;
    return TP_OK;
}

}  // namespace tpctx

using namespace tpctx;

extern "C" {

int tp_iterate_until(tp_context* c, const tp_params* p, int max_frames, double threshold, float* toterr, int* frames, float* relerr) {
    api_guard api_lock;
    if (!c) return TP_ERR_INVALID;
    if (int rc = validate_params(c, p, max_frames)) return rc;
    if (!toterr || !frames) return fail(c, TP_ERR_INVALID, "iterate_until: toterr / frames is NULL");
    *frames = 0;
    if (relerr) *relerr = 0.0f;
    if (max_frames == 0) return TP_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    c->mutations++; c->tail_is_finish = false;
    if (int rc = settle_persistent(c)) return rc;
    const float dp = resolve_dp(c, p->flavour, p->dp);
    const int NT = c->NT;
    float tot = *toterr, rel = 0.0f;
    // geterr (source/triangulation.hpp:653-674) on the base energies of one frame: float32, ascending t
    auto frame_test = [&](float newerr) {
        rel = (tot - newerr) / tot;
        tot = newerr;
        return (double)std::fabs(rel);   // (the reference compares the float with a double literal)
    };
    auto frame_err = [&](const int32_t* terr) {
        float newerr = 0.0f;
        for (int i = 0; i < NT; i++) { float err = 0.0f; err += (float)terr[i]; newerr += err; }
        return frame_test(newerr);
    };
    // the same sums for the frames of a chunk, eight frames side by side: a frame's sum is ONE chain of 3000 dependent float additions
    // (3 us of host time per frame, in series with the device), but the chains of different frames do not depend on each other -- only the
    // test that follows does, through the running total.  Each chain adds in the reference's order.
    std::vector<float> sums;
    auto chunk_sums = [&](const int32_t* terr, int C) {
        sums.assign((size_t)C, 0.0f);
        for (int j0 = 0; j0 < C; j0 += 8) {
            float acc[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
            const int32_t* f[8];
            for (int u = 0; u < 8; u++) f[u] = terr + (size_t)(j0 + u < C ? j0 + u : j0) * NT;
            for (int i = 0; i < NT; i++)
                for (int u = 0; u < 8; u++) acc[u] += (float)f[u][i];
            for (int u = 0; u < 8 && j0 + u < C; u++) sums[(size_t)(j0 + u)] = acc[u];
        }
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
            if (int rc = settle_epos(c)) return rc;
            enqueue_iter(c, *p, dp);
            HIP_TRY(c, hipGetLastError());
            HIP_TRY(c, hipMemcpyAsync(c->ering_host, c->ten, sizeof(int32_t) * (size_t)NT, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, wait_stream(c->stream));
            done++;
            converged = frame_err(c->ering_host) < threshold;
            continue;
        }
        // a chunk of frames inside one persistent launch: every frame leaves its base energies and its starting positions
        const bool shared = banded_rings(c);   // (band split: the rings are in the bands' mailboxes, every band writes into all of them)
        // (... in halves of PK_RING_FRAMES / 2 frames used in turn, chunk after chunk, over the life of the attachment: a band cannot be
        // two chunks ahead of another -- frames beyond the first need the others' positions of the same launch -- so a half is never
        // written while a host still reads the chunk before last from it)
        const int cap_frames = shared ? PK_RING_FRAMES / 2 : 256;
        const int C = left < chunk ? (left < cap_frames ? left : cap_frames) : (chunk < cap_frames ? chunk : cap_frames);
        if (shared) c->ring_half = (int)(c->ring_seq++ & 1);
        if (!shared) {
            if (int rc = grow(c, &c->ering, &c->cap_ering, (size_t)C * NT)) return rc;
            if (int rc = grow(c, &c->pring, &c->cap_pring, (size_t)C * c->NP)) return rc;
        }
        const int32_t* ering = shared ? band_ering(c, c->band) + (size_t)c->ring_half * (PK_RING_FRAMES / 2) * (size_t)NT : c->ering;
        const float2* pring = shared ? band_pring(c, c->band) + (size_t)c->ring_half * (PK_RING_FRAMES / 2) * (size_t)c->NP : c->pring;
        if (int rc = host_ring((size_t)256 * NT)) return rc;   // (for the longest chunk at once: freeing pinned memory waits for the device)
        // (the chunk that ends the call: its last frame leaves the buffers itself, so that a call that runs out of frames -- or converges in
        // its very last one -- runs no frame twice; 25 us of a 20-frame call)
        const bool emits = !shared && C == left;
        if (int rc = enqueue_persistent(c, *p, dp, C, true, false, emits)) return rc;
        // the frames' sums: on the device for the long chunks of a context on its own (one wave per frame, written into the pinned buffer:
        // 256 frames of 3000 energies are 3 MB across the link and 0.1 ms of host additions otherwise); short chunks and a band split's
        // rings (the bands' shared mailboxes) are read back and summed on the host -- a launch more would cost a short chunk more than it saves
        const bool device_sums = !shared && C >= 64;
        if (device_sums) tp_launch_frame_sums(ering, C, NT, (float*)c->ering_host, c->stream);
        else HIP_TRY(c, hipMemcpyAsync(c->ering_host, ering, sizeof(int32_t) * (size_t)C * NT, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, wait_stream(c->stream));
        {
            const int64_t fails = c->persist_failures;
            if (int rc = check_persist_status(c)) return rc;
            if (c->persist_failures != fails) { use = false; continue; }   // the chunk gave up (nothing changed): frame by frame from here
        }
        if (!device_sums) chunk_sums(c->ering_host, C);
        else sums.assign((const float*)c->ering_host, (const float*)c->ering_host + C);
        int j = 0;
        for (; j < C; j++) {
            done++;
            if (frame_test(sums[(size_t)j]) < threshold) { converged = true; break; }
        }
        if (converged || done >= max_frames) {
            // back to the start of the last frame that counts, and that frame once more on the two-kernel path: it writes the
            // buffers the reference reads back (`tenergy`, `colnum`, `colacc`, `gradient`) and takes the step
            const int last = converged ? j : C - 1;
            if (!(emits && last == C - 1)) {   // (the chunk's own last frame left everything in place)
                tp_launch_persist_finish(make_launch(c, p->image_slot, dp), pring + (size_t)last * c->NP, nullptr, nullptr, 1, c->stream);
                c->epos_stale = false; c->tail_is_finish = false;
                enqueue_iter(c, *p, dp);
                HIP_TRY(c, hipGetLastError());
            }
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

int tp_iterate_frames(tp_context* c, const tp_params* p, int max_frames, tp_frame_fn fn, void* user, int* frames) {
    api_guard api_lock;
    if (!c) return TP_ERR_INVALID;
    if (int rc = validate_params(c, p, max_frames)) return rc;
    if (!fn || !frames) return fail(c, TP_ERR_INVALID, "iterate_frames: fn / frames is NULL");
    *frames = 0;
    if (max_frames == 0) return TP_OK;
    if (c->n_bands > 1) return fail(c, TP_ERR_STATE, "iterate_frames: not available to the bands of a split descent");
    HIP_TRY(c, hipSetDevice(c->device));
    c->mutations++; c->tail_is_finish = false;
    if (int rc = settle_persistent(c)) return rc;
    const float dp = resolve_dp(c, p->flavour, p->dp);
    const int NT = c->NT;
    const size_t NP = (size_t)c->NP;
    auto host_rings = [&](size_t frames_cap) -> int {
        const size_t ints = frames_cap * (size_t)NT, pts = (frames_cap + 1) * NP;
        if (ints > c->cap_ering_host || !c->ering_host) {
            if (c->ering_host) hipHostFree(c->ering_host);
            c->ering_host = nullptr; c->cap_ering_host = 0;
            HIP_TRY(c, hipHostMalloc((void**)&c->ering_host, ints * sizeof(int32_t), hipHostMallocDefault));
            c->cap_ering_host = ints;
        }
        if (pts > c->cap_pring_host || !c->pring_host) {
            if (c->pring_host) hipHostFree(c->pring_host);
            c->pring_host = nullptr; c->cap_pring_host = 0;
            HIP_TRY(c, hipHostMalloc((void**)&c->pring_host, pts * sizeof(float2), hipHostMallocDefault));
            c->cap_pring_host = pts;
        }
        return TP_OK;
    };
    bool use = false;
    if (max_frames >= PK_MIN_ITERS) { if (int rc = ensure_plan(c, dp, &use, true)) return rc; }
    // vertices no triangle uses (the schedule's prune leaves some): the persistent kernel never writes their ring entries -- they are clamped
    // once by whatever ends the launch (k_persist_finish; shift.cs:25-43 runs for every i >= 4) and stand still from then on
    std::vector<int> loose;
    if (c->has_loose) {
        std::vector<char> used(NP, 0);
        for (int t = 0; t < NT; t++) for (int s = 0; s < 3; s++) used[(size_t)c->h_tris[4 * (size_t)t + s]] = 1;
        for (size_t v = 0; v < NP; v++) if (!used[v]) loose.push_back((int)v);
    }
    int done = 0;
    bool stopped = false;
    while (done < max_frames && !stopped) {
        const int left = max_frames - done;
        if (!use || left < PK_MIN_ITERS) {
            // frame by frame on the two-kernel path: the frame, then its base energies and positions come back
            if (int rc = host_rings(1)) return rc;
            if (int rc = settle_epos(c)) return rc;
            enqueue_iter(c, *p, dp);
            HIP_TRY(c, hipGetLastError());
            HIP_TRY(c, hipMemcpyAsync(c->ering_host, c->ten, sizeof(int32_t) * (size_t)NT, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipMemcpyAsync(c->pring_host, c->points, sizeof(float2) * NP, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, wait_stream(c->stream));
            const int verdict = fn(user, done, c->ering_host, reinterpret_cast<const float*>(c->pring_host));
            done++;
            stopped = verdict != TP_FRAME_GO_ON;   // (the frame has just run on the two-kernel path: everything stands as it left it)
            continue;
        }
        // a chunk of frames inside one persistent launch: every frame leaves its base energies and its starting positions
        const int C = left < 256 ? left : 256;
        if (int rc = grow(c, &c->ering, &c->cap_ering, (size_t)C * NT)) return rc;
        if (int rc = grow(c, &c->pring, &c->cap_pring, (size_t)C * NP)) return rc;
        if (int rc = host_rings(256)) return rc;   // (for the longest chunk at once: freeing pinned memory waits for the device)
        if (int rc = enqueue_persistent(c, *p, dp, C, true)) return rc;
        HIP_TRY(c, hipMemcpyAsync(c->ering_host, c->ering, sizeof(int32_t) * (size_t)C * NT, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->pring_host, c->pring, sizeof(float2) * (size_t)C * NP, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->pring_host + (size_t)C * NP, c->points, sizeof(float2) * NP, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, wait_stream(c->stream));
        {
            const int64_t fails = c->persist_failures;
            if (int rc = check_persist_status(c)) return rc;
            if (c->persist_failures != fails) { use = false; continue; }   // the chunk gave up (nothing changed): frame by frame from here
        }
        for (int v : loose)
            for (int j = 1; j < C; j++) c->pring_host[(size_t)j * NP + (size_t)v] = c->pring_host[(size_t)C * NP + (size_t)v];
        for (int j = 0; j < C; j++) {
            // (frame j started from pring[j] and ended with pring[j + 1] -- the chunk's last one with the positions the launch ended with)
            const int verdict = fn(user, done, c->ering_host + (size_t)j * NT, reinterpret_cast<const float*>(c->pring_host + (size_t)(j + 1) * NP));
            done++;
            if (verdict == TP_FRAME_GO_ON) continue;
            stopped = true;
            if (verdict == TP_FRAME_STOP_REPLAY) {
                tp_launch_persist_finish(make_launch(c, p->image_slot, dp), c->pring + (size_t)j * NP, nullptr, nullptr, 1, c->stream);
                c->epos_stale = false; c->tail_is_finish = false;
                enqueue_iter(c, *p, dp);
            } else if (j + 1 < C) {
                tp_launch_persist_finish(make_launch(c, p->image_slot, dp), c->pring + (size_t)(j + 1) * NP, nullptr, nullptr, 1, c->stream);
                c->epos_stale = false; c->tail_is_finish = false;
            }
            HIP_TRY(c, hipGetLastError());
            break;
        }
    }
    c->acc_slot = p->image_slot; c->last_flavour = p->flavour;
    c->accumulated = c->energized = false;
    *frames = done;
    return TP_OK;
}

#ifdef PK_DBG_BOUNDS
int tp_debug_persist_faults(tp_context* c, unsigned long long* out) {
    api_guard api_lock;
    hipStreamSynchronize(c->stream);
    return tp_persist_debug_faults(out);
}
#endif

#ifdef PK_DBG_STALE
// counting flavour only (tools/stale_counts.py): [512][4] what k_persist counted since the last reset; what the planner weighed the plan's patches with
int tp_debug_persist_counts(tp_context* c, unsigned long long* out, int reset) {
    api_guard api_lock;
    hipStreamSynchronize(c->stream);
    return tp_persist_debug_counts(out, reset);
}
int tp_debug_persist_vcounts(tp_context* c, unsigned long long* out, int reset) {   // [NP][2]: rows fetched again, rows walked
    api_guard api_lock;
    hipStreamSynchronize(c->stream);
    return tp_persist_debug_vcounts(out, c->NP, reset);
}
int tp_debug_plan_owner(tp_context* c, int32_t* out, int n) {   // [NP]: the patch that owns the vertex in the current plan (-1: none)
    api_guard api_lock;
    if (!c->plan.ok || n < c->NP || (int)c->plan.owner_v.size() < c->NP) return TP_ERR_STATE;
    for (int v = 0; v < c->NP; v++) out[v] = c->plan.owner_v[v];
    return TP_OK;
}
int tp_debug_plan_weights(tp_context* c, float* out, int n) {   // [patch][4]: work, rows, hot, rows per lane
    api_guard api_lock;
    const pk_plan& P = c->plan;
    if (!P.ok || (int)P.patch_work.size() != P.parts || n < 4 * P.parts) return TP_ERR_STATE;
    for (int p = 0; p < P.parts; p++) { out[4 * p] = P.patch_work[p]; out[4 * p + 1] = P.patch_rows[p]; out[4 * p + 2] = (float)P.wg[p].hot; out[4 * p + 3] = (float)P.wg[p].rows; }
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

}  // extern "C"
