// tp_replan.hip -- re-planning while a long descent runs: vertices drift, lines grow, and the patches
// of the plan a descent started with go out of balance.  The cut itself is tp_plan.h: pk_build_plan.
#include "tp_context.h"

namespace tpctx {

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
                      w->dp * 0.5f * (float)w->H, w->parts, PK_LDS_LIMIT, w->plan, w->base_every, w->rows_cap, w->speed.empty() ? nullptr : w->speed.data());
        if (w->plan.ok) {
            std::vector<float> rows; std::vector<double> wv; std::vector<int> deg;
            pk_vertex_work(w->NP, w->NT, w->tris.data(), w->points.data(), w->NE, w->edge_uv.data(), w->he_edge.data(), w->H, w->speed.empty() ? nullptr : w->speed.data(), rows, wv, deg);
            w->balance = pk_imbalance(w->plan.owner_v, wv, w->plan.parts);
        }
        lk.lock();
        w->busy = false; w->done = true;
    }
}
void start_replan(tp_context* c, const float* points, float dp, const std::vector<float>& speed_px) {
    if (!c->worker) {
        c->worker.reset(new tp_context::replan_worker());
        c->worker->th = std::thread(replan_worker_main, c->worker.get());
    }
    tp_context::replan_worker* w = c->worker.get();
    std::lock_guard<std::mutex> lk(w->m);
    if (w->busy || w->done) return;   // (one cut at a time: the one under way is from positions nearly as new)
    w->points.assign(points, points + 2 * (size_t)c->NP);
    w->speed = speed_px;
    w->tris = c->h_tris; w->edge_uv = c->h_edge_uv; w->he_edge = c->h_he_edge;
    w->NP = c->NP; w->NT = c->NT; w->NE = c->NE; w->W = c->W; w->H = c->H; w->parts = plan_patches(c);
    w->ratio = c->ratio; w->dp = dp; w->generation = c->generation; w->base_every = c->plan_base_every; w->rows_cap = plan_rows_cap(c);
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
    if (!w->speed.empty()) { c->last_speed_px = w->speed; c->speed_generation = c->generation; }
    c->plan_balance = w->balance;
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
    // how far every vertex moves per grad-iter, in pixels (the kernel's own average over the launch the snapshot follows)
    std::vector<float> speed;
    bool unbalanced = false;
    static const bool no_speed = getenv("TPOSE_NO_SPEED_PLAN") != nullptr;   // (A/B: plans balanced by rows alone, as before round 6)
    if (c->vspeed && c->snap_speed[k] && !no_speed) {
        const float* sp = c->snap_speed[k];
        speed.resize((size_t)c->NP);
        for (size_t v = 0, n = (size_t)c->NP; v < n; v++) {
            const float s = sp[2 * v] * sx + sp[2 * v + 1] * sy;
            speed[v] = s >= 0.0f ? s : 0.0f;   // (NaN: nothing known)
        }
        std::vector<float> rows; std::vector<double> wv; std::vector<int> deg;
        pk_vertex_work(c->NP, c->NT, c->h_tris.data(), q, c->NE, c->h_edge_uv.data(), c->h_he_edge.data(), c->H, speed.data(), rows, wv, deg);
        const double now = pk_imbalance(c->plan.owner_v, wv, c->plan.parts);
        unbalanced = now > PK_REPLAN_BALANCE && now > 1.08 * c->plan_balance;
    }
    if (worst <= PK_REPLAN_PX && !unbalanced) return TP_OK;
    if (worst <= PK_REPLAN_PX) c->replans_balance++;
    if (!more_chunks) { start_replan(c, q, dp, speed); return TP_OK; }
    if (c->worker) { std::lock_guard<std::mutex> lk(c->worker->m); c->worker->superseded = true; }
    bool ok = false;
    if (int rc = build_plan(c, q, dp, c->plan_slot ^ 1, &ok, speed.empty() ? nullptr : speed.data())) return rc;
    if (ok) c->replans++;
    return TP_OK;
}

}  // namespace tpctx
