// tp_context.hip -- the C ABI of include/tpose_hip.h: context, device memory, uploads, the piecewise API, the two-kernel
// path and its hipGraphs, tp_iterate (tp_context.h lists the other translation units of the host side).
#include "tp_context.h"

std::shared_mutex g_api_mutex;
thread_local int g_api_depth = 0;

namespace {
thread_local std::string g_create_error;
}

namespace tpctx {

int fail(tp_context* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->error = buf; else g_create_error = buf;
    return code;
}


void drop_graphs(tp_context* c) {
    for (auto& g : c->graphs)
        if (g.exec) hipGraphExecDestroy(g.exec);
    c->graphs.clear();
}

void free_triangulation(tp_context* c) {
    hipFree(c->vref); hipFree(c->vvar); c->vref = nullptr; c->vvar = nullptr;
    hipFree(c->points); hipFree(c->points_out); c->points_out = nullptr; c->cap_points_out = 0; hipFree(c->tris); hipFree(c->colors); hipFree(c->vtx_off); hipFree(c->vtx_adj);
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


// enqueue one grad-iter on the context stream (no sync)
int enqueue_iter(tp_context* c, const tp_params& p, float dp, bool mirror) {
    tp_launch L = make_launch(c, p.image_slot, dp);
    if (mirror) { if (int rc = frame_mirror_into(c, L)) return rc; }
    tp_launch_lines(L, c->stream);                       // vertex stage + the nine line sums of every edge
    tp_launch_update(L, p.flavour, p.rate, c->stream);  // variants + gradient + shift
    return TP_OK;
}

}  // namespace tpctx

using namespace tpctx;

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
    join_device(c);
    *out = c;
    return TP_OK;
}

int tp_destroy(tp_context* c) {
    api_guard api_lock;
    if (!c) return TP_OK;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    leave_device(c);
    if (c->ev_turn) hipEventDestroy(c->ev_turn);
    if (c->ev_tail) hipEventDestroy(c->ev_tail);
    stop_replan_worker(c);
    drop_graphs(c);
    free_triangulation(c);
    hipFree(c->img[0]); hipFree(c->img[1]); hipFree(c->prefix[0]); hipFree(c->prefix[1]); hipFree(c->px[0]); hipFree(c->px[1]); hipFree(c->pxt[0]); hipFree(c->pxt[1]);
    hipFree(c->render_pic); hipFree(c->render_pts);
    hipFree(c->d_wg); hipFree(c->d_pool); hipFree(c->posbox); hipFree(c->points_out); hipFree(c->d_status); hipFree(c->carry); hipFree(c->vspeed);
    if (c->h_status) hipHostFree(c->h_status);
    if (c->frame_mirror) hipHostFree(c->frame_mirror);
    hipFree(c->ering); hipFree(c->pring);
    if (c->ering_host) hipHostFree(c->ering_host);
    if (c->pring_host) hipHostFree(c->pring_host);
    for (int k = 0; k < 2; k++) {
        hipFree(c->plan_dev[k].wg); hipFree(c->plan_dev[k].pool);
        if (c->plan_dev[k].stage) hipHostFree(c->plan_dev[k].stage);
        if (c->snap_host[k]) hipHostFree(c->snap_host[k]);
        if (c->snap_speed[k]) hipHostFree(c->snap_speed[k]);
        if (c->snap_ev[k]) hipEventDestroy(c->snap_ev[k]);
    }
    if (c->eval_host) hipHostFree(c->eval_host);
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
    c->mutations++; c->tail_is_finish = false;   // (a retrieve no longer finds what the frame mirror holds)
    if (!(ratio > 0.0f)) return fail(c, TP_ERR_INVALID, "RATIO must be positive");
    // (a persistent launch that gave up is replayed at the next synchronisation -- with the RATIO its grad-iters were called with)
    if (c->persist_unchecked) { HIP_TRY(c, hipSetDevice(c->device)); if (int rc = settle_persistent(c)) return rc; }
    if (ratio != c->ratio) { c->ratio = ratio; c->generation++; c->accumulated = c->energized = false; }
    return TP_OK;
}

int tp_set_dp(tp_context* c, float dp) {
    api_guard api_lock;
    if (!c) return TP_ERR_INVALID;
    c->mutations++; c->tail_is_finish = false;   // (a retrieve no longer finds what the frame mirror holds)
    c->dp_override = dp;  // piecewise calls only; tp_iterate takes dp from its params (part of the graph key)
    drop_carry(c);        // (the next persistent launch cuts its lines itself)
    c->accumulated = c->energized = false;
    return TP_OK;
}

int tp_set_option(tp_context* c, int option, int64_t value) {
    api_guard api_lock;
    if (!c) return TP_ERR_INVALID;
    c->mutations++; c->tail_is_finish = false;   // (a retrieve no longer finds what the frame mirror holds)
    switch (option) {
        case TP_OPT_PERSISTENT:
            if (value != TP_PERSIST_OFF && value != TP_PERSIST_AUTO) return fail(c, TP_ERR_INVALID, "TP_OPT_PERSISTENT: bad value %lld", (long long)value);
            c->persist_mode = (int)value;
            return TP_OK;
        case TP_OPT_INJECT_GIVE_UP:
            if (value < 0 || value > 1000000) return fail(c, TP_ERR_INVALID, "TP_OPT_INJECT_GIVE_UP: bad value %lld", (long long)value);
            // (a fault injection for tests: a client cannot send a production context to the slow path with it)
            if (value > 0 && !getenv("TPOSE_ALLOW_FAULT_INJECTION")) return fail(c, TP_ERR_INVALID, "TP_OPT_INJECT_GIVE_UP needs TPOSE_ALLOW_FAULT_INJECTION in the environment");
            c->inject_give_up = (int)value;
            return TP_OK;
        default: return fail(c, TP_ERR_INVALID, "unknown option %d", option);
    }
}
int tp_get_ratio(const tp_context* c, float* ratio) {
    if (!c || !ratio) return TP_ERR_INVALID;
    *ratio = c->ratio;
    return TP_OK;
}

static int set_image_common(tp_context* c, int slot, const void* src, size_t stride, hipMemcpyKind kind) {
    if (!c) return TP_ERR_INVALID;
    c->mutations++; c->tail_is_finish = false;
    if (slot != TP_IMAGE_A && slot != TP_IMAGE_B) return fail(c, TP_ERR_INVALID, "bad image slot %d", slot);
    if (!src) return fail(c, TP_ERR_INVALID, "image pointer is NULL");
    if (stride < (size_t)c->W * 4) return fail(c, TP_ERR_INVALID, "stride %zu < 4*width", stride);
    HIP_TRY(c, hipSetDevice(c->device));
    // (a persistent launch that gave up is replayed at the next synchronisation -- over the tables its grad-iters were called with)
    if (int rc = settle_persistent(c)) return rc;
    if (!c->img[slot]) HIP_TRY(c, dev_alloc(&c->img[slot], (size_t)c->W * c->H * 4));
    HIP_TRY(c, hipMemcpy2DAsync(c->img[slot], (size_t)c->W * 4, src, stride, (size_t)c->W * 4, c->H, kind, c->stream));
    // row prefix table of this image (32-byte records per four pixels: 8 bytes per pixel): what k_lines reads, grad-iter after grad-iter
    if (!c->prefix[slot]) HIP_TRY(c, dev_alloc(&c->prefix[slot], (size_t)c->H * c->prefix_pitch * 2));
    tp_launch_prefix_table(c->img[slot], c->W * 4, c->W, c->H, c->prefix_pitch, c->prefix[slot], c->stream);
    if (c->px_pitch) {  // pixel records of the same sums: what the persistent grad-iter kernel reads
        if (!c->px[slot]) HIP_TRY(c, dev_alloc(&c->px[slot], (size_t)c->H * c->px_pitch));
        if (!c->pxt[slot]) HIP_TRY(c, dev_alloc(&c->pxt[slot], (size_t)tp_px_tiled_rows((uint32_t)c->H) * c->px_pitch));
        tp_launch_px_table(c->img[slot], c->W * 4, c->W, c->H, c->px_pitch, c->px[slot], c->pxt[slot], c->stream);
    }
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->have_img[slot] = true;
    c->accumulated = c->energized = false;
    drop_carry(c);   // (the next persistent launch cuts its lines itself)
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
    c->mutations++; c->tail_is_finish = false;   // (a retrieve no longer finds what the frame mirror holds)
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
        HIP_TRY(c, dev_alloc(&c->points_out, capP)); c->cap_points_out = (size_t)capP;   // (the two are swapped: the same size)
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
    c->has_loose = false;
    for (int v = 0; v < NP; v++) { if (off[v + 1] == 0) c->has_loose = true; off[v + 1] += off[v]; }
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
    c->epos_stale = false;   // (k_vertex_refs' launch filed every position with its edges)
    return TP_OK;
}

int tp_accumulate(tp_context* c, int flavour, int slot) {
    api_guard api_lock;
    if (!c) return TP_ERR_INVALID;
    c->mutations++; c->tail_is_finish = false;   // (a retrieve no longer finds what the frame mirror holds)
    if (flavour != TP_TRIANGULATE && flavour != TP_WARP) return fail(c, TP_ERR_INVALID, "bad flavour %d", flavour);
    if (!c->uploaded) return fail(c, TP_ERR_STATE, "accumulate before upload");
    if (int rc = check_slot(c, slot)) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    if (int rc = settle_persistent(c)) return rc;
    if (int rc = settle_epos(c)) return rc;
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
    c->mutations++; c->tail_is_finish = false;   // (a retrieve no longer finds what the frame mirror holds)
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
    c->mutations++; c->tail_is_finish = false;   // (a retrieve no longer finds what the frame mirror holds)
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

}  // extern "C"

namespace tpctx {
int validate_params(tp_context* c, const tp_params* p, int n_iters) {
    if (!p) return fail(c, TP_ERR_INVALID, "params is NULL");
    if (n_iters < 0) return fail(c, TP_ERR_INVALID, "n_iters < 0");
    if (p->flavour != TP_TRIANGULATE && p->flavour != TP_WARP) return fail(c, TP_ERR_INVALID, "bad flavour %d", p->flavour);
    if (!c->uploaded) return fail(c, TP_ERR_STATE, "iterate before upload");
    return check_slot(c, p->image_slot);
}

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
        if (e.generation == c->generation && e.points == c->points && e.iters == CHUNK && memcmp(&e.params, p, sizeof *p) == 0) { *out = &e; return TP_OK; }
    graph_entry e;
    if (int rc = capture_graph(c, [&] { for (int k = 0; k < CHUNK; k++) enqueue_iter(c, *p, dp); }, &e.exec)) return rc;
    e.params = *p; e.iters = CHUNK; e.generation = c->generation; e.points = c->points;
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
        if (int rc = settle_epos(c)) return rc;
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
    if (left <= 0) return TP_OK;
    if (int rc = settle_epos(c)) return rc;
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
}  // namespace tpctx

extern "C" {

int tp_iterate(tp_context* c, const tp_params* p, int n_iters) {
    api_guard api_lock;
    if (!c) return TP_ERR_INVALID;
    if (int rc = validate_params(c, p, n_iters)) return rc;
    if (n_iters == 0) return TP_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    c->mutations++; c->tail_is_finish = false;
    const int rc = enqueue_iters(c, p, n_iters);
    // A marker behind a single frame (the schedules run frame by frame and read every one back).  The runtime raises a completion signal
    // behind a bare kernel only when asked: a hipStreamQuery / hipStreamSynchronize behind one submits a barrier packet of its own and
    // waits for ITS round trip.  A marker queued right behind the frame's kernels is processed the moment they end, and the read-back polls
    // it (config 2: 0.39 -> 0.30 s of frames).  Queued only just before the wait it buys nothing (measured: the other waits keep
    // hipStreamQuery).
    if (rc == TP_OK && n_iters == 1) {
        if (!c->ev_tail) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_tail, hipEventDisableTiming));
        HIP_TRY(c, hipEventRecord(c->ev_tail, c->stream));
        c->tail_mark = c->mutations;
    }
    return rc;
}

int tp_prepare(tp_context* c, const tp_params* p) {
    api_guard api_lock;
    if (!c) return TP_ERR_INVALID;
    if (int rc = validate_params(c, p, 0)) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    bool use = false;
    if (int rc = ensure_plan(c, resolve_dp(c, p->flavour, p->dp), &use)) return rc;
    if (use) return probe_speeds(c, *p, resolve_dp(c, p->flavour, p->dp));
    graph_entry* g = nullptr;
    return chunk_graph(c, p, resolve_dp(c, p->flavour, p->dp), &g);
}
int tp_profile_iterate(tp_context* c, const tp_params* p, int n_iters, double* accumulate_us) {
    api_guard api_lock;
    if (!c || !accumulate_us) return TP_ERR_INVALID;
    c->mutations++; c->tail_is_finish = false;   // (a retrieve no longer finds what the frame mirror holds)
    if (int rc = validate_params(c, p, n_iters)) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    if (int rc = settle_persistent(c)) return rc;
    if (int rc = settle_epos(c)) return rc;
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
    c->mutations++; c->tail_is_finish = false;   // (a retrieve no longer finds what the frame mirror holds)
    if (int rc = validate_params(c, p, launches)) return rc;
    if (launches < 1) return fail(c, TP_ERR_INVALID, "launches < 1");
    if (int rc = tp_synchronize(c)) return rc;
    if (int rc = settle_epos(c)) return rc;
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
    HIP_TRY(c, wait_context(c));
    return check_persist_status(c);
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
        case 10: *value = c->box_finegrained ? 1 : 0; return TP_OK;
        case 11: *value = c->warm_launches; return TP_OK;
        case 12: {   // milliseconds until persistent launches are tried again after a give-up (0: they are in use, or were never possible here)
            const auto now = std::chrono::steady_clock::now();
            *value = c->census == -6 && c->persist_retry_at > now ? (int64_t)std::chrono::duration_cast<std::chrono::milliseconds>(c->persist_retry_at - now).count() + 1 : 0;
            return TP_OK;
        }
        case 13: *value = (c->plan_generation == c->generation && c->plan.ok) ? c->plan.rows_max : 0; return TP_OK;
        case 14: *value = c->replans_balance; return TP_OK;   // plans cut again because the patches were out of balance under the vertices' speeds
        case 15: *value = (int64_t)(c->plan_balance * 1000.0); return TP_OK;
        case 16: *value = (int64_t)(c->plan_heaviest_vertex * 1000.0); return TP_OK;   // heaviest vertex / mean patch, x 1000 (plans cut on the calling thread)   // heaviest patch / mean patch of the current plan, x 1000
        default: return fail(c, TP_ERR_INVALID, "unknown info %d", what);
    }
}

}  // extern "C"
