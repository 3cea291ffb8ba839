// tp_readback.hip -- Buffer::retrieve (software/triangulate/main.cpp:201-204, 221; software/warp/main.cpp:226-229): read-backs
// with one wait per batch, the pinned frame mirror behind single frames, and the flat-shaded picture (tp_render).
#include "tp_context.h"

namespace tpctx {

// the frame mirror of this triangulation (pinned; grown when the triangulation outgrows it) into a launch
int frame_mirror_into(tp_context* c, tp_launch& L) {
    // (NT + 64 entries: what the schedules' shortcuts look at.  Mirroring all 13 NT entries for callers that read whole buffers back, as the
    // reference's loop does, was tried in round 6: the frame's kernels then store 312 KB more across the link four bytes at a time, and the
    // frame went from 60 to 95 us -- the copy kernel's 16-byte stores behind the frame are the faster way)
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

}  // namespace tpctx

using namespace tpctx;

extern "C" {

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
        if (c->ev_tail && c->tail_mark == c->mutations) HIP_TRY(c, wait_event(c->ev_tail));   // (the marker behind the frame: tp_iterate)
        else HIP_TRY(c, wait_stream(c->stream));
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
    // everything rides the context's stream behind the enqueued work: ONE wait for the whole batch.  Small batches (the
    // per-frame read-backs of the schedules) are written into the pinned buffer by one kernel instead of one copy command each
    auto copy_all = [&]() -> int {
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
        return TP_OK;
    };
    const int64_t failures = c->persist_failures;
    if (int rc = copy_all()) return rc;
    if (int rc = check_persist_status(c)) return rc;
    // a persistent launch ahead of the copy had given up: its calls were run again just now, and what the copy took is from before that
    // (the positions may even live in the other buffer by now) -- once more
    if (c->persist_failures != failures) { if (int rc = copy_all()) return rc; }
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

}  // extern "C"
