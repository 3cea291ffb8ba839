// tp_bands.hip -- band split of ONE descent over several GPUs (include/tpose_hip.h: tp_band_attach).
// The unit being split is a direction of a pair: software/warp/main.cpp:214-283 with source/triangulation.hpp:492-520.
#include "tp_context.h"

namespace tpctx {

// a band's mailbox: [4][cap] position slots of 16 bytes, then the rings of tp_iterate_until -- PK_RING_FRAMES frames of
// cap_tris base energies (int32) and of cap positions (float2)
size_t band_slots_bytes(size_t cap) { return (cap * 64 + 255) & ~(size_t)255; }
size_t band_ering_bytes(size_t cap_tris) { return ((size_t)PK_RING_FRAMES * cap_tris * 4 + 255) & ~(size_t)255; }
// band split: the rings of tp_iterate_until live in the bands' mailboxes (behind the position slots)
bool banded_rings(const tp_context* c) { return c->n_bands > 1 && c->band_cap_tris > 0; }
int32_t* band_ering(const tp_context* c, int b) { return (int32_t*)((char*)c->band_box[b] + band_slots_bytes(c->band_cap)); }
float2* band_pring(const tp_context* c, int b) { return (float2*)((char*)c->band_box[b] + band_slots_bytes(c->band_cap) + band_ering_bytes(c->band_cap_tris)); }

}  // namespace tpctx

using namespace tpctx;

extern "C" {

size_t tp_band_mailbox_bytes(int points, int triangles) {
    const size_t cap = (size_t)(points > 0 ? points : 0), ct = (size_t)(triangles > 0 ? triangles : 0);
    return band_slots_bytes(cap) + band_ering_bytes(ct) + (size_t)PK_RING_FRAMES * cap * 8 + 256;
}

int tp_band_attach(tp_context* c, int band, int n_bands, void* const* mailboxes, size_t bytes_each, int points, int triangles, int patches_per_band) {
    api_guard api_lock;
    if (!c) return TP_ERR_INVALID;
    c->mutations++; c->tail_is_finish = false;   // (a retrieve no longer finds what the frame mirror holds)
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
    // a mailbox in another device's memory (bands of one process, one context per device): this device must be allowed to reach it
    for (int b = 0; b < n_bands; b++) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, mailboxes[b]) != hipSuccess) { (void)hipGetLastError(); continue; }   // (an IPC mapping may not answer: it is reachable as mapped)
        if (at.device != c->device) {
            const hipError_t e = hipDeviceEnablePeerAccess(at.device, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
                return fail(c, TP_ERR_HIP, "band attach: device %d cannot reach mailbox %d on device %d: %s", c->device, b, at.device, hipGetErrorString(e));
            (void)hipGetLastError();
        }
    }
    c->band = band; c->n_bands = n_bands; c->band_patches = ppb;
    c->band_cap = cap; c->band_cap_tris = cap_tris;
    for (int b = 0; b < n_bands; b++) c->band_box[b] = (unsigned long long*)mailboxes[b];
    c->plan_generation = 0;   // the plan is cut again, into n_bands * ppb patches
    // Tags and slot parities count from the attachment, not from the context's past: the mailboxes are zeroed by their owners
    // (a cleared granule matches no tag), so every band starts at grad-iter 1, launch 0, ring chunk 0 whatever it ran before --
    // a warm-up call on one band only would otherwise leave the bands' tags apart for good.
    c->epoch = 1; c->band_seq = 0; c->ring_seq = 0; c->ring_half = 0;
    return TP_OK;
}

// A band's mailbox is polled from inside a running kernel while OTHER devices write it, so it must be memory whose remote writes a
// running kernel can see: fine-grained device memory (hipDeviceMallocFinegrained).  Ordinary hipMalloc memory is coarse-grained --
// coherent with peer writes at kernel boundaries only -- and every launch of a band split on it would wait out its time limit.
int tp_band_mailbox_alloc(tp_context* c, size_t bytes, void** out) {
    api_guard api_lock;
    if (!c || !out) return TP_ERR_INVALID;
    *out = nullptr;
    HIP_TRY(c, hipSetDevice(c->device));
    void* p = nullptr;
    hipError_t e = hipExtMallocWithFlags(&p, bytes ? bytes : 1, hipDeviceMallocFinegrained);
    if (e != hipSuccess || !p) {
        (void)hipGetLastError();
        return fail(c, TP_ERR_HIP, "band mailbox: no fine-grained device memory (%s)", hipGetErrorString(e));
    }
    e = hipMemset(p, 0, bytes ? bytes : 1);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) { hipFree(p); return fail(c, TP_ERR_HIP, "band mailbox: %s", hipGetErrorString(e)); }
    c->box_finegrained = true;
    *out = p;
    return TP_OK;
}

int tp_band_mailbox_export(tp_context* c, void* box, void* handle) {
    api_guard api_lock;
    if (!c || !box || !handle) return TP_ERR_INVALID;
    static_assert(sizeof(hipIpcMemHandle_t) <= TP_MAILBOX_HANDLE_BYTES, "the handle travels in 64 bytes");
    HIP_TRY(c, hipSetDevice(c->device));
    hipIpcMemHandle_t h;
    memset(&h, 0, sizeof h);
    HIP_TRY(c, hipIpcGetMemHandle(&h, box));
    memset(handle, 0, TP_MAILBOX_HANDLE_BYTES);
    memcpy(handle, &h, sizeof h);
    return TP_OK;
}

int tp_band_mailbox_import(tp_context* c, const void* handle, void** box) {
    api_guard api_lock;
    if (!c || !handle || !box) return TP_ERR_INVALID;
    *box = nullptr;
    HIP_TRY(c, hipSetDevice(c->device));
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof h);
    HIP_TRY(c, hipIpcOpenMemHandle(box, h, hipIpcMemLazyEnablePeerAccess));
    return TP_OK;
}

int tp_band_mailbox_close(tp_context* c, void* box) {
    api_guard api_lock;
    if (!c) return TP_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    if (int rc = tp_synchronize(c)) return rc;
    for (auto& b : c->band_box) if (b == (unsigned long long*)box) b = nullptr;
    if (box) HIP_TRY(c, hipIpcCloseMemHandle(box));
    return TP_OK;
}

int tp_band_mailbox_free(tp_context* c, void* box) {
    api_guard api_lock;
    if (!c) return TP_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    if (int rc = tp_synchronize(c)) return rc;
    for (auto& b : c->band_box) if (b == (unsigned long long*)box) b = nullptr;
    if (box) HIP_TRY(c, hipFree(box));
    return TP_OK;
}

}  // extern "C"
