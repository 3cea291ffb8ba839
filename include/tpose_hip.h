/*
 * tpose_hip.h -- C ABI of the MI355X-native t-pose hot path (libtpose_hip.so).
 *
 * Drop-in boundary.  In the reference (weigert/t-pose) this path is not behind a plugin ABI but a
 * name-bound SSBO contract between C++ and GLSL: the eight global `Buffer*` handles and four host
 * mirrors of source/triangulation.hpp:578-590, bound by name to the shaders in
 * software/triangulate/main.cpp:84-101 / software/warp/main.cpp:86-108 and driven by the lambdas
 * computecolors/doreset, doenergy, doshift (triangulate/main.cpp:121-155, warp/main.cpp:140-178).
 * Each entry point below names the reference interface it replaces.  Plain pointers and sizes
 * only; every pointer argument is caller-owned and only touched during the call; the context owns
 * all device memory.  A context is single-threaded; distinct contexts (distinct GPUs) may be
 * driven from distinct threads.  Every function returns TP_OK (0) or an error code and records a
 * message retrievable with tp_last_error().  There is NO CPU fallback: without a HIP device
 * tp_create fails with TP_ERR_NO_DEVICE.
 *
 * Buffer layouts are the reference's: variant-major `id = i*NT + t`, i = 0..12 (TDIV), t = triangle
 * (triangle.vs:47-48); triangles ivec4 (x,y,z = vertex ids, w unused); points vec2 in t-pose space
 * x in [-RATIO,RATIO], y in [-1,1], y up; colours ivec4.
 */
#ifndef TPOSE_HIP_H
#define TPOSE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TP_ABI_VERSION 5
#define TP_MAXT (2 << 18) /* tpose::triangulation::MAXT, source/triangulation.hpp:95; 13*NT <= MAXT */

typedef struct tp_context tp_context;

enum tp_status {
    TP_OK = 0,
    TP_ERR_INVALID = 1,   /* bad argument */
    TP_ERR_NO_DEVICE = 2, /* no HIP device / device index out of range */
    TP_ERR_HIP = 3,       /* a HIP runtime call failed */
    TP_ERR_CAPACITY = 4,  /* 13*NT > MAXT or NP > MAXT, raster too large */
    TP_ERR_STATE = 5      /* call order violated (e.g. energy before accumulate/upload) */
};

/* which cost function: software/triangulate/shader/ vs software/warp/shader/ */
enum tp_flavour { TP_TRIANGULATE = 0, TP_WARP = 1 };

/* image slots: `imageTexture` (triangulate) = slot A; `imageA` / `imageB` (warp triangle.fs:23-24) */
enum tp_slot { TP_IMAGE_A = 0, TP_IMAGE_B = 1 };

/* buffers readable with tp_retrieve -- the SSBOs of triangle.vs:5-32 */
enum tp_buffer {
    TP_BUF_TENERGY = 0,  /* int32[13*NT]      `tenergy`  (tpose::tenergybuf -> terr) */
    TP_BUF_COLNUM = 1,   /* int32[13*NT]      `colnum`   (tpose::tcolnumbuf -> cn) */
    TP_BUF_COLACC = 2,   /* int32[4*13*NT]    `colacc`   (tpose::tcolaccbuf) ivec4 */
    TP_BUF_POINTS = 3,   /* float[2*NP]       `points`   (tpose::pointbuf) */
    TP_BUF_GRADIENT = 4, /* int32[2*NP]       `gradient` (tpose::pgradbuf) ivec2 */
    TP_BUF_PENERGY = 5,  /* int32[count]      `penergy`  (dead in the reference: lambda = 0,
                            triangle.vs:107) -- always zeros */
    TP_BUF_MOMENTS = 6   /* int64[6*13*NT]    {n, n_odd, sum r, sum g, sum b, sum r^2+g^2+b^2} per
                            variant: the exact pixel moments the energies derive from */
};

/* parameters of a fused grad-iter; tp_default_params fills the reference's hard-coded values */
typedef struct tp_params {
    int32_t flavour;    /* tp_flavour */
    int32_t image_slot; /* raster swept by the cost function.  triangulate: A.  warp: the OTHER
                           view (warp/shader/triangle.fs:49-50: warpA -> imageB) */
    float rate;         /* shift.cs:45 -- 0.00005 (triangulate) / 0.00003 (warp) */
    float dp;           /* vertex perturbation in t-pose units; <= 0 selects the reference law
                           0.05/(1+4NT/3000) (triangle.vs:60-62) or 0.05/(1+9NT/1000) (warp :63-65) */
} tp_params;

/* library / device ------------------------------------------------------------------------- */
int tp_abi_version(void);
int tp_device_count(int* count);
const char* tp_last_error(const tp_context* ctx); /* ctx may be NULL: last error of tp_create */

/* tpose::init() (source/triangulation.hpp:592-608) + Tiny::window/Texture setup: one context per
 * device, for a width x height raster.  RATIO defaults to (float)width/(float)height
 * (software/triangulate/main.cpp:54). */
int tp_create(int device, int width, int height, tp_context** out);
/* tpose::quit() (source/triangulation.hpp:610-626) */
int tp_destroy(tp_context* ctx);

/* tpose::RATIO (source/tpose.hpp:12); io::read overwrites it (source/io.hpp:81) */
int tp_set_ratio(tp_context* ctx, float ratio);
int tp_get_ratio(const tp_context* ctx, float* ratio);
/* override the vertex perturbation `dp` (triangle.vs:60-62) for the piecewise calls below;
 * dp <= 0 restores the reference law.  tp_iterate takes its dp from tp_params instead. */
int tp_set_dp(tp_context* ctx, float dp);

/* Execution options (no counterpart in the reference, whose frame loop is fixed).  Results never depend on them.
 * TP_OPT_PERSISTENT: TP_PERSIST_AUTO (default) lets tp_iterate run calls of >= 4 grad-iters inside
 * persistent launches (one workgroup per patch of the mesh, K grad-iters per launch) when the device keeps a full grid
 * resident; TP_PERSIST_OFF keeps every grad-iter on the two-kernel path (k_lines + k_update).
 * TP_OPT_INJECT_GIVE_UP (tests; refused unless TPOSE_ALLOW_FAULT_INJECTION is set in the environment): n > 0 makes one workgroup of the n-th persistent launch from now give up before its last grad-iter,
 * as if the launch's workgroups had not all been resident: the launch and those behind it are run again on the two-kernel path
 * (tp_get_info 9 counts it) and the context keeps to the two-kernel path for a while: 0.2 s, then 0.8, 3.2 and 12.8 s while give-ups come in a
 * row (64 launches that complete end a row); never for good -- tp_get_info 12 says how long is left. */
enum tp_option { TP_OPT_PERSISTENT = 1, TP_OPT_INJECT_GIVE_UP = 3 };
enum { TP_PERSIST_OFF = 0, TP_PERSIST_AUTO = 1 };
int tp_set_option(tp_context* ctx, int option, int64_t value);

/* Band split of ONE descent over several GPUs (no counterpart in the reference, which runs a
 * direction of a pair on one GPU: software/warp/main.cpp:214-283).  `n_bands` contexts -- one per process and GPU, all given
 * the same images, uploads and calls, in the same order -- run one descent together: a persistent launch of context `band`
 * runs the patches [band, band + 1) x patches_per_band of a plan of n_bands x patches_per_band patches, and the only thing that
 * crosses between them is what crosses between workgroups: vertex positions, 16 bytes per vertex and grad-iter, posted by a
 * vertex's owner into EVERY band's mailbox (peer memory, system-scope stores) and polled by the readers from their own.
 * mailboxes[b]: band b's mailbox as THIS process addresses it -- its own allocation at [band], the others' mapped in
 * (hipIpcOpenMemHandle between processes, peer access between devices); `bytes_each` >= tp_band_mailbox_bytes(points,
 * triangles) each (the most vertices and triangles any upload will have; besides the position slots it holds the per-frame
 * rings of tp_iterate_until), zeroed by their owners before any band iterates.  patches_per_band: 0 = one per compute unit.
 * After tp_iterate every band holds ALL positions (tp_retrieve(TP_BUF_POINTS) is complete); `tenergy`, `colnum`, `colacc` and
 * `gradient` hold the entries of the band's own patches only.  tp_iterate_until splits its frames the same way -- every band
 * writes the energies of its triangles into every band's ring, and every band's host applies the convergence test to all of
 * them -- and ends, like the unsplit call, with the last frame run whole (all buffers complete on every band).  Calls too
 * short for persistent launches and the piecewise API run whole on every band.  A band that waits a second for positions
 * gives up; every band then runs the call again on its own (tp_get_info 9 counts it).  n_bands = 1 detaches.
 * Attaching starts the bands' common history: mailbox tags, final-slot and ring parities count from the attachment on
 * every band, whatever a context ran before it (so the bands need the same calls AFTER attaching, not before).
 * MEMORY TYPE: a mailbox is polled by a running kernel while other devices write it, so it must be FINE-GRAINED device memory
 * (tp_band_mailbox_alloc: hipExtMallocWithFlags(hipDeviceMallocFinegrained), zeroed).  Ordinary hipMalloc memory is coherent
 * with a peer's writes at kernel boundaries only: bands on different devices would wait out their time limit in every launch
 * (bands that share ONE device, as on a one-GPU test box, work with either kind).  tp_band_attach enables peer access from this
 * context's device to the devices the other mailboxes live on (same-process bands); between processes map them with
 * hipIpcOpenMemHandle first. */
size_t tp_band_mailbox_bytes(int points, int triangles);
int tp_band_mailbox_alloc(tp_context* ctx, size_t bytes, void** mailbox);
int tp_band_mailbox_free(tp_context* ctx, void* mailbox);
/* bands in different PROCESSES (one per GPU): a mailbox's owner exports it as an opaque 64-byte handle (hipIpcGetMemHandle), which travels
 * by any means (a pipe, RCCL, torch.distributed); every other band imports it (hipIpcOpenMemHandle) and passes the address to
 * tp_band_attach.  tp_band_mailbox_close unmaps an imported mailbox. */
#define TP_MAILBOX_HANDLE_BYTES 64
int tp_band_mailbox_export(tp_context* ctx, void* mailbox, void* handle /* [TP_MAILBOX_HANDLE_BYTES] */);
int tp_band_mailbox_import(tp_context* ctx, const void* handle, void** mailbox);
int tp_band_mailbox_close(tp_context* ctx, void* mailbox);
int tp_band_attach(tp_context* ctx, int band, int n_bands, void* const* mailboxes, size_t bytes_each, int points, int triangles,
                   int patches_per_band);

/* `Texture tex(IMG)` (software/triangulate/main.cpp:74, warp/main.cpp:118-119): RGBA8, row 0 = top,
 * width x height texels, `stride_bytes` between rows.  Host pointer.  Like the reference's texture the image
 * is uploaded once and read by every iteration after it; the upload also builds the image's row prefix table
 * (8 bytes per pixel of device memory), which is what the iterations actually read. */
int tp_set_image(tp_context* ctx, int slot, const uint8_t* rgba, size_t stride_bytes);
/* same, source already in device memory of this context's device (e.g. a torch tensor) */
int tp_set_image_device(tp_context* ctx, int slot, const void* dev_rgba, size_t stride_bytes);

/* tpose::upload(tr, uploadcolor) (source/triangulation.hpp:628-643).  points float[2*NP],
 * triangles int32[4*NT] (ivec4), colors int32[4*NT] (ivec4) or NULL (= uploadcolor false: `colacc`
 * keeps its contents).  Colours are replicated into the 13 variant blocks device-side. */
int tp_upload(tp_context* ctx, const float* points, int NP, const int32_t* triangles, int NT,
              const int32_t* colors);

/* computecolors() / doreset() -- the mode-0 draw (triangulate/main.cpp:121-130, warp/main.cpp:140-151).
 * `flavour` says which program's vertex stage applies (its dp law).  The pixel sums of every edge
 * line over raster `slot` -- for TP_WARP the image the following mode-1 pass samples -- yield exact
 * per-variant moments, from which `colnum`/`colacc` (and the mode-1 energies) derive. */
int tp_accumulate(tp_context* ctx, int flavour, int slot);
/* doenergy() -- the mode-1 draw (triangulate/main.cpp:132-141, warp/main.cpp:153-164): fills
 * `tenergy` (and `colnum`, `colacc` for TP_TRIANGULATE) from the moments of the last tp_accumulate. */
int tp_energy(tp_context* ctx, int flavour);
/* doshift() -- gradient.cs + shift.cs (triangulate/main.cpp:143-155, warp/main.cpp:166-178) */
int tp_shift(tp_context* ctx, float rate);

void tp_default_params(int flavour, tp_params* p);
/* n_iters x { tp_accumulate(image_slot); tp_energy(flavour); tp_shift(rate) } with no host
 * round trip (the reference reads back four buffers every frame, triangulate/main.cpp:201-204).
 * Asynchronous: returns after enqueueing; tp_retrieve / tp_synchronize wait.  From 4 grad-iters on they run inside
 * persistent launches (tp_set_option); the last one writes the buffers tp_retrieve reads. */
int tp_iterate(tp_context* ctx, const tp_params* p, int n_iters);
/* The reference's frame loop up to its convergence test, without a read-back per frame: frames of { accumulate; energy;
 * shift } run until |relerr| < threshold, relerr = (toterr - newerr) / toterr with newerr = sum over t < NT of
 * float(tenergy[t]) in float32, ascending t, and toterr <- newerr after every frame -- tpose::geterr
 * (source/triangulation.hpp:653-674) as software/triangulate/main.cpp:201-210 (threshold 1e-4) and software/warp/main.cpp:
 * 226-231 (1e-6) apply it -- or until max_frames frames have run.  *toterr is the caller's running total (the reference's
 * global, initially 1.0), updated frame by frame.  On return the context is in the state after the LAST frame run
 * (*frames of them; its buffers readable with tp_retrieve, positions after its step), exactly as if the frames had
 * been issued one by one with a test in between.  Frames run in chunks inside persistent launches that keep each
 * frame's base energies and starting positions on the device; the host applies the test to a chunk at a time and the
 * converged frame is then re-run to leave its buffers.  The threshold is a double because the reference compares its float
 * against a double literal (`geterr(&tr) < 1E-4`).  Synchronous. */
int tp_iterate_until(tp_context* ctx, const tp_params* p, int max_frames, double threshold, float* toterr, int* frames,
                     float* relerr /* may be NULL */);
/* The reference's frame loop with the HOST in it, but off the device's critical path (round 6).  software/triangulate/main.cpp:196-346 looks
 * at EVERY frame: geterr over the base energies (:210), then the prune / wide-angle-flip / collapse sweeps over the positions the frame
 * ended with (:316-346) -- four read-backs and a wait per frame.  tp_iterate_frames runs up to max_frames frames of { accumulate; energy;
 * shift } in chunks inside persistent launches that keep every frame's base energies and positions on the device, and hands them to the
 * caller frame by frame, in order:  fn(user, k, tenergy, points)  with k the frame's number within this call, tenergy = int32[NT] (the
 * entries [0, NT) of `tenergy` as the frame's doenergy left them) and points = float[2 NP] (the positions after the frame's shift).
 * fn returns
 *   TP_FRAME_GO_ON      the next frame, please;
 *   TP_FRAME_STOP_REPLAY  the run ends WITH this frame and the caller wants the device as the frame left it: the positions of the frame's
 *                       start are restored and the frame is run once more on the two-kernel path -- `tenergy`, `colnum`, `colacc`,
 *                       `gradient` and the positions are then exactly what the reference's frame leaves (as tp_iterate_until's last frame);
 *   TP_FRAME_STOP       the run ends with this frame and the caller will upload next: the positions the frame ended with are restored,
 *                       the other buffers are unspecified.
 * Frames the device ran beyond the one that stops the run are discarded.  *frames = frames handed to fn.  When the run ends because
 * max_frames were handed over, the device holds the positions of the last frame; the other buffers are unspecified (the next
 * tp_iterate / tp_accumulate recomputes them).  What fn does to the caller's own copy of the mesh between frames -- the reference's
 * wide-angle flips, which it never uploads (main.cpp:322-331 does not set `updated`) -- is the caller's business: the device runs on the
 * mesh of the last tp_upload, as the reference's does.  Synchronous; not available to bands (TP_ERR_STATE). */
enum { TP_FRAME_GO_ON = 0, TP_FRAME_STOP_REPLAY = 1, TP_FRAME_STOP = 2 };
typedef int (*tp_frame_fn)(void* user, int frame, const int32_t* tenergy, const float* points);
int tp_iterate_frames(tp_context* ctx, const tp_params* p, int max_frames, tp_frame_fn fn, void* user, int* frames);
/* Optional: build what tp_iterate needs for these parameters now -- the plan of the persistent launches (or the launch graph of the two-kernel
 * path) -- so that no later call pays for it.  Round 6: where persistent launches are in use it also PROBES how far the mesh's vertices move per
 * grad-iter (8 grad-iters of the persistent kernel whose results nobody keeps: positions and every buffer of the context stay as they are) and
 * cuts the plan weighted by that -- on a photograph a tenth of the patches hold the vertices that jump a pixel per grad-iter, and a plan balanced
 * by rows alone makes everybody wait for them.  The reference has no counterpart -- its frame loop issues GL calls one by one
 * (triangulate/main.cpp:190-204). */
int tp_prepare(tp_context* ctx, const tp_params* p);

/* Buffer::retrieve (triangulate/main.cpp:201-204, 221): copies `count` elements (int32 / float /
 * int64 units as listed in tp_buffer) into dst after waiting for enqueued work. */
int tp_retrieve(tp_context* ctx, int what, void* dst, size_t count);
/* The energy (triangulate flavour: triangle.fs:27-43, against the triangle's own mean colour) and pixel count each of n HYPOTHETICAL
 * triangles would have at the context's current positions on image `slot`: vertices[3k .. 3k+2] index the uploaded points; the triangles
 * need not be in the uploaded mesh, and no buffer of the context changes.  variants: NULL (base variants), or per triangle 0..12 -- what
 * entry variant * NT + t of `tenergy` would hold if the triple were triangle t of the mesh (variant i > 0: vertex (i - 1) / 4 of the
 * triple displaced by move (i - 1) % 4 + 1 of the context's dp: tp_set_dp, or the reference's law at the uploaded NT).  What the reference's convergence step obtains for its flip set
 * by flipping on the host, uploading the topology, running computecolors + doenergy over everything and reading `tenergy` back
 * (software/triangulate/main.cpp:233-306) -- a triangle's energy depends on its own pixels only.  Rasters up to 4096 columns and rows
 * (TP_ERR_STATE beyond: the caller keeps to the upload path).  count may be NULL.  Waits for the result. */
int tp_evaluate_triangles(tp_context* ctx, int slot, int n, const int32_t* vertices, const int32_t* variants, int32_t* energy, int32_t* count);
/* the same for n buffers with ONE wait: what[k] -> dst[k], count[k] elements.  The reference reads four
 * buffers back every frame (software/triangulate/main.cpp:201-204, warp/main.cpp:226-229); done one by one,
 * each is a blocking round trip. */
int tp_retrieve_many(tp_context* ctx, int n, const int* what, void* const* dst, const size_t* count);
/* tp_synchronize waits for everything the LIBRARY has enqueued for this context.  Work a caller puts on the stream of tp_get_stream
 * itself is not covered: wait for that with the HIP runtime (hipStreamSynchronize). */
int tp_synchronize(tp_context* ctx);

/* measurement hooks (bench.py): the HIP stream the kernels run on (foreign work on it: see tp_synchronize); a pair of HIP events on that
 * stream around whatever is enqueued between the two calls (tp_timer_stop waits for the second event and returns the time between them) */
int tp_get_stream(tp_context* ctx, void** hip_stream);
int tp_timer_start(tp_context* ctx);
int tp_timer_stop(tp_context* ctx, double* elapsed_us);
/* runs n_iters grad-iters eagerly with the dispatch's own timestamps around every k_lines launch; returns the
 * average duration of that kernel in microseconds */
int tp_profile_iterate(tp_context* ctx, const tp_params* p, int n_iters, double* accumulate_us);
/* average duration of k_lines as it runs inside the fused path: `launches` back-to-back launches of the kernel
 * on the current state -- idempotent -- are captured into one hipGraph and a replay is bracketed by two HIP events
 * on the context's stream; returns elapsed / launches in microseconds (includes the gap between graph nodes).
 * The triangulation is not advanced.  (tp_profile_iterate's per-dispatch timestamps need eager launches, which
 * run ~1 us longer than the same kernel inside a graph replay.) */
int tp_profile_accumulate(tp_context* ctx, const tp_params* p, int launches, double* accumulate_us);

/* Flat-shaded picture of the triangulation: every raster pixel gets the colour of the base triangle
 * that covers it (same coverage rule as the cost function, so every covered pixel is written exactly once),
 * uncovered pixels are opaque black.  Replaces the display pass `mode == 2` of
 * software/triangulate/shader/triangle.fs:45-50 (source TP_RENDER_AVERAGE: colacc/colnum of the last
 * evaluation, as the reference draws it) and of software/view/shader/triangle.fs (TP_RENDER_STORED: the
 * colours given to tp_upload).  `points` (float[2*NP], host) overrides the vertex positions for this
 * picture only -- software/view/shader/triangle.vs draws mix(points, originpoints, s) -- or NULL for
 * the context's current positions.  dst: RGBA8, row 0 on top, `stride` bytes per row. */
enum { TP_RENDER_AVERAGE = 0, TP_RENDER_STORED = 1 };
int tp_render(tp_context* ctx, int source, const float* points, uint8_t* dst_rgba, size_t stride);

/* introspection for tests/benchmarks: 0 = records per row of the prefix table, 1 = chunks per line of k_lines for the
 * current triangulation (rows of a line are shared by that many lanes); persistent path: 2 = patches (workgroups) of
 * the current plan, 0 if none, 3 = its LDS bytes per workgroup, 4 = lines all patches walk per grad-iter (9 per edge if none were walked twice),
 * 5 = persistent launches so far, 6 = grad-iters run inside them, 7 = census (1 a full grid is resident, -1 not,
 * 0 not taken yet; below -1: why not), 8 = plans cut again during long descents (vertices had drifted from where the plan saw
 * them), 9 = persistent launches that gave up waiting and were run again on the two-kernel path, 10 = this band's mailbox came
 * from tp_band_mailbox_alloc (fine-grained memory), 11 = persistent launches that started from what the launch before them left (the cut of the
 * patches' lines and the lanes' lane-items: same plan, image and dp -- a launch after tp_upload, tp_set_image or tp_set_dp never does),
 * 12 = milliseconds until persistent launches are tried again after one gave up (0: in use), 13 = rows a lane of the walk takes in the current
 * plan's largest patch (above 16: the records of the rows beyond 16 live in LDS), 14 = plans cut again because the patches had gone out of balance
 * under the speeds of their vertices (round 6: the kernel measures how far every vertex moves per grad-iter, the planner weighs its rows by it),
 * 15 = heaviest patch / mean patch of the current plan under the weights it was cut with, x 1000 */
int tp_get_info(tp_context* ctx, int what, int64_t* value);

/* device self-test of the exact span walker (tp_raster.h): for each (N0, step, d), the 32 values
 * floor((N0 + r*step)/d), r = 0..31, as the picture pass derives them.  out = int32[32*n]. */
int tp_selftest_walker(tp_context* ctx, const int64_t* N0, const int32_t* step, const int32_t* d,
                       int n, int32_t* out);


/* device self-test of the whole-line walker of k_lines (tp_raster.h, tp_setup_line): for line k
 * through the snapped points (ends[4k], ends[4k+1]) - (ends[4k+2], ends[4k+3]) (1/256 pixel) on a raster of
 * H[k] rows, out[(rows+2)k] = first row, out[(rows+2)k+1] = last row (first > last: no row), and then the crossing
 * column (not clamped to the raster) of `rows` consecutive rows from the first, from the one set-up of the line. */
int tp_selftest_line(tp_context* ctx, const int32_t* ends, const int32_t* H, int n, int rows, int32_t* out);

/* device self-test of the arithmetic between a patch's line sums and a variant's energy as k_persist runs it (round 5: packed 64-bit
 * words, float reciprocal with a remainder fix) beside the general 64-bit form (triangle.fs:37-43 / warp :46-53 from exact moments).
 * Case k: sums[12k ..] = three lines x four words {n | n_odd << 32, sum r | sum g << 32, sum b, q}; meta[8k ..] = the lines' directions
 * (+1 / -1 / 0), flip bits, flavour, stored colour r, g, b.  out[10k ..] = n, n_odd, sum r, sum g, sum b, q low, q high (packed form),
 * energy (packed form), energy (general form), 1 when the six moments of the two forms agree. */
int tp_selftest_variant(tp_context* ctx, const uint64_t* sums, const int32_t* meta, int n, int32_t* out);

#ifdef __cplusplus
}
#endif
#endif /* TPOSE_HIP_H */
