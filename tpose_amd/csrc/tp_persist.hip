// tp_persist.hip -- K grad-iters in ONE launch: the persistent form of the reference's frame
// (software/triangulate/main.cpp:132-155: doenergy -> gradient.cs -> shift.cs; software/warp/main.cpp:153-178).
//
// One workgroup per patch of the mesh (tp_plan.h), all resident at once (at most one per CU).  What a lane does in
// each phase is in tp_persist.h; this file is the choreography: the phases of a grad-iter, the workgroup barriers
// between them and the ONE hand-over between workgroups,
//   positions  owner of a vertex  ->  the owners of its neighbours     two 8-byte granules {tag : 32, float : 32}
// Every granule is written by ONE agent-scope (sc1, write-through) store and read by agent-scope loads that bypass
// the reader's L1; the tag is the grad-iter's number, so a granule is valid on its own: no flags, no fences, no order
// between granules (MI355X guide, "R2: the data IS the flag").  The mailbox is double-buffered by the parity of the
// grad-iter: a vertex's owner cannot post the position of grad-iter e + 2 before every reader has taken that of e,
// because its own grad-iter e + 1 needs the positions e + 1 of all the vertex's neighbours, and their owners -- the
// readers -- post those only after a grad-iter e that read the vertex (every vertex of a triangle is posted every
// grad-iter, the four fixed corners included, so this holds for all of them).
// Nothing here is placement-dependent: workgroup b -> patch is a permutation chosen for L2 locality only.
//
// Every wait is bounded: a lane that polls longer than PK_TIMEOUT_TICKS raises the launch's status word and the whole
// grid drains without having changed anything the host can see; at its next synchronisation the host runs the grad-iters
// of that launch (and of the launches behind it, which do nothing once the word is raised) on the two-kernel path.
#include "tp_kernels.h"
#include "tp_persist.h"
#include <type_traits>

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned int gu32;
#define PK_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
#define PK_RLX_SYSTEM __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM
#define PK_TIMEOUT_TICKS 3000000ull  // 30 ms of the 100 MHz wall clock (a hand-over takes microseconds)
#ifndef PK_RECUT_ON_DEMAND
#define PK_RECUT_ON_DEMAND 0   /* (measured: while a cut that changes one line moves every lane-item behind it, cutting more often costs more than the rows it saves) */
#endif
#ifndef PK_RECUT_MIN_GAP
#define PK_RECUT_MIN_GAP 4    /* grad-iters between two cuts of a patch's lines, at least */
#endif
#ifndef PK_CHUNK_MAJOR
#define PK_CHUNK_MAJOR 0   /* 1: a cut of everything hands slots out chunk-major (tp_persist.h) -- measured: 5 % faster on the full-contrast raster, 2 % slower on
                             the bench's; 0: line by line, a line's chunks on lanes far apart */
#endif
#ifndef PK_ROT_ALL
#define PK_ROT_ALL 0       /* 1 (experiment): every patch folds in rotated word order, not only the hot ones */
#endif
#ifndef PK_UNCACHED_PARTS
#define PK_UNCACHED_PARTS 4   /* lanes that share a lane-item without cached records when a patch has few of those */
#endif
#ifndef PK_EXP_NOFILL
#define PK_EXP_NOFILL 0   /* 1 = the first grad-iter of a launch fetches like every other (row-major table, compare first) */
#endif
#define PK_TIMEOUT_BANDS 100000000ull  // 1 s when other processes take part (their launches start when their hosts get to it)

#ifdef TPOSE_DEBUG
#define PK_STAMP(k) do { if (threadIdx.x == 0 && A.dbg && it >= A.dbg_first && it < A.dbg_first + PK_DBG_ITERS) A.dbg[((size_t)blockIdx.x * PK_DBG_ITERS + (it - A.dbg_first)) * 16 + (k)] = wall_clock64(); } while (0)
// (before the first grad-iter: into the stamps of grad-iter 0)
#define PK_STAMP0(k) do { if (threadIdx.x == 0 && A.dbg && A.dbg_first == 0) A.dbg[(size_t)blockIdx.x * PK_DBG_ITERS * 16 + (k)] = wall_clock64(); } while (0)
// the same per WAVE (its first lane; 16 stamps x 8 waves x PK_DBG_WITERS grad-iters per workgroup, behind the per-workgroup stamps) -- a
// flavour of its own (-DTPOSE_DEBUG -DPK_DBG_WAVES, tools/wave_timeline.py): eleven more stamps per wave and grad-iter cost microseconds
#ifdef PK_DBG_WAVES
#define PK_WSTAMP(k) do { if ((threadIdx.x & 63) == 0 && A.dbg && it >= A.dbg_first && it < A.dbg_first + PK_DBG_WITERS) \
    A.dbg[PK_DBG_WBASE + (((size_t)blockIdx.x * PK_DBG_WITERS + (it - A.dbg_first)) * (PK_THREADS / 64) + (threadIdx.x >> 6)) * 16 + (k)] = wall_clock64(); } while (0)
// (a figure instead of the clock, from the wave's first lane)
#define PK_WNOTE(k, v) do { if ((threadIdx.x & 63) == 0 && A.dbg && it >= A.dbg_first && it < A.dbg_first + PK_DBG_WITERS) \
    A.dbg[PK_DBG_WBASE + (((size_t)blockIdx.x * PK_DBG_WITERS + (it - A.dbg_first)) * (PK_THREADS / 64) + (threadIdx.x >> 6)) * 16 + (k)] = (unsigned long long)(v); } while (0)
#else
#define PK_WSTAMP(k) do { } while (0)
#define PK_WNOTE(k, v) do { } while (0)
#endif
#else
#define PK_STAMP(k) do { } while (0)
#define PK_STAMP0(k) do { } while (0)
#define PK_WSTAMP(k) do { } while (0)
#define PK_WNOTE(k, v) do { } while (0)
#endif


namespace {

struct spin_state { unsigned spins; unsigned long long t0; };

// one more turn of a polling loop; true: give up (somebody raised the status word, or this lane waited too long)
__device__ __forceinline__ bool spin_fail(spin_state& st, gu32* status, unsigned long long limit = PK_TIMEOUT_TICKS) {
    if ((++st.spins & 63u) == 0u) {
        if (__hip_atomic_load(status, PK_RLX_AGENT) != 0u) return true;
        const unsigned long long now = wall_clock64();
        if (st.t0 == 0ull) st.t0 = now;
        else if (now - st.t0 > limit) { __hip_atomic_store(status, 1u, PK_RLX_AGENT); return true; }
    }
    __builtin_amdgcn_s_sleep(1);
    return false;
}

__device__ __forceinline__ int patch_of_block(int b, int parts) {
    // workgroup b runs on XCD b mod 8 (observed, never relied on): give every XCD one run of neighbouring patches
    return (parts & 7) == 0 ? (b & 7) * (parts >> 3) + (b >> 3) : b;
}

}  // namespace

// RR: table records a lane keeps per lane-item = the most rows per lane of any patch of the plan (the launcher picks the
// smallest instantiation that covers the plan: fewer rows, fewer registers and less straight-line code)
// MODE: 0 a plain tp_iterate, 1 with the rings of tp_iterate_until, 2 a band of a split descent (rings decided at run time).
// (One kernel for all three kept a dozen pointers of the rare cases in scalar registers -- 130 of them spilled to vector
// lanes -- and cost every grad-iter of the common case 0.3 us.)
template <int RT, int MODE>
__global__ __launch_bounds__(PK_THREADS) __attribute__((amdgpu_waves_per_eu(PK_THREADS / 256, PK_THREADS / 256))) void k_persist(pk_args A) {
    constexpr int RR = RT > PK_ROWS_PER_LANE ? PK_ROWS_PER_LANE : RT, RL = RT - RR;   // rows per lane whose records live in registers / in LDS
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int tid = threadIdx.x;
    const int part = A.part0 + patch_of_block((int)blockIdx.x, (int)gridDim.x);
    const pk_wg w = A.wg[part];
    pk_view V;
    pk_carve(smem, w, V);
    gu32* status = (gu32*)A.status;

    if (A.n_iters < 0) {
        // census (once per context): are all workgroups of this grid resident together?  Everyone arrives at one counter
        // and waits for the others; a launch whose last workgroups only start when earlier ones exit times out.
        if (tid == 0) {
            gu32* cnt = status + 1;
            __hip_atomic_fetch_add(cnt, 1u, PK_RLX_AGENT);
            spin_state st = {0u, 0ull};
            while (__hip_atomic_load(cnt, PK_RLX_AGENT) < gridDim.x)
                if (spin_fail(st, status)) break;
        }
        return;
    }

#ifdef PK_DBG_BOUNDS
    if (tid == 0) g_pk_fault[0] = (unsigned long long)A.vw.H * (unsigned long long)A.px_pitch * 16ull;
#endif
    // a launch behind one that gave up does nothing (tp_context.hip: the grad-iters are run again on the two-kernel path)
    // (one answer for the whole workgroup: the word may be raised while its threads look)
    if (__syncthreads_or(__hip_atomic_load(status, PK_RLX_AGENT) != 0u)) {
        if (A.host_status && tid == 0 && blockIdx.x == 0) __hip_atomic_store((gu32*)A.host_status, 1u, PK_RLX_SYSTEM);
        return;
    }
    PK_STAMP0(12);
    // carry (below): requested first, so that the words are on their way while the tables come over
    int32_t* const carry = A.carry && w.n_lines_all + 1 <= A.carry_cut_cap ? A.carry + (size_t)part * (size_t)A.carry_stride : nullptr;
    int32_t cy_hdr[5] = {0, 0, 0, 0, 0}, cy_item[PK_NI][3], cy_cut[3] = {0, 0, 0};
    if (carry) {
#pragma unroll
        for (int q = 0; q < 5; q++) cy_hdr[q] = carry[q];
#pragma unroll
        for (int i = 0; i < PK_NI; i++)
#pragma unroll
            for (int q = 0; q < 3; q++) cy_item[i][q] = carry[8 + 3 * A.carry_cut_cap + 3 * (tid + i * PK_THREADS) + q];
        if (tid <= w.n_lines_all)
#pragma unroll
            for (int q = 0; q < 3; q++) cy_cut[q] = carry[8 + q * A.carry_cut_cap + tid];   // (tl, nc, cut)
    }

    // ---- prologue: the patch's tables and positions into LDS.  The plan lays a patch's tables out in its pool the way pk_carve lays them
    // out in LDS -- {vid, edges, lines} and {corners, base}, every table padded to 16 bytes (tp_plan.h: put) -- so they come over as two runs
    // of 16-byte words, normally one load per thread and all of them in flight together (six loops of dependent 4-byte loads, and the
    // per-thread constants below read from the pool again, cost a short call 10 us before its first grad-iter; this way it is 2).
    {
        const int32_t* pool = A.pool;
        const int n_a = (pk_align16(w.n_slots * 4) + pk_align16(w.n_edges * 4) + pk_align16(w.n_lines_all * 4)) >> 4, n_b = w.n_corners + w.n_base;
        const pk_i4* src_a = reinterpret_cast<const pk_i4*>(pool + w.off_vid);
        const pk_i4* src_b = reinterpret_cast<const pk_i4*>(pool + w.off_corners);
        pk_i4* dst_a = reinterpret_cast<pk_i4*>(V.vid);
        pk_i4* dst_b = V.corners;
        for (int i = tid; i < n_a + n_b; i += PK_THREADS) {
            if (i < n_a) dst_a[i] = src_a[i]; else dst_b[i - n_a] = src_b[i - n_a];
        }
        for (int i = tid; i < w.n_slots; i += PK_THREADS) {
            const float2 p = A.points[pool[w.off_vid + i]];
            V.pos[i].x = p.x; V.pos[i].y = p.y;
        }
        for (int i = tid; i < PK_SUM_STRIDE * w.n_lines_all; i += PK_THREADS) V.sums[i] = 0ull;
        for (int k = tid; k < w.n_own_v; k += PK_THREADS) { V.gacc[2 * k] = 0ull; V.gacc[2 * k + 1] = 0ull; V.vdeg[k] = 0; V.spd[2 * k] = 0.0f; V.spd[2 * k + 1] = 0.0f; }
        if (tid < 16) V.flags[tid] = 0;   // ([3]: a lane gave up; [8]: the lines want cutting again; [9]: free slots filed while they are)
        for (int i = tid; i < PK_CACHED; i += PK_THREADS) V.st[i] = -1;
    }
    __syncthreads();
    PK_STAMP0(13);
    // corners of every own vertex: the lane that brings a vertex's count to this number has seen all its central differences
    for (int k = tid; k < w.n_corners; k += PK_THREADS) atomicAdd(&V.vdeg[(V.corners[k].y >> 2) & 0x3ff], 1);
    // the stored colour of this lane's variant (warp flavour: `colacc` as uploaded, triangle.fs:49-50) never changes
    // during a launch; the first pass of the corner lanes keeps it in registers
    pk_i4 col0 = {0, 0, 0, 0};
    if (A.flavour == 1 && tid < 4 * w.n_corners) {
        const int k = tid >> 2, m = (tid & 3) + 1;
        const int t = V.corners[k].x, s = V.corners[k].y & 3;
        const int4 c = A.ca[(size_t)(4 * s + m) * A.NT + t];
        col0.x = c.x; col0.y = c.y; col0.z = c.z; col0.w = c.w;
    }
    // the ends of this thread's first line never change during a launch: slot u | slot v << 10 | version << 20
    int my_ends = 0;
    float my_dxu = 0.0f, my_dyu = 0.0f, my_dxv = 0.0f, my_dyv = 0.0f;   // (... and how its endpoints are displaced)
    if (tid < w.n_lines_all) {
        const int ln_ = V.lines[tid], ed_ = V.edges[ln_ & 0xffff];
        my_ends = (ed_ & 0x3ff) | (((ed_ >> 16) & 0x3ff) << 10) | ((ln_ >> 16) << 20);
        const int q_ = ln_ >> 16, mu_ = (q_ >= 1 && q_ <= 4) ? q_ : 0, mv_ = q_ >= 5 ? q_ - 4 : 0;
        my_dxu = tp_move_dx(mu_, A.vw.dp); my_dyu = tp_move_dy(mu_, A.vw.dp); my_dxv = tp_move_dx(mv_, A.vw.dp); my_dyv = tp_move_dy(mv_, A.vw.dp);
    }
    // ... and neither do the line-sum slots of its first corner variant: edge leaving the vertex | arriving << 16; opposite | own slot << 16
    int my_c0 = 0, my_c1 = 0;
    if (tid < 4 * w.n_corners) {
        const pk_i4 cr_ = V.corners[tid >> 2];
        const int m_ = tid & 3;
        my_c0 = ((cr_.z & 0xffff) + m_) | ((((cr_.z >> 16) & 0xffff) + m_) << 16);
        my_c1 = (cr_.w & 0xffff) | (((cr_.y >> 2) & 0x3ff) << 16) | (((cr_.w >> 16) & 7) << 26);   // (... | flips << 26)
    }
    const char* table = reinterpret_cast<const char*>(A.px);
    const char* tiled = reinterpret_cast<const char*>(A.px_tiled);
    // this thread's lane-item of the walk and the table records of its rows: in registers for the whole launch
    pk_lane_cache<RR> cache[PK_NI];
    gu64* posbox = (gu64*)A.posbox;
    const bool banded = MODE == 2;
    int32_t* const ering = MODE == 0 ? nullptr : A.ering;
    float2* const pring = MODE == 0 ? nullptr : A.pring;
    int failed = 0;
    int n_unc_now = 0, n_unc_all_now = 0;   // lane-items WITHOUT a slot: of the lines walked every grad-iter / with the last one's base lines
    __syncthreads();
    PK_STAMP0(14);

    // the mailbox slot of the first foreign vertex this lane polls (its number is a table look-up the poll would otherwise start with)
    const int my_vid = w.n_own_v + tid < w.n_slots ? V.vid[w.n_own_v + tid] : 0;
    // ---- carry: what the launch before this one left for it (same plan, image, dp -- the tag says so): the cut of the patch's lines, this
    // thread's lane-item, the rows per lane and how long ago the lines were cut.  A launch that finds them neither counts chunks nor searches
    // for its lane-items in its first grad-iter (1.2 + 1.5 us of a short call), and cuts again when the lines are due, not at its start.
    int age0 = 0, last_cut = 0;
    bool warm = carry && (unsigned)cy_hdr[0] == A.carry_tag;
    if (warm) {
        age0 = cy_hdr[1] & (PK_RECUT - 1);
        warm = age0 != 0;   // (due at once: the first grad-iter cuts as a cold launch does)
        if (!warm) age0 = 0;
    }
    if (warm) {
        if (tid < w.n_lines_all) { V.tl[tid] = cy_cut[0]; V.nc[tid] = cy_cut[1]; }
        if (tid <= w.n_lines_all) V.cut[tid] = cy_cut[2];
        for (int i = tid + PK_THREADS; i <= w.n_lines_all; i += PK_THREADS) {
            if (i < w.n_lines_all) { V.tl[i] = carry[8 + i]; V.nc[i] = carry[8 + A.carry_cut_cap + i]; }
            V.cut[i] = carry[8 + 2 * A.carry_cut_cap + i];
        }
        if (tid == 0) { V.flags[1] = 0; V.flags[2] = cy_hdr[2]; }
        n_unc_now = cy_hdr[3]; n_unc_all_now = cy_hdr[4];
#pragma unroll
        for (int i = 0; i < PK_NI; i++) {
            const int lc = cy_item[i][0], TL = cy_item[i][1];
            cache[i].l = lc & 0xffff; cache[i].c = (int)((unsigned)lc >> 16); cache[i].TL = TL; cache[i].magic = (uint32_t)cy_item[i][2];
            cache[i].row0 = 0xffffffffu;   // (matches no row: the first walk drops the columns and fetches)
        }
        // (the table of the lane-items no thread keeps records for -- the patch's overflow, and the last grad-iter's base lines -- is listed
        // behind the first barrier of the first grad-iter, where the cut is visible to everybody)
    } else {
#pragma unroll
        for (int i = 0; i < PK_NI; i++) pk_slot_clear(cache[i]);   // (no slot has a lane-item before the first cut)
    }
    for (int it = 0; it < A.n_iters; it++) {
#if defined(__HIP_DEVICE_COMPILE__)
        // Nothing derived from the thread's number is kept across grad-iters: left alone, the compiler hoists some forty addresses out of
        // this loop, and with them the kernel needs 253 registers where 167 do -- the difference between two waves per SIMD and three.
        // (What a chain would wait for -- the polled vertex, the first line's ends, the first corner's slots -- is kept by name, above.)
        asm volatile("" : "+v"(tid));
#endif
        const uint32_t epoch = A.epoch + (uint32_t)it, tag = pk_tag(epoch), par = epoch & 1u;
        // the last grad-iter of a call that wants the reference's buffers also walks the base lines of the base variants
        const bool last = it + 1 == A.n_iters, emit = last && A.emit;
        // every PK_RECUT grad-iters the patch looks at the chunks of its lines again (tp_persist.h, pk_recut_line)
        // ... and at once when a lane has found more rows than it keeps records for (a line outgrew its chunks: those rows are fetched again every
        // grad-iter, a memory latency or two inside the sums of its wave, until the lines are cut again -- on a raster of three times the
        // contrast a tenth of the patches spent 2.4 us of every grad-iter waiting for such a wave; profiles/r05_experiments.txt)
        const bool recut = ((it + age0) & (PK_RECUT - 1)) == 0 ||   // (age0: grad-iters since the cut a warm launch inherited)
                           (PK_RECUT_ON_DEMAND && it > 0 && it - last_cut >= PK_RECUT_MIN_GAP && __hip_atomic_load(&V.flags[8], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0);
        if (recut) last_cut = it;
        const int n_lines = emit ? w.n_lines_all : w.n_lines, n_setup = recut ? w.n_lines_all : n_lines;
        PK_STAMP(0); PK_WSTAMP(0);
        // ---- P0: positions of the neighbouring vertices this patch uses (the first grad-iter of a launch read `points`)
        // (One request per granule in flight, issued by the lanes that have just finished the corners: round 5 measured the alternatives -- the
        // idle last wave polling from the end of its walk on, two or four requests in flight, a 16-byte load per vertex, a cache line or
        // 256 bytes per vertex, a pause before the first poll -- and every one that adds requests makes the hand-over SLOWER (4.75 -> 5.0 /
        // 5.3 / 5.6 us per grad-iter); the others change nothing.  tools/handover_bench.hip: 0.93 us per hand-over, the store's trip to the
        // memory side plus the load's, whatever the layout.)
        if (it > 0) {
            for (int s = w.n_own_v + tid; s < w.n_slots; s += PK_THREADS) {
                gu64* g = posbox + ((size_t)par * A.box_stride + (s < w.n_own_v + PK_THREADS ? my_vid : V.vid[s])) * 2;
                spin_state st = {0u, 0ull};
                unsigned long long a, b;
                for (;;) {
                    if (banded) { a = __hip_atomic_load(g, PK_RLX_SYSTEM); b = __hip_atomic_load(g + 1, PK_RLX_SYSTEM); }   // (written by another device)
                    else { a = __hip_atomic_load(g, PK_RLX_AGENT); b = __hip_atomic_load(g + 1, PK_RLX_AGENT); }
                    if ((uint32_t)(a >> 32) == tag && (uint32_t)(b >> 32) == tag) break;
                    if (spin_fail(st, status, banded ? PK_TIMEOUT_BANDS : PK_TIMEOUT_TICKS)) { failed = 1; break; }
                }
                V.pos[s].x = __uint_as_float((uint32_t)a); V.pos[s].y = __uint_as_float((uint32_t)b);
            }
        }
        // (a lane that gave up says so in LDS; ONE barrier, then everybody reads the word -- __syncthreads_or is a wave
        // reduction, three barriers and three dependent LDS operations: 0.2 us of every grad-iter)
        if (A.inject_give_up && last && tid == 0 && blockIdx.x == gridDim.x / 2u) { __hip_atomic_store(status, 1u, PK_RLX_AGENT); failed = 1; }
        if (failed) V.flags[3] = 1;
        PK_WSTAMP(1);
        __syncthreads();
        PK_WSTAMP(2);
        if (V.flags[3]) {
            if (A.host_status && tid == 0) __hip_atomic_store((gu32*)A.host_status, 1u, PK_RLX_SYSTEM);   // (nobody else will tell the host)
            return;
        }
        PK_STAMP(1);
        // ---- P1: line set-up (low threads), snapped positions (high threads), gradient reset
        for (int l = tid; l < n_setup; l += PK_THREADS) {
            pk_walker wk;
            int dir;
            if (l == tid) dir = pk_setup_moved(V, A.vw, my_ends & 0x3ff, (my_ends >> 10) & 0x3ff, my_dxu, my_dyu, my_dxv, my_dyv, wk);
            else dir = pk_setup_lane(V, A.vw, l, wk);
            V.wk[l] = wk; V.ldir[l] = dir;
        }
        {
            // (the line sums of the grad-iter before: every corner has read them by now -- the barrier behind P0 was passed)
            if (it > 0)
                for (int i = PK_THREADS - 1 - tid; i < PK_SUM_STRIDE * w.n_lines; i += PK_THREADS) V.sums[i] = 0ull;
            for (int k = tid; k < w.n_own_v; k += PK_THREADS) {
                if (pring) {   // (a frame can be returned to)
                    const size_t at = (size_t)it * A.NP + V.vid[k];
                    if (banded) {   // (system scope, write-through: the other bands' posts land in the same lines)
                        const unsigned long long w8 = (unsigned long long)__float_as_uint(V.pos[k].x) | ((unsigned long long)__float_as_uint(V.pos[k].y) << 32);
                        __hip_atomic_store((gu64*)pring + at, w8, PK_RLX_SYSTEM);
                        for (int b = 0; b < A.n_peers; b++) __hip_atomic_store((gu64*)A.peer_pring[b] + at, w8, PK_RLX_SYSTEM);
                    } else pring[at] = make_float2(V.pos[k].x, V.pos[k].y);
                }
            }
        }
        PK_WSTAMP(3);
        __syncthreads();
        PK_WSTAMP(4);
        PK_STAMP(6);
        if (warm && it == 0) {
            for (int l = tid; l < w.n_lines_all; l += PK_THREADS) pk_list_line(V, l, w.li_cap);
            __syncthreads();
        }
        if (recut) {
            // pass A (tp_persist.h, "SLOTS"): what every line wants now
            if (tid < 64) {
                // (the first cut of a launch without a carry cuts everything afresh: every slot's lane-item is decided there)
                int changed = 0, rpl = it == 0 ? w.rows : V.flags[2];   // (a warm launch never cuts in its first grad-iter)
                bool first = it == 0;
                for (;;) {
                    changed = 0;
                    int every = pk_cut_want(V, w.n_lines_all, w.n_lines, tid, 64, rpl, first, changed);
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) every += __shfl_xor(every, d);
                    if (every <= PK_CACHED || rpl >= RT) break;
                    rpl++; first = true;   // (more chunks than slots: a row more per lane, everything afresh)
                }
                // (afresh counts as a change whatever the lines say: a patch WITHOUT lines -- a crowded mesh leaves some -- must still get its
                // (empty) tables written, or its threads walk whatever the LDS held: found by config 3's coarsest level, 2 runs in 3)
                changed = first ? 2 : (__any(changed) ? 1 : 0);
                if (!changed) pk_cut_forget(V, w.n_lines_all, tid, 64);
                if (tid == 0) { V.flags[1] = changed; V.flags[2] = rpl; V.flags[8] = 0; V.flags[9] = 0; }
            }
            __syncthreads();
            PK_STAMP(7);
            const int how = V.flags[1];   // 0: no line changed, 1: some did, 2: everything is cut afresh
            if (how) {
                // pass B: a slot whose line dropped its chunk is given up (afresh: every slot); every free slot files itself
#pragma unroll
                for (int i = 0; i < PK_NI; i++)
                    if (pk_slot_release(cache[i], V, how == 2)) {
                        if (how != 2) V.freel[__hip_atomic_fetch_add(&V.flags[9], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)] = tid + i * PK_THREADS;
                        else if (!PK_CHUNK_MAJOR) V.freel[w.hot ? tid + i * PK_THREADS : pk_place_of_slot(tid + i * PK_THREADS)] = tid + i * PK_THREADS;   // (hot: a line's chunks on adjacent lanes)
                    }
                __syncthreads();
                // pass C: free slots to the chunks that want one; the uncached lane-items numbered
                if (tid < 64) {
                    int unc;
                    if (how == 2 && PK_CHUNK_MAJOR) {   // chunk-major: the chunks 0 of all lines, then the chunks 1, ... (slot = place)
                        pk_cut_fresh_begin(V, w.n_lines_all, tid, 64);
                        int base = 0;
                        for (int c = 0;; c++) {
                            const int k = pk_cut_level_count(V, w.n_lines, tid, 64, w.n_lines_all, c);
                            int incl = k;
#pragma unroll
                            for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); incl += tid >= d ? o : 0; }
                            const int total = __shfl(incl, 63);
                            if (total == 0) break;
                            pk_cut_level_assign(V, w.n_lines, tid, 64, w.n_lines_all, c, base + incl - k, PK_CACHED);
                            base += total;
                        }
                        unc = pk_cut_fresh_done(V, w.n_lines_all, tid, 64);
                    } else {
                        if (how == 2) pk_cut_fresh_begin(V, w.n_lines_all, tid, 64);   // (no chunk keeps a slot)
                        const int n_free = how == 2 ? PK_CACHED : V.flags[9];
                        const int need = pk_cut_need(V, w.n_lines_all, w.n_lines, tid, 64);
                        int incl = need;
#pragma unroll
                        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); incl += tid >= d ? o : 0; }
                        unc = pk_cut_alloc(V, w.n_lines_all, w.n_lines, tid, 64, incl - need, n_free);
                    }
                    int uincl = unc;
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(uincl, d); uincl += tid >= d ? o : 0; }
                    pk_cut_write(V, w.n_lines_all, tid, 64, uincl - unc);
                }
                __syncthreads();
                // pass D: the slots that were handed a lane-item take it; the table of the uncached ones
#pragma unroll
                for (int i = 0; i < PK_NI; i++) pk_slot_take(cache[i], V, tid + i * PK_THREADS);
                for (int l = tid; l < w.n_lines_all; l += PK_THREADS) pk_list_line(V, l, w.li_cap);
                __syncthreads();   // (the table is read by other threads than wrote it)
            }
            n_unc_now = V.cut[w.n_lines]; n_unc_all_now = V.cut[w.n_lines_all];
        }
        const int n_unc = emit ? n_unc_all_now : n_unc_now;
        PK_STAMP(2);
        // ---- P3: walk -- one table record per (line, row); chunks of a line meet in LDS
        auto fold = [&](int l, const pk_acc& a, int rot = 0) {
            if (a.xs | a.nodd | a.r | a.q) {
                unsigned long long* s = V.sums + (size_t)l * PK_SUM_STRIDE;
                unsigned long long wd[PK_SUM_WORDS];
                pk_fold_words(a, wd);
                if (w.hot || PK_ROT_ALL) {
                    // (a hot patch keeps a line's chunks on ADJACENT lanes: word q of all of them in one instruction is one address, and same-address
                    // LDS atomics serialise -- so lane k starts with word k mod 4: four neighbours, four words)
#pragma unroll
                    for (int q = 0; q < PK_SUM_WORDS; q++) {
                        const int k = (q + rot) & 3;
                        const unsigned long long v = k == 0 ? wd[0] : k == 1 ? wd[1] : k == 2 ? wd[2] : wd[3];
                        atomicAdd(&s[k], v);
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < PK_SUM_WORDS; q++) atomicAdd(&s[q], wd[q]);
                }
            }
        };
        // Lane-items WITHOUT a slot (a patch with more chunks than slots; the base lines of a call's last grad-iter), every other grad-iter
        // in the opposite order (the same set of lines every grad-iter and larger than the L2: taken the same way round each time nothing
        // of one grad-iter's reads would still be there for the next).  When a patch has few of them each is cut into PK_UNCACHED_PARTS
        // parts on as many lanes: one memory latency instead of four in a row.
        const bool back = (it & 1) != 0;
        const int extra = n_unc;
        const int parts = extra * PK_UNCACHED_PARTS <= PK_THREADS ? PK_UNCACHED_PARTS : 1;
        // (each step for every lane-item of the thread -- PK_NI: one, since round 5 -- before the next step: the fetches are in flight together)
        {
            int rows[PK_NI];
#pragma unroll
            for (int i = 0; i < PK_NI; i++)
                rows[i] = (it == 0 && tiled && !PK_EXP_NOFILL) ? pk_walk_fill<RR, RL>(cache[i], V, tid + i * PK_THREADS, A.px_pitch, tiled, table, A.vw.W)
                                                                : pk_walk_pass<RR, RL>(cache[i], V, tid + i * PK_THREADS, A.px_pitch, table, A.vw.W, tiled, w.hot != 0);
            PK_STAMP(8); PK_WSTAMP(5);
            if (RL > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the records requested into LDS have landed: the compiler does not count those)
#pragma unroll
            for (int i = 0; i < PK_NI; i++) {
                pk_acc a;
                pk_walk_sum<RR, RL>(cache[i], rows[i], V, tid + i * PK_THREADS, A.px_pitch, table, A.vw.W, a);
                if (rows[i] > RT) V.flags[8] = 1;   // (the lines want cutting again)
                fold(cache[i].l, a, tid);
            }
            PK_STAMP(9); PK_WSTAMP(15);
            // (how many of the wave's lanes have more rows than records | lane-items without a slot beyond those in LDS << 16 | those in LDS << 32)
            PK_WNOTE(9, (unsigned long long)__popcll(__ballot(rows[0] > RT)) | ((unsigned long long)extra << 16));
        }
        for (int k = tid; k < extra * parts; k += PK_THREADS) {
            const int kk = parts > 1 ? k / PK_UNCACHED_PARTS : k, sub = parts > 1 ? k % PK_UNCACHED_PARTS : 0;   // (sub: which part of the lane-item; `part` is the patch)
            const int j = back ? extra - 1 - kk : kk;
            pk_acc a;
            const int l = pk_walk_lane(V, table, tiled, A.px_pitch, A.vw.W, w.n_lines_all, 0, w.li_cap, j, a, sub, parts);
            fold(l, a);
        }
        PK_WSTAMP(6);
        __syncthreads();
        PK_STAMP(3); PK_WSTAMP(7);
        if (ering && !emit) {   // tp_iterate_until: the energy of the base variants, frame by frame (the plan walks their lines every grad-iter)
            // (by the LAST threads of the workgroup, round 6: the first ones form the corners right below, and a thread that did both had a
            // base variant's chain -- three line sums, an average, an energy: 0.4 us -- in front of its corner's, on the path to the step)
            for (int k = PK_THREADS - 1 - tid; k < w.n_base; k += PK_THREADS) {
                int t;
                const pk_var mv = pk_base_var(V, k, t);
                pk_i4 col = {0, 0, 0, 0};
                if (A.flavour == 1) { const int4 c = A.ca[t]; col.x = c.x; col.y = c.y; col.z = c.z; }
                const int32_t en = pk_energy_var(mv, A.flavour, col);
                const size_t at = (size_t)it * A.NT + t;
                if (banded) {
                    __hip_atomic_store((gu32*)ering + at, (unsigned)en, PK_RLX_SYSTEM);
                    for (int b = 0; b < A.n_peers; b++) __hip_atomic_store((gu32*)A.peer_ering[b] + at, (unsigned)en, PK_RLX_SYSTEM);
                } else ering[at] = en;
            }
        }
        // (band split: what a launch ends with is a release for the rings -- the last posts below must not overtake this grad-iter's ring stores)
        if (banded && last) __syncthreads();
        // ---- P6: corners -- four displaced variants each, central differences into the vertex's gradient (int32 wrapping
        // sums, like the reference's atomics: gradient.cs:24-35)
        for (int j = tid; j < 4 * w.n_corners; j += PK_THREADS) {
            const int k = j >> 2, m = (j & 3) + 1;
            pk_i4 col = col0;
            if (A.flavour == 1 && j >= PK_THREADS) {
                const pk_i4 cq = V.corners[k];
                const int4 c = A.ca[(size_t)(4 * (cq.y & 3) + m) * A.NT + cq.x];
                col.x = c.x; col.y = c.y; col.z = c.z;
            }
            int so, si, sopp, own, flips;
            if (j == tid) {
                // (decoded here, every grad-iter: hoisted out of the loop the slot addresses and flip masks take a dozen registers from the walk,
                // and the two that no longer fit come back from scratch memory right on the chain between the corners and the step)
                int c0_ = my_c0, c1_ = my_c1;
#if defined(__HIP_DEVICE_COMPILE__)
                asm volatile("" : "+v"(c0_), "+v"(c1_));
#endif
                so = c0_ & 0xffff; si = (int)((unsigned)c0_ >> 16); sopp = c1_ & 0xffff; own = (c1_ >> 16) & 0x3ff; flips = c1_ >> 26;
            }
            else {
                const pk_i4 cq = V.corners[k];
                so = (cq.z & 0xffff) + m - 1; si = ((cq.z >> 16) & 0xffff) + m - 1; sopp = cq.w & 0xffff; own = (cq.y >> 2) & 0x3ff; flips = (cq.w >> 16) & 7;
            }
            const pk_var mv = pk_signed_packed(V, so, si, sopp, flips);
#ifdef PK_DBG_WAVES
            if (j == tid) { asm volatile("" :: "v"(mv.n), "v"(mv.q)); PK_WSTAMP(11); }
#endif
#if defined(PK_EXP_OLDP6)   // timing experiments only (tools/build_variants.py): the general 64-bit form
            const int32_t e = pk_energy(pk_corner_moments(V, so, si, sopp, flips), A.flavour, col);
#else
            const int32_t e = pk_energy_var(mv, A.flavour, col);
#endif
            // lanes 4k+0/1: E(+dx), E(-dx); 4k+2/3: E(+dy), E(-dy) -- the neighbour's energy by a DPP quad permute [1,0,3,2] (__shfl_xor
            // goes through the LDS crossbar: a hundred cycles on this chain)
#ifdef PK_DBG_WAVES
            if (j == tid) { asm volatile("" :: "v"(e)); PK_WSTAMP(12); }
#endif
            const uint32_t d = (uint32_t)e - (uint32_t)__builtin_amdgcn_mov_dpp(e, 0xB1, 0xF, 0xF, true);
            if (emit) {
                const pk_i4 cr = V.corners[k];   // the variant's outputs in the reference's layout, id = i NT + t (triangle.vs:47-48)
                const size_t id = (size_t)(4 * (cr.y & 3) + m) * A.NT + cr.x;
                if (A.flavour == 0) A.ca_out[id] = make_int4((int32_t)mv.r, (int32_t)mv.g, (int32_t)mv.b, 0);
                A.ten[id] = e; A.cn[id] = (int32_t)mv.n;
            }
            // ---- P7, by the lane that completes an axis of a vertex.  Lanes 4k + 0 / 4k + 2 add the corner's x / y difference to the vertex's
            // accumulator of that axis with ONE returning atomic, (difference << 32) | 1: the lane whose returned count is the vertex's last
            // holds the sum of all the others in the same word -- nothing else to read, nothing to order -- and takes that axis' shift.cs
            // step right here, no workgroup barrier in between; the new coordinate goes to the mailbox of the next grad-iter (one granule per
            // coordinate) or, after the last one, to `points_out`.
            if ((j & 1) == 0) {
                const int ax = (j >> 1) & 1;   // 0: x, 1: y
                unsigned long long* acc = V.gacc + 2 * own + ax;
                const unsigned long long old = atomicAdd(acc, pk_gacc_word(d));
#ifdef PK_DBG_WAVES
                if (j == tid) { asm volatile("" :: "v"(old)); PK_WSTAMP(13); }
#endif
                if ((uint32_t)old + 1u == (uint32_t)V.vdeg[own]) {
                    const int32_t g = (int32_t)((uint32_t)(old >> 32) + d);
                    *acc = 0ull;   // (for the next grad-iter: every corner of the vertex has added to it)
                    const int v = V.vid[own];
                    float* const pc = ax ? &V.pos[own].y : &V.pos[own].x;
                    if (emit) reinterpret_cast<int32_t*>(A.gr + v)[ax] = g;
                    const float po = *pc;
#if defined(PK_EXP_FREEZE)  // timing experiments only: the mesh stands still
                    const float pn = po;
#else
                    const float pn = v < 4 ? po : pk_step_axis(po, g, ax ? 1.0f : A.vw.ratio, A.rate);   // (vertices 0..3 never move: shift.cs:20)
#endif
                    *pc = pn;
                    if (last) reinterpret_cast<float*>(A.points_out + v)[ax] = pn;
                    if (!last || banded) {
                        // (band split: the positions a launch ends with go to the slots 2, 3 of every band's mailbox, for tp_launch_band_collect)
                        const unsigned long long T = (unsigned long long)(last ? A.final_tag : pk_tag(epoch + 1u)) << 32;
                        const size_t at = ((size_t)(last ? 2u + A.final_slot : (epoch + 1u) & 1u) * A.box_stride + v) * 2 + ax;
                        const unsigned long long gw = T | __float_as_uint(pn);
                        if (banded) {
                            // (what a launch ends with is a RELEASE: whoever collects it also sees the rings this workgroup wrote -- the
                            // barriers between the phases, and the one in front of the last grad-iter's corners, order them before this)
                            if (last) __atomic_thread_fence(__ATOMIC_RELEASE);
                            __hip_atomic_store(posbox + at, gw, PK_RLX_SYSTEM);
                            for (int b = 0; b < A.n_peers; b++) __hip_atomic_store((gu64*)A.peer_box[b] + at, gw, PK_RLX_SYSTEM);
                        } else {
                            __hip_atomic_store(posbox + at, gw, PK_RLX_AGENT);
                        }
                    }
                    // (behind the post: how far the vertex went, for the planner -- one lane per vertex and axis and grad-iter, so a plain add)
                    if (A.vspeed) V.spd[2 * own + ax] += fabsf(pn - po);
                }
            }
        }
        if (emit) {   // base variants (i = 0) of the triangles whose first vertex this patch owns (the last threads: beside the corners, not behind them)
            for (int k = PK_THREADS - 1 - tid; k < w.n_base; k += PK_THREADS) {
                int t;
                const pk_var mv = pk_base_var(V, k, t);
                pk_i4 col = {0, 0, 0, 0};
                if (A.flavour == 1) { const int4 c = A.ca[t]; col.x = c.x; col.y = c.y; col.z = c.z; }
                if (A.flavour == 0) A.ca_out[t] = make_int4((int32_t)mv.r, (int32_t)mv.g, (int32_t)mv.b, 0);
                const int32_t en = pk_energy_var(mv, A.flavour, col);
                A.ten[t] = en; A.cn[t] = (int32_t)mv.n;
                // (tp_iterate_until's last chunk: its last frame leaves the reference's buffers AND its ring entry -- no frame run twice)
                if (ering && !banded) ering[(size_t)it * A.NT + t] = en;
            }
        }
        PK_STAMP(4); PK_WSTAMP(8);
        PK_STAMP(5); PK_WSTAMP(10);
        // (no barrier here: P0 of the next grad-iter touches foreign position slots only, and its barrier orders the rest)
    }
    // ---- what the planner wants to know: the mean step of every own vertex over this launch
    if (A.vspeed) {
        __syncthreads();
        const float inv = 1.0f / (float)A.n_iters;
        for (int k = tid; k < 2 * w.n_own_v; k += PK_THREADS) A.vspeed[2 * (size_t)V.vid[k >> 1] + (k & 1)] = V.spd[k] * inv;
    }
    // ---- carry for the next launch on this plan (no fence: it is work on the same stream, ordered behind the end of this kernel)
    if (carry) {
        __syncthreads();
        for (int i = tid; i <= w.n_lines_all; i += PK_THREADS) {
            if (i < w.n_lines_all) { carry[8 + i] = V.tl[i]; carry[8 + A.carry_cut_cap + i] = V.nc[i]; }
            carry[8 + 2 * A.carry_cut_cap + i] = V.cut[i];
        }
#pragma unroll
        for (int i = 0; i < PK_NI; i++) {
            int32_t* e = carry + 8 + 3 * A.carry_cut_cap + 3 * (tid + i * PK_THREADS);
            e[0] = (cache[i].l & 0xffff) | (cache[i].c << 16); e[1] = cache[i].TL; e[2] = (int32_t)cache[i].magic;
        }
        if (tid == 0) {
            carry[1] = (A.n_iters + age0) & (PK_RECUT - 1); carry[2] = V.flags[2]; carry[3] = n_unc_now; carry[4] = n_unc_all_now;
            carry[0] = (int32_t)A.carry_tag;
        }
    }
    // ---- a launch that finishes itself: the last workgroup through here counts the launch as completed and tells the host (a pinned word it
    // spins on).  A workgroup that gave up never takes a ticket, so a launch that reaches the full count has completed everywhere.
    // No fences: whatever reads this launch's results is work on the same stream, ordered behind the END of the kernel (the ABI hands out no
    // device pointers to them); the word only spares the host a question to the runtime.  (A release per workgroup is a cache write-back
    // per workgroup: 8 us at the end of every launch -- more than the small kernel this replaces; one per lane: 47 us.)
    if (MODE == 0 && A.host_status) {
        __syncthreads();
        if (tid == 0 && __hip_atomic_fetch_add(status + 3, 1u, PK_RLX_AGENT) == gridDim.x - 1u) {
            __hip_atomic_store(status + 3, 0u, PK_RLX_AGENT);
            const unsigned done = __hip_atomic_load(status + 2, PK_RLX_AGENT) + 1u;   // (launches complete one after the other)
            __hip_atomic_store(status + 2, done, PK_RLX_AGENT);
            __hip_atomic_store((gu32*)A.host_status + 2, done, PK_RLX_SYSTEM);
        }
    }
}

namespace {
template <int RR>
int set_lds_rr(int bytes) {
    int rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(k_persist<RR, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (!rc) rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(k_persist<RR, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (!rc) rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(k_persist<RR, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    return rc;
}
// TPOSE_COOPERATIVE=1 (experiment, round 5): hipLaunchCooperativeKernel instead of a plain launch -- the runtime then checks that the grid can be
// resident at once and sends it through the device's cooperative queue.  Measured and not made the default: profiles/r05_experiments.txt.
bool cooperative() { static const bool on = [] { const char* e = getenv("TPOSE_COOPERATIVE"); return e && e[0] == '1'; }(); return on; }
template <typename K>
void launch_one(K kernel, const pk_args& A, dim3 g, dim3 b, size_t lds, hipStream_t s) {
    if (cooperative()) {
        pk_args copy = A;
        void* args[] = {(void*)&copy};
        (void)hipLaunchCooperativeKernel(reinterpret_cast<const void*>(kernel), g, b, args, (unsigned)lds, s);
    } else hipLaunchKernelGGL(kernel, g, b, lds, s, A);
}
template <int RR>
void launch_rr(const pk_args& A, dim3 g, dim3 b, size_t lds, hipStream_t s) {
    if (A.n_peers > 0) launch_one(k_persist<RR, 2>, A, g, b, lds, s);
    else if (A.ering || A.pring) launch_one(k_persist<RR, 1>, A, g, b, lds, s);
    else launch_one(k_persist<RR, 0>, A, g, b, lds, s);
}
}  // namespace

#ifdef PK_DBG_BOUNDS
int tp_persist_debug_faults(unsigned long long out[16]) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pk_fault), 16 * sizeof(unsigned long long)); }
#endif
#ifdef PK_DBG_STALE
int tp_persist_debug_counts(unsigned long long* out, int reset) {   // counting flavour: g_pk_cnt, [512][4]
    int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pk_cnt), 512 * 4 * sizeof(unsigned long long));
    if (!rc && reset) { static unsigned long long zero[512 * 4]; rc = (int)hipMemcpyToSymbol(HIP_SYMBOL(g_pk_cnt), zero, sizeof(zero)); }
    return rc;
}
int tp_persist_debug_vcounts(unsigned long long* out, int n, int reset) {   // g_pk_vcnt, [n][2]
    if (n > PK_DBG_VCNT) n = PK_DBG_VCNT;
    int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pk_vcnt), 2 * (size_t)n * sizeof(unsigned long long));
    if (!rc && reset) { static unsigned long long zero[2 * PK_DBG_VCNT]; rc = (int)hipMemcpyToSymbol(HIP_SYMBOL(g_pk_vcnt), zero, sizeof(zero)); }
    return rc;
}
#endif
int tp_persist_set_lds(int bytes) {
    int rc = set_lds_rr<PK_RR0>(bytes);
    if (!rc) rc = set_lds_rr<PK_RR1>(bytes);
    if (!rc) rc = set_lds_rr<PK_RR2>(bytes);
    if (!rc) rc = set_lds_rr<PK_ROWS_PER_LANE>(bytes);
    if (!rc) rc = set_lds_rr<PK_ROWS_MID>(bytes);
    if (!rc) rc = set_lds_rr<PK_ROWS_MAX>(bytes);
    if (!rc) rc = set_lds_rr<PK_ROWS_BIG>(bytes);
    return rc;
}
// rows: the most rows per lane of any patch of the plan (pk_plan::rows_max); the census passes PK_ROWS_PER_LANE
void tp_launch_persist(const pk_args& A, int grid, int rows, int lds_bytes, hipStream_t s) {
    const dim3 g((unsigned)grid), b(PK_THREADS);
    if (rows > PK_ROWS_MAX) { launch_rr<PK_ROWS_BIG>(A, g, b, (size_t)lds_bytes, s); return; }   // (eight rows per lane in LDS: the plan has made the room)
    switch (pk_rr_for(rows)) {
        case PK_RR0: launch_rr<PK_RR0>(A, g, b, (size_t)lds_bytes, s); break;
        case PK_RR1: launch_rr<PK_RR1>(A, g, b, (size_t)lds_bytes, s); break;
        case PK_RR2: launch_rr<PK_RR2>(A, g, b, (size_t)lds_bytes, s); break;
        case PK_ROWS_PER_LANE: launch_rr<PK_ROWS_PER_LANE>(A, g, b, (size_t)lds_bytes, s); break;
        case PK_ROWS_MID: launch_rr<PK_ROWS_MID>(A, g, b, (size_t)lds_bytes, s); break;
        default: launch_rr<PK_ROWS_MAX>(A, g, b, (size_t)lds_bytes, s); break;   // (rows beyond the registers in LDS: the plan has made the room)
    }
}


// band split, after a launch: every band posted the positions its vertices ended with into slot 2 + (launch number mod 2) of
// every mailbox; each band takes them all from its own.  A position that does not arrive (a band gave up) raises the status
// word: this launch, too, is run again on the two-kernel path.
__global__ void k_band_collect(tp_launch L, pk_args A, float2* points_out) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    gu32* status = (gu32*)A.status;
    if (v >= L.NP || L.vtx_off[v + 1] <= L.vtx_off[v]) return;
    if (__hip_atomic_load(status, PK_RLX_AGENT) != 0u) return;
    gu64* g = (gu64*)A.posbox + ((size_t)(2u + A.final_slot) * A.box_stride + v) * 2;
    spin_state st = {0u, 0ull};
    for (;;) {
        const unsigned long long a = __hip_atomic_load(g, PK_RLX_SYSTEM), b = __hip_atomic_load(g + 1, PK_RLX_SYSTEM);
        if ((uint32_t)(a >> 32) == A.final_tag && (uint32_t)(b >> 32) == A.final_tag) {
            __atomic_thread_fence(__ATOMIC_ACQUIRE);   // (pairs with the posters' release: their rings are visible behind this kernel)
            points_out[v] = make_float2(__uint_as_float((uint32_t)a), __uint_as_float((uint32_t)b));
            return;
        }
        if (spin_fail(st, status, PK_TIMEOUT_BANDS)) return;
    }
}
void tp_launch_band_collect(const tp_launch& L, const pk_args& A, float2* points_out, hipStream_t s) {
    hipLaunchKernelGGL(k_band_collect, dim3((unsigned)((L.NP + 63) / 64)), dim3(64), 0, s, L, A, points_out);
}
