#!/usr/bin/env python
"""bench.py -- headline benchmark of the t-pose hot path on MI355X.

Metric (BASELINE.json): triangles*grad-iters / second at a 2048x2048 RGBA8 raster, 3000 triangles,
plus the achieved fraction of the HBM roofline of the dominant kernel (k_persist: K grad-iters per launch, one workgroup
per patch of the mesh; the line sums read the per-image row prefix table).

  python bench.py --gpus N --steps K --warmup W

A "step" is one grad-iter: the pixel moments of the 13 variants of every triangle -> energy -> gradient ->
shift, on inputs already resident in HBM, with no host round trip inside the
timed region.  N > 1 (launched by torch.distributed.run, one rank per GPU) runs independent
replicas -- one image + triangulation per GPU, no data-path collective -- and
reports the whole-job aggregate ("weak" scaling).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

W = H = 2048
NT = 3000
DOMINANT = "k_persist"  # the kernel the roofline figure is about: K grad-iters per launch
CONTRAST = 0.1          # contrast of the synthetic raster of rounds 3-5 (tpose_amd/synth.py: workload): now a figure BESIDE the headline
# Round 6: the headline workload is the reference's own picture -- resource/meninas.png (BASELINE config 2's), decoded into tests/golden/photos/ and
# resampled to the metric's 2048 x 2048 by an integer resampler (tpose_amd/photos.py) -- under the same 3000-triangle jittered grid.  The round-5
# review asked for exactly that unless the kernel's speed stopped depending on how fast the mesh moves; it still does (ms_per_step_by_contrast).
PHOTO = "meninas"
PHOTOS_BESIDE = ("fruit", "imageA", "shoeA")   # the other pictures the configs name, same flags (ms_per_step_on_reference_photos)
CHILD_ITERS = 256       # grad-iters per k_persist launch in the profiler passes
LONG_RUN = 131072       # grad-iters of the ageing figure (ms_per_step_long_run)
PAIR_SPLIT_DEADLINE_S = 150   # the one-pair-on-all-GPUs figure runs under this deadline (bench.py --gpus N)
CHILD_LAUNCHES = 16     # ... and launches per pass: the first 4096 grad-iters of the descent (the span the default warm-up and first timed regions cover)
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes(W, H, NT, NP):
    """one read of the RGBA8 plane + triangle indices + per-variant outputs
    in reference layout (ca 16 B + cn 4 B + ten 4 B) + points r/w and gradient."""
    return 4 * W * H + 16 * NT + 24 * 13 * NT + 24 * NP


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(img, pts, tris, ratio, budget_s=14.0):
    """The oracle in reference form (13 variants x 2 passes, per-fragment loops) on the host cores: single thread and
    OpenMP over variants with the fastest thread count.  Baseline only; this is the one place bench.py touches oracle/."""
    from oracle import oracle as O
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)

    def one(n):
        t = time.perf_counter()
        out = O.iterate(img, pts, tris, O.TRIANGULATE, ratio, 0.00005, 1, literal=True, nthreads=n)
        return time.perf_counter() - t, out["points"]

    one(1)  # warm-up (page the raster in)
    t1 = min(one(1)[0] for _ in range(2))
    # the affinity mask can exceed what the container may really use: take the fastest thread count
    best, cores = t1, 1
    for n in sorted({avail, 128, 64, 32, 16, 8}, reverse=True):
        if n > avail or n == 1:
            continue
        one(n)
        t = one(n)[0]
        if t < best:
            best, cores = t, n
    iters, t0, p = 0, time.perf_counter(), pts
    while True:
        dt_, p = one(cores)
        iters += 1
        dt = time.perf_counter() - t0
        if (dt >= budget_s - 3.0 * t1 and iters >= 3) or iters >= 5000:
            break
    return {
        "value": NT * iters / dt, "unit": "triangles*grad-iters/s", "cores": cores, "kind": "port",
        "single_thread_value": NT / t1, "cpu_model": cpu_model(), "cores_available": avail,
        "sample": "%d grad-iters of the same 2048x2048 / 3000-triangle workload, oracle/tp_oracle.c literal two-pass "
                  "form, OpenMP over variants with the fastest of {1..%d} threads (%.1f s); single thread: best of 2 "
                  "grad-iters" % (iters, avail, dt),
    }


def live_pmc_traffic(timeout_s=150):
    """HBM-side bytes per k_persist launch (CHILD_ITERS grad-iters), collected NOW: two separate rocprofv3 --pmc passes
    (FETCH_SIZE, WRITE_SIZE; counters only, no trace domains) over a child run of this script (the same workload),
    corrected as the MI355X guide prescribes for gfx950 (2 x FETCH_SIZE + WRITE_SIZE, units KB).
    Returns (bytes, note) or (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    per_launch = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="tpose_pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
        try:
            cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "pmc", "--", sys.executable,
                   os.path.abspath(__file__), "--pmc-child"]
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, cwd=d)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, "rocprofv3 --pmc %s failed (rc %d)" % (counter, r.returncode)
            vals = []
            for row in csv.DictReader(open(files[0])):
                # (the persistent kernel is a template: "void k_persist<12>(pk_args)"; k_persist_finish is another kernel)
                kname = row.get("Kernel_Name", "").split("(")[0].replace("void ", "").split("<")[0].strip()
                if row.get("Counter_Name") == counter and kname == DOMINANT:
                    vals.append(float(row["Counter_Value"]))
            if not vals:
                return None, "no %s rows in the %s pass" % (DOMINANT, counter)
            if len(vals) > 2:   # the census of resident workgroups and tp_prepare's 8-grad-iter probe: the same kernel, (next to) no table traffic
                med = sorted(vals)[len(vals) // 2]
                vals = [v for v in vals if v >= 0.25 * med]
            per_launch[counter] = sum(vals) / len(vals)
        except Exception as e:  # noqa: BLE001 -- measurement is best effort, the bench line must still appear
            return None, "%s pass: %s" % (counter, e)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return int((2.0 * per_launch["FETCH_SIZE"] + per_launch["WRITE_SIZE"]) * 1024), \
        "live: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, %d launches of %d grad-iters each), 2 x FETCH + WRITE, KB" % (CHILD_LAUNCHES, CHILD_ITERS)


def live_kernel_trace(timeout_s=150):
    """Average duration of the kernels as rocprofv3 sees them: `rocprofv3 --kernel-trace --stats` over a child run of this
    script (CHILD_LAUNCHES launches of CHILD_ITERS grad-iters of the same workload).  Returns ({kernel: avg_us}, note) or (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    d = tempfile.mkdtemp(prefix="tpose_kt_", dir=os.environ.get("TMPDIR", "/tmp"))
    try:
        cmd = [exe, "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "-o", "kt", "--", sys.executable,
               os.path.abspath(__file__), "--trace-child"]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, cwd=d)
        files = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
        traces = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
        if r.returncode != 0 or not files:
            return None, "rocprofv3 --kernel-trace failed (rc %d)" % r.returncode
        out, acc = {}, {}
        for row in csv.DictReader(open(files[0])):
            # (the persistent kernel is a template over the rows a lane keeps: k_persist<11>, k_persist<12> ... are one kernel here)
            name = row["Name"].split("(")[0].replace("void ", "").split("<")[0]
            if name.startswith("k_"):
                a = acc.setdefault(name, {"ns": 0.0, "calls": 0})
                a["ns"] += float(row["AverageNs"]) * int(row["Calls"])
                a["calls"] += int(row["Calls"])
        for name, a in acc.items():
            out[name] = {"avg_us": a["ns"] / a["calls"] / 1e3, "calls": a["calls"]}
        # the dominant kernel dispatch by dispatch (the same pass's kernel trace): the census of resident workgroups and tp_prepare's probe of the
        # vertices' speeds are launches of the same kernel, a few microseconds / 8 grad-iters long -- only the CHILD_ITERS-grad-iter launches count
        if traces:
            durs = []
            for row in csv.DictReader(open(traces[0])):
                kname = row.get("Kernel_Name", "").split("(")[0].replace("void ", "").split("<")[0].strip()
                if kname == DOMINANT:
                    durs.append((float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) / 1e3)
            if durs:
                long_ = [x for x in durs if x >= 0.5 * sorted(durs)[len(durs) // 2]]
                out[DOMINANT] = {"avg_us": sum(long_) / len(long_), "calls": len(long_), "dispatches_us": [round(x, 1) for x in durs],
                                 "left_out": "the census of resident workgroups and tp_prepare's 8-grad-iter probe (the same kernel)"}
            keep = os.environ.get("TPOSE_BENCH_KEEP_TRACE")
            if keep:   # (tools/collect_profiles.sh: the rows the figure is made of, for profiles/)
                os.makedirs(keep, exist_ok=True)
                with open(traces[0]) as fi, open(os.path.join(keep, "kernel_trace_%s_%d.csv" % (DOMINANT, CHILD_ITERS)), "w") as fo:
                    for k, line_ in enumerate(fi):
                        if k == 0 or DOMINANT in line_:
                            fo.write(line_)
                shutil.copy(files[0], os.path.join(keep, "kernel_stats_%d.csv" % CHILD_ITERS))
        return out, "live: rocprofv3 --kernel-trace --stats over %d launches of %d grad-iters (child run)" % (CHILD_LAUNCHES, CHILD_ITERS)
    except Exception as e:  # noqa: BLE001 -- measurement is best effort, the bench line must still appear
        return None, "kernel trace: %s" % e
    finally:
        shutil.rmtree(d, ignore_errors=True)


def cold_cache_figure(capi, synth, device, params, pts, tris, flavour, planes=5, steps_per_call=64, rounds=6):
    try:
        ctxs = []
        for k in range(planes):
            img_k = synth.workload(W, H, NT, seed=4321 + k, contrast=CONTRAST)[0]
            c = capi.Context(device, W, H)
            c.set_image(capi.IMAGE_A, img_k)
            colors = None
            if flavour == 1:
                c.set_image(capi.IMAGE_B, synth.displaced_raster(img_k))
                colors = synth.mean_colors(img_k, pts, tris, float(W) / float(H))
            c.upload(pts, tris, colors)
            c.prepare(params)
            c.iterate(params, 2048)   # (past the first, fast phase of the descent, like the context of the timed region)
            c.synchronize()
            ctxs.append(c)

        def loop(cs):
            t0 = time.perf_counter()
            for _ in range(rounds):
                for c in cs:
                    c.iterate(params, steps_per_call)
                    c.synchronize()
            return (time.perf_counter() - t0) / (rounds * len(cs) * steps_per_call) * 1e3

        loop(ctxs[:1] * planes)
        warm = loop(ctxs[:1] * planes)
        loop(ctxs)
        cold_ms = loop(ctxs)
        per_ctx = 4 * W * H + (W // 4 + 4) * H * 32 + (W + 8) * H * 16
        for c in ctxs:
            c.close()
        return {"ms_per_step": cold_ms, "ms_per_step_same_loop_one_context": warm, "contexts": planes,
                "steps_per_call": steps_per_call, "raster_and_table_bytes_cycled": planes * per_ctx,
                "note": "calls of %d grad-iters on %d contexts in turn (more tables than the 256 MB Infinity Cache holds) against "
                        "the same calls on one context; host-timed, one launch and one wait per call" % (steps_per_call, planes)}
    except Exception as e:  # noqa: BLE001 -- an extra figure: the bench line must still appear
        return {"error": str(e)}


def pair_split_figure(capi, synth, dist_util, dist, rank, world, device, local_rank, share_gpu, steps, warmup, repeats):
    """ONE image pair on all the GPUs of the job -- its directions (A->B, B->A: two of them from four GPUs on) each
    split into bands of patches, one band per GPU (tp_band_attach: vertex positions cross between the bands' kernels through fine-grained
    mailboxes mapped into every band's process; the unit being split is software/warp/main.cpp:214-283).  Every rank of the job calls
    this; every step that can fail is followed by an agreement, so that no rank waits in a collective another one never reaches.
    Returns a dict for rank 0's line (or {"error": ...})."""
    import torch

    def agree(ok):
        t = torch.tensor([0 if ok else 1], dtype=torch.int32, device=device)
        dist.all_reduce(t)
        return int(t.item()) == 0

    D = 2 if world >= 4 else 1          # directions of the pair that run side by side
    B = world // D                      # bands per direction
    if B < 2 or B > 4 or B * D != world:
        return {"error": "needs 2, 4 or 8 ranks"}
    my_dir, my_band = rank // B, rank % B
    err, ctx, box, mates = None, None, None, []
    try:
        imgA, pts, tris, he, ratio = synth.workload(W, H, NT, contrast=CONTRAST)
        imgB = synth.displaced_raster(imgA)
        sweep, other = (imgB, imgA) if my_dir == 0 else (imgA, imgB)   # direction 0: T(A) against image B (warp/shader/triangle.fs:49-50)
        colors = synth.mean_colors(other, pts, tris, ratio)
        slot = capi.IMAGE_B if my_dir == 0 else capi.IMAGE_A
        ctx = capi.Context(local_rank, W, H)
        ctx.set_image(slot, sweep)
        ctx.upload(pts, tris, colors)
        params = capi.default_params(capi.WARP)
        params.image_slot = slot
        cap_p, cap_t = pts.shape[0] + 64, tris.shape[0] + 64
        nbytes = capi.band_mailbox_bytes(cap_p, cap_t)
        box = ctx.band_mailbox_alloc(nbytes)
        handle = ctx.band_mailbox_export(box)
    except Exception as e:  # noqa: BLE001
        err, handle = "set-up: %s" % e, bytes(64)
    hs = [torch.zeros(64, dtype=torch.uint8, device=device) for _ in range(world)]
    dist.all_gather(hs, torch.tensor(list(handle), dtype=torch.uint8, device=device))
    if not agree(err is None):
        return {"error": err or "another rank failed in set-up"}
    try:
        boxes = []
        for b in range(B):
            if b == my_band:
                boxes.append(box)
            else:
                m = ctx.band_mailbox_import(bytes(hs[my_dir * B + b].cpu().numpy().tobytes()))
                mates.append(m)
                boxes.append(m)
        # (a plan of 512 patches between the bands of a direction -- 256 when the ranks share one GPU, where more cannot be resident)
        ppb = (256 if share_gpu else 512) // B
        ctx.band_attach(my_band, B, boxes, nbytes, cap_p, cap_t, ppb)
    except Exception as e:  # noqa: BLE001
        err = "attach: %s" % e
    if not agree(err is None):
        return {"error": err or "another rank failed to attach"}
    for r in range(world):   # (the census of resident workgroups wants its device to itself when ranks share one)
        if r == rank:
            try:
                ctx.prepare(params)
                ctx.synchronize()
            except Exception as e:  # noqa: BLE001
                err = "prepare: %s" % e
        dist.barrier()
    if not agree(err is None):
        return {"error": err or "another rank failed to prepare"}
    times = []
    info = {"persist_iters": 0, "gave_up": 0, "patches": 0}

    def guarded(fn):   # (a rank whose call fails still takes part in every collective of the loop below)
        nonlocal err
        if err is None:
            try:
                fn()
            except Exception as e:  # noqa: BLE001
                err = "iterate: %s" % e

    def step(n):
        ctx.iterate(params, n)
        ctx.synchronize()

    guarded(lambda: step(warmup))
    for rep in range(repeats):
        dist.barrier()
        t0 = time.perf_counter()
        guarded(lambda: step(steps))
        times.append(dist_util.max_over_ranks(dist, time.perf_counter() - t0, device))
    guarded(lambda: info.update(persist_iters=ctx.info(capi.INFO_PERSIST_ITERS), gave_up=ctx.info(capi.INFO_PERSIST_FAILURES), patches=ctx.info(capi.INFO_PATCHES)))
    if not agree(err is None):
        return {"error": err or "another rank failed while iterating"}
    gave_up = torch.tensor([info["gave_up"]], dtype=torch.int64, device=device)
    dist.all_reduce(gave_up)
    fine = torch.tensor([1 if ctx.info(capi.INFO_BOX_FINEGRAINED) == 1 else 0], dtype=torch.int64, device=device)
    dist.all_reduce(fine)   # (ranks whose band mailbox is fine-grained memory: all of them, or bands on different devices cannot see each other's posts)
    # the same pair on ONE GPU: its directions one after the other, unsplit (rank 0, the others wait)
    one = None
    if rank == 0:
        try:
            c1 = capi.Context(local_rank, W, H)
            c1.set_image(capi.IMAGE_B, imgB)
            c1.set_image(capi.IMAGE_A, imgA)
            c1.upload(pts, tris, colors)
            c1.prepare(params)
            c1.iterate(params, warmup)
            c1.synchronize()
            ts = []
            for rep in range(repeats):
                t0 = time.perf_counter()
                for d in range(D):   # (the second direction costs what the first does: the same mesh against the other raster)
                    c1.iterate(params, steps)
                c1.synchronize()
                ts.append(time.perf_counter() - t0)
            one = sorted(ts)[len(ts) // 2]
            c1.close()
        except Exception as e:  # noqa: BLE001
            one = None
            err = "one-GPU leg: %s" % e
    dist.barrier()
    dt = sorted(times)[len(times) // 2]
    out = {"directions": D, "bands_per_direction": B, "patches_per_band": ppb, "steps": steps,
           "ms_per_step": dt / steps * 1e3, "value": NT * D * steps / dt, "unit": "triangles*grad-iters/s of ONE pair (warp flavour, %d direction%s)" % (D, "s" if D > 1 else ""),
           "launches_given_up_all_ranks": int(gave_up.item()), "mailbox_fine_grained": int(fine.item()) == world, "rank0": info,
           "note": "one image pair on %d GPUs: %d direction(s) x %d bands of patches, positions handed over between the bands' persistent kernels "
                   "through fine-grained mailboxes (tp_band_attach); a band that waits a second for positions gives up and every band runs the "
                   "call whole -- launches_given_up counts that" % (world, D, B)}
    if one:
        out["same_pair_on_one_gpu_ms_per_step"] = one / steps * 1e3
        out["speedup_vs_one_gpu"] = one / dt
    if err:
        out["warning"] = err
    try:
        for m in mates:
            ctx.band_mailbox_close(m)
        ctx.close()
    except Exception:  # noqa: BLE001
        pass
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2048)
    ap.add_argument("--warmup", type=int, default=256)
    ap.add_argument("--repeats", type=int, default=5, help="timed regions of --steps steps each; the median is reported")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="do not spawn the two rocprofv3 --pmc passes (traffic from profiles/)")
    ap.add_argument("--no-cold", action="store_true", help="skip the cold-cache figure (five contexts cycled)")
    ap.add_argument("--no-extra", action="store_true", help="skip the full-contrast and all-13-variants figures")
    ap.add_argument("--no-pair-split", action="store_true", help="N > 1: skip the one-pair-on-all-GPUs figure (band split, SURVEY row e3)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--trace-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--flavour", type=int, default=0, help="0 triangulate (metric), 1 warp")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="testing only: every rank uses GPU 0 (with --backend gloo on a 1-GPU box)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
            # not under torchrun: start one rank per GPU ourselves and relay rank 0's line
            import subprocess
            port = 29500 + (os.getpid() % 2000)
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
                   "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
            sys.exit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")))
        args.gpus = world

    import torch
    from tpose_amd import capi, dist_util, synth
    dist, device = None, None
    if world > 1:
        dist, rank, world, device = dist_util.init(args.backend)  # RCCL; one rank per GPU
    if args.share_gpu:
        local_rank = 0

    # independent replica per rank: its own context, raster tables and triangulation (its own jitter of the grid) on the headline picture -- the work of a
    # GPU is the same at every N (weak scaling), so that the driver's efficiency compares like with like; the other pictures are timed at N = 1
    from tpose_amd import photos
    img_syn, pts, tris, he, ratio = synth.workload(W, H, NT, seed=dist_util.replica_seed(rank), contrast=CONTRAST)
    picture = PHOTO
    img = photos.resample_int(photos.load(picture), W, H)
    NP = pts.shape[0]
    ctx = capi.Context(local_rank, W, H)
    ctx.set_image(capi.IMAGE_A, img)
    imgB = None
    colors = None
    if args.flavour == 1:
        imgB = synth.displaced_raster(img)
        ctx.set_image(capi.IMAGE_B, imgB)
        colors = synth.mean_colors(img, pts, tris, ratio)
    ctx.upload(pts, tris, colors)
    params = capi.default_params(args.flavour)

    def sync_all():
        ctx.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    if args.pmc_child or args.trace_child:  # what the profiler passes sample: fused grad-iters, nothing else
        ctx.prepare(params)
        for _ in range(CHILD_LAUNCHES):
            ctx.iterate(params, CHILD_ITERS)
        ctx.synchronize()
        ctx.close()
        return
    ctx.prepare(params)  # the plan of the persistent launches (and the census of resident workgroups) is built here
    ctx.iterate(params, args.warmup)
    sync_all()
    # the timed region: exactly K steps, barrier + synchronize on both sides, max over ranks; repeated REPEATS times
    # (each repeat continues the descent from where the last one stopped), the MEDIAN is reported.
    times, dev_us = [], []
    for rep in range(args.repeats):
        sync_all()
        t0 = time.perf_counter()
        ctx.iterate(params, args.steps)
        ctx.synchronize()
        torch.cuda.synchronize()
        dt_rep = time.perf_counter() - t0
        if dist is not None:
            dt_rep = dist_util.max_over_ranks(dist, dt_rep, device)  # the job is as slow as its slowest rank
        times.append(dt_rep)
    # the same regions again with HIP events on the library's stream around them (what the device spent, without the host's
    # launch and wait) -- in regions of their own: recording the events costs the host a few microseconds per region
    for rep in range(args.repeats):
        sync_all()
        ctx.timer_start()
        ctx.iterate(params, args.steps)
        dev_us.append(ctx.timer_stop())   # (waits for the end of the region on the library's stream)
    dt = sorted(times)[len(times) // 2]
    ranks_in_collective = 1
    if dist is not None:
        dist.barrier()
        one_each = torch.ones(1, dtype=torch.int64, device=device)
        dist.all_reduce(one_each)   # (SUM over the process group -- RCCL when the driver launches this: how many ranks really took part)
        ranks_in_collective = int(one_each.item())
    # the same regions on SURVEY section 8(d)'s FULL-contrast raster (what rounds 1-2 timed; the reference's fixed-step descent throws
    # vertices hundreds of pixels on it -- profiles/r04_contrast_stats.txt: 69 x the drive of the reference's photographs), and with all 13
    # variants formed in every grad-iter (tp_iterate_until: every frame keeps its base energies for the host's convergence test, as the
    # reference's loop reads `terr` back every frame; a tp_iterate call forms the base variants in its last grad-iter only)
    full_ms, until_ms, by_contrast, full_kernel_us, long_run, syn_kernel_us, on_photos = None, None, None, None, None, None, None
    if rank == 0 and world == 1 and not args.no_extra:
        def regions(fn):
            ts = []
            for rep in range(args.repeats):
                t0 = time.perf_counter()
                fn()
                ts.append(time.perf_counter() - t0)
            return sorted(ts)[len(ts) // 2] / args.steps * 1e3
        by_contrast, full_kernel_us = {}, None
        for cval in (CONTRAST, 0.14, 0.3, 1.0):
            try:
                imgF = synth.workload(W, H, NT, seed=dist_util.replica_seed(rank), contrast=cval)[0]
                cf = capi.Context(local_rank, W, H)
                cf.set_image(capi.IMAGE_A, imgF)
                if args.flavour == 1:
                    cf.set_image(capi.IMAGE_B, synth.displaced_raster(imgF))
                cf.upload(pts, tris, colors)
                cf.prepare(params)
                cf.iterate(params, args.warmup)
                cf.synchronize()

                def one_full():
                    cf.iterate(params, args.steps)
                    cf.synchronize()
                by_contrast["%.2f" % cval] = regions(one_full)
                if cval in (1.0, CONTRAST):
                    evs = []
                    # the dominant kernel on this raster: launches of CHILD_ITERS grad-iters between HIP events -- 3 on SURVEY's raster (the median
                    # is taken), CHILD_LAUNCHES on the x0.10 raster (their MEAN: the same span of the descent the roofline's child run covers,
                    # which is how rounds 3-5 priced this raster)
                    for _ in range(3 if cval == 1.0 else CHILD_LAUNCHES):
                        cf.timer_start()
                        cf.iterate(params, CHILD_ITERS)
                        evs.append(cf.timer_stop())
                    if cval == 1.0:
                        full_ms = by_contrast["1.00"]
                        full_kernel_us = sorted(evs)[1]
                    else:
                        syn_kernel_us = sum(evs) / len(evs)
                cf.close()
            except Exception as e:  # noqa: BLE001 -- an extra figure: the bench line must still appear
                by_contrast["%.2f" % cval] = "error: %s" % e
                if cval == 1.0:
                    full_ms = "error: %s" % e
        on_photos = {}
        for name in PHOTOS_BESIDE:
            try:
                imgP = photos.resample_int(photos.load(name), W, H)
                cp = capi.Context(local_rank, W, H)
                cp.set_image(capi.IMAGE_A, imgP)
                if args.flavour == 1:
                    cp.set_image(capi.IMAGE_B, synth.displaced_raster(imgP))
                cp.upload(pts, tris, colors)
                cp.prepare(params)
                cp.iterate(params, args.warmup)
                cp.synchronize()

                def one_photo():
                    cp.iterate(params, args.steps)
                    cp.synchronize()
                on_photos[name] = regions(one_photo)
                cp.close()
            except Exception as e:  # noqa: BLE001
                on_photos[name] = "error: %s" % e
        try:
            state = {"tot": 1.0}
            cu = capi.Context(local_rank, W, H)   # (a context of its own: its plan walks every triangle's base lines in every grad-iter)
            cu.set_image(capi.IMAGE_A, img)
            if args.flavour == 1:
                cu.set_image(capi.IMAGE_B, imgB)
            cu.upload(pts, tris, colors)
            cu.prepare(params)   # (like the timed context: census, and the probe of the vertices' speeds for the first plan)

            def one_until():
                n, state["tot"], _ = cu.iterate_until(params, args.steps, 0.0, state["tot"])   # (a threshold nothing meets: exactly `steps` frames)
                assert n == args.steps
            for _ in range(max(1, args.warmup // max(1, args.steps))):
                one_until()
            until_ms = regions(one_until)
            cu.close()
        except Exception as e:  # noqa: BLE001
            until_ms = "error: %s" % e
    patches, persist_iters = ctx.info(capi.INFO_PATCHES), ctx.info(capi.INFO_PERSIST_ITERS)

    # dominant kernel: k_persist, CHILD_ITERS grad-iters per launch, HIP events on the library's stream right after the timed
    # region, same workload and state (the events also cover the small kernel that files the positions: < 3 us per launch)
    ev_samples = []
    for _ in range(5):
        ctx.timer_start()
        ctx.iterate(params, CHILD_ITERS)
        ev_samples.append(ctx.timer_stop())
    ev_samples.sort()
    kern_us_events = ev_samples[len(ev_samples) // 2]
    # the two-kernel path (k_lines + k_update per grad-iter: what every call ran through before round 3, and what calls
    # of fewer than 4 grad-iters and rasters wider than 4096 columns still use) and the reference's frame (software/triangulate/main.cpp:
    # 196-204: one grad-iter, then terr, perr, cn and the points read back whole) -- beside the fused figure, never as
    # `value`.  On a context of their own, `warmup` grad-iters into the descent like the timed regions (rounds 4-5 took them on the main
    # context after everything else: both kernels walk every row of every line, so their time follows the mesh's age -- which is what the
    # round-5 review read as a regression, 50.3 -> 58.2 us; the figure's state is now the same from round to round)
    cr = capi.Context(local_rank, W, H)
    cr.set_image(capi.IMAGE_A, img)
    if args.flavour == 1:
        cr.set_image(capi.IMAGE_B, imgB)
    cr.upload(pts, tris, colors)
    cr.set_persistent(False)
    cr.prepare(params)
    cr.iterate(params, max(args.warmup, 16))
    cr.synchronize()
    t0 = time.perf_counter()
    cr.iterate(params, 512)
    cr.synchronize()
    two_kernel_ms = (time.perf_counter() - t0) / 512 * 1e3
    cr.upload(pts, tris, colors)
    cr.iterate(params, max(args.warmup, 16))
    for _ in range(8):
        cr.iterate(params, 1)
        cr.retrieve_many([capi.BUF_TENERGY, capi.BUF_PENERGY, capi.BUF_COLNUM, capi.BUF_POINTS])
    t0 = time.perf_counter()
    for _ in range(256):
        cr.iterate(params, 1)
        cr.retrieve_many([capi.BUF_TENERGY, capi.BUF_PENERGY, capi.BUF_COLNUM, capi.BUF_POINTS])
    readback_ms = (time.perf_counter() - t0) / 256 * 1e3
    cr.close()
    # how the figure ages: LONG_RUN grad-iters more on the same context (calls of 4096), the last LONG_RUN / 4 of them timed
    if rank == 0 and world == 1 and not args.no_extra:
        try:
            ctx.prepare(params)
            done = 0
            while done < LONG_RUN * 3 // 4:
                ctx.iterate(params, 4096); done += 4096
            ctx.synchronize()
            t0 = time.perf_counter()
            n_timed = 0
            while done < LONG_RUN:
                ctx.iterate(params, 4096); done += 4096; n_timed += 4096
            ctx.synchronize()
            long_run = {"ms_per_step": (time.perf_counter() - t0) / n_timed * 1e3, "grad_iters_before": done - n_timed, "grad_iters_timed": n_timed,
                        "plans_cut_again": ctx.info(capi.INFO_REPLANS), "launches_given_up": ctx.info(capi.INFO_PERSIST_FAILURES),
                        "note": "the same context after the timed regions: calls of 4096 grad-iters until %d have run, the last quarter host-timed" % LONG_RUN}
        except Exception as e:  # noqa: BLE001
            long_run = {"error": str(e)}
    # SURVEY section 8d caveat (iii): the same calls over more tables than the 256 MB Infinity Cache holds -- five contexts
    # (five rasters: 5 x 118 MB of raster + tables), one call each in turn, against the same loop on ONE context.  What a
    # cold table costs the persistent path is the first grad-iter of a launch (every lane fetches its records) and the
    # re-fetches while the mesh moves; the grad-iters in between read registers.
    cold = None
    if rank == 0 and world == 1 and not args.no_cold:
        cold = cold_cache_figure(capi, synth, local_rank, params, pts, tris, args.flavour)
    bytes_iter = algorithmic_bytes(W, H, NT, NP)
    # the same kernel as rocprofv3 sees it (what profiles/ holds): the roofline figure uses THIS duration when it is
    # available, so that it can be reproduced from a kernel trace; the HIP-event figure stays beside it
    under_profiler = any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
    trace, trace_note = (None, "skipped")
    if rank == 0 and world == 1 and not args.no_pmc and not under_profiler:
        ctx.synchronize()
        trace, trace_note = live_kernel_trace()
    kern_us = kern_us_events
    if trace and DOMINANT in trace:
        kern_us = trace[DOMINANT]["avg_us"]
    achieved = bytes_iter * CHILD_ITERS / (kern_us * 1e-6) / 1e9
    # HBM-side traffic of the same kernel: 2 x FETCH_SIZE + WRITE_SIZE per launch (the gfx950 correction of the
    # guide), from two live counter passes over a child run
    traffic, traffic_source = None, None
    if rank == 0 and world == 1 and not args.no_pmc and not under_profiler:  # never nest profilers
        ctx.synchronize()
        traffic, traffic_source = live_pmc_traffic()

    line = None
    if rank == 0:
        line = {
            "metric": "triangles*grad-iters/sec at 2048^2/3000 tris; HBM GB/s vs roofline",
            "metric_conditions": "value: the reference's own picture %s.png resampled to 2048 x 2048 (round 6: the round-5 review's rule -- the kernel's speed still "
                                 "depends on how fast the mesh moves, so the headline is the photograph's figure, not the synthetic raster's: value_synthetic_contrast_0.10 / "
                                 "ms_per_step_by_contrast beside it), tp_iterate calls of `steps` grad-iters whose intermediate grad-iters form the 12 displaced "
                                 "variants (the base variants in the call's last one); like-for-like with the reference's loop shape: value_all_13_variants" % picture,
            "value": NT * args.steps * world / dt,
            "unit": "triangles*grad-iters/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ranks_in_collective": ranks_in_collective, "collective_backend": (args.backend if dist is not None else None),
            "ms_per_step": dt / args.steps * 1e3,
            "timing": "median of %d timed regions of %d steps; ms_per_step of each: %s" % (
                len(times), args.steps, ", ".join("%.5f" % (t / args.steps * 1e3) for t in times)),
            "ms_per_step_device": sorted(dev_us)[len(dev_us) // 2] / args.steps / 1e3,
            "ms_per_step_device_note": "HIP events on the library's stream around %d more regions of the same length (median): what "
                                       "the device spent, without the host's launch and wait" % args.repeats,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "ms_per_step_full_contrast": full_ms,
            "ms_per_step_full_contrast_note": "the same flags on SURVEY section 8(d)'s raster as written (uniform u8 site colours, no contrast scaling): "
                                              "the raster of rounds 1-2; vertices fly hundreds of pixels on it, lines outgrow the rows their lanes keep",
            "ms_per_step_on_reference_photos": dict({picture: dt / args.steps * 1e3}, **(on_photos or {})),
            "ms_per_step_on_reference_photos_note": "the same flags on the pictures BASELINE.json's configs name, each resampled to 2048 x 2048 (tests/golden/photos, tpose_amd/photos.py), same mesh",
            "value_synthetic_contrast_0.10": (NT / (by_contrast["%.2f" % CONTRAST] * 1e-3)) if by_contrast and isinstance(by_contrast.get("%.2f" % CONTRAST), float) else None,
            "roofline_frac_synthetic_contrast_0.10": (algorithmic_bytes(W, H, NT, NP) * CHILD_ITERS / (syn_kernel_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if syn_kernel_us else None,
            "value_synthetic_note": "rounds 3-5 quoted `value` on the synthetic Voronoi raster at x0.10 contrast: kept here (same flags, a context of its own; the fraction: the mean of %d "
                                    "launches of %d grad-iters between HIP events, behind the timed regions)" % (CHILD_LAUNCHES, CHILD_ITERS),
            "value_all_13_variants": (NT / (until_ms * 1e-3)) if isinstance(until_ms, float) else None,
            "roofline_frac_all_13_variants_by_bench_clock": (algorithmic_bytes(W, H, NT, NP) / (until_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if isinstance(until_ms, float) else None,
            "ms_per_step_by_contrast": by_contrast,
            "ms_per_step_by_contrast_note": "the same flags on the SYNTHETIC raster at contrasts about mid-grey (1.00 = SURVEY section 8(d)'s raster as written): the "
                                            "kernel keeps a lane's table records in registers and re-fetches a row only when its crossing column changes, so it is "
                                            "fastest when vertices move a fraction of a pixel per grad-iter; at higher contrast the fixed-step descent moves them "
                                            "further, more rows are re-fetched and lines outgrow the rows their lanes keep between two cuts",
            "roofline_frac_full_contrast": (algorithmic_bytes(W, H, NT, NP) * CHILD_ITERS / (full_kernel_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if full_kernel_us else None,
            "roofline_frac_full_contrast_note": "the dominant kernel on SURVEY section 8(d)'s raster as written: one launch of %d grad-iters between HIP events, algorithmic bytes / time / 8 TB/s" % CHILD_ITERS,
            "ms_per_step_long_run": long_run,
            "ms_per_step_all_13_variants": until_ms,
            "ms_per_step_all_13_variants_note": "tp_iterate_until with a threshold nothing meets, the same number of frames per call: base lines walked and "
                                                "base energies kept in EVERY grad-iter, the host's geterr over every frame, the last frame re-run to leave its buffers",
            "ms_per_step_two_kernel_path": two_kernel_ms,
            "ms_per_step_readback_every_iter": readback_ms,
            "dtype": "int64", "data": "reference photograph (decoded fixture of resource/%s.png), integer-resampled to the metric's 2048 x 2048; synthetic mesh" % picture,
            "config": {
                "workload": "2048x2048 RGBA8 raster = the reference's %s.png (%s) resampled by tpose_amd/photos.py: resample_int, 3000-triangle jittered grid "
                            "(50x30x2), %s flavour, one replica per GPU%s" % (picture, "BASELINE config 2's picture" if picture == "meninas" else "one of the configs' pictures",
                                                                            "warp" if args.flavour else "triangulate",
                                                                            "" if world == 1 else " (every rank the same picture, its own jitter of the grid)"),
                "raster": [W, H], "triangles": NT, "points": NP, "variants": 13 * NT,
                "parallelism": "replicas x%d (no data-path collective)" % world,
                "path": "persistent launches: %d patches (workgroups), %d grad-iters ran inside them.  Inside a tp_iterate call the intermediate "
                        "grad-iters form the 12 displaced variants of every triangle (all the step needs); the base variants (i = 0), which only "
                        "the buffers the reference reads back need, are formed in the call's last grad-iter -- ms_per_step_all_13_variants is the "
                        "mode that forms them every grad-iter" % (patches, persist_iters),
            },
            "roofline": {
                "bound": "latency", "nominal_bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "frac_note": "algorithmic bytes (SURVEY section 8d: one read of the raster plane + indices + per-variant outputs + points) / kernel "
                             "time / 8 TB/s -- the roofline the task prices this path against; the kernel itself is bound by latency (see "
                             "observed_bound), its measured HBM-side traffic is `traffic`",
                "frac_of_measured_traffic": (traffic / (kern_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                "traffic_source": traffic_source,
                "observed_bound": "latency, not bandwidth: VALU issue inside a workgroup (the walk of the edge lines) plus one "
                                  "position hand-over between workgroups per grad-iter; the table records a lane needs stay in "
                                  "its registers from one grad-iter to the next, so HBM-side traffic is far below the algorithmic bytes",
                "hbm_side_GBs": (traffic / (kern_us * 1e-6) / 1e9) if traffic else None,
                "kernel": DOMINANT, "kernel_us": kern_us, "grad_iters_per_launch": CHILD_ITERS,
                "us_per_grad_iter": kern_us / CHILD_ITERS,
                "algorithmic_bytes": bytes_iter * CHILD_ITERS, "algorithmic_bytes_per_grad_iter": bytes_iter,
                "kernel_timing": ("average duration in a rocprofv3 --kernel-trace --stats pass over a child run (%d launches of "
                                  "%d grad-iters: the first %d of the descent), collected by this command" % (CHILD_LAUNCHES, CHILD_ITERS, CHILD_LAUNCHES * CHILD_ITERS) if trace and DOMINANT in trace else
                                  "HIP events (kernel trace unavailable: %s)" % trace_note),
                "kernel_trace_us": trace,
                "kernel_us_hip_events": kern_us_events,
                "kernel_us_hip_events_note": "HIP events on the library's stream around one launch of %d grad-iters (median of 5), "
                                             "after the timed region, same state" % CHILD_ITERS,
                "kernel_us_samples": ev_samples,
            },
        }
        if cold is not None:
            line["cold_cache"] = cold
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(img if args.flavour == 0 else imgB, pts, tris, ratio)
    # N > 1: beside the replicas, ONE pair on all the GPUs (row e3) -- LAST, in a context of its own, with the headline already in hand:
    # it has only ever run with its ranks sharing one GPU, so it runs under a deadline -- if any rank is stuck in it (a collective another
    # rank never reaches, a mailbox that cannot be mapped across devices), rank 0 prints the line without the figure and every rank leaves.
    if dist is not None and world in (2, 4, 8) and not args.no_pair_split:
        import threading
        printed = threading.Lock()

        def give_up():
            if not printed.acquire(blocking=False):
                return
            if rank == 0 and line is not None:
                line["pair_split"] = {"error": "no answer within %d s: a rank is stuck in the band split; left out" % PAIR_SPLIT_DEADLINE_S}
                print(json.dumps(line), flush=True)
            os._exit(0)

        watchdog = threading.Timer(PAIR_SPLIT_DEADLINE_S, give_up)
        watchdog.daemon = True
        watchdog.start()
        try:
            pair = pair_split_figure(capi, synth, dist_util, dist, rank, world, device, local_rank, args.share_gpu, args.steps, args.warmup, args.repeats)
        except Exception as e:  # noqa: BLE001 -- an extra figure: the bench line must still appear
            pair = {"error": str(e)}
        watchdog.cancel()
        if not printed.acquire(blocking=False):
            time.sleep(60)   # (the watchdog is printing: it ends the process)
            return
        if line is not None and pair is not None:
            line["pair_split"] = pair
            # (for whoever reads a SCALE line without having seen the run: what makes the one-pair figure mean something, at the top level)
            line["pair_split_launches_given_up_all_ranks"] = pair.get("launches_given_up_all_ranks")
            line["pair_split_mailbox_fine_grained"] = pair.get("mailbox_fine_grained")
    if line is not None:
        print(json.dumps(line), flush=True)

    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
