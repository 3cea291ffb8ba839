#!/usr/bin/env python
"""BASELINE config 4: a batch of synthetic 4096^2 image pairs, 12 000 triangles each, embarrassingly parallel.

Pair k goes to rank k mod N (one rank per GPU; no data-path collective); on its GPU a rank runs
the two directions of a pair (A->B and B->A, warp flavour) CONCURRENTLY in two contexts -- the single-problem path is
latency-bound and two independent problems overlap to 1.6x (profiles/README.md).  The only collective is the final
gather of a small per-pair metrics record.

  python tools/run_batch.py --pairs 8 --iters 512                      (one GPU: all pairs, one after the other)
  python -m torch.distributed.run --nproc-per-node 8 tools/run_batch.py --pairs 8 --iters 512   (RCCL, one pair per GPU)
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from tpose_amd import capi, dist_util, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=8)
    ap.add_argument("--iters", type=int, default=512)
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--triangles", type=int, default=12000)
    ap.add_argument("--backend", default=None)
    ap.add_argument("--share-gpu", action="store_true", help="testing: every rank uses GPU 0")
    ap.add_argument("--split-directions", action="store_true",
                    help="one DIRECTION per GPU instead of one pair per GPU: direction d of pair k runs on rank (2k + d) mod N -- "
                         "with N = 4 and 2 pairs this is the '2 pairs x 2 directions' substitute for splitting one pair over 4 GPUs")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank, local = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    dist = device = None
    if world > 1:
        dist, rank, world, device = dist_util.init(args.backend)
    gpu = 0 if args.share_gpu else local
    W = H = args.size
    records = []
    t_all = time.perf_counter()
    busy = 0.0
    for k in range(args.pairs):
        mine = [d for d in (0, 1) if ((2 * k + d) % world == rank if args.split_directions else k % world == rank)]
        if not mine:
            continue
        A, pts, tris, he, ratio = synth.workload(W, H, args.triangles, seed=4000 + k)
        B = synth.displaced_raster(A)
        ctxs = []
        for d, (src, dst) in enumerate(((A, B), (B, A))):  # direction 0: T(A) against raster B; 1: T(B) against raster A
            if d not in mine:
                continue
            c = capi.Context(gpu, W, H)
            c.set_image(capi.IMAGE_A, src)
            c.set_image(capi.IMAGE_B, dst)
            c.upload(pts, tris, synth.mean_colors(src, pts, tris, ratio))
            ctxs.append(c)
        p = capi.default_params(capi.WARP)  # sweeps IMAGE_B = the other view
        for c in ctxs:
            c.iterate(p, 16)
        e0 = [int(c.retrieve(capi.BUF_TENERGY)[: args.triangles].astype(np.int64).sum()) for c in ctxs]
        t0 = time.perf_counter()
        for start in range(0, args.iters, 128):  # interleave the enqueues: neither stream runs dry
            n = min(128, args.iters - start)
            for c in ctxs:
                c.iterate(p, n)
        for c in ctxs:
            c.synchronize()
        dt = time.perf_counter() - t0
        busy += dt
        e1 = [int(c.retrieve(capi.BUF_TENERGY)[: args.triangles].astype(np.int64).sum()) for c in ctxs]
        moved = [float(np.abs(c.retrieve(capi.BUF_POINTS) - pts).max()) for c in ctxs]
        records.append(dict(pair=k, rank=rank, directions=mine, seconds=dt, energy_before=e0, energy_after=e1, max_vertex_shift=moved,
                            launches_given_up=[c.info(capi.INFO_PERSIST_FAILURES) for c in ctxs]))
        for c in ctxs:
            c.close()
    wall = time.perf_counter() - t_all
    if dist is not None:
        gathered = [None] * world
        dist.all_gather_object(gathered, dict(records=records, busy=busy, wall=wall))
        dist.barrier()
    else:
        gathered = [dict(records=records, busy=busy, wall=wall)]
    if rank == 0:
        recs = sorted((r for g in gathered for r in g["records"]), key=lambda r: (r["pair"], r["directions"]))
        slowest = max(g["busy"] for g in gathered)
        work = args.pairs * 2 * args.triangles * args.iters
        print(json.dumps(dict(config="%d pairs of %dx%d, %d triangles, %d warp grad-iters per direction" % (args.pairs, W, H, args.triangles, args.iters),
                              n_gpus=world, mapping="one direction per GPU" if args.split_directions else "one pair (both directions) per GPU",
                              triangles_iters_per_s=work / slowest, seconds_iterating_slowest_rank=slowest,
                              seconds_wall_incl_setup_slowest_rank=max(g["wall"] for g in gathered), pairs=recs)))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
