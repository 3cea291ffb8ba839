"""Worker for tests/test_distributed.py: one rank of the two-way warp driver on CPU (gloo), with the
oracle as the compute engine (tests only -- the product engine is tpose_amd.warp_dist.HipEngine)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import oracle as O  # noqa: E402
from tpose_amd import dist_util, hostlib, warp_dist  # noqa: E402


class OracleEngine:
    def __init__(self, imgA, imgB):
        self.img = [imgA, imgB]

    def optimise(self, tri, sweep_slot, frames, check=8, tol=1e-6):
        ratio = hostlib.get_ratio()
        pts, tris, colors = tri.points, tri.triangles, tri.colors
        state, done = [np.float32(1.0)], 0
        while done < frames:
            n = min(check, frames - done)
            if n > 1:
                out = O.iterate(self.img[sweep_slot], pts, tris, O.WARP, ratio, warp_dist.RATE_WARP, n - 1, colors=colors, literal=False)
                pts = out["points"]
                warp_dist.geterr32(out["ten"][: tri.NT], state)
            out = O.iterate(self.img[sweep_slot], pts, tris, O.WARP, ratio, warp_dist.RATE_WARP, 1, colors=colors, literal=False)
            pts = out["points"]
            done += n
            if warp_dist.geterr32(out["ten"][: tri.NT], state) < tol:
                break
        tri.points = pts
        return done


def main():
    workdir, frames = sys.argv[1], int(sys.argv[2])
    dist, rank, world, device = dist_util.init("gloo")
    group = dist.new_group([0, 1])
    A = np.load(os.path.join(workdir, "A.npy"))
    B = np.load(os.path.join(workdir, "B.npy"))
    hostlib.set_ratio(float(np.float32(A.shape[1]) / np.float32(A.shape[0])))
    forward = rank == 0
    levels = warp_dist.run_pair(dist, group, OracleEngine(A, B), os.path.join(workdir, "A.tri" if forward else "B.tri"),
                                1 if forward else 0, frames, device)
    worst = dist_util.max_over_ranks(dist, 10.0 + rank, device)
    json.dump(dict(levels=levels, worst=worst, seed=dist_util.replica_seed(rank)), open(os.path.join(workdir, "out%d.json" % rank), "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
