"""Two-way consistent hierarchical warp, one image (direction) per GPU, vertex buffers exchanged with
torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

Reference: software/warp/main.cpp:214-283 runs the two directions A->B and B->A one after the other on
one GPU and seeds each from the reverse warp of the other (README.md:49-53).  Here rank 2p optimises
T(A_p) against raster B_p while rank 2p+1 optimises T(B_p) against raster A_p, concurrently; once per
level and direction the ranks of a pair exchange {NT, NP, triangles, points, originpoints} -- a few
tens of KB, latency-bound -- so each can run `reversewarp` of its own origin points through the
peer's warped mesh (source/triangulation.hpp:492-520), re-seed, refine, and append its level to
<tri>.warp; the next finer level is read warped-on-read (source/io.hpp:139).  There is no collective
in the per-iteration data path: pairs are independent, and inside a pair the only exchange is this
per-level hand-over.

Launch:  python -m torch.distributed.run --nproc-per-node 2 -m tpose_amd.warp_dist \
             --ia A.ppm --ib B.ppm --ta A.tri --tb B.tri [--frames 200]
"""
import argparse
import os

import numpy as np

from . import hostlib

RATE_WARP = 0.00003  # software/warp/shader/shift.cs:45


def geterr32(terr, state):
    """tpose::geterr (source/triangulation.hpp:653-674): float32, ascending t; state = [toterr]."""
    e = terr.astype(np.float32)
    newerr = np.float32(0) if e.size == 0 else np.cumsum(e, dtype=np.float32)[-1]
    rel = (state[0] - newerr) / state[0]
    state[0] = newerr
    return abs(float(rel))


class HipEngine:
    """The HIP path behind the C ABI: one tp_context on this rank's GPU."""

    def __init__(self, device, imgA, imgB):
        from . import capi
        self.capi = capi
        H, W = imgA.shape[:2]
        self.ctx = capi.Context(device, W, H)
        self.ctx.set_image(capi.IMAGE_A, imgA)
        self.ctx.set_image(capi.IMAGE_B, imgB)

    def optimise(self, tri, sweep_slot, frames, check=8, tol=1e-6):
        """descend `tri` (stored colours) against raster `sweep_slot`; returns frames spent.  The reference tests
        the relative energy change between CONSECUTIVE frames every frame (software/warp/main.cpp:231); here the
        same test is sampled on the last two frames of every `check` fused iterations (two small readbacks)."""
        capi = self.capi
        self.ctx.set_ratio(hostlib.get_ratio())
        self.ctx.upload(tri.points, tri.triangles, tri.colors)
        params = capi.default_params(capi.WARP, image_slot=sweep_slot)
        state, done = [np.float32(1.0)], 0
        while done < frames:
            n = min(check, frames - done)
            if n > 1:
                self.ctx.iterate(params, n - 1)
                geterr32(self.ctx.retrieve(capi.BUF_TENERGY)[: tri.NT], state)
            self.ctx.iterate(params, 1)
            done += n
            if geterr32(self.ctx.retrieve(capi.BUF_TENERGY)[: tri.NT], state) < tol:
                break
        tri.points = self.ctx.retrieve(capi.BUF_POINTS)
        return done


def pack(tri):
    """{NT, NP, triangles, points, originpoints} as one int32 vector (floats bit-cast)"""
    head = np.array([tri.NT, tri.NP], np.int32)
    return np.concatenate([head, tri.triangles.ravel(), tri.points.view(np.int32).ravel(),
                           tri.originpoints.view(np.int32).ravel()])


def unpack(buf):
    NT, NP = int(buf[0]), int(buf[1])
    o = 2
    tris = buf[o:o + 4 * NT].reshape(NT, 4).copy(); o += 4 * NT
    pts = buf[o:o + 2 * NP].view(np.float32).reshape(NP, 2).copy(); o += 2 * NP
    org = buf[o:o + 2 * NP].view(np.float32).reshape(NP, 2).copy()
    t = hostlib.Triangulation()
    t.assign(tris, pts, org)
    return t


def exchange(dist, group, mine, device):
    """all-gather the (padded) packed meshes inside the pair; returns the peer's triangulation"""
    import torch
    me = pack(mine)
    size = torch.tensor([me.size], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(size) for _ in range(2)]
    dist.all_gather(sizes, size, group=group)
    n = int(max(int(s.item()) for s in sizes))
    send = torch.zeros(n, dtype=torch.int32, device=device)
    send[: me.size] = torch.from_numpy(me).to(device)
    recv = [torch.zeros_like(send) for _ in range(2)]
    dist.all_gather(recv, send, group=group)
    peer = 1 - dist.get_rank(group)
    return unpack(recv[peer].cpu().numpy()[: int(sizes[peer].item())])


def run_pair(dist, group, engine, my_tri_path, sweep_slot, frames, device, log=None):
    """the per-rank loop over hierarchy levels; returns a list of per-level dicts"""
    mine = hostlib.Triangulation()
    if not mine.read(my_tri_path):
        raise RuntimeError("empty triangulation file " + my_tri_path)
    out_path = my_tri_path + ".warp"
    if os.path.exists(out_path):
        os.remove(out_path)
    levels = []
    while True:
        f1 = engine.optimise(mine, sweep_slot, frames)
        peer = exchange(dist, group, mine, device)
        # what the peer's warp predicts for my vertices: my origin points pulled back through its mesh
        seed = peer.reversewarp(mine.originpoints)
        residual = float(np.abs(seed - mine.points).max())
        mine.points = seed
        f2 = engine.optimise(mine, sweep_slot, frames)
        mine.write(out_path)
        levels.append(dict(NT=mine.NT, NP=mine.NP, frames=(f1, f2), residual=residual))
        if log:
            log("level %d: NT=%d frames=%d+%d two-way residual=%.5f" % (len(levels) - 1, mine.NT, f1, f2, residual))
        more = mine.read(my_tri_path, dowarp=True)  # next finer level, warped by this one
        # both ranks of a pair must agree on continuing (stacks may differ in depth)
        import torch
        flag = torch.tensor([1 if more else 0], dtype=torch.int64, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if int(flag.item()) == 0:
            break
    return levels


def load_ppm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"P6"
        line = f.readline()
        while line.startswith(b"#"):
            line = f.readline()
        w, h = (int(v) for v in line.split())
        assert int(f.readline()) == 255
        rgb = np.frombuffer(f.read(w * h * 3), np.uint8).reshape(h, w, 3)
    img = np.full((h, w, 4), 255, np.uint8)
    img[:, :, :3] = rgb
    return img


def main():
    import torch
    import torch.distributed as dist
    ap = argparse.ArgumentParser()
    ap.add_argument("--ia", required=True); ap.add_argument("--ib", required=True)
    ap.add_argument("--ta", required=True); ap.add_argument("--tb", required=True)
    ap.add_argument("--frames", type=int, default=400)
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--share-gpu", action="store_true", help="testing only: all ranks use GPU 0 (with --backend gloo)")
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world % 2:
        raise SystemExit("warp_dist needs an even number of ranks (one direction per GPU)")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.backend == "nccl":
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        device = torch.device("cuda", local)
    else:
        dist.init_process_group(args.backend)
        device = torch.device("cpu")
    groups = [dist.new_group([2 * p, 2 * p + 1]) for p in range(world // 2)]
    group = groups[rank // 2]
    A, B = load_ppm(args.ia), load_ppm(args.ib)
    hostlib.set_ratio(float(np.float32(A.shape[1]) / np.float32(A.shape[0])))
    engine = HipEngine(0 if args.share_gpu else local, A, B)
    from . import capi
    forward = rank % 2 == 0  # even rank: T(A) against raster B
    levels = run_pair(dist, group, engine, args.ta if forward else args.tb,
                      capi.IMAGE_B if forward else capi.IMAGE_A, args.frames, device,
                      log=lambda s: print("[rank %d] %s" % (rank, s), flush=True))
    dist.barrier()
    if rank == 0:
        print("levels: %d" % len(levels))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
