"""One band of a band-split descent on the CPU (tests/test_distributed.py): the emulator of the persistent kernel replays this
rank's half of the patches; after every grad-iter the ranks all_gather the mailbox slot array they posted into (gloo) and every
rank keeps, per vertex, the granule that carries the grad-iter's tag -- what the bands' mailboxes do on the device.
argv: out_dir iters"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
from oracle import oracle as O  # noqa: E402
from util import RATE, case  # noqa: E402

out_dir, iters = sys.argv[1], int(sys.argv[2])
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
so = os.path.join(out_dir, "libtp_emul_persist_%d.so" % rank)
subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, os.path.join(HERE, "emul", "emul_persist.cpp")])
emp = C.CDLL(so)
W, H = 300, 200
img, imgB, pts, tris, ratio, colors = case(W, H, (15, 5))
exchanges = [0]


@C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_uint64), C.c_int, C.c_uint32)
def exchange(user, slot, NP, tag):
    mine = np.ctypeslib.as_array(slot, shape=(2 * NP,))
    t = torch.from_numpy(mine.astype(np.int64))
    got = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(got, t)
    for g in got:
        g = g.numpy().astype(np.uint64)
        fresh = ((g[0::2] >> np.uint64(32)) == np.uint64(tag)) & ((g[1::2] >> np.uint64(32)) == np.uint64(tag))
        mine[0::2][fresh] = g[0::2][fresh]
        mine[1::2][fresh] = g[1::2][fresh]
    exchanges[0] += 1


p = np.ascontiguousarray(pts.copy())
stats = np.zeros(16, np.int64)
rc = emp.emul_persist_band(img.ctypes.data_as(C.c_void_p), C.c_size_t(img.strides[0]), W, H, p.ctypes.data_as(C.c_void_p), p.shape[0],
                           tris.ctypes.data_as(C.c_void_p), tris.shape[0], None, 0, C.c_float(O.dp(0, tris.shape[0])), C.c_float(ratio),
                           C.c_float(RATE[0]), iters, 8, 160 * 1024, stats.ctypes.data_as(C.c_void_p), rank, world, exchange, None)
ref = O.iterate(img, pts, tris, 0, ratio, RATE[0], iters, literal=False)
used = np.zeros(pts.shape[0], bool)
used[tris[:, :3].ravel()] = True
same = bool(np.array_equal(p.view(np.uint32)[used], ref["points"].view(np.uint32)[used]))
json.dump({"rc": int(rc), "patches": int(stats[1]), "same_as_oracle": same, "exchanges": exchanges[0]}, open(os.path.join(out_dir, "band%d.json" % rank), "w"))
dist.destroy_process_group()
