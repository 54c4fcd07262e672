#!/usr/bin/env python3
"""Where the time of the one-launch kernels goes (dctz_kernels_one.hip): DCTZHIP_ONE_STAMPS=1 makes the first wave of every
workgroup leave 100 MHz time stamps at the kernel's phases; this prints, per phase, the median / max over the workgroups of
the time since the EARLIEST first stamp of the launch.   python3 tools/one_stamps.py [c1|c2] [qt]"""
import ctypes as C
import os
import sys

import numpy as np

os.environ["DCTZHIP_ONE_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import dctz_amd  # noqa: E402
from tests import workloads as W  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
mode = dctz_amd.QT if "qt" in sys.argv[2:] else dctz_amd.EC
x, eb = (W.c1(), 1e-3) if cfg == "c1" else (W.c2(), 1e-4)
ctx = dctz_amd.Context(0)
lib = ctx.lib
lib.dctzhip_debug_one_stamps.restype = C.c_int
lib.dctzhip_debug_one_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
xd = torch.from_numpy(x).to(ctx.device)
n = x.size
ntiles = (n // 64 + 63) // 64
nwg = (ntiles + 3) // 4 + (1 if n % 64 else 0)
NAMES_C = ["start", "stats posted (A1)", "A1 passed", "sweep A done", "binned (B1)", "B1 passed", "flush done", "sweep B done", "B2 passed", "end"]
NAMES_D = ["start", "counted (B1)", "B1 passed", "sweep done", "B2 passed", "dequantised", "transformed", "-", "-", "end"]


def report(names, what):
    buf = np.zeros(16 * nwg, np.uint64)
    rc = lib.dctzhip_debug_one_stamps(ctx.h, buf.ctypes.data, nwg)
    assert rc == 0, rc
    st = buf.reshape(nwg, 16).astype(np.int64)
    t0 = st[:, 0].min()
    print(what, f"({nwg} workgroups; us since the first workgroup's start)")
    for k, nm in enumerate(names):
        if nm == "-":
            continue
        v = (st[:, k] - t0) / 100.0
        print(f"  {nm:22s} min {v.min():7.2f}  median {np.median(v):7.2f}  max {v.max():7.2f}")


for rep in range(3):
    out, info = ctx.compress(xd, eb, mode)
    torch.cuda.synchronize()
assert info.flags & dctz_amd.hip.INFO_ONE_LAUNCH
report(NAMES_C, f"k_compress_one {cfg} p={info.cnt / n:.3f}")
tdt = torch.float64 if x.dtype == np.float64 else torch.float32
for rep in range(3):
    r = ctx.decompress(out, info.cnt, n, tdt, eb, info.sf, mode, qtable=np.array(info.qtable[:]))
    torch.cuda.synchronize()
report(NAMES_D, f"k_decompress_one {cfg}")
