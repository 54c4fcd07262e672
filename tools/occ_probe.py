import ctypes as C, os, sys
sys.path.insert(0, "/root/repo")
import torch, dctz_amd, numpy as np
from tests import workloads as W
ctx = dctz_amd.Context(0)
x = torch.from_numpy(W.c2()).to(ctx.device)
for mode in (0, 1):
    out, info = ctx.compress(x, 1e-4, mode)
    print("mode", mode, "flags", info.flags)
