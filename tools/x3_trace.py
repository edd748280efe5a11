"""Timeline of igemm3_kernel (csrc/igemm3.hpp) on the layer-1 spatial forward: s_memtime stamps of wave 0 of the first 64
workgroups (a library built with -DSLV_X3_TRACE: tools/build_variant.sh x3trace conv_x3_fwd.hip -- -DSLV_X3_TRACE).
    SELAVI_HIP_LIB=$PWD/tools/proto/libselavi_x3trace.so python tools/x3_trace.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from selavi_amd import ops, _lib

dev = torch.device("cuda")
B = 16
plan = ops.ConvPlan.get((B, 64, 16, 56, 56), 144, (1, 3, 3), (1, 1, 1), (0, 1, 1), dev)
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(B, 64, 16, 56, 56, device=dev, generator=g)
w = torch.randn(144, 64, 1, 3, 3, device=dev, generator=g) * 0.05
ss = torch.stack([torch.rand(64, device=dev, generator=g) + 0.5, torch.randn(64, device=dev, generator=g) * 0.1])
wf, _ = ops.conv_w_transform(plan, w, need_wt=False)
for _ in range(3):
    ops.conv_fwd(plan, x, w, in_ss=ss, in_relu=True, wf=wf)
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIBPATH)
buf = np.zeros((64, 160), dtype=np.uint64)
rc = lib.slv_debug_x3_trace(ctypes.c_void_p(buf.ctypes.data), ctypes.c_size_t(buf.nbytes))
assert rc == 0
t = buf.astype(np.int64)
names = ["top", "loads issued", "compute done", "barrier1", "dma issued", "store done", "barrier2"]
print("block: prologue, first stage, loop(17 chunks), last compute, epilogue, total   [cycles of s_memtime = 100 MHz? -> see ratios]")
for b in (0, 1, 8, 9, 32, 63):
    r = t[b]
    print(b, r[1] - r[0], r[2] - r[1], r[8 + 8 * 16 + 5] - r[2], r[3] - r[8 + 8 * 16 + 5], r[4] - r[3], r[4] - r[0])
seg = np.zeros((64, 17, 6))
for b in range(64):
    for c in range(17):
        base = 8 + 8 * c
        prev = t[b][base - 3] if c > 0 else t[b][2]
        seg[b, c, 0] = t[b][base] - prev
        for k in range(1, 6):
            seg[b, c, k] = t[b][base + k] - t[b][base + k - 1]
m = seg[:, 2:16].mean(axis=(0, 1))
print("mean per-chunk segments (chunks 2..15, 64 blocks):")
for k in range(6):
    print("  %-14s %8.1f" % (["loop overhead", "issue dma", "issue B loads", "compute", "split", "wait + barrier"][k], m[k]))
print("  sum %8.1f" % m.sum())
