"""Training step with the video trunk on the 16-bit MFMA path at a given shape (default cfg2's: B=16, 16 frames):
ms/step, clips/s, algorithmic TFLOP/s.  Usage: python tools/step16_bench.py [B] [T] [steps] [precision] [graph]
(run under `rocprofv3 --kernel-trace --stats` for the per-kernel breakdown)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from selavi_amd import model as smodel, ops, optim, train

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
prec = sys.argv[4] if len(sys.argv) > 4 else "bf16"
dev = torch.device("cuda")
hc, K = 10, 309
if prec == "fp32":
    ops.set_benchmark(True)
torch.manual_seed(31)
m = smodel.load_model(use_mlp=True, num_classes=K, norm_feat=False, headcount=hc).to(dev).train()
m.set_precision(prec)
opt = optim.SGD(m.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-5)
g = torch.Generator(device=dev).manual_seed(1)
video = torch.randn(B, 3, T, 112, 112, device=dev, generator=g)
audio = torch.randn(B, 1, 129, 100, device=dev, generator=g)
sl = torch.randint(0, K, (4096, hc), device=dev, generator=g)
sel = torch.randint(0, 4096, (B,), device=dev, generator=g)
graph = len(sys.argv) > 5 and sys.argv[5] == "graph"       # the whole step as one HIP graph (train.GraphedStep)
if graph:
    gs = train.GraphedStep(m, opt, video, audio, sl, sel, hc)
    step = gs.replay
else:
    step = lambda: train.train_step(m, opt, video, audio, sl, sel, hc)
for _ in range(3):
    loss = step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    loss = step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / steps * 1e3
# host time of one eager enqueue (no device wait): is the step launch-bound?
t0 = time.perf_counter()
loss = step()
host_ms = (time.perf_counter() - t0) * 1e3
torch.cuda.synchronize()
gflop = 3 * (81.04 * T / 16 + 0.506 + 0.0168) * B
with torch.no_grad():
    m(video, audio)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        m(video, audio)
    torch.cuda.synchronize()
    fwd = (time.perf_counter() - t0) / 5 * 1e3
print(f"{prec}{' graph' if graph else ''} B={B} T={T}: {ms:.2f} ms/step (host enqueue {host_ms:.2f} ms), {B / ms * 1e3:.1f} clips/s, {gflop / ms:.1f} TFLOP/s algorithmic, "
      f"train-mode forward {fwd:.2f} ms ({B / fwd * 1e3:.0f} clips/s), loss {float(loss):.4f}, "
      f"peak HBM {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
