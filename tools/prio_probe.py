"""Does a high-priority main stream (the weight-gradient / audio side streams stay at the default priority) shorten the
step?  Usage: python tools/prio_probe.py [B] [T] [steps] [precision]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from selavi_amd import model as smodel, ops, optim, train
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
prec = sys.argv[4] if len(sys.argv) > 4 else "fp32"
dev = torch.device("cuda")
print("priority range", torch.cuda.Stream.priority_range())
if prec == "fp32":
    ops.set_benchmark(True)
torch.manual_seed(31)
m = smodel.load_model(use_mlp=True, num_classes=309, norm_feat=False, headcount=10).to(dev).train()
m.set_precision(prec)
opt = optim.SGD(m.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-5)
g = torch.Generator(device=dev).manual_seed(1)
video = torch.randn(B, 3, T, 112, 112, device=dev, generator=g)
audio = torch.randn(B, 1, 129, 100, device=dev, generator=g)
sl = torch.randint(0, 309, (4096, 10), device=dev, generator=g)
sel = torch.randint(0, 4096, (B,), device=dev, generator=g)
def run(n):
    for _ in range(n):
        train.train_step(m, opt, video, audio, sl, sel, 10)
run(4)
for tag, prio in (("default", None), ("high-priority main", torch.cuda.Stream.priority_range()[1] if torch.cuda.Stream.priority_range()[1] < 0 else -1), ("default", None)):
    st = torch.cuda.Stream(priority=prio) if prio is not None else torch.cuda.current_stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        run(2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(steps)
        torch.cuda.synchronize()
        print(f"{prec} B={B} T={T} {tag}: {(time.perf_counter() - t0) / steps * 1e3:.2f} ms/step")
    torch.cuda.current_stream().wait_stream(st)
