import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from selavi_amd import model as smodel, ops, optim, train
ops.set_benchmark(True)
dev = torch.device("cuda")
m = smodel.load_model(use_mlp=True, num_classes=309, norm_feat=False, headcount=10).to(dev).train()
opt = optim.SGD(m.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-5)
g = torch.Generator(device=dev).manual_seed(1)
video = torch.randn(16, 3, 16, 112, 112, device=dev, generator=g); audio = torch.randn(16, 1, 129, 100, device=dev, generator=g)
sl = torch.randint(0, 309, (1024, 10), device=dev, generator=g); sel = torch.randint(0, 1024, (16,), device=dev, generator=g)
for _ in range(4): train.train_step(m, opt, video, audio, sl, sel, 10)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): train.train_step(m, opt, video, audio, sl, sel, 10)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.1f ms/step, total %.1f ms/step" % ((t1 - t0) * 100, (t2 - t0) * 100))
