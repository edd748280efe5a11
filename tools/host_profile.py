"""Where the HOST time of one eager training step goes (cProfile over N steps; the 16-clip 16-bit step is launch-bound).
Usage: python tools/host_profile.py [B] [T] [steps] [precision]"""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from selavi_amd import model as smodel, ops, optim, train

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
prec = sys.argv[4] if len(sys.argv) > 4 else "bf16"
dev = torch.device("cuda")
hc, K = 10, 309
torch.manual_seed(31)
m = smodel.load_model(use_mlp=True, num_classes=K, norm_feat=False, headcount=hc).to(dev).train()
m.set_precision(prec)
opt = optim.SGD(m.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-5)
g = torch.Generator(device=dev).manual_seed(1)
video = torch.randn(B, 3, T, 112, 112, device=dev, generator=g)
audio = torch.randn(B, 1, 129, 100, device=dev, generator=g)
sl = torch.randint(0, K, (4096, hc), device=dev, generator=g)
sel = torch.randint(0, 4096, (B,), device=dev, generator=g)
step = lambda: train.train_step(m, opt, video, audio, sl, sel, hc)
for _ in range(5):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumulative").print_stats(35)
