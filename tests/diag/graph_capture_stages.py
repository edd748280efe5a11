"""Which part of the step survives HIP-graph capture?  python -X faulthandler tests/diag/graph_capture_stages.py <stage>
stages: fwd (no_grad forward), fwdbwd, step"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from selavi_amd import model as smodel, optim, train, utils       # noqa: E402

stage = sys.argv[1]
if len(sys.argv) > 2:
    os.environ["SELAVI_OVERLAP_AUDIO"] = sys.argv[2]
dev = torch.device("cuda:0")
torch.manual_seed(31)
m = smodel.load_model(use_mlp=True, num_classes=28, norm_feat=False, headcount=1).to(dev).train()
opt = optim.SGD(m.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-5)
video = torch.randn(4, 3, 8, 112, 112, device=dev)
audio = torch.randn(4, 1, 40, 100, device=dev)
labels = torch.randint(0, 28, (3328, 1), device=dev)
sel = torch.randint(0, 3328, (4,), device=dev)


def body():
    if stage == "fwd":
        with torch.no_grad():
            return m(video, audio)[0]
    if stage == "fwdbwd":
        fv, fa = m(video, audio)
        loss = 0.5 * utils.get_loss(fv, labels[sel, 0], headcount=1) + 0.5 * utils.get_loss(fa, labels[sel, 0], headcount=1)
        opt.zero_grad()
        loss.backward()
        return loss
    return train.train_step(m, opt, video, audio, labels, sel, 1)


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        body()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
print("warm-up done", flush=True)
from selavi_amd import nn as snn       # noqa: E402
snn.dropout_device_state(dev, create=True)
g = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
with torch.cuda.graph(g):
    out = body()
    train._join_package_streams()
print("captured", flush=True)
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
print(stage, "replayed ok:", float(out.flatten()[0]), flush=True)
