"""Can a 16-bit feature pass give the fp32 pseudo labels?  (the "label-safe fast feature pass" question)

Train the synthetic run (examples/train_synthetic.py) for some epochs, then on the FINAL model build the Sinkhorn-Knopp
input of every head twice -- trunk forward in fp32 and in bf16 (selavi_amd/infer16.py), heads in fp32 both times -- and solve:
  * fp32 labels vs bf16 labels (what SELAVI_FEATURE_PASS=bf16 does);
  * hybrids: the rows whose top-2 margin of the SK SCORE lambda * log P + log alpha (bf16 pass, alpha = the column scaling
    of its own SK solution: that is what the label's arg max runs over) is below tau take their fp32 probabilities, i.e.
    "re-score the uncertain rows in fp32", for several tau: fraction of rows re-scored, label equality with the all-fp32 labels.
The column scaling of the SK solution couples all rows, so a re-scored row's label can still differ through the OTHER rows'
bf16 probabilities; the table shows how far that goes.   Usage: python tools/label_margin_study.py [epochs] [dataset size]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from examples import train_synthetic as ts
from selavi_amd import infer16, sk_utils
from selavi_amd.data import SyntheticAVDataset

epochs = sys.argv[1] if len(sys.argv) > 1 else "6"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
seed = sys.argv[3] if len(sys.argv) > 3 else "31"
K, hc = 16, 2
base = ["--epochs", epochs, "--dataset-size", str(n), "--batch", "16", "--frames", "4", "--size", "32", "--mel", "40", "36",
        "--num-clusters", str(K), "--headcount", str(hc), "--nopts", "8", "--seed", seed]
log, labels_run, model = ts.main(base)
print(f"trained: final loss {np.mean(log[-16:]):.4f}, NMI(labels, classes) {ts.main.last_nmi:.3f}", flush=True)
args = ts.parse(base)
args.rank, args.world_size = 0, 1
dataset = SyntheticAVDataset(n=n, T=4, S=32, F=40, Tp=36, n_classes=K)
model.eval()
model.return_features = True
eng = infer16.Engine(model)
fv32, fa32, fv16, fa16 = [], [], [], []
with torch.no_grad():
    for lo in range(0, n, 64):
        items = [dataset[i] for i in range(lo, min(lo + 64, n))]
        video = torch.stack([it[0] for it in items]).cuda()
        audio = torch.stack([it[1] for it in items]).cuda()
        a, b = model(video, audio)
        c, d = eng.features(video, audio)
        fv32.append(a); fa32.append(b); fv16.append(c); fa16.append(d)
fv32, fa32, fv16, fa16 = (torch.cat(t) for t in (fv32, fa32, fv16, fa16))
print(f"feature difference bf16 vs fp32 (relative L2): video {float((fv16 - fv32).norm() / fv32.norm()):.2e}, "
      f"audio {float((fa16 - fa32).norm() / fa32.norm()):.2e}")
taus = [0.0, 0.05, 0.2, 0.5, 1.0, 2.0, 5.0, 10.0]
with torch.no_grad():
    for head in range(hc):
        hv, ha = getattr(model, f"mlp_v{head}"), getattr(model, f"mlp_a{head}")
        P32 = sk_utils.head_probabilities(hv.forward(fv32), ha.forward(fa32))
        P16 = sk_utils.head_probabilities(hv.forward(fv16), ha.forward(fa16))
        _, L32 = sk_utils.optimize_L_sk_gpu(args, P32.clone(), head)
        _, L16 = sk_utils.optimize_L_sk_gpu(args, P16.clone(), head)
        alpha16 = sk_utils.optimize_L_sk_gpu.last_info["alpha"].reshape(1, -1)
        top2 = torch.topk(args.lamb * torch.log(P16) + torch.log(alpha16), 2, dim=1).values
        margin = top2[:, 0] - top2[:, 1]
        print(f"head {head}: bf16 pass alone: {int((L16 != L32).sum())} of {n} labels differ; SK-score top-2 margin quantiles "
              "0.1 / 1 / 10 / 50 %: " + " / ".join(f"{float(torch.quantile(margin, q)):.3f}" for q in (0.001, 0.01, 0.1, 0.5))
              + f"; margins of the differing rows: {[round(float(v), 4) for v in margin[L16 != L32][:8]]}")
        for tau in taus:
            resc = margin < tau
            Ph = torch.where(resc.unsqueeze(1), P32, P16)
            _, Lh = sk_utils.optimize_L_sk_gpu(args, Ph.clone(), head)
            eq = float((Lh == L32).float().mean())
            print(f"  tau {tau:5.2f}: {100 * float(resc.float().mean()):6.2f} % of the rows re-scored in fp32 -> "
                  f"{100 * eq:7.3f} % of the labels equal the fp32 labels ({int((Lh != L32).sum())} of {n} differ)", flush=True)
