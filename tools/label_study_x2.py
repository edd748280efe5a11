"""How many Sinkhorn-Knopp pseudo labels move when the feature pass changes its arithmetic?  Trained synthetic runs
(examples/train_synthetic.py), then on the FINAL model the SK input of every head from four trunk forwards -- the model's own
fp32 eval forward (the reference point), fp32 with BatchNorm folded into the weights (3 pieces), the two-piece form (fp32x2) and
bf16 (selavi_amd/infer16.py) -- heads in fp32 every time; labels of each against the reference point's.
Usage: python tools/label_study_x2.py [epochs] [dataset size] [seeds...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from examples import train_synthetic as ts
from selavi_amd import infer16, infer32, sk_utils
from selavi_amd.data import SyntheticAVDataset

epochs = sys.argv[1] if len(sys.argv) > 1 else "6"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
seeds = sys.argv[3:] or ["31"]
K, hc = 16, 2
tot = {}
for seed in seeds:
    base = ["--epochs", epochs, "--dataset-size", str(n), "--batch", "16", "--frames", "4", "--size", "32", "--mel", "40", "36",
            "--num-clusters", str(K), "--headcount", str(hc), "--nopts", "8", "--seed", seed]
    log, labels_run, model = ts.main(base)
    print(f"seed {seed}: trained {epochs} epochs on {n} clips: final loss {np.mean(log[-16:]):.4f}, NMI(labels, classes) {ts.main.last_nmi:.3f}", flush=True)
    args = ts.parse(base)
    args.rank, args.world_size = 0, 1
    dataset = SyntheticAVDataset(n=n, T=4, S=32, F=40, Tp=36, n_classes=K)
    model.eval()
    model.return_features = True
    eng = infer16.Engine(model)
    feats = {k: ([], []) for k in ("fp32", "fp32_folded", "fp32x2", "bf16")}
    with torch.no_grad():
        for lo in range(0, n, 64):
            items = [dataset[i] for i in range(lo, min(lo + 64, n))]
            video = torch.stack([it[0] for it in items]).cuda()
            audio = torch.stack([it[1] for it in items]).cuda()
            a, b = model(video, audio)
            feats["fp32"][0].append(a); feats["fp32"][1].append(b)
            for name, pieces in (("fp32_folded", 3), ("fp32x2", 2)):
                with infer32.folded_eval(model, pieces=pieces):
                    a, b = model(video, audio)
                feats[name][0].append(a); feats[name][1].append(b)
            a, b = eng.features(video, audio)
            feats["bf16"][0].append(a); feats["bf16"][1].append(b)
    feats = {k: (torch.cat(v[0]), torch.cat(v[1])) for k, v in feats.items()}
    ref = feats["fp32"]
    for k, (fv, fa) in feats.items():
        if k != "fp32":
            print(f"  features {k} vs fp32 (relative L2): video {float((fv - ref[0]).norm() / ref[0].norm()):.2e}, "
                  f"audio {float((fa - ref[1]).norm() / ref[1].norm()):.2e}")
    with torch.no_grad():
        for head in range(hc):
            hv, ha = getattr(model, f"mlp_v{head}"), getattr(model, f"mlp_a{head}")
            L = {}
            for k, (fv, fa) in feats.items():
                P = sk_utils.head_probabilities(hv.forward(fv), ha.forward(fa))
                _, L[k] = sk_utils.optimize_L_sk_gpu(args, P, head)
            for k in ("fp32_folded", "fp32x2", "bf16"):
                d = int((L[k] != L["fp32"]).sum())
                tot[k] = tot.get(k, 0) + d
                print(f"  head {head}: {k}: {d} of {n} labels differ from the fp32 pass's")
    model.return_features = False
print("TOTAL over", len(seeds), "runs x", hc, "heads x", n, "rows:", tot)
