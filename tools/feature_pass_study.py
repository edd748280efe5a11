"""What does 16-bit arithmetic cost the pseudo labels?  The same synthetic training run (examples/train_synthetic.py:
SK schedule, cluster(), loss on the pseudo labels) three times from one seed:
  A  fp32 training, fp32 SK feature pass            (the bit-exact default)
  B  fp32 training, bf16 SK feature pass            (args.feature_pass = "bf16")
  C  bf16 training (video trunk), bf16 feature pass (--use_fp16 of the reference + the above)
and compares the pseudo labels at the end: NMI to the synthetic classes, NMI between the runs' labels, final loss.
Usage: python tools/feature_pass_study.py [epochs] [dataset size]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sklearn.metrics.cluster import normalized_mutual_info_score as nmi
from examples import train_synthetic as ts

epochs = sys.argv[1] if len(sys.argv) > 1 else "6"
n = sys.argv[2] if len(sys.argv) > 2 else "512"
seeds = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [31]
base = ["--epochs", epochs, "--dataset-size", n, "--batch", "16", "--frames", "4", "--size", "32", "--mel", "40", "36",
        "--num-clusters", "8", "--headcount", "2", "--nopts", "8"]
runs = {"A fp32 / fp32 pass": [], "B fp32 / bf16 pass": ["--feature-pass", "bf16"],
        "C bf16 / bf16 pass": ["--precision", "bf16", "--feature-pass", "bf16"]}
summary = {name: [] for name in runs}
for seed in seeds:
    out = {}
    for name, extra in runs.items():
        log, labels, model = ts.main(base + extra + ["--seed", str(seed)])
        lab = labels[:, 0].cpu().numpy()
        out[name] = (lab, float(np.mean(log[-16:])), ts.main.last_nmi)
        summary[name].append(ts.main.last_nmi)
        print(f"== seed {seed} {name}: final loss {out[name][1]:.4f}, NMI(labels, classes) {out[name][2]:.4f}, "
              f"{len(np.unique(lab))} clusters in use", flush=True)
    names = list(out)
    for i in range(len(names)):
        for j in range(i + 1, len(names)):
            a, b = out[names[i]][0], out[names[j]][0]
            print(f"NMI seed {seed} ({names[i]} , {names[j]}) = {nmi(a, b):.4f}; identical labels: {(a == b).mean():.3f}")
for name, v in summary.items():
    print(f"== mean over seeds {seeds}: {name}: NMI(labels, classes) {np.mean(v):.4f} +- {np.std(v):.4f}")
