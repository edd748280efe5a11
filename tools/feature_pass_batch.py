"""Eval-mode feature pass (sk_utils.py:137-233) rate by arithmetic and batch size: plain eval forward, folded BatchNorm with
three / two pieces per operand (selavi_amd/infer32.py).  Usage: python tools/feature_pass_batch.py [batches...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from selavi_amd import infer32, model as smodel, ops

ops.set_benchmark(os.environ.get("SELAVI_BENCHMARK", "1") == "1")
dev = torch.device("cuda")
torch.manual_seed(31)
m = smodel.load_model(vid_base_arch='r2plus1d_18', aud_base_arch='resnet9', use_mlp=True, num_classes=309, pretrained=False,
                      norm_feat=False, use_max_pool=False, headcount=10).to(dev).eval()
m.return_features = True
batches = [int(a) for a in sys.argv[1:]] or [64, 128, 256]
g = torch.Generator(device=dev).manual_seed(77)
for B in batches:
    video = torch.randn(B, 3, 16, 112, 112, device=dev, generator=g)
    audio = torch.randn(B, 1, 129, 100, device=dev, generator=g)

    def rate(fn, reps=4):
        fn(); fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return B / ((time.perf_counter() - t0) / reps)
    with torch.no_grad():
        out = {"unfolded": rate(lambda: m(video, audio))}
        for name, pieces in (("folded x3", 3), ("folded x2", 2)):
            with infer32.folded_eval(m, pieces=pieces):
                out[name] = rate(lambda: m(video, audio))
    print(f"B={B}: " + "  ".join(f"{k} {v:7.0f} clips/s" for k, v in out.items()), flush=True)
    del video, audio
    torch.cuda.empty_cache()
