"""Is the 16-bit audio trunk (forward + backward on its side stream, inside the whole step) bit-reproducible?  Runs the same
step from the same state N times and compares hashes of the audio parameters' gradients and BatchNorm buffers."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle.model_ref import portable_fill_, portable_init_
from selavi_amd import model as smodel, train, optim

def run(n):
    res = []
    for it in range(n):
        m = smodel.load_model(use_mlp=True, num_classes=7, norm_feat=False, headcount=2)
        portable_init_(m, seed=31)
        from oracle import step_ref
        step_ref.set_dropout_p(m, 0.0)
        m = m.cuda().train()
        m.set_precision("bf16")
        opt = optim.SGD(m.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-5)
        video = portable_fill_(torch.empty(4, 3, 4, 32, 32), 5).cuda()
        audio = portable_fill_(torch.empty(4, 1, 40, 36), 6).cuda()
        sl = (torch.arange(64 * 2).reshape(64, 2) * 7919 % 7).cuda()
        sel = torch.tensor([3, 17, 42, 63]).cuda()
        for _ in range(2):
            loss = train.train_step(m, opt, video, audio, sl, sel, 2)
        torch.cuda.synchronize()
        h = {k: hashlib.md5(v.detach().cpu().numpy().tobytes()).hexdigest() for k, v in m.state_dict().items()}
        res.append(h)
    bad = sorted({k for h in res[1:] for k in h if h[k] != res[0][k]})
    print(f"{n} runs: {len(bad)} tensors differ between runs: {bad[:6]}")

run(int(sys.argv[1]) if len(sys.argv) > 1 else 8)
