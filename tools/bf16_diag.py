"""Per-tensor agreement of the bf16 training path with the fp32 path on the same inputs (cosine, norm ratio)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import yolo_from_scratch_amd as y

nc, S, B = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (80, 320, 2)))
x = torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(71)).cuda()
tg = [t.cuda() for t in y.synthetic_targets(B, nc, S, 8, 72)]
res = {}
for dt in ("f32", "bf16"):
    torch.manual_seed(0)
    m = y.YOLO(num_classes=nc, img_size=S).cuda()
    tr = y.HipTrainer(m, lr=1e-3, max_norm=None, dtype=dt)
    loss = tr.step(x, tg)[:4].cpu()
    # max_norm None: flat_g holds the raw gradient
    res[dt] = (loss, {n: p.grad.detach().clone().double().reshape(-1) for n, p in m.named_parameters()})
print("loss f32", res["f32"][0].tolist(), "bf16", res["bf16"][0].tolist())
tot = torch.sqrt(sum((g ** 2).sum() for g in res["f32"][1].values()))
for n, g in res["f32"][1].items():
    h = res["bf16"][1][n]
    cos = float(g @ h / (g.norm() * h.norm() + 1e-30))
    print(f"{n:50s} n={g.numel():7d} |g|/tot={float(g.norm()/tot):.2e} cos={cos:.4f} ratio={float(h.norm()/(g.norm()+1e-30)):.3f}")
