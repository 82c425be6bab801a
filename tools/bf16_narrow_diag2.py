import os, sys, copy, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import yolo_from_scratch_amd as y
nc, S, B = 3, 640, 1
torch.manual_seed(0)
ref = y.YOLO(num_classes=nc, img_size=S)
x = torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(71)).cuda()
outs = {}
for mode in ("1", "0"):
    os.environ["YH_BF16_NARROW"] = mode
    m = copy.deepcopy(ref).cuda().set_compute_dtype("bf16")
    m.train()
    preds = m(x)
    plan = m._plan_for(x)
    recs = [r for r in plan.recs if hasattr(r, "y") and getattr(r, "y", None) is not None]
    outs[mode] = ([p.detach().float().clone() for p in preds], [(r.cin, r.cout, r.k, r.s, r.y.float().clone(), r.out.buf.data.float().clone() if hasattr(r.out.buf, "data") else None) for r in recs[:8]])
for a, b in zip(outs["1"][0], outs["0"][0]):
    d = (a - b).abs()
    print("pred", tuple(a.shape), "max diff", float(d.max()), "rel", float(d.max() / b.abs().max()), "frac differing", float((d > 0).float().mean()))
for (c1, o1, k1, s1, y1, _), (c0, o0, k0, s0, y0, _) in zip(outs["1"][1], outs["0"][1]):
    d = (y1 - y0).abs()
    print("conv", c1, o1, k1, s1, "pre-BN y: differing", int((d > 0).sum()), "of", d.numel(), "max", float(d.max()))
