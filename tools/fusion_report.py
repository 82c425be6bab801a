#!/usr/bin/env python3
"""Which Conv+BN+SiLU outputs of the training plan are never materialised (their readers apply the activation), and how many
elements per image that takes out of the bn_silu_fwd pass.

    python tools/fusion_report.py [--batch 4] [--img 640] [--nc 1] [--dtype f32]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--img", type=int, default=640)
    ap.add_argument("--nc", type=int, default=1)
    ap.add_argument("--dtype", default="f32")
    a = ap.parse_args()
    import yolo_from_scratch_amd as y
    from yolo_from_scratch_amd import graph as G
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = y.YOLO(num_classes=a.nc, img_size=a.img).to(dev).train()
    x = torch.rand(a.batch, 3, a.img, a.img, device=dev)
    tr = y.HipTrainer(model, dtype=a.dtype)
    tg = [t.to(dev) for t in y.synthetic_targets(a.batch, a.nc, a.img)]
    print("loss", tr.step(x, tg).cpu().tolist()[:4])
    plan = model._plan_for(x)
    tot = virt = 0
    for r in plan.recs:
        if not isinstance(r, G.ConvRec) or r.bn is None:
            continue
        n = r.Ho * r.Wo * r.cout
        tot += n
        virt += n if r.virtual else 0
        fam = "wino" if r.wino_f else "narrow" if r.narrow_f else "pw2" if r.fwd2 else "pw" if r.pw_f else "gemm"
        print(f"{r.cin:4d}->{r.cout:4d} k{r.k} s{r.s} @{r.Ho:3d}  {fam:6s} {'VIRTUAL' if r.virtual else '       '} {'x_fused' if r.x_fused else ''}"
              f"{' res' if r.residual is not None else ''}{' up' if r.upsample else ''}")
    print(f"normalised elements per image: {tot}, never materialised: {virt} ({100.0 * virt / tot:.1f} %)")


if __name__ == "__main__":
    main()
