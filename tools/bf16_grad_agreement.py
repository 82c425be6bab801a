#!/usr/bin/env python3
"""bf16 vs fp32 HIP gradients at identical weights, as training progresses (GPU box): worst / mean cosine over the conv
weights and the loss gap after 0, 20, 40, 60, 100, 150 fp32 Adam steps on a seeded synthetic stream."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import yolo_from_scratch_amd as y
    nc, S, B = 3, 320, 4
    torch.manual_seed(0)
    m = y.YOLO(num_classes=nc, img_size=S).cuda()
    tr = y.HipTrainer(m, lr=1e-3, max_norm=10.0, dtype="f32")
    names = [n for n, p in m.named_parameters() if p.dim() == 4]

    def batch(i):
        x = torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(1000 + i)).cuda()
        return x, [t.cuda() for t in y.synthetic_targets(B, nc, S, 8, 2000 + i)]

    done = 0
    for upto in (0, 20, 40, 60, 100, 150):
        while done < upto:
            tr.step(*batch(done))
            done += 1
        x, tg = batch(999)
        res = {}
        for dtype in ("f32", "bf16"):
            m.set_compute_dtype(dtype)
            m.train()
            m.zero_grad()
            out = y.yolo_loss_multiscale(m(x), tg, m.anchors, nc)
            out[0].backward()
            res[dtype] = (float(out[0].detach()), {n: p.grad.detach().reshape(-1).double().clone() for n, p in m.named_parameters() if n in names})
        m.set_compute_dtype("f32")
        cos = {}
        for n in names:
            a, b = res["f32"][1][n], res["bf16"][1][n]
            if float(a.norm()) > 0:
                cos[n] = float(a @ b / (a.norm() * b.norm() + 1e-30))
        worst = min(cos, key=cos.get)
        srt = sorted(cos.values())
        print(f"after {upto:3d} steps: worst {cos[worst]:.4f} ({worst}), 2nd {srt[1]:.4f}, mean {sum(srt) / len(srt):.4f}; "
              f"loss bf16 {res['bf16'][0]:.5f} fp32 {res['f32'][0]:.5f} (rel {abs(res['bf16'][0] - res['f32'][0]) / res['f32'][0]:.2e})", flush=True)


if __name__ == "__main__":
    main()
