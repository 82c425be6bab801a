#!/usr/bin/env python3
"""Which bf16 rounding site costs the weight gradients their agreement with fp32?  CPU only (oracle's storage emulation):
gradients with every site rounded, with each site alone, and with all but one, as cosines against the fp32 gradients per
layer group (backbone / neck / heads).  Used for VERDICT r2 item 7; test infrastructure, not product."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import yolo_oracle as orc          # noqa: E402
import yolo_from_scratch_amd as y               # noqa: E402


def grads(P0, names, x, tg, nc, storage):
    P = {k: v.clone() for k, v in P0.items()}
    for n in names:
        P[n].requires_grad_(True)
    out = orc.loss_multiscale(orc.forward(P, x, nc, True, storage=storage), tg, orc.anchors_of(P), nc)
    out[0].backward()
    return float(out[0]), {n: P[n].grad.clone() for n in names}


def main():
    nc, S, B = int(os.environ.get("NC", 3)), int(os.environ.get("S", 320)), int(os.environ.get("B", 4))
    torch.manual_seed(0)
    m = y.YOLO(num_classes=nc, img_size=S)
    P0 = {k: v.clone() for k, v in m.state_dict().items()}
    names = [n for n, p in m.named_parameters() if n.endswith("conv.weight") or (n.endswith(".weight") and p.dim() == 4)]
    x = torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(5))
    tg = y.synthetic_targets(B, nc, S, 8, 77)
    l0, g0 = grads(P0, names, x, tg, nc, "f32")
    groups = {"stem+backbone": lambda n: n.startswith(("stem", "backbone", "sppf")),
              "neck": lambda n: n.startswith(("lateral", "reduce", "merge", "downsample", "panet")),
              "heads": lambda n: n.startswith("head")}

    def report(tag, storage):
        l, g = grads(P0, names, x, tg, nc, storage)
        cos = {n: float(torch.nn.functional.cosine_similarity(g[n].flatten(), g0[n].flatten(), dim=0)) for n in names}
        line = f"{tag:28s} loss {l:.5f} ({l - l0:+.5f})"
        for gname, f in groups.items():
            v = [cos[n] for n in names if f(n)]
            line += f"  {gname} min {min(v):.4f} mean {sum(v) / len(v):.4f}"
        print(line, flush=True)
        return cos

    sites = orc._Net.SITES
    report("all sites (bf16)", "bf16")
    for s in sites:
        report(f"only {s}", [s])
    for s in sites:
        report(f"all but {s}", [t for t in sites if t != s])
    worst = report("all sites (bf16) per tensor", "bf16")
    for n, c in sorted(worst.items(), key=lambda kv: kv[1])[:12]:
        print(f"    {c:.4f}  {n}")


if __name__ == "__main__":
    main()
