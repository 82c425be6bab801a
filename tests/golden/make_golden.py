#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (KhaledSharif/yolo-from-scratch, mounted
read-only at /root/reference) on seeded inputs.  Runs only in the build container; the reference
never travels to the GPU box -- only the small data fixtures written here do.

    python tests/golden/make_golden.py            # regenerates every fixture

Each fixture holds inputs and the reference's outputs (data only).  Where whole tensors would be
large (full-model weights, 640x640 activations) the fixture keeps checksums plus seeded samples.

torchvision is not installed in this image, and the reference's `predict` imports
`torchvision.ops.batched_nms` at its single call site (train.py:1232).  To record what the
REFERENCE hands to that call (its own candidate extraction, train.py:1152-1229) a recorder object
is placed in sys.modules under that name: it stores the three argument tensors and returns
"keep everything".  It performs no NMS, so no NMS result in any fixture comes from it; class-aware
NMS stays "parity unpinned" against the reference (see oracle/yolo_oracle.py header).
"""
import os
import sys
import tempfile
import types
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, "/root/reference")
sys.path.insert(0, str(ROOT))
import train as ref  # noqa: E402  (the reference)

from oracle import yolo_oracle as orc  # noqa: E402  (only for synthetic label lists = inputs)

torch.set_num_threads(8)


def npy(t):
    return t.detach().cpu().numpy()


def save(name, **arrs):
    path = HERE / f"{name}.npz"
    np.savez_compressed(path, **arrs)
    print(f"  wrote {path.name}: {path.stat().st_size / 1024:.1f} KiB, {len(arrs)} arrays")


def sample_idx(numel, k, seed):
    rng = np.random.default_rng(seed)
    return np.sort(rng.choice(numel, size=min(k, numel), replace=False)).astype(np.int64)


# ------------------------------------------------------------------------------------------------
def gen_blocks():
    """F1: micro-shape building blocks, train-mode fwd/bwd then eval-mode fwd."""
    cases = {
        "cb3x3":   (lambda: ref.ConvBlock(8, 16, 3, 1, 1), (2, 8, 12, 12)),
        "cb3x3s2": (lambda: ref.ConvBlock(8, 16, 3, 2, 1), (2, 8, 12, 12)),
        "cb1x1":   (lambda: ref.ConvBlock(8, 12, 1, 1, 0), (2, 8, 12, 12)),
        "bneck":   (lambda: ref.Bottleneck(8, 8), (2, 8, 10, 10)),
        "c3":      (lambda: ref.C3(16, 16, n=1), (2, 16, 10, 10)),
        "c3wide":  (lambda: ref.C3(24, 16, n=1), (2, 24, 8, 8)),
        "sppf":    (lambda: ref.SPPF(16, 16), (2, 16, 10, 10)),
    }
    out = {}
    for i, (name, (make, shape)) in enumerate(cases.items()):
        torch.manual_seed(100 + i)
        m = make()
        # non-trivial BN affine so gamma/beta gradients are exercised
        with torch.no_grad():
            for mod in m.modules():
                if isinstance(mod, torch.nn.BatchNorm2d):
                    mod.weight.uniform_(0.5, 1.5)
                    mod.bias.uniform_(-0.5, 0.5)
        x = torch.randn(*shape, requires_grad=True)
        for k, v in m.state_dict().items():
            out[f"{name}/init/{k}"] = npy(v).copy()
        m.train()
        y = m(x)
        w = torch.randn_like(y)
        (y * w).sum().backward()
        out[f"{name}/x"], out[f"{name}/y"], out[f"{name}/w"] = npy(x), npy(y), npy(w)
        out[f"{name}/dx"] = npy(x.grad)
        for k, p in m.named_parameters():
            out[f"{name}/grad/{k}"] = npy(p.grad)
        for k, v in m.state_dict().items():
            if "running" in k or "num_batches" in k:
                out[f"{name}/after/{k}"] = npy(v).copy()
        m.eval()
        with torch.no_grad():
            out[f"{name}/y_eval"] = npy(m(x))
    save("blocks", **out)


def gen_decode():
    """F2: decode_predictions on small grids."""
    out = {}
    for nc in (1, 3):
        for img in (640, 1280):
            torch.manual_seed(200 + nc + img)
            raw = torch.randn(2, 5, 7, 3, 5 + nc) * 2.0
            anc = torch.tensor([[30., 61.], [62., 45.], [59., 119.]])
            key = f"nc{nc}_img{img}"
            out[f"{key}/raw"], out[f"{key}/anchors"] = npy(raw), npy(anc)
            out[f"{key}/decoded"] = npy(ref.decode_predictions(raw, anc, img))
    raw = torch.randn(1, 4, 4, 3, 6)
    out["default/raw"] = npy(raw)
    out["default/anchors"] = npy(anc)
    out["default/decoded"] = npy(ref.decode_predictions(raw, anc))
    save("decode", **out)


def gen_ciou():
    """F3: ciou_loss on the reference tests' hand-made pairs and random pairs, with gradients."""
    out = {}
    hand = {
        "identical": ([[0.5, 0.5, 0.2, 0.3]], [[0.5, 0.5, 0.2, 0.3]]),
        "disjoint":  ([[0.1, 0.1, 0.1, 0.1]], [[0.9, 0.9, 0.1, 0.1]]),
        "partial":   ([[0.5, 0.5, 0.2, 0.2]], [[0.55, 0.55, 0.2, 0.2]]),
        "aspect":    ([[0.5, 0.5, 0.1, 0.4]], [[0.5, 0.5, 0.4, 0.1]]),
        "contained": ([[0.5, 0.5, 0.1, 0.1]], [[0.5, 0.5, 0.4, 0.4]]),
    }
    for k, (p, t) in hand.items():
        p = torch.tensor(p, requires_grad=True)
        t = torch.tensor(t)
        l = ref.ciou_loss(p, t)
        l.backward()
        out[f"{k}/pred"], out[f"{k}/tgt"], out[f"{k}/loss"], out[f"{k}/dpred"] = npy(p), npy(t), npy(l), npy(p.grad)
    torch.manual_seed(300)
    p = torch.cat([torch.rand(64, 2), torch.rand(64, 2) * 0.5 + 0.01], 1).requires_grad_(True)
    t = torch.cat([p.detach()[:, :2] + (torch.rand(64, 2) - 0.5) * 0.3, torch.rand(64, 2) * 0.5 + 0.01], 1)
    l = ref.ciou_loss(p, t)
    l.backward()
    out["rand/pred"], out["rand/tgt"], out["rand/loss"], out["rand/dpred"] = npy(p), npy(t), npy(l), npy(p.grad)
    save("ciou", **out)


def make_loss_case(nc, B, S, seed, n_obj=8, empty=False):
    grids = [S // 8, S // 16, S // 32]
    torch.manual_seed(seed)
    preds = [(torch.randn(B, g, g, 3, 5 + nc) * 1.5).requires_grad_(True) for g in grids]
    boxes = [[] for _ in range(B)] if empty else orc.synthetic_boxes(B, nc, S, n_obj, seed)
    targets = orc.assign_targets(boxes, S, nc)
    return preds, targets


def gen_loss():
    """F4: yolo_loss / yolo_loss_multiscale, full 640 grids, B=2, with dPred."""
    out = {}
    m = ref.YOLO(num_classes=1)
    anchors = m.anchors
    for tag, nc, empty in (("nc1", 1, False), ("nc3", 3, False), ("nc1_empty", 1, True)):
        preds, targets = make_loss_case(nc, 2, 640, 400 + nc + (7 if empty else 0), empty=empty)
        tot, b, o, c = ref.yolo_loss_multiscale(preds, targets, anchors, nc)
        tot.backward()
        out[f"{tag}/scalars"] = np.array([float(tot), float(b), float(o), float(c)], np.float64)
        per = []
        for p, t, a in zip(preds, targets, anchors):
            with torch.no_grad():
                per.append([float(v) for v in ref.yolo_loss(p, t, a, nc)])
        out[f"{tag}/per_scale"] = np.array(per, np.float64)          # (3, [total,box,obj,cls])
        for s, (p, t) in enumerate(zip(preds, targets)):
            pos = (t[..., 4] > 0.5).nonzero()
            out[f"{tag}/s{s}/pos_idx"] = npy(pos)                    # (N,4) b,i,j,a
            out[f"{tag}/s{s}/pos_tgt"] = npy(t[t[..., 4] > 0.5])     # (N,5+nc)
            g = p.grad
            out[f"{tag}/s{s}/dpred_sum"] = np.array([float(g.double().sum()), float(g.double().abs().sum())])
            idx = sample_idx(g.numel(), 512, 41 + s)
            out[f"{tag}/s{s}/sample_idx"] = idx
            out[f"{tag}/s{s}/dpred_sample"] = npy(g.reshape(-1)[idx])
            if len(pos):
                out[f"{tag}/s{s}/dpred_pos"] = npy(g[t[..., 4] > 0.5])   # grads at positive cells
        out[f"{tag}/seed"] = np.array([400 + nc + (7 if empty else 0)])
    save("loss", **out)


def _targets_for(nc, B, S, seed):
    return orc.assign_targets(orc.synthetic_boxes(B, nc, S, 8, seed), S, nc)


def gen_model(tag, nc, S, B, lr=1e-3):
    """F5 + F6: seeded full model, one training step (clip 10.0 + Adam), then a second forward."""
    out = {}
    torch.manual_seed(0)
    m = ref.YOLO(num_classes=nc, img_size=S)
    sd = m.state_dict()
    out["keys"] = np.array(list(sd.keys()))
    out["init_sum"] = np.array([float(v.double().sum()) for v in sd.values()])
    out["init_abs"] = np.array([float(v.double().abs().sum()) for v in sd.values()])
    out["numel"] = np.array([v.numel() for v in sd.values()])
    g = torch.Generator().manual_seed(123)
    x = torch.rand(B, 3, S, S, generator=g)
    targets = _targets_for(nc, B, S, 2000)
    out["x_sum"] = np.array([float(x.double().sum())])
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=lr)
    opt.zero_grad()
    preds = m(x)
    tot, b, o, c = ref.yolo_loss_multiscale(preds, targets, m.anchors, nc)
    tot.backward()
    out["scalars"] = np.array([float(tot), float(b), float(o), float(c)])
    for s, p in enumerate(preds):
        out[f"pred{s}_sum"] = np.array([float(p.double().sum()), float(p.double().abs().sum())])
        idx = sample_idx(p.numel(), 1024, 50 + s)
        out[f"pred{s}_idx"], out[f"pred{s}_sample"] = idx, npy(p.reshape(-1)[idx])
        # decoded boxes of the reference's own forward at sampled cells (decode_predictions with img_size = the model's,
        # as predict() calls it, train.py:1154): fp32 boxes are held to 1e-4 relative end to end
        dec = ref.decode_predictions(p.detach(), m.anchors[s], S).reshape(-1, 5 + nc)
        cells = sample_idx(dec.shape[0], 512, 90 + s)
        out[f"dec{s}_cells"], out[f"dec{s}_boxes"] = cells, npy(dec[cells, :4])
    names = [n for n, _ in m.named_parameters()]
    out["param_names"] = np.array(names)
    out["grad_norm"] = np.array([float(p.grad.double().norm()) for _, p in m.named_parameters()])
    gtot = torch.linalg.vector_norm(torch.stack([p.grad.norm() for p in m.parameters()]))
    out["total_grad_norm"] = np.array([float(gtot)])
    for n, p in m.named_parameters():
        idx = sample_idx(p.numel(), 64, 7)
        out[f"gsample/{n}"] = npy(p.grad.reshape(-1)[idx])
    # BN running statistics after the forward (all of them: small)
    for k, v in m.state_dict().items():
        if "running_" in k:
            out[f"bn/{k}"] = npy(v).copy()
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    torch.nn.utils.clip_grad_norm_(m.parameters(), max_norm=10.0)
    opt.step()
    out["delta_norm"] = np.array([float((p.detach() - before[n]).double().norm()) for n, p in m.named_parameters()])
    for n, p in m.named_parameters():
        idx = sample_idx(p.numel(), 64, 7)
        out[f"dsample/{n}"] = npy((p.detach() - before[n]).reshape(-1)[idx])
    # second step's loss: end-to-end check of the update
    opt.zero_grad()
    preds = m(x)
    tot2, b2, o2, c2 = ref.yolo_loss_multiscale(preds, targets, m.anchors, nc)
    out["scalars_step2"] = np.array([float(tot2), float(b2), float(o2), float(c2)])
    save(tag, **out)


def gen_nms():
    """F7: the reference's python nms() (train.py:1086) on seeded single-class box sets built so
    no pair sits exactly on the threshold; plus its known-answer cases from tests/test_inference.py."""
    out = {}
    for M in (1, 3, 64, 300, 1000):
        rng = np.random.default_rng(700 + M)
        ctr = rng.uniform(50, 590, size=(M, 2)) if M > 3 else rng.uniform(100, 140, size=(M, 2))
        wh = rng.uniform(20, 120, size=(M, 2))
        boxes = np.concatenate([ctr - wh / 2, ctr + wh / 2], 1).astype(np.float32)
        scores = rng.permutation(M).astype(np.float32) / np.float32(M) * np.float32(0.9) + np.float32(0.05)
        for thr in (0.4, 0.6):
            dets = [(float(b[0]), float(b[1]), float(b[2]), float(b[3]), float(s), 0) for b, s in zip(boxes, scores)]
            kept = ref.nms(dets, thr)
            index_of = {d: i for i, d in enumerate(dets)}
            out[f"M{M}_t{thr}/kept"] = np.array([index_of[d] for d in kept], np.int64)
        out[f"M{M}/boxes"], out[f"M{M}/scores"] = boxes, scores
    ka = [(10, 10, 50, 50, 0.9, 0), (12, 12, 52, 52, 0.8, 0), (100, 100, 150, 150, 0.85, 0)]
    out["ka3/dets"] = np.array(ka, np.float64)
    out["ka3/kept"] = np.array([ka.index(d) for d in ref.nms(ka, 0.5)], np.int64)
    ka2 = [(10, 10, 50, 50, 0.9, 0), (20, 20, 60, 60, 0.8, 0)]
    out["ka2/dets"] = np.array(ka2, np.float64)
    out["ka2/kept_0.3"] = np.array([ka2.index(d) for d in ref.nms(ka2, 0.3)], np.int64)
    out["ka2/kept_0.7"] = np.array([ka2.index(d) for d in ref.nms(ka2, 0.7)], np.int64)
    out["iou/half_shift"] = np.array([ref.compute_iou_corners((0, 0, 10, 10, 1, 0), (5, 0, 15, 10, 1, 0))])
    save("nms", **out)


class _Recorder(types.ModuleType):
    """Records the arguments the reference passes to batched_nms; performs no suppression."""
    calls = []

    @staticmethod
    def batched_nms(boxes, scores, idxs, iou_threshold):
        _Recorder.calls.append((boxes.clone(), scores.clone(), idxs.clone(), float(iou_threshold)))
        return torch.arange(len(boxes))


def gen_candidates():
    """F8: what the reference's predict() extracts from the network output (before NMS)."""
    from PIL import Image
    tv = types.ModuleType("torchvision")
    ops = _Recorder("torchvision.ops")
    tv.ops = ops
    sys.modules["torchvision"], sys.modules["torchvision.ops"] = tv, ops
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for tag, nc, size, prior, thr in (("nc1_sq", 1, (640, 640), 0.2, 0.25),
                                          ("nc3_rect", 3, (500, 375), 0.15, 0.2)):
            torch.manual_seed(800 + nc)
            m = ref.YOLO(num_classes=nc, img_size=640)
            m.initialize_detection_biases(prior=prior)
            # an untrained net in eval mode emits ~constant logits; warm the BN running statistics
            # and widen the output convs so that objectness spreads around the threshold
            m.train()
            with torch.no_grad():
                for _ in range(3):
                    m(torch.rand(2, 3, 640, 640))
                for hd in (m.head_p3, m.head_p4, m.head_p5):
                    hd[-1].weight.mul_(60.0)
            del thr  # chosen below from the output distribution so that M is a few thousand
            rng = np.random.default_rng(801 + nc)
            img = rng.integers(0, 256, size=(size[1], size[0], 3), dtype=np.uint8)
            p = os.path.join(td, f"{tag}.png")
            Image.fromarray(img).save(p)
            pil0 = ref.letterbox_resize(Image.open(p).convert("RGB"), 640)[0]
            x0 = (torch.from_numpy(np.array(pil0)).permute(2, 0, 1).float() / 255.0).unsqueeze(0)
            m.eval()
            with torch.no_grad():
                objs = torch.cat([torch.sigmoid(q[..., 4]).flatten() for q in m(x0)])
            # threshold in the widest gap between neighbouring objectness values near rank 2500, so that
            # 1-ulp differences between sigmoid implementations cannot move a cell across it
            srt = torch.sort(objs, descending=True).values.double()
            gaps = (srt[800:6000] - srt[801:6001])
            k = 800 + int(torch.argmax(gaps))
            thr = float(((srt[k] + srt[k + 1]) / 2).float())
            print(f"    threshold {thr!r}: gap {float(gaps.max()):.3e} = {float(gaps.max()) / 1.5e-8:.0f} fp32 ulps")
            assert float(gaps.max()) > 64 * 1.5e-8, "no safe threshold gap"
            _Recorder.calls.clear()
            dets = ref.predict(m, p, torch.device("cpu"), nc, conf_threshold=thr, iou_threshold=0.4)
            boxes, scores, classes, _ = _Recorder.calls[-1]
            pil, scale, pad_top, pad_left = ref.letterbox_resize(Image.open(p).convert("RGB"), 640)
            x = (torch.from_numpy(np.array(pil)).permute(2, 0, 1).float() / 255.0).unsqueeze(0)
            m.eval()
            with torch.no_grad():
                preds = m(x)
            for s, pr in enumerate(preds):
                out[f"{tag}/pred{s}"] = npy(pr)
            out[f"{tag}/meta"] = np.array([nc, 640, thr, scale, pad_top, pad_left], np.float64)
            out[f"{tag}/boxes"], out[f"{tag}/scores"], out[f"{tag}/classes"] = npy(boxes), npy(scores), npy(classes)
            out[f"{tag}/anchors"] = np.stack([npy(a) for a in m.anchors])
            print(f"    {tag}: M={len(boxes)} candidates, {len(dets)} returned")
    del sys.modules["torchvision"], sys.modules["torchvision.ops"]
    save("candidates", **out)


def gen_assign():
    """Pins the synthetic-target generator's assignment rule against YOLODataset.__getitem__."""
    from PIL import Image
    out = {}
    with tempfile.TemporaryDirectory() as td:
        idir, ldir = Path(td) / "images", Path(td) / "labels"
        idir.mkdir(), ldir.mkdir()
        for nc in (1, 3):
            boxes = orc.synthetic_boxes(3, nc, 640, 12, 900 + nc)
            for b, bl in enumerate(boxes):
                Image.fromarray(np.zeros((640, 640, 3), np.uint8)).save(idir / f"im{nc}_{b}.png")
                with open(ldir / f"im{nc}_{b}.txt", "w") as f:
                    for (c, xc, yc, w, h) in bl:
                        f.write(f"{c} {xc!r} {yc!r} {w!r} {h!r}\n")
            ds = ref.YOLODataset(str(idir), num_classes=nc, img_size=640)
            files = [Path(p).name for p in ds.imgs]
            for b in range(3):
                _, tg = ds[files.index(f"im{nc}_{b}.png")]
                for s in range(3):
                    pos = (tg[s][..., 4] > 0.5)
                    out[f"nc{nc}/b{b}/s{s}/idx"] = npy(pos.nonzero())
                    out[f"nc{nc}/b{b}/s{s}/val"] = npy(tg[s][pos])
            out[f"nc{nc}/labels"] = np.array([[list(t) for t in bl] for bl in boxes], np.float64)
    save("assign", **out)


def gen_evalcounts():
    """f-2: the grid-cell TP/FP/FN counting of the reference's eval_epoch (train.py:990-1024), run by the
    reference itself on recorded predictions (a stub module returns them; the loop is the reference's)."""
    out = {}
    for tag, nc, S, seed in (("nc1", 1, 320, 31), ("nc3", 3, 256, 32)):
        torch.manual_seed(seed)
        grids = [S // 8, S // 16, S // 32]
        targets = orc.assign_targets(orc.synthetic_boxes(2, nc, S, 10, seed), S, nc)
        preds = []
        for g_, t in zip(grids, targets):
            p = torch.randn(2, g_, g_, 3, 5 + nc) * 1.5
            # make a good share of the positive cells confident and roughly aligned so that TP/FP/FN all occur
            pos = t[..., 4] > 0.5
            p[..., 4][pos] = torch.randn(int(pos.sum())) * 2.0 + 1.0
            preds.append(p)

        class Stub(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.anchors = ref.YOLO(num_classes=nc, img_size=S).anchors
                self.grid_size_p3, self.grid_size_p4, self.grid_size_p5 = grids

            def forward(self, x):
                return [p.clone() for p in preds]

        loader = [(torch.zeros(2, 3, S, S), [[t[b] for t in targets] for b in range(2)])]
        for conf, iou in ((0.5, 0.5), (0.3, 0.2)):
            loss, p_, r_, f1 = ref.eval_epoch(Stub(), loader, torch.device("cpu"), nc, iou_threshold=iou, conf_threshold=conf)
            out[f"{tag}/c{conf}_i{iou}"] = np.array([loss, p_, r_, f1], np.float64)
        for s_, (p, t) in enumerate(zip(preds, targets)):
            out[f"{tag}/pred{s_}"] = npy(p)
            out[f"{tag}/pos_idx{s_}"] = npy((t[..., 4] > 0.5).nonzero())
            out[f"{tag}/pos_val{s_}"] = npy(t[t[..., 4] > 0.5])
        out[f"{tag}/meta"] = np.array([nc, S])
    save("evalcounts", **out)


if __name__ == "__main__":
    only = sys.argv[1:]
    jobs = {
        "blocks": gen_blocks, "decode": gen_decode, "ciou": gen_ciou, "loss": gen_loss,
        "nms": gen_nms, "candidates": gen_candidates, "assign": gen_assign, "evalcounts": gen_evalcounts,
        "model_nc1": lambda: gen_model("model_nc1", 1, 640, 2),
        "model_nc3": lambda: gen_model("model_nc3", 3, 320, 2),
    }
    for name, fn in jobs.items():
        if only and name not in only:
            continue
        print(f"[{name}]")
        fn()
