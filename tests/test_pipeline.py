"""Input pipeline.  CPU part: the raw dataset mode carries exactly the information of the
reference-style samples (image bytes -> /255 gives the same floats; labels -> the oracle's assignment rule gives the same
dense targets).  GPU part: DevicePrefetcher hands the training step bit-identical inputs, in order, and an epoch through it
gives the same losses as the reference-style loader."""
import numpy as np
import pytest
import torch

import yolo_from_scratch_amd as y


def make_dataset(root, n=7, seed=0):
    from PIL import Image
    rng = np.random.default_rng(seed)
    (root / "images").mkdir(parents=True)
    (root / "labels").mkdir()
    for i in range(n):
        w, h = int(rng.integers(40, 200)), int(rng.integers(40, 200))
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(root / "images" / f"im{i:02d}.png")
        if i == 3:
            continue                                   # an image without a label file
        with open(root / "labels" / f"im{i:02d}.txt", "w") as fh:
            for _ in range(int(rng.integers(0, 6))):
                bw, bh = rng.uniform(0.02, 0.6, 2)
                fh.write(f"{int(rng.integers(0, 3))} {rng.uniform(bw / 2, 1 - bw / 2):.6f} {rng.uniform(bh / 2, 1 - bh / 2):.6f} "
                         f"{bw:.6f} {bh:.6f}\n")
            if i == 5:
                fh.write("garbage line\n")             # skipped by both paths
    return str(root / "images")


@pytest.mark.parametrize("nc", [1, 3])
def test_raw_dataset_carries_the_reference_samples(tmp_path, nc):
    from oracle import yolo_oracle as orc
    d = make_dataset(tmp_path)
    S = 96
    std = y.YOLODataset(d, nc, img_size=S)
    raw = y.YOLODataset(d, nc, img_size=S, raw=True)
    assert len(std) == len(raw) == 7
    for i in range(len(std)):
        img, tg = std[i]
        u8, lab = raw[i]
        assert u8.dtype == torch.uint8 and tuple(u8.shape) == (S, S, 3) and lab.dtype == torch.float64
        assert torch.equal(u8.permute(2, 0, 1).float() / 255.0, img)
        boxes = [[(int(r[0]), float(r[1]), float(r[2]), float(r[3]), float(r[4])) for r in lab.tolist()]]
        dense = orc.assign_targets(boxes, S, nc)
        for a, b in zip(dense, tg):
            assert torch.equal(a[0], b)
    imgs, lab, cnt = y.raw_collate_fn([raw[i] for i in range(7)])
    assert tuple(imgs.shape) == (7, S, S, 3) and lab.shape[0] == 7 and lab.shape[2] == 5 and cnt.dtype == torch.int32
    assert int(cnt[3]) == 0 and lab.shape[1] == max(1, int(cnt.max()))
    for i in range(7):
        assert torch.equal(lab[i, : int(cnt[i])], raw[i][1]) and float(lab[i, int(cnt[i]):].abs().sum()) == 0.0


def test_prefetcher_refuses_cpu():
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        y.DevicePrefetcher([], "cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("nc", [1, 3])
def test_prefetcher_matches_reference_style_loader(tmp_path, nc):
    from torch.utils.data import DataLoader
    d = make_dataset(tmp_path, n=11, seed=nc)
    S, dev = 64, torch.device("cuda")
    std = DataLoader(y.YOLODataset(d, nc, img_size=S), batch_size=4, shuffle=False, collate_fn=y.yolo_collate_fn)
    raw = DataLoader(y.YOLODataset(d, nc, img_size=S, raw=True), batch_size=4, shuffle=False, collate_fn=y.raw_collate_fn)
    ref = [(imgs, y.stack_targets(tg, "cpu")) for imgs, tg in std]
    for loader in (raw, std):                           # raw mode and pass-through mode
        got = [(i.clone(), [t.clone() for t in tg]) for i, tg in y.DevicePrefetcher(loader, dev, img_size=S, num_classes=nc, depth=2)]
        assert len(got) == len(ref) == 3
        dyn = list(y.DevicePrefetcher(loader, dev, img_size=S, num_classes=nc, depth=1, static_buffers=False))
        for (gi, gt), (di, dt) in zip(got, dyn):        # fresh-tensor mode: batches may be kept without copying
            assert torch.equal(gi, di) and all(torch.equal(a, b) for a, b in zip(gt, dt))
        for (gi, gt), (ri, rt) in zip(got, ref):
            assert gi.is_cuda and all(t.is_cuda for t in gt)
            gc = gi.cpu()                               # the reference's /255 is a CPU division: compare on the CPU
            fi = gc.permute(0, 3, 1, 2).float() / 255.0 if gc.dtype == torch.uint8 else gc
            assert torch.equal(fi, ri)
            for a, b in zip(gt, rt):
                assert torch.equal(a.cpu(), b)


@pytest.mark.gpu
def test_uint8_input_is_bit_identical_to_float_input():
    dev = torch.device("cuda")
    torch.manual_seed(0)
    m = y.YOLO(num_classes=1, img_size=64).to(dev).train()
    u8c = torch.randint(0, 256, (2, 64, 64, 3), dtype=torch.uint8)
    u8, f = u8c.to(dev), (u8c.permute(0, 3, 1, 2).float() / 255.0).to(dev)     # the reference divides on the CPU
    with torch.no_grad():
        a = [t.clone() for t in m(f)]
        b = [t.clone() for t in m(u8)]
    for p, q in zip(a, b):
        assert torch.equal(p, q)


@pytest.mark.gpu
def test_train_epoch_through_the_prefetcher_equals_the_plain_loader(tmp_path):
    from torch.utils.data import DataLoader
    d = make_dataset(tmp_path, n=8, seed=5)
    S, nc, dev = 64, 1, torch.device("cuda")
    out = []
    for mode in ("plain", "prefetch_raw"):
        torch.manual_seed(3)
        m = y.YOLO(num_classes=nc, img_size=S).to(dev)
        opt = y.HipAdam(m, lr=1e-3)
        if mode == "plain":
            loader = DataLoader(y.YOLODataset(d, nc, img_size=S), batch_size=4, shuffle=False, collate_fn=y.yolo_collate_fn)
        else:
            loader = y.DevicePrefetcher(DataLoader(y.YOLODataset(d, nc, img_size=S, raw=True), batch_size=4, shuffle=False,
                                                   collate_fn=y.raw_collate_fn), dev, img_size=S, num_classes=nc)
        losses = [y.train_epoch(m, loader, opt, dev, nc) for _ in range(2)]
        out.append((losses, torch.cat([p.detach().flatten() for p in m.parameters()]).cpu()))
    assert out[0][0] == out[1][0]                       # identical inputs -> bitwise identical training
    assert torch.equal(out[0][1], out[1][1])
