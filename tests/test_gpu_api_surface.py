"""Drop-in behaviour of the reference's public surface on the HIP path: the same calls the
reference's own tests make -- module shapes at several sizes, train_epoch with a user-constructed torch optimizer
and with the fused HipAdam, eval_epoch metric ranges, predict's tuple format -- on a tiny synthetic dataset."""
import numpy as np
import pytest
import torch
from torch.utils.data import DataLoader

pytestmark = pytest.mark.gpu


def api():
    import yolo_from_scratch_amd as y
    return y


@pytest.fixture(scope="module")
def dataset_dir(tmp_path_factory):
    from PIL import Image
    root = tmp_path_factory.mktemp("ds")
    (root / "train" / "images").mkdir(parents=True)
    (root / "train" / "labels").mkdir(parents=True)
    rng = np.random.default_rng(0)
    for i in range(6):
        Image.fromarray(rng.integers(0, 255, (200, 260, 3), dtype=np.uint8)).save(root / "train" / "images" / f"im{i}.jpg")
        with open(root / "train" / "labels" / f"im{i}.txt", "w") as f:
            for _ in range(int(rng.integers(1, 4))):
                w, h = rng.uniform(0.1, 0.5, 2)
                f.write(f"{int(rng.integers(0, 2))} {rng.uniform(0.3, 0.7):.4f} {rng.uniform(0.3, 0.7):.4f} {w:.4f} {h:.4f}\n")
    return root


def test_block_and_model_shapes():
    y = api()
    dev = torch.device("cuda")
    assert y.ConvBlock(3, 16, 3, 1, 1).to(dev)(torch.randn(2, 3, 32, 32, device=dev)).shape == (2, 16, 32, 32)
    assert y.ConvBlock(16, 32, 3, 2, 1).to(dev)(torch.randn(1, 16, 32, 32, device=dev)).shape == (1, 32, 16, 16)
    assert y.Bottleneck(16, 16).to(dev)(torch.randn(2, 16, 8, 8, device=dev)).shape == (2, 16, 8, 8)
    assert y.C3(32, 64, n=2).to(dev)(torch.randn(2, 32, 8, 8, device=dev)).shape == (2, 64, 8, 8)
    assert y.SPPF(32, 32).to(dev)(torch.randn(2, 32, 8, 8, device=dev)).shape == (2, 32, 8, 8)
    for S, nc, B in ((512, 1, 1), (320, 3, 2), (1024, 1, 1)):
        m = y.YOLO(num_classes=nc, img_size=S).to(dev)
        outs = m(torch.randn(B, 3, S, S, device=dev))
        assert [tuple(o.shape) for o in outs] == [(B, S // s, S // s, 3, 5 + nc) for s in (8, 16, 32)]
        assert all(o.is_contiguous() and torch.isfinite(o).all() for o in outs)
    with pytest.raises(AssertionError):
        y.YOLO(img_size=100).to(dev)(torch.randn(1, 3, 100, 100, device=dev))      # not a multiple of 32 (train.py:606)


@pytest.mark.parametrize("fused", [False, True])
def test_train_epoch_eval_epoch_predict(dataset_dir, fused):
    y = api()
    dev = torch.device("cuda")
    nc, S = 2, 320
    ds = y.YOLODataset(str(dataset_dir / "train" / "images"), num_classes=nc, img_size=S)
    assert len(ds) == 6
    img, tg = ds[0]
    assert img.shape == (3, S, S) and [t.shape for t in tg] == [(S // s, S // s, 3, 5 + nc) for s in (8, 16, 32)]
    assert sum(int((t[..., 4] > 0.5).sum()) for t in tg) >= 1
    loader = DataLoader(ds, batch_size=3, shuffle=False, collate_fn=y.yolo_collate_fn)
    torch.manual_seed(0)
    model = y.YOLO(num_classes=nc, img_size=S).to(dev)
    before = [p.detach().clone() for p in model.parameters()]
    opt = y.HipAdam(model, lr=1e-3) if fused else torch.optim.Adam(model.parameters(), lr=1e-3)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, y.get_lr_lambda(1, 3, 1e-3, 1e-5))
    res = y.train_epoch(model, loader, opt, dev, nc)
    sched.step()
    res2 = y.train_epoch(model, loader, opt, dev, nc)
    assert len(res) == 4 and all(isinstance(v, float) and np.isfinite(v) and v >= 0 for v in res + res2)
    assert res[0] < 1e6 and any(not torch.equal(a, b.detach()) for a, b in zip(before, model.parameters()))
    loss, p, r, f1 = y.eval_epoch(model, loader, dev, nc)
    assert np.isfinite(loss) and 0 <= p <= 100 and 0 <= r <= 100 and 0 <= f1 <= 100
    assert all(q.grad is None or torch.isfinite(q.grad).all() for q in model.parameters())
    model.initialize_detection_biases(prior=0.3)
    dets = y.predict(model, str(dataset_dir / "train" / "images" / "im0.jpg"), dev, nc, conf_threshold=0.25, iou_threshold=0.4)
    assert isinstance(dets, list)
    for d in dets:
        assert len(d) == 6 and isinstance(d[5], int) and 0 <= d[5] < nc
    # state-dict round trip (reference quirk Q5 fixed: grid buffers are contiguous)
    m2 = y.YOLO(num_classes=nc, img_size=S)
    m2.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()})


def test_autograd_path_equals_fused_path():
    """Same seeded model and batch: loss.backward() through the autograd nodes and the fused trainer give the same
    gradients (bitwise: they replay the same op lists)."""
    y = api()
    dev = torch.device("cuda")
    x = torch.rand(2, 3, 320, 320, generator=torch.Generator().manual_seed(3)).to(dev)
    tg = [t.to(dev) for t in y.synthetic_targets(2, 1, 320, 6, 9)]
    torch.manual_seed(1)
    a = y.YOLO(num_classes=1, img_size=320).to(dev)
    loss = y.yolo_loss_multiscale(a(x), tg, a.anchors, 1)[0]
    loss.backward()
    torch.manual_seed(1)
    b = y.YOLO(num_classes=1, img_size=320).to(dev)
    tr = y.HipTrainer(b, lr=0.0, max_norm=None)
    out = tr.step(x, tg)
    assert abs(float(out[0]) - float(loss)) <= 1e-6 * abs(float(loss))
    for (n, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        assert torch.equal(pa.grad, pb.grad), n


@pytest.mark.parametrize("size,S,B,nc", [("n", 416, 1, 1), ("m", 224, 2, 2), ("s", 352, 3, 1), ("l", 160, 1, 1), ("x", 96, 2, 3)])
def test_other_sizes_and_ragged_shapes_match_oracle(size, S, B, nc):
    """Other width/depth multipliers (24/48/96... channel counts, 2 Bottlenecks per C3 for 'm'), odd grid sizes
    (13, 11, 7 cells: ragged GEMM tiles and row segments) and batch 1, one fused step vs the CPU oracle."""
    from oracle import yolo_oracle as orc
    y = api()
    wm, dm = y.YOLO_SIZES[size]
    torch.manual_seed(4)
    m = y.YOLO(num_classes=nc, img_size=S, width_mult=wm, depth_mult=dm)
    P = {k: v.clone() for k, v in m.state_dict().items()}
    names = [n for n, _ in m.named_parameters()]
    x = torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(8))
    targets = y.synthetic_targets(B, nc, S, 5, 31)
    m = m.cuda()
    tr = y.HipTrainer(m, lr=1e-3, max_norm=10.0)
    out = tr.step(x.cuda(), [t.cuda() for t in targets]).cpu().numpy()
    for n in names:
        P[n].requires_grad_(True)
    ref = orc.loss_multiscale(orc.forward(P, x, nc, True), targets, orc.anchors_of(P), nc)
    np.testing.assert_allclose(out[:4], [float(v) for v in ref], rtol=1e-4, atol=1e-6)
    ref[0].backward()
    total, coef = orc.clip_coef([P[n].grad for n in names], 10.0)
    assert abs(float(tr.norm) - total) <= 5e-4 * total
    params = dict(m.named_parameters())
    for n in names[::7]:
        if n.endswith(".bias") and not ("bn" in n or n.startswith("head")):
            continue                                      # conv biases in front of BN: noise gradients (Q2)
        r = P[n].grad * coef
        assert float((params[n].grad.cpu() - r).abs().max()) <= 4e-3 * float(r.abs().max()) + 1e-8, n


def test_no_object_batch_and_loss_identities():
    """yolo_loss with no positives: bbox == 0, cls == 0, obj > 0 (reference tests/test_loss.py:91-109), and the
    weight identity total == 0.05*b + 1.0*o + 0.5*c (111-129); gradients stay finite."""
    y = api()
    dev = torch.device("cuda")
    anc = torch.tensor([[30., 61.], [62., 45.], [59., 119.]], device=dev)
    pred = torch.randn(2, 13, 13, 3, 6, device=dev, requires_grad=True)
    tgt = torch.zeros(2, 13, 13, 3, 6, device=dev)
    tot, b, o, c = y.yolo_loss(pred, tgt, anc, 1)
    assert b.item() == 0.0 and c.item() == 0.0 and o.item() > 0
    tgt[0, 3, 4, 1] = torch.tensor([0.3, 0.3, 0.2, 0.25, 1.0, 1.0], device=dev)
    tot, b, o, c = y.yolo_loss(pred, tgt, anc, 1)
    assert abs(tot.item() - (0.05 * b.item() + 1.0 * o.item() + 0.5 * c.item())) < 1e-5 and b.item() > 0
    tot.backward()
    assert torch.isfinite(pred.grad).all() and float(pred.grad.abs().sum()) > 0
    dec = y.decode_predictions(pred.detach(), anc)
    assert dec.shape == pred.shape and torch.equal(dec[..., 4:], pred.detach()[..., 4:])
    assert float(dec[..., 2:4].min()) > 0 and -0.1 < float(dec[..., 0].min()) and float(dec[..., 0].max()) < 1.1


@pytest.mark.parametrize("tag", ["nc1", "nc3"])
def test_eval_epoch_counts_match_reference_loop(tag):
    """eval_epoch's grid-cell precision / recall / F1 (one HIP kernel) against the numbers the
    reference's own 4-deep Python loop produced on the same recorded predictions (tests/golden/evalcounts.npz)."""
    from conftest import load_golden
    y = api()
    g = load_golden("evalcounts")
    nc, S = (int(v) for v in g[f"{tag}/meta"])
    dev = torch.device("cuda")
    grids = [S // 8, S // 16, S // 32]
    preds = [torch.from_numpy(g[f"{tag}/pred{s}"]).to(dev) for s in range(3)]
    targets = [torch.zeros(2, gs, gs, 3, 5 + nc) for gs in grids]
    for s in range(3):
        for (b, i, j, a), v in zip(g[f"{tag}/pos_idx{s}"], g[f"{tag}/pos_val{s}"]):
            targets[s][b, i, j, a] = torch.from_numpy(v)

    class Stub(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.anchors = [torch.tensor(a, dtype=torch.float32, device=dev) for a in y.DEFAULT_ANCHORS]

        def forward(self, x):
            return [p.clone() for p in preds]

    loader = [(torch.zeros(2, 3, S, S), [[t[b] for t in targets] for b in range(2)])]
    for conf, iou in ((0.5, 0.5), (0.3, 0.2)):
        got = y.eval_epoch(Stub(), loader, dev, nc, iou_threshold=iou, conf_threshold=conf)
        want = g[f"{tag}/c{conf}_i{iou}"]
        np.testing.assert_allclose(got[0], want[0], rtol=1e-4)
        np.testing.assert_allclose(got[1:], want[1:], rtol=1e-9, atol=1e-12)     # integer counts -> identical ratios


@pytest.mark.parametrize("nc", [1, 3])
def test_device_target_assignment_matches_reference_dataset(nc):
    """yh_assign_targets against what the reference's YOLODataset.__getitem__ produced for the same
    label files (tests/golden/assign.npz) -- bit-exact cells, anchors and stored values -- and against the host rule."""
    from conftest import load_golden
    y = api()
    g = load_golden("assign")
    labels = [[(int(r[0]), float(r[1]), float(r[2]), float(r[3]), float(r[4])) for r in img] for img in g[f"nc{nc}/labels"]]
    tg = y.assign_targets_gpu(labels, 640, nc, torch.device("cuda"))
    for b in range(3):
        for s in range(3):
            t = tg[s][b].cpu()
            pos = t[..., 4] > 0.5
            np.testing.assert_array_equal(pos.nonzero().numpy(), g[f"nc{nc}/b{b}/s{s}/idx"])
            np.testing.assert_array_equal(t[pos].numpy(), g[f"nc{nc}/b{b}/s{s}/val"])
    host = y.synthetic_targets(4, nc, 320, 12, 5)
    rng = np.random.default_rng(5)
    import math
    lab = []
    for _ in range(4):
        c = rng.uniform(0.05, 0.95, size=(12, 2)); wh = np.exp(rng.uniform(math.log(8), math.log(320), size=(12, 2))) / 320
        k = rng.integers(0, max(nc, 1), size=12)
        lab.append([(int(k[i]), float(c[i, 0]), float(c[i, 1]), float(wh[i, 0]), float(wh[i, 1])) for i in range(12)])
    dev = y.assign_targets_gpu(lab, 320, nc, torch.device("cuda"))
    for a, b_ in zip(dev, host):
        assert torch.equal(a.cpu(), b_)


@pytest.mark.parametrize("nc", [1, 3])
def test_predict_batch_equals_predict_per_image(tmp_path, nc):
    """one batched forward + per-image candidate/NMS segments returns exactly what `predict` returns
    image by image (images of different sizes: different letterbox parameters), including an image without detections."""
    from PIL import Image
    y = api()
    dev = torch.device("cuda")
    rng = np.random.default_rng(7)
    paths = []
    for i, (h, w) in enumerate(((120, 200), (260, 140), (96, 96), (300, 310))):
        arr = rng.integers(0, 255, (h, w, 3), dtype=np.uint8) if i != 2 else np.zeros((h, w, 3), dtype=np.uint8)
        Image.fromarray(arr).save(tmp_path / f"p{i}.png")
        paths.append(str(tmp_path / f"p{i}.png"))
    torch.manual_seed(11)
    model = y.YOLO(num_classes=nc, img_size=256).to(dev)
    model.initialize_detection_biases(prior=0.3)
    single = [y.predict(model, p, dev, nc, conf_threshold=0.3, iou_threshold=0.4) for p in paths]
    batch = y.predict_batch(model, paths, dev, nc, conf_threshold=0.3, iou_threshold=0.4)
    assert len(batch) == len(single) == 4 and sum(len(d) for d in single) > 20
    for a, b in zip(single, batch):
        assert a == b                       # same boxes, scores, classes, order -- bit for bit
    assert y.predict_batch(model, [], dev, nc) == []
    again = y.predict_batch(model, paths[:2], dev, nc, conf_threshold=0.3, iou_threshold=0.4)     # smaller batch reuses buffers
    assert again == single[:2]


def test_two_contexts_two_streams():
    """Two host threads, each with its own model, HIP stream and (per-thread) yh_context, run forward passes
    concurrently: the contexts bind distinct side streams / events and every result equals the single-threaded
    one bit for bit.  An op failure on a context with the side lane forked leaves the context usable (joined)."""
    import threading
    from yolo_from_scratch_amd import _lib as L
    y = api()
    dev = torch.device("cuda:0")
    xs = [torch.rand(2, 3, 160, 160, generator=torch.Generator().manual_seed(5 + i)).to(dev) for i in range(2)]
    models = []
    for i in range(2):
        torch.manual_seed(i)
        models.append(y.YOLO(num_classes=2, img_size=160).to(dev).train())
    with torch.no_grad():
        want = [[o.clone() for o in m(x)] for m, x in zip(models, xs)]
    torch.cuda.synchronize()
    got, infos, errs = [None, None], [None, None], []

    def work(i):
        try:
            st = torch.cuda.Stream(dev)
            with torch.cuda.stream(st), torch.no_grad():
                for _ in range(5):
                    out = [o.clone() for o in models[i](xs[i])]
            st.synchronize()
            got[i] = out
            infos[i] = L.context_for(0).info()          # this thread's default context
        except Exception as e:                            # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    for i in range(2):
        for a, b in zip(got[i], want[i]):
            assert torch.equal(a, b)
    a, b = infos
    assert a["device"] == b["device"] == 0
    for k in ("side_stream", "fork_event", "join_event"):
        assert a[k] and b[k] and a[k] != b[k], k
    # failure with the side lane forked: FORK, a side-lane NOP, then an unknown op -> error, lane joined, context reusable
    ctx = L.Context()
    ops = (L.YhOp * 3)()
    ops[0].kind = L.OP_FORK
    ops[1].kind, ops[1].lane = L.OP_NOP, 1
    ops[2].kind = 9999
    st = torch.cuda.current_stream(dev).cuda_stream
    with pytest.raises(RuntimeError, match=r"op #2"):
        L.run_ops(ops, 3, st, ctx)
    L.run_ops(ops, 2, st, ctx)
    torch.cuda.synchronize()
    assert ctx.info()["side_stream"]
    ctx.close()


def test_plan_cache_is_bounded_lru():
    """every plan owns a full set of activation buffers; the per-module cache keeps the three most recently used."""
    y = api()
    m = y.ConvBlock(8, 16, 3, 1, 1).cuda().eval()
    with torch.no_grad():
        for hw in (8, 10, 12, 14, 16):
            m(torch.rand(1, 8, hw, hw, device="cuda"))
    keys = [k[0][2] for k in m._plans]
    assert keys == [12, 14, 16] and len(m._plans) == m.PLAN_CACHE_MAX
    with torch.no_grad():
        m(torch.rand(1, 8, 12, 12, device="cuda"))            # a hit moves the plan to the recent end
        m(torch.rand(1, 8, 20, 20, device="cuda"))
    assert [k[0][2] for k in m._plans] == [16, 12, 20]
