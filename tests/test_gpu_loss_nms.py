"""GPU parity of the loss / decode / CIoU kernels and of candidate extraction + NMS against the
golden vectors recorded from the reference and against the pinned CPU oracle."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import yolo_oracle as orc
from test_oracle_pinned import loss_inputs, T, close

pytestmark = pytest.mark.gpu


def api():
    import yolo_from_scratch_amd as y
    return y


def test_decode_matches_golden_and_grad():
    y = api()
    g = load_golden("decode")
    for key in ("nc1_img640", "nc1_img1280", "nc3_img640", "nc3_img1280", "default"):
        img = int(key.split("img")[1]) if "img" in key else 640
        raw = T(g[f"{key}/raw"]).cuda().requires_grad_(True)
        anc = T(g[f"{key}/anchors"]).cuda()
        out = y.decode_predictions(raw, anc, img) if "img" in key else y.decode_predictions(raw, anc)
        close(out.detach().cpu(), g[f"{key}/decoded"], 1e-5, 1e-6)          # fp32 boxes within 1e-4 relative
        w = torch.randn_like(out)
        (out * w).sum().backward()
        rc = T(g[f"{key}/raw"]).clone().requires_grad_(True)
        (orc.decode(rc, T(g[f"{key}/anchors"]), img) * w.cpu()).sum().backward()
        close(raw.grad.cpu(), rc.grad, 1e-4, 1e-6)


@pytest.mark.parametrize("case", ["identical", "disjoint", "partial", "aspect", "contained", "rand"])
def test_ciou_matches_golden(case):
    y = api()
    g = load_golden("ciou")
    p = T(g[f"{case}/pred"]).cuda().requires_grad_(True)
    l = y.ciou_loss(p, T(g[f"{case}/tgt"]).cuda())
    (l * 2.0).backward()
    close(l.item(), float(g[f"{case}/loss"]), 1e-5, 1e-6)
    close(p.grad.cpu() / 2.0, g[f"{case}/dpred"], 1e-4, 1e-6)


@pytest.mark.parametrize("tag", ["nc1", "nc3", "nc1_empty"])
def test_multiscale_loss_matches_golden(tag):
    y = api()
    g, nc, preds, targets = loss_inputs(tag)
    anchors = [torch.tensor(a, dtype=torch.float32).cuda() for a in orc.DEFAULT_ANCHORS]
    pg = [p.cuda().requires_grad_(True) for p in preds]
    tg = [t.cuda() for t in targets]
    tot, b, o, c = y.yolo_loss_multiscale(pg, tg, anchors, nc)
    close([tot.item(), b.item(), o.item(), c.item()], g[f"{tag}/scalars"], 1e-4, 1e-6)   # loss within 1e-4 relative
    tot.backward()
    for s, p in enumerate(pg):
        close(p.grad.cpu().reshape(-1)[T(g[f"{tag}/s{s}/sample_idx"])], g[f"{tag}/s{s}/dpred_sample"], 1e-4, 1e-9)
        if f"{tag}/s{s}/dpred_pos" in g.files:
            close(p.grad.cpu()[targets[s][..., 4] > 0.5], g[f"{tag}/s{s}/dpred_pos"], 2e-4, 1e-8)
        close([float(p.grad.double().sum()), float(p.grad.double().abs().sum())], g[f"{tag}/s{s}/dpred_sum"], 1e-4, 1e-7)
        one = y.yolo_loss(p.detach(), tg[s], anchors[s], nc)
        close([v.item() for v in one], g[f"{tag}/per_scale"][s], 1e-4, 1e-6)
        assert abs(one[0].item() - (0.05 * one[1].item() + 1.0 * one[2].item() + 0.5 * one[3].item())) < 1e-5


def test_loss_backward_from_component_outputs():
    y = api()
    g, nc, preds, targets = loss_inputs("nc3")
    anchors = [torch.tensor(a, dtype=torch.float32) for a in orc.DEFAULT_ANCHORS]
    pc = [p.clone().requires_grad_(True) for p in preds]
    tot, b, o, c = orc.loss_multiscale(pc, targets, anchors, nc)
    (0.5 * tot + 2.0 * b - o + 3.0 * c).backward()
    pg = [p.cuda().requires_grad_(True) for p in preds]
    tot, b, o, c = y.yolo_loss_multiscale(pg, [t.cuda() for t in targets], [a.cuda() for a in anchors], nc)
    (0.5 * tot + 2.0 * b - o + 3.0 * c).backward()
    for a, r in zip(pg, pc):
        assert float((a.grad.cpu() - r.grad).abs().max()) <= 2e-4 * float(r.grad.abs().max())


# ---- NMS ---------------------------------------------------------------------------------------------
def _rand_boxes(M, seed, ncls, dup=False, spread=600.0):
    rng = np.random.default_rng(seed)
    ctr = rng.uniform(0, spread, size=(M, 2))
    wh = rng.uniform(8, 120, size=(M, 2))
    boxes = np.concatenate([ctr - wh / 2, ctr + wh / 2], 1).astype(np.float32)
    scores = rng.uniform(0.01, 1.0, size=M).astype(np.float32)
    if dup and M > 8:
        scores[rng.integers(0, M, M // 3)] = scores[rng.integers(0, M, M // 3)]    # many exact score ties
        boxes[rng.integers(0, M, M // 8)] = boxes[rng.integers(0, M, M // 8)]      # identical boxes
    classes = rng.integers(0, ncls, size=M).astype(np.int64)
    return boxes, scores, classes


@pytest.mark.parametrize("mode", ["cpu", "vanilla", "trick", "cuda"])
@pytest.mark.parametrize("M,ncls,dup,thr", [(0, 1, False, 0.4), (1, 1, False, 0.4), (3, 1, False, 0.5), (64, 1, False, 0.4),
                                           (64, 3, True, 0.4), (64, 80, True, 0.4), (65, 3, True, 0.4), (129, 1, True, 0.6),
                                           (1000, 1, False, 0.4), (1000, 3, True, 0.4), (1000, 80, True, 0.4),
                                           (1001, 3, True, 0.4), (1001, 80, True, 0.4), (4096, 3, True, 0.45),
                                           (4096, 80, True, 0.4), (4097, 1, False, 0.3), (5000, 80, True, 0.4),
                                           (5001, 80, True, 0.4), (20000, 5, True, 0.4)])
def test_nms_indices_bit_exact_vs_oracle(M, ncls, dup, thr, mode):
    """Both branches of torchvision.ops.batched_nms and its size rule (M = 1000 / 1001 straddle the CPU limit of 4000
    elements, 5000 / 5001 the GPU limit), boxes with negative coordinates, score ties, identical boxes."""
    y = api()
    boxes, scores, classes = _rand_boxes(M, 1234 + M, ncls, dup, spread=300.0 if M <= 5001 else 2000.0)
    boxes -= np.float32(30.0)                                     # negative coordinates: cross-class overlaps under the trick
    want = orc.nms_batched(boxes, scores, classes, thr, mode)
    got = y.batched_nms(torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda(), torch.from_numpy(classes).cuda(), thr, mode)
    assert got.dtype == torch.int64
    np.testing.assert_array_equal(got.cpu().numpy(), want)


def test_nms_default_mode_is_the_reference_cpu_branch():
    y = api()
    for M in (1000, 1001):
        boxes, scores, classes = _rand_boxes(M, 99 + M, 80, True, spread=300.0)
        dflt = y.batched_nms(torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda(), torch.from_numpy(classes).cuda(), 0.4)
        np.testing.assert_array_equal(dflt.cpu().numpy(), orc.nms_batched(boxes, scores, classes, 0.4, "trick" if M == 1000 else "vanilla"))


def test_nms_branches_differ_where_torchvisions_do():
    """Constructed nc > 1 inputs on which the coordinate trick and the per-class form select differently (the oracle test
    of the same name explains them): the HIP path follows each branch, and the default follows the CPU rule (M <= 1000)."""
    y = api()
    s2 = torch.tensor([0.9, 0.8]).cuda()
    b = torch.tensor([[600, 600, 640, 640], [-41, -41, -1, -1]], dtype=torch.float32).cuda()
    c = torch.tensor([0, 1]).cuda()
    assert y.batched_nms(b, s2, c, 0.4, "vanilla").tolist() == [0, 1]
    assert y.batched_nms(b, s2, c, 0.4, "trick").tolist() == [0]
    assert y.batched_nms(b, s2, c, 0.4).tolist() == [0]
    b = torch.tensor([[303.3179016113281, 364.748291015625, 337.342529296875, 374.748291015625],
                      [317.89996337890625, 364.748291015625, 351.92462158203125, 374.748291015625], [0, 0, 640, 1]],
                     dtype=torch.float32).cuda()
    s3, c3 = torch.tensor([0.9, 0.8, 0.1]).cuda(), torch.tensor([79, 79, 0]).cuda()
    assert y.batched_nms(b, s3, c3, 0.4, "vanilla").tolist() == [0, 1, 2]
    assert y.batched_nms(b, s3, c3, 0.4, "trick").tolist() == [0, 2]
    with pytest.raises(ValueError):
        y.batched_nms(b, s3, c3, 0.4, "fastest")


def test_nms_threshold_edge_is_strict_greater():
    y = api()
    # IoU of these two boxes is exactly 1/3 in fp32: 50 / (100 + 100 - 50)
    b = torch.tensor([[0, 0, 10, 10], [5, 0, 15, 10]], dtype=torch.float32).cuda()
    s = torch.tensor([0.9, 0.8]).cuda()
    c = torch.zeros(2, dtype=torch.int64).cuda()
    third = float(np.float32(50.0) / np.float32(150.0))
    for mode in ("vanilla", "trick"):
        assert y.batched_nms(b, s, c, third, mode).tolist() == [0, 1]                # IoU == thr is kept
        assert y.batched_nms(b, s, c, float(np.nextafter(np.float32(third), np.float32(0))), mode).tolist() == [0]
        # the threshold is a double (torchvision's CPU kernel promotes the fp32 IoU): inter 2 / union 5 = float32(0.4) > 0.4
        b2 = torch.tensor([[0, 0, 3.5, 1], [1.5, 0, 5, 1]], dtype=torch.float32).cuda()
        assert y.batched_nms(b2, s, c, 0.4, mode).tolist() == [0]
        assert y.batched_nms(b2, s, c, float(np.float32(0.4)), mode).tolist() == [0, 1]
        assert y.batched_nms(b2, s, c, float(np.nextafter(np.float32(0.4), np.float32(0))), mode).tolist() == [0]


def test_nms_matches_reference_python_nms_known_answers():
    y = api()
    g = load_golden("nms")
    for M in (3, 64, 300, 1000):
        b, s = torch.from_numpy(g[f"M{M}/boxes"]).cuda(), torch.from_numpy(g[f"M{M}/scores"]).cuda()
        for thr in (0.4, 0.6):
            got = y.batched_nms(b, s, torch.zeros(M, dtype=torch.int64).cuda(), thr)
            np.testing.assert_array_equal(got.cpu().numpy(), g[f"M{M}_t{thr}/kept"])
    ka = g["ka3/dets"]
    got = y.batched_nms(torch.tensor(ka[:, :4], dtype=torch.float32).cuda(), torch.tensor(ka[:, 4], dtype=torch.float32).cuda(),
                        torch.zeros(3, dtype=torch.int64).cuda(), 0.5)
    assert got.tolist() == [0, 2]


@pytest.mark.parametrize("tag", ["nc1_sq", "nc3_rect"])
def test_candidates_and_nms_pipeline_vs_reference_capture(tag):
    y = api()
    g = load_golden("candidates")
    nc, img, thr, scale, pad_top, pad_left = g[f"{tag}/meta"]
    nc, img = int(nc), int(img)
    preds = [T(g[f"{tag}/pred{s}"]).cuda().contiguous() for s in range(3)]
    anchors = [T(a) for a in g[f"{tag}/anchors"]]
    det = y.Detector([p.shape[1] for p in preds], nc, preds[0].device)
    det.candidates(preds, anchors, img, float(thr), pad_left, pad_top, scale)
    M = int(det.count.item())
    assert M == len(g[f"{tag}/scores"])                       # same cells pass the threshold, same order
    np.testing.assert_array_equal(det.classes[:M].cpu().numpy(), g[f"{tag}/classes"])
    close(det.scores[:M].cpu(), g[f"{tag}/scores"], 1e-5, 1e-7)
    close(det.boxes[:M].cpu(), g[f"{tag}/boxes"], 1e-5, 2e-3)
    det.nms(0.4)
    got = det.keep[: int(det.nkeep.item())].cpu().numpy()
    want = orc.nms_batched(det.boxes[:M].cpu().numpy(), det.scores[:M].cpu().numpy(), det.classes[:M].cpu().numpy(), 0.4)
    np.testing.assert_array_equal(got, want)                  # bit-exact index selection on identical inputs
    # The reference-captured candidates are the same cells with boxes / scores that differ by fp32 rounding (GPU expf vs
    # torch-CPU sigmoid).  Every decision of the GPU's scan is replayed against the reference's values: it must agree
    # unless the pair's IoU straddles the threshold between the two evaluations (tests/nms_explain.py); with no such pair
    # the two selections must be identical.
    from nms_explain import explain_selection
    ambiguous, swaps = explain_selection(det.boxes[:M].cpu().numpy(), det.scores[:M].cpu().numpy(), g[f"{tag}/boxes"],
                                         g[f"{tag}/scores"], g[f"{tag}/classes"], got, 0.4)
    ref_sel = orc.nms_batched(g[f"{tag}/boxes"], g[f"{tag}/scores"], g[f"{tag}/classes"], 0.4)
    assert len(ambiguous) <= max(2, M // 200), ambiguous
    if not ambiguous and not swaps:
        np.testing.assert_array_equal(got, ref_sel)
