"""Test helper: account for EVERY difference between the NMS selection made on one evaluation of the candidates (the HIP
path's boxes / scores) and the selection the same greedy rule makes on another evaluation of the SAME candidates (the
reference's, which differ by fp32 rounding of the forward / decode arithmetic).

Greedy NMS is discontinuous: a pair whose IoU sits on the threshold, or two scores that swap order, flips one decision
and the flip cascades.  Instead of bounding the overlap of the two selections, `explain_selection` replays the greedy
scan of selection A (made on evaluation A) against evaluation B decision by decision:

  * candidate kept by A: no earlier same-class box kept by A may suppress it under B either -- unless that pair is
    AMBIGUOUS, i.e. the threshold lies between the pair's IoU under A and under B (widened by `ulps` fp32 ulps);
  * candidate suppressed by A: the box that suppressed it under A (the first kept one with IoU_A > thr) must suppress it
    under B too, unless that pair is ambiguous;
  * the scan order: A's descending-score order must also be a descending order of B's scores except between candidates
    whose B-scores differ by less than `score_tol`.

It returns the ambiguous pairs / order swaps; anything it cannot attribute to such a pair raises AssertionError.  With zero
ambiguous pairs and zero swaps the two selections are necessarily identical, which the callers assert as well."""
import numpy as np


def _iou_f32(b, i, js):
    x1, y1, x2, y2 = (b[:, k] for k in range(4))
    area = ((x2 - x1).astype(np.float32) * (y2 - y1).astype(np.float32)).astype(np.float32)
    w = np.maximum(np.float32(0), np.minimum(x2[i], x2[js]) - np.maximum(x1[i], x1[js])).astype(np.float32)
    h = np.maximum(np.float32(0), np.minimum(y2[i], y2[js]) - np.maximum(y1[i], y1[js])).astype(np.float32)
    inter = (w * h).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = (inter / ((area[i] + area[js]).astype(np.float32) - inter).astype(np.float32)).astype(np.float32)
    return iou.astype(np.float64)          # exact promotion: comparisons against the double threshold happen in double


def explain_selection(boxes_a, scores_a, boxes_b, scores_b, classes, sel_a, thr, ulps=4, score_tol=1e-6, mode="cpu"):
    """mode: the branch rule of torchvision.ops.batched_nms both selections were made with (oracle.nms_batched).  Under
    the coordinate trick each evaluation is replayed on ITS OWN shifted boxes (its own boxes.max()) with no class test,
    exactly what that branch computes; the threshold is compared as a double like torchvision's CPU kernel does."""
    from oracle import yolo_oracle as orc
    boxes_a, boxes_b = np.asarray(boxes_a, np.float32), np.asarray(boxes_b, np.float32)
    scores_a, scores_b = np.asarray(scores_a, np.float32), np.asarray(scores_b, np.float32)
    classes = np.asarray(classes)
    if len(scores_a) and orc.nms_uses_trick(len(scores_a), mode):
        boxes_a, boxes_b = orc.nms_shifted_boxes(boxes_a, classes), orc.nms_shifted_boxes(boxes_b, classes)
        classes = np.zeros_like(classes)
    thr32 = float(thr)                                   # fp32 IoUs are promoted to double for every comparison below
    pad = float(ulps) * float(np.spacing(np.float32(thr)))
    order = np.argsort(-scores_a, kind="stable")
    # order: inversions of B's scores along A's order must be near-ties
    sb = scores_b[order]
    swaps = [(int(order[k]), int(order[k + 1])) for k in np.nonzero(sb[1:] > sb[:-1])[0]]
    for i, j in swaps:
        assert abs(float(scores_b[i]) - float(scores_b[j])) <= score_tol, f"order of candidates {i},{j} differs beyond a near-tie"
    kept_set = set(int(v) for v in sel_a)
    assert [int(v) for v in sel_a] == [int(i) for i in order if int(i) in kept_set], "selection A is not in descending score order"
    ambiguous = []
    kept_so_far = []
    for i in order:
        i = int(i)
        prev = np.asarray([k for k in kept_so_far if classes[k] == classes[i]], np.int64)
        if len(prev):
            ia, ib = _iou_f32(boxes_a, i, prev), _iou_f32(boxes_b, i, prev)
            lo, hi = np.minimum(ia, ib) - pad, np.maximum(ia, ib) + pad
            amb = (lo <= thr32) & (thr32 <= hi)                  # threshold between the two evaluations of this pair's IoU
        if i in kept_set:
            if len(prev):
                assert not bool((ia > thr32).any()), f"candidate {i}: selection A is not the greedy outcome on its own inputs"
                bad = (ib > thr32) & ~amb
                assert not bool(bad.any()), f"kept candidate {i} is clearly suppressed by {prev[bad].tolist()} under evaluation B"
                ambiguous += [(int(k), i) for k in prev[(ib > thr32) & amb]]
            kept_so_far.append(i)
        else:
            assert len(prev) and bool((ia > thr32).any()), f"candidate {i} was dropped without a suppressor under evaluation A"
            first = int(np.nonzero(ia > thr32)[0][0])
            if not ib[first] > thr32:
                assert bool(amb[first]), f"candidate {i}: suppressor {int(prev[first])} does not suppress it under evaluation B " \
                                         f"(IoU {float(ia[first])} vs {float(ib[first])})"
                ambiguous.append((int(prev[first]), i))
    return ambiguous, swaps


def explain_detections(det, hip_preds, ref_preds, anchors, img_size, nc, conf_thr, iou_thr, letterbox, mode="cpu"):
    """Everything one image's post-process differs from the oracle pipeline by, accounted for (used by the inference-session
    and predict_batch tests).  det: the Detector after candidates + nms; hip_preds / ref_preds: the three (1,G,G,3,5+nc) head
    tensors of the HIP forward and of the oracle forward on the same image; letterbox = (pad_left, pad_top, scale).
      (a) kept indices are the bit-exact greedy NMS of the GPU's own candidates;
      (b) a cell is a candidate on one side only if |sigmoid_ref(obj) - thr| <= delta / 4, delta = max logit difference;
      (c) every NMS decision agrees with the oracle's boxes / scores of the same cells unless the pair's IoU straddles the
          threshold between the two evaluations; with nothing ambiguous the selections are identical.
    Returns (number of one-sided cells, ambiguous pairs)."""
    import torch
    from oracle import yolo_oracle as orc
    M = int(det.count.item())
    gb, gs, gc = det.boxes[:M].cpu().numpy(), det.scores[:M].cpu().numpy(), det.classes[:M].cpu().numpy()
    got = det.keep[: int(det.nkeep.item())].cpu().numpy()
    np.testing.assert_array_equal(got, orc.nms_batched(gb, gs, gc, iou_thr, mode))                 # (a)
    delta = max(float((a.cpu() - r).abs().max()) for a, r in zip(hip_preds, ref_preds))
    obj_ref = torch.cat([torch.sigmoid(p[0, ..., 4]).reshape(-1) for p in ref_preds])
    obj_hip = torch.cat([torch.sigmoid(p[0, ..., 4].cpu()).reshape(-1) for p in hip_preds])
    in_ref, in_hip = obj_ref > conf_thr, obj_hip > conf_thr
    assert int(in_hip.sum()) == M
    differ = in_ref != in_hip
    assert bool(((obj_ref[differ] - conf_thr).abs() <= delta / 4 + 1e-7).all())                    # (b)
    assert int(differ.sum()) <= max(2, M // 20), (int(differ.sum()), M, delta)      # sanity: the band of (b) is narrow
    ab, asx, ac = orc.candidates(ref_preds, anchors, img_size, nc, -1.0, *letterbox)             # every cell, candidate order
    cells = torch.nonzero(in_hip).reshape(-1)
    same_cls = ac[cells].numpy() == gc
    assert int((~same_cls).sum()) <= max(1, M // 200)                                              # argmax over near-tied class logits
    ambiguous, swaps = explain_selection(gb, gs, ab[cells].numpy(), asx[cells].numpy(), gc, got, iou_thr, ulps=4,
                                         score_tol=max(delta, 1e-6), mode=mode)                    # (c)
    assert len(ambiguous) <= max(2, M // 100), ambiguous
    if not ambiguous and not swaps and not bool(differ.any()) and bool(same_cls.all()):
        b, s, c = orc.candidates(ref_preds, anchors, img_size, nc, conf_thr, *letterbox)
        np.testing.assert_array_equal(got, orc.nms_batched(b.numpy(), s.numpy(), c.numpy(), iou_thr, mode))
    return int(differ.sum()), ambiguous
