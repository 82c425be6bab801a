"""BASELINE config 5 at its stated size -- bs=1, 640x640, BN folded, candidates + global NMS + result table captured into ONE
hipGraph -- against the CPU oracle pipeline (train.py:1114-1250), for nc=1 and nc=80, float and uint8 image input; and the
weight-state bookkeeping of eval plans / InferenceSession (folded BatchNorm copies must never go stale silently)."""
import numpy as np
import pytest
import torch

from oracle import yolo_oracle as orc

pytestmark = pytest.mark.gpu


def api():
    import yolo_from_scratch_amd as y
    return y


def _detecting_model(y, nc, S, seed, gain, prior=0.3):
    """Untrained weights produce no detections; spread the objectness / class logits and give BatchNorm non-trivial running
    statistics so that the folded path, the threshold and the NMS all have work."""
    torch.manual_seed(seed)
    m = y.YOLO(num_classes=nc, img_size=S)
    m.initialize_detection_biases(prior=prior)
    with torch.no_grad():
        for hd in (m.head_p3, m.head_p4, m.head_p5):
            hd[-1].weight.mul_(gain)
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.running_mean.uniform_(-0.2, 0.2); mod.running_var.uniform_(0.5, 1.5)
                mod.weight.uniform_(0.8, 1.2); mod.bias.uniform_(-0.1, 0.1)
    return m


def _calibrated_model(y, nc, S, seed, gain=6.0, prior=0.05, warm=24):
    """A random-weight model whose outputs DEPEND on the image: with the default running statistics (mean 0, var 1) the
    signal of a freshly initialised net dies out layer by layer and every cell sees the head bias.  The BatchNorm running
    statistics are therefore calibrated with `warm` training-mode forwards of the CPU oracle on random images."""
    torch.manual_seed(seed)
    m = y.YOLO(num_classes=nc, img_size=S)
    m.initialize_detection_biases(prior=prior)
    with torch.no_grad():
        for hd in (m.head_p3, m.head_p4, m.head_p5):
            hd[-1].weight.mul_(gain)
    P = {k: v.clone() for k, v in m.state_dict().items()}
    torch.set_num_threads(16)
    with torch.no_grad():
        for k in range(warm):
            orc.forward(P, torch.rand(1, 3, S, S, generator=torch.Generator().manual_seed(100 + k)), nc, training=True)
    m.load_state_dict(P)
    return m, P


@pytest.mark.parametrize("nc", [1, 80])
def test_config5_640_hipgraph_matches_eager_and_oracle(nc):
    from nms_explain import explain_detections
    y = api()
    S = 640
    m, P = _calibrated_model(y, nc, S, 11 + nc)
    m = m.cuda().eval()
    rng = np.random.default_rng(5)
    probe = torch.from_numpy((rng.random((S, S, 3)) * 255).astype(np.uint8))
    with torch.no_grad():
        obj = torch.cat([torch.sigmoid(p[..., 4]).flatten() for p in m(probe.unsqueeze(0).cuda())]).sort(descending=True).values

    def gap_threshold(k):
        """A threshold between two DISTINCT objectness values with about k cells above it (random images through random
        weights give large plateaus of identical logits; a threshold ON a plateau is decided by the last bit of the sigmoid)."""
        vals, counts = torch.unique(obj, return_counts=True)           # ascending
        above = counts.flip(0).cumsum(0).flip(0)                       # cells with obj >= vals[i]
        ok = (above >= k // 2) & (above <= k + k // 2)
        ok[0] = False
        gaps = torch.where(ok, vals - vals.roll(1), torch.zeros_like(vals))
        i = int(torch.argmax(gaps))
        assert float(gaps[i]) > 1e-6, "no usable gap in the objectness distribution"      # >= 8 ulps of the sigmoid either side
        return float((vals[i] + vals[i - 1]) / 2)

    # two thresholds: ~3000 candidates (torchvision's per-class branch) and ~600 (the coordinate trick, M <= 1000)
    for conf, want_trick in ((gap_threshold(3000), False), (gap_threshold(600), True)):
        eager = y.InferenceSession(m, conf_threshold=conf, iou_threshold=0.4, use_graph=False)
        graph = y.InferenceSession(m, conf_threshold=conf, iou_threshold=0.4, use_graph=True)
        for k in range(2):
            u8 = probe if k == 0 else torch.from_numpy((rng.random((S, S, 3)) * 255).astype(np.uint8))
            img = (u8.permute(2, 0, 1).float() / 255.0).unsqueeze(0)          # what train.py:1136-1137 builds on the host
            lb = (3.0 * k, 1.0 * k, 1.0 - 0.25 * k)
            de = eager.run(img, *lb)
            dg = graph.run(img.pin_memory(), *lb)
            du = graph.run(u8, *lb)                                            # uint8 bytes: /255 on the device, same bits
            assert len(de) > 5 and de == dg == du
            M = int(graph.det.count.item())
            assert 100 < M < 6000 and (k > 0 or orc.nms_uses_trick(M, "cpu") == want_trick)
            with torch.no_grad():
                preds = orc.forward(P, img, nc, training=False)
                hip_preds = m(img.cuda())
            for a, r in zip(hip_preds, preds):
                assert float((a.cpu() - r).abs().max()) < 2e-3 * max(1.0, float(r.abs().max()))
            # kept indices bit-exact on the GPU's own candidates (torchvision's branch for this M), every difference to the
            # oracle pipeline accounted for: tests/nms_explain.py
            explain_detections(graph.det, hip_preds, preds, orc.anchors_of(P), S, nc, conf, 0.4, lb)
            # the returned tuples are the table rows: kept candidates in NMS order
            kk = int(graph.det.nkeep.item())
            idx = graph.det.keep[:kk].long()
            got = np.array([d[:5] for d in dg], np.float32).reshape(-1, 5)
            np.testing.assert_array_equal(got[:, :4], graph.det.boxes[idx].cpu().numpy())
            np.testing.assert_array_equal(got[:, 4], graph.det.scores[idx].cpu().numpy())
            assert [d[5] for d in dg] == graph.det.classes[idx].cpu().tolist()


def test_result_table_longer_than_the_pinned_mirror():
    y = api()
    m = _detecting_model(y, 3, 320, 5, 30.0).cuda().eval()
    ses = y.InferenceSession(m, conf_threshold=0.3, iou_threshold=0.9, use_graph=True)
    img = torch.rand(1, 3, 320, 320, generator=torch.Generator().manual_seed(2))
    full = ses.run(img)
    assert len(full) > 64
    small = y.Detector(ses.det.grids, 3, ses.device, max_fetch=16)             # forces the second copy for rows 16...
    with torch.no_grad():
        preds = m(img.cuda())
    small.candidates(preds, m.anchors, 320, 0.3)
    small.nms(0.9)
    assert small.fetch() == full


def test_folded_weights_follow_every_sanctioned_write_and_invalidate_covers_the_rest():
    """eval plans fold BatchNorm once per weight state.  Writes torch's version counters see (in-place ops on the
    registered tensors), writes by this package (optimizer kernel, training forward, load_state_dict, HipTrainer's
    load_flat_parameters) are followed automatically; `.data` / flat-view writes need invalidate_folded_weights()."""
    y = api()
    S = 160
    m = _detecting_model(y, 2, S, 3, 10.0).cuda().eval()
    x = torch.rand(1, 3, S, S, generator=torch.Generator().manual_seed(9)).cuda()
    ses = y.InferenceSession(m, conf_threshold=0.0, iou_threshold=0.4, use_graph=True)

    def both():
        with torch.no_grad():
            out = [p.clone() for p in m(x)]
        ses.run(x, fetch=False)
        torch.cuda.synchronize()
        heads = [v.buf.data.view(1, v.H, v.W, 3, v.C // 3).clone() for v in ses.heads]
        for a, b in zip(out, heads):
            assert torch.equal(a, b)                      # the session and the module forward always agree
        return out

    base = both()
    w = m.stem[0].weight
    with torch.no_grad():
        w.mul_(1.25)                                      # version counter moves: automatic
    o1 = both()
    assert not torch.equal(base[0], o1[0])
    w.data.mul_(0.8)                                      # `.data` has its own counter: invisible ...
    m.invalidate_folded_weights()                         # ... so the contract is an explicit invalidation
    o2 = both()
    assert not torch.equal(o1[0], o2[0])
    for a, b in zip(base, o2):
        assert float((a - b).abs().max()) <= 2e-4 * float(a.abs().max())     # 1.25 * 0.8 = 1 up to rounding
    # through the trainer's flat buffer
    tr = y.HipTrainer(m, lr=1e-3)                         # parameters move into flat_p: plans re-trace, graphs re-capture
    m.eval()
    o3 = both()
    for a, b in zip(o2, o3):
        assert torch.equal(a, b)
    tr.flat_p.mul_(1.1)
    y.invalidate_folded_weights()
    o4 = both()
    assert not torch.equal(o3[0], o4[0])
    saved = tr.flat_p.clone()
    tr.load_flat_parameters(saved / 1.1)                  # sanctioned write: invalidates by itself
    o5 = both()
    for a, b in zip(o3, o5):
        assert float((a - b).abs().max()) <= 2e-4 * float(a.abs().max())
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    sd["stem.0.weight"] = sd["stem.0.weight"] * 2.0
    m.load_state_dict(sd)
    o6 = both()
    assert not torch.equal(o5[0], o6[0])
    # a training step through the fused kernels, then eval again
    imgs = torch.rand(2, 3, S, S, generator=torch.Generator().manual_seed(4)).cuda()
    tg = [t.cuda() for t in y.synthetic_targets(2, 2, S, 4, 8)]
    tr.step(imgs, tg)
    m.eval()
    o7 = both()
    assert not torch.equal(o6[0], o7[0])


def test_session_follows_reassigned_parameters():
    """op lists and captured hipGraphs hold raw addresses.  Re-assigning a Parameter attribute, or load_state_dict(
    assign=True) through a plain nn.Module parent, must make the session re-trace by itself -- eager and graph replay."""
    y = api()
    torch.manual_seed(5)
    m = y.YOLO(num_classes=1, img_size=320).cuda().eval()
    m.initialize_detection_biases(prior=0.3)
    img = torch.rand(1, 3, 320, 320, generator=torch.Generator().manual_seed(9))
    for use_graph in (False, True):
        ses = y.InferenceSession(m, conf_threshold=0.2, iou_threshold=0.4, use_graph=use_graph)
        base = ses.run(img)
        w = m.head_p3[0].conv.weight
        w0 = w.detach().clone()
        m.head_p3[0].conv.weight = torch.nn.Parameter((w.detach() * 1.7).clone())     # a NEW tensor at a new address
        moved = ses.run(img)
        fresh = y.InferenceSession(m, conf_threshold=0.2, iou_threshold=0.4, use_graph=False).run(img)
        assert moved == fresh and moved != base
        parent = torch.nn.Sequential(m)
        sd = {k: v.clone() for k, v in parent.state_dict().items()}
        sd["0.head_p3.0.conv.weight"] = w0
        parent.load_state_dict(sd, assign=True)
        assert ses.run(img) == base
