"""GPU parity of the module tree (ConvBlock/Bottleneck/C3/SPPF/YOLO) and of the fused training step
against golden vectors recorded from the reference, and against the pinned CPU oracle on the same
seeded inputs (full tensors)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import yolo_oracle as orc
from test_oracle_pinned import T, close

pytestmark = pytest.mark.gpu

Q2 = {"stem.0.bias", "stem.3.bias", "backbone_p3.1.bias", "backbone_p4.0.bias", "backbone_p5.0.bias",
      "sppf.conv1.bias", "sppf.conv2.bias"}     # conv biases cancelled by the following BN: gradients are rounding noise


def api():
    import yolo_from_scratch_amd as y
    return y


def make_block(y, name):
    return {"cb3x3": lambda: y.ConvBlock(8, 16, 3, 1, 1), "cb3x3s2": lambda: y.ConvBlock(8, 16, 3, 2, 1),
            "cb1x1": lambda: y.ConvBlock(8, 12, 1, 1, 0), "bneck": lambda: y.Bottleneck(8, 8),
            "c3": lambda: y.C3(16, 16, n=1), "c3wide": lambda: y.C3(24, 16, n=1), "sppf": lambda: y.SPPF(16, 16)}[name]()


@pytest.mark.parametrize("name", ["cb3x3", "cb3x3s2", "cb1x1", "bneck", "c3", "c3wide", "sppf"])
def test_blocks_match_reference_golden(name):
    y = api()
    g = load_golden("blocks")
    m = make_block(y, name)
    pre = f"{name}/init/"
    m.load_state_dict({k[len(pre):]: T(g[k]) for k in g.files if k.startswith(pre)})
    m = m.cuda().train()
    x = T(g[f"{name}/x"]).cuda().requires_grad_(True)
    out = m(x)
    assert out.shape == tuple(g[f"{name}/y"].shape) and out.is_contiguous()
    close(out.detach().cpu(), g[f"{name}/y"], 1e-4, 1e-5)
    (out * T(g[f"{name}/w"]).cuda()).sum().backward()
    close(x.grad.cpu(), g[f"{name}/dx"], 1e-3, 2e-5)
    params = dict(m.named_parameters())
    for k in g.files:
        if k.startswith(f"{name}/grad/"):
            if name == "sppf" and k.endswith(("conv1.bias", "conv2.bias")):
                continue          # conv bias directly in front of BatchNorm: its gradient is rounding noise (quirk Q2)
            ref = g[k]
            got = params[k.split("/grad/")[1]].grad.cpu().numpy()
            assert np.abs(got - ref).max() <= 2e-4 * max(np.abs(ref).max(), 1e-3), k
        if k.startswith(f"{name}/after/"):
            close(m.state_dict()[k.split("/after/")[1]].cpu(), g[k], 1e-5, 1e-6)
    m.eval()
    with torch.no_grad():
        close(m(x.detach()).cpu(), g[f"{name}/y_eval"], 1e-4, 1e-5)


def _model_and_inputs(nc, S, B=2):
    y = api()
    torch.manual_seed(0)
    m = y.YOLO(num_classes=nc, img_size=S)
    x = torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(123))
    targets = orc.assign_targets(orc.synthetic_boxes(B, nc, S, 8, 2000), S, nc)
    return y, m, x, targets


@pytest.mark.parametrize("tag,nc,S", [("model_nc1", 1, 640), ("model_nc3", 3, 320)])
def test_full_model_autograd_path_matches_reference_golden(tag, nc, S):
    g = load_golden(tag)
    y, m, x, targets = _model_and_inputs(nc, S)
    P = {k: v.clone() for k, v in m.state_dict().items()}            # CPU copy for the oracle
    m = m.cuda().train()
    preds = m(x.cuda())
    assert [tuple(p.shape) for p in preds] == [(2, S // s, S // s, 3, 5 + nc) for s in (8, 16, 32)]
    tg = [t.cuda() for t in targets]
    tot, b, o, c = y.yolo_loss_multiscale(preds, tg, m.anchors, nc)
    close([tot.item(), b.item(), o.item(), c.item()], g["scalars"], 1e-4, 1e-6)        # within 1e-4 relative
    for s, p in enumerate(preds):
        close(p.detach().cpu().reshape(-1)[T(g[f"pred{s}_idx"])], g[f"pred{s}_sample"], 1e-3, 2e-4)
        close([float(p.double().sum())], g[f"pred{s}_sum"][:1], 1e-4, 1e-2)
        # north_star: "fp32 boxes within 1e-4 relative" -- the HIP forward decoded by the HIP decode kernel against the boxes
        # the REFERENCE decodes from its own forward (sampled cells of every scale; 1e-6 absolute floor: normalised
        # coordinates of cells in the first grid column / row are ~1e-3)
        dec = y.decode_predictions(p.detach(), m.anchors[s], S).reshape(-1, 5 + nc)[T(g[f"dec{s}_cells"]).cuda(), :4]
        close(dec.cpu(), g[f"dec{s}_boxes"], 1e-4, 1e-6)
    tot.backward()
    names = list(g["param_names"])
    params = dict(m.named_parameters())
    gn = np.array([float(params[n].grad.double().norm()) for n in names])
    live = np.array([n not in Q2 for n in names])
    close(gn[live], g["grad_norm"][live], 2e-3, 1e-6)
    tn = float(torch.linalg.vector_norm(torch.stack([params[n].grad.norm() for n in names])))
    close(tn, g["total_grad_norm"][0], 2e-4, 0)
    for n in names:
        if n in Q2:
            continue
        ref = g[f"gsample/{n}"]
        idx = np.sort(np.random.default_rng(7).choice(params[n].numel(), size=min(64, params[n].numel()), replace=False))
        got = params[n].grad.cpu().reshape(-1)[torch.from_numpy(idx)].numpy()
        assert np.abs(got - ref).max() <= 2e-3 * max(np.abs(ref).max(), g["grad_norm"][names.index(n)] / np.sqrt(params[n].numel())), n
    for k in g.files:
        if k.startswith("bn/"):
            close(m.state_dict()[k[3:]].cpu(), g[k], 1e-4, 1e-6)
    # the same inputs through the pinned CPU oracle, full tensors
    for n in names:
        P[n].requires_grad_(True)
    op = orc.forward(P, x, nc, training=True)
    for a, r in zip(preds, op):
        assert float((a.detach().cpu() - r.detach()).abs().max()) < 1e-3
    orc.loss_multiscale(op, targets, orc.anchors_of(P), nc)[0].backward()
    for n in names:
        if n in Q2:
            continue
        r = P[n].grad
        assert float((params[n].grad.cpu() - r).abs().max()) <= 2e-3 * float(r.abs().max()) + 1e-7, n


@pytest.mark.parametrize("tag,nc,S", [("model_nc1", 1, 640), ("model_nc3", 3, 320)])
def test_fused_trainer_step_matches_reference_golden(tag, nc, S):
    g = load_golden(tag)
    y, m, x, targets = _model_and_inputs(nc, S)
    m = m.cuda()
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    tr = y.HipTrainer(m, lr=1e-3, max_norm=10.0)
    xg, tg = x.cuda(), [t.cuda() for t in targets]
    out = tr.step(xg, tg).cpu().numpy().copy()
    close(out[:4], g["scalars"], 1e-4, 1e-6)
    close(float(tr.norm), g["total_grad_norm"][0], 2e-4, 0)
    names = list(g["param_names"])
    params = dict(m.named_parameters())
    dn = np.array([float((params[n].detach() - before[n]).double().norm()) for n in names])
    live = np.array([n not in Q2 for n in names])
    close(dn[live], g["delta_norm"][live], 5e-3, 1e-7)
    names_live = [n for n in names if n not in Q2]
    # Adam's first step is lr*g/(|g|+eps): an element whose gradient sits below the fp32 noise floor moves by
    # +-lr in an implementation-dependent direction.  Elements with a gradient well above the floor must move
    # exactly as in the reference; the post-update loss therefore agrees to ~1e-3 rather than 1e-4.
    checked = 0
    for n in names_live:
        gref, dref = g[f"gsample/{n}"], g[f"dsample/{n}"]
        idx = np.sort(np.random.default_rng(7).choice(params[n].numel(), size=min(64, params[n].numel()), replace=False))
        dgot = (params[n].detach() - before[n]).cpu().reshape(-1)[torch.from_numpy(idx)].numpy()
        strong = np.abs(gref) > 1e-2 * max(np.abs(gref).max(), 1e-12)
        if strong.any():
            assert np.abs(dgot[strong] - dref[strong]).max() <= 2e-2 * np.abs(dref[strong]).max(), n
            checked += int(strong.sum())
    assert checked > 1000
    # the second-step loss is compared loosely here (the golden holds only samples of the update); the precise statement --
    # oracle loss + first-order effect of the implementation-dependent +-lr moves, to 2e-4 -- is
    # test_second_step_loss_is_explained_to_first_order
    out2 = tr.step(xg, tg).cpu().numpy()
    close(out2[:4], g["scalars_step2"], 2e-3, 1e-5)


@pytest.mark.parametrize("nc,S,B", [(1, 320, 2), (3, 320, 2)])
def test_second_step_loss_is_explained_to_first_order(nc, S, B):
    """Why post-update quantities cannot be compared at 1e-4 directly, and what CAN be: Adam's first step is
    -lr * g / (|g| + eps) = -lr * sign(g), so an element whose gradient is smaller than the fp32 error of the gradient
    itself moves by +-lr in an implementation-dependent direction (the seven Q2 conv biases in front of BatchNorm are the
    extreme case: their true gradient is 0).  The claims tested here, against the CPU oracle on full tensors:
      (1) only such noise-level elements move differently: every element whose first update differs from the oracle's by
          more than 1 % of lr has |g| <= 2e-3 * max|g| of its tensor (the fp32 gradient tolerance of the other tests), or
          belongs to a Q2 bias, whose whole gradient is below 1e-5 of the global gradient norm;
      (2) the second-step loss equals the oracle's second-step loss plus the FIRST-ORDER effect of exactly those moves,
          L2_ref + <dL2/dp (oracle), p1_hip - p1_ref>, to 2e-4 relative -- the remaining second-order term; the loss terms
          of the first step agree to 1e-4 (north_star's tolerance)."""
    y = api()
    torch.manual_seed(0)
    ref = y.YOLO(num_classes=nc, img_size=S)
    P = {k: v.clone() for k, v in ref.state_dict().items()}
    names = [n for n, _ in ref.named_parameters()]
    x = torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(91))
    tg = y.synthetic_targets(B, nc, S, 8, 92)
    # oracle: step 1 (clip 10 + Adam), then loss and gradient at the updated point
    for n in names:
        P[n].requires_grad_(True)
    l1 = orc.loss_multiscale(orc.forward(P, x, nc, True), tg, orc.anchors_of(P), nc)
    l1[0].backward()
    g1 = {n: P[n].grad.clone() for n in names}
    total, coef = orc.clip_coef([g1[n] for n in names], 10.0)
    p0 = {n: P[n].detach().clone() for n in names}
    with torch.no_grad():
        for n in names:
            orc.adam_step(P[n], g1[n] * coef, torch.zeros_like(P[n]), torch.zeros_like(P[n]), 1, 1e-3)
            P[n].grad = None
    l2 = orc.loss_multiscale(orc.forward(P, x, nc, True), tg, orc.anchors_of(P), nc)
    l2[0].backward()
    # HIP: two fused steps
    m = ref.cuda()
    tr = y.HipTrainer(m, lr=1e-3, max_norm=10.0)
    xg, tgg = x.cuda(), [t.cuda() for t in tg]
    o1 = tr.step(xg, tgg)[:4].cpu().numpy().copy()
    p1_hip = {n: p.detach().cpu().clone() for n, p in m.named_parameters()}
    o2 = tr.step(xg, tgg)[:4].cpu().numpy().copy()
    close(o1, [float(v) for v in l1], 1e-4, 1e-6)
    first_order, n_diff = 0.0, 0
    for n in names:
        d_hip, d_ref = p1_hip[n] - p0[n], P[n].detach() - p0[n]
        diff = (d_hip - d_ref).abs() > 1e-5                               # more than 1 % of lr apart
        if bool(diff.any()):
            if n in Q2:      # the whole tensor is rounding noise (true gradient 0): negligible against the global norm
                assert float(g1[n].abs().max()) <= 1e-5 * total, n
            else:
                assert float(g1[n].abs()[diff].max()) <= 2e-3 * float(g1[n].abs().max()) + 1e-12, n   # claim (1)
            n_diff += int(diff.sum())
        first_order += float(((p1_hip[n] - P[n].detach()).double() * P[n].grad.double()).sum())
    assert n_diff > 0                                                    # the effect exists (else the golden test could be 1e-4)
    want2 = float(l2[0]) + first_order
    assert abs(float(o2[0]) - want2) <= 2e-4 * abs(float(l2[0])), (float(o2[0]), float(l2[0]), first_order)   # claim (2)


def test_training_step_is_bitwise_reproducible():
    y, m, x, targets = _model_and_inputs(1, 320)
    res = []
    for _ in range(2):
        torch.manual_seed(0)
        mm = y.YOLO(num_classes=1, img_size=320).cuda()
        tr = y.HipTrainer(mm, lr=1e-3)
        xg, tg = x[:, :, :320, :320].contiguous().cuda(), [t.cuda() for t in orc.assign_targets(orc.synthetic_boxes(2, 1, 320, 8, 2000), 320, 1)]
        tr.step(xg, tg)
        tr.step(xg, tg)
        res.append((tr.flat_p.clone(), tr.loss_out.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


def test_eval_mode_forward_and_predict_pipeline(tmp_path):
    from PIL import Image
    y = api()
    torch.manual_seed(3)
    m = y.YOLO(num_classes=2, img_size=320)
    m.initialize_detection_biases(prior=0.3)
    P = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.cuda().eval()
    x = torch.rand(1, 3, 320, 320)
    with torch.no_grad():
        preds = m(x.cuda())
        ref = orc.forward(P, x, 2, training=False)
    for a, r in zip(preds, ref):
        assert float((a.cpu() - r).abs().max()) < 1e-4
    img = (np.random.default_rng(0).random((200, 300, 3)) * 255).astype(np.uint8)
    p = tmp_path / "im.png"
    Image.fromarray(img).save(p)
    dets = y.predict(m, str(p), torch.device("cuda"), num_classes=2, conf_threshold=0.29, iou_threshold=0.4)
    assert isinstance(dets, list)
    for d in dets:
        assert len(d) == 6 and isinstance(d[5], int) and 0 <= d[5] < 2 and 0.0 <= d[4] <= 1.0
    for i in range(len(dets)):
        for j in range(i + 1, len(dets)):
            if dets[i][5] == dets[j][5]:
                assert y.compute_iou_corners(dets[i], dets[j]) <= 0.4 + 1e-6


def test_eval_fast_kernels_match_oracle_and_generic_path(monkeypatch):
    """Eval plans at batch sizes that fill the chip route 3x3 stride-1 layers to the Winograd kernel and 1x1 layers to the
    pointwise GEMM, both with the folded-BatchNorm + bias + SiLU (+residual, +upsample) epilogue; layers with few pixels run the
    latency-oriented kernel (K split over the waves of a workgroup), the rest stays on the gather-GEMM.  Same results as the CPU oracle's eval forward and as the
    all-generic routing (YH_EVAL_FAST=0); weights are re-folded when they change."""
    from yolo_from_scratch_amd import _lib as L
    y = api()
    nc, S, B = 2, 320, 16
    torch.manual_seed(3)
    m = y.YOLO(num_classes=nc, img_size=S)
    with torch.no_grad():                                   # non-trivial running statistics and affine parameters
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.running_mean.uniform_(-0.2, 0.2)
                mod.running_var.uniform_(0.5, 1.5)
                mod.weight.uniform_(0.7, 1.3)
                mod.bias.uniform_(-0.1, 0.1)
    P = {k: v.clone() for k, v in m.state_dict().items()}
    x = torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        ref = orc.forward(P, x, nc, training=False)
    m = m.cuda().eval()
    outs = {}
    for fast in ("1", "0"):
        monkeypatch.setenv("YH_EVAL_FAST", fast)
        m._plans.clear()
        with torch.no_grad():
            outs[fast] = [p.clone() for p in m(x.cuda())]
        kinds = [op.kind for op in m._plan_for(x.cuda()).fwd_ops[0][: m._plan_for(x.cuda()).fwd_ops[1]]]
        nw, npw = kinds.count(L.OP_CONV_WINO_FWD_FUSED), kinds.count(L.OP_CONV_PW_FWD_FUSED)
        nrest = kinds.count(L.OP_CONV_FWD_FUSED) + kinds.count(L.OP_CONV_LAT_FWD_FUSED)     # gather-GEMM / few-pixel latency kernel
        assert (nw >= 8 and npw >= 8 and nrest >= 8) if fast == "1" else (nw == 0 and npw == 0 and kinds.count(L.OP_CONV_LAT_FWD_FUSED) == 0)
    for a, b, r in zip(outs["1"], outs["0"], ref):
        scale = float(r.abs().max())
        assert float((a.cpu() - r).abs().max()) < 1e-4 * max(scale, 1.0)
        assert float((a - b).abs().max()) < 1e-4 * max(scale, 1.0)
    # weights change (through torch: version counters; through the package's own kernels: the weights epoch) -> re-fold
    monkeypatch.setenv("YH_EVAL_FAST", "1")
    m._plans.clear()
    with torch.no_grad():
        before = [p.clone() for p in m(x.cuda())]
        m.head_p3[0].conv.weight.mul_(1.5)
        after = [p.clone() for p in m(x.cuda())]
    assert not torch.equal(before[0], after[0]) and torch.equal(before[2], after[2])


def test_cpu_tensor_is_rejected_loudly():
    y = api()
    m = y.ConvBlock(4, 8, 3, 1, 1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 4, 8, 8))


@pytest.mark.parametrize("nc,S,B", [(80, 320, 2), (80, 1280, 1), (1, 640, 64), (80, 640, 64), (80, 1280, 16)])
def test_other_baseline_configs_match_oracle(nc, S, B):
    """BASELINE configs 3/4 at reduced batch: nc=80 heads (255 output channels, scalar-dy kernels) and a
    1280x1280 input, where the loss keeps decoding with img_size=640 (reference quirk Q1) -- and the three configurations
    BASELINE.json names AT THEIR OWN SIZE (nc=1 640 bs=64; nc=80 640 bs=64; nc=80 1280 bs=16) against the pinned CPU oracle
    (train.py:888-926): one fused step, loss terms 1e-4, clipped global gradient norm 3e-4, full gradient tensors of five layers
    3e-3 of their maximum.  The oracle's forward + backward takes a few seconds per case on the box's host cores; these are the
    only places the large-grid kernel variants that batch 64 selects meet the reference arithmetic."""
    y = api()
    torch.manual_seed(0)
    m = y.YOLO(num_classes=nc, img_size=S)
    P = {k: v.clone() for k, v in m.state_dict().items()}
    names = [n for n, _ in m.named_parameters()]
    x = torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(5))
    targets = y.synthetic_targets(B, nc, S, 8, 77)
    m = m.cuda()
    tr = y.HipTrainer(m, lr=1e-3, max_norm=10.0)
    before = tr.flat_g.clone()
    out = tr.step(x.cuda(), [t.cuda() for t in targets]).cpu().numpy()
    for n in names:
        P[n].requires_grad_(True)
    torch.set_num_threads(16)
    ref = orc.loss_multiscale(orc.forward(P, x, nc, True), targets, orc.anchors_of(P), nc)
    close(out[:4], [float(v) for v in ref], 1e-4, 1e-6)
    ref[0].backward()
    total, coef = orc.clip_coef([P[n].grad for n in names], 10.0)
    close(float(tr.norm), total, 3e-4, 0)
    params = dict(m.named_parameters())
    for n in ("head_p3.2.weight", "head_p5.2.bias", "head_p4.0.conv.weight", "stem.0.weight", "merge_p3.conv3.bn.weight"):
        r = P[n].grad * coef
        assert float((params[n].grad.cpu() - r).abs().max()) <= 3e-3 * float(r.abs().max()) + 1e-8, n


def test_inference_session_graph_replay_matches_eager_and_oracle():
    """BASELINE config 5 plumbing: BN-folded fused forward + candidates + NMS under a captured hipGraph give the
    same detections as the eager launch sequence and as the CPU oracle pipeline."""
    y = api()
    torch.manual_seed(11)
    m = y.YOLO(num_classes=3, img_size=320)
    m.initialize_detection_biases(prior=0.3)
    with torch.no_grad():
        for hd in (m.head_p3, m.head_p4, m.head_p5):
            hd[-1].weight.mul_(30.0)
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.running_mean.uniform_(-0.2, 0.2); mod.running_var.uniform_(0.5, 1.5)
                mod.weight.uniform_(0.8, 1.2); mod.bias.uniform_(-0.1, 0.1)
    P = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.cuda().eval()
    imgs = [torch.rand(1, 3, 320, 320, generator=torch.Generator().manual_seed(s)) for s in (1, 2)]
    eager = y.InferenceSession(m, conf_threshold=0.3, iou_threshold=0.4, use_graph=False)
    graph = y.InferenceSession(m, conf_threshold=0.3, iou_threshold=0.4, use_graph=True)
    for k, img in enumerate(imgs):
        lb = (3.0 * k, 1.0 * k, 1.0 - 0.25 * k)
        de = eager.run(img, *lb)
        dg = graph.run(img, *lb)
        assert len(de) > 5 and de == dg
        with torch.no_grad():
            preds = orc.forward(P, img, 3, training=False)
        with torch.no_grad():
            hip_preds = m(img.cuda())
        for a, r in zip(hip_preds, preds):
            assert float((a.cpu() - r).abs().max()) < 2e-3 * max(1.0, float(r.abs().max()))
        # kept indices bit-exact on the GPU's own candidates, and every difference to the oracle pipeline accounted for
        # (threshold cells, IoU pairs straddling the NMS threshold): tests/nms_explain.py
        from nms_explain import explain_detections
        explain_detections(graph.det, hip_preds, preds, orc.anchors_of(P), 320, 3, 0.3, 0.4, lb)


def test_predict_batch_matches_oracle_pipeline_per_image(tmp_path):
    """predict_batch (one batched forward, per-image candidate / NMS segments) against the ORACLE pipeline run image by
    image -- letterbox, eval forward, candidates, class-aware NMS (train.py:1114-1250) -- not against predict()."""
    from PIL import Image
    from nms_explain import explain_detections
    y = api()
    nc, S = 2, 320
    torch.manual_seed(13)
    m = y.YOLO(num_classes=nc, img_size=S)
    m.initialize_detection_biases(prior=0.3)
    with torch.no_grad():
        for hd in (m.head_p3, m.head_p4, m.head_p5):
            hd[-1].weight.mul_(25.0)
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.running_mean.uniform_(-0.2, 0.2); mod.running_var.uniform_(0.5, 1.5)
                mod.weight.uniform_(0.8, 1.2); mod.bias.uniform_(-0.1, 0.1)
    P = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.cuda().eval()
    rng = np.random.default_rng(3)
    paths = []
    for k, (h, w) in enumerate(((200, 300), (320, 320), (400, 250), (90, 333))):
        pth = tmp_path / f"im{k}.png"
        Image.fromarray((rng.random((h, w, 3)) * 255).astype(np.uint8)).save(pth)
        paths.append(str(pth))
    dets = y.predict_batch(m, paths, torch.device("cuda"), num_classes=nc, conf_threshold=0.3, iou_threshold=0.4)
    assert len(dets) == len(paths)
    total_kept = 0
    for b, pth in enumerate(paths):
        pil, scale, pad_top, pad_left = y.letterbox_resize(Image.open(pth).convert("RGB"), S)
        x = (torch.from_numpy(np.array(pil)).permute(2, 0, 1).float() / 255.0).unsqueeze(0)
        with torch.no_grad():
            ref = orc.forward(P, x, nc, training=False)
            hip = m(x.cuda())
        det = m._detectors[b]
        explain_detections(det, hip, ref, orc.anchors_of(P), S, nc, 0.3, 0.4, (pad_left, pad_top, scale))
        # the returned tuples are the kept candidates, in order, in original-image pixels
        k = int(det.nkeep.item())
        idx = det.keep[:k].long()
        assert len(dets[b]) == k
        got = np.array([d[:5] for d in dets[b]], np.float32).reshape(-1, 5)
        np.testing.assert_array_equal(got[:, :4], det.boxes[idx].cpu().numpy())
        np.testing.assert_array_equal(got[:, 4], det.scores[idx].cpu().numpy())
        assert [d[5] for d in dets[b]] == det.classes[idx].cpu().tolist()
        total_kept += k
    assert total_kept > 20


def test_fused_and_unfused_plans_agree(monkeypatch):
    """Round 4: consumers that apply their producer's BatchNorm + SiLU while staging (virtual layers: the normalised tensor never
    exists) against the plan of the same model with every BatchNorm + SiLU pass launched (YH_FUSE_ACT=0, read when the plan is
    traced).  Different kernels and summation orders, the same arithmetic: losses to 2e-6 relative, the whole gradient vector to 1e-4
    relative (norm of the difference), every parameter tensor's gradient to cosine 1 - 1e-6; and the switch really switches."""
    y = api()
    nc, S, B = 3, 320, 4
    x = torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(31)).cuda()
    tg = [t.cuda() for t in y.synthetic_targets(B, nc, S, 8, 32)]
    runs = {}
    for fuse in ("1", "0"):
        monkeypatch.setenv("YH_FUSE_ACT", fuse)
        torch.manual_seed(0)
        m = y.YOLO(num_classes=nc, img_size=S).cuda()
        tr = y.HipTrainer(m, lr=1e-3, max_norm=None)
        out = tr.step(x, tg)[:4].cpu().double()
        plan = m._plan_for(x)
        nvirt = sum(1 for r in plan.recs if getattr(r, "virtual", False))
        grads = {n: p.grad.detach().double().reshape(-1).cpu().clone() for n, p in m.named_parameters() if p.grad is not None}
        if not grads:                                     # the fused trainer keeps gradients in its flat buffer
            off, grads = 0, {}
            g = tr.flat_g.detach().double().cpu()
            grads = {"flat": g.clone()}
        runs[fuse] = (out, nvirt, grads)
    monkeypatch.delenv("YH_FUSE_ACT")
    assert runs["1"][1] >= 30 and runs["0"][1] == 0, (runs["1"][1], runs["0"][1])
    a, b = runs["1"][0], runs["0"][0]
    assert float(((a - b).abs() / b.abs().clamp_min(1e-12)).max()) <= 2e-6, (a, b)
    for n in runs["1"][2]:
        ga, gb = runs["1"][2][n], runs["0"][2][n]
        assert float((ga - gb).norm()) <= 1e-4 * float(gb.norm()) + 1e-12, n
        if float(gb.norm()) > 1e-9:
            assert float(ga @ gb / (ga.norm() * gb.norm())) >= 1 - 1e-6, n


def test_side_lanes_do_not_change_results_and_trajectory_tracks_oracle():
    """(i) fork/join lanes are a pure scheduling change: parameters after 3 steps are bitwise identical with the
    side stream disabled; (ii) a 3-step trajectory (clip + Adam each step) tracks the CPU oracle's losses."""
    from yolo_from_scratch_amd import _lib as L
    y = api()
    nc, S, B = 1, 320, 2
    x = torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(21))
    targets = y.synthetic_targets(B, nc, S, 8, 22)
    runs = []
    for overlap in (1, 0):
        L.set_overlap(overlap)
        torch.manual_seed(0)
        m = y.YOLO(num_classes=nc, img_size=S).cuda()
        tr = y.HipTrainer(m, lr=1e-3, max_norm=10.0)
        losses = [tr.step(x.cuda(), [t.cuda() for t in targets])[:4].cpu().clone() for _ in range(3)]
        runs.append((tr.flat_p.clone(), torch.stack(losses)))
    L.set_overlap(1)
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    torch.manual_seed(0)
    ref = y.YOLO(num_classes=nc, img_size=S)
    P = {k: v.clone() for k, v in ref.state_dict().items()}
    params = [P[n].requires_grad_(True) for n, _ in ref.named_parameters()]
    opt = torch.optim.Adam(params, lr=1e-3)
    want = []
    for _ in range(3):
        opt.zero_grad()
        out = orc.loss_multiscale(orc.forward(P, x, nc, True), targets, orc.anchors_of(P), nc)
        out[0].backward()
        torch.nn.utils.clip_grad_norm_(params, 10.0)
        opt.step()
        want.append([float(v) for v in out])
    got = runs[0][1].numpy()
    np.testing.assert_allclose(got[0], want[0], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(got[1:], want[1:], rtol=5e-3, atol=1e-5)      # Adam's sign-like first steps amplify fp32 noise


def test_long_run_stays_finite_and_learns():
    y = api()
    torch.manual_seed(0)
    m = y.YOLO(num_classes=2, img_size=256).cuda()
    tr = y.HipTrainer(m, lr=2e-3, max_norm=10.0)
    x = torch.rand(4, 3, 256, 256, generator=torch.Generator().manual_seed(5)).cuda()
    tg = [t.cuda() for t in y.synthetic_targets(4, 2, 256, 6, 6)]
    hist = torch.stack([tr.step(x, tg)[:4].clone() for _ in range(150)]).cpu()
    assert torch.isfinite(hist).all() and torch.isfinite(tr.flat_p).all()
    assert float(hist[-10:, 0].mean()) < 0.5 * float(hist[:5, 0].mean())       # overfits the fixed batch


def test_full_size_step_properties(monkeypatch):
    """BASELINE configs[1] size (bs=64, 640x640, nc=1); parity with the oracle at this size is test_other_baseline_configs_match_oracle
    [1-640-64]; here: (i) the step is bitwise
    reproducible (every reduction has a fixed order, also in the large-grid kernel variants only this size selects);
    (ii) the specialised kernels (Winograd, pointwise, stem, merged stride-2) and the generic gather-GEMM / wgrad kernels
    are two independent product paths and agree: same loss to 1e-5, gradients to 5e-3 of each tensor's max (two fp32
    summation orders over up to 6.5 M pixels per weight, each within ~2e-3 of the exact sum)."""
    y = api()
    B, S, nc = 64, 640, 1
    x = torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(31)).cuda()
    tg = [t.cuda() for t in y.synthetic_targets(B, nc, S, 8, 32)]

    def run(generic):
        monkeypatch.setenv("YH_GENERIC", "1" if generic else "0")
        torch.manual_seed(0)
        m = y.YOLO(num_classes=nc, img_size=S).cuda()
        tr = y.HipTrainer(m, lr=1e-3, max_norm=10.0)
        loss = tr.step(x, tg)[:4].cpu().clone()
        grads = {n: p.grad.detach().clone() if p.grad is not None else None for n, p in m.named_parameters()}
        flat_g = tr.flat_g.clone()
        return loss, flat_g, tr.flat_p.clone(), grads

    a, b, c = run(False), run(False), run(True)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    assert torch.isfinite(a[0]).all() and torch.isfinite(a[1]).all()
    np.testing.assert_allclose(a[0].numpy(), c[0].numpy(), rtol=1e-5, atol=1e-7)
    checked = 0
    for (n, ga), gc in zip(a[3].items(), c[3].values()):
        if ga is None or ".bias" in n and "bn" not in n and "head" not in n:
            continue                                     # conv biases in front of BatchNorm: true gradient 0, rounding noise
        scale = float(gc.abs().max())
        if scale > 0:
            assert float((ga - gc).abs().max()) <= 5e-3 * scale, n
            checked += 1
    assert checked > 150

