"""CPU-side checks: the C-ABI library loads and exports every symbol include/yolohip.h declares, the
module tree reproduces the reference's state-dict, and nothing silently falls back to the CPU."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import load_golden


def test_library_exports_every_declared_symbol():
    from yolo_from_scratch_amd import _lib
    names = _lib.declared_symbols()
    assert len(names) >= 35 and "yh_conv_fwd" in names and "yh_nms" in names
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), f"{n} declared in yolohip.h but not exported"
    assert set(_lib._SIGS) == set(names)                   # the python binding covers the whole ABI
    assert _lib.lib().yh_version() >= 100
    assert ctypes.sizeof(_lib.YhOp) == 4 + 80 + 16 + 4 + 96 + 16   # struct yh_op layout (with padding)


def test_state_dict_contract_matches_reference():
    import yolo_from_scratch_amd as y
    g = load_golden("model_nc1")
    torch.manual_seed(0)
    m = y.YOLO(num_classes=1, img_size=640)
    sd = m.state_dict()
    assert len(sd) == 377 and list(sd.keys()) == list(g["keys"])
    assert sum(p.numel() for p in m.parameters()) == 3662838
    assert [v.numel() for v in sd.values()] == list(g["numel"])
    np.testing.assert_array_equal([float(v.double().sum()) for v in sd.values()], g["init_sum"])
    # grid buffers are contiguous so checkpoints round-trip (reference quirk Q5)
    m2 = y.YOLO(num_classes=1)
    m2.load_state_dict(sd)
    assert all(v.is_contiguous() for v in sd.values())
    assert m.head_p3[-1].bias is not None and abs(float(m.head_p3[-1].bias[4]) + 4.59512) < 1e-4
    assert [tuple(a.shape) for a in m.anchors] == [(3, 2)] * 3
    assert (m.grid_size_p3, m.grid_size_p4, m.grid_size_p5, m.output_channels) == (80, 40, 20, 18)


@pytest.mark.parametrize("size,nc", [("n", 2), ("m", 1)])
def test_other_sizes_construct(size, nc):
    import yolo_from_scratch_amd as y
    wm, dm = y.YOLO_SIZES[size]
    m = y.YOLO(num_classes=nc, img_size=320, width_mult=wm, depth_mult=dm)
    assert m.head_p5[-1].out_channels == 3 * (5 + nc)


def test_product_refuses_cpu_tensors():
    import yolo_from_scratch_amd as y
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        y.YOLO(num_classes=1, img_size=64)(torch.zeros(1, 3, 64, 64))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        y.decode_predictions(torch.zeros(1, 2, 2, 3, 6), torch.ones(3, 2))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        y.batched_nms(torch.zeros(1, 4), torch.zeros(1), torch.zeros(1, dtype=torch.long), 0.5)


def test_host_side_helpers_known_answers():
    import yolo_from_scratch_amd as y
    g = load_golden("nms")
    ka = [tuple(r) for r in g["ka3/dets"]]
    assert [ka.index(d) for d in y.nms(ka, 0.5)] == [0, 2]
    assert y.nms([], 0.5) == []
    assert abs(y.compute_iou_corners((0, 0, 10, 10), (5, 0, 15, 10)) - 1 / 3) < 1e-12
    lam = y.get_lr_lambda(3, 100, 1e-2, 1e-4, 1e-6)
    assert abs(lam(0) * 1e-2 - 1e-6) < 1e-12 and abs(lam(3) - 1.0) < 1e-9 and abs(lam(100) * 1e-2 - 1e-4) < 1e-9
    a = torch.tensor([0.5, 0.5, 0.2, 0.2])
    assert abs(float(y.compute_box_iou(a, a)) - 0.04 / (0.04 + 1e-6)) < 1e-6


def test_synthetic_targets_follow_the_reference_assignment_rule():
    """The bench's synthetic-target generator (product, host side) against the pinned oracle rule."""
    import numpy as np, math
    import yolo_from_scratch_amd as y
    from oracle import yolo_oracle as orc
    for nc in (1, 3):
        got = y.synthetic_targets(3, nc, 320, 8, 2000)
        rng = np.random.default_rng(2000)
        boxes = []
        for _ in range(3):
            c = rng.uniform(0.05, 0.95, size=(8, 2))
            wh = np.exp(rng.uniform(math.log(8), math.log(320), size=(8, 2))) / 320
            k = rng.integers(0, max(nc, 1), size=8)
            boxes.append([(int(k[i]), float(c[i, 0]), float(c[i, 1]), float(wh[i, 0]), float(wh[i, 1])) for i in range(8)])
        want = orc.assign_targets(boxes, 320, nc)
        for a, b in zip(got, want):
            assert torch.equal(a, b)


def test_checkpoint_round_trip_and_reference_style_grid_buffers(tmp_path):
    """the reference's checkpoint dict round-trips, including a file whose grid_* buffers are
    the reference's stride-0 expanded views (quirk Q5)."""
    import yolo_from_scratch_amd as y
    torch.manual_seed(2)
    m = y.YOLO(num_classes=3, img_size=320, width_mult=0.25, depth_mult=0.33)
    p = tmp_path / "yolo.pt"
    y.save_checkpoint(m, 7, p)
    ck = torch.load(p, weights_only=False)
    assert set(ck) == {"model", "epoch", "num_classes", "img_size", "width_mult", "depth_mult"} and ck["epoch"] == 7
    m2, ep = y.load_checkpoint(p)
    assert ep == 7 and m2.num_classes == 3 and m2.img_size == 320
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    sd = dict(ck["model"])
    g = 320 // 8
    gy, gx = torch.meshgrid(torch.arange(g, dtype=torch.float32), torch.arange(g, dtype=torch.float32), indexing="ij")
    sd["grid_x_p3"] = torch.arange(g, dtype=torch.float32).view(1, 1, g, 1).expand(1, g, g, 1)     # stride-0 view
    assert not sd["grid_x_p3"].is_contiguous()
    torch.save({**ck, "model": sd}, p)
    m3, _ = y.load_checkpoint(p)
    assert torch.equal(m3.grid_x_p3, gx.view(1, g, g, 1))


def test_fused_step_rejects_mismatched_targets():
    """HipTrainer.step hands raw pointers to yh_yolo_loss: targets of the wrong device / dtype / shape must raise
    before anything is launched (the reference raises a shape error in its loss, train.py:806-830)."""
    from yolo_from_scratch_amd.training import check_targets
    shapes = [(2, 8, 8, 3, 6), (2, 4, 4, 3, 6), (2, 2, 2, 3, 6)]
    good = [torch.zeros(s) for s in shapes]
    with pytest.raises(ValueError, match="no CPU fallback"):           # host tensors
        check_targets(good, shapes, "cuda:0")
    with pytest.raises(ValueError, match="expected 3 target tensors"):
        check_targets(good[:2], shapes, "cuda:0")
    meta = [torch.empty(s, device="meta") for s in shapes]             # shape / dtype rules, checked without a GPU
    import unittest.mock as mock

    def as_cuda(t, dtype=torch.float32, shape=None):
        m = mock.MagicMock(spec=torch.Tensor)
        m.device, m.dtype, m.shape = torch.device("cuda:0"), dtype, torch.Size(shape if shape is not None else t.shape)
        m.contiguous.return_value = m
        return m

    ok = check_targets([as_cuda(t) for t in meta], shapes, "cuda:0")
    assert len(ok) == 3
    with pytest.raises(ValueError, match="float32"):
        check_targets([as_cuda(meta[0], torch.float64)] + [as_cuda(t) for t in meta[1:]], shapes, "cuda:0")
    with pytest.raises(ValueError, match="has shape"):                 # smaller last batch vs the plan's batch
        check_targets([as_cuda(meta[0], shape=(1, 8, 8, 3, 6))] + [as_cuda(t) for t in meta[1:]], shapes, "cuda:0")
    with pytest.raises(ValueError, match="has shape"):                 # built for another num_classes
        check_targets([as_cuda(meta[0], shape=(2, 8, 8, 3, 8))] + [as_cuda(t) for t in meta[1:]], shapes, "cuda:0")


def test_contexts_are_independent_handles():
    """the only state between calls lives in the explicit yh_context.  Two handles share nothing: own
    overlap flag, own (lazily created) side stream / events; yh_create makes no HIP call so this runs without a GPU.
    The GPU half (distinct stream / event handles once bound, two host threads on two streams) is
    tests/test_gpu_api_surface.py::test_two_contexts_two_streams."""
    from yolo_from_scratch_amd import _lib as L
    a, b = L.Context(), L.Context()
    assert a.handle.value != b.handle.value
    assert a.info() == {"device": -1, "overlap": 1, "side_stream": None, "fork_event": None, "join_event": None}   # YH_OVERLAP is read at yh_create
    a.set_overlap(False)
    assert a.info()["overlap"] == 0 and b.info()["overlap"] == 1           # b untouched
    b.set_overlap(True)
    assert a.info()["overlap"] == 0 and b.info()["overlap"] == 1
    # an empty list, and a list of lane markers only, run without touching a device
    ops = (L.YhOp * 2)()
    ops[0].kind, ops[1].kind = L.OP_FORK, L.OP_JOIN
    L.run_ops(ops, 0, 0, a)
    L.run_ops(ops, 2, 0, a)                  # overlap off: markers are skipped
    L.run_ops(ops, 2, 0, None)               # no context: serial
    # a failing op reports its index and the message; the context stays usable
    bad = (L.YhOp * 2)()
    bad[0].kind, bad[1].kind = L.OP_NOP, 9999
    with pytest.raises(RuntimeError, match=r"op #1 \(kind 9999\).*unknown op kind"):
        L.run_ops(bad, 2, 0, a)
    L.run_ops(ops, 2, 0, a)
    assert L.context_for(0) is L.context_for(0) and L.context_for(0) is not L.context_for(1)
    a.close(); b.close()
    assert a.handle is None


def test_replacing_a_parameter_or_buffer_bumps_the_generation():
    """plans / captured hipGraphs bake raw addresses in; re-assigning a Parameter or buffer attribute (also through
    load_state_dict(assign=True) on a plain nn.Module parent) must be visible to the cheap per-image state token."""
    import torch
    from yolo_from_scratch_amd import graph as G
    y = __import__("yolo_from_scratch_amd")
    m = y.ConvBlock(4, 8, 3, 1, 1)
    g0 = G.PARAM_GENERATION[0]
    m.conv.weight = torch.nn.Parameter(torch.zeros_like(m.conv.weight))
    assert G.PARAM_GENERATION[0] == g0 + 1
    m.bn.running_mean = torch.ones(8)
    assert G.PARAM_GENERATION[0] == g0 + 2
    parent = torch.nn.Sequential(m)
    sd = {k: v.clone() for k, v in parent.state_dict().items()}
    g1 = G.PARAM_GENERATION[0]
    parent.load_state_dict(sd, assign=True)
    assert G.PARAM_GENERATION[0] > g1
    g2 = G.PARAM_GENERATION[0]
    torch.nn.Linear(3, 3)                      # constructing modules (first registration) does not count
    assert G.PARAM_GENERATION[0] == g2
