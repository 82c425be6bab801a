"""GPU parity of the Winograd F(2x2,3x3) kernels (forward, backward-data, backward-weight) through the C ABI
against an fp64 torch evaluation of the same convolution (train.py:260-265, 300-306 and their autograd).
Tolerance: 1e-4 relative to the tensor's max magnitude (measured ~5e-7); the direct kernel is checked beside it
so that the two product paths are also compared with each other."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _lib():
    import yolo_from_scratch_amd._lib as L
    return L


def rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def rup4(c):
    return (c + 3) // 4 * 4


CASES = [  # (B, H, W, Cin, Cout)
    (2, 16, 16, 32, 32),      # one 32-column tile (NT = 1 kernels)
    (1, 16, 16, 64, 64),
    (2, 8, 8, 128, 128),
    (1, 6, 6, 256, 256),      # 9 tiles per image: ragged last workgroup
    (3, 10, 14, 64, 128),     # non-square, tile rows of 7
    (1, 12, 12, 48, 96),      # K % 16 == 0 but not a power of two; partial 64-column tile (forward / backward-data only)
    (2, 4, 4, 16, 16),        # smallest legal map (tile row width 2)
]


@pytest.mark.parametrize("case", CASES)
def test_winograd_forward_and_backward_data(case):
    L = _lib()
    lib = L.lib()
    B, H, W, Cin, Cout = case
    torch.manual_seed(sum(case))
    x = torch.randn(B, Cin, H, W)
    w = torch.randn(Cout, Cin, 3, 3) / (Cin * 9) ** 0.5
    bias = torch.randn(Cout)
    dy = torch.randn(B, Cout, H, W)
    ref = F.conv2d(x.double(), w.double(), bias.double(), 1, 1)
    dref = F.conv_transpose2d(dy.double(), w.double(), None, 1, 1)
    st = torch.cuda.current_stream().cuda_stream
    wd = w.cuda()
    ldu, ldub = rup4(Cout), rup4(Cin)
    U = torch.empty(16 * Cin * ldu, device="cuda")
    Ub = torch.empty(16 * Cout * ldub, device="cuda")
    L.check(lib.yh_wino_weights(wd.data_ptr(), U.data_ptr(), Cout, Cin, ldu, 0, st))
    L.check(lib.yh_wino_weights(wd.data_ptr(), Ub.data_ptr(), Cout, Cin, ldub, 1, st))
    # the input lives in a channel slice of a wider buffer (ld > C), like the concat buffers of the network
    ldx, offx = Cin + 8, 4
    xbuf = torch.full((B, H, W, ldx), 7.0, device="cuda")
    xbuf[..., offx:offx + Cin] = x.permute(0, 2, 3, 1).cuda()
    ldy, offy = Cout + 4, 4
    ybuf = torch.full((B, H, W, ldy), -3.0, device="cuda")
    nblk = lib.yh_conv_wino_blocks(B, H, W)
    part = torch.zeros(nblk, 2, Cout, device="cuda")
    L.check(lib.yh_conv_wino_fwd(xbuf.data_ptr() + 4 * offx, ldx, U.data_ptr(), ldu, bias.cuda().data_ptr(),
                                 ybuf.data_ptr() + 4 * offy, ldy, part.data_ptr(), B, H, W, Cin, Cout, st))
    y = ybuf[..., offy:offy + Cout].permute(0, 3, 1, 2)
    assert rel_err(y, ref) < 1e-4
    assert float(ybuf[..., :offy].min()) == -3.0 == float(ybuf[..., :offy].max())     # neighbours untouched
    # BatchNorm partial sums (same contract as yh_conv_fwd): per-channel sum and sum of squares of the OUTPUT
    s = part.double().sum(0).cpu()
    yd = y.double().cpu()
    assert float((s[0] - yd.sum((0, 2, 3))).abs().max() / (yd.sum((0, 2, 3)).abs().max() + 1e-30)) < 1e-5
    assert float((s[1] - (yd * yd).sum((0, 2, 3))).abs().max() / (yd * yd).sum((0, 2, 3)).abs().max()) < 1e-5
    # no bias, no statistics
    y2 = torch.empty(B, H, W, Cout, device="cuda")
    L.check(lib.yh_conv_wino_fwd(xbuf.data_ptr() + 4 * offx, ldx, U.data_ptr(), ldu, None, y2.data_ptr(), Cout, None,
                                 B, H, W, Cin, Cout, st))
    assert rel_err(y2.permute(0, 3, 1, 2), F.conv2d(x.double(), w.double(), None, 1, 1)) < 1e-4

    # backward-data: write, then accumulate on top of existing contents
    dyd = dy.permute(0, 2, 3, 1).contiguous().cuda()
    dxbuf = torch.full((B, H, W, ldx), 5.0, device="cuda")
    L.check(lib.yh_conv_wino_bwd_data(dyd.data_ptr(), Cout, Ub.data_ptr(), ldub, dxbuf.data_ptr() + 4 * offx, ldx, B, H, W,
                                      Cin, Cout, 0, st))
    dx = dxbuf[..., offx:offx + Cin].permute(0, 3, 1, 2)
    assert rel_err(dx, dref) < 1e-4
    assert float(dxbuf[..., :offx].min()) == 5.0
    L.check(lib.yh_conv_wino_bwd_data(dyd.data_ptr(), Cout, Ub.data_ptr(), ldub, dxbuf.data_ptr() + 4 * offx, ldx, B, H, W,
                                      Cin, Cout, 1, st))
    assert rel_err(dxbuf[..., offx:offx + Cin].permute(0, 3, 1, 2), 2 * dref) < 1e-4

    # the direct kernel on the same operands: the two product paths agree
    ldwf = rup4(Cout)
    wf = torch.empty(9 * Cin * ldwf, device="cuda")
    wb = torch.empty(9 * Cout * Cin, device="cuda")
    L.check(lib.yh_pack_weights(wd.data_ptr(), wf.data_ptr(), wb.data_ptr(), Cout, Cin, 3, Cin, ldwf, Cin, st))
    y3 = torch.empty(B, H, W, Cout, device="cuda")
    L.check(lib.yh_conv_fwd(xbuf.data_ptr() + 4 * offx, ldx, wf.data_ptr(), ldwf, None, y3.data_ptr(), Cout, None, B, H, W,
                            Cin, Cout, 3, 1, st))
    assert rel_err(y2, y3) < 1e-5


@pytest.mark.parametrize("case", [c for c in CASES if c[3] % 32 == 0 and c[4] % 32 == 0])
def test_winograd_backward_weight(case):
    L = _lib()
    lib = L.lib()
    B, H, W, Cin, Cout = case
    torch.manual_seed(sum(case) + 1)
    x = torch.randn(B, Cin, H, W)
    dy = torch.randn(B, Cout, H, W)
    wref = torch.nn.grad.conv2d_weight(x.double(), (Cout, Cin, 3, 3), dy.double(), padding=1)
    st = torch.cuda.current_stream().cuda_stream
    ldx, offx = Cin + 4, 4
    xbuf = torch.full((B, H, W, ldx), 9.0, device="cuda")
    xbuf[..., offx:offx + Cin] = x.permute(0, 2, 3, 1).cuda()
    dyd = dy.permute(0, 2, 3, 1).contiguous().cuda()
    nws = lib.yh_conv_wino_bwd_weight_ws(B, H, W, Cin, Cout)
    assert nws > 0
    ws = torch.empty(nws, device="cuda")
    dw = torch.zeros(Cout, Cin, 3, 3, device="cuda")
    L.check(lib.yh_conv_wino_bwd_weight(xbuf.data_ptr() + 4 * offx, ldx, dyd.data_ptr(), Cout, dw.data_ptr(), ws.data_ptr(),
                                        nws, B, H, W, Cin, Cout, st))
    assert rel_err(dw, wref) < 1e-4
    # deterministic: a second run with a dirty workspace gives the same bits
    dw2 = torch.zeros_like(dw)
    ws.fill_(123.0)
    L.check(lib.yh_conv_wino_bwd_weight(xbuf.data_ptr() + 4 * offx, ldx, dyd.data_ptr(), Cout, dw2.data_ptr(), ws.data_ptr(),
                                        nws, B, H, W, Cin, Cout, st))
    assert torch.equal(dw, dw2)
    # too small a workspace is refused, nothing is launched
    rc = lib.yh_conv_wino_bwd_weight(xbuf.data_ptr() + 4 * offx, ldx, dyd.data_ptr(), Cout, dw2.data_ptr(), ws.data_ptr(),
                                     nws - 1, B, H, W, Cin, Cout, st)
    assert rc != 0 and b"workspace" in lib.yh_last_error()


def test_winograd_full_size_layer_properties():
    """BASELINE-size layer (bs=64, 80x80, 64->64): linearity in the input and agreement with the direct kernel --
    size-independent properties, no CPU reference at this size."""
    L = _lib()
    lib = L.lib()
    B, H, W, C = 64, 80, 80, 64
    torch.manual_seed(5)
    st = torch.cuda.current_stream().cuda_stream
    w = (torch.randn(C, C, 3, 3) / (C * 9) ** 0.5).cuda()
    U = torch.empty(16 * C * C, device="cuda")
    L.check(lib.yh_wino_weights(w.data_ptr(), U.data_ptr(), C, C, C, 0, st))
    wf, wb = torch.empty(9 * C * C, device="cuda"), torch.empty(9 * C * C, device="cuda")
    L.check(lib.yh_pack_weights(w.data_ptr(), wf.data_ptr(), wb.data_ptr(), C, C, 3, C, C, C, st))
    x1, x2 = torch.randn(B, H, W, C, device="cuda"), torch.randn(B, H, W, C, device="cuda")

    def wino(x):
        y = torch.empty(B, H, W, C, device="cuda")
        L.check(lib.yh_conv_wino_fwd(x.data_ptr(), C, U.data_ptr(), C, None, y.data_ptr(), C, None, B, H, W, C, C, st))
        return y
    y1, y2, y12 = wino(x1), wino(x2), wino(x1 + 2 * x2)
    assert rel_err(y12, y1 + 2 * y2) < 1e-5
    yd = torch.empty(B, H, W, C, device="cuda")
    L.check(lib.yh_conv_fwd(x1.data_ptr(), C, wf.data_ptr(), C, None, yd.data_ptr(), C, None, B, H, W, C, C, 3, 1, st))
    assert rel_err(y1, yd) < 1e-5
    # backward-weight at full size against the direct kernel
    nws = max(lib.yh_conv_wino_bwd_weight_ws(B, H, W, C, C), lib.yh_conv_bwd_weight_ws(B, H, W, C, C, 3, 1))
    ws = torch.empty(nws, device="cuda")
    dw1, dw2 = torch.zeros_like(w), torch.zeros_like(w)
    L.check(lib.yh_conv_wino_bwd_weight(x1.data_ptr(), C, x2.data_ptr(), C, dw1.data_ptr(), ws.data_ptr(), nws, B, H, W, C, C, st))
    L.check(lib.yh_conv_bwd_weight(x1.data_ptr(), C, x2.data_ptr(), C, dw2.data_ptr(), ws.data_ptr(), nws, B, H, W, C, C, C, 3, 1, st))
    assert rel_err(dw1, dw2) < 1e-4


def test_winograd_rejects_unsupported_shapes():
    L = _lib()
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    t = torch.zeros(1 << 16, device="cuda")
    p = t.data_ptr()
    assert lib.yh_conv_wino_fwd(p, 16, p, 16, None, p, 16, None, 1, 5, 4, 16, 16, st) != 0      # odd H
    assert b"even" in lib.yh_last_error()
    assert lib.yh_conv_wino_fwd(p, 8, p, 8, None, p, 8, None, 1, 4, 4, 8, 8, st) != 0          # K % 16
    assert lib.yh_conv_wino_bwd_weight(p, 16, p, 16, p, p, 1 << 16, 1, 4, 4, 16, 16, st) != 0     # channels % 32
    assert lib.yh_conv_wino_bwd_weight_ws(1, 4, 4, 16, 16) < 0
