"""GPU parity of the bf16 C-ABI kernels (BASELINE configs 3-4) against fp64 torch CPU evaluations of the same op ON
THE SAME bf16-ROUNDED INPUTS: the kernels multiply bf16 operands exactly and accumulate in fp32, so what is left is
(i) fp32 summation order (1e-5 of the tensor's max) and (ii) ONE bf16 rounding of each stored output (2^-9 relative per
element) -- tolerance 5e-3 of the tensor's max for bf16 outputs, 2e-4 for fp32 outputs (weight gradients, BN sums)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF16_OUT_TOL = 5e-3     # one bf16 rounding (8 mantissa bits: 2^-9 = 2e-3 relative) + fp32 accumulation order
F32_OUT_TOL = 2e-4


def _lib():
    import yolo_from_scratch_amd._lib as L
    return L


def rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def rup8(c):
    return (c + 7) // 8 * 8


def bf(t):          # round to bf16 and come back (the value the kernel sees)
    return t.to(torch.bfloat16).to(torch.float32)


def nhwc_bf16(t, cpad=None, ld=None, off=0):
    """NCHW fp32 cpu -> NHWC bf16 cuda view with `cpad` channels (zero padded) inside a buffer of ld channels."""
    B, C, H, W = t.shape
    cpad = cpad or C
    ld = ld or cpad
    buf = torch.full((B, H, W, ld), 3.0, dtype=torch.bfloat16, device="cuda")
    buf[..., off:off + cpad] = 0
    buf[..., off:off + C] = t.permute(0, 2, 3, 1).to(torch.bfloat16).cuda()
    return buf


def pack(L, w, cin_pad, koff=0, kpad=None, wb_buf=None, want_b=True):
    """OIHW fp32 cuda -> (wf, ldf, wb, ldb) through yh_bf16_pack_multi."""
    import struct
    Cout, Cin, k, _ = w.shape
    ldf, ldb = rup8(Cout), rup8(cin_pad)
    kpad = kpad or rup8(Cout)
    wf = torch.empty(k * k * cin_pad * ldf, dtype=torch.bfloat16, device="cuda")
    wb = wb_buf if wb_buf is not None else torch.zeros(k * k * kpad * ldb, dtype=torch.bfloat16, device="cuda")
    rec = struct.pack("<QQQiiiiiiii", w.data_ptr(), wf.data_ptr(), wb.data_ptr() if want_b else 0, Cout, Cin, k * k, cin_pad,
                      ldf, ldb, koff, kpad)
    tab = torch.frombuffer(bytearray(rec), dtype=torch.uint8).cuda()
    L.check(L.lib().yh_bf16_pack_multi(tab.data_ptr(), 1, torch.cuda.current_stream().cuda_stream), "bf16_pack")
    torch.cuda.synchronize()
    return wf, ldf, wb, ldb


CONV_CASES = [  # (B, H, W, Cin, Cout, k, s, bias)
    (2, 32, 32, 3, 16, 3, 2, True),      # stem.0 (Cin padded to 8)
    (2, 24, 24, 16, 32, 3, 2, True),     # stem.3
    (2, 20, 20, 32, 16, 1, 1, False),    # C3 1x1 small
    (2, 20, 20, 16, 16, 3, 1, False),    # bottleneck 16
    (1, 16, 16, 64, 64, 3, 1, False),    # head 3x3
    (1, 8, 8, 128, 128, 3, 1, False),
    (1, 6, 6, 256, 256, 3, 1, False),
    (1, 10, 10, 192, 64, 1, 1, False),   # panet conv (non power-of-two Cin)
    (1, 6, 6, 512, 256, 1, 1, True),     # sppf.conv2
    (2, 12, 12, 64, 64, 3, 2, False),    # downsample
    (2, 9, 11, 32, 64, 3, 2, True),      # odd sizes, stride 2
    (2, 10, 10, 64, 18, 1, 1, True),     # head out nc=1 (fp32 output)
    (1, 10, 10, 128, 255, 1, 1, True),   # head out nc=80 (fp32 output)
    (3, 7, 5, 8, 24, 3, 1, False),       # tiny odd
    (1, 40, 40, 32, 32, 3, 1, False),    # M not a multiple of the tile
    (2, 12, 100, 16, 32, 3, 1, False),   # wide rows: several wgrad segments per row
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_bf16_conv_fwd_dgrad_wgrad(case):
    L = _lib()
    lib = L.lib()
    B, H, W, Cin, Cout, k, s, has_bias = case
    head = Cout % 8 != 0                      # head outputs: fp32 out, gradient arrives zero-padded to a multiple of 8
    torch.manual_seed(sum(case))
    x = bf(torch.randn(B, Cin, H, W))
    w = torch.randn(Cout, Cin, k, k) / (Cin * k * k) ** 0.5
    wq = bf(w)                                # the kernels see the bf16-rounded pack of the fp32 master weights
    bias = torch.randn(Cout) if has_bias else None
    p = k // 2
    ref = F.conv2d(x.double(), wq.double(), bias.double() if has_bias else None, s, p)
    Ho, Wo = ref.shape[2], ref.shape[3]
    cin = rup8(Cin)
    st = torch.cuda.current_stream().cuda_stream
    xg = nhwc_bf16(x, cpad=cin, ld=cin + 8, off=8)          # a channel-slice view of a wider buffer
    xv = xg.view(-1)[8:]
    wf, ldf, wb, ldb = pack(L, w.cuda(), cin)
    nblk = lib.yh_bf16_conv_fwd_blocks(B, H, W, cin, Cout, k, s, 1 if head else 0, cin + 8, Cout if head else Cout + 8)
    part = torch.zeros(nblk * 2 * Cout, device="cuda")
    bd = bias.cuda() if has_bias else None
    if head:
        ybuf = torch.full((B, Ho, Wo, Cout), 7.0, device="cuda")
        L.check(lib.yh_bf16_conv_fwd(xv.data_ptr(), cin + 8, wf.data_ptr(), ldf, bd.data_ptr() if has_bias else None,
                                     ybuf.data_ptr(), Cout, 1, None, B, H, W, cin, Cout, k, s, st), "fwd")
        y = ybuf.permute(0, 3, 1, 2)
        assert rel_err(y, ref) < F32_OUT_TOL
    else:
        ld = Cout + 8
        ybuf = torch.full((B, Ho, Wo, ld), 7.0, dtype=torch.bfloat16, device="cuda")
        yv = ybuf.view(-1)[8:]
        L.check(lib.yh_bf16_conv_fwd(xv.data_ptr(), cin + 8, wf.data_ptr(), ldf, bd.data_ptr() if has_bias else None,
                                     yv.data_ptr(), ld, 0, part.data_ptr(), B, H, W, cin, Cout, k, s, st), "fwd")
        y = ybuf[..., 8:].float().permute(0, 3, 1, 2)
        assert rel_err(y, ref) < BF16_OUT_TOL
        assert bool((ybuf[..., :8] == 7.0).all())                            # neighbouring channels untouched
        # BatchNorm partials = column sums of the STORED values
        ps = part.view(nblk, 2, Cout).sum(0).cpu().double()
        yd = y.double().cpu()
        assert rel_err(ps[0], yd.sum((0, 2, 3))) < 1e-3
        assert rel_err(ps[1], (yd * yd).sum((0, 2, 3))) < 1e-4
    # ---- backward-data ---------------------------------------------------------------------------------------------
    dy = bf(torch.randn(B, Cout, Ho, Wo))
    kp = rup8(Cout)
    dyg = nhwc_bf16(dy, cpad=kp)                                              # zero-padded to a multiple of 8 channels
    want_dx = F.conv_transpose2d(dy.double(), wq.double(), None, s, p,
                                 output_padding=(H + 2 * p - k - (Ho - 1) * s, W + 2 * p - k - (Wo - 1) * s))
    dxbuf = torch.full((B, H, W, cin + 8), 5.0, dtype=torch.bfloat16, device="cuda")
    dxv = dxbuf.view(-1)[8:]
    if Cin >= 8:          # the network never needs the image gradient through the padded stem
        L.check(lib.yh_bf16_conv_bwd_data(dyg.data_ptr(), kp, None, 0, wb.data_ptr(), ldb, dxv.data_ptr(), cin + 8, B, H, W, cin,
                                          kp, k, s, 0, st), "bwd_data")
        dx = dxbuf[..., 8:8 + Cin].float().permute(0, 3, 1, 2)
        assert rel_err(dx, want_dx) < BF16_OUT_TOL
        assert bool((dxbuf[..., :8] == 5.0).all())
        # accumulate form: dx += ...
        L.check(lib.yh_bf16_conv_bwd_data(dyg.data_ptr(), kp, None, 0, wb.data_ptr(), ldb, dxv.data_ptr(), cin + 8, B, H, W, cin,
                                          kp, k, s, 1, st), "bwd_data acc")
        dx2 = dxbuf[..., 8:8 + Cin].float().permute(0, 3, 1, 2)
        assert rel_err(dx2, 2 * want_dx) < 2 * BF16_OUT_TOL
    # ---- backward-weight -------------------------------------------------------------------------------------------
    want_dw = torch.nn.grad.conv2d_weight(x.double(), w.shape, dy.double(), s, p)
    nws = int(lib.yh_bf16_conv_bwd_weight_ws(B, H, W, cin, Cout, k, s))
    ws = torch.empty(nws, device="cuda")
    dw = torch.full((Cout, Cin, k, k), 9.0, device="cuda")
    L.check(lib.yh_bf16_conv_bwd_weight(xv.data_ptr(), cin + 8, dyg.data_ptr(), kp, dw.data_ptr(), ws.data_ptr(), nws, B, H, W, cin,
                                        Cin, Cout, k, s, st), "bwd_weight")
    assert rel_err(dw, want_dw) < F32_OUT_TOL
    dw2 = torch.empty_like(dw)
    L.check(lib.yh_bf16_conv_bwd_weight(xv.data_ptr(), cin + 8, dyg.data_ptr(), kp, dw2.data_ptr(), ws.data_ptr(), nws, B, H, W, cin,
                                        Cin, Cout, k, s, st), "bwd_weight")
    assert torch.equal(dw, dw2)                                               # deterministic


STREAM_CASES = [  # (B, H, W, K, N, k): the flat-stream kernel forced (the dispatch only picks it where it measured faster)
    (2, 20, 20, 16, 16, 3), (1, 40, 40, 32, 32, 3), (2, 12, 100, 16, 32, 3), (1, 16, 16, 64, 64, 3), (3, 9, 7, 64, 32, 3),
    (2, 20, 20, 32, 16, 1), (2, 13, 11, 64, 24, 1), (1, 10, 10, 128, 128, 1), (1, 37, 5, 128, 64, 1), (2, 8, 8, 16, 128, 1),
    (1, 30, 30, 64, 64, 1), (1, 200, 3, 32, 64, 3),
]


@pytest.mark.parametrize("B,H,W,K,N,k", STREAM_CASES)
def test_bf16_flat_stream_fwd_and_bwd_data(B, H, W, K, N, k):
    """yh_bf16_conv_stream_fwd / _bwd_data (conv_bf16_stream.hip) on every tile shape they instantiate: output, untouched
    neighbouring channels of a view, BatchNorm partial rows (one per persistent workgroup, padding positions excluded),
    backward-data plain / accumulating and with K split over two source tensors (the C3 sibling pair)."""
    L = _lib()
    lib = L.lib()
    torch.manual_seed(B * 1000 + H * 10 + K + N)
    st = torch.cuda.current_stream().cuda_stream
    p = k // 2
    x = bf(torch.randn(B, K, H, W))
    w = torch.randn(N, K, k, k) / (K * k * k) ** 0.5
    wq = bf(w)
    bias = torch.randn(N)
    ref = F.conv2d(x.double(), wq.double(), bias.double(), 1, p)
    xg = nhwc_bf16(x, cpad=K, ld=K + 8, off=8)
    xv = xg.view(-1)[8:]
    wf, ldf, wb, ldb = pack(L, w.cuda(), K)
    nblk = lib.yh_bf16_conv_stream_blocks(B, H, W, K, N, k)
    assert nblk > 0
    part = torch.zeros(nblk * 2 * N, device="cuda")
    ld = N + 8
    ybuf = torch.full((B, H, W, ld), 7.0, dtype=torch.bfloat16, device="cuda")
    L.check(lib.yh_bf16_conv_stream_fwd(xv.data_ptr(), K + 8, wf.data_ptr(), ldf, bias.cuda().data_ptr(), ybuf.view(-1)[8:].data_ptr(), ld,
                                        part.data_ptr(), B, H, W, K, N, k, st), "stream fwd")
    y = ybuf[..., 8:].float().permute(0, 3, 1, 2)
    assert rel_err(y, ref) < BF16_OUT_TOL
    assert bool((ybuf[..., :8] == 7.0).all())
    ps = part.view(nblk, 2, N).sum(0).cpu().double()
    yd = y.double().cpu()
    assert rel_err(ps[0], yd.sum((0, 2, 3))) < 1e-3
    assert rel_err(ps[1], (yd * yd).sum((0, 2, 3))) < 1e-4
    # backward-data of a conv whose OUTPUT channels are streamed (K of the GEMM) and whose input channels are produced (N)
    if lib.yh_bf16_conv_stream_blocks(B, H, W, N, K, k) == 0 or N % 8:
        return
    dy = bf(torch.randn(B, N, H, W))
    dyg = nhwc_bf16(dy, cpad=N)
    want = F.conv_transpose2d(dy.double(), wq.double(), None, 1, p)
    dxbuf = torch.full((B, H, W, K + 8), 5.0, dtype=torch.bfloat16, device="cuda")
    dxv = dxbuf.view(-1)[8:]
    for acc, mult in ((0, 1), (1, 2)):
        L.check(lib.yh_bf16_conv_stream_bwd_data(dyg.data_ptr(), N, None, 0, wb.data_ptr(), ldb, dxv.data_ptr(), K + 8, B, H, W, K, N, k, acc,
                                                 st), "stream bwd_data")
        dx = dxbuf[..., 8:].float().permute(0, 3, 1, 2)
        assert rel_err(dx, mult * want) < mult * BF16_OUT_TOL
        assert bool((dxbuf[..., :8] == 5.0).all())
    if k == 1 and N >= 32:      # the C3 sibling pair: dY of the two convs in two tensors, one stacked backward pack
        n1 = N // 2
        d1, d2 = nhwc_bf16(dy[:, :n1], cpad=n1), nhwc_bf16(dy[:, n1:], cpad=N - n1)
        dx2 = torch.empty(B, H, W, K, dtype=torch.bfloat16, device="cuda")
        L.check(lib.yh_bf16_conv_stream_bwd_data(d1.data_ptr(), n1, d2.data_ptr(), n1, wb.data_ptr(), ldb, dx2.data_ptr(), K, B, H, W, K, N, k,
                                                 0, st), "stream bwd_data pair")
        assert rel_err(dx2.float().permute(0, 3, 1, 2), want) < BF16_OUT_TOL


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,s", [(2, 13, 11, 64, 24, 1, 1), (1, 17, 9, 32, 40, 3, 1), (2, 10, 14, 16, 16, 3, 2)])
def test_bf16_conv_direct_store_variants(B, H, W, Cin, Cout, k, s):
    """Outputs whose row pitch is not a multiple of 8 elements cannot take the staged 16-byte stores: forward and
    backward-data (plain and accumulating) then run the element-wise epilogue variants of bf16_gemm_kernel, ragged
    tiles included.  Same oracle and tolerance as above; channels next to the view must stay untouched."""
    L = _lib()
    lib = L.lib()
    torch.manual_seed(B * 1000 + H * 100 + Cout)
    st = torch.cuda.current_stream().cuda_stream
    x = bf(torch.randn(B, Cin, H, W))
    w = torch.randn(Cout, Cin, k, k) / (Cin * k * k) ** 0.5
    wq = bf(w)
    p = k // 2
    ref = F.conv2d(x.double(), wq.double(), None, s, p)
    Ho, Wo = ref.shape[2], ref.shape[3]
    xg = nhwc_bf16(x)
    wf, ldf, wb, ldb = pack(L, w.cuda(), Cin)
    nblk = lib.yh_bf16_conv_fwd_blocks(B, H, W, Cin, Cout, k, s, 0, Cin, Cout + 4)
    part = torch.zeros(nblk * 2 * Cout, device="cuda")
    ld = Cout + 4                                   # 8-byte aligned rows only
    ybuf = torch.full((B, Ho, Wo, ld), 7.0, dtype=torch.bfloat16, device="cuda")
    yv = ybuf.view(-1)[4:]
    L.check(lib.yh_bf16_conv_fwd(xg.data_ptr(), Cin, wf.data_ptr(), ldf, None, yv.data_ptr(), ld, 0, part.data_ptr(),
                                 B, H, W, Cin, Cout, k, s, st), "fwd")
    y = ybuf[..., 4:].float().permute(0, 3, 1, 2)
    assert rel_err(y, ref) < BF16_OUT_TOL
    assert bool((ybuf[..., :4] == 7.0).all())
    ps = part.view(nblk, 2, Cout).sum(0).cpu().double()
    yd = y.double().cpu()
    assert rel_err(ps[0], yd.sum((0, 2, 3))) < 1e-3
    assert rel_err(ps[1], (yd * yd).sum((0, 2, 3))) < 1e-4
    # backward-data into a 4-element-offset view, then accumulating
    dy = bf(torch.randn(B, Cout, Ho, Wo))
    dyg = nhwc_bf16(dy, cpad=rup8(Cout))
    want = F.conv_transpose2d(dy.double(), wq.double(), None, s, p,
                              output_padding=(H + 2 * p - k - (Ho - 1) * s, W + 2 * p - k - (Wo - 1) * s))
    ldx = Cin + 4
    dxbuf = torch.full((B, H, W, ldx), 5.0, dtype=torch.bfloat16, device="cuda")
    dxv = dxbuf.view(-1)[4:]
    for acc, mult in ((0, 1), (1, 2)):
        L.check(lib.yh_bf16_conv_bwd_data(dyg.data_ptr(), rup8(Cout), None, 0, wb.data_ptr(), ldb, dxv.data_ptr(), ldx, B, H, W, Cin,
                                          rup8(Cout), k, s, acc, st), "bwd_data")
        dx = dxbuf[..., 4:].float().permute(0, 3, 1, 2)
        assert rel_err(dx, mult * want) < mult * BF16_OUT_TOL
        assert bool((dxbuf[..., :4] == 5.0).all())


def test_bf16_mfma_operand_maps_with_exact_integers():
    """A = I (through the conv: 1x1, weights = identity) and an ASYMMETRIC second operand, small integers that bf16 and
    fp32 hold exactly: any swapped row/column or k map gives a wrong integer, not a rounding difference."""
    L = _lib()
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    B, H, W, C = 1, 8, 16, 64
    x = torch.zeros(B, C, H, W)
    for c in range(C):
        for h in range(H):
            for w_ in range(W):
                x[0, c, h, w_] = float((3 * c + 5 * h + 7 * w_) % 13 - 6)
    eye = torch.eye(C).reshape(C, C, 1, 1).contiguous()
    wf, ldf, wb, ldb = pack(L, eye.cuda(), C)
    xg = nhwc_bf16(x)
    y = torch.empty(B, H, W, C, dtype=torch.bfloat16, device="cuda")
    L.check(lib.yh_bf16_conv_fwd(xg.data_ptr(), C, wf.data_ptr(), ldf, None, y.data_ptr(), C, 0, None, B, H, W, C, C, 1, 1, st))
    assert torch.equal(y.float().cpu(), x.permute(0, 2, 3, 1))
    # asymmetric integer weights, exact result
    w = torch.zeros(C, C, 1, 1)
    for o in range(C):
        for i in range(C):
            w[o, i, 0, 0] = float((o + 2 * i) % 5 - 2)
    wf, ldf, wb, ldb = pack(L, w.cuda(), C)
    L.check(lib.yh_bf16_conv_fwd(xg.data_ptr(), C, wf.data_ptr(), ldf, None, y.data_ptr(), C, 0, None, B, H, W, C, C, 1, 1, st))
    want = F.conv2d(x, w)          # |values| <= 64 * 6 * 2: exact in bf16? no -- compare in fp32 via the fp32-output form
    yf = torch.empty(B, H, W, C, device="cuda")
    L.check(lib.yh_bf16_conv_fwd(xg.data_ptr(), C, wf.data_ptr(), ldf, None, yf.data_ptr(), C, 1, None, B, H, W, C, C, 1, 1, st))
    assert torch.equal(yf.cpu(), want.permute(0, 2, 3, 1))
    # weight gradient with exact integers: dw[o][i] = sum_p x[p][i] dy[p][o] (the transposing LDS reads)
    dy = torch.zeros(B, C, H, W)
    for c in range(C):
        for h in range(H):
            for w_ in range(W):
                dy[0, c, h, w_] = float((c + 3 * h + 2 * w_) % 7 - 3)
    dyg = nhwc_bf16(dy)
    nws = int(lib.yh_bf16_conv_bwd_weight_ws(B, H, W, C, C, 1, 1))
    ws = torch.empty(nws, device="cuda")
    dw = torch.empty(C, C, 1, 1, device="cuda")
    L.check(lib.yh_bf16_conv_bwd_weight(xg.data_ptr(), C, dyg.data_ptr(), C, dw.data_ptr(), ws.data_ptr(), nws, B, H, W, C, C, C, 1, 1, st))
    want_dw = torch.nn.grad.conv2d_weight(x, w.shape, dy, 1, 0)
    assert torch.equal(dw.cpu(), want_dw)


def test_bf16_pair_backward_data():
    """The C3 sibling pair: dx = dy1 W1^T + dy2 W2^T as one GEMM over K = c1 + c2 from two sources."""
    L = _lib()
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    B, H, W, Cin, c1 = 2, 12, 12, 64, 32
    torch.manual_seed(3)
    w1 = torch.randn(c1, Cin, 1, 1) / 8
    w2 = torch.randn(c1, Cin, 1, 1) / 8
    dy1, dy2 = bf(torch.randn(B, c1, H, W)), bf(torch.randn(B, c1, H, W))
    stacked = torch.zeros(2 * c1 * Cin, dtype=torch.bfloat16, device="cuda")
    pack(L, w1.cuda(), Cin, koff=0, kpad=2 * c1, wb_buf=stacked)
    pack(L, w2.cuda(), Cin, koff=c1, kpad=2 * c1, wb_buf=stacked)
    cat = torch.zeros(B, H, W, 2 * c1 + 16, dtype=torch.bfloat16, device="cuda")     # the two gradients live in one buffer
    cat[..., :c1] = dy1.permute(0, 2, 3, 1).to(torch.bfloat16).cuda()
    cat[..., c1 + 16:] = dy2.permute(0, 2, 3, 1).to(torch.bfloat16).cuda()
    ld = 2 * c1 + 16
    dx = torch.empty(B, H, W, Cin, dtype=torch.bfloat16, device="cuda")
    L.check(lib.yh_bf16_conv_bwd_data(cat.data_ptr(), ld, cat.view(-1)[c1 + 16:].data_ptr(), c1, stacked.data_ptr(), Cin, dx.data_ptr(),
                                      Cin, B, H, W, Cin, 2 * c1, 1, 1, 0, st), "pair")
    want = F.conv_transpose2d(dy1.double(), bf(w1).double()) + F.conv_transpose2d(dy2.double(), bf(w2).double())
    assert rel_err(dx.float().permute(0, 3, 1, 2), want) < BF16_OUT_TOL


@pytest.mark.parametrize("C,H,W,up,res", [(16, 12, 12, 0, 0), (64, 10, 10, 0, 1), (128, 6, 6, 1, 0), (32, 7, 9, 0, 1)])
def test_bf16_bn_silu_forward_backward(C, H, W, up, res):
    L = _lib()
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    B = 3
    M = B * H * W
    torch.manual_seed(C + H)
    y = bf(torch.randn(M, C) * 1.5 + 0.3)
    gamma, beta = torch.rand(C) + 0.5, torch.randn(C) * 0.1
    mean, var = y.double().mean(0), y.double().var(0, unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    scale = gamma.double() * invstd
    coef = torch.cat([scale, beta.double() - mean * scale, mean, invstd]).float().cuda()
    r = bf(torch.randn(M, C)) if res else None
    z = y.double() * coef[:C].cpu().double() + coef[C:2 * C].cpu().double()
    a = z * torch.sigmoid(z) + (r.double() if res else 0)
    f = 2 if up else 1
    yg = y.to(torch.bfloat16).cuda()
    rg = r.to(torch.bfloat16).cuda() if res else None
    out = torch.empty(B, H * f, W * f, C, dtype=torch.bfloat16, device="cuda")
    L.check(lib.yh_bf16_bn_silu_fwd(yg.data_ptr(), C, coef.data_ptr(), rg.data_ptr() if res else None, C, out.data_ptr(), C, M, C, H, W,
                                    up, st), "fwd")
    want = a.view(B, H, W, C)
    if up:
        want = want.repeat_interleave(2, 1).repeat_interleave(2, 2)
    assert rel_err(out.float(), want) < BF16_OUT_TOL
    # backward: da (bf16) -> dy (bf16), dgamma / dbeta (fp32), residual route
    da = bf(torch.randn(B, H * f, W * f, C))
    dag = da.to(torch.bfloat16).cuda()
    g = da.double()
    if up:
        g = g.view(B, H, 2, W, 2, C).sum((2, 4))
    g = g.reshape(M, C)
    sg = torch.sigmoid(z)
    dz = g * (sg * (1 + z * (1 - sg)))
    xh = (y.double() - mean) * invstd
    want_dbeta, want_dgamma = dz.sum(0), (dz * xh).sum(0)
    want_dy = scale * (dz - want_dbeta / M - xh * want_dgamma / M)
    nb = lib.yh_bn_bwd_blocks(M, C)
    part = torch.empty(nb * 2 * C, device="cuda")
    L.check(lib.yh_bf16_bn_silu_bwd_reduce(dag.data_ptr(), C, yg.data_ptr(), C, coef.data_ptr(), part.data_ptr(), M, C, H, W, up, st))
    dgam, dbet = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    dyo = torch.empty(M, C, dtype=torch.bfloat16, device="cuda")
    dres = torch.full((M, C), 1.0, dtype=torch.bfloat16, device="cuda") if res else None
    L.check(lib.yh_bf16_bn_silu_bwd_apply(dag.data_ptr(), C, yg.data_ptr(), C, coef.data_ptr(), part.data_ptr(), nb, None,
                                          dgam.data_ptr(), dbet.data_ptr(), dyo.data_ptr(), C, dres.data_ptr() if res else None, C, 1,
                                          M, C, H, W, up, st))
    assert rel_err(dgam, want_dgamma) < 2e-3 and rel_err(dbet, want_dbeta) < 2e-3      # fast-sigmoid (1 ulp exp/rcp) sums
    assert rel_err(dyo.float(), want_dy) < BF16_OUT_TOL
    if res:
        assert rel_err(dres.float(), g + 1.0) < BF16_OUT_TOL


def test_bf16_pool_layout_colsum():
    L = _lib()
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    B, H, W, C = 2, 9, 7, 32
    torch.manual_seed(1)
    x = bf(torch.randn(B, C, H, W))
    x[0, :, 2:5, 2:5] = 1.25                                   # exact ties: first max wins
    xg = nhwc_bf16(x)
    y = torch.empty_like(xg)
    arg = torch.empty(B, H, W, C, dtype=torch.uint8, device="cuda")
    L.check(lib.yh_bf16_maxpool5_fwd(xg.data_ptr(), C, y.data_ptr(), C, arg.data_ptr(), B, H, W, C, st))
    xr = x.clone().requires_grad_(True)
    ref = F.max_pool2d(xr, 5, 1, 2)
    assert torch.equal(y.float().cpu().permute(0, 3, 1, 2), ref.detach())
    dy = bf(torch.randn(B, C, H, W))
    ref.backward(dy)
    dx = torch.zeros(B, H, W, C, dtype=torch.bfloat16, device="cuda")
    L.check(lib.yh_bf16_maxpool5_bwd(nhwc_bf16(dy).data_ptr(), C, arg.data_ptr(), dx.data_ptr(), C, B, H, W, C, st))
    assert rel_err(dx.float().permute(0, 3, 1, 2), xr.grad) < BF16_OUT_TOL
    # layout: NCHW fp32 -> NHWC bf16 (padded) and back
    img = torch.rand(B, 3, H, W)
    dst = torch.full((B, H, W, 8), 2.0, dtype=torch.bfloat16, device="cuda")
    L.check(lib.yh_bf16_nchw_to_nhwc(img.cuda().data_ptr(), dst.data_ptr(), B, 3, H, W, 8, 8, st))
    assert torch.equal(dst[..., :3].float().cpu(), bf(img).permute(0, 2, 3, 1)) and bool((dst[..., 3:] == 0).all())
    back = torch.empty(B, 3, H, W, device="cuda")
    L.check(lib.yh_bf16_nhwc_to_nchw(dst.data_ptr(), back.data_ptr(), B, 3, H, W, 8, 0, st))
    assert torch.equal(back.cpu(), bf(img))
    u8 = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8)
    L.check(lib.yh_bf16_u8hwc_to_nhwc(u8.cuda().data_ptr(), dst.data_ptr(), B, H, W, 3, 8, 8, st))
    assert torch.equal(dst[..., :3].float().cpu(), bf(u8.float() / 255.0))
    # column sums (bias gradient)
    M = B * H * W
    out = torch.empty(C, device="cuda")
    ws = torch.empty(int(lib.yh_colsum_ws(M, C)), device="cuda")
    L.check(lib.yh_bf16_colsum(xg.data_ptr(), C, M, C, out.data_ptr(), ws.data_ptr(), st))
    assert rel_err(out, x.double().sum((0, 2, 3))) < 1e-5


def test_loss_writes_padded_bf16_head_gradient():
    """yh_yolo_loss_ex with a bf16, pixel-padded dpred equals the fp32 gradient rounded once, padding zeroed."""
    import yolo_from_scratch_amd as y
    L = _lib()
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    nc, S, B = 3, 64, 2
    grids = [S // 8, S // 16, S // 32]
    ch = 5 + nc
    torch.manual_seed(5)
    preds = [torch.randn(B, g, g, 3, ch, device="cuda") for g in grids]
    tg = [t.cuda() for t in y.synthetic_targets(B, nc, S, 4, 11)]
    a18 = [float(v) for s in y.DEFAULT_ANCHORS for p in s for v in p]
    ws = torch.empty(int(lib.yh_loss_ws(L.int3(grids), B)) + 8, device="cuda")
    out1, out2 = torch.empty(13, device="cuda"), torch.empty(13, device="cuda")
    d32 = [torch.full_like(p, float("nan")) for p in preds]      # the loss writes EVERY element of the gradient (whole rows)
    L.check(lib.yh_yolo_loss(L.ptr3(preds), L.ptr3(tg), L.ptr3(d32), L.floats(a18), L.int3(grids), B, nc, 640.0, None, None,
                             out1.data_ptr(), ws.data_ptr(), st))
    ldd = rup8(3 * ch)
    d16 = [torch.full((B, g, g, ldd), 1.0, dtype=torch.bfloat16, device="cuda") for g in grids]
    L.check(lib.yh_yolo_loss_ex(L.ptr3(preds), L.ptr3(tg), L.ptr3(d16), 1, L.int3([ldd] * 3), L.floats(a18), L.int3(grids), B, nc,
                                640.0, None, None, out2.data_ptr(), ws.data_ptr(), st))
    assert torch.equal(out1, out2)
    for a, b in zip(d32, d16):
        assert not bool(torch.isnan(a).any())
        assert torch.equal(b[..., :3 * ch], a.reshape(a.shape[0], a.shape[1], a.shape[2], 3 * ch).to(torch.bfloat16))
        assert bool((b[..., 3 * ch:] == 0).all())


@pytest.mark.parametrize("B,H,W,Cin,Cout,s", [(2, 24, 64, 16, 16, 1), (3, 23, 45, 16, 16, 1), (2, 32, 96, 16, 32, 2), (2, 21, 67, 16, 32, 2),
                                             (2, 64, 96, 3, 16, 2), (3, 37, 51, 3, 16, 2), (16, 125, 160, 16, 16, 1)])
def test_bf16_narrow_layer_kernels(B, H, W, Cin, Cout, s):
    """The narrow-layer direct kernels with bf16 storage (yh_bf16_conv_narrow / _dgrad_s2 / _bwd_weight: bf16 activations and
    packs widened into LDS, fp32 MFMA loop) against fp64 torch on the bf16-rounded operands and against the generic bf16
    kernels: forward with bias + BatchNorm partials of the STORED values into a channel slice, stride-1 (flipped taps) and
    stride-2 backward-data with accumulate, weight gradient (fp32, bitwise reproducible).  Last case: more patches than
    persistent workgroups."""
    L = _lib()
    lib = L.lib()
    torch.manual_seed(H * W + s + Cin)
    x = bf(torch.randn(B, Cin, H, W))
    w = torch.randn(Cout, Cin, 3, 3) / (Cin * 9) ** 0.5
    wq = bf(w)
    bias = torch.randn(Cout)
    ref = F.conv2d(x.double(), wq.double(), bias.double(), s, 1)
    Ho, Wo = ref.shape[2], ref.shape[3]
    cin = rup8(Cin)
    cin_k = 4 if Cin <= 4 else Cin                 # channels the narrow kernel reads
    st = torch.cuda.current_stream().cuda_stream
    xg = nhwc_bf16(x, cpad=cin, ld=cin + 8, off=8)
    xv = xg.view(-1)[8:]
    wf, ldf, wb, ldb = pack(L, w.cuda(), cin)
    assert lib.yh_conv_narrow_ok(cin_k, Cout, 3, s) == 1
    nblk = lib.yh_conv_narrow_blocks(B, H, W, cin_k, s)
    part = torch.zeros(nblk * 2 * Cout, device="cuda")
    ld = Cout + 8
    ybuf = torch.full((B, Ho, Wo, ld), 7.0, dtype=torch.bfloat16, device="cuda")
    yv = ybuf.view(-1)[8:]
    L.check(lib.yh_bf16_conv_narrow(xv.data_ptr(), cin + 8, wf.data_ptr(), ldf, cin, bias.cuda().data_ptr(), yv.data_ptr(), ld,
                                    part.data_ptr(), B, H, W, cin_k, Cout, s, 0, 0, st), "bf16 narrow fwd")
    y = ybuf[..., 8:].float().permute(0, 3, 1, 2)
    assert rel_err(y, ref) < BF16_OUT_TOL
    assert bool((ybuf[..., :8] == 7.0).all())
    ps = part.view(nblk, 2, Cout).sum(0).cpu().double()
    yd = y.double().cpu()
    assert rel_err(ps[0], yd.sum((0, 2, 3))) < 1e-3 and rel_err(ps[1], (yd * yd).sum((0, 2, 3))) < 1e-4
    # the generic bf16 kernel on the same operands: same values up to the rounding of the last bf16 bit
    y2buf = torch.empty(B, Ho, Wo, Cout, dtype=torch.bfloat16, device="cuda")
    L.check(lib.yh_bf16_conv_fwd(xv.data_ptr(), cin + 8, wf.data_ptr(), ldf, bias.cuda().data_ptr(), y2buf.data_ptr(), Cout, 0, None,
                                 B, H, W, cin, Cout, 3, s, st), "generic fwd")
    assert rel_err(ybuf[..., 8:].float(), y2buf.float()) < BF16_OUT_TOL
    # ---- backward-data ---------------------------------------------------------------------------------------------
    dy = bf(torch.randn(B, Cout, Ho, Wo))
    kp = rup8(Cout)
    dyg = nhwc_bf16(dy, cpad=kp)
    if Cin >= 8:
        want_dx = F.conv_transpose2d(dy.double(), wq.double(), None, s, 1, output_padding=(H + 2 - 3 - (Ho - 1) * s, W + 2 - 3 - (Wo - 1) * s))
        dxbuf = torch.full((B, H, W, cin + 8), 5.0, dtype=torch.bfloat16, device="cuda")
        dxv = dxbuf.view(-1)[8:]
        for acc in (0, 1):
            if s == 1:
                L.check(lib.yh_bf16_conv_narrow(dyg.data_ptr(), kp, wb.data_ptr(), ldb, kp, None, dxv.data_ptr(), cin + 8, None, B, H, W,
                                                Cout, Cin, 1, 1, acc, st), "bf16 narrow dgrad")
            else:
                assert lib.yh_conv_narrow_dgrad_s2_ok(Cin, Cout) == 1
                L.check(lib.yh_bf16_conv_narrow_dgrad_s2(dyg.data_ptr(), kp, wb.data_ptr(), ldb, kp, dxv.data_ptr(), cin + 8, B, H, W, Cin,
                                                         Cout, acc, st), "bf16 narrow dgrad s2")
            dx = dxbuf[..., 8:8 + Cin].float().permute(0, 3, 1, 2)
            assert rel_err(dx, (1 + acc) * want_dx) < (1 + acc) * BF16_OUT_TOL
            assert bool((dxbuf[..., :8] == 5.0).all())
    # ---- backward-weight -------------------------------------------------------------------------------------------
    want_dw = torch.nn.grad.conv2d_weight(x.double(), w.shape, dy.double(), s, 1)
    assert lib.yh_conv_narrow_bwd_weight_ok(cin_k, min(Cin, cin_k), Cout, 3, s) == 1
    nws = int(lib.yh_conv_narrow_bwd_weight_ws(B, H, W, cin_k, Cout, s))
    ws = torch.empty(nws, device="cuda")
    dw = torch.full((Cout, Cin, 3, 3), 9.0, device="cuda")
    db = torch.full((Cout,), 9.0, device="cuda")
    args = (xv.data_ptr(), cin + 8, dyg.data_ptr(), kp)
    L.check(lib.yh_bf16_conv_narrow_bwd_weight(*args, dw.data_ptr(), db.data_ptr(), ws.data_ptr(), nws, B, H, W, cin_k, min(Cin, cin_k), Cout, s, st), "wgrad")
    assert rel_err(dw, want_dw) < F32_OUT_TOL
    assert rel_err(db, dy.double().sum((0, 2, 3))) < F32_OUT_TOL
    dw2 = torch.empty_like(dw)
    L.check(lib.yh_bf16_conv_narrow_bwd_weight(*args, dw2.data_ptr(), None, ws.data_ptr(), nws, B, H, W, cin_k, min(Cin, cin_k), Cout, s, st), "wgrad")
    assert torch.equal(dw, dw2)


def test_bf16_conv_stream_eligible_views_must_be_aligned():
    """the bf16 convolution entry points pick the flat-stream or the gather kernel from shapes and strides alone (the planner
    sizes the BatchNorm partial rows from the same predicate, which cannot see pointers).  For a stream-eligible problem (both strides
    multiples of 8) a view at a channel offset of 4 (8 bytes) is therefore an argument error with a message that says so -- not a
    silent change of route with a different partial-row count; the same view with a stride that is not a multiple of 8 runs on the
    gather kernel (test_bf16_conv_direct_store_variants)."""
    L = _lib()
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    B, H, W, cin, cout, k, s = 2, 8, 8, 32, 32, 3, 1
    ld = cin + 8
    x = torch.zeros(B, H, W, ld, dtype=torch.bfloat16, device="cuda")
    wf = torch.zeros(k * k * cin * rup8(cout), dtype=torch.bfloat16, device="cuda")
    y = torch.zeros(B, H, W, cout, dtype=torch.bfloat16, device="cuda")
    args = (wf.data_ptr(), rup8(cout), None, y.data_ptr(), cout, 0, None, B, H, W, cin, cout, k, s, st)
    assert lib.yh_bf16_conv_fwd(x.data_ptr(), ld, *args) == 0, lib.yh_last_error()
    rc = lib.yh_bf16_conv_fwd(x.data_ptr() + 8, ld, *args)
    assert rc != 0 and b"16-byte aligned" in lib.yh_last_error()
    dx = torch.zeros(B, H, W, ld, dtype=torch.bfloat16, device="cuda")
    rc = lib.yh_bf16_conv_bwd_data(y.data_ptr(), cout, None, 0, wf.data_ptr(), rup8(cin), dx.data_ptr() + 8, ld, B, H, W, cin, cout, k, s, 0, st)
    assert rc != 0 and b"16-byte aligned" in lib.yh_last_error()
    torch.cuda.synchronize()
