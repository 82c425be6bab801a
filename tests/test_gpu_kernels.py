"""GPU parity of the individual C-ABI kernels against a plain fp64/fp32 torch CPU evaluation of the
same op (conv fwd / bwd-data / bwd-weight on every layer shape family of the network, BN+SiLU, pool,
Adam).  Tolerances: fp32 MFMA is an fmaf chain, so 1e-4 relative to the tensor's max magnitude."""
import numpy as np
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


def nhwc(t):   # NCHW cpu -> NHWC cuda contiguous
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def rup4(c):
    return (c + 3) // 4 * 4


CONV_CASES = [  # (B, H, W, Cin, Cout, k, s, bias)  -- shape families of the 's' model at reduced size
    (2, 32, 32, 3, 16, 3, 2, True),      # stem.0 (Cin padded to 4)
    (2, 24, 24, 16, 32, 3, 2, True),     # stem.3
    (2, 20, 20, 32, 16, 1, 1, False),    # C3 1x1 small
    (2, 20, 20, 16, 16, 3, 1, False),    # bottleneck 16
    (1, 16, 16, 64, 64, 3, 1, False),    # head 3x3
    (1, 8, 8, 128, 128, 3, 1, False),
    (1, 6, 6, 256, 256, 3, 1, False),
    (1, 10, 10, 192, 64, 1, 1, False),   # panet conv (non power-of-two Cin)
    (1, 6, 6, 384, 128, 1, 1, False),
    (1, 6, 6, 512, 256, 1, 1, True),     # sppf.conv2
    (2, 12, 12, 64, 64, 3, 2, False),    # downsample
    (2, 9, 11, 32, 64, 3, 2, True),      # odd sizes, stride 2
    (2, 10, 10, 64, 18, 1, 1, True),     # head out nc=1
    (1, 10, 10, 128, 255, 1, 1, True),   # head out nc=80
    (3, 7, 5, 8, 12, 3, 1, False),       # tiny odd
    (1, 40, 40, 32, 32, 3, 1, False),    # M not a multiple of the tile
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(case):
    L = _lib()
    lib = L.lib()
    B, H, W, Cin, Cout, k, s, has_bias = case
    torch.manual_seed(hash(case) % 1000)
    x = torch.randn(B, Cin, H, W)
    w = torch.randn(Cout, Cin, k, k) / (Cin * k * k) ** 0.5
    bias = torch.randn(Cout) if has_bias else None
    p = k // 2
    ref = F.conv2d(x.double(), w.double(), bias.double() if has_bias else None, s, p)
    Ho, Wo = ref.shape[2], ref.shape[3]
    cin = rup4(Cin)
    st = torch.cuda.current_stream().cuda_stream
    xg = torch.zeros(B, H, W, cin, device="cuda")
    xg[..., :Cin] = nhwc(x)
    wd = w.cuda()
    ldwf, ldwb = rup4(Cout), cin
    wf = torch.empty(k * k * cin * ldwf, device="cuda")
    wb = torch.empty(k * k * Cout * ldwb, device="cuda")
    L.check(lib.yh_pack_weights(wd.data_ptr(), wf.data_ptr(), wb.data_ptr(), Cout, Cin, k, cin, ldwf, ldwb, st))
    # forward into a channel slice of a wider buffer (ld > C) with BN partial sums
    ld = Cout + 8
    ybuf = torch.full((B, Ho, Wo, ld), 7.0, device="cuda")
    nblk = lib.yh_conv_fwd_blocks(B, H, W, Cout, k, s)
    part = torch.zeros(nblk * 2 * Cout, device="cuda")
    bd = bias.cuda() if has_bias else None
    yview = ybuf.view(-1)[4:]
    L.check(lib.yh_conv_fwd(xg.data_ptr(), cin, wf.data_ptr(), ldwf, bd.data_ptr() if has_bias else None,
                            yview.data_ptr(), ld, part.data_ptr(), B, H, W, cin, Cout, k, s, st))
    y = ybuf[..., 4:4 + Cout].permute(0, 3, 1, 2)
    assert rel_err(y, ref) < 1e-4
    assert float(ybuf[..., :4].min()) == 7.0 and float(ybuf[..., 4 + Cout:].min()) == 7.0   # neighbours untouched
    ps = part.view(nblk, 2, Cout).double().sum(0).cpu()
    assert rel_err(ps[0], ref.sum((0, 2, 3))) < 1e-3 * max(1.0, float(ref.abs().sum((0, 2, 3)).max() / (ref.sum((0, 2, 3)).abs().max() + 1e-9)))
    assert rel_err(ps[1], (ref ** 2).sum((0, 2, 3))) < 1e-4
    # backward
    dy = torch.randn(B, Cout, Ho, Wo)
    xr = x.double().requires_grad_(True)
    wr = w.double().requires_grad_(True)
    F.conv2d(xr, wr, None, s, p).backward(dy.double())
    dyg = nhwc(dy)
    dx = torch.full((B, H, W, cin), 3.0, device="cuda")
    L.check(lib.yh_conv_bwd_data(dyg.data_ptr(), Cout, wb.data_ptr(), ldwb, dx.data_ptr(), cin, B, H, W, cin, Cout, k, s, 0, st))
    assert rel_err(dx[..., :Cin].permute(0, 3, 1, 2), xr.grad) < 1e-4
    L.check(lib.yh_conv_bwd_data(dyg.data_ptr(), Cout, wb.data_ptr(), ldwb, dx.data_ptr(), cin, B, H, W, cin, Cout, k, s, 1, st))
    assert rel_err(dx[..., :Cin].permute(0, 3, 1, 2), 2 * xr.grad) < 1e-4        # accumulate flag
    nws = lib.yh_conv_bwd_weight_ws(B, H, W, cin, Cout, k, s)
    assert nws > 0
    ws = torch.empty(nws, device="cuda")
    dw = torch.zeros(Cout, Cin, k, k, device="cuda")
    L.check(lib.yh_conv_bwd_weight(xg.data_ptr(), cin, dyg.data_ptr(), Cout, dw.data_ptr(), ws.data_ptr(), nws, B, H, W, cin,
                                   Cin, Cout, k, s, st))
    assert rel_err(dw, wr.grad) < 1e-4
    if has_bias:
        wsb = torch.empty(lib.yh_colsum_ws(B * Ho * Wo, Cout), device="cuda")
        db = torch.zeros(Cout, device="cuda")
        L.check(lib.yh_colsum(dyg.data_ptr(), Cout, B * Ho * Wo, Cout, db.data_ptr(), wsb.data_ptr(), st))
        assert rel_err(db, dy.double().sum((0, 2, 3))) < 1e-5


def test_conv_rejects_bad_arguments():
    L = _lib()
    lib = L.lib()
    t = torch.zeros(64, device="cuda")
    assert lib.yh_conv_fwd(t.data_ptr(), 4, t.data_ptr(), 4, None, t.data_ptr(), 4, None, 1, 4, 4, 4, 4, 5, 1, 0) != 0
    assert b"unsupported" in lib.yh_last_error()
    assert lib.yh_conv_fwd(None, 4, t.data_ptr(), 4, None, t.data_ptr(), 4, None, 1, 4, 4, 4, 4, 3, 1, 0) != 0
    with pytest.raises(RuntimeError):
        L.check(lib.yh_adam_step(None, None, None, None, 0, 1e-3, 0.9, 0.999, 1e-8, 1, 0.0, None, 1.0, 0), "adam")


@pytest.mark.parametrize("C,H,W,up,res", [(16, 9, 7, False, False), (64, 6, 6, True, False), (32, 8, 8, False, True),
                                          (256, 4, 4, False, False), (12, 5, 5, False, True)])
def test_bn_silu_fwd_bwd(C, H, W, up, res):
    L = _lib()
    lib = L.lib()
    B = 3
    st = torch.cuda.current_stream().cuda_stream
    torch.manual_seed(C + H)
    y = (torch.randn(B, C, H, W) * 2 + 0.5).double().requires_grad_(True)
    gamma = (torch.rand(C) + 0.5).double().requires_grad_(True)
    beta = (torch.randn(C) * 0.3).double().requires_grad_(True)
    r = torch.randn(B, C, H, W).double().requires_grad_(True) if res else None
    rm, rv = torch.zeros(C).double(), torch.ones(C).double()
    z = F.batch_norm(y, rm, rv, gamma, beta, True, 0.1, 1e-5)
    a = F.silu(z)
    if res:
        a = a + r
    if up:
        a = F.interpolate(a, scale_factor=2, mode="nearest")
    f = 2 if up else 1
    g = torch.randn(B, C, H * f, W * f).double()
    a.backward(g)
    M = B * H * W
    yg = nhwc(y.detach().float())
    # statistics through the same partial-sum interface the conv epilogue uses (1 block)
    part = torch.stack([yg.view(M, C).sum(0), (yg.view(M, C) ** 2).sum(0)]).contiguous()
    coef = torch.empty(4 * C, device="cuda")
    rmg, rvg = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    gm, bt = gamma.detach().float().cuda(), beta.detach().float().cuda()
    L.check(lib.yh_bn_finalize(part.data_ptr(), 1, M, gm.data_ptr(), bt.data_ptr(), rmg.data_ptr(), rvg.data_ptr(), 0.1, 1e-5,
                               coef.data_ptr(), C, None, st))
    assert rel_err(rmg, rm) < 1e-5 and rel_err(rvg, rv) < 1e-5
    rg = nhwc(r.detach().float()) if res else None
    out = torch.empty(B, H * f, W * f, C, device="cuda")
    L.check(lib.yh_bn_silu_fwd(yg.data_ptr(), C, coef.data_ptr(), rg.data_ptr() if res else None, C, out.data_ptr(), C, M, C, H,
                               W, int(up), st))
    assert rel_err(out.permute(0, 3, 1, 2), a.detach()) < 1e-5
    dag = nhwc(g.float())
    nb = lib.yh_bn_bwd_blocks(M, C)
    pb = torch.empty(nb * 2 * C, device="cuda")
    L.check(lib.yh_bn_silu_bwd_reduce(dag.data_ptr(), C, yg.data_ptr(), C, coef.data_ptr(), pb.data_ptr(), M, C, H, W, int(up), st))
    dgm, dbt = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    dres = torch.ones(B, H, W, C, device="cuda") if res else None
    dy = torch.empty_like(yg)
    L.check(lib.yh_bn_silu_bwd_apply(dag.data_ptr(), C, yg.data_ptr(), C, coef.data_ptr(), pb.data_ptr(), nb, gm.data_ptr(),
                                     dgm.data_ptr(), dbt.data_ptr(), dy.data_ptr(), C, dres.data_ptr() if res else None, C, 1, M,
                                     C, H, W, int(up), st))
    assert rel_err(dy.permute(0, 3, 1, 2), y.grad) < 2e-4
    assert rel_err(dgm, gamma.grad) < 2e-4 and rel_err(dbt, beta.grad) < 2e-4
    if res:
        assert rel_err(dres.permute(0, 3, 1, 2) - 1.0, r.grad) < 1e-5      # accumulated onto the ones


def test_maxpool5_chain_matches_torch_including_ties():
    L = _lib()
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    B, C, H, W = 2, 8, 11, 9
    torch.manual_seed(5)
    x = torch.randint(0, 4, (B, C, H, W)).double().requires_grad_(True)      # many exact ties
    y1 = F.max_pool2d(x, 5, 1, 2); y2 = F.max_pool2d(y1, 5, 1, 2); y3 = F.max_pool2d(y2, 5, 1, 2)
    g = torch.randn(B, 4 * C, H, W).double()
    torch.cat([x, y1, y2, y3], 1).backward(g)
    cat = torch.zeros(B, H, W, 4 * C, device="cuda")
    cat[..., :C] = nhwc(x.detach().float())
    args = [torch.empty(B, H, W, C, dtype=torch.uint8, device="cuda") for _ in range(3)]
    base = cat.data_ptr()
    for i in range(3):
        L.check(lib.yh_maxpool5_fwd(base + 4 * C * i, 4 * C, base + 4 * C * (i + 1), 4 * C, args[i].data_ptr(), B, H, W, C, st))
    ref = torch.cat([x, y1, y2, y3], 1).detach()
    assert rel_err(cat.permute(0, 3, 1, 2), ref) == 0.0
    dcat = nhwc(g.float())
    gb = dcat.data_ptr()
    for i in (2, 1, 0):
        L.check(lib.yh_maxpool5_bwd(gb + 4 * C * (i + 1), 4 * C, args[i].data_ptr(), gb + 4 * C * i, 4 * C, B, H, W, C, st))
    assert rel_err(dcat[..., :C].permute(0, 3, 1, 2), x.grad) < 1e-5


@pytest.mark.parametrize("B,C,H,W", [(1, 256, 20, 20), (3, 8, 11, 9), (2, 12, 64, 64), (1, 4, 3, 2)])
def test_sppf_pool3_one_launch_matches_torch_cascade(B, C, H, W):
    """inference form of SPPF's pooling (train.py:246-248): three cascaded pools in one launch, NaN wins like torch"""
    L = _lib()
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    torch.manual_seed(B * 1000 + C)
    x = torch.randn(B, C, H, W)
    x[0, 0, H // 2, W // 2] = float("nan")
    x[-1, C - 1, 0, 0] = float("inf")
    y1 = F.max_pool2d(x, 5, 1, 2); y2 = F.max_pool2d(y1, 5, 1, 2); y3 = F.max_pool2d(y2, 5, 1, 2)
    assert lib.yh_sppf_pool3_ok(H, W) == 1 and lib.yh_sppf_pool3_ok(65, 64) == 0
    cat = torch.zeros(B, H, W, 4 * C, device="cuda")
    cat[..., :C] = nhwc(x)
    base = cat.data_ptr()
    L.check(lib.yh_sppf_pool3_fwd(base, 4 * C, base + 4 * C, base + 8 * C, base + 12 * C, 4 * C, B, H, W, C, st))
    ref = torch.cat([x, y1, y2, y3], 1)
    got = cat.permute(0, 3, 1, 2).cpu()
    assert torch.equal(torch.isnan(got), torch.isnan(ref))
    assert torch.equal(torch.nan_to_num(got, nan=0.0), torch.nan_to_num(ref, nan=0.0))
    assert lib.yh_sppf_pool3_fwd(base, 4 * C, base + 4 * C, base + 8 * C, base + 12 * C, 4 * C, B, 65, 64, C, st) != 0   # too large for LDS: refused


@pytest.mark.parametrize("n,world", [(1003, 1), (4096, 4)])
def test_clip_adam_matches_oracle(n, world):
    from oracle import yolo_oracle as orc
    L = _lib()
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    torch.manual_seed(n)
    npad = rup4(n)
    p0, g0 = torch.randn(npad), torch.randn(npad) * 3
    p, g = p0.clone().cuda(), g0.clone().cuda()
    m, v = torch.zeros(npad, device="cuda"), torch.zeros(npad, device="cuda")
    norm = torch.zeros(1, device="cuda")
    ws = torch.empty(lib.yh_sqnorm_ws(npad) + 2, dtype=torch.float64, device="cuda")
    pr, mr, vr = p0.clone(), torch.zeros(npad), torch.zeros(npad)
    for step in (1, 2, 3):
        L.check(lib.yh_grad_sqnorm(g.data_ptr(), npad, 1.0 / world, norm.data_ptr(), ws.data_ptr(), st))
        gavg = g.cpu() / world
        total, coef = orc.clip_coef([gavg], 10.0)
        assert abs(float(norm) - total) / total < 1e-6
        L.check(lib.yh_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), npad, 1e-3, 0.9, 0.999, 1e-8, step, 10.0,
                                 norm.data_ptr(), 1.0 / world, st))
        gr = gavg * coef
        orc.adam_step(pr, gr, mr, vr, step, 1e-3)
        assert rel_err(g, gr) < 1e-6            # clipped gradient is written back, as clip_grad_norm_ does
        assert rel_err(p, pr) < 1e-6 and rel_err(m, mr) < 1e-6 and rel_err(v, vr) < 1e-6
        g = (torch.randn(npad) * 3).cuda()


@pytest.mark.parametrize("Cin,Cout,k,s,res,up,act", [(16, 32, 3, 1, True, False, True), (32, 16, 1, 1, False, True, True),
                                                    (8, 20, 1, 1, False, False, False), (64, 64, 3, 2, False, False, True)])
def test_fused_inference_conv_with_folded_bn(Cin, Cout, k, s, res, up, act):
    """yh_pack_fold_multi + yh_conv_fwd_fused == eval-mode conv -> BatchNorm -> SiLU (+residual) (+x2 nearest)."""
    import struct
    L = _lib()
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    B, H, W = 2, 12, 10
    torch.manual_seed(Cin * Cout + k)
    x = torch.randn(B, Cin, H, W)
    w = torch.randn(Cout, Cin, k, k) / (Cin * k * k) ** 0.5
    bias = torch.randn(Cout) * 0.1
    gamma, beta = torch.rand(Cout) + 0.5, torch.randn(Cout) * 0.2
    rm, rv = torch.randn(Cout) * 0.3, torch.rand(Cout) + 0.5
    y = F.conv2d(x.double(), w.double(), bias.double(), s, k // 2)
    if act:
        y = F.silu(F.batch_norm(y, rm.double(), rv.double(), gamma.double(), beta.double(), False, 0.1, 1e-5))
    Ho, Wo = y.shape[2], y.shape[3]
    r = torch.randn(B, Cout, Ho, Wo) if res else None
    if res:
        y = y + r.double()
    if up:
        y = F.interpolate(y, scale_factor=2, mode="nearest")
    dev = "cuda"
    ldwf = rup4(Cout)
    wd, bd = w.to(dev), bias.to(dev)
    g_, b_, rm_, rv_ = gamma.to(dev), beta.to(dev), rm.to(dev), rv.to(dev)
    wf = torch.empty(k * k * Cin * ldwf, device=dev)
    fb = torch.empty(Cout, device=dev)
    rec = struct.pack("<QQQQQQQQiiiiif", wd.data_ptr(), bd.data_ptr(), g_.data_ptr() if act else 0, b_.data_ptr() if act else 0,
                      rm_.data_ptr() if act else 0, rv_.data_ptr() if act else 0, wf.data_ptr(), fb.data_ptr(), Cout, Cin, k * k,
                      Cin, ldwf, 1e-5)
    tab = torch.frombuffer(bytearray(rec), dtype=torch.uint8).to(dev)
    L.check(lib.yh_pack_fold_multi(tab.data_ptr(), 1, st))
    xg = nhwc(x)
    rg = nhwc(r) if res else None
    f = 2 if up else 1
    out = torch.empty(B, Ho * f, Wo * f, Cout, device=dev)
    L.check(lib.yh_conv_fwd_fused(xg.data_ptr(), Cin, wf.data_ptr(), ldwf, fb.data_ptr(), rg.data_ptr() if res else None, Cout,
                                  out.data_ptr(), Cout, B, H, W, Cin, Cout, k, s, int(act), int(up), st))
    assert rel_err(out.permute(0, 3, 1, 2), y) < 1e-4


@pytest.mark.parametrize("B,H,W", [(2, 64, 64), (1, 50, 38), (3, 33, 47), (8, 250, 320)])
def test_first_layer_on_the_narrow_kernel(B, H, W):
    """The first layer (3->16, k3 s2 p1, NHWC4 input) on narrow_conv_kernel<4,16,2> against fp64 torch and the BatchNorm
    partial-sum contract; odd sizes and a ragged last workgroup included.  (The VALU kernel it replaced was retired.)"""
    L = _lib()
    lib = L.lib()
    torch.manual_seed(B * 100 + H)
    x = torch.randn(B, 3, H, W)
    w = torch.randn(16, 3, 3, 3) / 27 ** 0.5
    bias = torch.randn(16)
    ref = F.conv2d(x.double(), w.double(), bias.double(), 2, 1)
    Ho, Wo = ref.shape[2], ref.shape[3]
    st = torch.cuda.current_stream().cuda_stream
    x4 = torch.zeros(B, H, W, 4, device="cuda")
    x4[..., :3] = x.permute(0, 2, 3, 1).cuda()
    wf = torch.empty(9 * 4 * 16, device="cuda")
    L.check(lib.yh_pack_weights(w.cuda().data_ptr(), wf.data_ptr(), None, 16, 3, 3, 4, 16, 4, st))
    ldy = 24
    # the same layer on the direct MFMA kernel (CIN = 4 padded channels): output, untouched padding columns, partial sums
    assert lib.yh_conv_narrow_ok(4, 16, 3, 2) == 1
    y3 = torch.full((B, Ho, Wo, ldy), -1.0, device="cuda")
    nb3 = lib.yh_conv_narrow_blocks(B, H, W, 4, 2)
    part3 = torch.zeros(nb3, 2, 16, device="cuda")
    L.check(lib.yh_conv_narrow(x4.data_ptr(), 4, wf.data_ptr(), 16, bias.cuda().data_ptr(), y3.data_ptr(), ldy, part3.data_ptr(),
                               B, H, W, 4, 16, 2, 0, 0, st), "narrow first layer")
    assert rel_err(y3[..., :16].permute(0, 3, 1, 2), ref) < 1e-5
    assert float(y3[..., 16:].min()) == -1.0 == float(y3[..., 16:].max())
    s3 = part3.double().sum(0).cpu()
    od = y3[..., :16].permute(0, 3, 1, 2).double().cpu()
    assert float((s3[0] - od.sum((0, 2, 3))).abs().max() / od.sum((0, 2, 3)).abs().max()) < 1e-5
    assert float((s3[1] - (od * od).sum((0, 2, 3))).abs().max() / (od * od).sum((0, 2, 3)).abs().max()) < 1e-5


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 32, 32, 16, 32), (1, 17, 22, 16, 32), (2, 10, 12, 8, 24), (1, 8, 8, 32, 64)])
def test_stride2_backward_data_merged_parities(B, H, W, Cin, Cout):
    """yh_conv_bwd_data_s2m (column parities merged into the channel axis) = the input gradient of a 3x3 stride-2 conv:
    against fp64 torch and the generic four-class kernel; odd heights, accumulate."""
    L = _lib()
    lib = L.lib()
    torch.manual_seed(H * 7 + Cin)
    w = torch.randn(Cout, Cin, 3, 3) / (Cin * 9) ** 0.5
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    dy = torch.randn(B, Cout, Ho, Wo)
    ref = torch.nn.grad.conv2d_input((B, Cin, H, W), w.double(), dy.double(), stride=2, padding=1)
    st = torch.cuda.current_stream().cuda_stream
    ldw = rup4(2 * Cin)
    wbm = torch.empty(6 * Cout * ldw, device="cuda")
    L.check(lib.yh_pack_weights_s2m(w.cuda().data_ptr(), wbm.data_ptr(), Cout, Cin, ldw, st))
    dyd = nhwc(dy)
    dx = torch.full((B, H, W, Cin), 3.0, device="cuda")
    L.check(lib.yh_conv_bwd_data_s2m(dyd.data_ptr(), Cout, wbm.data_ptr(), ldw, dx.data_ptr(), Cin, B, H, W, Cin, Cout, 0, st))
    assert rel_err(dx.permute(0, 3, 1, 2), ref) < 1e-4
    L.check(lib.yh_conv_bwd_data_s2m(dyd.data_ptr(), Cout, wbm.data_ptr(), ldw, dx.data_ptr(), Cin, B, H, W, Cin, Cout, 1, st))
    assert rel_err(dx.permute(0, 3, 1, 2), 2 * ref) < 1e-4
    wb = torch.empty(9 * Cout * rup4(Cin), device="cuda")
    L.check(lib.yh_pack_weights(w.cuda().data_ptr(), None, wb.data_ptr(), Cout, Cin, 3, Cin, rup4(Cout), rup4(Cin), st))
    dx2 = torch.empty(B, H, W, Cin, device="cuda")
    L.check(lib.yh_conv_bwd_data(dyd.data_ptr(), Cout, wb.data_ptr(), rup4(Cin), dx2.data_ptr(), Cin, B, H, W, Cin, Cout, 3, 2, 0, st))
    dx3 = torch.empty(B, H, W, Cin, device="cuda")
    L.check(lib.yh_conv_bwd_data_s2m(dyd.data_ptr(), Cout, wbm.data_ptr(), ldw, dx3.data_ptr(), Cin, B, H, W, Cin, Cout, 0, st))
    assert rel_err(dx3, dx2) < 1e-5
    assert lib.yh_conv_bwd_data_s2m(dyd.data_ptr(), Cout, wbm.data_ptr(), ldw, dx3.data_ptr(), Cin + 4, B, H, W, Cin, Cout, 0, st) != 0


@pytest.mark.parametrize("B,H,W,Cin,Cout,s,res,up", [(1, 20, 20, 256, 256, 1, True, False), (1, 40, 40, 64, 64, 1, False, True),
                                                      (1, 21, 13, 32, 48, 2, False, False), (2, 16, 16, 128, 128, 1, False, False)])
def test_small_m_tap_split_inference_conv(B, H, W, Cin, Cout, s, res, up):
    """yh_conv_fwd_fused_splitk (K chunk ranges dealt to S workgroups per tile, last arriver adds the slabs in fixed order) =
    yh_conv_fwd_fused = fp64 torch, with residual / upsample / odd sizes; repeated calls on one workspace agree bitwise (the
    reducer leaves the tickets zero); it is a plain call when the layer is large or no workspace is given."""
    L = _lib()
    lib = L.lib()
    torch.manual_seed(H + Cin + s)
    x = torch.randn(B, Cin, H, W)
    w = torch.randn(Cout, Cin, 3, 3) / (Cin * 9) ** 0.5
    bias = torch.randn(Cout)
    ref = F.silu(F.conv2d(x.double(), w.double(), bias.double(), s, 1))
    Ho, Wo = ref.shape[2], ref.shape[3]
    r = torch.randn(B, Cout, Ho, Wo) if res else None
    if res:
        ref = ref + r.double()
    if up:
        ref = F.interpolate(ref, scale_factor=2, mode="nearest")
    st = torch.cuda.current_stream().cuda_stream
    wf = torch.empty(9 * Cin * rup4(Cout), device="cuda")
    L.check(lib.yh_pack_weights(w.cuda().data_ptr(), wf.data_ptr(), None, Cout, Cin, 3, Cin, rup4(Cout), rup4(Cin), st))
    xd, bd = nhwc(x), bias.cuda()
    rd = nhwc(r) if res else None
    f = 2 if up else 1
    nws = lib.yh_conv_fwd_fused_ws(B, H, W, Cin, Cout, 3, s)
    assert nws > 0
    ws = torch.zeros(nws, device="cuda")            # contract: the tickets behind the slabs are zero before the first call
    y1 = torch.empty(B, Ho * f, Wo * f, Cout, device="cuda")
    y2 = torch.empty_like(y1)
    args = (xd.data_ptr(), Cin, wf.data_ptr(), rup4(Cout), bd.data_ptr(), rd.data_ptr() if res else None, Cout if res else 0)
    L.check(lib.yh_conv_fwd_fused_splitk(*args, y1.data_ptr(), Cout, ws.data_ptr(), nws, B, H, W, Cin, Cout, 3, s, 1, int(up), st))
    L.check(lib.yh_conv_fwd_fused(*args, y2.data_ptr(), Cout, B, H, W, Cin, Cout, 3, s, 1, int(up), st))
    assert rel_err(y1.permute(0, 3, 1, 2), ref) < 1e-5 and rel_err(y1, y2) < 1e-5
    for _ in range(3):                                # tickets were reset: same bits again, on the same workspace
        y4 = torch.empty_like(y1)
        L.check(lib.yh_conv_fwd_fused_splitk(*args, y4.data_ptr(), Cout, ws.data_ptr(), nws, B, H, W, Cin, Cout, 3, s, 1, int(up), st))
        assert torch.equal(y4, y1)
    y3 = torch.empty_like(y1)
    L.check(lib.yh_conv_fwd_fused_splitk(*args, y3.data_ptr(), Cout, None, 0, B, H, W, Cin, Cout, 3, s, 1, int(up), st))
    assert torch.equal(y3, y2)
    assert lib.yh_conv_fwd_fused_splitk(*args, y3.data_ptr(), Cout, ws.data_ptr(), nws - 1, B, H, W, Cin, Cout, 3, s, 1, int(up), st) != 0
    assert lib.yh_conv_fwd_fused_ws(64, 160, 160, 32, 32, 3, 1) == 0


@pytest.mark.parametrize("B,H,W,Cin,Cout,s", [(2, 24, 64, 16, 16, 1), (3, 23, 45, 16, 16, 1), (2, 32, 96, 16, 32, 2), (2, 21, 67, 16, 32, 2),
                                             (1, 8, 32, 16, 16, 1),
                                             # more patches than persistent workgroups (several patches per workgroup, ragged edges)
                                             (16, 125, 160, 16, 16, 1), (16, 128, 190, 16, 32, 2)])
def test_narrow_direct_conv_kernel(B, H, W, Cin, Cout, s):
    """yh_conv_narrow (LDS halo patch + register-resident filter on the 16x16x4 MFMA) = fp64 torch for the two narrow layer
    shapes: forward with bias + BatchNorm partial sums into a channel slice of a wider buffer, ragged patches, and the
    stride-1 backward-data form (flipped taps, backward pack, accumulate)."""
    L = _lib()
    lib = L.lib()
    assert lib.yh_conv_narrow_ok(Cin, Cout, 3, s) == 1 and lib.yh_conv_narrow_ok(32, 32, 3, 1) == 0
    torch.manual_seed(H * W + s)
    x = torch.randn(B, Cin, H, W)
    w = torch.randn(Cout, Cin, 3, 3) / (Cin * 9) ** 0.5
    bias = torch.randn(Cout)
    ref = F.conv2d(x.double(), w.double(), bias.double(), s, 1)
    Ho, Wo = ref.shape[2], ref.shape[3]
    st = torch.cuda.current_stream().cuda_stream
    ldx = Cin + 4
    xbuf = torch.full((B, H, W, ldx), 9.0, device="cuda")
    xbuf[..., 4:] = nhwc(x)
    xv = xbuf.view(-1)[4:]
    ldwf, ldwb = rup4(Cout), rup4(Cin)
    wf = torch.empty(9 * Cin * ldwf, device="cuda")
    wb = torch.empty(9 * Cout * ldwb, device="cuda")
    L.check(lib.yh_pack_weights(w.cuda().data_ptr(), wf.data_ptr(), wb.data_ptr(), Cout, Cin, 3, Cin, ldwf, ldwb, st))
    ldy = Cout + 8
    ybuf = torch.full((B, Ho, Wo, ldy), 7.0, device="cuda")
    yv = ybuf.view(-1)[4:]
    nblk = lib.yh_conv_narrow_blocks(B, H, W, Cin, s)
    part = torch.zeros(nblk * 2 * Cout, device="cuda")
    L.check(lib.yh_conv_narrow(xv.data_ptr(), ldx, wf.data_ptr(), ldwf, bias.cuda().data_ptr(), yv.data_ptr(), ldy, part.data_ptr(),
                               B, H, W, Cin, Cout, s, 0, 0, st), "narrow fwd")
    y = ybuf[..., 4:4 + Cout].permute(0, 3, 1, 2)
    assert rel_err(y, ref) < 1e-5
    assert bool((ybuf[..., :4] == 7.0).all()) and bool((ybuf[..., 4 + Cout:] == 7.0).all())
    ps = part.view(nblk, 2, Cout).double().sum(0).cpu()
    assert rel_err(ps[0], ref.sum((0, 2, 3))) < 1e-4 and rel_err(ps[1], (ref * ref).sum((0, 2, 3))) < 1e-5
    if s == 1:
        dy = torch.randn(B, Cout, Ho, Wo)
        want = F.conv_transpose2d(dy.double(), w.double(), None, 1, 1)
        dx = torch.full((B, H, W, Cin), 1.0, device="cuda")
        dyg = nhwc(dy)
        L.check(lib.yh_conv_narrow(dyg.data_ptr(), Cout, wb.data_ptr(), ldwb, None, dx.data_ptr(), Cin, None, B, H, W, Cout, Cin, 1, 1, 0, st),
                "narrow dgrad")
        assert rel_err(dx.permute(0, 3, 1, 2), want) < 1e-5
        L.check(lib.yh_conv_narrow(dyg.data_ptr(), Cout, wb.data_ptr(), ldwb, None, dx.data_ptr(), Cin, None, B, H, W, Cout, Cin, 1, 1, 1, st),
                "narrow dgrad acc")
        assert rel_err(dx.permute(0, 3, 1, 2), 2 * want) < 1e-5
    else:       # stride 2: the parity-class backward-data kernel against conv_transpose2d and against the generic kernel
        assert lib.yh_conv_narrow_dgrad_s2_ok(Cin, Cout) == 1
        dy = torch.randn(B, Cout, Ho, Wo)
        want = F.conv_transpose2d(dy.double(), w.double(), None, 2, 1, output_padding=(H + 2 - 3 - (Ho - 1) * 2, W + 2 - 3 - (Wo - 1) * 2))
        dxbuf = torch.full((B, H, W, Cin + 4), 3.0, device="cuda")
        dxv = dxbuf.view(-1)[4:]
        dyg = nhwc(dy)
        L.check(lib.yh_conv_narrow_dgrad_s2(dyg.data_ptr(), Cout, wb.data_ptr(), ldwb, dxv.data_ptr(), Cin + 4, B, H, W, Cin, Cout, 0, st), "dgrad s2")
        assert rel_err(dxbuf[..., 4:].permute(0, 3, 1, 2), want) < 1e-5 and bool((dxbuf[..., :4] == 3.0).all())
        L.check(lib.yh_conv_narrow_dgrad_s2(dyg.data_ptr(), Cout, wb.data_ptr(), ldwb, dxv.data_ptr(), Cin + 4, B, H, W, Cin, Cout, 1, st), "dgrad s2 acc")
        assert rel_err(dxbuf[..., 4:].permute(0, 3, 1, 2), 2 * want) < 1e-5


@pytest.mark.parametrize("B,H,W,Cin,creal,Cout,s", [(2, 24, 64, 16, 16, 16, 1), (3, 23, 45, 16, 16, 16, 1), (2, 32, 96, 16, 16, 32, 2),
                                                   (2, 21, 67, 16, 16, 32, 2), (2, 64, 96, 4, 3, 16, 2), (3, 37, 51, 4, 3, 16, 2),
                                                   (16, 128, 160, 16, 16, 16, 1), (8, 256, 320, 4, 3, 16, 2)])
def test_narrow_weight_gradient_kernel(B, H, W, Cin, creal, Cout, s):
    """yh_conv_narrow_bwd_weight (pixels as the MFMA k dimension, persistent workgroups, fixed-order slab sum) = fp64 torch
    weight gradient for the three narrow layer shapes: strided views of wider buffers, ragged patches, more patches than
    workgroups (last case), bitwise reproducible, and equal to the generic yh_conv_bwd_weight to rounding."""
    L = _lib()
    lib = L.lib()
    assert lib.yh_conv_narrow_bwd_weight_ok(Cin, creal, Cout, 3, s) == 1 and lib.yh_conv_narrow_bwd_weight_ok(32, 32, 32, 3, 1) == 0
    torch.manual_seed(H * W + s)
    x = torch.randn(B, creal, H, W)
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    dy = torch.randn(B, Cout, Ho, Wo)
    xd = x.double().requires_grad_(False)
    w = torch.zeros(Cout, creal, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(xd, w, None, s, 1).backward(dy.double())
    want = w.grad
    st = torch.cuda.current_stream().cuda_stream
    ldx = Cin + 4
    xbuf = torch.full((B, H, W, ldx), 9.0, device="cuda")
    xbuf[..., 4:] = 0.0
    xbuf[..., 4:4 + creal] = nhwc(x)
    xv = xbuf.view(-1)[4:]
    lddy = Cout + 8
    dbuf = torch.full((B, Ho, Wo, lddy), 5.0, device="cuda")
    dbuf[..., 4:4 + Cout] = nhwc(dy)
    dv = dbuf.view(-1)[4:]
    nws = lib.yh_conv_narrow_bwd_weight_ws(B, H, W, Cin, Cout, s)
    ws = torch.full((nws,), float("nan"), device="cuda")
    dw = torch.full((Cout, creal, 3, 3), 7.0, device="cuda")
    db = torch.full((Cout,), 7.0, device="cuda")                          # bias gradient = column sums of dY, from the same launch
    args = (xv.data_ptr(), ldx, dv.data_ptr(), lddy)
    L.check(lib.yh_conv_narrow_bwd_weight(*args, dw.data_ptr(), db.data_ptr(), ws.data_ptr(), nws, B, H, W, Cin, creal, Cout, s, st), "narrow wgrad")
    assert rel_err(dw, want) < 1e-5
    assert rel_err(db, dy.double().sum((0, 2, 3))) < 1e-5
    dw2 = torch.empty_like(dw)
    L.check(lib.yh_conv_narrow_bwd_weight(*args, dw2.data_ptr(), None, ws.data_ptr(), nws, B, H, W, Cin, creal, Cout, s, st), "narrow wgrad")
    assert torch.equal(dw, dw2)
    nws_g = lib.yh_conv_bwd_weight_ws(B, H, W, Cin, Cout, 3, s)
    ws_g = torch.empty(nws_g, device="cuda")
    dw3 = torch.empty_like(dw)
    L.check(lib.yh_conv_bwd_weight(*args, dw3.data_ptr(), ws_g.data_ptr(), nws_g, B, H, W, Cin, creal, Cout, 3, s, st), "generic wgrad")
    assert rel_err(dw, dw3) < 1e-5
    assert lib.yh_conv_narrow_bwd_weight(*args, dw2.data_ptr(), None, ws.data_ptr(), nws - 1, B, H, W, Cin, creal, Cout, s, st) != 0


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,s,res,up,act", [
    (1, 20, 20, 256, 256, 3, 1, True, False, True),      # K = 2304: 288 chunks over eight waves
    (1, 40, 40, 64, 64, 3, 1, False, True, True),        # x2 upsample on write
    (1, 9, 11, 8, 24, 3, 1, False, False, True),         # ONE chunk per tap, 9 chunks: waves past the end of K, ragged M and N
    (2, 7, 5, 16, 18, 1, 1, False, False, False),        # two chunks, four waves, head-like N = 18, no activation
    (1, 13, 10, 24, 255, 1, 1, True, False, True),       # three chunks, N = 255
    (1, 21, 17, 32, 40, 3, 2, False, False, True),       # stride 2, odd sizes
    (3, 8, 8, 72, 32, 3, 1, True, True, True),           # 81 chunks, residual + upsample
])
def test_latency_inference_conv(B, H, W, Cin, Cout, k, s, res, up, act):
    """conv_lat.hip (batch-1 inference layers: K split over the waves of a workgroup, no split-K slabs): conv + bias + SiLU
    (+ residual) (+ x2 upsample) against an fp64 evaluation, through the C ABI; shapes that leave waves without K chunks, rows
    and columns past the tile, padding on every side."""
    L = _lib()
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    torch.manual_seed(B * 100 + H + Cin)
    x = torch.randn(B, Cin, H, W)
    w = torch.randn(Cout, Cin, k, k) / (Cin * k * k) ** 0.5
    bias = torch.randn(Cout)
    Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
    r = torch.randn(B, Cout, Ho, Wo) if res else None
    ref = F.conv2d(x.double(), w.double(), bias.double(), stride=s, padding=k // 2)
    if act:
        ref = ref * torch.sigmoid(ref)
    if res:
        ref = ref + r.double()
    if up:
        ref = ref.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)
    assert lib.yh_conv_lat_ok(B, H, W, Cin, Cout, k, s) == 1
    xd, wd, bd = nhwc(x), w.cuda(), bias.cuda()
    ldw = rup4(Cout)
    wq = torch.full((k * k * Cin * ldw,), float("nan"), device="cuda")       # every element the kernel reads must be written by the pack
    tab = torch.cat([torch.tensor([wd.data_ptr(), wq.data_ptr()], dtype=torch.int64).view(torch.uint8),
                     torch.tensor([Cout, Cin, k * k, ldw], dtype=torch.int32).view(torch.uint8)]).cuda()
    L.check(lib.yh_lat_pack_multi(tab.data_ptr(), 1, st))
    rd = nhwc(r) if res else None
    f = 2 if up else 1
    y = torch.full((B, Ho * f, Wo * f, Cout), float("nan"), device="cuda")
    L.check(lib.yh_conv_lat_fwd_fused(xd.data_ptr(), Cin, wq.data_ptr(), ldw, bd.data_ptr(), rd.data_ptr() if res else None, Cout,
                                      y.data_ptr(), Cout, B, H, W, Cin, Cout, k, s, int(act), int(up), st))
    torch.cuda.synchronize()
    assert rel_err(y.permute(0, 3, 1, 2), ref) < 1e-5
    assert lib.yh_conv_lat_ok(B, H, W, Cin + 4, Cout, k, s) == 0                 # Cin % 8 != 0: not this kernel's problem


@pytest.mark.parametrize("B,H,W,Cin,Cout,act", [
    (2, 32, 32, 32, 64, False),      # two column tiles, 64-pixel blocks
    (3, 20, 20, 64, 128, True),      # four column tiles; blocks span image boundaries (gap rows); input prologue
    (2, 9, 11, 16, 32, True),        # odd sizes: ragged blocks, padding on every side
    (1, 40, 40, 128, 256, False),    # two column workgroups per pixel block
    (5, 6, 6, 8, 96, True),          # one chunk, three column tiles of the last workgroup unused... Cout = 96: ragged column workgroup
])
def test_stride2_lds_forward(B, H, W, Cin, Cout, act):
    """conv_s2.hip: 3x3 stride-2 forward with the input patch staged through LDS, with and without the input prologue (producer
    BatchNorm scale / shift + SiLU, every fifth channel linear), output, bias and the BatchNorm partial sums against fp64."""
    L = _lib()
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    torch.manual_seed(B * 10 + H + Cin)
    x = torch.randn(B, Cin, H, W)
    w = torch.randn(Cout, Cin, 3, 3) / (Cin * 9) ** 0.5
    bias = torch.randn(Cout)
    assert lib.yh_conv_s2_ok(B, H, W, Cin, Cout) == 1
    xin = x.double()
    tab = None
    if act:
        sc, sh = torch.rand(Cin) + 0.5, torch.randn(Cin) * 0.3
        gate = (torch.arange(Cin) % 5 != 0).float()
        ldc = rup4(Cin) + 4
        tab = torch.zeros(3, ldc)
        tab[0, :Cin], tab[1, :Cin], tab[2, :Cin] = sc, sh, gate
        z = x.double() * sc.double()[None, :, None, None] + sh.double()[None, :, None, None]
        xin = torch.where(gate.bool()[None, :, None, None], z * torch.sigmoid(z), z)
        tab = tab.cuda()
    ref = F.conv2d(xin, w.double(), bias.double(), stride=2, padding=1)
    Ho, Wo = ref.shape[2], ref.shape[3]
    xd, wd, bd = nhwc(x), w.cuda(), bias.cuda()
    ldw = rup4(Cout)
    wq = torch.full((9 * Cin * ldw,), float("nan"), device="cuda")
    desc = torch.cat([torch.tensor([wd.data_ptr(), wq.data_ptr()], dtype=torch.int64).view(torch.uint8),
                      torch.tensor([Cout, Cin, 9, ldw], dtype=torch.int32).view(torch.uint8)]).cuda()
    L.check(lib.yh_lat_pack_multi(desc.data_ptr(), 1, st))
    nb = lib.yh_conv_s2_blocks(B, H, W, Cout)
    part = torch.full((nb, 2, Cout), float("nan"), device="cuda")
    y = torch.full((B, Ho, Wo, Cout), float("nan"), device="cuda")
    L.check(lib.yh_conv_s2_fwd_act(xd.data_ptr(), Cin, tab.data_ptr() if act else None, tab.shape[1] if act else 0, wq.data_ptr(), ldw,
                                   bd.data_ptr(), y.data_ptr(), Cout, part.data_ptr(), B, H, W, Cin, Cout, st))
    torch.cuda.synchronize()
    assert rel_err(y.permute(0, 3, 1, 2), ref) < 1e-5
    s = part.double().sum(0).cpu()
    assert float((s[0] - ref.sum((0, 2, 3))).abs().max() / ref.abs().sum((0, 2, 3)).max()) < 1e-5
    assert float((s[1] - (ref ** 2).sum((0, 2, 3))).abs().max() / (ref ** 2).sum((0, 2, 3)).max()) < 1e-5
