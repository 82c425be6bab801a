"""GPU parity of the pointwise (1x1) register-direct MFMA kernels through the C ABI against fp64 torch: forward (bias,
BatchNorm partial sums, views with ld > C, ragged M), the fused sibling pair (one GEMM, two outputs), backward-data
(single and two-source K, accumulate) and backward-weight (all vector-width variants, odd ld, determinism)."""
import struct

import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib():
    import yolo_from_scratch_amd._lib as L
    return L


def rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def rup4(c):
    return (c + 3) // 4 * 4


def pack(L, lib, w, wq_f, wq_b, ldwf, ldwb, koff=0, noff=0):
    Cout, Cin = w.shape
    tab = torch.frombuffer(bytearray(struct.pack("<QQQiiiiii", w.data_ptr(), wq_f.data_ptr() if wq_f is not None else 0,
                                                 wq_b.data_ptr() if wq_b is not None else 0, Cout, Cin, ldwf, ldwb, koff, noff)),
                           dtype=torch.uint8).cuda()
    L.check(lib.yh_pw_pack_multi(tab.data_ptr(), 1, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()


CASES = [(300, 32, 16), (1000, 64, 64), (777, 128, 128), (260, 256, 128), (513, 192, 64), (129, 64, 24), (4100, 32, 32), (96, 384, 136),
         # M >= 32768 with small K: the streaming form (weights in registers, fixed grid, one partial row per workgroup)
         (40003, 32, 32), (70001, 64, 64), (33000, 32, 128), (50000, 16, 16), (36000, 64, 24), (32768, 32, 16), (33000, 128, 32),
         # K >= 64 with more pixel tiles than persistent workgroups: the LDS-staged kernel walks several tiles per workgroup
         # (next tile prefetched, one partial row per workgroup), ragged last tile, N = 128 / 64 / 24 wave layouts
         (140001, 128, 128), (70003, 128, 64), (150001, 96, 24), (40000, 256, 256)]


@pytest.mark.parametrize("M,Cin,Cout", CASES)
def test_pointwise_forward_and_backward_data(M, Cin, Cout):
    L = _lib()
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    torch.manual_seed(M + Cin)
    x, w, bias = torch.randn(M, Cin), torch.randn(Cout, Cin) / Cin ** 0.5, torch.randn(Cout)
    dy = torch.randn(M, Cout)
    ref = x.double() @ w.double().t() + bias.double()
    dref = dy.double() @ w.double()
    wd = w.cuda()
    ldwf, ldwb = rup4(Cout), rup4(Cin)
    qf, qb = torch.zeros(Cin * ldwf, device="cuda"), torch.zeros(Cout * ldwb, device="cuda")
    pack(L, lib, wd, qf, qb, ldwf, ldwb)
    ldx, offx = Cin + 12, 8                    # the input is a channel slice of a wider buffer
    xb = torch.full((M, ldx), 5.0, device="cuda")
    xb[:, offx:offx + Cin] = x.cuda()
    ldy, offy = Cout + 4, 4
    yb = torch.full((M, ldy), -2.0, device="cuda")
    nblk = lib.yh_conv_pw_blocks(M, Cin, Cout)
    part = torch.zeros(nblk, 2, Cout, device="cuda")
    L.check(lib.yh_conv_pw_fwd(xb.data_ptr() + 4 * offx, ldx, qf.data_ptr(), ldwf, bias.cuda().data_ptr(), yb.data_ptr() + 4 * offy, ldy,
                               part.data_ptr(), M, Cin, Cout, st))
    y = yb[:, offy:offy + Cout]
    assert rel_err(y, ref) < 1e-5
    assert float(yb[:, :offy].min()) == -2.0 == float(yb[:, :offy].max())
    s = part.double().sum(0).cpu()
    assert float((s[0] - y.double().cpu().sum(0)).abs().max() / y.double().cpu().sum(0).abs().max()) < 1e-5
    assert float((s[1] - (y.double().cpu() ** 2).sum(0)).abs().max() / (y.double().cpu() ** 2).sum(0).abs().max()) < 1e-5
    if Cout % 8 == 0:
        dyd = dy.cuda()
        dxb = torch.full((M, ldx), 1.5, device="cuda")
        L.check(lib.yh_conv_pw_bwd_data(dyd.data_ptr(), Cout, None, 0, Cout, qb.data_ptr(), ldwb, dxb.data_ptr() + 4 * offx, ldx, M, Cin, 0, st))
        assert rel_err(dxb[:, offx:offx + Cin], dref) < 1e-5 and float(dxb[:, :offx].min()) == 1.5
        L.check(lib.yh_conv_pw_bwd_data(dyd.data_ptr(), Cout, None, 0, Cout, qb.data_ptr(), ldwb, dxb.data_ptr() + 4 * offx, ldx, M, Cin, 1, st))
        assert rel_err(dxb[:, offx:offx + Cin], 2 * dref) < 1e-5
    else:
        assert lib.yh_conv_pw_bwd_data(dy.cuda().data_ptr(), Cout, None, 0, Cout, qb.data_ptr(), ldwb, xb.data_ptr(), ldx, M, Cin, 0, st) != 0


@pytest.mark.parametrize("M,Cin,c1,c2", [(500, 64, 32, 32), (1300, 32, 16, 16), (260, 256, 128, 128), (333, 128, 64, 40),
                                         (45000, 64, 32, 32), (60001, 32, 16, 16),           # these two: streaming form
                                         (70001, 128, 64, 64), (50002, 128, 32, 32)])         # LDS-staged form, several tiles per workgroup
def test_pointwise_sibling_pair_forward_and_backward_data(M, Cin, c1, c2):
    """two convs reading the same x: fused forward (two outputs, own bias / partial sums) and fused backward-data (K from
    two tensors) equal the two separate convolutions"""
    L = _lib()
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    torch.manual_seed(M + c1)
    x = torch.randn(M, Cin)
    w1, w2 = torch.randn(c1, Cin) / Cin ** 0.5, torch.randn(c2, Cin) / Cin ** 0.5
    b2 = torch.randn(c2)
    ldw = rup4(c1 + c2)
    q = torch.zeros(Cin * ldw, device="cuda")
    ldwb = rup4(Cin)
    qb = torch.zeros((c1 + c2) * ldwb, device="cuda")
    w1d, w2d = w1.cuda(), w2.cuda()
    pack(L, lib, w1d, q, qb, ldw, ldwb, koff=0, noff=0)
    pack(L, lib, w2d, q, qb, ldw, ldwb, koff=c1, noff=c1)
    xd = x.cuda()
    y1, y2 = torch.empty(M, c1, device="cuda"), torch.full((M, c2 + 4), 9.0, device="cuda")
    nblk = lib.yh_conv_pw_blocks(M, Cin, c1 + c2)
    p1, p2 = torch.zeros(nblk, 2, c1, device="cuda"), torch.zeros(nblk, 2, c2, device="cuda")
    L.check(lib.yh_conv_pw_fwd2(xd.data_ptr(), Cin, q.data_ptr(), ldw, None, y1.data_ptr(), c1, p1.data_ptr(), c1, b2.cuda().data_ptr(),
                                y2.data_ptr(), c2 + 4, p2.data_ptr(), c2, M, Cin, st))
    r1, r2 = x.double() @ w1.double().t(), x.double() @ w2.double().t() + b2.double()
    assert rel_err(y1, r1) < 1e-5 and rel_err(y2[:, :c2], r2) < 1e-5 and float(y2[:, c2:].min()) == 9.0
    for part, r in ((p1, r1), (p2, r2)):
        s = part.double().sum(0).cpu()
        assert float((s[0] - r.sum(0)).abs().max() / r.sum(0).abs().max()) < 1e-4
        assert float((s[1] - (r * r).sum(0)).abs().max() / (r * r).sum(0).abs().max()) < 1e-4
    if c1 % 8 == 0 and c2 % 8 == 0:
        d1, d2 = torch.randn(M, c1), torch.randn(M, c2)
        dref = d1.double() @ w1.double() + d2.double() @ w2.double()
        d1d = d1.cuda()
        d2d = torch.zeros(M, c1, device="cuda")          # same ld for both sources
        d2d[:, :c2] = d2.cuda()
        dx = torch.empty(M, Cin, device="cuda")
        L.check(lib.yh_conv_pw_bwd_data(d1d.data_ptr(), c1, d2d.data_ptr(), c2, c1, qb.data_ptr(), ldwb, dx.data_ptr(), Cin, M, Cin, 0, st))
        assert rel_err(dx, dref) < 1e-5


@pytest.mark.parametrize("M,Cin,Cout,ldd", [(4096, 32, 16, 16), (3000, 32, 32, 32), (1111, 64, 64, 64), (2048, 128, 128, 128), (900, 256, 64, 64),
                                            (640, 64, 18, 18), (640, 128, 255, 255), (515, 192, 64, 72), (100, 16, 8, 8)])
def test_pointwise_backward_weight(M, Cin, Cout, ldd):
    L = _lib()
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    torch.manual_seed(M + Cout)
    x, dy = torch.randn(M, Cin), torch.randn(M, Cout)
    ref = dy.double().t() @ x.double()
    xd = x.cuda()
    dyb = torch.zeros(M, ldd, device="cuda")
    dyb[:, :Cout] = dy.cuda()
    nws = lib.yh_conv_pw_bwd_weight_ws(M, Cin, Cout)
    ws = torch.empty(nws, device="cuda")
    dw = torch.zeros(Cout, Cin, device="cuda")
    L.check(lib.yh_conv_pw_bwd_weight(xd.data_ptr(), Cin, dyb.data_ptr(), ldd, dw.data_ptr(), ws.data_ptr(), nws, M, Cin, Cout, st))
    assert rel_err(dw, ref) < 1e-5
    ws.fill_(7.0)
    dw2 = torch.zeros_like(dw)
    L.check(lib.yh_conv_pw_bwd_weight(xd.data_ptr(), Cin, dyb.data_ptr(), ldd, dw2.data_ptr(), ws.data_ptr(), nws, M, Cin, Cout, st))
    assert torch.equal(dw, dw2)
    assert lib.yh_conv_pw_bwd_weight(xd.data_ptr(), Cin, dyb.data_ptr(), ldd, dw2.data_ptr(), ws.data_ptr(), 10, M, Cin, Cout, st) != 0


@pytest.mark.parametrize("M,Cin,Cout", [(40003, 64, 128), (70001, 64, 64), (33000, 32, 32), (50000, 16, 16), (36000, 128, 64), (4100, 32, 24),
                                        (150001, 64, 96)])
def test_pointwise_forward_on_the_bf16_pipe(M, Cin, Cout):
    """yh_conv_pw_fwd_x6: the fp32 GEMM as six exact bf16 products per fp32 product (operands split into three bf16 terms), fp32
    accumulation.  Held to the fp64 reference at the fp32 kernels' tolerance AND to the claim that makes it admissible as an fp32 path:
    its error is not larger than 1.5x that of the fp32-MFMA kernel on the same data (measured: smaller).  Views with ld > C, ragged M and N,
    bias, BatchNorm partial sums, and the input prologue."""
    L = _lib()
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    torch.manual_seed(M + Cin + Cout)
    x, w, bias = torch.randn(M, Cin), torch.randn(Cout, Cin) / Cin ** 0.5, torch.randn(Cout)
    wd = w.cuda()
    ldwf = rup4(Cout)
    qf = torch.zeros(Cin * ldwf, device="cuda")
    pack(L, lib, wd, qf, None, ldwf, rup4(Cin))
    ldx, offx = Cin + 12, 8
    xb = torch.full((M, ldx), 5.0, device="cuda")
    xb[:, offx:offx + Cin] = x.cuda()
    ldy, offy = Cout + 4, 4
    nblk = lib.yh_conv_pw_x6_blocks(M, Cin, Cout)
    assert nblk > 0
    bd = bias.cuda()
    for act in (False, True):
        if act:
            ic = torch.zeros(3, rup4(Cin) + 4, device="cuda")
            ic[0] = torch.rand(rup4(Cin) + 4, device="cuda") + 0.5
            ic[1] = torch.randn(rup4(Cin) + 4, device="cuda") * 0.3
            ic[2, :Cin:2] = 1.0                                  # SiLU on every other channel
            z = x.double() * ic[0, :Cin].double().cpu() + ic[1, :Cin].double().cpu()
            xin = torch.where(ic[2, :Cin].cpu() != 0, z * torch.sigmoid(z), z)
            icp, icld = ic.data_ptr(), ic.shape[1]
        else:
            xin, icp, icld = x.double(), None, 0
        ref = xin @ w.double().t() + bias.double()
        outs = {}
        for name, fn in (("x6", lib.yh_conv_pw_fwd_x6), ("f32", lib.yh_conv_pw_fwd_act)):
            if name == "f32" and act and not lib.yh_conv_pw_prologue_ok(M, Cin, Cout):
                continue                                         # the fp32 kernel of this shape has no prologue: fp64 reference only
            yb = torch.full((M, ldy), 7.0, device="cuda")
            nb = nblk if name == "x6" else lib.yh_conv_pw_blocks(M, Cin, Cout)
            part = torch.zeros(nb * 2 * Cout, device="cuda")
            L.check(fn(xb.data_ptr() + 4 * offx, ldx, icp, icld, qf.data_ptr(), ldwf, bd.data_ptr(), yb.data_ptr() + 4 * offy, ldy,
                       part.data_ptr(), M, Cin, Cout, st), name)
            y = yb[:, offy:offy + Cout]
            assert bool((yb[:, :offy] == 7.0).all())
            outs[name] = (rel_err(y, ref), y, part.view(nb, 2, Cout).sum(0))
        e6, y6, p6 = outs["x6"]
        e32 = outs["f32"][0] if "f32" in outs else None
        assert e6 < (2e-5 if act else 2e-6), (e6, e32)
        assert e32 is None or e6 <= 1.5 * e32 + 1e-8, (e6, e32)
        yd = y6.double().cpu()
        assert rel_err(p6[0], yd.sum(0)) < 1e-4 and rel_err(p6[1], (yd * yd).sum(0)) < 1e-4
