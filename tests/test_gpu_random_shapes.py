"""Randomised shape sweep of the specialised convolution kernels against fp64 torch (seeded: the same 40 cases every run).
Catches tile-edge mistakes the hand-picked cases might miss: tiny maps, single-tile grids, channel counts just above a tile."""
import random
import struct

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


def _cases(n, seed):
    rng = random.Random(seed)
    out = []
    for _ in range(n):
        B = rng.choice([1, 1, 2, 3])
        H, W = 2 * rng.randint(2, 13), 2 * rng.randint(2, 13)
        Cin, Cout = 16 * rng.randint(1, 9), 16 * rng.randint(1, 9)
        out.append((B, H, W, Cin, Cout))
    return out


@pytest.mark.parametrize("case", _cases(20, 7))
def test_winograd_random_shapes(case):
    L = _lib()
    lib = L.lib()
    B, H, W, Cin, Cout = case
    st = torch.cuda.current_stream().cuda_stream
    torch.manual_seed(sum(case))
    x, w = torch.randn(B, Cin, H, W), torch.randn(Cout, Cin, 3, 3) / (Cin * 9) ** 0.5
    dy = torch.randn(B, Cout, H, W)
    wd = w.cuda()
    U, Ub = torch.empty(16 * Cin * rup4(Cout), device="cuda"), torch.empty(16 * Cout * rup4(Cin), device="cuda")
    L.check(lib.yh_wino_weights(wd.data_ptr(), U.data_ptr(), Cout, Cin, rup4(Cout), 0, st))
    L.check(lib.yh_wino_weights(wd.data_ptr(), Ub.data_ptr(), Cout, Cin, rup4(Cin), 1, st))
    xd, dyd = x.permute(0, 2, 3, 1).contiguous().cuda(), dy.permute(0, 2, 3, 1).contiguous().cuda()
    y, dx = torch.empty(B, H, W, Cout, device="cuda"), torch.empty(B, H, W, Cin, device="cuda")
    L.check(lib.yh_conv_wino_fwd(xd.data_ptr(), Cin, U.data_ptr(), rup4(Cout), None, y.data_ptr(), Cout, None, B, H, W, Cin, Cout, st))
    L.check(lib.yh_conv_wino_bwd_data(dyd.data_ptr(), Cout, Ub.data_ptr(), rup4(Cin), dx.data_ptr(), Cin, B, H, W, Cin, Cout, 0, st))
    assert rel_err(y.permute(0, 3, 1, 2), F.conv2d(x.double(), w.double(), None, 1, 1)) < 1e-4
    assert rel_err(dx.permute(0, 3, 1, 2), F.conv_transpose2d(dy.double(), w.double(), None, 1, 1)) < 1e-4
    if Cin % 32 == 0 and Cout % 32 == 0 and W >= 4:
        nws = lib.yh_conv_wino_bwd_weight_ws(B, H, W, Cin, Cout)
        ws, dw = torch.empty(nws, device="cuda"), torch.zeros(Cout, Cin, 3, 3, device="cuda")
        L.check(lib.yh_conv_wino_bwd_weight(xd.data_ptr(), Cin, dyd.data_ptr(), Cout, dw.data_ptr(), ws.data_ptr(), nws, B, H, W, Cin, Cout, st))
        assert rel_err(dw, torch.nn.grad.conv2d_weight(x.double(), w.shape, dy.double(), padding=1)) < 1e-4


@pytest.mark.parametrize("case", _cases(20, 11))
def test_pointwise_random_shapes(case):
    L = _lib()
    lib = L.lib()
    B, H, W, Cin, Cout = case
    Cin, Cout = Cin // 2, Cout // 2                  # multiples of 8
    M = B * H * W
    st = torch.cuda.current_stream().cuda_stream
    torch.manual_seed(sum(case) + 3)
    x, w, dy = torch.randn(M, Cin), torch.randn(Cout, Cin) / Cin ** 0.5, torch.randn(M, Cout)
    wd = w.cuda()
    ldwf, ldwb = rup4(Cout), rup4(Cin)
    qf, qb = torch.zeros(Cin * ldwf, device="cuda"), torch.zeros(Cout * ldwb, device="cuda")
    tab = torch.frombuffer(bytearray(struct.pack("<QQQiiiiii", wd.data_ptr(), qf.data_ptr(), qb.data_ptr(), Cout, Cin, ldwf, ldwb, 0, 0)),
                           dtype=torch.uint8).cuda()
    L.check(lib.yh_pw_pack_multi(tab.data_ptr(), 1, st))
    xd, dyd = x.cuda(), dy.cuda()
    y, dx = torch.empty(M, Cout, device="cuda"), torch.empty(M, Cin, device="cuda")
    L.check(lib.yh_conv_pw_fwd(xd.data_ptr(), Cin, qf.data_ptr(), ldwf, None, y.data_ptr(), Cout, None, M, Cin, Cout, st))
    L.check(lib.yh_conv_pw_bwd_data(dyd.data_ptr(), Cout, None, 0, Cout, qb.data_ptr(), ldwb, dx.data_ptr(), Cin, M, Cin, 0, st))
    assert rel_err(y, x.double() @ w.double().t()) < 1e-5
    assert rel_err(dx, dy.double() @ w.double()) < 1e-5
    nws = lib.yh_conv_pw_bwd_weight_ws(M, Cin, Cout)
    ws, dw = torch.empty(nws, device="cuda"), torch.zeros(Cout, Cin, device="cuda")
    L.check(lib.yh_conv_pw_bwd_weight(xd.data_ptr(), Cin, dyd.data_ptr(), Cout, dw.data_ptr(), ws.data_ptr(), nws, M, Cin, Cout, st))
    assert rel_err(dw, dy.double().t() @ x.double()) < 1e-5
