#!/usr/bin/env python3
"""Winograd F(2x2,3x3) kernel vs the direct gather-GEMM on the 3x3 / stride-1 layers of the model:
correctness against torch conv2d (fp32) and per-layer time, forward and backward-data.

    python tools/wino_bench.py [--batch 64] [--iters 10]
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

LAYERS = [(160, 160, 32, 32), (80, 80, 64, 64), (40, 40, 64, 64), (40, 40, 128, 128), (20, 20, 128, 128),
          (20, 20, 256, 256), (80, 80, 128, 64), (20, 20, 64, 64), (26, 26, 64, 128)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    from yolo_from_scratch_amd import _lib as L
    lib = L.lib()
    dev = "cuda"
    st = torch.cuda.current_stream().cuda_stream
    B = a.batch

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

    for (H, W, Cin, Cout) in LAYERS:
        torch.manual_seed(H * 1000 + Cin)
        x = torch.randn(B, H, W, Cin, device=dev)
        w = torch.randn(Cout, Cin, 3, 3, device=dev) / (Cin * 9) ** 0.5
        bias = torch.randn(Cout, device=dev)
        dy = torch.randn(B, H, W, Cout, device=dev)
        ldwf, ldwb = (Cout + 3) // 4 * 4, Cin
        wf = torch.empty(9 * Cin * ldwf, device=dev)
        wb = torch.empty(9 * Cout * ldwb, device=dev)
        L.check(lib.yh_pack_weights(w.data_ptr(), wf.data_ptr(), wb.data_ptr(), Cout, Cin, 3, Cin, ldwf, ldwb, st))
        ldu, ldub = (Cout + 3) // 4 * 4, (Cin + 3) // 4 * 4
        U = torch.empty(16 * Cin * ldu, device=dev)
        Ub = torch.empty(16 * Cout * ldub, device=dev)
        L.check(lib.yh_wino_weights(w.data_ptr(), U.data_ptr(), Cout, Cin, ldu, 0, st))
        L.check(lib.yh_wino_weights(w.data_ptr(), Ub.data_ptr(), Cout, Cin, ldub, 1, st))
        y0, y1 = torch.empty(B, H, W, Cout, device=dev), torch.empty(B, H, W, Cout, device=dev)
        dx0, dx1 = torch.empty_like(x), torch.empty_like(x)
        p0 = torch.empty(lib.yh_conv_fwd_blocks(B, H, W, Cout, 3, 1) * 2 * Cout, device=dev)
        nb1 = lib.yh_conv_wino_blocks(B, H, W)
        p1 = torch.empty(nb1 * 2 * Cout, device=dev)

        f_dir = lambda: L.check(lib.yh_conv_fwd(x.data_ptr(), Cin, wf.data_ptr(), ldwf, bias.data_ptr(), y0.data_ptr(), Cout,
                                                p0.data_ptr(), B, H, W, Cin, Cout, 3, 1, st))
        f_win = lambda: L.check(lib.yh_conv_wino_fwd(x.data_ptr(), Cin, U.data_ptr(), ldu, bias.data_ptr(), y1.data_ptr(), Cout,
                                                     p1.data_ptr(), B, H, W, Cin, Cout, st))
        b_dir = lambda: L.check(lib.yh_conv_bwd_data(dy.data_ptr(), Cout, wb.data_ptr(), ldwb, dx0.data_ptr(), Cin, B, H, W,
                                                     Cin, Cout, 3, 1, 0, st))
        b_win = lambda: L.check(lib.yh_conv_wino_bwd_data(dy.data_ptr(), Cout, Ub.data_ptr(), ldub, dx1.data_ptr(), Cin, B, H,
                                                          W, Cin, Cout, 0, st))
        nws0 = lib.yh_conv_bwd_weight_ws(B, H, W, Cin, Cout, 3, 1)
        nws1 = lib.yh_conv_wino_bwd_weight_ws(B, H, W, Cin, Cout) if Cin % 32 == 0 and Cout % 32 == 0 else 0
        ws = torch.empty(max(nws0, nws1, 1), device=dev)
        dw0, dw1 = torch.zeros_like(w), torch.zeros_like(w)
        w_dir = lambda: L.check(lib.yh_conv_bwd_weight(x.data_ptr(), Cin, dy.data_ptr(), Cout, dw0.data_ptr(), ws.data_ptr(), nws0,
                                                       B, H, W, Cin, Cin, Cout, 3, 1, st))
        w_win = lambda: L.check(lib.yh_conv_wino_bwd_weight(x.data_ptr(), Cin, dy.data_ptr(), Cout, dw1.data_ptr(), ws.data_ptr(),
                                                            nws1, B, H, W, Cin, Cout, st))
        t = [timed(f) for f in (f_dir, f_win, b_dir, b_win)]
        tw = [timed(w_dir), timed(w_win) if nws1 else float("nan")]
        wref = torch.nn.grad.conv2d_weight(x.permute(0, 3, 1, 2).double(), w.shape, dy.permute(0, 3, 1, 2).double(), padding=1)
        wsc = wref.abs().max().item()
        ew = [(dw0 - wref).abs().max().item() / wsc, (dw1 - wref).abs().max().item() / wsc if nws1 else float("nan")]
        xr = x.permute(0, 3, 1, 2).double()
        ref = F.conv2d(xr, w.double(), bias.double(), padding=1).permute(0, 2, 3, 1)
        dref = F.conv_transpose2d(dy.permute(0, 3, 1, 2).double(), w.double(), padding=1).permute(0, 2, 3, 1)
        sc, dsc = ref.abs().max().item(), dref.abs().max().item()
        e = [((y0 - ref).abs().max().item()) / sc, ((y1 - ref).abs().max().item()) / sc,
             ((dx0 - dref).abs().max().item()) / dsc, ((dx1 - dref).abs().max().item()) / dsc]
        s1 = p1.view(nb1, 2, Cout).double().sum(0)
        es = max(((s1[0] - y1.double().sum((0, 1, 2))).abs().max() / y1.double().sum((0, 1, 2)).abs().max()).item(),
                 ((s1[1] - (y1.double() ** 2).sum((0, 1, 2))).abs().max() / (y1.double() ** 2).sum((0, 1, 2)).abs().max()).item())
        gf = 2.0 * B * H * W * Cin * Cout * 9 / 1e9
        print(f"{H}x{W} {Cin}->{Cout}: fwd direct {t[0]:.3f} ms ({gf / t[0]:.0f} TF) wino {t[1]:.3f} ms ({gf / t[1]:.0f} TF-eq) | "
              f"dgrad direct {t[2]:.3f} wino {t[3]:.3f} | wgrad direct {tw[0]:.3f} wino {tw[1]:.3f} err {ew[0]:.1e}/{ew[1]:.1e} | err fwd {e[0]:.1e}/{e[1]:.1e} dgrad {e[2]:.1e}/{e[3]:.1e} stats {es:.1e}",
              flush=True)


if __name__ == "__main__":
    main()
