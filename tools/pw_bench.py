#!/usr/bin/env python3
"""Pointwise (1x1) kernels vs the generic gather-GEMM / wgrad kernels on the 1x1 layers of the model: correctness
against fp64 torch and per-layer time.

    python tools/pw_bench.py [--batch 64] [--iters 10]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

LAYERS = [(40, 40, 128, 128), (80, 80, 64, 64), (160, 160, 32, 16), (20, 20, 256, 128), (160, 160, 32, 32), (80, 80, 128, 32),
          (40, 40, 128, 64), (20, 20, 256, 256), (40, 40, 256, 64), (80, 80, 64, 32), (40, 40, 192, 64), (20, 20, 384, 128),
          (20, 20, 512, 256), (80, 80, 64, 18), (40, 40, 128, 18), (20, 20, 256, 18)]


def gf_of(M, Cin, Cout):
    return 2.0 * M * Cin * Cout / 1e9


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

    import struct
    tot = [0.0, 0.0]
    totg = [0.0, 0.0, 0.0, 0.0]
    for (H, W, Cin, Cout) in LAYERS:
        torch.manual_seed(H + Cin)
        M = B * H * W
        x = torch.randn(M, Cin, device=dev)
        dy = torch.randn(M, Cout, device=dev)
        nws0 = lib.yh_conv_bwd_weight_ws(B, H, W, Cin, Cout, 1, 1)
        nws1 = lib.yh_conv_pw_bwd_weight_ws(M, Cin, Cout)
        ws = torch.empty(max(nws0, nws1), device=dev)
        dw0, dw1 = torch.zeros(Cout, Cin, device=dev), torch.zeros(Cout, Cin, device=dev)
        w_dir = lambda: L.check(lib.yh_conv_bwd_weight(x.data_ptr(), Cin, dy.data_ptr(), Cout, dw0.data_ptr(), ws.data_ptr(), nws0,
                                                       B, H, W, Cin, Cin, Cout, 1, 1, st))
        w_pw = lambda: L.check(lib.yh_conv_pw_bwd_weight(x.data_ptr(), Cin, dy.data_ptr(), Cout, dw1.data_ptr(), ws.data_ptr(), nws1,
                                                         M, Cin, Cout, st))
        # forward / backward-data: generic gather-GEMM vs pointwise GEMM
        w = torch.randn(Cout, Cin, 1, 1, device=dev) / Cin ** 0.5
        bias = torch.randn(Cout, device=dev)
        ldwf, ldwb = (Cout + 3) // 4 * 4, (Cin + 3) // 4 * 4
        wf, wb = torch.empty(Cin * ldwf, device=dev), torch.empty(Cout * ldwb, device=dev)
        L.check(lib.yh_pack_weights(w.data_ptr(), wf.data_ptr(), wb.data_ptr(), Cout, Cin, 1, Cin, ldwf, ldwb, st))
        K8 = (Cout + 7) // 8 * 8
        qf, qb = torch.zeros(Cin * ldwf, device=dev), torch.zeros(K8 * ldwb, device=dev)
        tab = torch.frombuffer(bytearray(struct.pack("<QQQiiiiii", w.data_ptr(), qf.data_ptr(), qb.data_ptr(), Cout, Cin, ldwf, ldwb, 0, 0)),
                               dtype=torch.uint8).to(dev)
        L.check(lib.yh_pw_pack_multi(tab.data_ptr(), 1, st))
        y0, y1 = torch.empty(M, Cout, device=dev), torch.empty(M, Cout, device=dev)
        dx0, dx1 = torch.empty(M, Cin, device=dev), torch.empty(M, Cin, device=dev)
        p0 = torch.empty(lib.yh_conv_fwd_blocks(B, H, W, Cout, 1, 1) * 2 * Cout, device=dev)
        nb1 = lib.yh_conv_pw_blocks(M, Cin, Cout)
        p1 = torch.empty(nb1 * 2 * Cout, device=dev)
        f_dir = lambda: L.check(lib.yh_conv_fwd(x.data_ptr(), Cin, wf.data_ptr(), ldwf, bias.data_ptr(), y0.data_ptr(), Cout,
                                                p0.data_ptr(), B, H, W, Cin, Cout, 1, 1, st))
        f_pw = lambda: L.check(lib.yh_conv_pw_fwd(x.data_ptr(), Cin, qf.data_ptr(), ldwf, bias.data_ptr(), y1.data_ptr(), Cout,
                                                  p1.data_ptr(), M, Cin, Cout, st))
        b_dir = lambda: L.check(lib.yh_conv_bwd_data(dy.data_ptr(), Cout, wb.data_ptr(), ldwb, dx0.data_ptr(), Cin, B, H, W, Cin, Cout,
                                                     1, 1, 0, st))
        pw_ok = Cin % 8 == 0 and Cout % 8 == 0
        b_pw = lambda: L.check(lib.yh_conv_pw_bwd_data(dy.data_ptr(), Cout, None, 0, Cout, qb.data_ptr(), ldwb, dx1.data_ptr(), Cin, M, Cin,
                                                       0, st))
        tg = [timed(f_dir), timed(f_pw), timed(b_dir), timed(b_pw) if pw_ok else float("nan")]
        yref = x.double() @ w.view(Cout, Cin).double().t() + bias.double()
        dref = dy.double() @ w.view(Cout, Cin).double()
        ysc, dsc = yref.abs().max().item(), dref.abs().max().item()
        eg = [(y0 - yref).abs().max().item() / ysc, (y1 - yref).abs().max().item() / ysc, (dx0 - dref).abs().max().item() / dsc,
              (dx1 - dref).abs().max().item() / dsc if pw_ok else float("nan")]
        s1 = p1.view(nb1, 2, Cout).double().sum(0)
        es = ((s1[0] - y1.double().sum(0)).abs().max() / y1.double().sum(0).abs().max()).item()
        for i in range(4):
            totg[i] += tg[i] if tg[i] == tg[i] else tg[i - 1]
        print(f"   fwd generic {tg[0]:.3f} pw {tg[1]:.3f} ({gf_of(M, Cin, Cout) / tg[1]:.0f} TF, {4e-9 * M * (Cin + Cout) / tg[1]:.2f} TB/s) | dgrad generic {tg[2]:.3f} "
              f"pw {tg[3]:.3f} | err {eg[0]:.1e}/{eg[1]:.1e} {eg[2]:.1e}/{eg[3]:.1e} stats {es:.1e}")
        t0, t1 = timed(w_dir), timed(w_pw)
        ref = dy.double().t() @ x.double()
        sc = ref.abs().max().item()
        e0, e1 = (dw0 - ref).abs().max().item() / sc, (dw1 - ref).abs().max().item() / sc
        gf = 2.0 * M * Cin * Cout / 1e9
        gb = 4.0 * M * (Cin + Cout) / 1e9
        tot[0] += t0; tot[1] += t1
        print(f"{H}x{W} {Cin}->{Cout}: wgrad generic {t0:.3f} ms ({gf / t0:.0f} TF) pointwise {t1:.3f} ms ({gf / t1:.0f} TF, {gb / t1:.2f} TB/s) "
              f"err {e0:.1e}/{e1:.1e}", flush=True)
    print(f"total wgrad generic {tot[0]:.3f} pointwise {tot[1]:.3f} | fwd generic {totg[0]:.3f} pw {totg[1]:.3f} | dgrad generic {totg[2]:.3f} pw {totg[3]:.3f}")


if __name__ == "__main__":
    main()
