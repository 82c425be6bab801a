#!/usr/bin/env python3
"""Timing of yh_conv_narrow_bwd_weight on the three benchmark-shape layers (GPU box only); prints ms, TFLOP/s and TB/s.

    python tools/narrow_w_bench.py [--iters 20]            (YH_NARROW_W_BLOCKS=<n> to sweep the persistent grid)
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    from yolo_from_scratch_amd import _lib as L
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    for (B, H, W, Cin, creal, Cout, s) in [(64, 160, 160, 16, 16, 16, 1), (64, 320, 320, 16, 16, 32, 2), (64, 640, 640, 4, 3, 16, 2)]:
        Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
        x = torch.randn(B, H, W, Cin, device="cuda")
        dy = torch.randn(B, Ho, Wo, Cout, device="cuda")
        nws = lib.yh_conv_narrow_bwd_weight_ws(B, H, W, Cin, Cout, s)
        ws = torch.empty(nws, device="cuda")
        dw = torch.empty(Cout, creal, 3, 3, device="cuda")

        def run():
            L.check(lib.yh_conv_narrow_bwd_weight(x.data_ptr(), Cin, dy.data_ptr(), Cout, dw.data_ptr(), None, ws.data_ptr(), nws, B, H, W, Cin,
                                                  creal, Cout, s, st))
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.iters
        gf = 2.0 * B * Ho * Wo * 9 * creal * Cout / 1e9
        mb = (x.numel() + dy.numel()) * 4 / 1e6
        print(f"{(B, H, W, Cin, Cout, s)}: {ms:.3f} ms  {gf / ms:.1f} TFLOP/s  {mb / ms / 1e3:.2f} TB/s  (incl. reduce)", flush=True)


if __name__ == "__main__":
    main()
