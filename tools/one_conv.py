#!/usr/bin/env python3
"""Run ONE convolution kernel repeatedly (for rocprofv3 --pmc / --kernel-trace runs on the GPU box).

    python tools/one_conv.py --shape 64,80,80,64,64,3,1 --what fwd --iters 20
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="64,80,80,64,64,3,1")
    ap.add_argument("--what", default="fwd", choices=["fwd", "dgrad", "wgrad"])
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    from yolo_from_scratch_amd import _lib as L
    lib = L.lib()
    B, H, W, Cin, Cout, k, s = [int(v) for v in a.shape.split(",")]
    p = k // 2
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    dev = "cuda"
    x = torch.randn(B, H, W, Cin, device=dev)
    w = torch.randn(Cout, Cin, k, k, device=dev) / (Cin * k * k) ** 0.5
    ldwf, ldwb = (Cout + 3) // 4 * 4, Cin
    wf = torch.empty(k * k * Cin * ldwf, device=dev)
    wb = torch.empty(k * k * Cout * ldwb, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    L.check(lib.yh_pack_weights(w.data_ptr(), wf.data_ptr(), wb.data_ptr(), Cout, Cin, k, Cin, ldwf, ldwb, st))
    y = torch.empty(B, Ho, Wo, Cout, device=dev)
    dy = torch.randn(B, Ho, Wo, Cout, device=dev)
    dx = torch.empty_like(x)
    part = torch.empty(lib.yh_conv_fwd_blocks(B, H, W, Cout, k, s) * 2 * Cout, device=dev)
    nws = lib.yh_conv_bwd_weight_ws(B, H, W, Cin, Cout, k, s)
    ws = torch.empty(nws, device=dev)
    dw = torch.empty_like(w)

    def run():
        if a.what == "fwd":
            L.check(lib.yh_conv_fwd(x.data_ptr(), Cin, wf.data_ptr(), ldwf, None, y.data_ptr(), Cout, part.data_ptr(), B, H, W, Cin, Cout, k, s, st))
        elif a.what == "dgrad":
            L.check(lib.yh_conv_bwd_data(dy.data_ptr(), Cout, wb.data_ptr(), ldwb, dx.data_ptr(), Cin, B, H, W, Cin, Cout, k, s, 0, st))
        else:
            L.check(lib.yh_conv_bwd_weight(x.data_ptr(), Cin, dy.data_ptr(), Cout, dw.data_ptr(), ws.data_ptr(), nws, B, H, W, Cin, Cin, Cout, k, s, st))
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    gf = 2.0 * B * Ho * Wo * Cin * Cout * k * k / 1e9
    print(f"{a.what} {a.shape}: {ms:.4f} ms  {gf / ms:.1f} TFLOP/s")


if __name__ == "__main__":
    main()
