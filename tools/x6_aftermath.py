#!/usr/bin/env python3
"""What a dense bf16-MFMA launch does to the launches AFTER it: the duration of one fixed fp32 Winograd layer and of one
HBM-bound BatchNorm pass, each measured with its own event pair, when the launch in front of it is (a) the fp32-MFMA 1x1 forward,
(b) the split-bf16 form of the same GEMM, (c) nothing.

    python tools/x6_aftermath.py [--iters 30]
"""
import argparse
import os
import struct
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    a = ap.parse_args()
    from yolo_from_scratch_amd import _lib as L
    lib = L.lib()
    dev = "cuda"
    st = torch.cuda.current_stream().cuda_stream
    torch.manual_seed(0)
    B, H, W, C = 64, 80, 80, 64
    M = B * H * W
    # the 1x1 sibling pair 64 -> 128 at 80^2 (the launch in front)
    x = torch.randn(M, C, device=dev)
    w1 = torch.randn(128, C, device=dev) / 8
    q1 = torch.zeros(C * 128, device=dev)
    tab = torch.frombuffer(bytearray(struct.pack("<QQQiiiiii", w1.data_ptr(), q1.data_ptr(), 0, 128, C, 128, C, 0, 0)), dtype=torch.uint8).cuda()
    L.check(lib.yh_pw_pack_multi(tab.data_ptr(), 1, st))
    y1 = torch.empty(M, 128, device=dev)
    pf = torch.empty(max(lib.yh_conv_pw_blocks(M, C, 128), lib.yh_conv_pw_x6_blocks(M, C, 128)) * 2 * 128, device=dev)
    front = {
        "fp32 1x1": lambda: L.check(lib.yh_conv_pw_fwd_act(x.data_ptr(), C, None, 0, q1.data_ptr(), 128, None, y1.data_ptr(), 128, pf.data_ptr(), M, C, 128, st)),
        "x6 1x1": lambda: L.check(lib.yh_conv_pw_fwd_x6(x.data_ptr(), C, None, 0, q1.data_ptr(), 128, None, y1.data_ptr(), 128, pf.data_ptr(), M, C, 128, st)),
        "nothing": lambda: None,
    }
    # the launches behind it: a Winograd 64 -> 64 layer at 80^2 and a BatchNorm + SiLU forward pass over the same tensor
    xw = torch.randn(B, H, W, C, device=dev)
    w3 = torch.randn(C, C, 3, 3, device=dev) / 24
    U = torch.empty(16 * C * C, device=dev)
    L.check(lib.yh_wino_weights(w3.data_ptr(), U.data_ptr(), C, C, C, 0, st))
    yw = torch.empty(B, H, W, C, device=dev)
    pw_ = torch.empty(lib.yh_conv_wino_blocks(B, H, W) * 2 * C, device=dev)
    coef = torch.rand(4 * C, device=dev) + 0.5
    out = torch.empty(B, H, W, C, device=dev)
    behind = {
        "wino 64->64 80^2": lambda: L.check(lib.yh_conv_wino_fwd(xw.data_ptr(), C, U.data_ptr(), C, None, yw.data_ptr(), C, pw_.data_ptr(), B, H, W, C, C, st)),
        "bn_silu_fwd 26 Mpx x 64": lambda: L.check(lib.yh_bn_silu_fwd(yw.data_ptr(), C, coef.data_ptr(), None, 0, out.data_ptr(), C, M, C, H, W, 0, st)),
    }
    # steady state of ONE kind of launch: 40 back-to-back, time per launch (no other kernel in between)
    for fname, ffn in front.items():
        if fname == "nothing":
            continue
        for _ in range(10):
            ffn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40):
            ffn()
        e1.record()
        torch.cuda.synchronize()
        print(f"steady state, 40 back-to-back {fname:9s}: {e0.elapsed_time(e1) / 40 * 1e3:7.1f} us per launch")
    for bname, bfn in behind.items():
        for fname, ffn in front.items():
            ffn(); bfn()
            torch.cuda.synchronize()
            tf = tb = 0.0
            for _ in range(a.iters):
                e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                e[0].record(); ffn(); e[1].record(); bfn(); e[2].record()
                torch.cuda.synchronize()
                tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
            print(f"{bname:26s} after {fname:9s}: front {tf / a.iters * 1e3:7.1f} us, behind {tb / a.iters * 1e3:7.1f} us")
        # a burst: five launches in front
        for fname, ffn in front.items():
            if fname == "nothing":
                continue
            tb = 0.0
            for _ in range(a.iters):
                e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
                for _ in range(5):
                    ffn()
                e[0].record(); bfn(); e[1].record()
                torch.cuda.synchronize()
                tb += e[0].elapsed_time(e[1])
            print(f"{bname:26s} after 5 x {fname:9s}: behind {tb / a.iters * 1e3:7.1f} us")


if __name__ == "__main__":
    main()
