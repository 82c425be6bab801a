#!/usr/bin/env python3
"""Run ONE bf16 forward convolution (bf16_gemm_kernel) or weight gradient (bf16_wgrad_kernel) repeatedly, for rocprofv3 --pmc / --kernel-trace runs on the GPU box.

    python tools/one_conv_bf16.py --shape 64,80,80,64,64,3,1 --iters 20
"""
import argparse
import os
import struct
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="64,80,80,64,64,3,1")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--what", default="fwd", choices=["fwd", "wgrad"])
    a = ap.parse_args()
    from yolo_from_scratch_amd import _lib as L
    lib = L.lib()
    B, H, W, Cin, Cout, k, s = [int(v) for v in a.shape.split(",")]
    p = k // 2
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    dev = "cuda"
    st = torch.cuda.current_stream().cuda_stream
    x = torch.randn(B, H, W, Cin, device=dev).to(torch.bfloat16)
    w = torch.randn(Cout, Cin, k, k, device=dev) / (Cin * k * k) ** 0.5
    ldf, ldb = (Cout + 7) // 8 * 8, Cin
    wf = torch.empty(k * k * Cin * ldf, dtype=torch.bfloat16, device=dev)
    rec = struct.pack("<QQQiiiiiiii", w.data_ptr(), wf.data_ptr(), 0, Cout, Cin, k * k, Cin, ldf, ldb, 0, ldf)
    tab = torch.frombuffer(bytearray(rec), dtype=torch.uint8).cuda()
    L.check(lib.yh_bf16_pack_multi(tab.data_ptr(), 1, st), "pack")
    y = torch.empty(B, Ho, Wo, Cout, dtype=torch.bfloat16, device=dev)
    part = torch.empty(max(lib.yh_bf16_conv_blocks(B * Ho * Wo), 512) * 2 * Cout, device=dev)

    dy = torch.randn(B, Ho, Wo, ldf, device=dev).to(torch.bfloat16)
    nws = int(lib.yh_bf16_conv_bwd_weight_ws(B, H, W, Cin, Cout, k, s))
    ws = torch.empty(max(nws, 1), device=dev)
    dw = torch.empty(Cout, Cin, k, k, device=dev)

    def run():
        if a.what == "wgrad":
            L.check(lib.yh_bf16_conv_bwd_weight(x.data_ptr(), Cin, dy.data_ptr(), ldf, dw.data_ptr(), ws.data_ptr(), nws, B, H, W, Cin,
                                                Cin, Cout, k, s, st), "wgrad")
            return
        L.check(lib.yh_bf16_conv_fwd(x.data_ptr(), Cin, wf.data_ptr(), ldf, None, y.data_ptr(), Cout, 0, part.data_ptr(), B, H, W, Cin,
                                     Cout, k, s, st), "fwd")
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
    gb = (x.numel() + y.numel()) * 2 / 1e9
    print(f"bf16 {a.what} {a.shape}: {ms * 1e3:.1f} us  {gf / ms:.1f} TFLOP/s  {gb / ms * 1e3:.0f} GB/s of activations")


if __name__ == "__main__":
    main()
