#!/usr/bin/env python3
"""Time yh_bf16_conv_bwd_weight (main kernel + slab reduction) per shape through the C ABI (GPU box only).

    python tools/wgrad_bench.py [B,H,W,Cin,Cout,k,s ...]        default: the stride-1 shapes of the nc=80 model at batch 64
Under rocprofv3 --kernel-trace --stats one shape per run separates the stream kernel from the reduction."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

DEFAULT = ["64,40,40,64,64,3,1", "64,80,80,32,32,3,1", "64,80,80,64,64,3,1", "64,40,40,128,128,3,1", "64,20,20,256,256,3,1",
           "64,20,20,128,128,3,1", "64,160,160,32,16,1,1", "64,40,40,128,128,1,1", "64,80,80,64,64,1,1", "64,80,80,64,255,1,1",
           "64,20,20,512,256,1,1"]


def main():
    from yolo_from_scratch_amd import _lib as L
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    iters = int(os.environ.get("ITERS", "30"))
    for spec in (sys.argv[1:] or DEFAULT):
        B, H, W, Cin, Cout, k, s = (int(v) for v in spec.split(","))
        p = k // 2
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        kp = (Cout + 7) // 8 * 8
        x = torch.randn(B, H, W, Cin, device="cuda").to(torch.bfloat16)
        dy = torch.randn(B, Ho, Wo, kp, device="cuda").to(torch.bfloat16)
        if kp != Cout:
            dy[..., Cout:] = 0
        nws = int(lib.yh_bf16_conv_bwd_weight_ws(B, H, W, Cin, Cout, k, s))
        ws = torch.empty(nws, device="cuda")
        dw = torch.empty(Cout, Cin, k, k, device="cuda")

        def run():
            L.check(lib.yh_bf16_conv_bwd_weight(x.data_ptr(), Cin, dy.data_ptr(), kp, dw.data_ptr(), ws.data_ptr(), nws, B, H, W, Cin, Cin,
                                                Cout, k, s, st), "bwd_weight")
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / iters * 1e3
        mb = (x.numel() + dy.numel()) * 2 / 1e6
        print(f"{spec:28s} {us:8.1f} us   activations {mb:6.1f} MB = {mb / us:5.2f} TB/s   slabs {nws * 4 / 1e6:6.1f} MB", flush=True)


if __name__ == "__main__":
    main()
