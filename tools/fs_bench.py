#!/usr/bin/env python3
"""Forced flat-stream forward kernel vs whatever yh_bf16_conv_fwd dispatches, per shape (GPU box).
    python tools/fs_bench.py [B,H,W,K,N,k ...]"""
import os, struct, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
DEFAULT = ["64,40,40,64,64,3", "64,80,80,64,64,3", "64,80,80,32,32,3", "64,80,80,64,64,1", "64,80,80,128,32,1", "64,40,40,128,128,1",
           "64,40,40,128,64,1", "64,80,80,64,32,1", "64,20,20,128,128,1"]


def main():
    from yolo_from_scratch_amd import _lib as L
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    iters = int(os.environ.get("ITERS", "30"))
    for spec in (sys.argv[1:] or DEFAULT):
        B, H, W, K, N, k = (int(v) for v in spec.split(","))
        x = torch.randn(B, H, W, K, device="cuda").to(torch.bfloat16)
        w = torch.randn(N, K, k, k, device="cuda") / (K * k * k) ** 0.5
        wf = torch.empty(k * k * K * N, dtype=torch.bfloat16, device="cuda")
        rec = struct.pack("<QQQiiiiiiii", w.data_ptr(), wf.data_ptr(), 0, N, K, k * k, K, N, K, 0, N)
        tab = torch.frombuffer(bytearray(rec), dtype=torch.uint8).cuda()
        L.check(lib.yh_bf16_pack_multi(tab.data_ptr(), 1, st), "pack")
        y = torch.empty(B, H, W, N, dtype=torch.bfloat16, device="cuda")
        part = torch.zeros(max(lib.yh_bf16_conv_stream_blocks(B, H, W, K, N, k), lib.yh_bf16_conv_blocks(B * H * W)) * 2 * N, device="cuda")

        def forced():
            L.check(lib.yh_bf16_conv_stream_fwd(x.data_ptr(), K, wf.data_ptr(), N, None, y.data_ptr(), N, part.data_ptr(), B, H, W, K, N, k, st), "s")

        def auto():
            L.check(lib.yh_bf16_conv_fwd(x.data_ptr(), K, wf.data_ptr(), N, None, y.data_ptr(), N, 0, part.data_ptr(), B, H, W, K, N, k, 1, st), "a")
        res = []
        for fn in (forced, auto):
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / iters * 1e3)
        mb = (x.numel() + y.numel()) * 2 / 1e6
        print(f"{spec:24s} stream {res[0]:7.1f} us = {mb / res[0]:5.2f} TB/s   dispatched {res[1]:7.1f} us = {mb / res[1]:5.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
