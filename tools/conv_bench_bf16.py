#!/usr/bin/env python3
"""Time yh_bf16_conv_fwd / yh_bf16_conv_bwd_data per shape through the C ABI (GPU box only).
    python tools/conv_bench_bf16.py [B,H,W,Cin,Cout,k,s ...]"""
import os
import struct
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

DEFAULT = ["64,40,40,64,64,3,1", "64,80,80,32,32,3,1", "64,80,80,64,64,3,1", "64,160,160,16,16,3,1", "64,160,160,32,16,1,1",
           "64,160,160,32,32,1,1", "64,80,80,64,64,1,1", "64,80,80,128,32,1,1", "64,40,40,128,128,1,1", "64,40,40,128,64,1,1",
           "64,40,40,128,128,3,1"]


def main():
    from yolo_from_scratch_amd import _lib as L
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    iters = int(os.environ.get("ITERS", "30"))
    for spec in (sys.argv[1:] or DEFAULT):
        B, H, W, Cin, Cout, k, s = (int(v) for v in spec.split(","))
        x = torch.randn(B, H, W, Cin, device="cuda").to(torch.bfloat16)
        w = torch.randn(Cout, Cin, k, k, device="cuda") / (Cin * k * k) ** 0.5
        ldf, ldb = Cout, Cin
        wf = torch.empty(k * k * Cin * ldf, dtype=torch.bfloat16, device="cuda")
        wb = torch.zeros(k * k * Cout * ldb, dtype=torch.bfloat16, device="cuda")
        rec = struct.pack("<QQQiiiiiiii", w.data_ptr(), wf.data_ptr(), wb.data_ptr(), Cout, Cin, k * k, Cin, ldf, ldb, 0, Cout)
        tab = torch.frombuffer(bytearray(rec), dtype=torch.uint8).cuda()
        L.check(lib.yh_bf16_pack_multi(tab.data_ptr(), 1, st), "pack")
        y = torch.empty(B, H, W, Cout, dtype=torch.bfloat16, device="cuda")
        dx = torch.empty(B, H, W, Cin, dtype=torch.bfloat16, device="cuda")
        nblk = lib.yh_bf16_conv_fwd_blocks(B, H, W, Cin, Cout, k, s, 0, Cin, Cout)
        part = torch.zeros(nblk * 2 * Cout, device="cuda")

        def fwd():
            L.check(lib.yh_bf16_conv_fwd(x.data_ptr(), Cin, wf.data_ptr(), ldf, None, y.data_ptr(), Cout, 0, part.data_ptr(), B, H, W, Cin,
                                         Cout, k, s, st), "fwd")

        def bwd():
            L.check(lib.yh_bf16_conv_bwd_data(y.data_ptr(), Cout, None, 0, wb.data_ptr(), ldb, dx.data_ptr(), Cin, B, H, W, Cin, Cout, k, s,
                                              0, st), "bwd_data")
        res = []
        for fn in (fwd, bwd):
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
        print(f"{spec:28s} fwd {res[0]:7.1f} us = {mb / res[0]:5.2f} TB/s   dgrad {res[1]:7.1f} us = {mb / res[1]:5.2f} TB/s   ({mb:.0f} MB, {nblk} BN rows)",
              flush=True)


if __name__ == "__main__":
    main()
