#!/usr/bin/env python3
"""Diagnostic: bf16 narrow-layer kernels vs the generic bf16 kernels on identical operands -- how many stored bf16 values differ
(they should differ only where the fp32 sum sits on a rounding boundary)."""
import os
import struct
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from yolo_from_scratch_amd import _lib as L
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    for (B, H, W, Cin, Cout, s) in [(1, 640, 640, 3, 16, 2), (1, 320, 320, 16, 32, 2), (1, 160, 160, 16, 16, 1)]:
        torch.manual_seed(1)
        cin = (Cin + 7) // 8 * 8
        cin_k = 4 if Cin <= 4 else Cin
        x = torch.zeros(B, H, W, cin, dtype=torch.bfloat16, device="cuda")
        x[..., :Cin] = torch.rand(B, H, W, Cin, device="cuda").to(torch.bfloat16)
        w = (torch.randn(Cout, Cin, 3, 3, device="cuda") / (Cin * 9) ** 0.5).contiguous()
        bias = torch.randn(Cout, device="cuda")
        ldf, ldb = (Cout + 7) // 8 * 8, cin
        wf = torch.empty(9 * cin * ldf, dtype=torch.bfloat16, device="cuda")
        wb = torch.zeros(9 * ldf * ldb, dtype=torch.bfloat16, device="cuda")
        rec = struct.pack("<QQQiiiiiiii", w.data_ptr(), wf.data_ptr(), wb.data_ptr(), Cout, Cin, 9, cin, ldf, ldb, 0, ldf)
        tab = torch.frombuffer(bytearray(rec), dtype=torch.uint8).cuda()
        L.check(lib.yh_bf16_pack_multi(tab.data_ptr(), 1, st))
        Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
        y1 = torch.empty(B, Ho, Wo, Cout, dtype=torch.bfloat16, device="cuda")
        y2 = torch.empty_like(y1)
        n1 = lib.yh_conv_narrow_blocks(B, H, W, cin_k, s)
        n2 = max(lib.yh_bf16_conv_blocks(B * Ho * Wo), 512)
        p1, p2 = torch.zeros(n1 * 2 * Cout, device="cuda"), torch.zeros(n2 * 2 * Cout, device="cuda")
        L.check(lib.yh_bf16_conv_narrow(x.data_ptr(), cin, wf.data_ptr(), ldf, cin, bias.data_ptr(), y1.data_ptr(), Cout, p1.data_ptr(),
                                        B, H, W, cin_k, Cout, s, 0, 0, st))
        L.check(lib.yh_bf16_conv_fwd(x.data_ptr(), cin, wf.data_ptr(), ldf, bias.data_ptr(), y2.data_ptr(), Cout, 0, p2.data_ptr(), B, H, W,
                                     cin, Cout, 3, s, st))
        torch.cuda.synchronize()
        d = (y1.float() - y2.float()).abs()
        s1, s2 = p1.view(n1, 2, Cout).double().sum(0), p2.view(n2, 2, Cout).double().sum(0)
        print(f"{(B, H, W, Cin, Cout, s)}: differing values {int((d > 0).sum())} of {d.numel()}, max |diff| {float(d.max()):.3e} "
              f"(max |y| {float(y2.float().abs().max()):.2f}); partial sums rel diff {float(((s1 - s2).abs() / (s2.abs() + 1e-9)).max()):.2e}", flush=True)


if __name__ == "__main__":
    main()
