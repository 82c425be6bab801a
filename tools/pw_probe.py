#!/usr/bin/env python3
"""One pointwise (1x1) forward GEMM, timed in isolation with cold or warm inputs (GPU box only; also the target of
rocprofv3 --pmc runs).

    python tools/pw_probe.py --m 102400 --k 128 --n 128 [--nbuf 8] [--iters 40]
nbuf > 1 rotates over that many input / output buffers so that every launch streams its operands from HBM
(8 x (52 + 52) MB exceeds the 256 MB Infinity Cache); nbuf = 1 measures the cache-resident case.
"""
import argparse
import os
import struct
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=102400)
    ap.add_argument("--k", type=int, default=128)
    ap.add_argument("--n", type=int, default=128)
    ap.add_argument("--nbuf", type=int, default=8)
    ap.add_argument("--iters", type=int, default=40)
    a = ap.parse_args()
    from yolo_from_scratch_amd import _lib as L
    lib = L.lib()
    dev = "cuda"
    st = torch.cuda.current_stream().cuda_stream
    M, K, N = a.m, a.k, a.n
    xs = [torch.randn(M, K, device=dev) for _ in range(a.nbuf)]
    ys = [torch.empty(M, N, device=dev) for _ in range(a.nbuf)]
    w = torch.randn(N, K, 1, 1, device=dev) / K ** 0.5
    bias = torch.randn(N, device=dev)
    ldwf, ldwb = (N + 3) // 4 * 4, (K + 3) // 4 * 4
    qf, qb = torch.zeros(K * ldwf, device=dev), torch.zeros((N + 7) // 8 * 8 * ldwb, device=dev)
    tab = torch.frombuffer(bytearray(struct.pack("<QQQiiiiii", w.data_ptr(), qf.data_ptr(), qb.data_ptr(), N, K, ldwf, ldwb, 0, 0)),
                           dtype=torch.uint8).to(dev)
    L.check(lib.yh_pw_pack_multi(tab.data_ptr(), 1, st))
    nb = lib.yh_conv_pw_blocks(M, K, N)
    part = torch.empty(nb * 2 * N, device=dev)

    def run(i):
        x, y = xs[i % a.nbuf], ys[i % a.nbuf]
        L.check(lib.yh_conv_pw_fwd(x.data_ptr(), K, qf.data_ptr(), ldwf, bias.data_ptr(), y.data_ptr(), N, part.data_ptr(), M, K, N, st))
    for i in range(a.nbuf):
        run(i)
    torch.cuda.synchronize()
    ref = xs[0].double() @ w.view(N, K).double().t() + bias.double()
    err = float((ys[0] - ref).abs().max() / ref.abs().max())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(a.iters):
        run(i)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    print(f"M={M} K={K} N={N} nbuf={a.nbuf}: {ms * 1e3:.1f} us  {2.0 * M * K * N / ms / 1e9:.1f} TFLOP/s  {4.0 * M * (K + N) / ms / 1e9:.2f} TB/s  err {err:.1e}"
          f"  [TM={os.environ.get('YH_PW_TM', '-')} NT={os.environ.get('YH_PW_NT', '-')}]", flush=True)


if __name__ == "__main__":
    main()
