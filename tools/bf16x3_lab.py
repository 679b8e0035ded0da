#!/usr/bin/env python3
"""LAB ONLY (round-5 review, item 10; never the product path, never the headline): what lies past 0.95 of the fp32 matrix pipe
for cfg 5's big products -- split-bf16 emulation.  Each fp32 operand is split into three bf16 terms (hi + mid + lo: 3 x 8
mantissa bits), the six cross products of weight <= 2 (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid) run on the bf16 MFMA with
fp32 accumulation.  Here as ONE vendor-library bf16 GEMM over K' = 6 K (the terms concatenated along K), to answer two
questions before anyone writes the kernel: how accurate is it against the fp32 fmaf chain, and what does the bf16 pipe give back
after paying 6x the MACs?  dtype of this experiment: bf16 x 3 operands, fp32 accumulate."""
import sys, time, torch
dev = 'cuda'
torch.manual_seed(0)


def split3(x):
    hi = x.to(torch.bfloat16)
    r1 = x - hi.float()
    mid = r1.to(torch.bfloat16)
    lo = (r1 - mid.float()).to(torch.bfloat16)
    return hi, mid, lo


def cat6(a, b):
    ah, am, al = split3(a)
    bh, bm, bl = split3(b)
    return torch.cat([ah, ah, am, ah, al, am], 1).contiguous(), torch.cat([bh, bm, bh, bl, bh, bm], 1).contiguous()


def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(n):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


for (M, N, K) in ((2048, 2048, 2048), (8192, 8192, 2048), (8192, 40000, 2048)):
    a = torch.randn(M, K, device=dev)
    b = torch.randn(N, K, device=dev) / K ** 0.5
    ref32 = a @ b.t()                                       # the vendor fp32 product (fp32 MFMA, exact fp32 accumulation)
    a6, b6 = cat6(a, b)
    try:
        out = torch.mm(a6, b6.t(), out_dtype=torch.float32)
        mm = lambda: torch.mm(a6, b6.t(), out_dtype=torch.float32)
        how = 'bf16 GEMM with fp32 output'
    except TypeError:
        out = (a6 @ b6.t()).float()
        mm = lambda: a6 @ b6.t()
        how = 'bf16 GEMM with bf16 OUTPUT (this torch has no out_dtype: the error below is the output rounding, not the split)'
    # float64 reference on a sample of rows
    rows = torch.arange(0, M, max(1, M // 64), device=dev)
    ref64 = a[rows].double() @ b.double().t()
    e_emul = float((out[rows].double() - ref64).norm() / ref64.norm())
    e_fp32 = float((ref32[rows].double() - ref64).norm() / ref64.norm())
    t_split = timed(lambda: cat6(a, b))
    t_mm = timed(mm)
    t_fp32 = timed(lambda: a @ b.t())
    fl = 2.0 * M * N * K
    print('%5d x %5d x %4d: fp32 product %.3f ms = %.1f TF/s (norm-wise error vs float64 %.2e) | split pass %.3f ms + %s over K'
          ' = 6K %.3f ms => %.1f TF/s-equivalent incl. the split (%.1f without), error %.2e'
          % (M, N, K, t_fp32, fl / t_fp32 / 1e9, e_fp32, t_split, how, t_mm, fl / (t_mm + t_split) / 1e9, fl / t_mm / 1e9, e_emul), flush=True)
