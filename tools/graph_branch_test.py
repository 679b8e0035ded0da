"""Does a hipGraph captured from two streams run its branches concurrently? (tuning probe)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import drvae_amd.kernels as K
from drvae_amd import _lib
_lib.load()
dev = torch.device('cuda:0')
x = [torch.randn(1 << 14, device=dev) for _ in range(4)]
y = [torch.zeros(1 << 14, device=dev) for _ in range(4)]
A = torch.randn(600, 600, device=dev); B = torch.randn(2000, 600, device=dev); Cm = torch.empty(600, 2000, device=dev)
A2 = torch.randn(448, 200, device=dev); B2 = torch.randn(200, 200, device=dev); C2 = torch.empty(448, 200, device=dev)

def chain(i, n, kind):
    for _ in range(n):
        if kind == 'small':
            K.axpby(y[i], x[i], 1.0, 0.5)
        elif kind == 'sgemm':
            K.gemm(C2, A2, B2, True, True)
        else:
            K.gemm(Cm, A, B, True, True)

def run(build):
    g = torch.cuda.CUDAGraph()
    build()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        build()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3)
    return best

side = torch.cuda.Stream()
def two(n, ka, kb):
    def b():
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            chain(1, n, kb)
        chain(0, n, ka)
        torch.cuda.current_stream().wait_stream(side)
    return b
def serial(n, ka, kb):
    def b():
        chain(1, n, kb); chain(0, n, ka)
    return b
for ka, kb in [('small', 'small'), ('big', 'small'), ('big', 'sgemm'), ('sgemm', 'sgemm'), ('big', 'big')]:
    print('%-6s|%-6s x20 each: serial %8.1f us   two-branch %8.1f us' % (ka, kb, run(serial(20, ka, kb)), run(two(20, ka, kb))))
