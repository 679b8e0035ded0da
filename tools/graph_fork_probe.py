"""How does the ROCm hipGraph executor run a fork/join captured from several streams?  (tuning probe)
prefix -> fork -> [main: 6 small GEMMs | side: 14 tiny kernels] -> join -> suffix, several capture orders."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import drvae_amd.kernels as K
from drvae_amd import _lib
_lib.load()
dev = torch.device('cuda:0')
x = [torch.randn(1 << 14, device=dev) for _ in range(4)]
y = [torch.zeros(1 << 14, device=dev) for _ in range(4)]
A2 = torch.randn(448, 200, device=dev); B2 = torch.randn(200, 200, device=dev); C2 = [torch.empty(448, 200, device=dev) for _ in range(3)]
NM, NS = int(os.environ.get('NM', 6)), int(os.environ.get('NS', 14))

def small(i, n):
    for _ in range(n):
        K.axpby(y[i], x[i], 1.0, 0.5)
def sg(i, n):
    for _ in range(n):
        K.gemm(C2[i], A2, B2, True, True)

s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
cur = torch.cuda.current_stream

def serial():
    small(0, 3); sg(0, NM); small(1, NS); small(0, 1)
def side_first():          # what the engine records today
    small(0, 3)
    s1.wait_stream(cur())
    with torch.cuda.stream(s1):
        small(1, NS)
    sg(0, NM)
    cur().wait_stream(s1)
    small(0, 1)
def main_first():
    small(0, 3)
    s1.wait_stream(cur())
    sg(0, NM)
    with torch.cuda.stream(s1):
        small(1, NS)
    cur().wait_stream(s1)
    small(0, 1)
def both_fresh():          # origin stream idles between fork and join
    small(0, 3)
    s1.wait_stream(cur()); s2.wait_stream(cur())
    with torch.cuda.stream(s1):
        small(1, NS)
    with torch.cuda.stream(s2):
        sg(0, NM)
    cur().wait_stream(s1); cur().wait_stream(s2)
    small(0, 1)
def both_fresh_main_first():
    small(0, 3)
    s1.wait_stream(cur()); s2.wait_stream(cur())
    with torch.cuda.stream(s2):
        sg(0, NM)
    with torch.cuda.stream(s1):
        small(1, NS)
    cur().wait_stream(s2); cur().wait_stream(s1)
    small(0, 1)
def only_main():
    small(0, 3); sg(0, NM); small(0, 1)
def only_side():
    small(0, 3); small(1, NS); small(0, 1)

def run(build):
    g = torch.cuda.CUDAGraph()
    build(); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        build()
    best = 1e9
    for _ in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3)
    return best

for f in (only_main, only_side, serial, side_first, main_first, both_fresh, both_fresh_main_first):
    print('%-24s %8.1f us' % (f.__name__, run(f)))
