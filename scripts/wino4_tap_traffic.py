"""does the F(4,3) kernel's 2.5x fabric traffic (each XCD's L2 pulls its own copy of the 6.3 MB of
transformed taps) cost time?  Same launch geometry with the taps read modulo 128 output channels (1.6 MB: fits a 4 MB L2;
results wrong by design, timing only), interleaved with the real kernel in one process, 20 launches per captured graph.
usage: python scripts/wino4_tap_traffic.py > profiles/r04_wino4_tap_traffic.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepards_amd import hip_ops as H, _lib

torch.manual_seed(0)
rows, l, c = 1280, 7, 512
x = torch.randn(rows, l, c, device='cuda')
u = H.wino_weights(torch.randn(c, c, 3, device='cuda') * 0.02, points=6)
out = torch.empty(rows, l, c, device='cuda')


def graph_time(n=20, reps=30):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        H.conv3_winograd(x, u, out=out)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                H.conv3_winograd(x, u, out=out)
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps / n * 1e6


L = _lib.lib()
print('conv3_wino4k_kernel, 512 -> 512 channels, B = 64 (1280 rows x 7): us per launch, 20 launches per graph, 30 replays')
for rep in range(4):
    res = []
    for mod in (0, 128, 0, 128):
        L.da_wino_debug_tapmod(mod)
        res.append((mod, graph_time()))
    print('  pass %d: ' % rep + '   '.join('%s %.2f' % ('taps mod 128' if m else 'real taps   ', t) for m, t in res))
L.da_wino_debug_tapmod(0)
