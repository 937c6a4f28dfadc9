import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepards_amd import hip_ops as H
def graph_time(fn, reps=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for ci, L, rowsl in ((512, 7, (1024, 1152, 1280, 1536, 2048)), (256, 14, (1170, 1280, 1463, 2340)), (64, 56, (1170, 1280, 2340))):
    for rows in rowsl:
        x = torch.randn(rows, L, ci, device='cuda'); w = torch.randn(ci, ci, 3, device='cuda') * 0.05
        u = H.wino_weights(w); y = torch.empty_like(x)
        t = graph_time(lambda: H.conv3_winograd(x, u, out=y))
        tiles = ((rows * ((L + 1) // 2) + 63) // 64) * (ci // 32)
        print('C %3d L %2d rows %4d tiles %5d (%.3f/CU) %7.1f us  %.2f us per tile/CU' % (ci, L, rows, tiles, tiles / 256, t, t / (tiles / 256)))
