"""k3 s1 conv at the bench batch: direct implicit GEMM vs Winograd F(2,3), CUDA-graph timed (no launch gaps).
usage: python scripts/bench_wino.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepards_amd import hip_ops as H
ROWS = int(os.environ.get('ROWS', 1280))

def graph_time(fn, reps=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

for ci, co, L in ((64, 64, 56), (128, 128, 28), (256, 256, 14), (512, 512, 7), (128, 32, 56)):
    x = torch.randn(ROWS, L, ci, device='cuda'); w = torch.randn(co, ci, 3, device='cuda') * 0.05
    wf, wd = H.repack_weight(w, True, True)
    u = H.wino_weights(w)
    y = H.conv_fwd(x, wf, 1, 1); y2 = torch.empty_like(y)
    H.conv3_winograd(x, u, out=y2)
    err = (y - y2).abs().max().item() / y.abs().max().item()
    td = graph_time(lambda: H.conv_fwd(x, wf, 1, 1, out=y))
    tw = graph_time(lambda: H.conv3_winograd(x, u, out=y2))
    u6 = H.wino_weights(w, points=6)
    y3 = torch.empty_like(y)
    H.conv3_winograd(x, u6, out=y3)
    err4 = (y - y3).abs().max().item() / y.abs().max().item()
    t4 = graph_time(lambda: H.conv3_winograd(x, u6, out=y3))
    fl = 2.0 * ROWS * L * ci * co * 3
    print('   F(4,3) %7.1f us %6.1f TF(alg)  x%.2f vs F(2,3)  maxdiff %.1e' % (t4, fl / t4 / 1e6, tw / t4, err4))
    print('%4d->%4d L %2d  direct %7.1f us %6.1f TF | winograd %7.1f us %6.1f TF(alg)  x%.2f  maxdiff %.1e' % (ci, co, L, td, fl / td / 1e6, tw, fl / tw / 1e6, td / tw, err))
