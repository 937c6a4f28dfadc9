"""conv3_x3p at shapes that isolate one tile kind (A/B of the DA_X3_KERNEL variants): usage python scripts/x3p_shape_probe.py"""
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


# (rows, L, C, what): M = rows * L positions, 256 x 64 tiles = ceil(M / 256) * C / 64
for rows, L, c, what in ((1024, 8, 512, '256 full tiles, one round'), (2048, 8, 512, '512 full tiles, two rounds'),
                         (256, 8, 512, '256 quarter tiles'), (1280, 7, 512, 'bench: 96 quarter + 256 full'),
                         (1024, 16, 256, '256 full tiles, one round'), (1280, 14, 256, 'bench'),
                         (1280, 28, 128, 'bench'), (1280, 56, 64, 'bench')):
    torch.manual_seed(0)
    x = torch.randn(rows, L, c, device='cuda'); w = torch.randn(c, c, 3, device='cuda') * (2.0 / (3 * c)) ** 0.5
    uf = H.repack_multi([w], [49])[0][2]
    x3 = H.x3_split(x)
    y = H.conv3_x3p(x3, uf)
    t = graph_time(lambda: H.conv3_x3p(x3, uf, out=y))
    fl = 2.0 * rows * L * c * c * 3
    print('rows %5d L %2d C %3d  %-32s %7.1f us  %6.1f TF(alg)' % (rows, L, c, what, t, fl / t / 1e6), flush=True)
