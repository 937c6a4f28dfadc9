"""The stride-2 block entry: conv_x3p_s2 (x3 operands, one launch per direction) against the fp32 pair launches it
replaces (conv_gemm_multi_kernel), hipGraph timed at the bench batch.  usage: python scripts/bench_x3p_s2.py"""
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
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


tot = [0.0, 0.0]
for ci, co, lin in ((64, 128, 56), (128, 256, 28), (256, 512, 14)):
    torch.manual_seed(0)
    x = torch.randn(ROWS, lin, ci, device='cuda')
    w1 = torch.randn(co, ci, 3, device='cuda') * 0.05; wd = torch.randn(co, ci, 1, device='cuda') * 0.05
    (wf1, wd1, _, _), (wfd, wdd, _, _) = H.repack_multi([w1, wd])
    (_, _, uf1, ud1), (_, _, ufd, udd) = H.repack_multi([w1, wd], [49, 49])
    x3 = H.x3_split(x)
    y1, yd = H.conv_x3p_s2_fwd(x3, uf1, ufd)
    dy1, dyd = torch.randn_like(y1), torch.randn_like(yd)
    dy13, dyd3 = H.x3_split(dy1), H.x3_split(dyd)
    dx = H.conv_x3p_s2_dgrad(dy13, ud1, dyd3, udd)
    t0 = graph_time(lambda: H.conv_fwd_multi([(x, wf1, 2, 1), (x, wfd, 2, 0)]))
    t1 = graph_time(lambda: H.conv_x3p_s2_fwd(x3, uf1, ufd))
    t2 = graph_time(lambda: H.conv_dgrad_s2_pair(dy1, wd1, dyd, wdd, lin))
    t3 = graph_time(lambda: H.conv_x3p_s2_dgrad(dy13, ud1, dyd3, udd, out=dx))
    fl = 2.0 * ROWS * (lin // 2) * ci * co * 4
    print('%3d->%3d Lin %2d | fwd pair fp32 %6.1f us  x3p %6.1f us (%5.1f TF alg)  | dgrad pair fp32 %6.1f us  x3p %6.1f us (%5.1f TF alg)' %
          (ci, co, lin, t0, t1, fl / t1 / 1e6, t2, t3, fl / t3 / 1e6), flush=True)
    tot[0] += t0 + t2; tot[1] += t1 + t3
print('all three stages, both directions: fp32 %.1f us, x3p %.1f us' % tuple(tot))
