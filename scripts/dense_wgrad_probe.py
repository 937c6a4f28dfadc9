"""A/B of the dense-block 1x1 weight gradients at the bench shape (B = 64): X recomputed while staged (xform) against the
same jobs on a materialised relu(norm(x)) tensor.  usage: python scripts/dense_wgrad_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepards_amd import hip_ops as H

torch.manual_seed(0)
rows, R = 1280, 20
jobs_x, jobs_h, jobs_xc, jobs_hp = [], [], [], []
for l in (56, 28, 14, 7):
    buf = torch.randn(rows, l, 128, device='cuda')
    st = torch.empty(2, rows // R, 128, device='cuda')
    H.bn_stats_fused(buf, R, st[0], st[1])
    for ck in (64, 96):
        g, b = torch.rand(ck, device='cuda') + 0.5, torch.randn(ck, device='cuda') * 0.1
        dy = torch.randn(rows, l, 128, device='cuda')
        xv = buf[:, :, :ck]
        jobs_x.append((dy, xv, 1, 1, 0, {'xform': (st[0][:, :ck], st[1][:, :ck], g, b, R)}))
        h = H.bn_relu_ss(xv, R, st[0][:, :ck], st[1][:, :ck], g, b)
        jobs_h.append((dy, h, 1, 1, 0))
        jobs_xc.append((dy, xv.contiguous(), 1, 1, 0, {'xform': (st[0][:, :ck], st[1][:, :ck], g, b, R)}))      # xform, contiguous x
        hp = torch.empty_like(buf)
        hp[:, :, :ck] = h
        jobs_hp.append((dy, hp[:, :, :ck], 1, 1, 0))                                                             # plain, pitched h


def timeit(jobs, n=30):
    for _ in range(3):
        H.conv_wgrad_multi(jobs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        H.conv_wgrad_multi(jobs)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for rep in range(3):
    print('xform %.1f us   materialised %.1f us' % (timeit(jobs_x), timeit(jobs_h)))
for i in range(len(jobs_x)):
    print(tuple(jobs_x[i][1].shape), 'xform %.1f us   materialised %.1f us   xform on contiguous x %.1f us   plain on pitched h %.1f us' %
          (timeit([jobs_x[i]]), timeit([jobs_h[i]]), timeit([jobs_xc[i]]), timeit([jobs_hp[i]])))
