"""BatchNorm kernels against plain streaming kernels (torch copy / add) on tensors of the bench shapes, hipGraph timed:
how far the per-window kernels are from the byte path.  usage: python scripts/bw_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepards_amd import hip_ops as H
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
rows, R = 1280, 20
for L, c in ((56, 64), (28, 128), (14, 256), (7, 512)):
    x = torch.randn(rows, L, c, device='cuda'); y = torch.empty_like(x); r = torch.randn_like(x); z = torch.empty_like(x)
    mb = x.numel() * 4 / 1e6
    tc = graph_time(lambda: y.copy_(x))
    ta = graph_time(lambda: torch.add(x, r, out=z))
    gamma, beta = torch.ones(c, device='cuda'), torch.zeros(c, device='cuda')
    t0 = graph_time(lambda: H.bn_fwd(x, R, gamma, beta, relu=True))
    t1 = graph_time(lambda: H.bn_fwd(x, R, gamma, beta, relu=True, res=r, want_mask=True))
    _, mean, invstd, mask = H.bn_fwd(x, R, gamma, beta, relu=True, res=r, want_mask=True)
    d = torch.randn_like(x)
    t2 = graph_time(lambda: H.bn_bwd(d, x, R, mean, invstd, gamma, beta, 1, want_g=False, defer_param_grads=True))
    print('L %2d C %3d (%.1f MB)  copy %.1f us (%.2f TB/s)  add3 %.1f us (%.2f TB/s) | bn_fwd %.1f us (%.2f TB/s)  bn_fwd+res+mask %.1f us (%.2f TB/s)  bn_bwd %.1f us (%.2f TB/s)' %
          (L, c, mb, tc, 2 * mb / tc, ta, 3 * mb / ta, t0, 2 * mb / t0, t1, 3 * mb / t1, t2, 3 * mb / t2), flush=True)
