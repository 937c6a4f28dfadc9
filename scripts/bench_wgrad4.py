"""The 512-channel k3 s1 weight gradients at the bench batch (3 jobs, as one step issues them): Winograd F(4,3) form against
the F(2,3) form, hipGraph timed.  usage: python scripts/bench_wgrad4.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepards_amd import hip_ops as H
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rows, L, C = B * 20, 7, 512
g = torch.Generator().manual_seed(0)
jobs = [(torch.randn(rows, L, C, generator=g).cuda(), torch.randn(rows, L, C, generator=g).cuda(), 3, 1, 1) for _ in range(3)]
def timed(minc):
    H.WINO4_WGRAD_MIN_C = minc
    for _ in range(3):
        H.conv_wgrad_multi(jobs)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(10):
            H.conv_wgrad_multi(jobs)
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 50 * 1e3
flops = 3 * 2.0 * rows * L * C * C * 3
for name, minc in (('F(2,3)', 1 << 30), ('F(4,3)', 512), ('F(2,3)', 1 << 30), ('F(4,3)', 512)):
    us = timed(minc)
    print('%s: %.1f us for the three 512-channel jobs = %.1f TF/s algorithmic' % (name, us, flops / us / 1e6))
