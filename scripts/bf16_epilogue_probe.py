"""conv3_bf16_kernel at the bench batch with bf16 storage: plain / accumulating launches, per stage (hipGraph of 20 launches).
DA_LIB_PATH=<a build with -DCB_NO_STORE> shows what the epilogue's stores (and the accumulate loads) cost: timing only.
usage: python scripts/bf16_epilogue_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepards_amd import hip_ops as H
H.set_act_dtype('bf16')

def graph_time(fn, reps=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best

for c, L in ((64, 56), (128, 28), (256, 14), (512, 7)):
    x = torch.randn(1280, L, c, device='cuda').bfloat16()
    w = torch.randn(c, c, 3, device='cuda') * (2.0 / (3 * c)) ** 0.5
    wf, wd = H.pack_conv3_bf16(w)
    y = torch.zeros_like(x)
    t0 = graph_time(lambda: H.conv3_bf16(x, wf, out=y))
    t1 = graph_time(lambda: H.conv3_bf16(x, wf, out=y, accumulate=True))
    print('%4d ch L %2d: plain %5.1f us  accumulate %5.1f us' % (c, L, t0, t1))
