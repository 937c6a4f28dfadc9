"""All weight-gradient GEMMs of one resnet18 step at the bench batch in one da_conv_wgrad_multi call: the fp32 kernels
(Winograd F(2,3) form for k3 s1, direct for the stride-2 ones) vs the split-bf16 ("f32x3") kernels vs bf16 operands;
error of each against an fp64 reference.   usage: python scripts/bench_wgrad_x3.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepards_amd import hip_ops as H
ROWS = int(os.environ.get('ROWS', 1280))


def graph_time(fn, reps=5):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


# (ci, co, L_in, k, stride, pad, count) of resnet18's convs behind the stem at L = 224
SHAPES = [(64, 64, 56, 3, 1, 1, 4), (64, 128, 56, 3, 2, 1, 1), (64, 128, 56, 1, 2, 0, 1), (128, 128, 28, 3, 1, 1, 3),
          (128, 256, 28, 3, 2, 1, 1), (128, 256, 28, 1, 2, 0, 1), (256, 256, 14, 3, 1, 1, 3),
          (256, 512, 14, 3, 2, 1, 1), (256, 512, 14, 1, 2, 0, 1), (512, 512, 7, 3, 1, 1, 3)]
torch.manual_seed(0)
jobs, refs = [], []
for ci, co, l, k, st, pd, cnt in SHAPES:
    lo = (l + 2 * pd - k) // st + 1
    for _ in range(cnt):
        x = torch.randn(ROWS, l, ci, device='cuda')
        dy = torch.randn(ROWS, lo, co, device='cuda') * 1e-4
        jobs.append((dy, x, k, st, pd))
for (dy, x, k, st, pd) in jobs[::3]:
    xd, dyd = x.double().transpose(1, 2), dy.double().transpose(1, 2)
    w = torch.zeros(dy.shape[2], x.shape[2], k, device='cuda', dtype=torch.float64, requires_grad=True)
    y = torch.nn.functional.conv1d(xd, w, stride=st, padding=pd)
    (gw,) = torch.autograd.grad(y, w, dyd)
    refs.append(gw)


def run(mode):
    H.WGRAD_BF16, H.WGRAD_X3 = mode == 'bf16', mode == 'x3'
    slabs = H.conv_wgrad_multi(jobs)
    dws = [torch.zeros(s[3], s[4], s[2], device='cuda') for s in slabs]
    H.wgrad_reduce_multi(list(zip(slabs, dws)), accumulate=False)
    return dws


for mode in ('f32', 'x3', 'bf16'):
    dws = run(mode)
    errs = [((d.double() - r).abs().max() / r.abs().max()).item() for d, r in zip(dws[::3], refs)]
    rms = [((d.double() - r).pow(2).mean().sqrt() / r.pow(2).mean().sqrt()).item() for d, r in zip(dws[::3], refs)]
    H.WGRAD_BF16, H.WGRAD_X3 = mode == 'bf16', mode == 'x3'
    t = graph_time(lambda: H.conv_wgrad_multi(jobs))
    print('%-5s slabs of all %d weight gradients %7.1f us   max err vs fp64 %.1e (worst job)  rms err %.1e' %
          (mode, len(jobs), t, max(errs), max(rms)), flush=True)
H.WGRAD_BF16 = H.WGRAD_X3 = False
