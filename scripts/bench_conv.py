"""Per-shape microbenchmark of the conv GEMM kernels at the bench batch (B=64 windows -> 1280 rows).
usage: python scripts/bench_conv.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepards_amd import hip_ops as H

ROWS = int(os.environ.get('ROWS', 1280))
from deepards_amd import _lib
if os.environ.get('TILE'): _lib.lib().da_debug_set(0, int(os.environ['TILE']))
if os.environ.get('HALO'): _lib.lib().da_debug_set(2, int(os.environ['HALO']))
if os.environ.get('WGB'): _lib.lib().da_debug_set(1, int(os.environ['WGB']))
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 20
SHAPES = [  # ci, co, k, stride, L
    (64, 64, 3, 1, 56), (64, 128, 3, 2, 56), (64, 128, 1, 2, 56), (128, 128, 3, 1, 28),
    (128, 256, 3, 2, 28), (128, 256, 1, 2, 28), (256, 256, 3, 1, 14),
    (256, 512, 3, 2, 14), (256, 512, 1, 2, 14), (512, 512, 3, 1, 7),
    (64, 128, 1, 1, 56), (96, 128, 1, 1, 56), (128, 32, 3, 1, 56), (128, 64, 1, 1, 56),
]

def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REPS * 1e3   # us

print('%-26s %10s | %8s %7s | %8s %7s | %8s %7s' % ('shape ci,co,k,s,L', 'GFLOP', 'fwd us', 'TF', 'dgrad us', 'TF', 'wgrad us', 'TF'))
tot = [0, 0, 0, 0]
for ci, co, k, s, L in SHAPES:
    pad = (k - 1) // 2
    x = torch.randn(ROWS, L, ci, device='cuda')
    w = torch.randn(co, ci, k, device='cuda') * 0.05
    wf, wd = H.repack_weight(w, True, True)
    y = H.conv_fwd(x, wf, s, pad)
    dy = torch.randn_like(y)
    dx = torch.empty_like(x)
    dw = torch.empty_like(w)
    fl = 2.0 * ROWS * y.shape[1] * ci * co * k
    t_f = timeit(lambda: H.conv_fwd(x, wf, s, pad, out=y))
    t_d = timeit(lambda: H.conv_dgrad(dy, wd, s, pad, L, out=dx, accumulate=(k == 1 and s == 2)))
    t_w = timeit(lambda: H.conv_wgrad(dy, x, k, s, pad, out=dw))
    print('%-26s %10.3f | %8.1f %7.1f | %8.1f %7.1f | %8.1f %7.1f' % ((ci, co, k, s, L), fl / 1e9, t_f, fl / t_f / 1e6, t_d, fl / t_d / 1e6, t_w, fl / t_w / 1e6))
