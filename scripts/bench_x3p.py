"""Per-layer timing of the pre-split split-bf16 kernels (conv arithmetic 'f32x3p') against the fp32 Winograd kernels at the
bench batch, hipGraph timed: the k3 s1 conv (conv3_x3p_kernel), the 15 k3 s1 weight gradients in one call
(wgrad_x3p_multi_kernel vs wino_wgrad_multi_kernel), and the BatchNorm kernels with float / x3 stores.
usage: python scripts/bench_x3p.py"""
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


for ci, co, L in ((64, 64, 56), (128, 128, 28), (256, 256, 14), (512, 512, 7)):
    torch.manual_seed(0)
    x = torch.randn(ROWS, L, ci, device='cuda'); w = torch.randn(co, ci, 3, device='cuda') * (2.0 / (3 * co)) ** 0.5
    u = H.wino_weights(w, points=6 if ci >= 512 else 4)
    yw = torch.empty(ROWS, L, co, device='cuda'); H.conv3_winograd(x, u, out=yw)
    uf, ud = H.repack_multi([w], [49])[0][2:]
    x3 = H.x3_split(x)
    yx = H.conv3_x3p(x3, uf)
    tw = graph_time(lambda: H.conv3_winograd(x, u, out=yw))
    tx = graph_time(lambda: H.conv3_x3p(x3, uf, out=yx))
    fl = 2.0 * ROWS * L * ci * co * 3
    print('conv %4d->%4d L %2d | fp32 winograd %6.1f us %6.1f TF(alg) | x3p %6.1f us %6.1f TF(alg) = %4.0f TF bf16-executed  x%.2f' %
          (ci, co, L, tw, fl / tw / 1e6, tx, fl / tx / 1e6, 6 * fl / tx / 1e6, tw / tx), flush=True)
    R = 20
    gamma, beta = torch.ones(co, device='cuda'), torch.zeros(co, device='cuda')
    res = torch.randn_like(yw)
    res3 = H.x3_split(res)
    t0 = graph_time(lambda: H.bn_fwd(yw, R, gamma, beta, relu=True, res=res, want_mask=True))
    t1 = graph_time(lambda: H.bn_fwd_x(yw, R, gamma, beta, relu=True, res=res3, want_mask=True, out_x3=True))
    _, mean, invstd, mask = H.bn_fwd(yw, R, gamma, beta, relu=True, res=res, want_mask=True)
    dout = torch.randn_like(yw)
    t2 = graph_time(lambda: H.bn_bwd(dout, yw, R, mean, invstd, gamma, beta, 2, want_g=True, defer_param_grads=True, mask=mask))
    t3 = graph_time(lambda: H.bn_bwd_x(dout, yw, R, mean, invstd, gamma, beta, 3, want_g=True, mask=mask))
    print('      BatchNorm(+res, ReLU mask) fwd float %5.1f us / x3 stores %5.1f us;  bwd float %5.1f us / x3 dx %5.1f us' %
          (t0, t1, t2, t3), flush=True)

SHAPES = [(64, 64, 56, 4), (128, 128, 28, 3), (256, 256, 14, 3), (512, 512, 7, 3)]
jobs, jobs3 = [], []
for ci, co, l, cnt in SHAPES:
    for _ in range(cnt):
        x = torch.randn(ROWS, l, ci, device='cuda')
        dy = torch.randn(ROWS, l, co, device='cuda') * 1e-4
        jobs.append((dy, x, 3, 1, 1))
        jobs3.append((H.x3_split(dy), H.x3_split(x), 3, 1, 1))
fl = sum(2.0 * ROWS * l * ci * co * 3 * cnt for ci, co, l, cnt in SHAPES)
for name, jj in (('fp32 winograd form', jobs), ('x3 operands', jobs3)):
    t = graph_time(lambda: H.conv_wgrad_multi(jj), reps=5)
    print('all 13 k3 s1 weight gradients, %-20s %7.1f us  %6.1f TF(alg)' % (name, t, fl / t / 1e6), flush=True)
for (ci, co, l, cnt), j in zip(SHAPES, (jobs3[0], jobs3[4], jobs3[7], jobs3[10])):
    t = graph_time(lambda: H.conv_wgrad_multi([j]), reps=10)
    print('   one job %4d->%4d L %2d: x3 operands %6.1f us (%.0f TF alg)' % (ci, co, l, t, 2.0 * ROWS * l * ci * co * 3 / t / 1e6), flush=True)
