"""Randomised shape fuzz of the conv kernels (not part of the test suite): Winograd forward / data gradient / weight
gradient and the shared-launch forms against the direct kernels and, on small cases, the numpy oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from deepards_amd import hip_ops as H
from oracle import np_ref
rng = np.random.default_rng(int(os.environ.get('SEED', 0)))
n_cases = int(os.environ.get('CASES', 120))
worst = 0.0
for case in range(n_cases):
    ci = 32 * int(rng.integers(1, 9)); co = 32 * int(rng.integers(1, 9))
    L = int(rng.integers(1, 61)); rows = int(rng.choice([1, 2, 3, 7, 20, 40, 61, 200, 333]))
    if rows * L * max(ci, co) > 6e6: rows = max(1, int(6e6 / (L * max(ci, co))))
    x = torch.randn(rows, L, ci, device='cuda'); w = torch.randn(co, ci, 3, device='cuda') * (2.0 / (3 * co)) ** 0.5
    dy = torch.randn(rows, L, co, device='cuda')
    wf, wd = H.repack_weight(w, True, True)
    (_, _, uf, ud), = H.repack_multi([w], [True])
    y_d = H.conv_fwd(x, wf, 1, 1); y_w = H.conv3_winograd(x, uf)
    dx_d = H.conv_dgrad(dy, wd, 1, 1, L); dx_w = H.conv3_winograd(dy, ud)
    sc = lambda a: float(a.abs().max()) + 1e-12
    e1 = float((y_d - y_w).abs().max()) / sc(y_d); e2 = float((dx_d - dx_w).abs().max()) / sc(dx_d)
    (_, _, uf6, ud6), = H.repack_multi([w], [6])                     # F(4,3): both K-step variants of the kernel
    from deepards_amd import _lib
    _lib.lib().da_wino_debug_tail(2 + case % 2)
    e1 = max(e1, float((y_d - H.conv3_winograd(x, uf6)).abs().max()) / sc(y_d))
    base = torch.randn_like(dx_d); acc = base.clone(); H.conv3_winograd(dy, ud6, out=acc, accumulate=True)
    e2 = max(e2, float((dx_d + base - acc).abs().max()) / sc(dx_d))
    _lib.lib().da_wino_debug_tail(3)
    try:          # the direct weight-gradient kernels exist for the tile pairs the networks need, not for every (co, ci)
        dw_d = H.conv_wgrad(dy, x, 3, 1, 1)
        slabs = H.conv_wgrad_multi([(dy, x, 3, 1, 1)])
        dw_m = torch.zeros_like(dw_d); H.wgrad_reduce_multi(list(zip(slabs, [dw_m])), accumulate=True)
        e3 = float((dw_d - dw_m).abs().max()) / sc(dw_d)
    except H.HipError:
        e3 = 0.0
    e4 = 0.0
    if rows * L * ci * co < 3e7:
        xn, wn, dyn = [t.permute(0, 2, 1).double().cpu().numpy() if t.dim() == 3 and t is not w else t.double().cpu().numpy() for t in (x, w, dy)]
        y_ref = np_ref.conv1d_fwd(xn, wn, 1, 1)
        e4 = float(np.abs(y_w.permute(0, 2, 1).double().cpu().numpy() - y_ref).max()) / (np.abs(y_ref).max() + 1e-12)
    m = max(e1, e2, e3, e4); worst = max(worst, m)
    flag = '' if m < 2e-5 else '   <-- CHECK'
    if flag or case % 20 == 0:
        print('case %3d ci %3d co %3d L %2d rows %3d: fwd %.1e dgrad %.1e wgrad %.1e oracle %.1e%s' % (case, ci, co, L, rows, e1, e2, e3, e4, flag), flush=True)
# stride-2 block heads
for case in range(30):
    ci = 64 * int(rng.integers(1, 5)); co = 2 * ci; L = 2 * int(rng.integers(1, 30)); rows = int(rng.choice([1, 5, 40, 200]))
    x = torch.randn(rows, L, ci, device='cuda')
    w1 = torch.randn(co, ci, 3, device='cuda') * 0.05; wdn = torch.randn(co, ci, 1, device='cuda') * 0.05
    wf1, wd1 = H.repack_weight(w1, True, True); wfd, wdd = H.repack_weight(wdn, True, True)
    y1, yd = H.conv_fwd_multi([(x, wf1, 2, 1), (x, wfd, 2, 0)])
    e1 = float((y1 - H.conv_fwd(x, wf1, 2, 1)).abs().max()) + float((yd - H.conv_fwd(x, wfd, 2, 0)).abs().max())
    dy1, dyd = torch.randn_like(y1), torch.randn_like(yd)
    dx = H.conv_dgrad_s2_pair(dy1, wd1, dyd, wdd, L)
    one = H.conv_dgrad(dy1, wd1, 2, 1, L); H.conv_dgrad(dyd, wdd, 2, 0, L, out=one, accumulate=True)
    e2 = float((dx - one).abs().max()) / (float(one.abs().max()) + 1e-12)
    worst = max(worst, e1, e2)
    if max(e1, e2) > 2e-5: print('s2 case', case, ci, co, L, rows, e1, e2, '<-- CHECK')
print('worst relative difference %.2e over %d + 30 cases' % (worst, n_cases))
