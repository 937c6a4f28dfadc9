"""The recomputing default stem against the stored-map kernels it replaces, piece by piece, hipGraph timed at the bench batch.
usage: python scripts/stem_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepards_amd import hip_ops as H, _lib
def graph_time(fn, reps=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
rows, R, lin, c = 1280, 20, 224, 64
x = torch.randn(rows, lin, device='cuda'); w = torch.randn(c, 1, 7, device='cuda') * 0.4
gamma = torch.rand(c, device='cuda') + 0.5; beta = torch.randn(c, device='cuda') * 0.3
y0 = H.stem_conv_fwd(x, w); mean, invstd = H.bn_stats(y0, R, 1e-5)
out = H.bn_relu_pool_fwd(y0, R, mean, invstd, gamma, beta, 0)
dout = torch.randn_like(out)
L = _lib.lib(); P = H._p; S = H._stream
wn = rows // R; lc = lin // 2
part = H._bn_ws(wn, R * lc, c, x.device)
print('old fwd: conv %.1f  stats %.1f  apply+pool %.1f' % (graph_time(lambda: H.stem_conv_fwd(x, w)), graph_time(lambda: H.bn_stats(y0, R, 1e-5)), graph_time(lambda: H.bn_relu_pool_fwd(y0, R, mean, invstd, gamma, beta, 0))))
print('new fwd: stats partial %.1f  apply+pool %.1f  (whole %.1f)' % (
    graph_time(lambda: L.da_stem_stats_partial(P(x), P(w), rows, R, lin, c, P(part), S())),
    graph_time(lambda: L.da_stem_bn_relu_pool_fwd(P(x), P(w), P(out), c, rows, R, lin, c, P(mean), P(invstd), P(gamma), P(beta), 0, 0, S())),
    graph_time(lambda: H.stem_fused_fwd(x, w, R, gamma, beta, 0))))
dz = torch.empty_like(y0)
def old_bwd():
    d = H.pool_bwd(dout, y0, R, mean, invstd, gamma, beta, 0)
    H.bn_bwd(d, y0, R, mean, invstd, gamma, beta, 1, dx=d, defer_param_grads=True)
    H.stem_conv_wgrad(d, x)
print('old bwd: pool_bwd %.1f  whole %.1f' % (graph_time(lambda: H.pool_bwd(dout, y0, R, mean, invstd, gamma, beta, 0)), graph_time(old_bwd)))
print('new bwd: whole %.1f' % graph_time(lambda: H.stem_fused_bwd(dout, x, w, R, mean, invstd, gamma, beta, 0)))
