"""Per-kernel times of the dense-block forward kernels at one shape, each captured 20x into a graph (no host launch cost):
usage: python scripts/dense_kernel_probe.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepards_amd import hip_ops as H

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
R, rows = 20, 20 * B
torch.manual_seed(0)


def graph_time(fn, n=20, reps=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps / n * 1e6


for l in (56, 7):
    buf = torch.randn(rows, l, 128, device='cuda')
    st = torch.empty(2, rows // R, 128, device='cuda')
    H.bn_stats_fused(buf[:, :, :64], R, st[0][:, :64], st[1][:, :64])
    g1, b1 = torch.rand(64, device='cuda') + 0.5, torch.randn(64, device='cuda') * 0.1
    g2, b2 = torch.rand(128, device='cuda') + 0.5, torch.randn(128, device='cuda') * 0.1
    w1 = torch.randn(128, 64, 1, device='cuda') * 0.1
    u = H.wino_weights(torch.randn(32, 128, 3, device='cuda') * 0.05)
    y1 = torch.empty(rows, l, 128, device='cuda')
    xv, mv, iv = buf[:, :, :64], st[0][:, :64], st[1][:, :64]
    _, rec1 = H.conv1x1_bn(xv, w1, R, mv, iv, g1, b1, y1, want_records=True)
    m2, i2 = torch.empty(rows // R, 128, device='cuda'), torch.empty(rows // R, 128, device='cuda')
    new = buf[:, :, 64:96]
    h2 = torch.relu(y1)
    _, rec2 = H.conv3_winograd(h2, u, out=new, stats_R=R)
    pl = (l + 1) // 2
    w3 = torch.randn(128, 96, 1, device='cuda') * 0.1
    g3, b3 = torch.rand(96, device='cuda') + 0.5, torch.randn(96, device='cuda') * 0.1
    y3 = torch.empty(rows, l, 128, device='cuda')
    res = [
        ('bn_stats_fused 64ch', lambda: H.bn_stats_fused(xv, R, mv, iv)),
        ('bn_fwd 128ch', lambda: H.bn_fwd(y1, R, g2, b2, relu=True)),
        ('conv1x1_bn plain', lambda: H.conv1x1_bn(xv, w1, R, mv, iv, g1, b1, y1)),
        ('conv1x1_bn + records out', lambda: H.conv1x1_bn(xv, w1, R, mv, iv, g1, b1, y1, want_records=True)),
        ('conv1x1_bn 96ch + pending merge', lambda: H.conv1x1_bn(buf[:, :, :96], w3, R, st[0][:, :96], st[1][:, :96], g3, b3, y3,
                                                                pend=(rec2, 64, rows * pl, R * pl))),
        ('conv3_winograd plain', lambda: H.conv3_winograd(h2, u, out=new)),
        ('conv3_winograd + records out', lambda: H.conv3_winograd(h2, u, out=new, stats_R=R)),
        ('conv3_winograd_bn', lambda: H.conv3_winograd_bn(y1, u, R, rec1, m2, i2, g2, b2, new)),
        ('conv3_winograd_bn + records out', lambda: H.conv3_winograd_bn(y1, u, R, rec1, m2, i2, g2, b2, new, want_records=True)),
    ]
    for name, fn in res:
        print('B %3d L %2d  %-36s %6.1f us' % (B, l, name, graph_time(fn)))
