"""Times da_conv3_x3p of stand-alone builds of conv_x3p.hip (ablations / experiments: build_variants/*.so, each one the
file compiled alone with some -D switch) at shapes that isolate one tile kind.  Timing only: ablated variants compute
garbage.  usage: python scripts/x3p_variant_probe.py build_variants/a.so [b.so ...]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepards_amd import hip_ops as H

SHAPES = ((1024, 8, 512, 'one full round'), (2048, 8, 512, 'two full rounds'), (256, 8, 512, '256 quarter tiles'),
          (1280, 7, 512, 'bench 512'), (1280, 14, 256, 'bench 256'), (1280, 28, 128, 'bench 128'), (1280, 56, 64, 'bench 64'))


def graph_time(fn, reps=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


ops = []
for rows, L, c, what in SHAPES:
    torch.manual_seed(0)
    x = torch.randn(rows, L, c, device='cuda'); w = torch.randn(c, c, 3, device='cuda') * (2.0 / (3 * c)) ** 0.5
    uf = H.repack_multi([w], [49])[0][2]
    x3 = H.x3_split(x)
    ops.append((rows, L, c, what, x3, uf, H.conv3_x3p(x3, uf)))
print('%-28s' % 'variant' + ''.join('%18s' % s[3] for s in SHAPES))
P, I = ctypes.c_void_p, ctypes.c_int
libs = []
for path in sys.argv[1:]:
    lib = ctypes.CDLL(os.path.abspath(path))
    f = lib.da_conv3_x3p; f.restype = I; f.argtypes = [P, P, P] + [I] * 6 + [P]
    libs.append((path, f))


def runner(f, op, y):
    rows, L, c, what, x3, uf, ref = op
    def run():
        rc = f(x3.data_ptr(), uf.data_ptr(), y.data_ptr(), rows, L, c, c, c, 0, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
    return run


# the first measurements of a process run slow (clocks): warm up on the first variant, then take the best of 3 passes
ys = [torch.empty_like(op[6]) for op in ops]
for _ in range(3):
    for op, y in zip(ops, ys): graph_time(runner(libs[0][1], op, y))
best = {}
for rep in range(3):
    for path, f in libs:
        for i, (op, y) in enumerate(zip(ops, ys)):
            t = graph_time(runner(f, op, y))
            ok = bool(torch.equal(y, op[6]))
            k = (path, i)
            if k not in best or t < best[k][0]: best[k] = (t, ok)
for path, f in libs:
    print('%-28s' % os.path.basename(path) + ''.join('%18s' % ('%12.1f us %s' % (best[(path, i)][0], '=' if best[(path, i)][1] else 'x'))
                                                    for i in range(len(ops))), flush=True)
