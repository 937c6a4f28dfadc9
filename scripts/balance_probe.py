import os, sys
sys.path.insert(0, '/root/repo' if os.path.isdir('/root/repo/deepards_amd') else os.getcwd())
import torch
from deepards_amd import hip_ops as H
def timeit(fn, reps=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for rows, L in ((1024, 8), (1152, 8), (1170, 7), (1280, 7), (1280, 8), (1463, 7), (1536, 8), (2048, 8), (2304, 8)):
    for ci, co in ((512, 512), (256, 256)):
        x = torch.randn(rows, L, ci, device='cuda'); w = torch.randn(co, ci, 3, device='cuda') * 0.05
        wf, _ = H.repack_weight(w, True, True)
        y = H.conv_fwd(x, wf, 1, 1)
        t = timeit(lambda: H.conv_fwd(x, wf, 1, 1, out=y))
        fl = 2.0 * rows * L * ci * co * 3
        tiles = ((rows * L + 63) // 64) * (co // 64)
        print('rows %5d L %d %d->%d tiles %5d (%.3f/CU) %7.1f us %6.1f TF' % (rows, L, ci, co, tiles, tiles / 256, t, fl / t / 1e6))
