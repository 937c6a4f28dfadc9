import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepards_amd import hip_ops as H
P, I = ctypes.c_void_p, ctypes.c_int
ops = {}
for c in (256, 512):
    rows, L = 1024 * 512 // c, 8          # 256 full tiles, one round
    x = torch.randn(rows, L, c, device='cuda'); w = torch.randn(c, c, 3, device='cuda') * 0.03
    ops[c] = (rows, L, H.repack_multi([w], [49])[0][2], H.x3_split(x), torch.zeros(rows, L, c, device='cuda'))
for path in sys.argv[1:]:
    lib = ctypes.CDLL(os.path.abspath(path)); f = lib.da_conv3_x3p; f.restype = I; f.argtypes = [P, P, P] + [I] * 6 + [P]
    res = {}
    for c in (256, 512):
        rows, L, uf, x3, y = ops[c]
        best = None
        for rep in range(3):
            for _ in range(200): f(x3.data_ptr(), uf.data_ptr(), y.data_ptr(), rows, L, c, c, c, 0, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            cyc, rt = y.view(-1)[0].item(), y.view(-1)[1].item()
            if rt > 0 and (best is None or cyc < best[0]): best = (cyc, rt)
        res[c] = best
    (c1, r1), (c2, r2) = res[256], res[512]
    print('%-14s 16 steps %7.0f cyc  32 steps %7.0f cyc  -> %5.0f cycles/step, fixed %6.0f;  clock %.2f GHz' %
          (os.path.basename(path), c1, c2, (c2 - c1) / 16, c1 - (c2 - c1), c2 / r2 / 10), flush=True)
