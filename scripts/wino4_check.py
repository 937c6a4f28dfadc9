import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepards_amd import hip_ops as H
torch.manual_seed(0)
for (rows, L, ci, co) in ((3, 7, 32, 32), (5, 14, 64, 32), (20, 56, 64, 64), (7, 9, 96, 64), (2, 1, 32, 32), (2, 2, 32, 32), (2, 3, 32, 32), (4, 5, 32, 64), (41, 28, 128, 128), (300, 7, 512, 512), (1280, 56, 64, 64)):
    x = torch.randn(rows, L, ci, device='cuda'); w = torch.randn(co, ci, 3, device='cuda') * 0.05
    ref = torch.nn.functional.conv1d(x.permute(0, 2, 1).double(), w.double(), padding=1).permute(0, 2, 1)
    u6 = H.wino_weights(w, points=6)
    y = H.conv3_winograd(x, u6)
    e = (y.double() - ref).abs().max().item() / ref.abs().max().item()
    # dgrad taps + accumulate
    ud = H.wino_weights(w, transpose=True, points=6)
    dy = torch.randn(rows, L, co, device='cuda')
    base = torch.randn(rows, L, ci, device='cuda')
    out = base.clone()
    H.conv3_winograd(dy, ud, out=out, accumulate=True)
    refd = torch.nn.grad.conv1d_input((rows, ci, L), w.double(), dy.permute(0, 2, 1).double(), padding=1).permute(0, 2, 1) + base.double()
    ed = (out.double() - refd).abs().max().item() / refd.abs().max().item()
    print(rows, L, ci, co, 'fwd %.2e dgrad+acc %.2e' % (e, ed))
    assert e < 2e-5 and ed < 2e-5
print('ok')
