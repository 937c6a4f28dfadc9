"""Counter probe: Winograd forward (L4 / L3 shapes) and the Winograd weight gradient at the bench batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepards_amd import hip_ops as H
rows = 1280
for ci, L in ((512, 7), (256, 14)):
    x = torch.randn(rows, L, ci, device='cuda'); w = torch.randn(ci, ci, 3, device='cuda') * 0.05
    u = H.wino_weights(w); y = torch.empty_like(x); dy = torch.randn_like(y)
    for _ in range(3):
        H.conv3_winograd(x, u, out=y)
        sl = H.conv_wgrad_multi([(dy, x, 3, 1, 1)] * 3)
torch.cuda.synchronize()
