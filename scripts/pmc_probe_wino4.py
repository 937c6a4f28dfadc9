"""Counter probe: Winograd F(2,3) and F(4,3) forward at the layer4 shape (bench batch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepards_amd import hip_ops as H
rows = 1280
for ci, L in ((512, 7),):
    x = torch.randn(rows, L, ci, device='cuda'); w = torch.randn(ci, ci, 3, device='cuda') * 0.05
    u4, u6 = H.wino_weights(w), H.wino_weights(w, points=6)
    y = torch.empty_like(x)
    for _ in range(3):
        H.conv3_winograd(x, u4, out=y)
        H.conv3_winograd(x, u6, out=y)
torch.cuda.synchronize()
