"""Two launches for counter collection: the k3 conv forward and its weight gradient at high occupancy.
usage (own rocprofv3 run per counter group): rocprofv3 --pmc <counters> --kernel-trace -d out -- python3 scripts/pmc_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepards_amd import hip_ops as H
rows, L, ci, co = int(os.environ.get('ROWS', 2340)), 7, 512, 512      # 2340*7/64 = 256 m-tiles x 8 = 2048 tiles: 8 per CU
x = torch.randn(rows, L, ci, device='cuda')
w = torch.randn(co, ci, 3, device='cuda') * 0.05
wf, wd = H.repack_weight(w, True, True)
y = H.conv_fwd(x, wf, 1, 1)
dy = torch.randn_like(y)
dw = torch.empty_like(w)
for _ in range(3):
    H.conv_fwd(x, wf, 1, 1, out=y)
    H.conv_wgrad(dy, x, 3, 1, 1, out=dw)
torch.cuda.synchronize()
