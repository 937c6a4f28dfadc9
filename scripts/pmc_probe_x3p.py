"""Launches for counter collection: the k3 s1 conv on x3 operands (conv3_x3p_kernel) and the x3 weight gradient at the
bench shapes.  usage (own rocprofv3 run per counter group):
rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d out -- python3 scripts/pmc_probe_x3p.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepards_amd import hip_ops as H
ROWS = int(os.environ.get('ROWS', 1280))
for ci, co, L in ((512, 512, 7), (256, 256, 14)):
    x = torch.randn(ROWS, L, ci, device='cuda')
    w = torch.randn(co, ci, 3, device='cuda') * 0.05
    uf, ud = H.repack_multi([w], [49])[0][2:]
    x3 = H.x3_split(x)
    y = H.conv3_x3p(x3, uf)
    dy3 = H.x3_split(torch.randn_like(y))
    for _ in range(3):
        H.conv3_x3p(x3, uf, out=y)
        H.conv_wgrad_multi([(dy3, x3, 3, 1, 1)] * 3)
torch.cuda.synchronize()
