"""Run one conv shape repeatedly (for PMC profiling).  env: SHAPE=ci,co,k,s,L  ROWS  MODE=fwd|dgrad|wgrad"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepards_amd import hip_ops as H, _lib
if os.environ.get('TILE'): _lib.lib().da_debug_set(0, int(os.environ['TILE']))
ci, co, k, s, L = [int(v) for v in os.environ.get('SHAPE', '512,512,3,1,7').split(',')]
rows = int(os.environ.get('ROWS', 1280)); mode = os.environ.get('MODE', 'fwd')
pad = (k - 1) // 2
x = torch.randn(rows, L, ci, device='cuda'); w = torch.randn(co, ci, k, device='cuda') * 0.05
wf, wd = H.repack_weight(w, True, True)
y = H.conv_fwd(x, wf, s, pad); dy = torch.randn_like(y); dx = torch.empty_like(x); dw = torch.empty_like(w)
for _ in range(int(os.environ.get('REPS', 10))):
    if mode == 'fwd': H.conv_fwd(x, wf, s, pad, out=y)
    elif mode == 'dgrad': H.conv_dgrad(dy, wd, s, pad, L, out=dx)
    else: H.conv_wgrad(dy, x, k, s, pad, out=dw)
torch.cuda.synchronize()
