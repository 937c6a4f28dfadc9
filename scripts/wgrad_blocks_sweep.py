"""Whole fp32 step (bench shape) against the per-job block target of the direct weight-gradient plan (da_debug_set(1, n);
0 = the plan's own choice among 512 / 768 / 1024): do the six direct jobs of a step -- one launch -- fill whole rounds?
usage: python scripts/wgrad_blocks_sweep.py [--densenet] [targets...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepards_amd import _lib
import deepards_amd.models as M
from deepards_amd.train import HotPathTrainer
backbone = 'densenet18' if '--densenet' in sys.argv else 'resnet18'
targets = [int(a) for a in sys.argv[1:] if not a.startswith('--')] or [0, 256, 384, 512, 640, 768, 1024]
B = int(os.environ.get('BATCH', 64))
x = torch.randn(B, 20, 1, 224, device='cuda'); t = torch.zeros(B, 2, device='cuda'); t[:, 0] = 1
for tg in targets + targets[:1]:
    _lib.lib().da_debug_set(1, tg)
    torch.manual_seed(0)
    tr = HotPathTrainer(M.CNNLinearNetwork(getattr(M, backbone)(), 20, 0).cuda(), use_graph=True)
    for _ in range(5):
        tr.train_step(x, t)
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50):
            tr.train_step(x, t)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 50)
    print('target %5d: %.4f ms/step' % (tg, best * 1e3))
    tr.release_graphs()
_lib.lib().da_debug_set(1, 0)
