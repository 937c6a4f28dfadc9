"""Whole bf16 step (conv dtype bf16 + bf16 storage, bench shape) against the padded positions per split of the bf16 weight
gradient (da_wino_debug_pchunk(-n)): 2 368 blocks of 3 a CU at 2 048 -- three rounds and 64 blocks; 2 176 (the default since) fits three.
usage: python scripts/bf16_pchunk_sweep.py [pchunks...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepards_amd import _lib
import deepards_amd.models as M
import deepards_amd.functional as F_
from deepards_amd.train import HotPathTrainer
vals = [int(a) for a in sys.argv[1:]] or [2048, 1536, 1792, 2176, 2560, 3072]
F_.set_conv_dtype('bf16'); F_.set_storage_dtype('bf16')
x = torch.randn(64, 20, 1, 224, device='cuda'); t = torch.zeros(64, 2, device='cuda'); t[:, 0] = 1
for v in vals + vals[:1]:
    _lib.lib().da_wino_debug_pchunk(-v)
    torch.manual_seed(0)
    tr = HotPathTrainer(M.CNNLinearNetwork(M.resnet18(), 20, 0).cuda(), use_graph=True)
    for _ in range(5):
        tr.train_step(x, t)
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50):
            tr.train_step(x, t)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 50)
    print('pchunk %5d: %.4f ms/step' % (v, best * 1e3))
    tr.release_graphs()
_lib.lib().da_wino_debug_pchunk(-2176)
