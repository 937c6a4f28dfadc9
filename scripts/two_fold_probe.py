"""Two independent training replicas (two k-folds of the reference's 5-fold protocol, each its own model / optimizer /
captured step) on ONE GPU: replayed back to back on one stream vs concurrently on streams placed by measurement
(train.place_replicas_on_streams), with and without the stem-backward fork inside the captured step.
usage: python scripts/two_fold_probe.py [B] [n_replicas] [resnet18|densenet18]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deepards_amd.models as M
from deepards_amd import functional as F_
from deepards_amd.train import HotPathTrainer, place_replicas_on_streams

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
NR = int(sys.argv[2]) if len(sys.argv) > 2 else 2
BB = sys.argv[3] if len(sys.argv) > 3 else 'resnet18'
torch.manual_seed(0)
x = torch.randn(B, 20, 1, 224, device='cuda')
t = torch.zeros(B, 2, device='cuda'); t[torch.arange(B), torch.randint(0, 2, (B,))] = 1
for fork in (False,):        # (the captured step has been a single chain since round 3)
    trs = []
    for r in range(NR):
        torch.manual_seed(r)
        m = M.CNNLinearNetwork(getattr(M, BB)(), 20, 0).cuda()
        tr = HotPathTrainer(m, optimizer='sgd', use_graph=True)
        for _ in range(3):
            tr.train_step(x, t)
        torch.cuda.synchronize()
        trs.append(tr)
    statics = [tr.static_batch() for tr in trs]

    def run(streams, steps=40):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            for tr, s, st in zip(trs, streams, statics):
                with torch.cuda.stream(s):
                    tr.train_step(st[0], st[1])
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps
    one = torch.cuda.Stream()
    a = min(run([one] * NR) for _ in range(2))
    placed = place_replicas_on_streams(trs)
    b = min(run(placed) for _ in range(2))
    print('%s stem fork %-5s %d replicas, B=%d each: one stream %.3f ms per round (%.0f breath-seq/s)   own streams %.3f ms '
          '(%.0f breath-seq/s)  x%.3f' % (BB, fork, NR, B, a * 1e3, NR * B * 20 / a, b * 1e3, NR * B * 20 / b, a / b), flush=True)
    for tr in trs:
        tr.release_graphs()
