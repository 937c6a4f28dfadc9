"""Is there idle GPU time between graph replays?  train_step (copies + replay) vs bare replay loops."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deepards_amd.models as M
from deepards_amd.train import HotPathTrainer
torch.manual_seed(0)
dev = torch.device('cuda:0')
model = M.CNNLinearNetwork(M.densenet18() if os.environ.get('DN') else M.resnet18(), 20, 0).to(dev)
B = 64
x = torch.randn(B, 20, 1, 224, device=dev); t = torch.zeros(B, 2, device=dev); t[:, 0] = 1
tr = HotPathTrainer(model)
for _ in range(4): tr.train_step(x, t)
torch.cuda.synchronize()
def timeit(fn, n=50):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print('train_step        %.4f ms' % timeit(lambda: tr.train_step(x, t)))
print('bare graph replay %.4f ms' % timeit(lambda: tr._graph[0].replay()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): tr._graph[0].replay()
e1.record(); torch.cuda.synchronize()
print('bare replay (events) %.4f ms' % (e0.elapsed_time(e1) / 50))
t0 = time.perf_counter()
for _ in range(50): tr._graph[0].replay()
print('cpu time per replay call %.4f ms' % ((time.perf_counter() - t0) / 50 * 1e3)); torch.cuda.synchronize()
