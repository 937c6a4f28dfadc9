"""Would two independent half-batch chains on two streams beat one B=64 chain?  (windows are independent all the way
through the network.)  Proxy: two trainers with B=32 each, their graphs replayed on two streams, vs one B=64 trainer."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deepards_amd.models as M
from deepards_amd.train import HotPathTrainer
dev = torch.device('cuda:0')
def mk(B):
    torch.manual_seed(0)
    m = M.CNNLinearNetwork(M.resnet18(), 20, 0).to(dev)
    tr = HotPathTrainer(m)
    x = torch.randn(B, 20, 1, 224, device=dev); t = torch.zeros(B, 2, device=dev); t[:, 0] = 1
    for _ in range(4): tr.train_step(x, t)
    return tr
one = mk(64)
a, b = mk(32), mk(32)
torch.cuda.synchronize()
def timeit(fn, n=40):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
t1 = timeit(lambda: one._graph[0].replay())
th = timeit(lambda: a._graph[0].replay())
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def both():
    with torch.cuda.stream(s1): a._graph[0].replay()
    with torch.cuda.stream(s2): b._graph[0].replay()
t2 = timeit(both)
print('one B=64 chain      %.3f ms  -> %.0f breath-seq/s' % (t1, 1280 / t1 * 1e3))
print('one B=32 chain      %.3f ms  -> %.0f' % (th, 640 / th * 1e3))
print('two B=32 chains ||  %.3f ms  -> %.0f breath-seq/s' % (t2, 1280 / t2 * 1e3))
