"""RCCL smoke on ONE GPU: a world-size-1 'nccl' process group carries every collective the data-parallel path issues
(flat fp32 / int64 broadcasts of sync_replicas, the seed broadcast of shared_generator, the sum all-reduce of the 15.5 MB
gradient bucket between two graph replays, the float64 MAX all-reduce and the barrier of bench.py).  World size 1 moves no
data between GPUs -- this only proves that the calls, dtypes and stream ordering are accepted by RCCL on this stack."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29577')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
import deepards_amd.models as M
from deepards_amd.train import HotPathTrainer, shared_generator
model = M.CNNLinearNetwork(M.resnet18(), 20, 0).cuda()
tr = HotPathTrainer(model, world_size=1, use_graph=True)
tr.world_size = 2                       # take the data-parallel code paths (a lone rank "sums" its own gradients: x1)
tr._synced = False
tr.sync_replicas()
g = shared_generator(tr, None)
print('shared generator seed ok', torch.randperm(5, generator=g).tolist())
x = torch.randn(8, 20, 1, 224, device='cuda'); t = torch.zeros(8, 2, device='cuda'); t[:, 0] = 1
losses = [float(tr.train_step(x, t)) for _ in range(4)]
print('losses', losses, 'allreduce calls', tr.allreduce_calls, 'graphs', len(tr._graphs))
tt = torch.tensor([1.5], device='cuda', dtype=torch.float64)
dist.all_reduce(tt, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    tr.bucket.allreduce()
torch.cuda.synchronize()
print('all-reduce of %d bytes on a 1-rank RCCL group: %.3f ms each' % (tr.bucket.numel * 4, (time.perf_counter() - t0) / 20 * 1e3))
assert all(l == l for l in losses) and tr.allreduce_calls == 4
print('all-reduce inside the captured step:', tr.allreduce_in_graph)

# the captured data-parallel step (backward | RCCL all-reduce | update in ONE graph when the capture is accepted) against
# the same step run eagerly: bit for bit, 3 steps, from the same initialisation
def run(use_graph, capture):
    torch.manual_seed(7)
    m = M.CNNLinearNetwork(M.resnet18(), 20, 0).cuda()
    r = HotPathTrainer(m, world_size=1, use_graph=use_graph)
    r.world_size, r._synced, r._capture_allreduce = 2, True, capture
    ls = [float(r.train_step(x, t)) for _ in range(4)]
    torch.cuda.synchronize()
    flat = r.bucket.p.clone()
    in_graph = r.allreduce_in_graph
    r.release_graphs()
    return ls, flat, in_graph
le, pe, _ = run(False, False)
lt, pt, two = run(True, False)
lg, pg, one = run(True, True)
assert not two and le == lt and torch.equal(pe, pt), 'two-graph form differs from eager'
assert le == lg and torch.equal(pe, pg), 'single-graph form differs from eager'
print('captured all-reduce == eager bit for bit; in-graph capture accepted by RCCL as ONE chain: %s' % one)

# the capture's failure path: the all-reduce raises INSIDE the first capture (what a process group that refuses to be
# captured does) -- the trainer must end in the two-graph form, every rank agreeing, with the eager run's losses
from deepards_amd.train import FlatBucket
_orig = FlatBucket.allreduce
_fired = []
def _refusing(self, group=None):
    if torch.cuda.is_current_stream_capturing() and not _fired:
        _fired.append(1)
        raise RuntimeError('injected: collective refused inside a capture')
    return _orig(self, group)
FlatBucket.allreduce = _refusing
import warnings
with warnings.catch_warnings(record=True) as wlist:
    warnings.simplefilter('always')
    lf, pf, in_graph = run(True, True)
FlatBucket.allreduce = _orig
assert _fired and not in_graph, 'the injected refusal did not select the two-graph form'
assert any('two-graph form' in str(w.message) for w in wlist), [str(w.message) for w in wlist]
assert le == lf and torch.equal(pe, pf), 'two-graph form after a refused capture differs from eager'
print('refused capture -> two-graph form, losses and parameters == eager bit for bit')
dist.destroy_process_group()
print('rccl world-1 probe ok')
