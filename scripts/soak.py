"""Soak: many training steps (graph replays) with fresh batches + interleaved test steps; memory must stay flat."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deepards_amd.models as M
from deepards_amd.train import HotPathTrainer
torch.manual_seed(0)
dev = torch.device('cuda:0')
for name in ('resnet18', 'densenet18'):
    model = M.CNNLinearNetwork(getattr(M, name)(), 20, 0).to(dev)
    tr = HotPathTrainer(model)
    B = 64
    g = torch.Generator(device='cuda').manual_seed(1)
    mem0 = None
    t0 = time.time()
    for step in range(int(os.environ.get('STEPS', 2000))):
        x = torch.randn(B, 20, 1, 224, device=dev, generator=g)
        t = torch.zeros(B, 2, device=dev); t[:, step % 2] = 1
        loss = tr.train_step(x, t)
        if step % 500 == 3:
            tr.test_step(x, t)
            torch.cuda.synchronize()
            mem = torch.cuda.memory_allocated() / 2**20
            if mem0 is None: mem0 = mem
            print(name, 'step', step, 'loss %.5f' % float(loss), 'allocated %.0f MiB reserved %.0f MiB' % (mem, torch.cuda.memory_reserved() / 2**20), flush=True)
            assert abs(mem - mem0) < 64, 'memory grows'
    torch.cuda.synchronize()
    assert torch.isfinite(loss).all()
    print(name, 'ok %.1f s' % (time.time() - t0))
