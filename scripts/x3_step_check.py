"""One cnn_linear+resnet18 train step at the bench batch under conv arithmetic 'f32' (native fp32 MFMA, Winograd) and
'f32x3' (split-bf16 products): logits / gradient differences between the two and against each other's noise, and the
captured step time of each.   usage: python scripts/x3_step_check.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deepards_amd.models as M
from deepards_amd import functional as F_
from deepards_amd.train import HotPathTrainer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
torch.manual_seed(0)
x = torch.randn(B, 20, 1, 224, device='cuda')
t = torch.zeros(B, 2, device='cuda'); t[torch.arange(B), torch.randint(0, 2, (B,))] = 1


def run(mode, steps=30):
    F_.set_conv_dtype(mode)
    torch.manual_seed(1)
    m = M.CNNLinearNetwork(M.resnet18(), 20, 0).cuda()
    tr = HotPathTrainer(m, optimizer='sgd', use_graph=True)
    with torch.no_grad():
        logits = m(x, None).clone()
    losses = [float(tr.train_step(x, t)) for _ in range(5)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.train_step(x, t)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    p = torch.cat([q.detach().flatten() for q in m.parameters()]).clone()
    tr.release_graphs()
    return logits, losses, p, dt


res = {mode: run(mode) for mode in ('f32', 'f32x3')}
(l0, s0, p0, d0), (l1, s1, p1, d1) = res['f32'], res['f32x3']
print('logits   max |f32 - f32x3| = %.2e (scale %.2e)' % ((l0 - l1).abs().max().item(), l0.abs().max().item()))
print('losses   f32  %s\n         x3   %s' % (' '.join('%.6f' % v for v in s0), ' '.join('%.6f' % v for v in s1)))
print('params after 35 steps: max |diff| %.2e (scale %.2e)' % ((p0 - p1).abs().max().item(), p0.abs().max().item()))
print('step time  f32 %.3f ms   f32x3 %.3f ms   (x%.3f)' % (d0 * 1e3, d1 * 1e3, d0 / d1))
