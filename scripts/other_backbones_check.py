import sys, os, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
import deepards_amd.models as M
from deepards_amd.train import HotPathTrainer
for name in ('resnet34', 'densenet121', 'densenet169', 'densenet201'):
    torch.manual_seed(0)
    model = M.CNNLinearNetwork(M.base_networks[name](), 20, 0).cuda()
    tr = HotPathTrainer(model)
    x = torch.randn(16, 20, 1, 224, device='cuda'); t = torch.zeros(16, 2, device='cuda'); t[:, 0] = 1
    l = [float(tr.train_step(x, t)) for _ in range(4)]
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(10): tr.train_step(x, t)
    torch.cuda.synchronize()
    print(name, 'losses', ['%.4f' % v for v in l], '%.2f ms/step B=16' % ((time.time() - t0) * 100))
