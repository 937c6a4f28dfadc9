import sys, torch, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepards_amd.models as M
from deepards_amd.functional import bce_with_logits
for bb in ('resnet18', 'densenet18'):
    for nb in (1, 3, 20):
        model = M.CNNLinearNetwork(M.base_networks[bb](), nb, 0).cuda().train()
        for b in (0, 1):
            x = torch.randn(b, nb, 1, 224, device='cuda')
            try:
                out = model(x, None)
                msg = 'out %s' % (tuple(out.shape),)
                if b:
                    t = torch.zeros(b, 2, device='cuda'); t[:, 0] = 1
                    bce_with_logits(out, t).backward()
                    msg += ' finite grads %s' % all(torch.isfinite(p.grad).all().item() for p in model.parameters() if p.grad is not None)
            except Exception as e:
                msg = 'raises %s: %s' % (type(e).__name__, str(e)[:80])
            print(bb, 'nb', nb, 'B', b, msg, flush=True)
