"""Ad-hoc robustness sweep: whole-model forward/backward at many batch sizes, fast paths (Winograd, tail tiles, paired
launches, fused BN) vs the plain direct kernels (subprocess with the switches off): logits and gradients must agree."""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    import numpy as np, torch
    import deepards_amd.models as M
    from deepards_amd.functional import bce_with_logits
    from oracle.weights import seeded_params, seeded_batch
    out = {}
    for backbone in ('resnet18', 'densenet18'):
        for b in (1, 2, 3, 5, 13, 16, 33, 64, 65, 96, 128, 200):
            bb = M.resnet18() if backbone == 'resnet18' else M.densenet18(drop_rate=0.0)
            model = M.CNNLinearNetwork(bb, 20, 0)
            model.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_params(backbone, 1).items()}, strict=False)
            model = model.cuda().train()
            x, t = seeded_batch(b, 20, b)
            o = model(torch.from_numpy(x).cuda(), None)
            loss = bce_with_logits(o, torch.from_numpy(t).cuda())
            loss.backward()
            gs = [float(p.grad.double().abs().sum()) for p in model.parameters() if p.grad is not None]
            out['%s_%d' % (backbone, b)] = dict(logits=o.detach().cpu().double().numpy().tolist(), loss=float(loss), gsum=gs)
    json.dump(out, open(sys.argv[2], 'w'))
    sys.exit(0)
BF16 = os.environ.get('SWEEP_BF16') == '1'      # bf16-operand convs vs the plain fp32 kernels (looser bounds)
env_fast = dict(os.environ, DA_CONV_DTYPE='bf16') if BF16 else dict(os.environ)
env_slow = dict(os.environ, DA_WINOGRAD='0')
for name, env in (('fast', env_fast), ('slow', env_slow)):
    subprocess.check_call([sys.executable, __file__, 'child', '/tmp/sweep_%s.json' % name], env=env)
import numpy as np
a, b = json.load(open('/tmp/sweep_fast.json')), json.load(open('/tmp/sweep_slow.json'))
worst = 0
for k in a:
    la, lb = np.array(a[k]['logits']), np.array(b[k]['logits'])
    e = np.abs(la - lb).max()
    ga, gb = np.array(a[k]['gsum']), np.array(b[k]['gsum'])
    ge = (np.abs(ga - gb) / (np.abs(gb) + 1e-12)).max()
    worst = max(worst, e)
    flag = '' if e < (5e-2 if BF16 else 2e-5) and ge < (0.3 if BF16 else 2e-2) else '   <-- CHECK'
    print('%-16s logits diff %.2e  loss %.7f vs %.7f  grad-abs-sum rel diff %.2e%s' % (k, e, a[k]['loss'], b[k]['loss'], ge, flag))
print('worst logits diff %.2e' % worst)
