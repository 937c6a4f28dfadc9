"""Generate tests/golden/*.npz from the REAL reference (run in the build container only).

    python oracle/make_golden.py            # needs /root/reference; never runs on the GPU box

Imports ``deepards.models.{resnet,densenet,torch_cnn_linear_network}`` from /root/reference
(torch-only modules), loads the deterministic weights of ``oracle/weights.py``,
and records logits / loss / gradients / optimiser trajectories as small fixtures.  Nothing from
the reference is copied: the fixtures hold inputs and numeric outputs only.
"""
import os
import sys
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, '/root/reference')

from oracle.weights import param_spec, seeded_params, seeded_batch, DEAD_RESNET_PARAMS, digest as sample  # noqa: E402
from deepards.models.resnet import resnet18                                            # noqa: E402
from deepards.models.densenet import densenet18                                        # noqa: E402
from deepards.models.torch_cnn_linear_network import CNNLinearNetwork                   # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
CLIP, LR, MOM, WD = 0.01, 1e-3, 0.9, 1e-4


def build(backbone, seed, dtype, first_pool_type='max', shift=0.0):
    if backbone == 'resnet18':
        bb = resnet18(first_pool_type=first_pool_type)
    else:
        bb = densenet18(drop_rate=0)
    model = CNNLinearNetwork(bb, 20, 0)
    names = [n for n, _ in model.named_parameters()]
    spec = param_spec(backbone)
    assert names == [s[0] for s in spec], 'param_spec order differs from the reference'
    for (n, shp, _), (_, p) in zip(spec, model.named_parameters()):
        assert tuple(p.shape) == tuple(shp), (n, p.shape, shp)
    sd = {k: torch.from_numpy(v) for k, v in seeded_params(backbone, seed, bn_bias_shift=shift).items()}
    missing = model.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys
    assert all(('running_' in k or 'num_batches' in k) for k in missing.missing_keys), missing
    return model.to(dtype).train()


def run_case(tag, backbone, b, seed, kind, first_pool_type='max', shift=0.0):
    x, tgt = seeded_batch(b, 20, seed, kind)
    rec = dict(x=x, target=tgt, backbone=backbone, seed=seed, b=b,
               first_pool_type=first_pool_type, kind=kind, bn_bias_shift=shift)
    for dt, sfx in ((torch.float64, '64'), (torch.float32, '32')):
        model = build(backbone, seed, dt, first_pool_type, shift)
        xt, tt = torch.from_numpy(x).to(dt), torch.from_numpy(tgt).to(dt)
        out = model(xt, None)
        loss = torch.nn.BCEWithLogitsLoss()(out, tt)
        loss.backward()
        rec['logits' + sfx] = out.detach().numpy().astype(np.float64)
        rec['loss' + sfx] = float(loss)
        # per-window independence (SURVEY finding 3) is implied by the reference's loop
        feat = model.breath_block(xt[0]).detach().numpy()
        rec['feat0_' + sfx] = sample(feat, 256)
        for n, p in model.named_parameters():
            if p.grad is not None:
                rec['grad%s/%s' % (sfx, n)] = sample(p.grad.numpy())
        if backbone == 'resnet18':
            rec['rm%s/bn1' % sfx] = model.breath_block.bn1.running_mean.numpy().astype(np.float64)
            rec['rv%s/bn1' % sfx] = model.breath_block.bn1.running_var.numpy().astype(np.float64)
            l4 = model.breath_block.layer4[1].bn2
            rec['rm%s/layer4.1.bn2' % sfx] = l4.running_mean.numpy().astype(np.float64)
            rec['rv%s/layer4.1.bn2' % sfx] = l4.running_var.numpy().astype(np.float64)

        # optimiser trajectories with the clamp hooks (train_ards_detector.py:416-422,474-476)
        for opt_name in ('sgd', 'adam'):
            model = build(backbone, seed, dt, first_pool_type, shift)
            for p in model.parameters():
                p.register_hook(lambda g: torch.clamp(g, -CLIP, CLIP))
            if opt_name == 'sgd':
                opt = torch.optim.SGD(model.parameters(), lr=LR, momentum=MOM, weight_decay=WD, nesterov=True)
            else:
                opt = torch.optim.Adam(model.parameters(), lr=LR)
            losses = []
            for step in range(3):
                model.zero_grad()
                out = model(xt, None)
                loss = torch.nn.BCEWithLogitsLoss()(out, tt)
                loss.backward()
                opt.step()
                opt.zero_grad()
                losses.append(float(loss))
            rec['%s_losses%s' % (opt_name, sfx)] = np.array(losses)
            for n, p in model.named_parameters():
                if n in DEAD_RESNET_PARAMS:
                    continue
                rec['%s_p%s/%s' % (opt_name, sfx, n)] = sample(p.detach().numpy())
            rec['%s_logits_after%s' % (opt_name, sfx)] = model(xt, None).detach().numpy().astype(np.float64)
    path = os.path.join(OUT, tag + '.npz')
    np.savez_compressed(path, **rec)
    print(tag, 'logits64', rec['logits64'].ravel(), 'loss', rec['loss64'],
          '|l32-l64|max', np.abs(rec['logits32'] - rec['logits64']).max(), os.path.getsize(path), 'B')


if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    run_case('resnet18_b2_randn', 'resnet18', 2, 0, 'randn')
    run_case('resnet18_b4_flow', 'resnet18', 4, 1, 'flow')
    run_case('resnet18_b2_avgpool', 'resnet18', 2, 2, 'randn', first_pool_type='avg')
    run_case('densenet18_b2_randn', 'densenet18', 2, 0, 'randn')
    run_case('densenet18_b4_flow', 'densenet18', 4, 1, 'flow')
    # every ReLU active (BN beta += 6): no activation decision can flip under fp32 rounding
    run_case('resnet18_b2_active', 'resnet18', 2, 3, 'randn', first_pool_type='avg', shift=6.0)
    run_case('densenet18_b2_active', 'densenet18', 2, 3, 'randn', shift=6.0)
