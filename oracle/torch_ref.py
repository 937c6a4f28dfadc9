"""CPU oracle #2 (TEST / BASELINE INFRASTRUCTURE ONLY): the reference's CPU path re-assembled from
stock ``torch.nn.functional`` ops.

The reference's hot path *is* stock PyTorch (SURVEY.md finding 1), so its honest CPU baseline is the
same ATen/MKLDNN kernels in the same order: per window i, ``breath_block(x[i])`` then
``linear_final(view(-1))`` then ``torch.cat`` (models/torch_cnn_linear_network.py:104-113), BCE-with-
logits, backward, +-clip clamp hooks, SGD-Nesterov (train_ards_detector.py:161-173,416-422,474-476).
This file states that path functionally over a ``{state_dict key: tensor}`` dict (names from
``oracle/weights.py:param_spec``).  ``tests/test_oracle_golden.py`` pins it to the golden vectors and
(in the build container) checks it bit-for-bit against the imported reference.

Used by ``bench.py`` only for the ``cpu_baseline`` leg ("kind": "port") and by tests.
"""
import torch
import torch.nn.functional as F


def _bn(p, prefix, x):
    # train-mode statistics, never eval (train_ards_detector.py:448 is commented out)
    return F.batch_norm(x, None, None, p[prefix + '.weight'], p[prefix + '.bias'], True, 0.1, 1e-5)


def _stem(p, x, conv, bn, pool='max'):
    x = F.relu(_bn(p, bn, F.conv1d(x, p[conv], None, 2, 3)))
    return F.max_pool1d(x, 3, 2, 1) if pool == 'max' else F.avg_pool1d(x, 3, 2, 1)


def resnet18_block(p, x, prefix='breath_block.', first_pool_type='max'):
    """models/resnet.py:141-163 with BasicBlock :24-40."""
    h = _stem(p, x, prefix + 'conv1.weight', prefix + 'bn1', first_pool_type)
    inpl = 64
    for li, planes in enumerate((64, 128, 256, 512)):
        for bi in range(2):
            s = 2 if (li > 0 and bi == 0) else 1
            bp = '%slayer%d.%d.' % (prefix, li + 1, bi)
            o = F.relu(_bn(p, bp + 'bn1', F.conv1d(h, p[bp + 'conv1.weight'], None, s, 1)))
            o = _bn(p, bp + 'bn2', F.conv1d(o, p[bp + 'conv2.weight'], None, 1, 1))
            r = h
            if s != 1 or inpl != planes:
                r = _bn(p, bp + 'downsample.1', F.conv1d(h, p[bp + 'downsample.0.weight'], None, s, 0))
            h = F.relu(o + r)
            inpl = planes
    return F.avg_pool1d(h, 7, 1).flatten(1)


def densenet18_block(p, x, prefix='breath_block.', drop_rate=0.0):
    """models/densenet.py:179-189 with _DenseLayer :35-40 and _Transition :68-79."""
    fp = prefix + 'features.'
    h = _stem(p, x, fp + 'conv0.weight', fp + 'norm0')
    for bi in range(1, 5):
        for li in range(1, 3):
            lp = '%sdenseblock%d.denselayer%d.' % (fp, bi, li)
            o = F.conv1d(F.relu(_bn(p, lp + 'norm1', h)), p[lp + 'conv1.weight'])
            o = F.conv1d(F.relu(_bn(p, lp + 'norm2', o)), p[lp + 'conv2.weight'], None, 1, 1)
            if drop_rate > 0:
                o = F.dropout(o, drop_rate, True)
            h = torch.cat([h, o], 1)
        if bi != 4:
            tp = '%stransition%d.' % (fp, bi)
            h = F.avg_pool1d(F.conv1d(F.relu(_bn(p, tp + 'norm', h)), p[tp + 'conv.weight']), 2, 2)
    return F.avg_pool1d(F.relu(_bn(p, fp + 'norm5', h)), 7, 1).flatten(1)


def cnn_linear(p, x, backbone='resnet18', first_pool_type='max', drop_rate=0.0):
    """CNNLinearNetwork.forward: the reference's Python loop over windows + growing torch.cat."""
    if x.shape[-1] != 224:
        raise Exception('input breaths must have sequence length of 224')
    outs = None
    for i in range(x.shape[0]):
        if backbone == 'resnet18':
            f = resnet18_block(p, x[i], first_pool_type=first_pool_type)
        else:
            f = densenet18_block(p, x[i], drop_rate=drop_rate)
        o = F.linear(f.reshape(-1), p['linear_final.weight'], p['linear_final.bias']).unsqueeze(0)
        outs = o if outs is None else torch.cat([outs, o], dim=0)
    return outs


class CpuReferenceTrainer(object):
    """zero_grad -> forward -> BCEWithLogits -> backward (clamp hooks) -> SGD-Nesterov step."""

    def __init__(self, params, backbone='resnet18', lr=1e-3, wd=1e-4, clip=0.01, drop_rate=0.0, live=None):
        self.backbone = backbone
        self.drop_rate = drop_rate
        self.p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        for v in self.p.values():
            v.register_hook(lambda g: torch.clamp(g, -clip, clip))
        self.opt = torch.optim.SGD(list(self.p.values()), lr=lr, momentum=0.9, weight_decay=wd, nesterov=True)
        self.crit = torch.nn.BCEWithLogitsLoss()

    def step(self, x, target):
        self.opt.zero_grad()
        out = cnn_linear(self.p, x, self.backbone, drop_rate=self.drop_rate)
        loss = self.crit(out, target)
        loss.backward()
        self.opt.step()
        return loss.detach()
