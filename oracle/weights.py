"""Oracle helper (TEST INFRASTRUCTURE ONLY): parameter inventory + deterministic weights.

``param_spec`` lists the reference's ``state_dict`` parameter names and shapes for
``CNNLinearNetwork(resnet18|densenet18)`` (resnet.py:81-139, densenet.py:83-167,
torch_cnn_linear_network.py:97-102) -- verified against the real import by
``oracle/make_golden.py``.  ``seeded_params`` fills them from a name-keyed numpy RNG so that the
golden fixtures under ``tests/golden`` need not store 15 MB of weights: the same formula runs in
this container (against the reference) and on the GPU box (against the HIP path).
"""
import zlib
import numpy as np


def param_spec(backbone, n_sub_batches=20, in_ch=1, n_meta=0, head='linear', lstm_hidden=16):
    """-> list of (name, shape, kind) with kind in {'conv','bn_w','bn_b','lin_w','lin_b'},
    in the reference's ``named_parameters()`` order."""
    out = []

    def conv(name, co, ci, k):
        out.append((name + '.weight', (co, ci, k), 'conv'))

    def bn(name, c):
        out.append((name + '.weight', (c,), 'bn_w'))
        out.append((name + '.bias', (c,), 'bn_b'))

    p = 'breath_block.'
    layers = {'resnet18': (2, 2, 2, 2), 'resnet34': (3, 4, 6, 3)}.get(backbone)
    blocks = {'densenet18': (2, 2, 2, 2), 'densenet121': (6, 12, 24, 16), 'densenet169': (6, 12, 32, 32),
              'densenet201': (6, 12, 48, 32)}.get(backbone)
    if layers is not None:
        conv(p + 'conv1', 64, 1, 7)
        conv(p + 'conv1_alt', 64, 1, 3)      # dead unless double_conv_first (SURVEY finding 6)
        bn(p + 'bn1', 64)
        conv(p + 'conv2', 64, 64, 7)         # dead
        bn(p + 'bn2', 64)                    # dead
        inpl = 64
        for li, planes in enumerate([64, 128, 256, 512]):
            for bi in range(layers[li]):
                bp = '%slayer%d.%d.' % (p, li + 1, bi)
                stride = 2 if (li > 0 and bi == 0) else 1
                conv(bp + 'conv1', planes, inpl, 3)
                bn(bp + 'bn1', planes)
                conv(bp + 'conv2', planes, planes, 3)
                bn(bp + 'bn2', planes)
                if stride != 1 or inpl != planes:
                    conv(bp + 'downsample.0', planes, inpl, 1)
                    bn(bp + 'downsample.1', planes)
                inpl = planes
        feat = 512
    elif blocks is not None:
        fp = p + 'features.'
        conv(fp + 'conv0', 64, in_ch, 7)
        bn(fp + 'norm0', 64)
        nf = 64
        for bi in range(1, 5):
            for li in range(1, blocks[bi - 1] + 1):
                lp = '%sdenseblock%d.denselayer%d.' % (fp, bi, li)
                bn(lp + 'norm1', nf)
                conv(lp + 'conv1', 128, nf, 1)
                bn(lp + 'norm2', 128)
                conv(lp + 'conv2', 32, 128, 3)
                nf += 32
            if bi != 4:
                tp = '%stransition%d.' % (fp, bi)
                bn(tp + 'norm', nf)
                conv(tp + 'conv', nf // 2, nf, 1)
                nf //= 2
        bn(fp + 'norm5', nf)
        feat = nf
    else:
        raise ValueError(backbone)
    # heads of torch_cnn_linear_network.py: 'linear' (CNNLinearNetwork :92-103), 'to_mean' / 'compr_to_rf' /
    # 'single_breath' (Linear(F, 2), :7-67), 'double_linear' (Linear(F, 2) then Linear(2 NB + meta, 2), :70-89)
    if head == 'linear':
        out.append(('linear_final.weight', (2, feat * n_sub_batches + n_meta), 'lin_w'))
    elif head in ('to_mean', 'compr_to_rf', 'single_breath'):
        out.append(('linear_final.weight', (2, feat), 'lin_w'))
    elif head == 'lstm':          # CNNLSTMNetwork (torch_cnn_lstm_combo.py:6-50), no metadata: nn.LSTM(F, H) + Linear(H, 2)
        hdim = lstm_hidden
        out.append(('lstm.weight_ih_l0', (4 * hdim, feat), 'lin_w'))
        out.append(('lstm.weight_hh_l0', (4 * hdim, hdim), 'lin_w'))
        out.append(('lstm.bias_ih_l0', (4 * hdim,), 'lin_b'))
        out.append(('lstm.bias_hh_l0', (4 * hdim,), 'lin_b'))
        out.append(('linear_final.weight', (2, hdim), 'lin_w'))
    elif head == 'double_linear':
        out.append(('linear_intermediate.weight', (2, feat), 'lin_w'))
        out.append(('linear_intermediate.bias', (2,), 'lin_b'))
        out.append(('linear_final.weight', (2, 2 * n_sub_batches + n_meta), 'lin_w'))
    else:
        raise ValueError(head)
    out.append(('linear_final.bias', (2,), 'lin_b'))
    return out


DEAD_RESNET_PARAMS = ('breath_block.conv1_alt.weight', 'breath_block.conv2.weight',
                      'breath_block.bn2.weight', 'breath_block.bn2.bias')


def seeded_params(backbone, seed=0, n_sub_batches=20, dtype=np.float32, bn_bias_shift=0.0, head='linear', lstm_hidden=16,
                  in_ch=1):
    """Deterministic weights: conv ~ N(0, sqrt(2/(k*C_out))) as the reference's init
    (resnet.py:115-118, densenet.py:154-157); BN gamma ~ U(.5,1.5), beta ~ N(0,.1) (NOT the
    reference's 1/0 -- randomised so that parity tests see gamma/beta); linear ~ U(+-1/sqrt(in)).
    bn_bias_shift > 0 moves every BN beta up so that (almost) every ReLU is active: the network then has
    no activation decisions an fp32 rounding could flip, which makes whole-model GRADIENT parity a
    well-posed 1e-4 comparison (see tests/test_model_gpu.py)."""
    params = {}
    for name, shape, kind in param_spec(backbone, n_sub_batches, in_ch=in_ch, head=head, lstm_hidden=lstm_hidden):
        rng = np.random.default_rng([seed, zlib.crc32(name.encode())])
        if kind == 'conv':
            std = np.sqrt(2.0 / (shape[2] * shape[0]))
            a = rng.standard_normal(shape) * std
        elif kind == 'bn_w':
            a = rng.uniform(0.5, 1.5, shape)
        elif kind == 'bn_b':
            a = rng.standard_normal(shape) * 0.1 + bn_bias_shift
        elif kind == 'lin_w':
            bound = 1.0 / np.sqrt(shape[1])
            a = rng.uniform(-bound, bound, shape)
        else:
            a = rng.uniform(-0.01, 0.01, shape)
        params[name] = a.astype(dtype)
    return params


def seeded_batch(b, n_sub_batches=20, seed=0, kind='randn', dtype=np.float32):
    """Synthetic (B, NB, 1, 224) z-scored flow batch + one-hot targets.
    kind='randn' : N(0,1).  kind='flow': breath-like waveform (inspiratory square-ish
    pulse + exponential expiratory decay + noise) z-scored with the fixture's mu/std."""
    rng = np.random.default_rng([seed, b, 77])
    if kind == 'randn':
        x = rng.standard_normal((b, n_sub_batches, 1, 224))
    else:
        t = np.arange(224)[None, None, None, :]
        per = rng.uniform(60, 140, (b, n_sub_batches, 1, 1))
        ph = rng.uniform(0, 1, (b, n_sub_batches, 1, 1)) * per
        u = ((t + ph) % per) / per
        insp = (u < 0.33) * rng.uniform(20, 60, (b, n_sub_batches, 1, 1))
        exp_ = (u >= 0.33) * (-rng.uniform(20, 65, (b, n_sub_batches, 1, 1))) * np.exp(-(u - 0.33) * 6)
        x = insp + exp_ + rng.standard_normal((b, n_sub_batches, 1, 224)) * 0.8
        x = (x - 2.0560646853765587) / 28.08296533428954
    lab = rng.integers(0, 2, b)
    tgt = np.zeros((b, 2))
    tgt[np.arange(b), lab] = 1
    return x.astype(dtype), tgt.astype(dtype)


def digest(a, n=96):
    """Small deterministic digest of a tensor for the fixtures: the full tensor if tiny, else a
    strided sample followed by [sum, abs-sum, square-sum]."""
    f = np.asarray(a, dtype=np.float64).ravel()
    if f.size <= 1024:
        return f.copy()
    idx = np.linspace(0, f.size - 1, n).astype(np.int64)
    return np.concatenate([f[idx], [f.sum(), np.abs(f).sum(), np.square(f).sum()]])
