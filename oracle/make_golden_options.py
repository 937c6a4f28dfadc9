"""Goldens for the constructor OPTIONS of the drop-in boundary, from the REAL reference (build container only):

    python oracle/make_golden_options.py        # needs /root/reference; never runs on the GPU box

* ``densenet18(with_fft / only_fft / fft_real_only)`` (models/densenet.py:109-115: conv0 gets 3 / 2 / 2 / 1 input
  channels) inside ``CNNLinearNetwork`` on (B, NB, C, 224) inputs whose extra channels are the spectrum channels the
  dataset appends (dataset.py:1330-1341; built here with ``deepards_amd.tiles.perform_fft`` -- the reference's
  dataset.py does not import in this container, and for a model golden the input only has to be SOME (C, 224) signal);
* ``resnet18(double_conv_first=True)`` (models/resnet.py:90-96,142-149: conv1_alt -> bn1 -> conv2 -> bn2, the four
  parameters that are dead in the default stem become live and conv1 dies instead).

Recorded per case: inputs, fp64 / fp32 logits and loss, per-parameter gradient digests, a 3-step SGD-Nesterov
trajectory with the clamp hooks (losses + parameter digests).  Numbers only; nothing of the reference is copied.
"""
import os
import sys
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, '/root/reference')

from oracle.weights import param_spec, seeded_params, seeded_batch, digest as sample      # noqa: E402
from deepards.models.resnet import resnet18                                               # noqa: E402
from deepards.models.densenet import densenet18                                           # noqa: E402
from deepards.models.torch_cnn_linear_network import CNNLinearNetwork                     # noqa: E402
from deepards_amd.tiles import perform_fft                                                # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
CLIP, LR, MOM, WD = 0.01, 1e-3, 0.9, 1e-4


def build(backbone, kwargs, seed, dtype, shift, in_ch):
    bb = resnet18(**kwargs) if backbone == 'resnet18' else densenet18(drop_rate=0, **kwargs)
    model = CNNLinearNetwork(bb, 20, 0)
    spec = param_spec(backbone, in_ch=in_ch)
    assert [n for n, _ in model.named_parameters()] == [s[0] for s in spec]
    for (n, shp, _), (_, p) in zip(spec, model.named_parameters()):
        assert tuple(p.shape) == tuple(shp), (n, p.shape, shp)
    sd = {k: torch.from_numpy(v) for k, v in seeded_params(backbone, seed, bn_bias_shift=shift, in_ch=in_ch).items()}
    assert not model.load_state_dict(sd, strict=False).unexpected_keys
    return model.to(dtype).train()


def inputs(b, seed, fft):
    x, tgt = seeded_batch(b, 20, seed, 'flow')
    if fft:
        x = perform_fft(x.astype(np.float64), **fft)
        mu = x.mean(axis=(0, 1, 3), keepdims=True)
        sd = x.std(axis=(0, 1, 3), keepdims=True)
        x = ((x - mu) / sd).astype(np.float32)           # per-channel z-score, as the dataset's factors would
    return x, tgt


def run_case(tag, backbone, kwargs, b, seed, fft=None, shift=0.0):
    x, tgt = inputs(b, seed, fft)
    in_ch = x.shape[2]
    rec = dict(x=x, target=tgt, backbone=backbone, seed=seed, b=b, bn_bias_shift=shift, in_ch=in_ch,
               first_pool_type=str(kwargs.get('first_pool_type', 'max')),
               **{'opt_' + k: int(bool(v)) for k, v in kwargs.items() if k != 'first_pool_type'})
    for dt, sfx in ((torch.float64, '64'), (torch.float32, '32')):
        model = build(backbone, kwargs, seed, dt, shift, in_ch)
        xt, tt = torch.from_numpy(x).to(dt), torch.from_numpy(tgt).to(dt)
        out = model(xt, None)
        loss = torch.nn.BCEWithLogitsLoss()(out, tt)
        loss.backward()
        rec['logits' + sfx] = out.detach().numpy().astype(np.float64)
        rec['loss' + sfx] = float(loss)
        for n, p in model.named_parameters():
            if p.grad is not None:
                rec['grad%s/%s' % (sfx, n)] = sample(p.grad.numpy())
        model = build(backbone, kwargs, seed, dt, shift, in_ch)
        for p in model.parameters():
            p.register_hook(lambda g: torch.clamp(g, -CLIP, CLIP))
        opt = torch.optim.SGD(model.parameters(), lr=LR, momentum=MOM, weight_decay=WD, nesterov=True)
        losses = []
        for step in range(3):
            model.zero_grad()
            loss = torch.nn.BCEWithLogitsLoss()(model(xt, None), tt)
            loss.backward()
            opt.step()
            losses.append(float(loss))
        rec['sgd_losses' + sfx] = np.array(losses)
        live = {n for n, p in model.named_parameters() if p.grad is not None}
        for n, p in model.named_parameters():
            if n in live:
                rec['sgd_p%s/%s' % (sfx, n)] = sample(p.detach().numpy())
    path = os.path.join(OUT, tag + '.npz')
    np.savez_compressed(path, **rec)
    print(tag, 'in_ch', in_ch, 'logits64', rec['logits64'].ravel(), 'loss', rec['loss64'],
          '|l32-l64|max', np.abs(rec['logits32'] - rec['logits64']).max(), os.path.getsize(path), 'B')


if __name__ == '__main__':
    torch.manual_seed(0)
    torch.set_num_threads(8)
    run_case('opt_densenet18_with_fft_b2', 'densenet18', dict(with_fft=True), 2, 21, fft=dict(add_fft=True))
    run_case('opt_densenet18_only_fft_b2', 'densenet18', dict(only_fft=True), 2, 22, fft=dict(only_fft=True))
    run_case('opt_densenet18_with_fft_real_only_b2', 'densenet18', dict(with_fft=True, fft_real_only=True), 2, 23,
             fft=dict(add_fft=True, fft_real_only=True))
    run_case('opt_densenet18_only_fft_real_only_b2', 'densenet18', dict(only_fft=True, fft_real_only=True), 2, 24,
             fft=dict(only_fft=True, fft_real_only=True))
    run_case('opt_densenet18_with_fft_b2_active', 'densenet18', dict(with_fft=True), 2, 25, fft=dict(add_fft=True), shift=6.0)
    run_case('opt_resnet18_double_conv_b2', 'resnet18', dict(double_conv_first=True), 2, 26)
    run_case('opt_resnet18_double_conv_b2_active', 'resnet18', dict(double_conv_first=True, first_pool_type='avg'), 2, 27,
             shift=6.0)
