"""Goldens for the sibling heads of CNNLinearNetwork from the REAL reference classes
(torch_cnn_linear_network.py:7-89), run in the build container only:

    python oracle/make_golden_heads.py        # needs /root/reference; never runs on the GPU box

Writes tests/golden/head_<head>_<backbone>_b2[_active].npz: inputs, fp64 / fp32 logits, loss, and digests of every
parameter gradient (loss as the reference's calc_loss: per-breath outputs repeat the target, train_ards_detector.py:
540-543).  `_active` = every ReLU active (BN beta + 6, avg first pool): gradients without flippable decisions.
"""
import os
import sys
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, '/root/reference')

from oracle.weights import param_spec, seeded_params, seeded_batch, digest as sample        # noqa: E402
from deepards.models.resnet import resnet18                                                 # noqa: E402
from deepards.models.densenet import densenet18                                             # noqa: E402
from deepards.models import torch_cnn_linear_network as ref                                 # noqa: E402
from deepards.models.torch_cnn_lstm_combo import CNNLSTMNetwork                             # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
HEADS = {'to_mean': lambda bb: ref.CNNLinearToMean(bb), 'compr_to_rf': lambda bb: ref.CNNLinearComprToRF(bb),
         'single_breath': lambda bb: ref.CNNSingleBreathLinearNetwork(bb),
         'double_linear': lambda bb: ref.CNNDoubleLinearNetwork(bb, 20, 0),
         'lstm': lambda bb: CNNLSTMNetwork(bb, 0, False, 16)}       # defaults.yml:35 time_series_hidden_units = 16


def build(head, backbone, seed, dtype, shift):
    bb = resnet18(first_pool_type='avg' if shift else 'max') if backbone == 'resnet18' else densenet18(drop_rate=0)
    model = HEADS[head](bb)
    spec = param_spec(backbone, head=head)
    assert [n for n, _ in model.named_parameters()] == [s[0] for s in spec], 'param_spec order differs'
    sd = {k: torch.from_numpy(v) for k, v in seeded_params(backbone, seed, bn_bias_shift=shift, head=head).items()}
    missing = model.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys and all(('running_' in k or 'num_batches' in k) for k in missing.missing_keys)
    return model.to(dtype).train()


def main():
    for head in HEADS:
        for backbone in ('resnet18', 'densenet18'):
            for shift in (0.0, 6.0):
                seed, b = 11, 2
                x, tgt = seeded_batch(b, 20, seed, 'randn')
                rec = dict(x=x, target=tgt, backbone=backbone, head=head, seed=seed, b=b, bn_bias_shift=shift,
                           first_pool_type='avg' if (shift and backbone == 'resnet18') else 'max')
                for dt, sfx in ((torch.float64, '64'), (torch.float32, '32')):
                    model = build(head, backbone, seed, dt, shift)
                    xt, tt = torch.from_numpy(x).to(dt), torch.from_numpy(tgt).to(dt)
                    if head == 'lstm':       # zero initial state, NaN metadata (= none), train_ards_detector.py:846-847
                        out, (hx, cx) = model(xt, torch.full((b,), float('nan'), dtype=dt), None)
                        rec['hx' + sfx] = hx.detach().numpy().astype(np.float64)
                        rec['cx' + sfx] = cx.detach().numpy().astype(np.float64)
                    else:
                        out = model(xt, None)
                    tl = tt.unsqueeze(1).repeat((1, out.shape[1], 1)) if out.dim() == 3 else tt
                    loss = torch.nn.BCEWithLogitsLoss()(out, tl)
                    loss.backward()
                    rec['logits' + sfx] = out.detach().numpy().astype(np.float64)
                    rec['loss' + sfx] = float(loss)
                    for n, p in model.named_parameters():
                        if p.grad is not None:
                            rec['grad%s/%s' % (sfx, n)] = sample(p.grad.numpy())
                path = os.path.join(OUT, 'head_%s_%s_b2%s.npz' % (head, backbone, '_active' if shift else ''))
                np.savez_compressed(path, **rec)
                print(path, os.path.getsize(path), 'loss', rec['loss64'])


if __name__ == '__main__':
    main()
