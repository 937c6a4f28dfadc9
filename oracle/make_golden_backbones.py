"""Goldens for the other base networks that run on the same kernels (reference models/resnet.py:178 resnet34,
models/densenet.py:234 densenet121), from the REAL reference, build container only:

    python oracle/make_golden_backbones.py

tests/golden/bb_<name>_b2.npz: inputs, fp64 / fp32 logits, loss, digests of every parameter gradient (cnn_linear head).
"""
import os
import sys
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, '/root/reference')

from oracle.weights import param_spec, seeded_params, seeded_batch, digest as sample        # noqa: E402
from deepards.models.resnet import resnet34                                                 # noqa: E402
from deepards.models.densenet import densenet121                                            # noqa: E402
from deepards.models.torch_cnn_linear_network import CNNLinearNetwork                       # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')


def main():
    for name, ctor in (('resnet34', lambda: resnet34()), ('densenet121', lambda: densenet121(drop_rate=0))):
        seed, b = 21, 2
        x, tgt = seeded_batch(b, 20, seed, 'randn')
        rec = dict(x=x, target=tgt, backbone=name, seed=seed, b=b, bn_bias_shift=0.0, first_pool_type='max')
        for dt, sfx in ((torch.float64, '64'), (torch.float32, '32')):
            model = CNNLinearNetwork(ctor(), 20, 0)
            spec = param_spec(name)
            assert [n for n, _ in model.named_parameters()] == [s[0] for s in spec], 'param_spec order differs'
            sd = {k: torch.from_numpy(v) for k, v in seeded_params(name, seed).items()}
            missing = model.load_state_dict(sd, strict=False)
            assert not missing.unexpected_keys
            model = model.to(dt).train()
            xt, tt = torch.from_numpy(x).to(dt), torch.from_numpy(tgt).to(dt)
            out = model(xt, None)
            loss = torch.nn.BCEWithLogitsLoss()(out, tt)
            loss.backward()
            rec['logits' + sfx] = out.detach().numpy().astype(np.float64)
            rec['loss' + sfx] = float(loss.detach())
            if sfx == '64':
                for n, p in model.named_parameters():
                    if p.grad is not None:
                        rec['grad64/' + n] = sample(p.grad.numpy(), 24)
        path = os.path.join(OUT, 'bb_%s_b2.npz' % name)
        np.savez_compressed(path, **rec)
        print(path, os.path.getsize(path), rec['loss64'], rec['logits64'].ravel())


if __name__ == '__main__':
    main()
