"""Golden for the explainer surface: the forward/backward that reference gradcam.py:38-108
(CamExtractor.forward_pass + GradCam._generate_grad_and_output) performs on a cnn_linear + densenet18 model,
restated on the REAL reference modules (gradcam.py itself needs cv2 / matplotlib, absent here):

    conv_output = model.breath_block.features(x); x.register_hook(save);  relu -> avgpool -> view(-1) -> linear_final
    one-hot(target) * output -> backward;  guided gradients = d output[target] / d conv_output

    python oracle/make_golden_gradcam.py      # build container only
"""
import os
import sys
import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, '/root/reference')

from oracle.weights import seeded_params, seeded_batch                                  # noqa: E402
from deepards.models.densenet import densenet18                                         # noqa: E402
from deepards.models.torch_cnn_linear_network import CNNLinearNetwork                   # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden', 'gradcam_densenet18.npz')


def main():
    seed = 4
    x, _ = seeded_batch(1, 20, seed, 'flow')
    rec = dict(x=x[0], seed=seed)
    for dt, sfx in ((torch.float64, '64'), (torch.float32, '32')):
        model = CNNLinearNetwork(densenet18(drop_rate=0), 20, 0)
        sd = {k: torch.from_numpy(v) for k, v in seeded_params('densenet18', seed).items()}
        model.load_state_dict(sd, strict=False)
        model = model.to(dt).train()
        grads = {}
        xt = torch.from_numpy(x[0]).to(dt)
        conv = model.breath_block.features(xt)
        conv.register_hook(lambda g: grads.__setitem__('g', g))
        h = F.relu(conv)
        h = model.breath_block.avgpool(h).view(-1)
        out = model.linear_final(h).unsqueeze(0)
        target = int(np.argmax(out.detach().numpy()))
        one_hot = torch.zeros((1, 2), dtype=dt)
        one_hot[0, target] = 1
        model.zero_grad()
        torch.sum(one_hot * out).backward()
        rec['conv' + sfx] = conv.detach().numpy().astype(np.float64)
        rec['grad' + sfx] = grads['g'].numpy().astype(np.float64)
        rec['out' + sfx] = out.detach().numpy().astype(np.float64)
        rec['target' + sfx] = target
    np.savez_compressed(OUT, **rec)
    print(OUT, os.path.getsize(OUT), rec['out64'], rec['conv64'].shape)


if __name__ == '__main__':
    main()
