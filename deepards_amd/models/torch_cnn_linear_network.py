"""CNNLinearNetwork on MI355X.

Operator surface of reference ``deepards/models/torch_cnn_linear_network.py:92-113``: same
constructor, ``breath_block`` / ``linear_final`` / ``seq_size`` attributes, same exception for a
wrong sequence length, same output ``(B, 2)``.  Instead of looping over the batch in Python and
``torch.cat``-ing the per-window logits, all B windows go through the breath block in one batched
pass (BatchNorm statistics still per window) and one head kernel.
"""
import torch.nn as nn

from .. import functional as F_


class CNNLinearNetwork(nn.Module):
    def __init__(self, breath_block, sequence_size, metadata_features):
        super(CNNLinearNetwork, self).__init__()
        self.seq_size = 224
        self.breath_block = breath_block
        self.n_sub_batches = sequence_size
        self.metadata_features = metadata_features
        self.linear_final = nn.Linear(self.breath_block.n_out_filters * sequence_size + metadata_features, 2)

    def forward(self, x, metadata):
        # input should be in shape: (batches, breaths in seq, chans, 224)
        if x.shape[-1] != 224:
            raise Exception('input breaths must have sequence length of 224')
        if self.metadata_features:
            # the reference builds the head wider but never concatenates metadata -> shape error there too
            raise NotImplementedError('metadata_features > 0 is not runnable in the reference either '
                                      '(SURVEY.md finding 8)')
        b, nb, c, l = x.shape
        feat = self.breath_block.forward_windows(x.reshape(b * nb, c, l), nb)        # (B*NB, F)
        flat = feat.view(b, nb * feat.shape[1])                                     # == view(-1) per window
        return F_.Linear2Function.apply(flat, self.linear_final.weight, self.linear_final.bias)
