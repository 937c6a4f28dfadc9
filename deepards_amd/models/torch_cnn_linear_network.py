"""CNNLinearNetwork and its sibling heads on MI355X.

Operator surface of reference ``deepards/models/torch_cnn_linear_network.py``: same constructors,
``breath_block`` / ``linear_final`` (/ ``linear_intermediate``) / ``seq_size`` attributes, same exception for a
wrong sequence length, same output shapes.  Instead of looping over the batch in Python and ``torch.cat``-ing
the per-window results, all B windows go through the breath block in one batched pass (BatchNorm statistics
still per window) and one head kernel.

    CNNLinearNetwork              :92-113   Linear(F*NB, 2) on the flattened (NB, F) block        -> (B, 2)
    CNNLinearToMean               :7-26     Linear(F, 2) on the mean over the NB breaths          -> (B, 2)
    CNNLinearComprToRF            :29-47    Linear(F, 2) on the (lower) median over the breaths   -> (B, 2)
    CNNSingleBreathLinearNetwork  :50-67    Linear(F, 2) per breath                               -> (B, NB, 2)
    CNNDoubleLinearNetwork        :70-89    Linear(2*NB, 2) on the flattened per-breath Linear(F, 2) -> (B, 2)
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
        if x.shape[0] == 0:           # the reference indexes x[0] first (torch_cnn_linear_network.py:110)
            raise IndexError('index 0 is out of bounds for dimension 0 with size 0')
        if self.metadata_features:
            # the reference builds the head wider but never concatenates metadata -> shape error there too
            raise NotImplementedError('metadata_features > 0 is not runnable in the reference either '
                                      '(SURVEY.md finding 8)')
        b, nb, c, l = x.shape
        feat = self.breath_block.forward_windows(x.reshape(b * nb, c, l), nb)        # (B*NB, F)
        flat = feat.view(b, nb * feat.shape[1])                                     # == view(-1) per window
        return F_.Linear2Function.apply(flat, self.linear_final.weight, self.linear_final.bias)


def _windows(model, x):
    # input should be in shape: (batches, breaths in seq, chans, 224)
    if x.shape[-1] != 224:
        raise Exception('input breaths must have sequence length of 224')
    if x.shape[0] == 0:           # the reference indexes x[0] first (torch_cnn_linear_network.py:110)
        raise IndexError('index 0 is out of bounds for dimension 0 with size 0')
    b, nb, c, l = x.shape
    return b, nb, model.breath_block.forward_windows(x.reshape(b * nb, c, l), nb)    # (B*NB, F)


class CNNLinearToMean(nn.Module):
    def __init__(self, breath_block):
        super(CNNLinearToMean, self).__init__()
        self.seq_size = 224
        self.breath_block = breath_block
        self.linear_final = nn.Linear(self.breath_block.n_out_filters, 2)

    def forward(self, x, metadata):
        b, nb, feat = _windows(self, x)
        return F_.Linear2Function.apply(F_.WindowMeanFunction.apply(feat, nb), self.linear_final.weight,
                                        self.linear_final.bias)


class CNNLinearComprToRF(nn.Module):
    def __init__(self, breath_block):
        super(CNNLinearComprToRF, self).__init__()
        self.seq_size = 224
        self.breath_block = breath_block
        self.linear_final = nn.Linear(self.breath_block.n_out_filters, 2)

    def forward(self, x, metadata):
        b, nb, feat = _windows(self, x)
        return F_.Linear2Function.apply(F_.WindowMedianFunction.apply(feat, nb), self.linear_final.weight,
                                        self.linear_final.bias)


class CNNSingleBreathLinearNetwork(nn.Module):
    def __init__(self, breath_block):
        super(CNNSingleBreathLinearNetwork, self).__init__()
        self.seq_size = 224
        self.breath_block = breath_block
        self.linear_final = nn.Linear(self.breath_block.n_out_filters, 2)

    def forward(self, x, metadata):
        b, nb, feat = _windows(self, x)
        return F_.Linear2Function.apply(feat, self.linear_final.weight, self.linear_final.bias).view(b, nb, 2)


class CNNDoubleLinearNetwork(nn.Module):
    def __init__(self, breath_block, sequence_size, metadata_features):
        super(CNNDoubleLinearNetwork, self).__init__()
        self.seq_size = 224
        self.breath_block = breath_block
        self.metadata_features = metadata_features
        self.linear_intermediate = nn.Linear(self.breath_block.n_out_filters, 2)
        self.linear_final = nn.Linear(2 * sequence_size + metadata_features, 2)

    def forward(self, x, metadata):
        if self.metadata_features:
            raise NotImplementedError('metadata_features > 0 is not runnable in the reference either '
                                      '(SURVEY.md finding 8)')
        b, nb, feat = _windows(self, x)
        inter = F_.Linear2Function.apply(feat, self.linear_intermediate.weight, self.linear_intermediate.bias)
        return F_.Linear2Function.apply(inter.view(b, nb * 2), self.linear_final.weight, self.linear_final.bias)


class CNNLSTMNetwork(nn.Module):
    """reference models/torch_cnn_lstm_combo.py:6-50: breath block -> nn.LSTM over the NB breaths -> Linear(H, 2) per
    breath; returns (logits (B, NB, 2), (hx, cx)).  Metadata features are not on the accelerated path (NaN metadata =
    none, as the reference's default run)."""

    def __init__(self, breath_block, metadata_features, bm_to_linear, lstm_hidden_units):
        super(CNNLSTMNetwork, self).__init__()
        if metadata_features:
            raise NotImplementedError('metadata features are outside the accelerated path')
        if lstm_hidden_units % 8 or not 8 <= lstm_hidden_units <= 64:
            raise NotImplementedError('lstm_hidden_units must be a multiple of 8 in [8, 64] (defaults.yml: 16)')
        self.seq_size = 224
        self.breath_block = breath_block
        self.lstm_hidden_units = lstm_hidden_units
        self.lstm_layers = 1
        self.bm_to_linear = bm_to_linear
        self.lstm = nn.LSTM(breath_block.n_out_filters, lstm_hidden_units, num_layers=1, batch_first=True)
        self.linear_final = nn.Linear(lstm_hidden_units, 2)

    def forward(self, x, metadata, hx_cx=None):
        b, nb, feat = _windows(self, x)
        h0 = c0 = None
        if hx_cx is not None:
            h0 = hx_cx[0].detach().reshape(b, -1).contiguous()
            c0 = hx_cx[1].detach().reshape(b, -1).contiguous()
        lstm = self.lstm
        hs, hx, cx = F_.LSTMFunction.apply(feat, lstm.weight_ih_l0, lstm.weight_hh_l0, lstm.bias_ih_l0, lstm.bias_hh_l0,
                                           nb, h0, c0)
        out = F_.Linear2Function.apply(hs.view(b * nb, -1), self.linear_final.weight, self.linear_final.bias)
        return out.view(b, nb, 2), (hx, cx)
