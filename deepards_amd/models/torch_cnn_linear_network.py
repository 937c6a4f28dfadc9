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


SEQ_LEN = 224                 # ARDSRawDataset.seq_len (dataset.py:345): the only row length the reference's heads accept


class _WindowHead(nn.Module):
    """What every head of the reference file shares: ``seq_size``, the ``breath_block`` child, a ``linear_final`` child
    (attribute and state_dict names are the contract), the 224-sample check and the batched breath-block pass."""

    def __init__(self, breath_block, head_inputs):
        nn.Module.__init__(self)
        self.seq_size = SEQ_LEN
        self.breath_block = breath_block
        self.linear_final = nn.Linear(head_inputs, 2)

    def _features(self, x):
        """x (B, NB, C, 224) -> (B, NB, features (B * NB, F)): all windows in one pass, BatchNorm still per window."""
        if x.shape[-1] != SEQ_LEN:
            raise Exception('input breaths must have sequence length of 224')        # the reference's message (:106-107)
        if x.shape[0] == 0:           # the reference indexes x[0] first (torch_cnn_linear_network.py:110)
            raise IndexError('index 0 is out of bounds for dimension 0 with size 0')
        b, nb, c, l = x.shape
        return b, nb, self.breath_block.forward_windows(x.reshape(b * nb, c, l), nb)

    def _head(self, flat, layer=None):
        layer = self.linear_final if layer is None else layer
        return F_.Linear2Function.apply(flat, layer.weight, layer.bias)

    @staticmethod
    def _no_metadata(metadata_features):
        if metadata_features:
            # the reference builds the head wider but never concatenates metadata -> shape error there too
            raise NotImplementedError('metadata_features > 0 is not runnable in the reference either '
                                      '(SURVEY.md finding 8)')


class CNNLinearNetwork(_WindowHead):
    def __init__(self, breath_block, sequence_size, metadata_features):
        _WindowHead.__init__(self, breath_block, breath_block.n_out_filters * sequence_size + metadata_features)
        self.n_sub_batches, self.metadata_features = sequence_size, metadata_features

    def forward(self, x, metadata):
        b, nb, feat = self._features(x)
        self._no_metadata(self.metadata_features)
        return self._head(feat.view(b, nb * feat.shape[1]))                           # == view(-1) per window

    def forward_loss(self, x, target):
        """(loss (1,), logits (B, 2)) of BCEWithLogitsLoss()(self(x, None), target) with the head chain -- global average
        pool, view(-1), linear_final, the loss and its first gradient -- as ONE autograd node in two launches
        (functional.HeadLossFunction); ``loss.backward()`` then runs the whole backward.  With gradients required the
        two returned tensors are FILLED BY THAT BACKWARD (its first kernel derives them from the forward's partial dot
        products): read them after it; under no_grad they are complete on return.  None when the breath block's last
        map is not the 7-position one the fused head pools (seq_len != 224 cannot happen here; a backbone without
        ``forward_windows(..., pooled=False)``): the caller then takes forward() + the loss kernels."""
        if x.shape[-1] != SEQ_LEN:
            raise Exception('input breaths must have sequence length of 224')
        if x.shape[0] == 0:
            raise IndexError('index 0 is out of bounds for dimension 0 with size 0')
        self._no_metadata(self.metadata_features)
        b, nb, c, l = x.shape
        try:       # ('fused': a backbone whose last block can pool for the head hands over (rows, F) features instead of the map)
            hmap = self.breath_block.forward_windows(x.reshape(b * nb, c, l), nb,
                                                     pooled='fused' if getattr(self.breath_block, 'fused_tail', False) else False)
        except TypeError:
            return None
        return F_.head_loss(hmap, self.linear_final.weight, self.linear_final.bias, target, nb)


def _windows(model, x):
    return model._features(x)


class CNNLinearToMean(_WindowHead):
    def __init__(self, breath_block):
        _WindowHead.__init__(self, breath_block, breath_block.n_out_filters)

    def forward(self, x, metadata):
        b, nb, feat = self._features(x)
        return self._head(F_.WindowMeanFunction.apply(feat, nb))


class CNNLinearComprToRF(_WindowHead):
    def __init__(self, breath_block):
        _WindowHead.__init__(self, breath_block, breath_block.n_out_filters)

    def forward(self, x, metadata):
        b, nb, feat = self._features(x)
        return self._head(F_.WindowMedianFunction.apply(feat, nb))


class CNNSingleBreathLinearNetwork(_WindowHead):
    def __init__(self, breath_block):
        _WindowHead.__init__(self, breath_block, breath_block.n_out_filters)

    def forward(self, x, metadata):
        b, nb, feat = self._features(x)
        return self._head(feat).view(b, nb, 2)


class CNNDoubleLinearNetwork(_WindowHead):
    def __init__(self, breath_block, sequence_size, metadata_features):
        nn.Module.__init__(self)
        self.seq_size, self.breath_block, self.metadata_features = SEQ_LEN, breath_block, metadata_features
        self.linear_intermediate = nn.Linear(breath_block.n_out_filters, 2)       # registered before linear_final (:79-80)
        self.linear_final = nn.Linear(2 * sequence_size + metadata_features, 2)

    def forward(self, x, metadata):
        self._no_metadata(self.metadata_features)
        b, nb, feat = self._features(x)
        inter = self._head(feat, self.linear_intermediate)
        return self._head(inter.view(b, nb * 2))


class CNNLSTMNetwork(_WindowHead):
    """reference models/torch_cnn_lstm_combo.py:6-50: breath block -> nn.LSTM over the NB breaths -> Linear(H, 2) per
    breath; returns (logits (B, NB, 2), (hx, cx)).  Metadata features are not on the accelerated path (NaN metadata =
    none, as the reference's default run)."""

    def __init__(self, breath_block, metadata_features, bm_to_linear, lstm_hidden_units):
        nn.Module.__init__(self)
        if metadata_features:
            raise NotImplementedError('metadata features are outside the accelerated path')
        if lstm_hidden_units % 8 or not 8 <= lstm_hidden_units <= 64:
            raise NotImplementedError('lstm_hidden_units must be a multiple of 8 in [8, 64] (defaults.yml: 16)')
        self.seq_size, self.breath_block = SEQ_LEN, breath_block
        self.lstm_hidden_units = lstm_hidden_units
        self.lstm_layers = 1
        self.bm_to_linear = bm_to_linear
        self.lstm = nn.LSTM(breath_block.n_out_filters, lstm_hidden_units, num_layers=1, batch_first=True)
        self.linear_final = nn.Linear(lstm_hidden_units, 2)

    def forward(self, x, metadata, hx_cx=None):
        b, nb, feat = self._features(x)
        h0 = c0 = None
        if hx_cx is not None:
            h0 = hx_cx[0].detach().reshape(b, -1).contiguous()
            c0 = hx_cx[1].detach().reshape(b, -1).contiguous()
        lstm = self.lstm
        hs, hx, cx = F_.LSTMFunction.apply(feat, lstm.weight_ih_l0, lstm.weight_hh_l0, lstm.bias_ih_l0, lstm.bias_hh_l0,
                                           nb, h0, c0)
        out = F_.Linear2Function.apply(hs.view(b * nb, -1), self.linear_final.weight, self.linear_final.bias)
        return out.view(b, nb, 2), (hx, cx)


class BreathBlockLinear(nn.Module):
    """NOT a class of the reference: the model around BASELINE configs[4]'s tile shape (resnet18, nb 40, seq_len 512),
    which the reference cannot run -- ``CNNLinearNetwork.forward`` refuses anything but 224 samples
    (torch_cnn_linear_network.py:106-107) and never concatenates the 9 metadata inputs its head is sized for (SURVEY
    finding 8).  STATED, not mirrored: the breath block on (B * NB, 1, L) rows with per-window BatchNorm exactly as
    ``breath_block(x[i])`` would see them, its features (``AvgPool1d(7, 1)`` leaves L / 32 - 6 positions per channel,
    flattened channel-major like ``view(N, -1)``, resnet.py:159-160) flattened per window like ``view(-1)``
    (:110-112), one ``Linear(F * NB, 2)`` head, no metadata.  Used by ``bench.py --nb 40 --seq-len 512`` and by the
    model-level parity tests of that shape."""

    def __init__(self, breath_block, nb, seq_len):
        nn.Module.__init__(self)
        if seq_len % 32 or seq_len < 224:
            raise ValueError('seq_len must be a multiple of 32, at least 224')
        self.breath_block, self.nb, self.seq_size = breath_block, nb, seq_len
        self.linear_final = nn.Linear(breath_block.n_out_filters * (seq_len // 32 - 6) * nb, 2)

    def forward(self, x, metadata):
        b, nb, c, l = x.shape
        if l != self.seq_size or nb != self.nb:
            raise Exception('input breaths must have sequence length of %d in windows of %d' % (self.seq_size, self.nb))
        feat = self.breath_block.forward_windows(x.reshape(b * nb, c, l), nb)
        return F_.Linear2Function.apply(feat.view(b, -1), self.linear_final.weight, self.linear_final.bias)
