"""1-D DenseNet-BC (densenet18: block_config (2,2,2,2), growth 32, bn_size 4) on MI355X.

Operator surface of reference ``deepards/models/densenet.py`` (DenseNet :83-194, _DenseLayer :18-44,
_DenseBlock :46-66, _Transition :68-81, densenet18 :223-231): ``features`` Sequential with the same
module names (state_dict keys match), ``avgpool``, ``n_out_filters``, ``network_name``,
``forward_no_pool`` and ``conv_info()``.  BatchNorms are built with track_running_stats=False and
dropout stays active whenever the module is in training mode -- which is always on this path
(SURVEY.md finding 4).  Compute runs in the HIP kernels via ``deepards_amd.functional``.
"""
from collections import OrderedDict

import torch
import torch.nn as nn

from .. import functional as F_
from .resnet import _require_cuda, conv1d, init_like_reference


def _bn(c):
    return nn.BatchNorm1d(c, track_running_stats=False)        # densenet.py:107: always batch statistics


class _DenseLayer(nn.Sequential):
    """norm1 -> relu1 -> conv1 (1x1, to bn_size * growth) -> norm2 -> relu2 -> conv2 (k3, to growth) -> dropout -> cat:
    the child names are the state_dict contract (reference models/densenet.py:18-33)."""
    num_layers = 2

    def __init__(self, num_input_features, growth_rate, bn_size, drop_rate, track_running_stats=False):
        super(_DenseLayer, self).__init__()
        mid = bn_size * growth_rate
        for name, mod in (('norm1', _bn(num_input_features)), ('relu1', nn.ReLU(inplace=True)),
                          ('conv1', conv1d(num_input_features, mid, 1)), ('norm2', _bn(mid)), ('relu2', nn.ReLU(inplace=True)),
                          ('conv2', conv1d(mid, growth_rate, 3))):
            self.add_module(name, mod)
        self.drop_rate = float(drop_rate)

    def conv_info(self):
        return [1, 3], [1, 1], [0, 1]

    def forward_rlc(self, x, R, seed, salt):
        p = self.drop_rate if self.training else 0.0
        return F_.DenseLayerFunction.apply(
            x, self.norm1.weight, self.norm1.bias, self.conv1.weight,
            self.norm2.weight, self.norm2.bias, self.conv2.weight,
            R, F_.BNState(self.norm1), F_.BNState(self.norm2), p, seed if p > 0 else None, salt)


class _DenseBlock(nn.Sequential):
    def __init__(self, num_layers, num_input_features, bn_size, growth_rate, drop_rate, track_running_stats=False):
        super(_DenseBlock, self).__init__()
        self.track_running_stats = track_running_stats
        for i in range(num_layers):
            self.add_module('denselayer%d' % (i + 1),
                            _DenseLayer(num_input_features + i * growth_rate, growth_rate, bn_size, drop_rate))
        self.num_layers = _DenseLayer.num_layers * num_layers

    def conv_info(self):
        """(kernel sizes, strides, paddings) of the block's convs in order (protopnet's receptive-field bookkeeping)."""
        ks, st, pd = [], [], []
        for layer in self.children():
            a, b, c = layer.conv_info()
            ks, st, pd = ks + a, st + b, pd + c
        return ks, st, pd

    @property
    def kernel_sizes(self):
        return self.conv_info()[0]

    @property
    def strides(self):
        return self.conv_info()[1]

    @property
    def paddings(self):
        return self.conv_info()[2]


class _Transition(nn.Sequential):
    """norm -> relu -> conv (1x1, halves the channels) -> pool (AvgPool1d(2, 2)); reference models/densenet.py:68-79."""
    num_layers = 1

    def __init__(self, num_input_features, num_output_features, track_running_stats=False):
        super(_Transition, self).__init__()
        for name, mod in (('norm', _bn(num_input_features)), ('relu', nn.ReLU(inplace=True)),
                          ('conv', conv1d(num_input_features, num_output_features, 1)),
                          ('pool', nn.AvgPool1d(kernel_size=2, stride=2))):
            self.add_module(name, mod)

    def conv_info(self):
        return [1, 2], [1, 2], [0, 0]

    def forward_rlc(self, x, R):
        return F_.TransitionFunction.apply(x, self.norm.weight, self.norm.bias, self.conv.weight, R,
                                           F_.BNState(self.norm))


class _Features(nn.Sequential):
    """``DenseNet.features``: the reference's Sequential (same child names / state_dict keys); calling it runs the
    HIP pipeline and returns the norm5 output in the reference's (N, C, L) layout WITHOUT the final ReLU, as
    ``nn.Sequential.__call__`` does there -- what gradcam.py:45 hooks."""
    drop_rate = 0.0

    def forward_rlc(self, x, R, relu):
        _require_cuda(x, 'DenseNet')
        cin = self.conv0.in_channels                       # 1, or 2 / 3 with the FFT channels (densenet.py:109-115)
        if x.dim() != 3 or x.shape[1] != cin:
            raise ValueError('expected (rows, %d, L) input, got %s' % (cin, tuple(x.shape)))
        rows, _, l = x.shape
        if rows % R:
            raise ValueError('rows not a multiple of rows_per_window')
        x2d = x.contiguous().float()                       # (rows, C_in, L): the stem kernel reads the NCL rows directly
        use_drop = self.training and self.drop_rate > 0
        plan = self._block_plan(rows, R, l, use_drop)
        if plan is not None:
            return self._forward_blocks(x2d, R, relu, use_drop, plan)
        h = F_.StemFunction.apply(x2d, self.conv0.weight, self.norm0.weight, self.norm0.bias, R, F_.POOL_MAX,
                                  F_.BNState(self.norm0))
        if use_drop:
            self._drop_seed.add_(0x9E3779B97F4A7C15 >> 1)
        salt = 0
        for name, mod in self.named_children():
            if isinstance(mod, _DenseBlock):
                for layer in mod.children():
                    salt += 1
                    h = layer.forward_rlc(h, R, self._drop_seed, salt)
            elif isinstance(mod, _Transition):
                h = mod.forward_rlc(h, R)
        return F_.NormReluFunction.apply(h, self.norm5.weight, self.norm5.bias, R, F_.BNState(self.norm5), relu)

    def _block_plan(self, rows, R, l, use_drop):
        """[(block, its transition or None, length, C0, Cb)] when EVERY dense block of the network can run as one
        F_.DenseBlockFunction (one pitched buffer + one statistics table per block, see there), else None: the per-layer
        Functions below then take the whole network (other shapes, bf16 convs, DA_DENSE_BLOCK=0)."""
        mods = list(self.children())
        lb = ((l + 2 * 3 - 7) // 2 + 1 - 1) // 2 + 1            # conv0 k7 s2 p3, pool0 (3, 2, 1)
        c0 = self.conv0.out_channels
        plan = []
        for i, mod in enumerate(mods):
            if not isinstance(mod, _DenseBlock):
                continue
            layers = list(mod.children())
            growth, mid = layers[0].conv2.out_channels, layers[0].conv1.out_channels
            tail = mods[i + 1] if i + 1 < len(mods) and isinstance(mods[i + 1], _Transition) else None
            cb = c0 + len(layers) * growth
            if any(ly.conv2.out_channels != growth or ly.conv1.out_channels != mid for ly in layers) or \
                    not F_.dense_block_ok(rows, R, lb, c0, growth, len(layers), mid, tail.conv.out_channels if tail else 0,
                                          use_drop):
                return None
            plan.append((mod, tail, lb, c0, cb))
            if tail is not None:
                c0, lb = tail.conv.out_channels, lb // 2
        return plan or None

    def _forward_blocks(self, x2d, R, relu, use_drop, plan):
        """The network as one Function per dense block (+ its transition / norm5), F_.DenseBlockFunction."""
        h = F_.StemFunction.apply(x2d, self.conv0.weight, self.norm0.weight, self.norm0.bias, R, F_.POOL_MAX,
                                  F_.BNState(self.norm0), False, plan[0][4])
        if use_drop:
            self._drop_seed.add_(0x9E3779B97F4A7C15 >> 1)
        salt = 1
        p = self.drop_rate if use_drop else 0.0
        rec = None                                     # statistics records of the buffer's first channels (from a transition)
        for n, (blk, tail, lb, c0, cb) in enumerate(plan):
            layers = list(blk.children())
            params = []
            for ly in layers:
                params += [ly.norm1.weight, ly.norm1.bias, ly.conv1.weight, ly.norm2.weight, ly.norm2.bias, ly.conv2.weight]
            if tail is not None:
                params += [tail.norm.weight, tail.norm.bias, tail.conv.weight]
                tail_cb, eps = plan[n + 1][4], tail.norm.eps
            else:
                params += [self.norm5.weight, self.norm5.bias]
                tail_cb, eps = 0, self.norm5.eps
            h = F_.DenseBlockFunction.apply(h, rec, R, c0, layers[0].conv2.out_channels, len(layers), p,
                                            self._drop_seed if use_drop else None, salt, eps, tail_cb, relu, *params)
            if tail is not None:
                h, rec = h
            salt += len(layers)
        return h

    def forward(self, x):
        return self.forward_rlc(x, x.shape[0], False).permute(0, 2, 1)


class DenseNet(nn.Module):
    """DenseNet-BC on 1-D windows: stem (conv0 k7 s2, norm0, relu0, pool0), `block_config` dense blocks of growth-rate
    layers with halving transitions between them, norm5.  Module names / order / shapes are the reference's
    (densenet.py:83-166): 64 state_dict keys for cnn_linear + densenet18."""

    def __init__(self, growth_rate=32, block_config=(6, 12, 24, 16), num_init_features=64, bn_size=4,
                 drop_rate=0.2, num_classes=1000, with_fft=False, only_fft=False, fft_real_only=False):
        super(DenseNet, self).__init__()
        # densenet.py:109-115: flow + Re/Im of its spectrum (with_fft: 3 channels), the spectrum alone (only_fft: 2), one
        # less without the imaginary part (fft_real_only); the channels come from the dataset (dataset.py:1330-1341,
        # deepards_amd.tiles.perform_fft)
        fft_real_modifier = -1 if fft_real_only else 0
        initial_chans = 3 + fft_real_modifier if with_fft else (2 + fft_real_modifier if only_fft else 1)
        if growth_rate % 32 or (bn_size * growth_rate) % 32 or num_init_features % 32 or 256 % num_init_features:
            raise NotImplementedError('channel counts must be multiples of 32')
        self.drop_rate = drop_rate
        self.features = _Features(OrderedDict((
            ('conv0', conv1d(initial_chans, num_init_features, 7, 2)), ('norm0', _bn(num_init_features)),
            ('relu0', nn.ReLU(inplace=True)), ('pool0', nn.MaxPool1d(kernel_size=3, stride=2, padding=1)))))
        width, self.n_layers = num_init_features, 0
        for i, n in enumerate(block_config):
            block = _DenseBlock(n, width, bn_size, growth_rate, drop_rate)
            self.features.add_module('denseblock%d' % (i + 1), block)
            self.n_layers += block.num_layers
            width += n * growth_rate
            if i + 1 < len(block_config):                            # a transition halves the width between blocks
                self.features.add_module('transition%d' % (i + 1), _Transition(width, width // 2))
                self.n_layers += 2 * _Transition.num_layers          # the reference counts a transition twice (:137-140)
                width //= 2
        self.features.add_module('norm5', _bn(width))
        init_like_reference(self)
        self.inplanes = num_init_features
        self.n_out_filters = width
        self.avgpool = nn.AvgPool1d(7, stride=1)
        # the stem's (kernel, stride, padding) then every block's / transition's, as conv_info() reports them
        self.kernel_sizes, self.strides, self.paddings = [7, 3], [2, 2], [3, 1]
        for mod in self.features.children():
            if isinstance(mod, (_DenseBlock, _Transition)):
                ks, st, pd = mod.conv_info()
                self.kernel_sizes += ks
                self.strides += st
                self.paddings += pd
        # device-resident dropout seed: bumped on the device each forward so a captured graph replays
        # with fresh masks
        self.features.drop_rate = drop_rate
        self.features.register_buffer('_drop_seed', torch.zeros(1, dtype=torch.int64), persistent=False)

    def conv_info(self):
        return self.kernel_sizes, self.strides, self.paddings

    def _features_rlc(self, x, R):
        return self.features.forward_rlc(x, R, True)

    def forward_windows(self, x, rows_per_window, pooled=True):
        """pooled=False: the last map (rows, L, C) itself, for a head that pools it in its own kernel (CNNLinearNetwork.forward_loss)."""
        h = self._features_rlc(x, rows_per_window)
        if h.shape[1] < 7:
            raise ValueError('AvgPool1d(7, stride=1) needs a final length >= 7 (seq_len >= 224); got %d' % h.shape[1])
        if not pooled:
            if h.shape[1] != 7:
                raise TypeError('the un-pooled map is only handed out at the 7-position length the fused head pools')
            return h
        return F_.GlobalAvgPoolFunction.apply(h)

    def forward(self, x):
        return self.forward_windows(x, x.shape[0])

    def forward_no_pool(self, x):
        """relu(features(x)) in the reference's (N, C, L) layout (densenet.py:191-193)."""
        return self._features_rlc(x, x.shape[0]).permute(0, 2, 1)


def densenet18(pretrained=False, progress=True, **kwargs):
    if pretrained:
        raise NotImplementedError('no pretrained weights exist for the 1-D densenet18')
    model = DenseNet(32, (2, 2, 2, 2), 64, **kwargs)
    model.network_name = 'densenet18'
    return model


def _densenet(name, block_config, pretrained, **kwargs):
    if pretrained:
        raise NotImplementedError('no pretrained weights exist for the 1-D %s' % name)
    model = DenseNet(32, block_config, 64, **kwargs)
    model.network_name = name
    return model


def densenet121(pretrained=False, progress=True, **kwargs):
    """reference models/densenet.py:234-242 (growth 32, blocks (6, 12, 24, 16))."""
    return _densenet('densenet121', (6, 12, 24, 16), pretrained, **kwargs)


def densenet169(pretrained=False, progress=True, **kwargs):
    """reference models/densenet.py:256-264."""
    return _densenet('densenet169', (6, 12, 32, 32), pretrained, **kwargs)


def densenet201(pretrained=False, progress=True, **kwargs):
    """reference models/densenet.py:267-275."""
    return _densenet('densenet201', (6, 12, 48, 32), pretrained, **kwargs)
