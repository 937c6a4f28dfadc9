"""1-D DenseNet-BC (densenet18: block_config (2,2,2,2), growth 32, bn_size 4) on MI355X.

Operator surface of reference ``deepards/models/densenet.py`` (DenseNet :83-194, _DenseLayer :18-44,
_DenseBlock :46-66, _Transition :68-81, densenet18 :223-231): ``features`` Sequential with the same
module names (state_dict keys match), ``avgpool``, ``n_out_filters``, ``network_name``,
``forward_no_pool`` and ``conv_info()``.  BatchNorms are built with track_running_stats=False and
dropout stays active whenever the module is in training mode -- which is always on this path
(SURVEY.md finding 4).  Compute runs in the HIP kernels via ``deepards_amd.functional``.
"""
import math
from collections import OrderedDict

import torch
import torch.nn as nn

from .. import functional as F_
from .resnet import _require_cuda


class _DenseLayer(nn.Sequential):
    num_layers = 2

    def __init__(self, num_input_features, growth_rate, bn_size, drop_rate, track_running_stats):
        super(_DenseLayer, self).__init__()
        self.add_module('norm1', nn.BatchNorm1d(num_input_features, track_running_stats=track_running_stats))
        self.add_module('relu1', nn.ReLU(inplace=True))
        self.add_module('conv1', nn.Conv1d(num_input_features, bn_size * growth_rate, kernel_size=1, stride=1,
                                           bias=False))
        self.add_module('norm2', nn.BatchNorm1d(bn_size * growth_rate, track_running_stats=track_running_stats))
        self.add_module('relu2', nn.ReLU(inplace=True))
        self.add_module('conv2', nn.Conv1d(bn_size * growth_rate, growth_rate, kernel_size=3, stride=1, padding=1,
                                           bias=False))
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
    def __init__(self, num_layers, num_input_features, bn_size, growth_rate, drop_rate, track_running_stats):
        super(_DenseBlock, self).__init__()
        self.kernel_sizes, self.strides, self.paddings = [], [], []
        self.track_running_stats = track_running_stats
        for i in range(num_layers):
            layer = _DenseLayer(num_input_features + i * growth_rate, growth_rate, bn_size, drop_rate,
                                track_running_stats)
            lks, ls, lp = layer.conv_info()
            self.kernel_sizes.extend(lks)
            self.strides.extend(ls)
            self.paddings.extend(lp)
            self.add_module('denselayer%d' % (i + 1), layer)
        self.num_layers = _DenseLayer.num_layers * num_layers

    def conv_info(self):
        return self.kernel_sizes, self.strides, self.paddings


class _Transition(nn.Sequential):
    num_layers = 1

    def __init__(self, num_input_features, num_output_features, track_running_stats):
        super(_Transition, self).__init__()
        self.add_module('norm', nn.BatchNorm1d(num_input_features, track_running_stats=track_running_stats))
        self.add_module('relu', nn.ReLU(inplace=True))
        self.add_module('conv', nn.Conv1d(num_input_features, num_output_features, kernel_size=1, stride=1,
                                          bias=False))
        self.add_module('pool', nn.AvgPool1d(kernel_size=2, stride=2))

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
        if x.dim() != 3 or x.shape[1] != 1:
            raise ValueError('expected (rows, 1, L) input, got %s' % (tuple(x.shape),))
        rows, _, l = x.shape
        if rows % R:
            raise ValueError('rows not a multiple of rows_per_window')
        x2d = x.contiguous().float().view(rows, l)
        h = F_.StemFunction.apply(x2d, self.conv0.weight, self.norm0.weight, self.norm0.bias, R, F_.POOL_MAX,
                                  F_.BNState(self.norm0))
        use_drop = self.training and self.drop_rate > 0
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

    def forward(self, x):
        return self.forward_rlc(x, x.shape[0], False).permute(0, 2, 1)


class DenseNet(nn.Module):
    def __init__(self, growth_rate=32, block_config=(6, 12, 24, 16), num_init_features=64, bn_size=4,
                 drop_rate=0.2, num_classes=1000, with_fft=False, only_fft=False, fft_real_only=False):
        super(DenseNet, self).__init__()
        if with_fft or only_fft:
            raise NotImplementedError('FFT input channels are outside the accelerated hot path (in_channels=1)')
        if growth_rate % 32 or (bn_size * growth_rate) % 32 or num_init_features % 32 or 256 % num_init_features:
            raise NotImplementedError('channel counts must be multiples of 32')
        self.kernel_sizes, self.strides, self.paddings = [], [], []
        self.n_layers = 0
        self.inplanes = num_init_features
        self.drop_rate = drop_rate
        track_running_stats = False
        self.features = _Features(OrderedDict([
            ('conv0', nn.Conv1d(1, num_init_features, kernel_size=7, stride=2, padding=3, bias=False)),
            ('norm0', nn.BatchNorm1d(num_init_features, track_running_stats=track_running_stats)),
            ('relu0', nn.ReLU(inplace=True)),
            ('pool0', nn.MaxPool1d(kernel_size=3, stride=2, padding=1)),
        ]))
        self.kernel_sizes.extend([7, 3])
        self.strides.extend([2, 2])
        self.paddings.extend([3, 1])
        num_features = num_init_features
        for i, num_layers in enumerate(block_config):
            block = _DenseBlock(num_layers=num_layers, num_input_features=num_features, bn_size=bn_size,
                                growth_rate=growth_rate, drop_rate=drop_rate,
                                track_running_stats=track_running_stats)
            self.update_conv_info(block)
            self.features.add_module('denseblock%d' % (i + 1), block)
            num_features = num_features + num_layers * growth_rate
            if i != len(block_config) - 1:
                trans = _Transition(num_input_features=num_features, num_output_features=num_features // 2,
                                    track_running_stats=track_running_stats)
                self.update_conv_info(trans)
                self.n_layers += trans.num_layers
                self.features.add_module('transition%d' % (i + 1), trans)
                num_features = num_features // 2
        self.features.add_module('norm5', nn.BatchNorm1d(num_features, track_running_stats=track_running_stats))
        for m in self.modules():
            if isinstance(m, nn.Conv1d):
                n = m.kernel_size[0] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n))
            elif isinstance(m, nn.BatchNorm1d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        self.n_out_filters = num_features
        self.avgpool = nn.AvgPool1d(7, stride=1)
        # device-resident dropout seed: bumped on the device each forward so a captured graph replays
        # with fresh masks
        self.features.drop_rate = drop_rate
        self.features.register_buffer('_drop_seed', torch.zeros(1, dtype=torch.int64), persistent=False)

    def update_conv_info(self, obj):
        bks, bs, bp = obj.conv_info()
        self.n_layers += obj.num_layers
        self.kernel_sizes.extend(bks)
        self.strides.extend(bs)
        self.paddings.extend(bp)

    def conv_info(self):
        return self.kernel_sizes, self.strides, self.paddings

    def _features_rlc(self, x, R):
        return self.features.forward_rlc(x, R, True)

    def forward_windows(self, x, rows_per_window):
        h = self._features_rlc(x, rows_per_window)
        if h.shape[1] < 7:
            raise ValueError('AvgPool1d(7, stride=1) needs a final length >= 7 (seq_len >= 224); got %d' % h.shape[1])
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
