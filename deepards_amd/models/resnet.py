"""1-D ResNet-18 breath block on MI355X.

Operator surface of reference ``deepards/models/resnet.py`` (ResNet :81-163, BasicBlock :11-40,
resnet18 :166-174): same constructor arguments, sub-module / parameter names (so ``state_dict`` keys
match, including the four parameters the reference never uses: conv1_alt, conv2, bn2 -- SURVEY.md
finding 6), ``n_out_filters`` and ``network_name``.  The torch.nn leaf modules are only parameter
containers here: ``forward`` runs the hand-written HIP kernels through
``deepards_amd.functional`` on a whole batch of windows.
"""
import math

import torch.nn as nn

from .. import functional as F_


def _require_cuda(x, what):
    if not x.is_cuda:
        raise RuntimeError('%s: the deepards_amd models run on MI355X only (input is on %s); there is no '
                           'CPU fallback' % (what, x.device))


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super(BasicBlock, self).__init__()
        self.conv1 = nn.Conv1d(inplanes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm1d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv1d(planes, planes, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn2 = nn.BatchNorm1d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward_rlc(self, x, R):
        """x: (rows, L, C) channels-last; R rows per BatchNorm window."""
        ds = self.downsample
        return F_.BasicBlockFunction.apply(
            x, self.conv1.weight, self.bn1.weight, self.bn1.bias,
            self.conv2.weight, self.bn2.weight, self.bn2.bias,
            None if ds is None else ds[0].weight,
            None if ds is None else ds[1].weight,
            None if ds is None else ds[1].bias,
            self.stride, R, F_.BNState(self.bn1), F_.BNState(self.bn2),
            None if ds is None else F_.BNState(ds[1]))


class ResNet(nn.Module):
    def __init__(self, block, layers, initial_planes=64, first_pool_type='max', double_conv_first=False):
        super(ResNet, self).__init__()
        if block is not BasicBlock:
            raise NotImplementedError('only BasicBlock (resnet18/34 style) is on the accelerated path')
        if double_conv_first:
            raise NotImplementedError('double_conv_first is outside the accelerated hot path')
        if initial_planes not in (64, 128, 256):
            raise NotImplementedError('initial_planes must be 64, 128 or 256 on the accelerated path')
        self.inplanes = initial_planes
        self.expansion = block.expansion
        self.conv1 = nn.Conv1d(1, self.inplanes, kernel_size=7, stride=2, padding=3, bias=False)
        self.conv1_alt = nn.Conv1d(1, self.inplanes, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn1 = nn.BatchNorm1d(self.inplanes)
        self.conv2 = nn.Conv1d(self.inplanes, self.inplanes, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn2 = nn.BatchNorm1d(self.inplanes)
        self.double_conv_first = double_conv_first
        self.relu = nn.ReLU(inplace=True)
        if first_pool_type == 'max':
            self.first_pool = nn.MaxPool1d(kernel_size=3, stride=2, padding=1)
        elif first_pool_type == 'avg':
            self.first_pool = nn.AvgPool1d(kernel_size=3, stride=2, padding=1)
        else:
            raise ValueError('first_pool_type must be "max" or "avg"')
        self.first_pool_type = first_pool_type
        self.layer1 = self._make_layer(block, initial_planes, layers[0])
        self.layer2 = self._make_layer(block, initial_planes * 2, layers[1], stride=2)
        self.layer3 = self._make_layer(block, initial_planes * 4, layers[2], stride=2)
        self.layer4 = self._make_layer(block, initial_planes * 8, layers[3], stride=2)
        self.avgpool = nn.AvgPool1d(7, stride=1)
        for m in self.modules():
            if isinstance(m, nn.Conv1d):
                n = m.kernel_size[0] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n))
            elif isinstance(m, nn.BatchNorm1d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()
        self.n_out_filters = self.inplanes * block.expansion

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                nn.Conv1d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                nn.BatchNorm1d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)

    def forward_windows(self, x, rows_per_window):
        """x: (rows, 1, L) with rows = windows * rows_per_window; BatchNorm statistics are taken per
        window, exactly as when the reference feeds one (NB, 1, L) window at a time."""
        _require_cuda(x, 'ResNet')
        if x.dim() != 3 or x.shape[1] != 1:
            raise ValueError('expected (rows, 1, L) input, got %s' % (tuple(x.shape),))
        rows, _, l = x.shape
        if rows % rows_per_window:
            raise ValueError('rows not a multiple of rows_per_window')
        x2d = x.contiguous().float().view(rows, l)
        pool = F_.POOL_MAX if self.first_pool_type == 'max' else F_.POOL_AVG
        h = F_.StemFunction.apply(x2d, self.conv1.weight, self.bn1.weight, self.bn1.bias, rows_per_window, pool,
                                  F_.BNState(self.bn1))
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in layer:
                h = blk.forward_rlc(h, rows_per_window)
        if h.shape[1] < 7:
            raise ValueError('AvgPool1d(7, stride=1) needs a final length >= 7 (seq_len >= 224); got %d' % h.shape[1])
        return F_.GlobalAvgPoolFunction.apply(h)

    def forward(self, x):
        # one call == one BatchNorm batch, like the reference's breath_block(x[i])
        return self.forward_windows(x, x.shape[0])


def resnet18(pretrained=False, **kwargs):
    model = ResNet(BasicBlock, [2, 2, 2, 2], **kwargs)
    model.network_name = 'resnet18'
    return model


def resnet34(pretrained=False, **kwargs):
    """reference models/resnet.py:178-187: BasicBlock [3, 4, 6, 3]."""
    model = ResNet(BasicBlock, [3, 4, 6, 3], **kwargs)
    model.network_name = 'resnet34'
    return model
