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


def conv1d(cin, cout, k, stride=1):
    """Bias-free Conv1d with 'same'-style padding k // 2 (every conv of both backbones: k7 p3, k3 p1, k1 p0)."""
    return nn.Conv1d(cin, cout, kernel_size=k, stride=stride, padding=k // 2, bias=False)


def init_like_reference(net):
    """The reference's initialisation (resnet.py:115-121, densenet.py:155-164): conv weights N(0, sqrt(2 / (k * C_out))),
    BatchNorm gamma 1 / beta 0."""
    for m in net.modules():
        if isinstance(m, nn.Conv1d):
            m.weight.data.normal_(0, math.sqrt(2.0 / (m.kernel_size[0] * m.out_channels)))
        elif isinstance(m, nn.BatchNorm1d):
            m.weight.data.fill_(1)
            m.bias.data.zero_()


class BasicBlock(nn.Module):
    """Parameter container of one residual block; child names and their order are the state_dict contract
    (conv1, bn1, relu, conv2, bn2, downsample -- reference models/resnet.py:14-22)."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super(BasicBlock, self).__init__()
        children = (('conv1', conv1d(inplanes, planes, 3, stride)), ('bn1', nn.BatchNorm1d(planes)),
                    ('relu', nn.ReLU(inplace=True)), ('conv2', conv1d(planes, planes, 3)), ('bn2', nn.BatchNorm1d(planes)))
        for name, mod in children:
            self.add_module(name, mod)
        self.downsample, self.stride = downsample, stride

    def forward_rlc(self, x, R, x3=None, want_out3=False, pool_out=False, split_dx=False):
        """x: (rows, L, C) channels-last; R rows per BatchNorm window.  Conv arithmetic 'f32x3p': ``x3`` = the input in the
        x3 format (``x`` is then the autograd handle) and ``want_out3`` asks for ``(handle, out3)`` (functional: the x3 flow)."""
        ds = self.downsample
        dsw = (None, None, None, None) if ds is None else (ds[0].weight, ds[1].weight, ds[1].bias, F_.BNState(ds[1]))
        return F_.BasicBlockFunction.apply(
            x, self.conv1.weight, self.bn1.weight, self.bn1.bias, self.conv2.weight, self.bn2.weight, self.bn2.bias,
            dsw[0], dsw[1], dsw[2], self.stride, R, F_.BNState(self.bn1), F_.BNState(self.bn2), dsw[3], x3, want_out3, pool_out, split_dx)

    def takes_x3(self, rows, l_in, R):
        """Whether this block's conv1 reads the x3 format under the current conv arithmetic: a k3 s1 conv (no downsample)
        or a stride-2 block entry, on a shape that has the x3 store forms."""
        if not F_.x3_block_ok(rows, l_in, self.conv1.in_channels, R):
            return False
        if self.downsample is not None:           # the stride-2 block entry: conv1 + the 1x1 downsample conv in one launch
            return self.stride == 2 and F_.s2_x3_ok(self.conv1.weight, self.downsample[0].weight, l_in)
        return self.stride == 1 and F_._is_wino(self.conv1.weight, 1, 1) == 49


_POOLS = {'max': nn.MaxPool1d, 'avg': nn.AvgPool1d}
_SPLIT_DX = True          # an identity block's input gradient as two terms, summed by the bn2 backward in front (False, tests: accumulated by its data-gradient conv)
_FUSED_TAIL = True        # the last block's BatchNorm pools for the head (False, tests: its map is stored and the head pools it)


class ResNet(nn.Module):
    """Four stages of BasicBlocks behind a k7 s2 stem.  The module tree (names, order, shapes -- including the stem's
    unused conv1_alt / conv2 / bn2, SURVEY finding 6) is the reference's (resnet.py:81-135): 129 state_dict keys for
    cnn_linear + resnet18."""

    fused_tail = True     # forward_windows(pooled='fused')

    def __init__(self, block, layers, initial_planes=64, first_pool_type='max', double_conv_first=False):
        super(ResNet, self).__init__()
        if block is not BasicBlock:
            raise NotImplementedError('only BasicBlock (resnet18/34 style) is on the accelerated path')
        if initial_planes not in (64, 128, 256):
            raise NotImplementedError('initial_planes must be 64, 128 or 256 on the accelerated path')
        if first_pool_type not in _POOLS:
            raise ValueError('first_pool_type must be "max" or "avg"')
        p = initial_planes
        self.expansion, self.double_conv_first, self.first_pool_type = block.expansion, double_conv_first, first_pool_type
        stem = (('conv1', conv1d(1, p, 7, 2)), ('conv1_alt', conv1d(1, p, 3)), ('bn1', nn.BatchNorm1d(p)),
                ('conv2', conv1d(p, p, 7, 2)), ('bn2', nn.BatchNorm1d(p)), ('relu', nn.ReLU(inplace=True)),
                ('first_pool', _POOLS[first_pool_type](kernel_size=3, stride=2, padding=1)))
        for name, mod in stem:
            self.add_module(name, mod)
        width = p
        for i, n_blocks in enumerate(layers):                       # stage i: p * 2^i planes, stride 2 from stage 2 on
            planes, stride = p << i, (1 if i == 0 else 2)
            out = planes * block.expansion
            shortcut = None
            if stride != 1 or width != out:
                shortcut = nn.Sequential(conv1d(width, out, 1, stride), nn.BatchNorm1d(out))
            stage = [block(width, planes, stride, shortcut)] + [block(out, planes) for _ in range(n_blocks - 1)]
            self.add_module('layer%d' % (i + 1), nn.Sequential(*stage))
            width = out
        self.inplanes = width
        self.avgpool = nn.AvgPool1d(7, stride=1)
        init_like_reference(self)
        self.n_out_filters = width

    def forward_windows(self, x, rows_per_window, pooled=True):
        """x: (rows, 1, L) with rows = windows * rows_per_window; BatchNorm statistics are taken per
        window, exactly as when the reference feeds one (NB, 1, L) window at a time.  pooled=False: the last map
        (rows, 7, C) itself, for a head that pools it in its own kernel; pooled='fused' (CNNLinearNetwork.forward_loss):
        that map, or -- where the last block can pool it itself -- the pooled features (rows, C) without the map."""
        _require_cuda(x, 'ResNet')
        if x.dim() != 3 or x.shape[1] != 1:
            raise ValueError('expected (rows, 1, L) input, got %s' % (tuple(x.shape),))
        rows, _, l = x.shape
        if rows % rows_per_window:
            raise ValueError('rows not a multiple of rows_per_window')
        x2d = x.contiguous().float().view(rows, l)
        pool = F_.POOL_MAX if self.first_pool_type == 'max' else F_.POOL_AVG
        # which blocks read their input in the x3 format (conv arithmetic 'f32x3p'): decided from the shapes up front, so
        # that every producer knows what its consumer wants
        blocks = [blk for layer in (self.layer1, self.layer2, self.layer3, self.layer4) for blk in layer]
        lens, cur = [], ((l // 2) - 1) // 2 + 1              # stem conv s2, then pool(3, 2, 1)
        for blk in blocks:
            lens.append(cur)
            cur = (cur - 1) // blk.stride + 1 if blk.stride > 1 else cur
        takes3 = [blk.takes_x3(rows, li, rows_per_window) for blk, li in zip(blocks, lens)] + [False]
        if self.double_conv_first:          # resnet.py:144-149: conv1_alt -> bn1 -> conv2 -> bn2 (conv1 is the dead one then)
            h = F_.DoubleStemFunction.apply(x2d, self.conv1_alt.weight, self.bn1.weight, self.bn1.bias, self.conv2.weight,
                                            self.bn2.weight, self.bn2.bias, rows_per_window, pool, F_.BNState(self.bn1),
                                            F_.BNState(self.bn2), takes3[0])
        else:
            h = F_.StemFunction.apply(x2d, self.conv1.weight, self.bn1.weight, self.bn1.bias, rows_per_window, pool,
                                      F_.BNState(self.bn1), takes3[0])
        for i, blk in enumerate(blocks):
            # pooled='fused' (CNNLinearNetwork.forward_loss): the last block hands over its POOLED output (rows, C) -- its map
            # is never stored (BasicBlockFunction pool_out) -- when it is an identity block on the 7-position map
            pool_out = pooled == 'fused' and _FUSED_TAIL and i == len(blocks) - 1 and not takes3[i] and \
                blk.downsample is None and blk.stride == 1 and lens[i] == 7 and F_.H.bn_pool_ok(h, rows_per_window)
            # an identity block behind another block hands its input gradient back as two terms (BasicBlockFunction split_dx)
            split_dx = _SPLIT_DX and i > 0 and blk.downsample is None and not takes3[i] and not takes3[i - 1]
            if takes3[i]:
                h, h3 = h
                h = blk.forward_rlc(h, rows_per_window, h3, takes3[i + 1])
            else:
                h = blk.forward_rlc(h, rows_per_window, None, takes3[i + 1], pool_out, split_dx)
        if pooled == 'fused':
            if h.dim() == 3 and h.shape[1] != 7:
                raise TypeError('the un-pooled map is only handed out at the 7-position length the fused head pools')
            return h
        if h.shape[1] < 7:
            raise ValueError('AvgPool1d(7, stride=1) needs a final length >= 7 (seq_len >= 224); got %d' % h.shape[1])
        if not pooled:
            if h.shape[1] != 7:
                raise TypeError('the un-pooled map is only handed out at the 7-position length the fused head pools')
            return h
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
