"""Block-level autograd Functions: each one orchestrates the HIP kernels of one structural unit of the
reference networks (stem, BasicBlock, _DenseLayer, _Transition, final pool, classifier head) for a
WHOLE batch of windows at once -- the reference's Python loop over windows
(models/torch_cnn_linear_network.py:108-113) disappears, BatchNorm stays grouped per window.

Every unit is a single autograd node with explicit gradient accumulation inside, so the autograd
graph of a model is a plain chain and no PyTorch arithmetic kernel runs on the hot path.
"""
import torch
from torch.autograd import Function

from . import hip_ops as H

POOL_MAX, POOL_AVG = 0, 1


class BNState(object):
    """Running-stat buffers of one BatchNorm1d (None when track_running_stats=False)."""
    __slots__ = ('running_mean', 'running_var', 'num_batches_tracked', 'momentum', 'eps')

    def __init__(self, bn):
        self.running_mean = bn.running_mean if bn.track_running_stats else None
        self.running_var = bn.running_var if bn.track_running_stats else None
        self.num_batches_tracked = bn.num_batches_tracked if bn.track_running_stats else None
        self.momentum = 0.1 if bn.momentum is None else bn.momentum
        self.eps = bn.eps


def _stats(x, R, st):
    """Per-window batch statistics + the reference's sequential running-stat update
    (one momentum step per window: SURVEY.md finding 5)."""
    mean, invstd = H.bn_stats(x, R, st.eps)
    if st.running_mean is not None:
        H.bn_running_update(mean, invstd, R * x.shape[1], st.running_mean, st.running_var, st.momentum, st.eps)
        st.num_batches_tracked.add_(mean.shape[0])
    return mean, invstd


class StemFunction(Function):
    """conv k7 s2 p3 (C_in=1) -> BN -> ReLU -> {Max,Avg}Pool1d(3,2,1).
    reference models/resnet.py:141-153, models/densenet.py:118-124."""

    @staticmethod
    def forward(ctx, x2d, w, gamma, beta, R, pool_mode, st):
        y0 = H.stem_conv_fwd(x2d, w)
        mean, invstd = _stats(y0, R, st)
        out = H.bn_relu_pool_fwd(y0, R, mean, invstd, gamma, beta, pool_mode)
        ctx.save_for_backward(x2d, y0, mean, invstd, gamma, beta)
        ctx.R, ctx.pool_mode = R, pool_mode
        return out

    @staticmethod
    def backward(ctx, dout):
        x2d, y0, mean, invstd, gamma, beta = ctx.saved_tensors
        dz = H.pool_bwd(dout.contiguous(), y0, ctx.R, mean, invstd, gamma, beta, ctx.pool_mode)
        dy0, dgamma, dbeta, _ = H.bn_bwd(dz, y0, ctx.R, mean, invstd, gamma, beta, 1, dx=dz)
        dw = H.stem_conv_wgrad(dy0, x2d)
        return None, dw, dgamma, dbeta, None, None, None


class BasicBlockFunction(Function):
    """conv3(s) -> BN -> ReLU -> conv3 -> BN -> (+ identity | BN(conv1x1(s))) -> ReLU.
    reference models/resnet.py:24-40 (BasicBlock.forward), :123-131 (downsample)."""

    @staticmethod
    def forward(ctx, x, w1, g1, b1, w2, g2, b2, wd, gd, bd, stride, R, st1, st2, std):
        wf1, _ = H.repack_weight(w1)
        y1 = H.conv_fwd(x, wf1, stride, 1)
        m1, i1 = _stats(y1, R, st1)
        h1 = H.bn_apply(y1, R, m1, i1, g1, b1, relu=True)
        wf2, _ = H.repack_weight(w2)
        y2 = H.conv_fwd(h1, wf2, 1, 1)
        m2, i2 = _stats(y2, R, st2)
        if wd is not None:
            wfd, _ = H.repack_weight(wd)
            yd = H.conv_fwd(x, wfd, stride, 0)
            md, idd = _stats(yd, R, std)
            res = H.bn_apply(yd, R, md, idd, gd, bd, relu=False)
        else:
            yd = md = idd = None
            res = x
        out = H.bn_apply(y2, R, m2, i2, g2, b2, relu=True, res=res)
        ctx.has_ds = wd is not None
        ctx.stride, ctx.R = stride, R
        saved = [x, w1, g1, b1, w2, g2, b2, y1, m1, i1, h1, y2, m2, i2, out]
        if ctx.has_ds:
            saved += [wd, gd, bd, yd, md, idd]
        ctx.save_for_backward(*saved)
        return out

    @staticmethod
    def backward(ctx, dout):
        s = ctx.saved_tensors
        x, w1, g1, b1, w2, g2, b2, y1, m1, i1, h1, y2, m2, i2, out = s[:15]
        R, stride = ctx.R, ctx.stride
        lin = x.shape[1]
        dout = dout.contiguous()
        # relu + residual add + bn2
        dy2, dg2, db2, g = H.bn_bwd(dout, y2, R, m2, i2, g2, b2, 2, out=out, want_g=True)
        dw2 = H.conv_wgrad(dy2, h1, 3, 1, 1)
        _, wdd2 = H.repack_weight(w2, need_fwd=False, need_dgrad=True)
        dh1 = H.conv_dgrad(dy2, wdd2, 1, 1, h1.shape[1])
        dy1, dg1, db1, _ = H.bn_bwd(dh1, y1, R, m1, i1, g1, b1, 1, dx=dh1)
        dw1 = H.conv_wgrad(dy1, x, 3, stride, 1)
        _, wdd1 = H.repack_weight(w1, need_fwd=False, need_dgrad=True)
        if ctx.has_ds:
            wd, gd, bd, yd, md, idd = s[15:]
            dyd, dgd, dbd, _ = H.bn_bwd(g, yd, R, md, idd, gd, bd, 0, dx=g)
            dwd = H.conv_wgrad(dyd, x, 1, stride, 0)
            dx = H.conv_dgrad(dy1, wdd1, stride, 1, lin)
            _, wddd = H.repack_weight(wd, need_fwd=False, need_dgrad=True)
            H.conv_dgrad(dyd, wddd, stride, 0, lin, out=dx, accumulate=True)
        else:
            dwd = dgd = dbd = None
            dx = H.conv_dgrad(dy1, wdd1, stride, 1, lin, out=g, accumulate=True)   # identity grad + conv path
        return dx, dw1, dg1, db1, dw2, dg2, db2, dwd, dgd, dbd, None, None, None, None, None


class DenseLayerFunction(Function):
    """BN -> ReLU -> conv1x1 -> BN -> ReLU -> conv3 -> dropout -> cat([x, new], C).
    reference models/densenet.py:18-41 (_DenseLayer)."""

    @staticmethod
    def forward(ctx, x, g1, b1, w1, g2, b2, w2, R, st1, st2, drop_p, seed, salt):
        m1, i1 = _stats(x, R, st1)
        h = H.bn_apply(x, R, m1, i1, g1, b1, relu=True)
        wf1, _ = H.repack_weight(w1)
        y1 = H.conv_fwd(h, wf1, 1, 0)
        m2, i2 = _stats(y1, R, st2)
        h2 = H.bn_apply(y1, R, m2, i2, g2, b2, relu=True)
        wf2, _ = H.repack_weight(w2)
        new = H.conv_fwd(h2, wf2, 1, 1)
        if drop_p > 0:
            new = H.dropout(new, seed, salt, drop_p)
        out = H.concat2(x, new)
        ctx.R, ctx.drop_p, ctx.salt = R, drop_p, salt
        ctx.save_for_backward(x, g1, b1, w1, g2, b2, w2, m1, i1, h, y1, m2, i2, h2, seed)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, g1, b1, w1, g2, b2, w2, m1, i1, h, y1, m2, i2, h2, seed = ctx.saved_tensors
        R = ctx.R
        cin = x.shape[2]
        dout = dout.contiguous()
        dnew = H.slice_channels(dout, cin, w2.shape[0])
        if ctx.drop_p > 0:
            dnew = H.dropout(dnew, seed, ctx.salt, ctx.drop_p)
        dw2 = H.conv_wgrad(dnew, h2, 3, 1, 1)
        _, wdd2 = H.repack_weight(w2, need_fwd=False, need_dgrad=True)
        dh2 = H.conv_dgrad(dnew, wdd2, 1, 1, h2.shape[1])
        dy1, dg2, db2, _ = H.bn_bwd(dh2, y1, R, m2, i2, g2, b2, 1, dx=dh2)
        dw1 = H.conv_wgrad(dy1, h, 1, 1, 0)
        _, wdd1 = H.repack_weight(w1, need_fwd=False, need_dgrad=True)
        dh = H.conv_dgrad(dy1, wdd1, 1, 0, h.shape[1])
        dx, dg1, db1, _ = H.bn_bwd(dh, x, R, m1, i1, g1, b1, 1, dx=dh)
        H.slice_channels(dout, 0, cin, out=dx, accumulate=True)                     # pass-through half of the cat
        return dx, dg1, db1, dw1, dg2, db2, dw2, None, None, None, None, None, None


class TransitionFunction(Function):
    """BN -> ReLU -> conv1x1 -> AvgPool1d(2,2).  reference models/densenet.py:68-79 (_Transition)."""

    @staticmethod
    def forward(ctx, x, g, b, w, R, st):
        m, i = _stats(x, R, st)
        h = H.bn_apply(x, R, m, i, g, b, relu=True)
        wf, _ = H.repack_weight(w)
        y = H.conv_fwd(h, wf, 1, 0)
        out = H.avgpool_fwd(y, 2)
        ctx.R = R
        ctx.save_for_backward(x, g, b, w, m, i, h)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, g, b, w, m, i, h = ctx.saved_tensors
        dy = H.avgpool_bwd(dout.contiguous(), x.shape[1], 2)
        dw = H.conv_wgrad(dy, h, 1, 1, 0)
        _, wdd = H.repack_weight(w, need_fwd=False, need_dgrad=True)
        dh = H.conv_dgrad(dy, wdd, 1, 0, h.shape[1])
        dx, dg, db, _ = H.bn_bwd(dh, x, ctx.R, m, i, g, b, 1, dx=dh)
        return dx, dg, db, dw, None, None


class NormReluFunction(Function):
    """BN -> ReLU (densenet norm5 + F.relu, models/densenet.py:146,181-182)."""

    @staticmethod
    def forward(ctx, x, g, b, R, st):
        m, i = _stats(x, R, st)
        out = H.bn_apply(x, R, m, i, g, b, relu=True)
        ctx.R = R
        ctx.save_for_backward(x, g, b, m, i)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, g, b, m, i = ctx.saved_tensors
        dx, dg, db, _ = H.bn_bwd(dout.contiguous(), x, ctx.R, m, i, g, b, 1)
        return dx, dg, db, None, None


class GlobalAvgPoolFunction(Function):
    """AvgPool1d(7, stride=1) on an L=7 map -> (rows, C)  (resnet.py:112,159-160; densenet.py:167,183-184)."""

    @staticmethod
    def forward(ctx, x):
        ctx.lin = x.shape[1]
        return H.avgpool_fwd(x, x.shape[1]).view(x.shape[0], x.shape[2])

    @staticmethod
    def backward(ctx, dfeat):
        d = dfeat.contiguous().view(dfeat.shape[0], 1, dfeat.shape[1])
        return H.avgpool_bwd(d, ctx.lin, ctx.lin)


class Linear2Function(Function):
    """linear_final on the flattened (NB, F) block of every window
    (models/torch_cnn_linear_network.py:102,110-112)."""

    @staticmethod
    def forward(ctx, flat, w, bias):
        ctx.save_for_backward(flat, w)
        return H.linear2_fwd(flat, w, bias)

    @staticmethod
    def backward(ctx, dlogits):
        flat, w = ctx.saved_tensors
        dflat, dw, dbias = H.linear2_bwd(dlogits.contiguous(), flat, w, need_input=ctx.needs_input_grad[0])
        return dflat, dw, dbias


class BCEWithLogitsFunction(Function):
    """torch.nn.BCEWithLogitsLoss() (mean) -- train_ards_detector.py:530,929-930."""

    @staticmethod
    def forward(ctx, logits, target):
        loss, d = H.bce_logits(logits.contiguous(), target.contiguous(), want_grad=True)
        ctx.save_for_backward(d)
        return loss.view(())

    @staticmethod
    def backward(ctx, gout):
        (d,) = ctx.saved_tensors
        return d * gout, None


def bce_with_logits(logits, target):
    return BCEWithLogitsFunction.apply(logits, target)
