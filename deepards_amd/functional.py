"""Block-level autograd Functions: each one orchestrates the HIP kernels of one structural unit of the
reference networks (stem, BasicBlock, _DenseLayer, _Transition, final pool, classifier head) for a
WHOLE batch of windows at once -- the reference's Python loop over windows
(models/torch_cnn_linear_network.py:108-113) disappears, BatchNorm stays grouped per window.

Every unit is a single autograd node with explicit gradient accumulation inside, so the autograd
graph of a model is a plain chain and no PyTorch arithmetic kernel runs on the hot path.
"""
import contextlib
import os

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


# ---- per-step context: packed-weight cache + deferred small kernels -----------------------------------
# Inside ``training_step(model)`` (one step: the weights do not change between its forward and backward)
#   * every conv weight is repacked ONCE, all of them by one batched launch (Wf[k][co][ci], Wd[k][ci][co]);
#   * the ~60 tiny per-layer launches -- BN running statistics, BN dgamma/dbeta folds, split-K slab
#     reductions of the weight gradients -- are queued and served by three batched launches.
# Outside it (plain autograd use, tests) everything runs immediately.
_STEP = {'on': False, 'pack': {}, 'running': [], 'pgrad': [], 'wgrad': [], 'wslab': [], 'stemred': None}


@contextlib.contextmanager
def training_step(model=None):
    _STEP.update(on=True, pack={}, running=[], pgrad=[], wgrad=[], wslab=[], stemred=None, dout2={})
    try:
        if model is not None:
            ms = [m for m in model.modules()
                  if isinstance(m, torch.nn.Conv1d) and m.kernel_size[0] <= 3 and m.in_channels % 32 == 0
                  and m.weight.is_cuda]
            ws = [m.weight for m in ms]
            wino = [_step_pack_code(m) for m in ms]
            for w, c, e in zip(ws, wino, H.repack_multi(ws, wino)):
                _STEP['pack'][(w.data_ptr(), int(c))] = e
        yield
    finally:
        _STEP.update(on=False, pack={}, running=[], pgrad=[], wgrad=[], wslab=[], stemred=None, dout2={})


def flush_forward(defer=False):
    """Run the queued BN running-statistics updates (call after the forward of the step).  defer: a training step leaves
    them queued for flush_backward, whose one launch carries them (nothing reads the running statistics in between)."""
    if defer:
        return
    H.bn_running_multi(_STEP['running'])
    _STEP['running'] = []


def flush_backward():
    """Run the queued parameter-gradient folds (call after the backward, before the optimiser)."""
    if _STEP.get('dout2'):      # a block left the second term of its input gradient for a consumer that never came
        raise RuntimeError('a two-term input gradient (BasicBlockFunction split_dx) was not picked up by the block in front')
    _launch_wgrads()
    # one launch: slab reductions + dgamma / dbeta folds + the running statistics a deferred flush_forward left queued
    dst = [dw.data_ptr() for _, dw in _STEP['wslab']] + [t.data_ptr() for _, tg, tb in _STEP['pgrad'] for t in (tg, tb)]
    if len(set(dst)) != len(dst):                   # a destination written twice in one step (shared parameters)
        _plain_writer()
    H.step_tail_multi(_STEP['wslab'], _STEP['pgrad'], _STEP['running'], accumulate=_acc(), stem=_STEP.get('stemred'))
    _STEP['pgrad'], _STEP['wslab'], _STEP['running'], _STEP['stemred'] = [], [], [], None


# The captured step is ONE chain of kernels on one stream.  Rounds 1-2 could fork the stem's backward (or a stage's weight
# gradients) onto a side stream inside the capture; that never paid once the weight gradients took the Winograd form
# (DESIGN_APPENDIX.md 7a) and a hipGraphExec with parallel branches owns streams that die with ANOTHER such exec (segfault in
# hip::Graph::UpdateStreams) -- so the forked paths and their switches are gone, not just off.
def _launch_wgrads():
    """Launch the queued weight-gradient GEMMs ((dy, x, k, stride, pad, target) jobs), all in one batched call."""
    jobs = _STEP['wgrad']
    if not jobs:
        return
    # the slab reductions chained: each launch of the call folds, as its first blocks, the slabs the launch before it wrote
    # (still in the Infinity Cache, and memory-bound blocks beside matrix-bound ones); the step's tail keeps the last launch's
    specs = [j[:5] + (j[6],) for j in jobs]
    if _WGRAD_CHAIN:
        if len({j[5].data_ptr() for j in jobs}) != len(jobs):
            _plain_writer()
        slabs, reduced = H.conv_wgrad_multi(specs, dws=[j[5] for j in jobs], accumulate=_acc())
    else:
        slabs, reduced = H.conv_wgrad_multi(specs), [False] * len(jobs)
    _STEP['wslab'] += [(sl, j[5]) for sl, j, r in zip(slabs, jobs, reduced) if not r]
    _STEP['wgrad'] = []


_WGRAD_CHAIN = True       # (tests switch it off to compare with the one reduction launch behind all weight gradients)
_WINOGRAD = os.environ.get('DA_WINOGRAD', '1') != '0'   # 0: the direct fp32 kernels (the second fp32 implementation the tests compare)
_WINO4_MIN_C = 512        # channels from which F(4,3) beats F(2,3) (scripts/bench_wino.py; DESIGN appendix)


# Arithmetic of the k3 s1 p1 convs' forward / data gradient: 'f32' (Winograd on the fp32 matrix cores), 'bf16' (BASELINE
# config C3: operands rounded to bf16, fp32 sums, conv_bf16.hip; also the k3 s1 weight gradients) or 'f32x3p'
# (fp32-equivalent products from exact three-term bf16 splits on the bf16 matrix cores, the split done by the PRODUCERS:
# the BatchNorm / pool kernels in front of a k3 s1 conv store the x3 format, conv_x3p.hip and the x3 weight-gradient
# kernel read it; forward, data gradient and weight gradient; opt-in, frozen since round 3).
_CONV_DTYPE = os.environ.get('DA_CONV_DTYPE', 'f32')
_STEM_FUSED = os.environ.get('DA_STEM_FUSED', '1') != '0'   # the default stem recomputes its conv output instead of storing it
CONV_DTYPES = ('f32', 'bf16', 'f32x3p')


def set_conv_dtype(name):
    """One of CONV_DTYPES (see _CONV_DTYPE); captured steps keep the arithmetic they were captured with."""
    global _CONV_DTYPE
    if name not in CONV_DTYPES:
        raise ValueError('conv dtype must be one of %s' % (CONV_DTYPES,))
    if name != 'bf16' and H.act_dtype() == 'bf16':
        H.set_act_dtype('f32')                     # fp32 convs read fp32 activations
    _CONV_DTYPE = name
    H.WGRAD_BF16 = name == 'bf16'


def conv_dtype():
    return _CONV_DTYPE


def set_storage_dtype(name):
    """'f32' or 'bf16': storage of every activation / activation-gradient tensor between two kernels (BASELINE's bf16
    configs C3 / C5: bf16 storage with fp32 statistics and accumulators).  bf16 storage needs conv dtype 'bf16' and a
    network whose convs all have bf16 kernels (the ResNets: channel counts multiples of 64, even lengths at the
    stride-2 heads); captured steps keep the storage they were captured with."""
    if name == 'bf16' and _CONV_DTYPE != 'bf16':
        raise ValueError("bf16 storage needs conv dtype 'bf16' (set_conv_dtype('bf16') first)")
    H.set_act_dtype(name)


def storage_dtype():
    return H.act_dtype()


set_conv_dtype(_CONV_DTYPE)


def _is_wino(w, stride, pad):
    """How a conv's forward / data gradient runs: 0 = direct fp32 kernel; k3 s1 p1 convs: 4 = Winograd F(2,3) (2/3 of
    the direct conv's MFMAs, fp32 throughout), 6 = F(4,3) (1/2 of them; pays once both channel counts reach
    _WINO4_MIN_C), 16 = bf16 products with fp32 sums (conv dtype 'bf16', channel counts multiples of 64; also the
    k3 s2 p1 and k1 s2 p0 convs)."""
    if _CONV_DTYPE == 'bf16' and w.shape[0] % 64 == 0 and w.shape[1] % 64 == 0 and (
            (w.shape[2] == 3 and pad == 1 and stride in (1, 2)) or (w.shape[2] == 1 and pad == 0 and stride == 2)):
        return 16                                   # also the stride-2 block heads and 1x1 downsamples (even lengths)
    if not (w.shape[2] == 3 and stride == 1 and pad == 1 and w.shape[0] % 32 == 0 and w.shape[1] % 32 == 0):
        return 0
    if _CONV_DTYPE == 'f32x3p' and w.shape[0] % 64 == 0 and w.shape[1] % 64 == 0:
        return 49                                   # the same on x3 (pre-split) operands (conv_x3p.hip); float operands: _fp32_code
    return _fp32_code(w)


def s2_x3_ok(w1, wd, l_in):
    """Whether a stride-2 block entry (k3 s2 p1 conv + 1x1 s2 downsample) runs on x3 operands (conv arithmetic 'f32x3p':
    H.conv_x3p_s2_fwd / _dgrad and the x3 weight-gradient jobs): even input length, channel counts multiples of 64."""
    return _CONV_DTYPE == 'f32x3p' and H.act_dtype() == 'f32' and wd is not None and l_in % 2 == 0 and \
        tuple(w1.shape[2:]) == (3,) and tuple(wd.shape[2:]) == (1,) and tuple(wd.shape[:2]) == tuple(w1.shape[:2]) and \
        w1.shape[0] % 64 == 0 and w1.shape[1] % 64 == 0


def _step_pack_code(m):
    """The pack form the step's batched repack prepares for a conv module (a consumer that needs another form packs it
    itself, once: _pack)."""
    w, stride, pad = m.weight, m.stride[0], m.padding[0]
    if _CONV_DTYPE == 'f32x3p' and H.act_dtype() == 'f32' and stride == 2 and w.shape[0] % 64 == 0 and \
            w.shape[1] % 64 == 0 and (w.shape[2], pad) in ((3, 1), (1, 0)):
        return 49                                   # the stride-2 block entries on x3 operands (s2_x3_ok)
    return _is_wino(w, stride, pad)


def _fp32_code(w):
    """The fp32 kernel of a k3 s1 p1 conv whose operand is a float tensor: Winograd F(2,3) / F(4,3) or direct."""
    if not _WINOGRAD:
        return 0
    return 6 if min(w.shape[0], w.shape[1]) >= _WINO4_MIN_C else 4


def _pack(w, code):
    """(wf, wd, uf, ud) of a conv weight: direct packs (code 0), Winograd taps or bf16 tap packs (in the uf / ud
    places), repacked once per step."""
    key = (w.data_ptr(), int(code))                 # a weight may be packed in two forms in one step (x3 and fp32 consumers)
    e = _STEP['pack'].get(key) if _STEP['on'] else None
    if e is None:
        e = H.repack_multi([w], [code])[0]
        if _STEP['on']:
            _STEP['pack'][key] = e
    return e


def _need_bf16_kernel(code, w, stride, pad):
    if code != 16 and H.act_dtype() == 'bf16':
        raise NotImplementedError('bf16 activation storage: the conv %s stride %d pad %d has no bf16 kernel (channel counts '
                                  'must be multiples of 64, stride-2 inputs of even length)' % (tuple(w.shape), stride, pad))


def _conv_fwd(x, w, stride, pad):
    code = _is_wino(w, stride, pad)
    if code == 49:
        if H.is_x3(x):
            return H.conv3_x3p(x, _pack(w, 49)[2])
        code = _fp32_code(w)                        # a float operand (a shape without x3 producers): the fp32 kernels
    if code == 16 and stride == 2 and x.shape[1] % 2:
        code = 0                                    # odd length: the fp32 kernel
    _need_bf16_kernel(code, w, stride, pad)
    if code == 16:
        return H.conv3_bf16(x, _pack(w, code)[2]) if stride == 1 else H.conv_fwd_bf16_s2(x, _pack(w, code)[2])
    if code:
        return H.conv3_winograd(x, _pack(w, code)[2])
    return H.conv_fwd(x, _pack(w, 0)[0], stride, pad)


def _conv_dgrad(dy, w, stride, pad, l_in, out=None, accumulate=False):
    code = _is_wino(w, stride, pad)
    if code == 49:
        if H.is_x3(dy):
            return H.conv3_x3p(dy, _pack(w, 49)[3], out=out, accumulate=accumulate)
        code = _fp32_code(w)
    if code == 16 and stride == 2 and l_in % 2:
        code = 0
    _need_bf16_kernel(code, w, stride, pad)
    if code == 16:
        if stride == 2:
            return H.conv_dgrad_bf16_s2(dy, _pack(w, code)[3], l_in, out=out, accumulate=accumulate)
        return H.conv3_bf16(dy, _pack(w, code)[3], out=out, accumulate=accumulate)
    if code:
        return H.conv3_winograd(dy, _pack(w, code)[3], out=out, accumulate=accumulate)
    return H.conv_dgrad(dy, _pack(w, 0)[1], stride, pad, l_in, out=out, accumulate=accumulate)


# A captured training step may run WITHOUT the zero-fill of the gradient bucket: every gradient destination is then written
# exactly once, by a writer that has an overwrite form (the step's queued folds / slab reductions, the fused head).  The
# trainer learns whether that holds from the eager warm-up pass in front of the capture (_OV['ok'] stays True) and switches
# the form on for the capture only (grad_overwrite); a writer that can only accumulate refuses to run in such a step.
_OV = {'on': False, 'ok': True}


def grad_overwrite(on):
    _OV['on'] = bool(on)


def _acc():
    """accumulate flag of the writers that have an overwrite form."""
    return not _OV['on']


def _plain_writer():
    """A backward is about to ACCUMULATE into a trainer's gradient destination (no overwrite form)."""
    _OV['ok'] = False
    if _OV['on']:
        raise RuntimeError('a gradient writer without an overwrite form ran in a step captured without the gradient zero-fill')


def _tgt(*params):
    """Gradient destinations a trainer attached to the Parameters (``p._da_grad``: a view into its flat
    gradient bucket).  With a destination the backward kernels accumulate straight into it and autograd
    gets None (no AccumulateGrad add kernel); without, gradients are returned the usual way."""
    return tuple(None if p is None else getattr(p, '_da_grad', None) for p in params)


# ---- the x3 flow (conv arithmetic 'f32x3p') -------------------------------------------------------------------------------
# An activation that feeds a k3 s1 conv or a stride-2 block entry (k3 s2 conv + 1x1 s2 downsample) is stored ONLY in the x3 format (H.is_x3: bf16 (rows, L, C/16, 3, 16)).  autograd
# checks a gradient's shape against the tensor it belongs to, and the gradients stay float (rows, L, C) -- so the tensor
# that carries the autograd edge between two blocks is a zero-stride float "handle" of the logical shape (no memory, never
# read), and the x3 data travels beside it as a second, non-differentiable argument / result of the block Functions.
_HANDLES = {}


def x3_handle(x3):
    rows, l, g = x3.shape[:3]
    z = _HANDLES.get(x3.device)
    if z is None:
        z = _HANDLES[x3.device] = torch.zeros(1, device=x3.device, dtype=torch.float32)
    return z.expand(rows, l, g * 16)


def x3_block_ok(rows, l, c, R):
    """Whether a block whose activations are (rows, L, C) in windows of R rows can run its k3 s1 convs on x3 operands:
    conv arithmetic 'f32x3p', float storage, 64-multiple channels and the single-pass BatchNorm geometry (its store forms)."""
    return _CONV_DTYPE == 'f32x3p' and H.act_dtype() == 'f32' and c % 64 == 0 and H.bn_x3_ok(rows, l, c, R)


def _bn_apply_x(x, R, s, st, gamma, beta, relu, res=None, want_mask=False, out_x3=True):
    """_bn_apply through the x3 store forms (H.bn_fwd_x): the output in the x3 format and / or an x3 residual."""
    r = H.bn_fwd_x(x, R, gamma, beta, relu=relu, res=res, eps=st.eps, want_mask=want_mask, out_x3=out_x3)
    s.mean, s.invstd = r[1], r[2]
    s.mask = r[3] if want_mask else None
    _running(x, R, s, st)
    return r[0]


def _bn_bwd_x(dout, x, R, mean, invstd, gamma, beta, mode, tg, tb, want_g=False, mask=None, dx_x3=True):
    """_bn_bwd through H.bn_bwd_x (dx in the x3 format); -> (dx, dgamma|None, dbeta|None[, g])."""
    dx, g, ds = H.bn_bwd_x(dout, x, R, mean, invstd, gamma, beta, mode, want_g=want_g, mask=mask, dx_x3=dx_x3)
    direct = tg is not None and tb is not None
    dg = db = None
    if direct and _STEP['on']:
        _STEP['pgrad'].append((ds, tg, tb))
    elif direct:
        H.bn_param_grad_multi([(ds, tg, tb)], accumulate=True)
    else:
        dg, db = torch.empty_like(gamma), torch.empty_like(beta)
        H.bn_param_grad_multi([(ds, dg, db)], accumulate=False)
    return (dx, dg, db, g) if want_g else (dx, dg, db)


# Test instrumentation: with a list here, every block Function appends the post-ReLU activation it stores (float RLC or x3),
# in forward order -- the ReLU DECISIONS this run took, which the parity tests hand to the oracle's backward
# (tests/tools/decision_match.py) instead of searching for them.  None (the default): nothing is recorded.
DECISION_TAP = None


def _tap(t):
    if DECISION_TAP is not None:
        DECISION_TAP.append(t)


class _Stats(object):
    """Per-window statistics of one BatchNorm input (filled by the BatchNorm's consumer, _bn_apply)."""
    __slots__ = ('mean', 'invstd', 'mask')


def _bn_apply(x, R, s, st, gamma, beta, relu, res=None, want_mask=False):
    """act(bn(x)(+res)): statistics and normalisation in one call; fills s.mean / s.invstd (and s.mask: the ReLU
    decisions as bits, for the backward) and books the running update."""
    if want_mask:
        out, s.mean, s.invstd, s.mask = H.bn_fwd(x, R, gamma, beta, relu=relu, res=res, eps=st.eps, want_mask=True)
    else:
        out, s.mean, s.invstd = H.bn_fwd(x, R, gamma, beta, relu=relu, res=res, eps=st.eps)
    _running(x, R, s, st)
    return out


def _running(x, R, s, st):
    if st.running_mean is None:
        return
    item = (s.mean, s.invstd, R * x.shape[1], st.running_mean, st.running_var, st.num_batches_tracked, st.momentum,
            st.eps)
    if _STEP['on']:
        _STEP['running'].append(item)
    else:
        H.bn_running_multi([item])


class StemFunction(Function):
    """conv k7 s2 p3 (C_in = 1, or 2 / 3 with the DenseNet FFT channels) -> BN -> ReLU -> {Max,Avg}Pool1d(3,2,1).
    reference models/resnet.py:141-153, models/densenet.py:109-124.  x2d: (rows, L) or (rows, C_in, L)."""

    @staticmethod
    def forward(ctx, x2d, w, gamma, beta, R, pool_mode, st, want_out3=False, out_cb=0):
        # out_cb > C0: the output is the first C0 channels of a fresh (rows, Lp, out_cb) buffer -- a dense block's pitched
        # buffer, which its layers then fill (DenseBlockFunction); the buffer is what is returned
        # The default stem (one input channel) never stores its conv output -- 36.7 MB at B = 64 for 7 FMAs an element: the
        # statistics, the apply + pool pass and the whole backward recompute it from the raw rows (bit for bit the same
        # forward values; H.stem_fused_fwd / stem_fused_bwd).  The stems with FFT channels keep the stored map.
        ctx.fused = _STEM_FUSED and H.stem_fused_ok(x2d, w, R)
        s_ = _Stats()
        c0 = w.shape[0]
        buf = view = None
        if out_cb > c0:
            lc = (x2d.shape[-1] + 2 * (w.shape[2] // 2) - w.shape[2]) // 2 + 1
            buf = torch.empty((x2d.shape[0], (lc - 1) // 2 + 1, out_cb), device=x2d.device, dtype=torch.float32)
            view = buf[:, :, :c0]
        ctx.c0 = c0 if buf is not None else 0
        if ctx.fused:
            out, mean, invstd = H.stem_fused_fwd(x2d, w, R, gamma, beta, pool_mode, st.eps, out_x3=want_out3, out=view)
            y0 = w                                      # (nothing of the stem's resolution is kept for the backward)
            wn = R * (x2d.shape[-1] // 2)
        else:
            y0 = H.stem_conv_fwd(x2d, w)
            mean, invstd = H.bn_stats(y0, R, st.eps)
            out = H.bn_relu_pool_fwd(y0, R, mean, invstd, gamma, beta, pool_mode, out_x3=want_out3, out=view)
            wn = R * y0.shape[1]
        if buf is not None:
            out = buf
        s_.mean, s_.invstd = mean, invstd
        if st.running_mean is not None:
            item = (mean, invstd, wn, st.running_mean, st.running_var, st.num_batches_tracked, st.momentum, st.eps)
            if _STEP['on']:
                _STEP['running'].append(item)
            else:
                H.bn_running_multi([item])
        ctx.save_for_backward(x2d, y0, mean, invstd, gamma, beta)
        ctx.R, ctx.pool_mode = R, pool_mode
        ctx.gt = _tgt(w, gamma, beta)
        if want_out3:                               # (handle, x3 data): see "the x3 flow" above
            ctx.mark_non_differentiable(out)
            ctx.set_materialize_grads(False)        # (or autograd fills a zero "gradient" of the x3 tensor every backward)
            return x3_handle(out), out
        return out

    @staticmethod
    def backward(ctx, dout, _d3=None):
        x2d, y0, mean, invstd, gamma, beta = ctx.saved_tensors
        tw, tg, tb = ctx.gt
        dout = dout.contiguous()
        if ctx.c0:                                      # the gradient of a pitched buffer: the stem's channels are its first C0
            dout = dout[:, :, :ctx.c0]
        if ctx.fused:                                   # y0 holds the conv weight here
            if _STEM_TAIL and tw is not None and _STEP['on'] and _STEP.get('stemred') is None:
                # inside a training step: the last fold of the weight gradient rides on the step's tail launch
                (part, nblk, n), ds = H.stem_fused_bwd(dout, x2d, y0, ctx.R, mean, invstd, gamma, beta, ctx.pool_mode, defer=True)
                _STEP['stemred'] = (part, nblk, n, tw)
                dw = None
            else:
                if tw is not None:
                    _plain_writer()
                dw, ds = H.stem_fused_bwd(dout, x2d, y0, ctx.R, mean, invstd, gamma, beta, ctx.pool_mode, dw=tw,
                                          accumulate=tw is not None)
            dgamma = dbeta = None
            if tg is not None and tb is not None:
                if _STEP['on']:
                    _STEP['pgrad'].append((ds, tg, tb))
                else:
                    _plain_writer()
                    H.bn_param_grad_multi([(ds, tg, tb)], accumulate=True)
            else:
                dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(beta)
                H.bn_param_grad_multi([(ds, dgamma, dbeta)], accumulate=False)
            return None, None if tw is not None else dw, dgamma, dbeta, None, None, None, None, None
        dz = H.pool_bwd(dout, y0, ctx.R, mean, invstd, gamma, beta, ctx.pool_mode)
        dy0, dgamma, dbeta = _bn_bwd(dz, y0, ctx.R, mean, invstd, gamma, beta, 1, tg, tb, dx=dz)
        if tw is not None:
            _plain_writer()
        dw = H.stem_conv_wgrad(dy0, x2d, out=tw, accumulate=tw is not None)
        return None, None if tw is not None else dw, dgamma, dbeta, None, None, None, None, None


class DoubleStemFunction(Function):
    """ResNet(double_conv_first=True), resnet.py:144-153: conv1_alt k3 s1 p1 (1 -> C0) -> bn1 -> conv2 k7 s2 p3 (C0 -> C0)
    -> bn2 -> ReLU -> pool.  No ReLU between bn1 and conv2; the four parameters that are dead in the default stem
    (conv1_alt, conv2, bn2.weight / bias, SURVEY finding 6) are live here and conv1 is dead instead.  An option path:
    the k7 conv runs on the fp32 direct GEMM kernels in three tap groups (forward, each half of the data gradient, the
    weight gradient) rather than on a kernel of its own."""

    @staticmethod
    def forward(ctx, x2d, wa, g1, b1, w2, g2, b2, R, pool_mode, st1, st2, want_out3=False):
        if H.act_dtype() != 'f32':
            raise NotImplementedError('double_conv_first runs with fp32 activation storage only')
        ya = H.stem_conv_fwd(x2d, wa, stride=1)                       # (rows, L, C0)
        s1 = _Stats()
        h = _bn_apply(ya, R, s1, st1, g1, b1, False)
        wf, wd = H.repack_weight(w2, True, True)
        y2 = H.conv_fwd(h, wf, 2, 3)                                  # (rows, L / 2, C0)
        m2, i2 = H.bn_stats(y2, R, st2.eps)
        out = H.bn_relu_pool_fwd(y2, R, m2, i2, g2, b2, pool_mode, out_x3=want_out3)
        s2 = _Stats()
        s2.mean, s2.invstd = m2, i2
        _running(y2, R, s2, st2)
        ctx.save_for_backward(x2d, ya, s1.mean, s1.invstd, g1, b1, h, wd, y2, m2, i2, g2, b2)
        ctx.R, ctx.pool_mode = R, pool_mode
        ctx.gt = _tgt(wa, g1, b1, w2, g2, b2)
        if want_out3:
            ctx.mark_non_differentiable(out)
            ctx.set_materialize_grads(False)
            return x3_handle(out), out
        return out

    @staticmethod
    def backward(ctx, dout, _d3=None):
        x2d, ya, m1, i1, g1, b1, h, wd, y2, m2, i2, g2, b2 = ctx.saved_tensors
        twa, tg1, tb1, tw2, tg2, tb2 = ctx.gt
        R = ctx.R
        if twa is not None or tw2 is not None:
            _plain_writer()
        dz = H.pool_bwd(dout.contiguous(), y2, R, m2, i2, g2, b2, ctx.pool_mode)
        dy2, dg2, db2 = _bn_bwd(dz, y2, R, m2, i2, g2, b2, 1, tg2, tb2, dx=dz)
        dw2 = H.conv_wgrad(dy2, h, 7, 2, 3, out=tw2, accumulate=tw2 is not None)
        dh = H.conv_dgrad(dy2, wd, 2, 3, h.shape[1])
        dya, dg1, db1 = _bn_bwd(dh, ya, R, m1, i1, g1, b1, 0, tg1, tb1, dx=dh)
        dwa = H.stem_conv_wgrad(dya, x2d, out=twa, accumulate=twa is not None, k=3, stride=1)
        return (None, None if twa is not None else dwa, dg1, db1, None if tw2 is not None else dw2, dg2, db2,
                None, None, None, None, None)


def _bn_bwd(dout, x, R, mean, invstd, gamma, beta, mode, tg, tb, out=None, dx=None, want_g=False, add=None, mask=None):
    """bn_bwd with optional direct gradient destinations; returns (dx, dgamma|None, dbeta|None[, g]).
    add = (tensor, channel offset): that slice is added to dx in the same pass (a concatenation's pass-through)."""
    direct = tg is not None and tb is not None
    defer = direct and _STEP['on']
    if direct and not defer:
        _plain_writer()
    dx, dg, db, g, ds = H.bn_bwd(dout, x, R, mean, invstd, gamma, beta, mode, out=out, want_g=want_g, dx=dx,
                                 dgamma=tg if direct else None, dbeta=tb if direct else None, accumulate=direct,
                                 defer_param_grads=defer, add=add, mask=mask)
    if defer:
        _STEP['pgrad'].append((ds, tg, tb))
    res = (dx, None if direct else dg, None if direct else db)
    return res + (g,) if want_g else res


def _bn_pgrad(ds, gamma, beta, tg, tb):
    """dgamma / dbeta from a BatchNorm backward's window sums ``ds``: queued for (or run as) the batched fold into the trainer's
    destinations (-> (None, None)), or returned."""
    if tg is not None and tb is not None:
        if _STEP['on']:
            _STEP['pgrad'].append((ds, tg, tb))
        else:
            _plain_writer()
            H.bn_param_grad_multi([(ds, tg, tb)], accumulate=True)
        return None, None
    dg, db = torch.empty_like(gamma), torch.empty_like(beta)
    H.bn_param_grad_multi([(ds, dg, db)], accumulate=False)
    return dg, db


def _wgrad(dy, x, k, stride, pad, tw, extra=None):
    """Weight gradient of a conv: queued for the step's batched launch (a trainer's gradient destination ``tw``), or run now.
    ``extra``: the dense-block operand forms of H.conv_wgrad_multi (x recomputed as relu(norm(x)), dy at half resolution);
    dy / x may then be channel slices of pitched buffers."""
    if tw is not None and _STEP['on']:
        _STEP['wgrad'].append((dy, x, k, stride, pad, tw, extra))     # launched with all the others by flush_backward()
        return None
    if tw is not None:
        _plain_writer()
    if H.WGRAD_BF16 or H.act_dtype() == 'bf16' or H.is_x3(dy) or extra or not (dy.is_contiguous() and x.is_contiguous()):
        # (the bf16-pipe kernels, the operand forms and pitched operands exist in the batched form only)
        (slab,) = H.conv_wgrad_multi([(dy, x, k, stride, pad, extra)])
        co, ci = (dy.shape[2] * 16, x.shape[2] * 16) if H.is_x3(dy) else (dy.shape[2], x.shape[2])
        dw = tw if tw is not None else torch.empty((co, ci, k), device=x.device, dtype=torch.float32)
        H.wgrad_reduce_multi([(slab, dw)], accumulate=tw is not None)
        return None if tw is not None else dw
    dw = H.conv_wgrad(dy, x, k, stride, pad, out=tw, accumulate=tw is not None)
    return None if tw is not None else dw


class BasicBlockFunction(Function):
    """conv3(s) -> BN -> ReLU -> conv3 -> BN -> (+ identity | BN(conv1x1(s))) -> ReLU.
    reference models/resnet.py:24-40 (BasicBlock.forward), :123-131 (downsample).

    Conv arithmetic 'f32x3p' (the x3 flow above): ``x3`` is the block input in the x3 format when its conv1 is a k3 s1 conv
    or the block is a stride-2 entry whose two input convs run as H.conv_x3p_s2_fwd (``x`` is then only the autograd handle),
    ``want_out3`` asks for the output in the x3 format (the next block reads one); the hidden activation h1 and the gradients
    in front of the data / weight-gradient convs (dy2, dy1, the downsample branch's dyd) take the x3 format whenever the
    shape has the store forms.  Returns ``out`` or ``(handle, out3)``."""

    @staticmethod
    def forward(ctx, x, w1, g1, b1, w2, g2, b2, wd, gd, bd, stride, R, st1, st2, std, x3=None, want_out3=False, pool_out=False,
                split_dx=False):
        # split_dx (an identity block BEHIND another BasicBlock, inside a training step): the input gradient conv1-dgrad(dy1) +
        # g is handed back as its two terms -- the conv output is the returned gradient, g waits in _STEP['dout2'] under its
        # address -- and the block in front sums them while its bn2 backward loads them (H.bn_bwd_two / bn_bwd_pair(dout2)):
        # an accumulating conv epilogue costs 4 ... 8 us a launch more than a plain one
        ctx.split_dx = bool(split_dx) and wd is None and x3 is None
        # pool_out (the LAST block in front of a head that pools its map, CNNLinearNetwork.forward_loss): the block's output
        # is never stored -- bn2 + residual + ReLU hand over the pooled features (rows, C) float (H.bn_fwd_pool), and the
        # backward takes their gradient (H.bn_bwd_pool): two 18 MB passes and the pooling launch less at B = 64
        in3 = x3 is not None
        bf16_pair = wd is not None and stride == 2 and x.shape[1] % 2 == 0 and \
            _is_wino(w1, stride, 1) == 16 and _is_wino(wd, stride, 0) == 16
        pair = bf16_pair or (wd is not None and stride == 2 and not _is_wino(w1, stride, 1))
        s2x = in3 and stride == 2
        # conv dtype bf16, a stride-1 block: bn1 has no pass of its own (H.conv3_bf16_bn at both ends); a window must cover a
        # 128-position tile and its halo
        fuse1 = _BN1_FUSED and not in3 and stride == 1 and R * x.shape[1] >= 130 and x.shape[0] % R == 0 and \
            _is_wino(w1, 1, 1) == 16 and _is_wino(w2, 1, 1) == 16 and \
            H.bn_single_pass(x.shape[0] // R, R * x.shape[1], w1.shape[0])
        ctx.fuse1 = fuse1
        ctx.bf16_pair = bf16_pair
        if s2x:           # the stride-2 block entry on the pre-split input: conv1 and the downsample conv in one launch
            if not s2_x3_ok(w1, wd, x3.shape[1]):
                raise ValueError('x3 input handed to a stride-2 block whose shape has no x3 kernels')
            y1, yd = H.conv_x3p_s2_fwd(x3, _pack(w1, 49)[2], _pack(wd, 49)[2])
            pair = True
        elif in3:         # k3 s1 conv on the pre-split input
            y1 = _conv_fwd(x3, w1, 1, 1)
        elif fuse1:       # conv dtype bf16, stride 1: the statistics records of y1 come out of the conv's epilogue
            y1, rec1 = H.conv3_bf16_bn(x, _pack(w1, 16)[2], R, want_records=True)
        elif bf16_pair:   # conv dtype bf16: the same shared launch on the bf16 kernel
            y1, yd = H.conv_fwd_bf16_s2(x, _pack(w1, 16)[2], _pack(wd, 16)[2])
        elif pair:    # the stride-2 conv and the 1x1 downsample read the same input: one launch
            y1, yd = H.conv_fwd_multi([(x, _pack(w1, False)[0], stride, 1), (x, _pack(wd, False)[0], stride, 0)])
        else:
            y1 = _conv_fwd(x, w1, stride, 1)
        mid3 = x3_block_ok(y1.shape[0], y1.shape[1], y1.shape[2], R) and _is_wino(w2, 1, 1) == 49
        if (in3 or want_out3) and not mid3:
            raise ValueError('x3 input / output asked of a block whose shape has no x3 store forms')
        s1 = _Stats()
        # the two BatchNorms behind the shared conv launch (the downsample's and bn1) in ONE launch when the shape has the
        # single-pass geometry: two ~10 us latency chains side by side instead of one after the other
        bn_pair = pair and not mid3 and not fuse1 and _BN_PAIR and std.eps == st1.eps and \
            y1.shape[0] % R == 0 and H.bn_single_pass(y1.shape[0] // R, R * y1.shape[1], y1.shape[2])
        ctx.bn_pair = bn_pair
        if pair:
            sd = _Stats()
        if bn_pair:
            (res, sd.mean, sd.invstd, _), (h1, s1.mean, s1.invstd, _) = H.bn_fwd_pair(
                [(yd, gd, bd, False, None, False), (y1, g1, b1, True, None, False)], R, st1.eps)
            _running(yd, R, sd, std)
            _running(y1, R, s1, st1)
        elif pair:        # the downsample BatchNorm first: yd is still in L2 / MALL right after the shared launch
            res = _bn_apply(yd, R, sd, std, gd, bd, False)   # (forking it beside bn1 / conv2 on another stream was measured slower)
        if fuse1:         # bn1 + ReLU applied while conv2 stages its operand: statistics from the records conv1's epilogue wrote
            wn_ = y1.shape[0] // R
            s1.mean = torch.empty((wn_, y1.shape[2]), device=y1.device, dtype=torch.float32)
            s1.invstd = torch.empty_like(s1.mean)
            y2 = H.conv3_bf16_bn(y1, _pack(w2, 16)[2], R, rec=rec1, mean=s1.mean, invstd=s1.invstd, gamma=g1, beta=b1, eps=st1.eps)
            _running(y1, R, s1, st1)
            h1 = y1                                     # (placeholder in the saved list: the backward rebuilds h1 for the weight gradient)
            if DECISION_TAP is not None:
                _tap(H.bn_fwd(y1, R, g1, b1, relu=True, eps=st1.eps)[0])
        else:
            if not bn_pair:
                h1 = _bn_apply_x(y1, R, s1, st1, g1, b1, True) if mid3 else _bn_apply(y1, R, s1, st1, g1, b1, True)
            _tap(h1)
            y2 = _conv_fwd(h1, w2, 1, 1)
        s2 = _Stats()
        if wd is not None:
            if not pair:
                yd = _conv_fwd(x, wd, stride, 0)
                sd = _Stats()
                res = _bn_apply(yd, R, sd, std, gd, bd, False)
            md, idd = sd.mean, sd.invstd
        else:
            yd = md = idd = None
            res = x3 if in3 else x
        ctx.pool_out = bool(pool_out)
        if pool_out:
            if mid3 or want_out3 or wd is not None or not H.bn_pool_ok(y2, R):
                raise ValueError('pool_out: an identity block on a map with the pooled BatchNorm form (pool_out_ok)')
            out, s2.mean, s2.invstd, s2.mask = H.bn_fwd_pool(y2, R, g2, b2, res=res, eps=st2.eps)
            _running(y2, R, s2, st2)
            if DECISION_TAP is not None:
                _tap(H.bn_fwd(y2, R, g2, b2, relu=True, res=res, eps=st2.eps)[0])
        elif mid3:        # x3 residual and / or x3 output: the store forms (always with the ReLU bit mask)
            out = _bn_apply_x(y2, R, s2, st2, g2, b2, True, res=res, want_mask=True, out_x3=want_out3)
            _tap(out)
        else:
            out = _bn_apply(y2, R, s2, st2, g2, b2, True, res=res, want_mask=True)
            _tap(out)
        m1, i1, m2, i2 = s1.mean, s1.invstd, s2.mean, s2.invstd
        ctx.relu_mask = s2.mask     # 8 bytes per thread instead of re-reading `out` for its sign (None: two-stage geometry)
        ctx.has_ds, ctx.in3, ctx.mid3, ctx.s2x = wd is not None, in3, mid3, s2x
        ctx.stride, ctx.R, ctx.lin = stride, R, x.shape[1]
        ctx.gt = _tgt(w1, g1, b1, w2, g2, b2, wd, gd, bd)
        # (the float block output is only kept when the backward reads it for its sign: no bit mask)
        saved = [x3 if in3 else x, w1, g1, b1, w2, g2, b2, y1, m1, i1, h1, y2, m2, i2, out if ctx.relu_mask is None else m2]
        if ctx.has_ds:
            saved += [wd, gd, bd, yd, md, idd]
        ctx.save_for_backward(*saved)
        if want_out3:
            ctx.mark_non_differentiable(out)
            ctx.set_materialize_grads(False)
            return x3_handle(out), out
        return out

    @staticmethod
    def backward(ctx, dout, _d3=None):
        s = ctx.saved_tensors
        x, w1, g1, b1, w2, g2, b2, y1, m1, i1, h1, y2, m2, i2, out = s[:15]
        tw1, tg1, tb1, tw2, tg2, tb2, twd, tgd, tbd = ctx.gt
        R, stride, lin = ctx.R, ctx.stride, ctx.lin
        in3, mid3 = ctx.in3, ctx.mid3
        # the second term of this block's output gradient, if the block behind left one (its split_dx)
        d2 = _STEP['dout2'].pop(dout.data_ptr(), None) if _STEP['on'] and _STEP.get('dout2') else None
        dout = dout.contiguous()
        # relu + residual add + bn2
        bwd_pair = ctx.has_ds and ctx.bn_pair and ctx.relu_mask is not None and not mid3 and not ctx.s2x
        two = d2 is not None and not ctx.pool_out and not mid3 and ctx.relu_mask is not None and H.bn_two_ok(y2, R)
        if d2 is not None and not two:      # (a shape without the two-term kernels: sum them here)
            dout = dout + d2
        if bwd_pair:      # bn2 and the downsample's BatchNorm take the same masked gradient: one launch, no g tensor
            wd, gd, bd, yd, md, idd = s[15:]
            (dy2, ds2), (dyd, dsd) = H.bn_bwd_pair(dout, [(y2, m2, i2, g2, b2, None), (yd, md, idd, gd, bd, None)], R, ctx.relu_mask,
                                                   dout2=d2 if two else None)
            dg2, db2 = _bn_pgrad(ds2, g2, b2, tg2, tb2)
            dgd, dbd = _bn_pgrad(dsd, gd, bd, tgd, tbd)
            g = None
        elif two:         # bn2 of an identity block with the two-term upstream gradient
            dy2, g, ds2 = H.bn_bwd_two(dout, d2, y2, R, m2, i2, g2, b2, ctx.relu_mask, want_g=True)
            dg2, db2 = _bn_pgrad(ds2, g2, b2, tg2, tb2)
        elif ctx.pool_out:  # dout = the gradient of the pooled features (rows, C)
            dy2, g, ds2 = H.bn_bwd_pool(dout, y2, R, m2, i2, g2, b2, ctx.relu_mask, want_g=True)
            dg2, db2 = _bn_pgrad(ds2, g2, b2, tg2, tb2)
        elif mid3:        # dy2 feeds the k3 s1 data-gradient and weight-gradient convs: stored pre-split
            dy2, dg2, db2, g = _bn_bwd_x(dout, y2, R, m2, i2, g2, b2, 3, tg2, tb2, want_g=True, mask=ctx.relu_mask)
        else:
            dy2, dg2, db2, g = _bn_bwd(dout, y2, R, m2, i2, g2, b2, 2, tg2, tb2, out=out, want_g=True, mask=ctx.relu_mask)
        if ctx.has_ds and not bwd_pair:    # the downsample BatchNorm's backward right away: g is still cache-resident
            wd, gd, bd, yd, md, idd = s[15:]
            if ctx.s2x:       # x3: it feeds the stride-2 data-gradient and weight-gradient kernels
                dyd, dgd, dbd = _bn_bwd_x(g, yd, R, md, idd, gd, bd, 0, tgd, tbd)
            else:
                dyd, dgd, dbd = _bn_bwd(g, yd, R, md, idd, gd, bd, 0, tgd, tbd, dx=g)
        if ctx.fuse1:     # h1 was never stored: the BatchNorm backward rebuilds it (same fused multiply-add) for conv2's weight gradient
            dh1 = _conv_dgrad(dy2, w2, 1, 1, y1.shape[1])
            h1 = torch.empty_like(y1)
            ds1 = H.bn_bwd_ss(dh1, y1, R, m1, i1, g1, b1, 1, dh1, hout=h1)
            dy1 = dh1
            dg1, db1 = _bn_pgrad(ds1, g1, b1, tg1, tb1)
            dw2 = _wgrad(dy2, h1, 3, 1, 1, tw2)
        else:
            dw2 = _wgrad(dy2, h1, 3, 1, 1, tw2)
            dh1 = _conv_dgrad(dy2, w2, 1, 1, y1.shape[1])
            if in3:       # conv1 is a k3 s1 conv on x3 operands too
                dy1, dg1, db1 = _bn_bwd_x(dh1, y1, R, m1, i1, g1, b1, 1, tg1, tb1)
            else:
                dy1, dg1, db1 = _bn_bwd(dh1, y1, R, m1, i1, g1, b1, 1, tg1, tb1, dx=dh1)
        dw1 = _wgrad(dy1, x, 3, stride, 1, tw1)
        if ctx.has_ds:
            dwd = _wgrad(dyd, x, 1, stride, 0, twd)
            if ctx.s2x:
                dx = H.conv_x3p_s2_dgrad(dy1, _pack(w1, 49)[3], dyd, _pack(wd, 49)[3])
            elif stride == 2 and not _is_wino(w1, stride, 1):
                dx = H.conv_dgrad_s2_pair(dy1, _pack(w1, False)[1], dyd, _pack(wd, False)[1], lin)
            elif ctx.bf16_pair and _BF16_DGRAD_PAIR:    # conv dtype bf16: the same shared launch (two sources in the even problem)
                dx = H.conv_dgrad_bf16_s2_pair(dy1, _pack(w1, 16)[3], dyd, _pack(wd, 16)[3], lin)
            else:
                dx = _conv_dgrad(dy1, w1, stride, 1, lin)
                _conv_dgrad(dyd, wd, stride, 0, lin, out=dx, accumulate=True)
        else:
            dwd = dgd = dbd = None
            if ctx.split_dx and _STEP['on'] and not in3 and g is not None:
                dx = _conv_dgrad(dy1, w1, stride, 1, lin)                       # the conv term; the identity term g travels beside it
                _STEP['dout2'][dx.data_ptr()] = g
            else:
                dx = _conv_dgrad(dy1, w1, stride, 1, lin, out=g, accumulate=True)   # identity grad + conv path
        return dx, dw1, dg1, db1, dw2, dg2, db2, dwd, dgd, dbd, None, None, None, None, None, None, None, None, None


_STEM_TAIL = True         # inside a training step the stem's last weight-gradient fold rides on the tail launch
_BF16_DGRAD_PAIR = True   # conv dtype bf16: a stride-2 block entry's two data gradients in one launch (tests compare with the two launches)
_BN_PAIR = True           # a block entry's two independent BatchNorms (forward: bn1 | downsample; backward: bn2 | downsample) share a launch
_BN1_FUSED = os.environ.get('DA_BN1_FUSED', '0') == '1'   # conv dtype bf16: bn1 of a stride-1 residual block without a pass of its own -- measured slower (profiles/r04_bf16_bn1_fusion.txt): opt-in
_DENSE_BLOCK = os.environ.get('DA_DENSE_BLOCK', '1') != '0'   # 0: the per-layer Functions below (the path shapes without the block kernels take)


def dense_block_ok(rows, R, l, c0, growth, n_layers, mid, tail_out, use_drop):
    """Whether a _DenseBlock of ``n_layers`` layers on (rows, l, c0) inputs, followed by a transition to ``tail_out`` channels
    (0: by norm5), runs as ONE DenseBlockFunction: float storage, the single-pass BatchNorm geometry for every channel count
    of the block, an even length in front of a transition (its pooling is folded in front of its conv), channel counts the
    kernels tile (growth % 32, 1x1 outputs % 64), and -- with dropout on -- the Winograd growth conv (its epilogue drops)."""
    if not (_DENSE_BLOCK and _CONV_DTYPE != 'bf16' and growth % 32 == 0 and c0 % 32 == 0 and mid % 64 == 0 and n_layers >= 1):
        return False
    if tail_out and (l % 2 or tail_out % 64 or R * (l // 2) < 64):
        return False
    if use_drop and not _WINOGRAD:
        return False
    return H.dense_fused_ok(rows, R, l, sorted(set([c0 + k * growth for k in range(n_layers + 1)] + [mid, growth])))


class DenseBlockFunction(Function):
    """A whole _DenseBlock AND the unit that consumes it -- its _Transition, or norm5 (+ReLU) behind the last block -- on ONE
    pitched buffer (reference models/densenet.py:18-44 _DenseLayer, :46-66 _DenseBlock, :68-81 _Transition, :146,181-182):

    * ``buf`` (rows, L, Cb): the block's input in its first C0 channels (written there by the stem / the previous
      transition); every layer's growth conv writes its G new channels at their offset, with F.dropout in its epilogue
      -- torch.cat and the dropout / slice kernels are gone;
    * ONE statistics table (2, W, Cb) per block: a channel's per-window mean / invstd are computed once, when the channel
      is written, and reused by every later norm1 and by the transition norm (statistics depend on the data only; gamma and
      beta differ);
    * h = relu(norm1(x)) is never stored: the 1x1 conv applies it while it stages its operand (H.conv1x1_bn), its weight
      gradient recomputes it (H.conv_wgrad_multi xform), the BatchNorm backward takes the ReLU decision from the same fused
      multiply-add (H.bn_bwd_ss);
    * backward: ONE gradient buffer (rows, L, Cb); each norm1 backward accumulates into its first Ck channels in place and
      applies the previous layer's dropout mask to the G channels it is the last to touch;
    * a transition's AvgPool1d(2,2) is folded in FRONT of its (linear) 1x1 conv: half the products, no full-resolution conv
      output, and the next block's first C0' channels land in ITS buffer -- which is what this Function returns.

    params: per layer (norm1.weight, norm1.bias, conv1.weight, norm2.weight, norm2.bias, conv2.weight), then the tail's:
    (norm.weight, norm.bias, conv.weight) for a transition (``tail_cb`` = the next block's buffer width) or
    (norm5.weight, norm5.bias) for the final norm (``tail_cb`` = 0; ``tail_relu``: F.relu behind it or the bare map)."""

    @staticmethod
    def forward(ctx, buf, c0_rec, R, c0, growth, n_layers, drop_p, seed, salt0, eps, tail_cb, tail_relu, *params):
        # Who computes a channel's statistics: the kernel that WRITES the channel, from its epilogue, as per-(tile, window)
        # records (H.stat_records) that the next 1x1 conv merges in its prologue and publishes to the block's table -- the
        # growth conv for its G new channels, the previous transition's conv for this block's first C0 (``c0_rec``).  Only
        # where no such producer exists (the stem's output, a growth conv off the Winograd kernel, windows shorter than a
        # record tile) a statistics-only pass runs (H.bn_stats_fused).
        rows, l, cb = buf.shape
        w_ = rows // R
        G = growth
        stats = torch.empty((2, w_, cb), device=buf.device, dtype=torch.float32)
        mean_t, invstd_t = stats[0], stats[1]
        pend = None                                     # (records, first channel, units, units per window) not yet in the table
        if c0_rec is not None:
            pend = (c0_rec, 0, rows * l, R * l)
        else:
            H.bn_stats_fused(buf[:, :, :c0], R, mean_t[:, :c0], invstd_t[:, :c0], eps)
        pl = (l + 1) // 2
        rec_ok = R * pl >= 64
        drop = drop_p > 0
        keep, fused2 = [], []
        for k in range(n_layers):
            g1, b1, w1, g2, b2, w2 = params[6 * k:6 * k + 6]
            ck = c0 + k * G
            xk = buf[:, :, :ck]
            mid = w1.shape[0]
            y1 = torch.empty((rows, l, mid), device=buf.device, dtype=torch.float32)
            code = _is_wino(w2, 1, 1)
            # norm2 -> relu2 -> conv2 as ONE kernel (the Winograd growth conv normalises while it stages, from the records
            # the 1x1 conv's epilogue hands over): h2 = relu(norm2(y1)) is then never stored either -- ``fuse2``
            fuse2 = code == 4 and rec_ok and mid <= 128 and R * l >= 64
            r1 = H.conv1x1_bn(xk, w1, R, mean_t[:, :ck], invstd_t[:, :ck], g1, b1, y1, pend=pend, eps=eps, want_records=fuse2)
            pend = None
            if DECISION_TAP is not None:
                _tap(H.bn_relu_ss(xk, R, mean_t[:, :ck], invstd_t[:, :ck], g1, b1))
            new = buf[:, :, ck:ck + G]
            consumed = k + 1 < n_layers or tail_cb            # (norm5 takes its own statistics of the whole buffer)
            dr = (seed, salt0 + k, drop_p) if drop else None
            if fuse2:
                m2 = torch.empty((w_, mid), device=buf.device, dtype=torch.float32)
                i2 = torch.empty_like(m2)
                r_ = H.conv3_winograd_bn(y1, _pack(w2, code)[2], R, r1[1], m2, i2, g2, b2, new, eps=eps, drop=dr,
                                         want_records=bool(consumed))
                if consumed:
                    pend = (r_[1], ck, rows * pl, R * pl)
                h2 = y1                                        # (placeholder in the saved list: the backward recomputes h2)
                if DECISION_TAP is not None:
                    _tap(H.bn_relu_ss(y1, R, m2, i2, g2, b2))
            else:
                h2, m2, i2 = H.bn_fwd(y1, R, g2, b2, relu=True, eps=eps)
                _tap(h2)
                if code == 4:
                    r_ = H.conv3_winograd(h2, _pack(w2, code)[2], out=new, drop=dr, stats_R=R if consumed and rec_ok else 0)
                    if consumed and rec_ok:
                        pend = (r_[1], ck, rows * pl, R * pl)
                elif code == 6:
                    H.conv3_winograd(h2, _pack(w2, code)[2], out=new)
                else:
                    H.conv_fwd(h2, _pack(w2, 0)[0], 1, 1, out=new)
            if consumed and pend is None:
                H.bn_stats_fused(new, R, mean_t[:, ck:ck + G], invstd_t[:, ck:ck + G], eps)
            keep += [y1, m2, i2, h2]
            fused2.append(fuse2)
        out_rec = None
        if tail_cb:
            gt, bt, wt = params[6 * n_layers:6 * n_layers + 3]
            out = torch.empty((rows, l // 2, tail_cb), device=buf.device, dtype=torch.float32)
            r_ = H.conv1x1_bn(buf, wt, R, mean_t, invstd_t, gt, bt, out[:, :, :wt.shape[0]], pool=True, pend=pend, eps=eps,
                              want_records=True)
            out_rec = r_[1]                             # the next block's first channels: their statistics, from this epilogue
            if DECISION_TAP is not None:
                _tap(H.bn_relu_ss(buf, R, mean_t, invstd_t, gt, bt))
            keep += [mean_t, invstd_t]          # (placeholders: the tail's statistics are the table's)
        else:
            g5, b5 = params[6 * n_layers:6 * n_layers + 2]
            out, m5, i5 = H.bn_fwd(buf, R, g5, b5, relu=tail_relu, eps=eps)
            if tail_relu:
                _tap(out)
            keep += [m5, i5]
            ctx.tail_out = out if tail_relu else None      # (its sign is the backward's ReLU decision)
        ctx.cfg = (R, c0, G, n_layers, drop_p, salt0, tail_cb, tail_relu)
        ctx.fused2 = fused2
        ctx.gt = _tgt(*params)
        ctx.save_for_backward(buf, stats, seed if drop else stats, *keep, *params)
        if tail_cb:                                     # (next buffer, the records of its first channels)
            ctx.mark_non_differentiable(out_rec)
            ctx.set_materialize_grads(False)
            return out, out_rec
        return out

    @staticmethod
    def backward(ctx, dout, _drec=None):
        R, c0, G, n_layers, drop_p, salt0, tail_cb, tail_relu = ctx.cfg
        sv = ctx.saved_tensors
        buf, stats, seed = sv[0], sv[1], sv[2]
        keep = sv[3:3 + 4 * n_layers + 2]
        params = sv[3 + 4 * n_layers + 2:]
        tg = ctx.gt
        rows, l, cb = buf.shape
        mean_t, invstd_t = stats[0], stats[1]
        drop = drop_p > 0
        dout = dout.contiguous()
        dbuf = torch.empty_like(buf)
        grads = [None] * len(params)

        def fold(ds, gamma, beta, ig, ib):                  # dgamma / dbeta from a BatchNorm backward's window sums
            if tg[ig] is not None and tg[ib] is not None:
                if _STEP['on']:
                    _STEP['pgrad'].append((ds, tg[ig], tg[ib]))
                else:
                    _plain_writer()
                    H.bn_param_grad_multi([(ds, tg[ig], tg[ib])], accumulate=True)
            else:
                grads[ig], grads[ib] = torch.empty_like(gamma), torch.empty_like(beta)
                H.bn_param_grad_multi([(ds, grads[ig], grads[ib])], accumulate=False)

        last_drop = (seed, salt0 + n_layers - 1, drop_p, G) if drop else None     # the last layer's new channels
        pt = 6 * n_layers
        if tail_cb:
            gt_, bt_, wt = params[pt:pt + 3]
            dy = dout[:, :, :wt.shape[0]]
            grads[pt + 2] = _wgrad(dy, buf, 1, 1, 0, tg[pt + 2],
                                   {'xform': (mean_t, invstd_t, gt_, bt_, R), 'dy_half': True})
            dpool = H.conv_dgrad(dy, _pack(wt, 0)[1], 1, 0, l // 2)
            ds = H.bn_bwd_ss(dpool, buf, R, mean_t, invstd_t, gt_, bt_, 1, dbuf, half_dout=True, drop=last_drop)
        else:
            g5, b5 = params[pt:pt + 2]
            m5, i5 = keep[4 * n_layers], keep[4 * n_layers + 1]
            ds = H.bn_bwd_ss(dout, buf, R, m5, i5, g5, b5, 2 if tail_relu else 0, dbuf, drop=last_drop, out=ctx.tail_out)
            gt_, bt_ = g5, b5
        fold(ds, gt_, bt_, pt, pt + 1)
        for k in range(n_layers - 1, -1, -1):
            g1, b1, w1, g2, b2, w2 = params[6 * k:6 * k + 6]
            y1, m2, i2, h2 = keep[4 * k:4 * k + 4]
            ck = c0 + k * G
            dnew = dbuf[:, :, ck:ck + G]                    # (its dropout mask was applied by the kernel that wrote it last)
            f2 = ctx.fused2[k]
            grads[6 * k + 5] = _wgrad(dnew, y1 if f2 else h2, 3, 1, 1, tg[6 * k + 5], {'xform': (m2, i2, g2, b2, R)} if f2 else None)
            code = _is_wino(w2, 1, 1)
            if code in (4, 6):
                dh2 = H.conv3_winograd(dnew, _pack(w2, code)[3])
            else:
                dh2 = H.conv_dgrad(dnew, _pack(w2, 0)[1], 1, 1, l)
            if f2:      # the ReLU decision of the form the forward applied (fused multiply-add), in place
                ds2 = H.bn_bwd_ss(dh2, y1, R, m2, i2, g2, b2, 1, dh2)
                fold(ds2, g2, b2, 6 * k + 3, 6 * k + 4)
                dy1 = dh2
            else:
                dy1, grads[6 * k + 3], grads[6 * k + 4] = _bn_bwd(dh2, y1, R, m2, i2, g2, b2, 1, tg[6 * k + 3], tg[6 * k + 4], dx=dh2)
            xk, mk, ik = buf[:, :, :ck], mean_t[:, :ck], invstd_t[:, :ck]
            grads[6 * k + 2] = _wgrad(dy1, xk, 1, 1, 0, tg[6 * k + 2], {'xform': (mk, ik, g1, b1, R)})
            dh = H.conv_dgrad(dy1, _pack(w1, 0)[1], 1, 0, l)
            dxk = dbuf[:, :, :ck]
            ds = H.bn_bwd_ss(dh, xk, R, mk, ik, g1, b1, 1, dxk, add=dxk,
                             drop=(seed, salt0 + k - 1, drop_p, G) if drop and k > 0 else None)
            fold(ds, g1, b1, 6 * k, 6 * k + 1)
        return (dbuf,) + (None,) * 11 + tuple(grads)


class DenseLayerFunction(Function):
    """BN -> ReLU -> conv1x1 -> BN -> ReLU -> conv3 -> dropout -> cat([x, new], C).
    reference models/densenet.py:18-41 (_DenseLayer)."""

    @staticmethod
    def forward(ctx, x, g1, b1, w1, g2, b2, w2, R, st1, st2, drop_p, seed, salt):
        s1 = _Stats()
        h = _bn_apply(x, R, s1, st1, g1, b1, True)
        _tap(h)
        y1 = _conv_fwd(h, w1, 1, 0)
        s2 = _Stats()
        h2 = _bn_apply(y1, R, s2, st2, g2, b2, True)
        _tap(h2)
        m1, i1, m2, i2 = s1.mean, s1.invstd, s2.mean, s2.invstd
        new = _conv_fwd(h2, w2, 1, 1)
        out = H.concat2(x, new, drop=(seed, salt, drop_p) if drop_p > 0 else None)      # dropout rides on the concat
        ctx.R, ctx.drop_p, ctx.salt = R, drop_p, salt
        ctx.gt = _tgt(g1, b1, w1, g2, b2, w2)
        ctx.save_for_backward(x, g1, b1, w1, g2, b2, w2, m1, i1, h, y1, m2, i2, h2, seed)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, g1, b1, w1, g2, b2, w2, m1, i1, h, y1, m2, i2, h2, seed = ctx.saved_tensors
        tg1, tb1, tw1, tg2, tb2, tw2 = ctx.gt
        R = ctx.R
        cin = x.shape[2]
        dout = dout.contiguous()
        dnew = H.slice_channels(dout, cin, w2.shape[0], drop=(seed, ctx.salt, ctx.drop_p) if ctx.drop_p > 0 else None)
        dw2 = _wgrad(dnew, h2, 3, 1, 1, tw2)
        dh2 = _conv_dgrad(dnew, w2, 1, 1, h2.shape[1])
        dy1, dg2, db2 = _bn_bwd(dh2, y1, R, m2, i2, g2, b2, 1, tg2, tb2, dx=dh2)
        dw1 = _wgrad(dy1, h, 1, 1, 0, tw1)
        dh = _conv_dgrad(dy1, w1, 1, 0, h.shape[1])
        dx, dg1, db1 = _bn_bwd(dh, x, R, m1, i1, g1, b1, 1, tg1, tb1, dx=dh, add=(dout, 0))     # + pass-through half of the cat
        return dx, dg1, db1, dw1, dg2, db2, dw2, None, None, None, None, None, None


class TransitionFunction(Function):
    """BN -> ReLU -> conv1x1 -> AvgPool1d(2,2).  reference models/densenet.py:68-79 (_Transition)."""

    @staticmethod
    def forward(ctx, x, g, b, w, R, st):
        s_ = _Stats()
        h = _bn_apply(x, R, s_, st, g, b, True)
        _tap(h)
        m, i = s_.mean, s_.invstd
        y = _conv_fwd(h, w, 1, 0)
        out = H.avgpool_fwd(y, 2)
        ctx.R = R
        ctx.gt = _tgt(g, b, w)
        ctx.save_for_backward(x, g, b, w, m, i, h)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, g, b, w, m, i, h = ctx.saved_tensors
        tg, tb, tw = ctx.gt
        dy = H.avgpool_bwd(dout.contiguous(), x.shape[1], 2)
        dw = _wgrad(dy, h, 1, 1, 0, tw)
        dh = _conv_dgrad(dy, w, 1, 0, h.shape[1])
        dx, dg, db = _bn_bwd(dh, x, ctx.R, m, i, g, b, 1, tg, tb, dx=dh)
        return dx, dg, db, dw, None, None


class NormReluFunction(Function):
    """BN -> ReLU (densenet norm5 + F.relu, models/densenet.py:146,181-182); relu=False: the bare norm5 output
    that ``features(x)`` returns to the explainers (gradcam.py:45)."""

    @staticmethod
    def forward(ctx, x, g, b, R, st, relu=True):
        s_ = _Stats()
        out = _bn_apply(x, R, s_, st, g, b, relu)
        if relu:
            _tap(out)
        m, i = s_.mean, s_.invstd
        ctx.R, ctx.relu = R, relu
        ctx.gt = _tgt(g, b)
        ctx.save_for_backward(x, g, b, m, i)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, g, b, m, i = ctx.saved_tensors
        tg, tb = ctx.gt
        dx, dg, db = _bn_bwd(dout.contiguous(), x, ctx.R, m, i, g, b, 1 if ctx.relu else 0, tg, tb)
        return dx, dg, db, None, None, None


class GlobalAvgPoolFunction(Function):
    """AvgPool1d(7, stride=1) + view(rows, -1): (rows, C) on an L=7 map, (rows, C*(L-6)) on longer ones
    (resnet.py:112,159-160; densenet.py:167,183-184)."""

    @staticmethod
    def forward(ctx, x):
        ctx.lin, ctx.c = x.shape[1], x.shape[2]
        if ctx.lin > 7:             # seq_len > 224: sliding window, flattened channel-major like view(N, -1)
            return H.avgpool_slide_fwd(x, 7)
        return H.global_avgpool_fwd(x)

    @staticmethod
    def backward(ctx, dfeat):
        if ctx.lin > 7:
            return H.avgpool_slide_bwd(dfeat.contiguous(), ctx.lin, 7, ctx.c)
        return H.global_avgpool_bwd(dfeat.contiguous(), ctx.lin)


class Linear2Function(Function):
    """linear_final on the flattened (NB, F) block of every window
    (models/torch_cnn_linear_network.py:102,110-112)."""

    @staticmethod
    def forward(ctx, flat, w, bias):
        ctx.save_for_backward(flat, w)
        ctx.gt = _tgt(w, bias)
        return H.linear2_fwd(flat, w, bias)

    @staticmethod
    def backward(ctx, dlogits):
        flat, w = ctx.saved_tensors
        tw, tb = ctx.gt
        direct = tw is not None and tb is not None
        if direct:
            _plain_writer()
        dflat, dw, dbias = H.linear2_bwd(dlogits.contiguous(), flat, w, need_input=ctx.needs_input_grad[0],
                                         dw=tw if direct else None, dbias=tb if direct else None, accumulate=direct)
        return dflat, None if direct else dw, None if direct else dbias


class HeadLossFunction(Function):
    """AvgPool1d(7,1) + view(-1) + linear_final + BCEWithLogitsLoss of CNNLinearNetwork as ONE autograd node in two launches
    (H.head_fwd / head_bwd) instead of six: xmap (B * R, L, F) the breath block's last map, target (B, 2) -> (loss (1,), logits
    (B, 2)).  In a training step the logits and the loss are FILLED BY THE BACKWARD (they are the first thing it needs, and
    nothing reads them earlier: the forward kernel leaves the dot products as row-group partials); backward takes no upstream
    gradient into account beyond d(loss) = 1: what the trainer's ``loss.backward()`` means.
    reference models/resnet.py:112,159-160, densenet.py:167,183-184; torch_cnn_linear_network.py:102,110-112;
    train_ards_detector.py:161-173,530."""

    @staticmethod
    def forward(ctx, xmap, w, bias, target, R, grad_mode):
        # (grad mode is always off in here and needs_input_grad ignores no_grad: the caller passes torch.is_grad_enabled())
        # xmap (B * R, F) float: features the last block pooled already (BasicBlockFunction pool_out) -- the same two launches
        # on a "map" of one position, their gradient (B * R, F) float
        need_grad = grad_mode and any(ctx.needs_input_grad[:3])
        target = target.contiguous()
        ctx.pooled = xmap.dim() == 2
        if ctx.pooled:
            flat, part, logits, loss = H.head_flat_fwd(xmap.contiguous(), w, bias, target, R, finish=not need_grad)
        else:
            flat, part, logits, loss = H.head_fwd(xmap.contiguous(), w, bias, target, R, finish=not need_grad)
        ctx.save_for_backward(flat, part, w, bias, target, logits, loss)
        ctx.R, ctx.l = R, 1 if ctx.pooled else xmap.shape[1]
        ctx.gt = _tgt(w, bias)
        ctx.mark_non_differentiable(logits)
        ctx.set_materialize_grads(False)               # (or autograd fills a zero d(logits) every step: one more launch)
        return loss, logits

    @staticmethod
    def backward(ctx, _dloss, _dlogits=None):
        flat, part, w, bias, target, logits, loss = ctx.saved_tensors
        tw, tb = ctx.gt
        direct = tw is not None and tb is not None
        if ctx.pooled:
            dx, dw, db = H.head_flat_bwd(part, bias, target, flat, w, logits, loss, ctx.R, dw=tw if direct else None,
                                         dbias=tb if direct else None, accumulate=direct and _acc())
        else:
            dx, dw, db = H.head_bwd(part, bias, target, flat, w, logits, loss, ctx.R, ctx.l, dw=tw if direct else None,
                                    dbias=tb if direct else None, accumulate=direct and _acc())
        return dx, None if direct else dw, None if direct else db, None, None, None


def head_loss(xmap, w, bias, target, R):
    """(loss, logits) through HeadLossFunction; see its note on WHEN the two are filled."""
    return HeadLossFunction.apply(xmap, w, bias, target, R, torch.is_grad_enabled())


class WindowMeanFunction(Function):
    """torch.mean(outputs, dim=1) over the NB breaths of every window (CNNLinearToMean,
    models/torch_cnn_linear_network.py:25): (B*NB, F) -> (B, F)."""

    @staticmethod
    def forward(ctx, feat, nb):
        ctx.nb = nb
        b = feat.shape[0] // nb
        return H.avgpool_fwd(feat.view(b, nb, feat.shape[1]), nb).view(b, feat.shape[1])

    @staticmethod
    def backward(ctx, dout):
        d = dout.contiguous().view(dout.shape[0], 1, dout.shape[1])
        dx = H.avgpool_bwd(d, ctx.nb, ctx.nb)                       # (B, NB, F)
        return dx.view(dout.shape[0] * ctx.nb, dout.shape[1]), None


class WindowMedianFunction(Function):
    """torch.median(outputs, dim=1)[0] (the lower median) over the NB breaths (CNNLinearComprToRF,
    models/torch_cnn_linear_network.py:47): (B*NB, F) -> (B, F); the gradient goes to the selected breath."""

    @staticmethod
    def forward(ctx, feat, nb):
        out, idx = H.window_median_fwd(feat.contiguous(), nb)
        ctx.save_for_backward(idx)
        ctx.nb = nb
        return out

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        return H.window_median_bwd(dout.contiguous(), idx, ctx.nb), None


class LSTMFunction(Function):
    """nn.LSTM(F, H, num_layers=1, batch_first=True) over the NB breath features of every window
    (CNNLSTMNetwork, models/torch_cnn_lstm_combo.py:17,46).  feat (B*T, F) -> hs (B, T, H), hT, cT (1, B, H).
    The input projection and the weight / input gradients run on the conv GEMM kernels (1x1 conv = GEMM), the
    recurrence in da_lstm_fwd / da_lstm_bwd.  The initial state gets no gradient (the reference detaches it,
    train_ards_detector.py:848)."""

    @staticmethod
    def forward(ctx, feat, w_ih, w_hh, b_ih, b_hh, T, h0, c0):
        rows, f = feat.shape
        b, g4 = rows // T, w_ih.shape[0]
        x3 = feat.contiguous().view(rows, 1, f)
        gx = H.conv_fwd(x3, w_ih.view(1, g4, f), 1, 0)                      # W_ih is already the forward pack
        hs, cs, gates, ht, ct = H.lstm_fwd(gx.view(b, T, g4), w_hh, b_ih, b_hh, h0, c0)
        ctx.T = T
        ctx.gt = _tgt(w_ih, w_hh, b_ih, b_hh)
        ctx.save_for_backward(x3, w_ih, w_hh, hs, cs, gates, h0 if h0 is not None else hs.new_empty(0),
                              c0 if c0 is not None else hs.new_empty(0))
        ctx.mark_non_differentiable(ht, ct)
        return hs, ht.view(1, b, -1), ct.view(1, b, -1)

    @staticmethod
    def backward(ctx, dhs, _dht, _dct):
        x3, w_ih, w_hh, hs, cs, gates, h0, c0 = ctx.saved_tensors
        h0 = h0 if h0.numel() else None
        c0 = c0 if c0.numel() else None
        tih, thh, tbi, tbh = ctx.gt
        rows, _, f = x3.shape
        g4 = w_ih.shape[0]
        dgates, part = H.lstm_bwd(dhs.contiguous(), w_hh, hs, cs, gates, h0, c0)
        dg3 = dgates.view(rows, 1, g4)
        dfeat = H.conv_dgrad(dg3, H.repack_weight(w_ih.view(g4, f, 1), False, True)[1], 1, 0, 1)
        direct = tih is not None
        if direct:
            _plain_writer()
        dw_ih = H.conv_wgrad(dg3, x3, 1, 1, 0, out=tih.view(g4, f, 1) if direct else None, accumulate=direct)
        dw_hh = H.reduce_rows(part, out=thh if direct else None, accumulate=direct)
        db = H.reduce_rows(dgates.view(rows, g4), out=tbi if direct else None, accumulate=direct)
        if direct:
            H.reduce_rows(dgates.view(rows, g4), out=tbh, accumulate=True)
            return dfeat.view(rows, f), None, None, None, None, None, None, None
        return dfeat.view(rows, f), dw_ih.view(g4, f), dw_hh, db, db.clone(), None, None, None


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
