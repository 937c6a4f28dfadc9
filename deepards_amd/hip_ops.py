"""Thin Python wrappers over the C ABI (include/deepards_hip.h): shape checks on the host, raw device
pointers + the current HIP stream into the library.  No arithmetic happens here and there is no
fallback -- every function launches a hand-written gfx950 kernel or raises.

Activation layout "RLC": contiguous ``(rows, L, C)`` float32 CUDA tensors (channels last); a
BatchNorm window = ``R`` consecutive rows (reference models/torch_cnn_linear_network.py:108-113).
"""
import ctypes
import os

import torch

from . import _lib

_DEBUG_SYNC = os.environ.get('DEEPARDS_DEBUG_SYNC') == '1'      # debugging aid: synchronise after every launch


class HipError(RuntimeError):
    pass


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _chk(rc, name):
    if rc != 0:
        raise HipError('%s failed with code %d' % (name, rc))
    if _DEBUG_SYNC:                      # debugging aid: serialise every launch
        torch.cuda.synchronize()


# Storage type of the RLC activation / activation-gradient tensors (include/deepards_hip.h, da_act_t): float32, or
# bfloat16 after set_act_dtype('bf16') (BASELINE's bf16 configs).  Features, statistics, parameters stay float32.
ACT = torch.float32


def set_act_dtype(name):
    """'f32' or 'bf16': how every activation tensor between two kernels is stored from now on (process-wide; the
    library dispatches its activation kernels on it, the wrappers allocate with it)."""
    global ACT
    if name not in ('f32', 'bf16'):
        raise ValueError("activation dtype must be 'f32' or 'bf16'")
    _chk(_lib.lib().da_set_act_dtype(1 if name == 'bf16' else 0), 'da_set_act_dtype')
    ACT = torch.bfloat16 if name == 'bf16' else torch.float32


def act_dtype():
    return 'bf16' if ACT == torch.bfloat16 else 'f32'


def _rlc(t, name='tensor'):
    if not (t.is_cuda and t.dtype == ACT and t.dim() == 3 and t.is_contiguous()):
        raise ValueError('%s must be a contiguous %s CUDA (rows, L, C) tensor, got %s %s %s' %
                         (name, ACT, tuple(t.shape), t.dtype, t.device))
    return t


def _pv(t, name='tensor'):
    """A (rows, L, C) float32 CUDA activation whose C channels are a slice of a contiguous (rows, L, ld) buffer (a dense
    block's pitched buffer, or a plain contiguous tensor: ld == C).  -> the channel pitch ld."""
    ok = t.is_cuda and t.dtype == torch.float32 and t.dim() == 3 and t.shape[1] >= 1 and t.shape[2] >= 1
    if ok:
        ld = t.stride(1) if t.shape[1] > 1 else max(t.shape[2], t.stride(0) // max(t.shape[1], 1))
        ok = t.stride(2) == 1 and ld >= t.shape[2] and ld % 4 == 0 and t.data_ptr() % 16 == 0 and \
            (t.shape[0] == 1 or t.stride(0) == t.shape[1] * ld) and (t.shape[1] == 1 or t.stride(1) == ld)
    if not ok:
        raise ValueError('%s must be a float32 CUDA (rows, L, C) channel slice of a contiguous buffer, got %s strides %s %s' %
                         (name, tuple(t.shape), t.stride() if t.dim() == 3 else None, t.dtype))
    return ld


def _sv(t, w, c, name='statistics'):
    """A (W, C) float32 CUDA slice of a per-block statistics table (W, ldstat) -> ldstat."""
    if not (t.is_cuda and t.dtype == torch.float32 and tuple(t.shape) == (w, c) and t.stride(1) == 1 and
            (w == 1 or (t.stride(0) >= c and t.stride(0) % 4 == 0)) and t.data_ptr() % 16 == 0):
        raise ValueError('%s must be a (W, C) = (%d, %d) float32 CUDA slice of a (W, ld) table, got %s strides %s' %
                         (name, w, c, tuple(t.shape), t.stride()))
    return t.stride(0) if w > 1 else max(c, t.stride(0))


def _f32(t, name='tensor'):
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise ValueError('%s must be a contiguous float32 CUDA tensor' % name)
    return t


def _ints(vals):
    return (ctypes.c_int * len(vals))(*vals)


# ------------------------------------------------------------------------------------------------
# convolution
# ------------------------------------------------------------------------------------------------
def repack_weight(w, need_fwd=True, need_dgrad=False):
    """torch (Co,Ci,K) -> Wf (K,Co,Ci) and/or Wd (K,Ci,Co)."""
    _f32(w, 'w')
    co, ci, k = w.shape
    wf = torch.empty((k, co, ci), device=w.device, dtype=torch.float32) if need_fwd else None
    wd = torch.empty((k, ci, co), device=w.device, dtype=torch.float32) if need_dgrad else None
    _chk(_lib.lib().da_repack_conv_weight(_p(w), _p(wf), _p(wd), co, ci, k, _stream()), 'da_repack_conv_weight')
    return wf, wd


def conv_out_len(l, k, stride, pad):
    return (l + 2 * pad - k) // stride + 1


def conv_fwd(x, wf, stride, pad, out=None):
    """x (rows,L,Ci), wf packed (K,Co,Ci) -> (rows,Lo,Co).  K in {1,3} is one launch; a longer kernel (the k7 conv2 of
    ResNet(double_conv_first), resnet.py:92-93) runs as ceil(K/3) launches of <= 3 taps that accumulate into the output."""
    ldx = _pv(x, 'x') if ACT == torch.float32 else _rlc(x, 'x').shape[2]
    k, co, ci = wf.shape
    rows, l, c = x.shape
    if c != ci or ci % 32 or co % 32:
        raise ValueError('conv_fwd: unsupported shape x%s wf%s' % (tuple(x.shape), tuple(wf.shape)))
    lo = conv_out_len(l, k, stride, pad)
    if out is None:
        out = torch.empty((rows, lo, co), device=x.device, dtype=torch.float32)
    elif tuple(out.shape) != (rows, lo, co):
        raise ValueError('conv_fwd: bad out shape')
    ldy = _pv(out, 'out')
    so = [t - pad for t in range(k)]
    wt = list(range(k))
    for g in range(0, k, 3):
        _chk(_lib.lib().da_conv_gemm(_p(x), _p(wf), _p(out), rows, lo, l, ldx, ci, lo, ldy, co, 1, 0, stride, len(so[g:g + 3]),
                                     _ints(so[g:g + 3]), _ints(wt[g:g + 3]), 1 if g else 0, _stream()), 'da_conv_gemm(fwd)')
    return out


def wino_weights(w, transpose=False, points=4):
    """Winograd taps of a (Co, Ci, 3) conv weight: (points, Co, Ci) for the forward, (points, Ci, Co) (transpose) for
    the data gradient; points = 4: F(2,3), 6: F(4,3)."""
    co, ci, k = w.shape
    if k != 3 or not w.is_contiguous() or points not in (4, 6):
        raise ValueError('wino_weights: (Co, Ci, 3) contiguous weight expected, points 4 or 6')
    u = torch.empty((points, ci, co) if transpose else (points, co, ci), device=w.device, dtype=torch.float32)
    fn = _lib.lib().da_wino_weights if points == 4 else _lib.lib().da_wino4_weights
    _chk(fn(_p(w), _p(u), co, ci, 1 if transpose else 0, _stream()), 'da_wino_weights')
    return u


def stat_records(units, n, device):
    """An empty statistics-record buffer for ``units`` record units of ``n`` channels (include/deepards_hip.h,
    "Statistics records")."""
    return torch.empty((_lib.lib().da_stat_records_floats(units, n),), device=device, dtype=torch.float32)


def conv3_winograd(x, u, out=None, accumulate=False, drop=None, stats_R=0):
    """k3 s1 p1 conv of x (rows, L, C) with taps u from wino_weights -> (rows, L, N): (4, N, C) taps run
    F(2,3), (6, N, C) taps F(4,3).  x and out may be channel slices of pitched buffers (a dense block's);
    drop = (seed, salt, p): F.dropout on the result in the epilogue (F(2,3) only; the mask of dropout() on the contiguous
    (rows, L, N) tensor); stats_R > 0 (F(2,3) only): -> (out, records) with the statistics records of the output for
    windows of stats_R rows, written by the epilogue (units: output pairs)."""
    ldx = _pv(x, 'x') if ACT == torch.float32 else _rlc(x, 'x').shape[2]      # (bf16 storage: the library refuses, as before)
    rows, l, c = x.shape
    four, n, c2 = u.shape
    if four not in (4, 6) or c2 != c or c % 32 or n % 32:
        raise ValueError('conv3_winograd: unsupported shape x%s u%s' % (tuple(x.shape), tuple(u.shape)))
    if out is None:
        if accumulate:
            raise ValueError('accumulate needs out')
        out = torch.empty((rows, l, n), device=x.device, dtype=torch.float32)
    elif tuple(out.shape) != (rows, l, n):
        raise ValueError('conv3_winograd: bad out shape')
    ldy = _pv(out, 'out') if out.dtype == torch.float32 else n
    if (drop is not None and drop[2] > 0) or stats_R:
        if four != 4 or accumulate:
            raise ValueError('conv3_winograd: dropout / statistics records belong to the F(2,3) kernel, without accumulate')
        seed, salt, p = drop if drop is not None and drop[2] > 0 else (None, 0, 0.0)
        part = stat_records(rows * ((l + 1) // 2), n, x.device) if stats_R else None
        _chk(_lib.lib().da_conv3_winograd_drop(_p(x), _p(u), _p(out), rows, l, ldx, c, ldy, n, _p(seed), salt, p, _p(part),
                                               stats_R, _stream()), 'da_conv3_winograd_drop')
        return (out, part) if stats_R else out
    fn = _lib.lib().da_conv3_winograd if four == 4 else _lib.lib().da_conv3_winograd4
    _chk(fn(_p(x), _p(u), _p(out), rows, l, ldx, c, ldy, n, 1 if accumulate else 0, _stream()), 'da_conv3_winograd')
    return out


def conv3_winograd_bn(x, u, R, rec, mean, invstd, gamma, beta, out, eps=1e-5, drop=None, want_records=False):
    """The growth conv of a dense layer on the 1x1 conv's output x (rows, L, C <= 128, contiguous) with relu(norm2(x)) applied
    while x is staged: the statistics of x come from ``rec`` (the records conv1x1_bn(want_records=True) wrote) and are
    published to mean / invstd (W, C) for the backward; out: a (rows, L, N) channel slice of the block's buffer;
    drop = (seed, salt, p): F.dropout in the epilogue; want_records: -> (out, records of the output, units = pairs)."""
    _rlc32(x, 'x')
    rows, l, c = x.shape
    four, n, c2 = u.shape
    if four != 4 or c2 != c or c % 32 or c > 128 or n % 32 or tuple(out.shape) != (rows, l, n) or rows % R:
        raise ValueError('conv3_winograd_bn: unsupported shape x%s u%s out%s' % (tuple(x.shape), tuple(u.shape), tuple(out.shape)))
    w = rows // R
    if tuple(mean.shape) != (w, c) or tuple(invstd.shape) != (w, c) or not (mean.is_contiguous() and invstd.is_contiguous()):
        raise ValueError('conv3_winograd_bn: mean / invstd must be contiguous (W, C)')
    if rec.numel() != _lib.lib().da_stat_records_floats(rows * l, c):
        raise ValueError('conv3_winograd_bn: the records do not have %d positions of %d channels' % (rows * l, c))
    ldy = _pv(out, 'out')
    seed, salt, p = drop if drop is not None and drop[2] > 0 else (None, 0, 0.0)
    part = stat_records(rows * ((l + 1) // 2), n, x.device) if want_records else None
    _chk(_lib.lib().da_conv3_winograd_bn(_p(x), _p(u), _p(out), rows, l, c, ldy, n, R, _p(rec), _p(mean), _p(invstd),
                                         _p(_f32(gamma)), _p(_f32(beta)), eps, _p(seed), salt, p, _p(part), _stream()),
         'da_conv3_winograd_bn')
    return (out, part) if want_records else out


def pack_conv3_bf16(w):
    """(Co, Ci, 3) fp32 conv weight -> (wf (3, Co, Ci), wd (3, Ci, Co)) bf16 tap packs for conv3_bf16."""
    _f32(w, 'w')
    co, ci, k = w.shape
    if k != 3:
        raise ValueError('pack_conv3_bf16: (Co, Ci, 3) weight expected')
    wf = torch.empty((3, co, ci), device=w.device, dtype=torch.bfloat16)
    wd = torch.empty((3, ci, co), device=w.device, dtype=torch.bfloat16)
    _chk(_lib.lib().da_pack_conv3_bf16(_p(w), _p(wf), _p(wd), co, ci, _stream()), 'da_pack_conv3_bf16')
    return wf, wd


def conv3_bf16(x, wpk, out=None, accumulate=False):
    """k3 s1 p1 conv of x (rows, L, C) fp32 with bf16 taps wpk (3, N, C): bf16 products, fp32 sums -> (rows, L, N)."""
    _rlc(x, 'x')
    rows, l, c = x.shape
    three, n, c2 = wpk.shape
    if three != 3 or c2 != c or c % 32 or n % 64 or wpk.dtype != torch.bfloat16 or not wpk.is_contiguous():
        raise ValueError('conv3_bf16: unsupported shape x%s w%s' % (tuple(x.shape), tuple(wpk.shape)))
    if out is None:
        if accumulate:
            raise ValueError('accumulate needs out')
        out = torch.empty((rows, l, n), device=x.device, dtype=ACT)
    elif tuple(out.shape) != (rows, l, n):
        raise ValueError('conv3_bf16: bad out shape')
    _chk(_lib.lib().da_conv3_bf16(_p(x), _p(wpk), _p(out), rows, l, c, c, n, n, 1 if accumulate else 0, _stream()),
         'da_conv3_bf16')
    return out



def conv3_bf16_bn(x, wpk, R, rec=None, mean=None, invstd=None, gamma=None, beta=None, eps=1e-5, want_records=False):
    """conv3_bf16 with a BatchNorm (windows of R rows) folded into either end: rec (+ mean / invstd (W, C) tables to publish
    into, gamma, beta): x is a raw conv output whose statistics records are ``rec`` -- relu(norm(x)) is applied while x is
    staged; want_records: -> (out, records of the output as stored; units = positions)."""
    _rlc(x, 'x')
    rows, l, c = x.shape
    three, n, c2 = wpk.shape
    if three != 3 or c2 != c or c % 32 or n % 64 or wpk.dtype != torch.bfloat16 or not wpk.is_contiguous() or rows % R:
        raise ValueError('conv3_bf16_bn: unsupported shape x%s w%s' % (tuple(x.shape), tuple(wpk.shape)))
    L = _lib.lib()
    if rec is not None:
        w = rows // R
        if rec.numel() != L.da_stat_records_floats(rows * l, c) or tuple(mean.shape) != (w, c) or tuple(invstd.shape) != (w, c) \
                or not (mean.is_contiguous() and invstd.is_contiguous()):
            raise ValueError('conv3_bf16_bn: records / tables do not fit x%s' % (tuple(x.shape),))
    out = torch.empty((rows, l, n), device=x.device, dtype=ACT)
    part = stat_records(rows * l, n, x.device) if want_records else None
    _chk(L.da_conv3_bf16_bn(_p(x), _p(wpk), _p(out), rows, l, c, c, n, n, R, _p(rec), _p(mean), _p(invstd),
                            _p(_f32(gamma)) if gamma is not None else None, _p(_f32(beta)) if beta is not None else None, eps,
                            _p(part), _stream()), 'da_conv3_bf16_bn')
    return (out, part) if want_records else out


def is_x3(t):
    """An activation in the x3 format (exact three-term bf16 split, include/deepards_hip.h): (rows, L, C/16, 3, 16) bf16."""
    return t is not None and t.dtype == torch.bfloat16 and t.dim() == 5 and t.shape[3] == 3 and t.shape[4] == 16


def x3_empty(rows, l, c, device):
    if c % 16:
        raise ValueError('x3 format needs a channel count that is a multiple of 16')
    return torch.empty((rows, l, c // 16, 3, 16), device=device, dtype=torch.bfloat16)


def x3_split(x):
    """fp32 (rows, L, C) -> x3 (rows, L, C/16, 3, 16) bf16: h | m | l with h + m + l == x exactly."""
    _rlc32(x, 'x')
    rows, l, c = x.shape
    out = x3_empty(rows, l, c, x.device)
    _chk(_lib.lib().da_x3_split(_p(x), c, _p(out), rows * l, c, _stream()), 'da_x3_split')
    return out


def x3_merge(x3):
    """x3 (rows, L, C/16, 3, 16) -> fp32 (rows, L, C) = h + m + l (exact)."""
    if not (is_x3(x3) and x3.is_cuda and x3.is_contiguous()):
        raise ValueError('x3_merge: a contiguous x3 CUDA tensor expected')
    rows, l, g = x3.shape[:3]
    out = torch.empty((rows, l, g * 16), device=x3.device, dtype=torch.float32)
    _chk(_lib.lib().da_x3_merge(_p(x3), _p(out), g * 16, rows * l, g * 16, _stream()), 'da_x3_merge')
    return out


def conv3_x3p(x3, wpk, out=None, accumulate=False):
    """k3 s1 p1 conv of an x3 activation (rows, L, C/16, 3, 16) with the chunked split-bf16 pack wpk (N/64, C/16, 18, 64, 8)
    (repack_multi code 49): fp32-equivalent products on the bf16 matrix cores, fp32 sums -> (rows, L, N) fp32."""
    if not (is_x3(x3) and x3.is_cuda and x3.is_contiguous()):
        raise ValueError('conv3_x3p: a contiguous x3 CUDA tensor expected, got %s %s' % (tuple(x3.shape), x3.dtype))
    rows, l, g = x3.shape[:3]
    c = g * 16
    if wpk.dim() != 5 or wpk.shape[1] != g or tuple(wpk.shape[2:]) != (18, 64, 8) or wpk.dtype != torch.bfloat16 or \
            not wpk.is_contiguous():
        raise ValueError('conv3_x3p: unsupported shape x%s w%s' % (tuple(x3.shape), tuple(wpk.shape)))
    n = wpk.shape[0] * 64
    if out is None:
        if accumulate:
            raise ValueError('accumulate needs out')
        out = torch.empty((rows, l, n), device=x3.device, dtype=torch.float32)
    elif tuple(out.shape) != (rows, l, n) or out.dtype != torch.float32 or not out.is_contiguous():
        raise ValueError('conv3_x3p: bad out')
    _chk(_lib.lib().da_conv3_x3p(_p(x3), _p(wpk), _p(out), rows, l, c, n, n, 1 if accumulate else 0, _stream()),
         'da_conv3_x3p')
    return out


def _x3p_pack_ok(pk, nt, kc):
    return pk.dim() == 5 and tuple(pk.shape) == (nt, kc, 18, 64, 8) and pk.dtype == torch.bfloat16 and pk.is_contiguous()


def conv_x3p_s2_fwd(x3, w1pk, wdpk):
    """The stride-2 block entry on x3 operands: (y1, yd) = (conv k3 s2 p1, conv 1x1 s2) of the x3 activation
    (rows, Lin, C/16, 3, 16), Lin even, from ONE read of it.  w1pk / wdpk: forward packs of repack_multi code 49
    (N/64, C/16, 18, 64, 8), the 1x1 weights packed with K = 1 (tap 1 of the chunks)."""
    if not (is_x3(x3) and x3.is_cuda and x3.is_contiguous()):
        raise ValueError('conv_x3p_s2_fwd: a contiguous x3 CUDA tensor expected')
    rows, lin, g = x3.shape[:3]
    if lin % 2 or not _x3p_pack_ok(w1pk, w1pk.shape[0], g) or not _x3p_pack_ok(wdpk, w1pk.shape[0], g):
        raise ValueError('conv_x3p_s2_fwd: unsupported shape x%s w1%s wd%s' % (tuple(x3.shape), tuple(w1pk.shape), tuple(wdpk.shape)))
    n = w1pk.shape[0] * 64
    y1 = torch.empty((rows, lin // 2, n), device=x3.device, dtype=torch.float32)
    yd = torch.empty_like(y1)
    _chk(_lib.lib().da_conv_x3p_s2_fwd(_p(x3), _p(w1pk), _p(wdpk), _p(y1), _p(yd), rows, lin, g * 16, n, _stream()), 'da_conv_x3p_s2_fwd')
    return y1, yd


def conv_x3p_s2_dgrad(dy1_3, w1pk, dyd_3, wdpk, out=None):
    """dx (rows, 2 Lout, C) fp32 = the data gradients of the two convs of conv_x3p_s2_fwd, summed; dy1_3 / dyd_3 x3
    activations (rows, Lout, N/16, 3, 16), packs: the DATA-GRADIENT side of repack_multi code 49 (C/64, N/16, 18, 64, 8)."""
    for t in (dy1_3, dyd_3):
        if not (is_x3(t) and t.is_cuda and t.is_contiguous()):
            raise ValueError('conv_x3p_s2_dgrad: contiguous x3 CUDA tensors expected')
    rows, lout, g = dy1_3.shape[:3]
    if tuple(dyd_3.shape) != tuple(dy1_3.shape) or not _x3p_pack_ok(w1pk, w1pk.shape[0], g) or not _x3p_pack_ok(wdpk, w1pk.shape[0], g):
        raise ValueError('conv_x3p_s2_dgrad: unsupported shapes')
    c = w1pk.shape[0] * 64
    if out is None:
        out = torch.empty((rows, 2 * lout, c), device=dy1_3.device, dtype=torch.float32)
    elif tuple(out.shape) != (rows, 2 * lout, c) or out.dtype != torch.float32 or not out.is_contiguous():
        raise ValueError('conv_x3p_s2_dgrad: bad out')
    _chk(_lib.lib().da_conv_x3p_s2_dgrad(_p(dy1_3), _p(w1pk), _p(dyd_3), _p(wdpk), _p(out), rows, lout, g * 16, c, _stream()),
         'da_conv_x3p_s2_dgrad')
    return out


def _conv_bf16_multi(jobs):
    """jobs: [(x, wpk, out, lm, lsrc, ldst, dst_stride, dst_off, src_stride, src_off, wtap, accumulate)] in one call."""
    arr = (_lib.ConvJob * len(jobs))()
    for d, (x, wpk, out, lm, lsrc, ldst, dst_stride, dst_off, src_stride, so, wt, acc) in zip(arr, jobs):
        rows, _, c = x.shape
        n = wpk.shape[1]
        _conv_job(d, x, wpk, out, rows, lm, lsrc, c, c, ldst, n, n, dst_stride, dst_off, src_stride, so, wt, acc)
    _chk(_lib.lib().da_conv_bf16_multi(arr, len(jobs), _stream()), 'da_conv_bf16_multi')


def _check_bf16_pack(x, w16, name):
    k, n, c2 = w16.shape
    if k not in (1, 3) or c2 != x.shape[2] or c2 % 32 or n % 64 or w16.dtype != torch.bfloat16 or not w16.is_contiguous():
        raise ValueError('%s: unsupported shape x%s w%s' % (name, tuple(x.shape), tuple(w16.shape)))
    return k, n


def conv_fwd_bf16_s2(x, *packs):
    """Stride-2 forward with bf16 operands, one launch for all packs: wf16 (3, N, C) = k3 s2 p1 block head, (1, N, C) = k1
    s2 p0 downsample; x (rows, L, C) fp32 with L even -> (rows, L / 2, N) fp32 per pack (a single tensor for one pack)."""
    _rlc(x, 'x')
    rows, l, c = x.shape
    if l % 2 or not packs:
        raise ValueError('conv_fwd_bf16_s2: even length and at least one pack expected')
    jobs, outs = [], []
    for wf16 in packs:
        k, n = _check_bf16_pack(x, wf16, 'conv_fwd_bf16_s2')
        out = torch.empty((rows, l // 2, n), device=x.device, dtype=ACT)
        so, wt = ([-1, 0, 1], [0, 1, 2]) if k == 3 else ([0], [0])
        jobs.append((x, wf16, out, l // 2, l, l // 2, 1, 0, 2, so, wt, False))
        outs.append(out)
    _conv_bf16_multi(jobs)
    return outs[0] if len(outs) == 1 else outs


def conv_dgrad_bf16_s2(dy, wd16, l_in, out=None, accumulate=False):
    """Data gradient of the stride-2 convs with bf16 operands: wd16 (3, Ci, Co) (taps reversed, as the packs come) or
    (1, Ci, Co); dy (rows, l_in / 2, Co) -> dx (rows, l_in, Ci).  k3: even positions take tap 1, odd ones taps 0 / 2
    of the neighbouring outputs (two problems, one launch); k1: even positions only (odd ones zero unless accumulating)."""
    _rlc(dy, 'dy')
    rows, lo, co = dy.shape
    k, ci = _check_bf16_pack(dy, wd16, 'conv_dgrad_bf16_s2')
    if l_in != 2 * lo:
        raise ValueError('conv_dgrad_bf16_s2: l_in must be twice the output length')
    if out is None:
        if accumulate:
            raise ValueError('accumulate needs out')
        out = (torch.zeros if k == 1 else torch.empty)((rows, l_in, ci), device=dy.device, dtype=ACT)
    elif tuple(out.shape) != (rows, l_in, ci):
        raise ValueError('conv_dgrad_bf16_s2: bad out shape')
    elif k == 1 and not accumulate:
        out.zero_()
    if k == 3:       # wd16[t'] = w[..][2 - t']: dx[2j] = dy[j] w1;  dx[2j+1] = dy[j] w2 + dy[j+1] w0
        _conv_bf16_multi([(dy, wd16, out, lo, lo, l_in, 2, 0, 1, [0], [1], accumulate),
                          (dy, wd16, out, lo, lo, l_in, 2, 1, 1, [0, 1], [0, 2], accumulate)])
    else:
        _conv_bf16_multi([(dy, wd16, out, lo, lo, l_in, 2, 0, 1, [0], [0], accumulate)])
    return out


def conv_dgrad_bf16_s2_pair(dy1, wd1_16, dyd, wdd_16, l_in):
    """dx = dgrad(k3 s2 p1 conv, dy1) + dgrad(k1 s2 p0 downsample, dyd) of a stride-2 block entry with bf16 operands in ONE
    launch (conv_dgrad_s2_pair's form): the odd input positions take the conv's taps 0 / 2, the even ones the conv's tap 1 from
    dy1 and the downsample's tap from dyd as one contraction over two sources.  wd1_16 (3, Ci, Co), wdd_16 (1, Ci, Co)."""
    _rlc(dy1, 'dy1')
    _rlc(dyd, 'dyd')
    rows, lo, co = dy1.shape
    k1, ci = _check_bf16_pack(dy1, wd1_16, 'conv_dgrad_bf16_s2_pair')
    kd, ci2 = _check_bf16_pack(dyd, wdd_16, 'conv_dgrad_bf16_s2_pair')
    if (k1, kd) != (3, 1) or ci != ci2 or tuple(dyd.shape) != tuple(dy1.shape) or l_in != 2 * lo:
        raise ValueError('conv_dgrad_bf16_s2_pair: shapes dy1%s dyd%s' % (tuple(dy1.shape), tuple(dyd.shape)))
    out = torch.empty((rows, l_in, ci), device=dy1.device, dtype=ACT)
    a = (_lib.ConvJob * 2)()
    _conv_job(a[0], dy1, wd1_16, out, rows, lo, lo, co, co, l_in, ci, ci, 2, 1, 1, [0, 1], [0, 2], False)
    _conv_job(a[1], dy1, wd1_16, out, rows, lo, lo, co, co, l_in, ci, ci, 2, 0, 1, [0, 0], [1, 0], False,
              x2=dyd, w2=wdd_16, tap_split=1)
    _chk(_lib.lib().da_conv_bf16_multi(a, 2, _stream()), 'da_conv_bf16_multi(dgrad pair)')
    return out


def conv_dgrad(dy, wd, stride, pad, l_in, out=None, accumulate=False):
    """dy (rows,Lo,Co), wd packed (K,Ci,Co) -> dx (rows,l_in,Ci).  With accumulate the result is
    added into `out`; positions no tap reaches are left untouched (accumulate) or zeroed."""
    lddy = _pv(dy, 'dy') if ACT == torch.float32 else _rlc(dy, 'dy').shape[2]
    k, ci, co = wd.shape
    rows, lo, c = dy.shape
    if c != co or ci % 32 or co % 32:
        raise ValueError('conv_dgrad: unsupported shape')
    if out is None:
        if accumulate:
            raise ValueError('accumulate needs out')
        out = torch.empty((rows, l_in, ci), device=dy.device, dtype=torch.float32)
    elif tuple(out.shape) != (rows, l_in, ci):
        raise ValueError('conv_dgrad: bad out shape')
    ldo = _pv(out, 'out')
    L = _lib.lib()
    for r in range(stride):
        lm = (l_in - r + stride - 1) // stride           # positions l_in = stride*j + r
        if lm <= 0:
            continue
        taps = [t for t in range(k) if (r + pad - t) % stride == 0]
        if not taps:
            if not accumulate:
                out[:, r::stride, :].zero_()
            continue
        so = [(r + pad - t) // stride for t in taps]
        for g in range(0, len(taps), 3):                 # <= 3 taps per launch; later groups add to the first
            _chk(L.da_conv_gemm(_p(dy), _p(wd), _p(out), rows, lm, lo, lddy, co, l_in, ldo, ci, stride, r, 1,
                                len(taps[g:g + 3]), _ints(so[g:g + 3]), _ints(taps[g:g + 3]),
                                1 if (accumulate or g) else 0, _stream()), 'da_conv_gemm(dgrad)')
    return out


def _conv_job(d, x, w, y, rows, lm, lsrc, ldx, c, ldst, ldy, n, dst_stride, dst_off, src_stride, so, wt, accumulate,
              x2=None, w2=None, tap_split=0):
    d.x, d.w, d.y = x.data_ptr(), w.data_ptr(), y.data_ptr()
    d.x2, d.w2, d.tap_split = _p(x2), _p(w2), tap_split
    d.rows, d.Lm, d.Lsrc, d.ldx, d.C, d.Ldst, d.ldy, d.N = rows, lm, lsrc, ldx, c, ldst, ldy, n
    d.dst_stride, d.dst_off, d.src_stride, d.ntaps = dst_stride, dst_off, src_stride, len(so)
    for t in range(3):
        d.src_off[t] = so[t] if t < len(so) else 0
        d.wtap[t] = wt[t] if t < len(wt) else 0
    d.accumulate = 1 if accumulate else 0


def conv_fwd_multi(problems):
    """[(x, wf, stride, pad)] (<= 4, Co % 64 == 0) -> [y]: independent forward convs in ONE launch (a block's stride-2
    conv and its 1x1 downsample read the same input; alone each has 2-4 tiles per CU)."""
    if len(problems) > 4 or any(wf.shape[1] % 64 for _, wf, _, _ in problems):
        return [conv_fwd(x, wf, s_, p_) for x, wf, s_, p_ in problems]
    arr = (_lib.ConvJob * len(problems))()
    outs = []
    for d, (x, wf, stride, pad) in zip(arr, problems):
        _rlc(x, 'x')
        k, co, ci = wf.shape
        rows, l, c = x.shape
        if c != ci or k > 3 or ci % 32:
            raise ValueError('conv_fwd_multi: unsupported shape x%s wf%s' % (tuple(x.shape), tuple(wf.shape)))
        lo = conv_out_len(l, k, stride, pad)
        y = torch.empty((rows, lo, co), device=x.device, dtype=torch.float32)
        _conv_job(d, x, wf, y, rows, lo, l, c, ci, lo, co, co, 1, 0, stride, [t - pad for t in range(k)], list(range(k)), False)
        outs.append(y)
    _chk(_lib.lib().da_conv_gemm_multi(arr, len(problems), _stream()), 'da_conv_gemm_multi(fwd)')
    return outs


def conv_dgrad_s2_pair(dy1, wd1, dyd, wdd, l_in):
    """dx = dgrad(k3 s2 p1 conv, dy1) + dgrad(k1 s2 p0 downsample, dyd) of one block in ONE launch instead of three:
    the odd input positions (the conv's taps 0 and 2) and the even ones (the conv's tap 1 and the downsample's tap as
    one contraction over two sources) are two problems with the same work per tile."""
    _rlc(dy1, 'dy1')
    _rlc(dyd, 'dyd')
    k1, ci, co = wd1.shape
    kd, ci2, co2 = wdd.shape
    rows, lo, c = dy1.shape
    if (k1, kd) != (3, 1) or ci != ci2 or co != co2 or c != co or tuple(dyd.shape) != tuple(dy1.shape) or ci % 64 or l_in % 2:
        dx = conv_dgrad(dy1, wd1, 2, 1, l_in)
        return conv_dgrad(dyd, wdd, 2, 0, l_in, out=dx, accumulate=True)
    dx = torch.empty((rows, l_in, ci), device=dy1.device, dtype=torch.float32)
    lm = l_in // 2
    a = (_lib.ConvJob * 2)()
    # odd input positions 2j+1: taps t = 0, 2 of the k3 conv, source positions (1 + 1 - t) / 2 + j
    _conv_job(a[0], dy1, wd1, dx, rows, lm, lo, co, co, l_in, ci, ci, 2, 1, 1, [1, 0], [0, 2], False)
    # even input positions 2j: tap 1 of the k3 conv from dy1 AND the downsample's only tap from dyd, one contraction
    _conv_job(a[1], dy1, wd1, dx, rows, lm, lo, co, co, l_in, ci, ci, 2, 0, 1, [0, 0], [1, 0], False,
              x2=dyd, w2=wdd, tap_split=1)
    _chk(_lib.lib().da_conv_gemm_multi(a, 2, _stream()), 'da_conv_gemm_multi(dgrad)')
    return dx


def conv_wgrad(dy, x, k, stride, pad, out=None, accumulate=False, defer=False):
    """dW (Co,Ci,K) torch layout = sum_positions dy (x) x.  defer=True: only the split-K slabs are produced;
    returns (slab, splits, k, co, ci) for wgrad_reduce_multi."""
    _rlc(dy, 'dy')
    _rlc(x, 'x')
    rows, lo, co = dy.shape
    rows2, l, ci = x.shape
    if rows != rows2 or lo != conv_out_len(l, k, stride, pad) or ci % 32 or co % 32 or (k > 3 and defer):
        raise ValueError('conv_wgrad: unsupported shape')
    if k > 3:         # a kernel longer than 3 taps (resnet's k7 conv2, double_conv_first only): tap groups into slices
        if out is None:
            if accumulate:
                raise ValueError('accumulate needs out')
            out = torch.empty((co, ci, k), device=x.device, dtype=torch.float32)
        for g in range(0, k, 3):
            n = min(3, k - g)
            part = _conv_wgrad_taps(dy, x, [t - pad for t in range(g, g + n)], stride)
            if accumulate:
                out[:, :, g:g + n].add_(part)
            else:
                out[:, :, g:g + n].copy_(part)
        return out
    L = _lib.lib()
    nbytes = L.da_conv_wgrad_workspace(rows, lo, co, ci, k)
    ws = torch.empty((nbytes // 4,), device=x.device, dtype=torch.float32)
    so = [t - pad for t in range(k)]
    if defer:
        _chk(L.da_conv_wgrad(_p(dy), _p(x), None, _p(ws), rows, lo, lo, co, co, l, ci, ci, 1, 0, stride, k,
                             _ints(so), 0, _stream()), 'da_conv_wgrad')
        return ws, L.da_conv_wgrad_splits(rows, lo, co, ci, k), k, co, ci
    if out is None:
        if accumulate:
            raise ValueError('accumulate needs out')
        out = torch.empty((co, ci, k), device=x.device, dtype=torch.float32)
    _chk(L.da_conv_wgrad(_p(dy), _p(x), _p(out), _p(ws), rows, lo, lo, co, co, l, ci, ci, 1, 0, stride, k,
                         _ints(so), 1 if accumulate else 0, _stream()), 'da_conv_wgrad')
    return out


def _conv_wgrad_taps(dy, x, src_off, stride):
    """(Co, Ci, len(src_off)) weight gradient of the taps that read x at stride * j + src_off[t]."""
    rows, lo, co = dy.shape
    l, ci = x.shape[1], x.shape[2]
    n = len(src_off)
    L = _lib.lib()
    ws = torch.empty((L.da_conv_wgrad_workspace(rows, lo, co, ci, n) // 4,), device=x.device, dtype=torch.float32)
    out = torch.empty((co, ci, n), device=x.device, dtype=torch.float32)
    _chk(L.da_conv_wgrad(_p(dy), _p(x), _p(out), _p(ws), rows, lo, lo, co, co, l, ci, ci, 1, 0, stride, n,
                         _ints(src_off), 0, _stream()), 'da_conv_wgrad')
    return out


WINOGRAD_WGRAD = os.environ.get('DA_WINOGRAD', '1') != '0'     # 0: the direct fp32 kernels (as functional._WINOGRAD)
WGRAD_BF16 = False        # set by functional.set_conv_dtype('bf16'): k3 s1 weight gradients on the bf16 matrix cores
WINO4_WGRAD_MIN_C = 512  # channels from which the F(4,3) weight-gradient form replaces F(2,3) (the 512-channel stage, as the forward)


def conv_wgrad_multi(jobs, dws=None, accumulate=True):
    """jobs: [(dy, x, k, stride, pad)] -> [(slab, splits, k, co, ci)]: every weight-gradient GEMM of the list in
    one launch per tile shape (slabs only; reduce with wgrad_reduce_multi).  k3 s1 p1 jobs with 64-multiple
    channel counts take the Winograd F(2,3) form (F(4,3) from WINO4_WGRAD_MIN_C channels), or the bf16-operand kernel
    while WGRAD_BF16 is set.
    dws (one (co, ci, k) gradient destination per job, or None entries): the slab reductions are chained
    (da_conv_wgrad_multi_reduce) -- every launch of the call carries, as its first blocks, the reduction of the slabs the
    launch before it wrote; -> (slabs, reduced) with reduced[i] False for the jobs whose reduction the caller still owes
    (those of the call's last launch)."""
    if not jobs:
        return [] if dws is None else ([], [])
    L = _lib.lib()
    arr = (_lib.WgradJob * len(jobs))()
    plan = (ctypes.c_int * 4)()
    outs, caps = [], []
    for d, job in zip(arr, jobs):
        dy, x, k, stride, pad = job[:5]
        extra = job[5] if len(job) > 5 and job[5] else {}         # dense-block operand forms (da_wgrad_job.xform / dy_half)
        both_x3 = is_x3(dy) and is_x3(x)
        if both_x3:                                   # x3 operands (conv arithmetic 'f32x3p'): the split-bf16 kernels
            if not (dy.is_cuda and dy.is_contiguous() and x.is_contiguous() and (
                    (k == 3 and stride == 1 and pad == 1) or
                    (stride == 2 and x.shape[1] % 2 == 0 and ((k == 3 and pad == 1) or (k == 1 and pad == 0))))):
                raise ValueError('conv_wgrad_multi: x3 operands belong to k3 s1 p1 / k3 s2 p1 / k1 s2 p0 (even length) jobs')
            rows, lo, co = dy.shape[0], dy.shape[1], dy.shape[2] * 16
            rows2, l, ci = x.shape[0], x.shape[1], x.shape[2] * 16
            lddy, ldx = co, ci
        elif ACT == torch.float32:                    # float operands may be channel slices of pitched (dense-block) buffers
            lddy, ldx = _pv(dy, 'dy'), _pv(x, 'x')
            rows, lo, co = dy.shape
            rows2, l, ci = x.shape
        else:
            _rlc(dy, 'dy')
            _rlc(x, 'x')
            rows, lo, co = dy.shape
            rows2, l, ci = x.shape
            lddy, ldx = co, ci
        ldy_len = lo
        if extra.get('dy_half'):                      # dY at half resolution: position j of x reads dy[j / 2] / 2
            if k != 1 or stride != 1 or pad != 0 or l % 2 or lo * 2 != l:
                raise ValueError('conv_wgrad_multi: dy_half belongs to a 1x1 job whose dy has half the positions of x')
            lo = l
        if rows != rows2 or lo != conv_out_len(l, k, stride, pad) or k > 3 or ci % 32 or co % 32:
            raise ValueError('conv_wgrad_multi: unsupported shape')
        wino = 1 if (WINOGRAD_WGRAD and k == 3 and stride == 1 and pad == 1 and co % 64 == 0 and ci % 64 == 0) else 0
        if wino and min(co, ci) >= WINO4_WGRAD_MIN_C:
            wino = 6                                 # F(4,3) form: 6 contractions over quads instead of 8 over pairs
        if WGRAD_BF16 and co % 64 == 0 and ci % 64 == 0 and (
                (k == 3 and stride == 1 and pad == 1) or
                (stride == 2 and l % 2 == 0 and ((k == 3 and pad == 1) or (k == 1 and pad == 0)))):
            wino = 16                                # bf16 operands / fp32 sums (conv dtype bf16)
        if both_x3:
            if co % 64 or ci % 64:
                raise ValueError('conv_wgrad_multi: x3 operands need channel counts that are multiples of 64')
            wino = 49
        _chk(L.da_conv_wgrad_plan(rows, lo, co, ci, k, wino, plan), 'da_conv_wgrad_plan')
        cap = plan[2] * (2 if dws is not None and wino == 1 else 1)      # (chained: a last-round job may write twice the slabs)
        ws = torch.empty((cap * k * co * ci,), device=x.device, dtype=torch.float32)
        d.dy, d.x, d.workspace = dy.data_ptr(), x.data_ptr(), ws.data_ptr()
        d.rows, d.Lm, d.Ldy, d.lddy, d.N, d.Lx, d.ldx, d.C = rows, lo, ldy_len, lddy, co, l, ldx, ci
        d.dy_stride, d.dy_off, d.src_stride, d.ntaps = 1, 0, stride, k
        d.winograd = wino
        if extra:
            wino = 0                                     # (the operand forms run on the direct kernels)
            _chk(L.da_conv_wgrad_plan(rows, lo, co, ci, k, 0, plan), 'da_conv_wgrad_plan')
            cap = plan[2]
            ws = torch.empty((cap * k * co * ci,), device=x.device, dtype=torch.float32)
            d.workspace, d.winograd = ws.data_ptr(), 0
            if stride != 1:
                raise ValueError('conv_wgrad_multi: the dense-block operand forms belong to stride-1 jobs')
            d.dy_half = 1 if extra.get('dy_half') else 0
            if extra.get('xform') is not None:
                mean_v, invstd_v, gamma, beta, R = extra['xform']
                if rows % R:
                    raise ValueError('conv_wgrad_multi: rows not a multiple of rows_per_window')
                d.xform, d.Wn = 1, R * l
                d.ldstat = _sv(mean_v, rows // R, ci, 'mean')
                if _sv(invstd_v, rows // R, ci, 'invstd') != d.ldstat:
                    raise ValueError('conv_wgrad_multi: mean / invstd must be slices of tables with one pitch')
                d.mean, d.invstd, d.gamma, d.beta = mean_v.data_ptr(), invstd_v.data_ptr(), _f32(gamma).data_ptr(), _f32(beta).data_ptr()
        for t in range(3):
            d.src_off[t] = t - pad if t < k else 0
        outs.append((ws, plan[2], k, co, ci))
        caps.append(cap)
    if dws is not None:
        if len(dws) != len(jobs):
            raise ValueError('conv_wgrad_multi: one destination (or None) per job')
        ptrs = (ctypes.c_void_p * len(jobs))()
        for i, (dw, (_, _, k, co, ci)) in enumerate(zip(dws, outs)):
            if dw is not None:
                if tuple(dw.shape) != (co, ci, k) or not dw.is_contiguous():
                    raise ValueError('conv_wgrad_multi: bad dw shape')
                ptrs[i] = _f32(dw, 'dw').data_ptr()
        red, spl = (ctypes.c_int * len(jobs))(), (ctypes.c_int * len(jobs))()
        _chk(L.da_conv_wgrad_multi_reduce(arr, len(jobs), ptrs, 1 if accumulate else 0, red, spl, _stream()), 'da_conv_wgrad_multi_reduce')
        for i, (ws, planned, k, co, ci) in enumerate(outs):       # the slabs actually written (a batch plan: fewer; a last-round job: twice)
            if not 1 <= spl[i] <= caps[i]:
                raise RuntimeError('conv_wgrad_multi: job %d wrote %d slabs into a workspace of %d' % (i, spl[i], caps[i]))
            outs[i] = (ws, spl[i], k, co, ci)
        return outs, [bool(r) for r in red]
    _chk(L.da_conv_wgrad_multi(arr, len(jobs), _stream()), 'da_conv_wgrad_multi')
    return outs


def wgrad_reduce_multi(items, accumulate=True):
    """items: ((slab, splits, k, co, ci), dw) -- one launch per 32 convolutions."""
    if not items:
        return
    arr = (_lib.WgradReduceDesc * len(items))()
    for d, ((slab, splits, k, co, ci), dw) in zip(arr, items):
        if tuple(dw.shape) != (co, ci, k):
            raise ValueError('wgrad_reduce_multi: bad dw shape')
        d.slab, d.dw, d.splits, d.ntaps, d.N, d.C = slab.data_ptr(), dw.data_ptr(), splits, k, co, ci
    _chk(_lib.lib().da_wgrad_reduce_multi(arr, len(items), 1 if accumulate else 0, _stream()), 'da_wgrad_reduce_multi')


def _running_descs(items):
    arr = (_lib.BnRunningDesc * len(items))()
    for d, (mean, invstd, wn, rm, rv, nbt, mom, eps) in zip(arr, items):
        if nbt is not None and nbt.dtype != torch.int64:
            raise ValueError('num_batches_tracked must be int64')
        d.mean, d.invstd, d.running_mean, d.running_var = mean.data_ptr(), invstd.data_ptr(), rm.data_ptr(), rv.data_ptr()
        d.num_batches_tracked = None if nbt is None else nbt.data_ptr()
        d.W, d.C, d.Wn, d.eps, d.momentum = mean.shape[0], mean.shape[1], wn, eps, mom
    return arr


def step_tail_multi(items, pgrad, running, accumulate=True, stem=None):
    """The tail of a training step in ONE launch: wgrad_reduce_multi(items), bn_param_grad_multi(pgrad) and
    bn_running_multi(running) (the library falls back to the three launches beyond 32 reductions / 24 BatchNorms).
    stem = (partial, nblk, n, dw): the partials a deferred stem_fused_bwd left, folded into dw by the same launch."""
    if not items:
        bn_running_multi(running)
        bn_param_grad_multi(pgrad, accumulate)
        if stem is not None:
            part, nblk, n, dw = stem
            _chk(_lib.lib().da_stem_wgrad_reduce(_p(part), nblk, n, _p(dw), 1 if accumulate else 0, _stream()), 'da_stem_wgrad_reduce')
        return
    arr = (_lib.WgradReduceDesc * len(items))()
    for d, ((slab, splits, k, co, ci), dw) in zip(arr, items):
        if tuple(dw.shape) != (co, ci, k):
            raise ValueError('step_tail_multi: bad dw shape')
        d.slab, d.dw, d.splits, d.ntaps, d.N, d.C = slab.data_ptr(), dw.data_ptr(), splits, k, co, ci
    pg = (_lib.BnPgradDesc * max(1, len(pgrad)))()
    for d, (ds, dg, db) in zip(pg, pgrad):
        _, w, c = ds.shape
        d.s1, d.s2 = ds.data_ptr(), ds.data_ptr() + 4 * w * c
        d.dgamma, d.dbeta, d.W, d.C = dg.data_ptr(), db.data_ptr(), w, c
    run = _running_descs(running) if running else None
    sp, snb, sn, sdw = (None, 0, 0, None) if stem is None else (_p(stem[0]), stem[1], stem[2], _p(stem[3]))
    _chk(_lib.lib().da_step_tail_multi(arr, len(items), pg if pgrad else None, len(pgrad), run, len(running), sp, snb, sn, sdw,
                                       1 if accumulate else 0, _stream()), 'da_step_tail_multi')


def repack_multi(weights, winograd=None):
    """[(Co,Ci,K) weights] -> [(wf, wd, uf, ud)] with one launch per 32 weights.  winograd[i] (K == 3; True / 4:
    F(2,3), 6: F(4,3)): emit the Winograd taps uf (points,Co,Ci) / ud (points,Ci,Co) INSTEAD of the direct packs
    wf / wd (None in the tuple); 16: bf16 tap packs of conv3_bf16 in the uf / ud places; 49: the chunked split-bf16
    packs of conv3_x3p there."""
    outs, descs = [], []
    for n, w in enumerate(weights):
        _f32(w, 'w')
        co, ci, k = w.shape
        code = winograd[n] if winograd is not None else 0
        wino = bool(code)
        if wino and k != 3 and not (code in (16, 49) and k == 1):
            raise ValueError('winograd taps need a 3-tap weight')
        pts = code if wino and code in (6, 16, 49) else 4
        mk = lambda *shape: torch.empty(shape, device=w.device, dtype=torch.float32)
        wf, wd = (None, None) if wino else (mk(k, co, ci), mk(k, ci, co))
        if pts == 16:                                # bf16 tap packs (3, Co, Ci) / (3, Ci, Co)
            if co % 32 or ci % 32:
                raise ValueError('bf16 tap packs need channel counts that are multiples of 32')
            uf = torch.empty((k, co, ci), device=w.device, dtype=torch.bfloat16)
            ud = torch.empty((k, ci, co), device=w.device, dtype=torch.bfloat16)
        elif pts == 49:                              # chunked split-bf16 packs of conv3_x3p (18 KB per 64 x 16 chunk)
            if co % 64 or ci % 64:
                raise ValueError('chunked split-bf16 packs need channel counts that are multiples of 64')
            uf = torch.empty((co // 64, ci // 16, 18, 64, 8), device=w.device, dtype=torch.bfloat16)
            ud = torch.empty((ci // 64, co // 16, 18, 64, 8), device=w.device, dtype=torch.bfloat16)
        else:
            uf, ud = (mk(pts, co, ci), mk(pts, ci, co)) if wino else (None, None)
        descs.append((w.data_ptr(), _p(wf), _p(wd), _p(uf), _p(ud), co, ci, k, pts))
        outs.append((wf, wd, uf, ud))
    if descs:
        arr = (_lib.RepackDesc * len(descs))()
        for d, v in zip(arr, descs):
            d.W, d.Wf, d.Wd, d.Uf, d.Ud, d.Co, d.Ci, d.K, d.points = v
        _chk(_lib.lib().da_repack_multi(arr, len(descs), _stream()), 'da_repack_multi')
    return outs


STEM_SHAPES = ((1, 7, 2), (2, 7, 2), (3, 7, 2), (1, 3, 1))      # (Cin, K, stride) the library instantiates; pad = K // 2


def stem_conv_fwd(x, w, stride=2):
    """First convolution of a breath block on the raw rows.  x (rows, Lin) [one channel] or (rows, Cin, Lin),
    w (C0, Cin, K) -> (rows, Lin / stride, C0) RLC; pad = K // 2.  (Cin, K, stride) in STEM_SHAPES."""
    _f32(x, 'x')
    _f32(w, 'w')
    if x.dim() == 2:
        x = x.view(x.shape[0], 1, x.shape[1])
    rows, cin, lin = x.shape
    c0, cin_w, k = w.shape
    if cin_w != cin or (cin, k, stride) not in STEM_SHAPES:
        raise ValueError('stem conv: x %s with w %s stride %d is not one of the (Cin, K, stride) shapes %s' %
                         (tuple(x.shape), tuple(w.shape), stride, STEM_SHAPES))
    y = torch.empty((rows, lin // stride, c0), device=x.device, dtype=ACT)
    _chk(_lib.lib().da_stem_conv_fwd_g(_p(x), _p(w), _p(y), rows, lin, cin, k, stride, c0, c0, _stream()),
         'da_stem_conv_fwd_g')
    return y


def stem_conv_wgrad(dy, x, out=None, accumulate=False, k=7, stride=2):
    _rlc(dy, 'dy')
    if x.dim() == 2:
        x = x.view(x.shape[0], 1, x.shape[1])
    rows, lo, c0 = dy.shape
    cin, lin = x.shape[1], x.shape[2]
    if (cin, k, stride) not in STEM_SHAPES or lo * stride != lin:
        raise ValueError('stem conv weight gradient: unsupported shape')
    if out is None:
        out = torch.empty((c0, cin, k), device=x.device, dtype=torch.float32)
    L = _lib.lib()
    ws = torch.empty((L.da_stem_wgrad_workspace_g(rows, c0, cin, k) // 4,), device=x.device, dtype=torch.float32)
    _chk(L.da_stem_conv_wgrad_g(_p(dy), c0, _p(x), _p(out), _p(ws), rows, lin, cin, k, stride, c0,
                                1 if accumulate else 0, _stream()), 'da_stem_conv_wgrad_g')
    return out


# ------------------------------------------------------------------------------------------------
# window-grouped batch norm
# ------------------------------------------------------------------------------------------------
def _bn_ws(w, wn, c, dev):
    return torch.empty((_lib.lib().da_bn_workspace(w, wn, c) // 4,), device=dev, dtype=torch.float32)


def bn_stats_partial(x, R):
    """Stage 1 of the per-window statistics: chunk records part[w][p][{mean,M2}][C] (opaque tensor)."""
    _rlc(x, 'x')
    rows, l, c = x.shape
    if rows % R:
        raise ValueError('rows %d not a multiple of rows_per_window %d' % (rows, R))
    w = rows // R
    part = _bn_ws(w, R * l, c, x.device)
    _chk(_lib.lib().da_bn_stats_partial(_p(x), c, w, R * l, c, _p(part), _stream()), 'da_bn_stats_partial')
    return part


def bn_running_multi(items):
    """items: (mean, invstd, Wn, running_mean, running_var, num_batches_tracked|None, momentum, eps).
    One launch per 32 BatchNorms: the reference's one momentum update per window, in window order."""
    if not items:
        return
    _chk(_lib.lib().da_bn_running_multi(_running_descs(items), len(items), _stream()), 'da_bn_running_multi')


def bn_stats(x, R, eps=1e-5, running_mean=None, running_var=None, num_batches_tracked=None, momentum=0.1):
    """-> mean, invstd of shape (W, C); window = R rows (standalone form: partial + merge).  With running
    buffers the reference's per-window momentum updates are applied to them in place."""
    part = bn_stats_partial(x, R)
    rows, l, c = x.shape
    w = rows // R
    mean = torch.empty((w, c), device=x.device, dtype=torch.float32)
    invstd = torch.empty((w, c), device=x.device, dtype=torch.float32)
    _chk(_lib.lib().da_bn_stats_merge(_p(part), w, R * l, c, eps, _p(mean), _p(invstd), _stream()), 'da_bn_stats_merge')
    if running_mean is not None:
        bn_running_multi([(mean, invstd, R * l, running_mean, running_var, num_batches_tracked, momentum, eps)])
    return mean, invstd


def bn_apply(x, R, mean, invstd, gamma, beta, relu=True, res=None, out=None, part=None, eps=1e-5):
    """out = act(bn(x) (+res)).  With `part` (bn_stats_partial) mean/invstd are OUTPUTS filled on the way."""
    _rlc(x, 'x')
    rows, l, c = x.shape
    w = rows // R
    if out is None:
        out = torch.empty_like(x)
    if res is not None and tuple(res.shape) != tuple(x.shape):
        raise ValueError('residual shape mismatch')
    if tuple(mean.shape) != (w, c) or tuple(invstd.shape) != (w, c):
        raise ValueError('mean/invstd must be (W, C)')
    _chk(_lib.lib().da_bn_apply(_p(x), c, _p(res), c, _p(out), c, w, R * l, c, _p(mean), _p(invstd), _p(gamma),
                                _p(beta), 1 if relu else 0, _p(part), eps, _stream()), 'da_bn_apply')
    return out


def bn_fwd(x, R, gamma, beta, relu=True, res=None, eps=1e-5, out=None, want_mask=False):
    """-> out, mean, invstd: per-window statistics and out = act(bn(x) (+res)) in one call (single pass over x when
    a window slab fits a block's registers).  want_mask (relu only): -> out, mean, invstd, mask with the ReLU decisions
    as a bit mask for bn_bwd(mask=...) (None when the shape takes the two-stage kernels)."""
    _rlc(x, 'x')
    rows, l, c = x.shape
    if rows % R:
        raise ValueError('rows %d not a multiple of rows_per_window %d' % (rows, R))
    w = rows // R
    if out is None:
        out = torch.empty_like(x)
    if res is not None and tuple(res.shape) != tuple(x.shape):
        raise ValueError('residual shape mismatch')
    mean = torch.empty((w, c), device=x.device, dtype=torch.float32)
    invstd = torch.empty((w, c), device=x.device, dtype=torch.float32)
    scratch = _bn_ws(w, R * l, c, x.device)
    mask = None
    if want_mask and relu:
        words = _lib.lib().da_bn_mask_words(w, R * l, c)
        if words:
            mask = torch.empty((words,), device=x.device, dtype=torch.int64)
    if mask is not None:
        _chk(_lib.lib().da_bn_fwd_mask(_p(x), c, _p(res), c, _p(out), c, w, R * l, c, _p(mean), _p(invstd), _p(gamma),
                                       _p(beta), eps, _p(scratch), _p(mask), _stream()), 'da_bn_fwd_mask')
    else:
        _chk(_lib.lib().da_bn_fwd(_p(x), c, _p(res), c, _p(out), c, w, R * l, c, _p(mean), _p(invstd), _p(gamma), _p(beta),
                                  1 if relu else 0, eps, _p(scratch), _stream()), 'da_bn_fwd')
    return (out, mean, invstd, mask) if want_mask else (out, mean, invstd)


def bn_fwd_pair(items, R, eps=1e-5):
    """Two BatchNorm forwards of ONE shape in one launch (a stride-2 block entry: bn1 + ReLU on conv1's output and the
    downsample's BatchNorm): items = two (x, gamma, beta, relu, res | None, want_mask) -> [(out, mean, invstd, mask | None)].
    Single-pass geometry only (bn_single_pass)."""
    (x0, x1) = items[0][0], items[1][0]
    _rlc(x0, 'x')
    _rlc(x1, 'x')
    rows, l, c = x0.shape
    if len(items) != 2 or tuple(x1.shape) != (rows, l, c) or rows % R:
        raise ValueError('bn_fwd_pair: two tensors of one (rows, L, C) shape, rows a multiple of rows_per_window')
    w = rows // R
    L = _lib.lib()
    words = L.da_bn_mask_words(w, R * l, c)
    arr = (_lib.BnFwdDesc * 2)()
    outs = []
    for d, (x, gamma, beta, relu, res, want_mask) in zip(arr, items):
        if res is not None and tuple(res.shape) != (rows, l, c):
            raise ValueError('residual shape mismatch')
        out = torch.empty_like(x)
        mean = torch.empty((w, c), device=x.device, dtype=torch.float32)
        invstd = torch.empty((w, c), device=x.device, dtype=torch.float32)
        mask = torch.empty((words,), device=x.device, dtype=torch.int64) if (want_mask and relu and words) else None
        d.x, d.ldx, d.res, d.ldr, d.out, d.ldo = x.data_ptr(), c, (res.data_ptr() if res is not None else None), c, out.data_ptr(), c
        d.mean, d.invstd, d.gamma, d.beta = mean.data_ptr(), invstd.data_ptr(), _f32(gamma).data_ptr(), _f32(beta).data_ptr()
        d.relu, d.mask = (1 if relu else 0), (mask.data_ptr() if mask is not None else None)
        outs.append((out, mean, invstd, mask))
    _chk(L.da_bn_fwd_pair(arr, w, R * l, c, eps, _stream()), 'da_bn_fwd_pair')
    return outs


def bn_two_ok(x, R):
    """Whether bn_bwd_two / bn_bwd_pair(dout2=...) take this (rows, L, C) map (single-pass geometry of at most 512 threads)."""
    rows, l, c = x.shape
    return rows % R == 0 and rows > 0 and bool(_lib.lib().da_bn_two_ok(rows // R, R * l, c))


def bn_bwd_two(dout, dout2, x, R, mean, invstd, gamma, beta, mask, want_g=False, dx=None):
    """bn_bwd(mask=...) whose upstream gradient is dout + dout2 (a residual block left its input gradient as its two terms:
    no accumulating conv epilogue).  -> dx, g (the masked SUM; only with want_g), ds (2, W, C)."""
    _rlc(dout, 'dout')
    _rlc(dout2, 'dout2')
    _rlc(x, 'x')
    rows, l, c = x.shape
    if tuple(dout.shape) != (rows, l, c) or tuple(dout2.shape) != (rows, l, c) or not bn_two_ok(x, R):
        raise ValueError('bn_bwd_two: shapes dout%s dout2%s x%s' % (tuple(dout.shape), tuple(dout2.shape), tuple(x.shape)))
    w = rows // R
    if dx is None:
        dx = torch.empty_like(x)
    g = torch.empty_like(x) if want_g else None
    ds = torch.empty((2, w, c), device=x.device, dtype=torch.float32)
    _chk(_lib.lib().da_bn_bwd_mask2(_p(dout), c, _p(dout2), c, _p(x), c, _p(dx), c, _p(g), c, w, R * l, c, _p(mean), _p(invstd),
                                    _p(_f32(gamma)), _p(_f32(beta)), _p(ds), _p(mask), _stream()), 'da_bn_bwd_mask2')
    return dx, g, ds


def bn_bwd_pair(dout, items, R, mask, dout2=None):
    """The two BatchNorm backwards that share one masked gradient dout * [out > 0] (a block entry's bn2 and its downsample's
    BatchNorm; ``mask``: the ReLU bit mask of the block output's bn_fwd(want_mask=True)) in one launch: items = two
    (x, mean, invstd, gamma, beta, dx | None) -> [(dx, ds (2, W, C))].  dout2: the upstream gradient is dout + dout2."""
    _rlc(dout, 'dout')
    rows, l, c = dout.shape
    if len(items) != 2 or rows % R or mask is None:
        raise ValueError('bn_bwd_pair: two items, rows a multiple of rows_per_window, a ReLU bit mask')
    w = rows // R
    arr = (_lib.BnBwdDesc * 2)()
    outs = []
    for d, (x, mean, invstd, gamma, beta, dx) in zip(arr, items):
        _rlc(x, 'x')
        if tuple(x.shape) != (rows, l, c):
            raise ValueError('bn_bwd_pair: shape mismatch')
        if dx is None:
            dx = torch.empty_like(x)
        ds = torch.empty((2, w, c), device=x.device, dtype=torch.float32)
        d.x, d.ldx, d.dx, d.lddx = x.data_ptr(), c, dx.data_ptr(), c
        d.mean, d.invstd, d.gamma, d.beta, d.ds = mean.data_ptr(), invstd.data_ptr(), _f32(gamma).data_ptr(), _f32(beta).data_ptr(), ds.data_ptr()
        outs.append((dx, ds))
    if dout2 is not None:
        _rlc(dout2, 'dout2')
        if tuple(dout2.shape) != (rows, l, c):
            raise ValueError('bn_bwd_pair: dout2 shape mismatch')
        _chk(_lib.lib().da_bn_bwd_pair2(_p(dout), c, _p(dout2), c, arr, w, R * l, c, _p(mask), _stream()), 'da_bn_bwd_pair2')
    else:
        _chk(_lib.lib().da_bn_bwd_pair(_p(dout), c, arr, w, R * l, c, _p(mask), _stream()), 'da_bn_bwd_pair')
    return outs


def bn_x3_ok(rows, l, c, R):
    """Whether the BatchNorm of a (rows, L, C) tensor in windows of R rows has the single-pass geometry the x3 store forms
    exist for (a window slab fits one block's registers: R * L <= 1280 at the usual channel counts)."""
    return c % 16 == 0 and rows % R == 0 and _lib.lib().da_bn_mask_words(rows // R, R * l, c) > 0


def bn_fwd_x(x, R, gamma, beta, relu=True, res=None, eps=1e-5, want_mask=False, out_x3=True):
    """bn_fwd on float activations whose output (out_x3) and / or residual (an x3 tensor) are in the x3 format: the
    producers of the k3 s1 convs' operands under conv arithmetic 'f32x3'.  -> out, mean, invstd[, mask]."""
    _rlc32(x, 'x')
    rows, l, c = x.shape
    if rows % R:
        raise ValueError('rows %d not a multiple of rows_per_window %d' % (rows, R))
    w = rows // R
    res_x3 = is_x3(res)
    if res is not None and (tuple(res.shape) != ((rows, l, c // 16, 3, 16) if res_x3 else (rows, l, c)) or not res.is_contiguous()):
        raise ValueError('residual shape mismatch')
    out = x3_empty(rows, l, c, x.device) if out_x3 else torch.empty_like(x)
    mean = torch.empty((w, c), device=x.device, dtype=torch.float32)
    invstd = torch.empty((w, c), device=x.device, dtype=torch.float32)
    mask = None
    if want_mask and relu:
        mask = torch.empty((_lib.lib().da_bn_mask_words(w, R * l, c),), device=x.device, dtype=torch.int64)
    _chk(_lib.lib().da_bn_fwd_x(_p(x), c, _p(res), c, _p(out), c, w, R * l, c, _p(mean), _p(invstd), _p(gamma), _p(beta),
                                1 if relu else 0, eps, _p(mask), 1 if res_x3 else 0, 1 if out_x3 else 0, _stream()),
         'da_bn_fwd_x')
    return (out, mean, invstd, mask) if want_mask else (out, mean, invstd)


def bn_bwd_x(dout, x, R, mean, invstd, gamma, beta, mask_mode, want_g=False, mask=None, dx_x3=True):
    """bn_bwd on float activations with dx stored in the x3 format (dx_x3); parameter gradients are always deferred
    (fold ds with bn_param_grad_multi).  -> dx, g (float, only when want_g), ds."""
    _rlc32(dout, 'dout')
    _rlc32(x, 'x')
    rows, l, c = x.shape
    w = rows // R
    dx = x3_empty(rows, l, c, x.device) if dx_x3 else torch.empty_like(x)
    g = torch.empty_like(x) if want_g else None
    ds = torch.empty((2, w, c), device=x.device, dtype=torch.float32)
    _chk(_lib.lib().da_bn_bwd_x(_p(dout), c, _p(x), c, _p(dx), c, _p(g), c, w, R * l, c, _p(mean), _p(invstd), _p(gamma),
                                _p(beta), mask_mode, _p(ds), _p(mask), 1 if dx_x3 else 0, _stream()), 'da_bn_bwd_x')
    return dx, g, ds


def bn_debug_two_stage(on):
    _chk(_lib.lib().da_bn_debug_two_stage(1 if on else 0), 'da_bn_debug_two_stage')


def bn_bwd(dout, x, R, mean, invstd, gamma, beta, mask_mode, out=None, want_g=False, dx=None,
           dgamma=None, dbeta=None, accumulate=False, defer_param_grads=False, add=None, mask=None):
    """-> dx, dgamma, dbeta, g, ds.  g = masked upstream gradient (only when want_g); ds (2,W,C) holds the
    per-window totals; with defer_param_grads dgamma/dbeta are not computed (fold ds with bn_param_grad_multi).
    add = (tensor (rows, L, Ca >= C), channel offset): dx += tensor[:, :, off:off+C] in the same pass."""
    _rlc(dout, 'dout')
    _rlc(x, 'x')
    rows, l, c = x.shape
    w = rows // R
    if dx is None:
        dx = torch.empty_like(x)
    g = torch.empty_like(x) if want_g else None
    if not defer_param_grads and dgamma is None:
        dgamma = torch.empty((c,), device=x.device, dtype=torch.float32)
        dbeta = torch.empty((c,), device=x.device, dtype=torch.float32)
    scratch = _bn_ws(w, R * l, c, x.device)
    ds = torch.empty((2, w, c), device=x.device, dtype=torch.float32)
    args = (_p(dout), c, _p(x), c, _p(out), c, _p(dx), c, _p(g), c, w, R * l, c, _p(mean),
            _p(invstd), _p(gamma), _p(beta), mask_mode, _p(scratch), _p(ds),
            None if defer_param_grads else _p(dgamma), None if defer_param_grads else _p(dbeta),
            1 if accumulate else 0)
    if mask is not None:          # ReLU decisions from bn_fwd(want_mask=True) instead of reading `out`
        if add is not None:
            raise ValueError('bn_bwd: mask and add are exclusive')
        _chk(_lib.lib().da_bn_bwd_mask(_p(dout), c, _p(x), c, _p(dx), c, _p(g), c, w, R * l, c, _p(mean), _p(invstd),
                                       _p(gamma), _p(beta), _p(scratch), _p(ds),
                                       None if defer_param_grads else _p(dgamma), None if defer_param_grads else _p(dbeta),
                                       1 if accumulate else 0, _p(mask), _stream()), 'da_bn_bwd_mask')
    elif add is None:
        _chk(_lib.lib().da_bn_bwd(*args, _stream()), 'da_bn_bwd')
    else:
        at, off = add
        _rlc(at, 'add')
        if tuple(at.shape[:2]) != (rows, l) or off % 4 or off + c > at.shape[2]:
            raise ValueError('bn_bwd: bad add operand')
        ap = ctypes.c_void_p(at.data_ptr() + 4 * off)
        _chk(_lib.lib().da_bn_bwd_add(*args, ap, at.shape[2], _stream()), 'da_bn_bwd_add')
    return dx, dgamma, dbeta, g, ds


def bn_pool_ok(x, R):
    """Whether bn_fwd_pool / bn_bwd_pool take this (rows, L, C) map with windows of R rows."""
    rows, l, c = x.shape
    return rows % R == 0 and rows > 0 and bool(_lib.lib().da_bn_pool_ok(rows // R, R * l, c, l))


def bn_fwd_pool(x, R, gamma, beta, res=None, eps=1e-5):
    """relu(bn(x) (+res)) WITHOUT storing it: -> flat (rows, C) float, the average over the L positions of every row (what
    head_fwd / global_avgpool_fwd pool from the stored map, bit for bit), mean, invstd (W, C), mask (the ReLU decisions for
    bn_bwd_pool).  The block-output BatchNorm of the last residual block in front of the head."""
    _rlc(x, 'x')
    rows, l, c = x.shape
    if not bn_pool_ok(x, R):
        raise ValueError('bn_fwd_pool: shape %s / R %d has no pooled form' % (tuple(x.shape), R))
    if res is not None and tuple(res.shape) != tuple(x.shape):
        raise ValueError('residual shape mismatch')
    w = rows // R
    mk = lambda *sh: torch.empty(sh, device=x.device, dtype=torch.float32)
    flat, mean, invstd = mk(rows, c), mk(w, c), mk(w, c)
    mask = torch.empty((_lib.lib().da_bn_mask_words(w, R * l, c),), device=x.device, dtype=torch.int64)
    _chk(_lib.lib().da_bn_fwd_pool(_p(x), c, _p(res), c, _p(flat), w, R * l, c, l, _p(mean), _p(invstd), _p(_f32(gamma)),
                                   _p(_f32(beta)), eps, _p(mask), _stream()), 'da_bn_fwd_pool')
    return flat, mean, invstd, mask


def bn_bwd_pool(dflat, x, R, mean, invstd, gamma, beta, mask, want_g=False, dx=None):
    """The backward of bn_fwd_pool: dflat (rows, C) float = the gradient of the pooled features -> dx, g (the masked upstream
    gradient, activation storage type; only with want_g), ds (2, W, C)."""
    _rlc(x, 'x')
    _f32(dflat, 'dflat')
    rows, l, c = x.shape
    if tuple(dflat.shape) != (rows, c) or not bn_pool_ok(x, R):
        raise ValueError('bn_bwd_pool: shapes dflat%s x%s' % (tuple(dflat.shape), tuple(x.shape)))
    w = rows // R
    if dx is None:
        dx = torch.empty_like(x)
    g = torch.empty_like(x) if want_g else None
    ds = torch.empty((2, w, c), device=x.device, dtype=torch.float32)
    _chk(_lib.lib().da_bn_bwd_pool(_p(dflat), c, _p(x), c, _p(dx), c, _p(g), c, w, R * l, c, l, _p(mean), _p(invstd),
                                   _p(_f32(gamma)), _p(_f32(beta)), _p(ds), _p(mask), _stream()), 'da_bn_bwd_pool')
    return dx, g, ds


def bn_param_grad_multi(items, accumulate=True):
    """items: (ds (2,W,C), dgamma (C,), dbeta (C,)) -- one launch per 32 BatchNorms."""
    if not items:
        return
    arr = (_lib.BnPgradDesc * len(items))()
    for d, (ds, dg, db) in zip(arr, items):
        _, w, c = ds.shape
        d.s1, d.s2 = ds.data_ptr(), ds.data_ptr() + 4 * w * c
        d.dgamma, d.dbeta, d.W, d.C = dg.data_ptr(), db.data_ptr(), w, c
    _chk(_lib.lib().da_bn_param_grad_multi(arr, len(items), 1 if accumulate else 0, _stream()),
         'da_bn_param_grad_multi')


# ------------------------------------------------------------------------------------------------
# the dense block as one design (include/deepards_hip.h "the dense block as one design"): a pitched buffer per block, one
# pitched statistics table per block, relu(norm1(x)) never stored
# ------------------------------------------------------------------------------------------------
def bn_single_pass(w, wn, c):
    """Whether a BatchNorm over W windows of wn positions x C channels has the single-pass geometry (a window slab in one
    block's registers): what bn_bwd_ss and the block-fused forms need."""
    return c % 32 == 0 and _lib.lib().da_bn_mask_words(w, wn, c) > 0


def dense_fused_ok(rows, R, l, channels):
    """Whether BatchNorms over (rows, l, C) for every C in ``channels`` have the single-pass geometry the dense-block
    kernels need (float storage, a window slab in one block's registers, a conv tile within two windows)."""
    if ACT != torch.float32 or rows % R or R * l < 64:
        return False
    L = _lib.lib()
    return all(c % 32 == 0 and L.da_bn_mask_words(rows // R, R * l, c) > 0 for c in channels)


def bn_stats_fused(xv, R, mean_v, invstd_v, eps=1e-5):
    """Per-window statistics of the channel slice xv (rows, L, C) of a pitched buffer, written into the (W, C) slices
    mean_v / invstd_v of the block's statistics tables."""
    ldx = _pv(xv, 'x')
    rows, l, c = xv.shape
    w = rows // R
    ldstat = _sv(mean_v, w, c, 'mean')
    if rows % R or _sv(invstd_v, w, c, 'invstd') != ldstat:
        raise ValueError('bn_stats_fused: bad statistics slices')
    _chk(_lib.lib().da_bn_stats_fused(_p(xv), ldx, w, R * l, c, _p(mean_v), _p(invstd_v), ldstat, eps, _stream()),
         'da_bn_stats_fused')


def bn_relu_ss(xv, R, mean_v, invstd_v, gamma, beta):
    """relu(norm(xv)) in the fused-multiply-add form the dense-block kernels apply on the fly -> contiguous (rows, L, C)
    (tests / explainers: the hot path never stores it)."""
    ldx = _pv(xv, 'x')
    rows, l, c = xv.shape
    w = rows // R
    ldstat = _sv(mean_v, w, c, 'mean')
    if rows % R or _sv(invstd_v, w, c, 'invstd') != ldstat:
        raise ValueError('bn_relu_ss: bad statistics slices')
    out = torch.empty((rows, l, c), device=xv.device, dtype=torch.float32)
    _chk(_lib.lib().da_bn_relu_ss(_p(xv), ldx, _p(out), c, w, R * l, c, _p(mean_v), _p(invstd_v), ldstat, _p(_f32(gamma)),
                                  _p(_f32(beta)), _stream()), 'da_bn_relu_ss')
    return out


def bn_bwd_ss(dout, xv, R, mean_v, invstd_v, gamma, beta, relu, dx, add=None, half_dout=False, drop=None, out=None, hout=None):
    """Backward of relu(norm(xv)) (relu: decision from the fused-multiply-add form) or norm(xv): dout (rows, L, C) -- or
    (rows, L / 2, C) with half_dout (a transition's pooling in front of its conv) --; relu = 1: decision of the fused
    multiply-add form, 2: the sign of ``out`` (the stored output of a bn_fwd forward), 0: none; dx a (rows, L, C) channel slice that
    receives the input gradient (+ ``add``, which may be dx itself: in-place accumulation into the block's gradient
    buffer); drop = (seed, salt, p, g): the dropout mask on the last g channels of dx; hout (relu = 1): receives
    relu(norm(xv)) itself (the activation the forward never stored).  -> ds (2, W, C) window sums.  Either storage type
    (bf16: contiguous tensors)."""
    _ld = _pv if ACT == torch.float32 else (lambda t, name: _rlc(t, name).shape[2])
    ldd, ldx, lddx = _ld(dout, 'dout'), _ld(xv, 'x'), _ld(dx, 'dx')
    rows, l, c = xv.shape
    w = rows // R
    if rows % R or tuple(dx.shape) != (rows, l, c) or tuple(dout.shape) != (rows, l // 2 if half_dout else l, c) or \
            (half_dout and l % 2):
        raise ValueError('bn_bwd_ss: shape mismatch x%s dout%s dx%s' % (tuple(xv.shape), tuple(dout.shape), tuple(dx.shape)))
    ldstat = _sv(mean_v, w, c, 'mean')
    if _sv(invstd_v, w, c, 'invstd') != ldstat:
        raise ValueError('bn_bwd_ss: bad statistics slices')
    ldadd = 0
    if add is not None:
        if tuple(add.shape) != (rows, l, c):
            raise ValueError('bn_bwd_ss: bad add operand')
        ldadd = _ld(add, 'add')
    seed, salt, p, g = drop if drop is not None else (None, 0, 0.0, 0)
    ldo = ldh = 0
    if relu == 2:
        if out is None or tuple(out.shape) != (rows, l, c):
            raise ValueError('bn_bwd_ss: relu = 2 takes its decisions from the stored output')
        ldo = _ld(out, 'out')
    if hout is not None:
        if relu != 1 or tuple(hout.shape) != (rows, l, c):
            raise ValueError('bn_bwd_ss: hout is the (rows, L, C) activation of the relu = 1 form')
        ldh = _ld(hout, 'hout')
    ds = torch.empty((2, w, c), device=xv.device, dtype=torch.float32)
    _chk(_lib.lib().da_bn_bwd_ss(_p(dout), ldd, _p(xv), ldx, _p(out) if relu == 2 else None, ldo, _p(dx), lddx, _p(add), ldadd, w,
                                 R * l, c, _p(mean_v), _p(invstd_v), ldstat, _p(_f32(gamma)), _p(_f32(beta)), int(relu), 1 if half_dout else 0,
                                 _p(seed) if p > 0 else None, salt, p, g, _p(ds), _p(hout), ldh, _stream()), 'da_bn_bwd_ss')
    return ds


def conv1x1_bn(xv, w, R, mean_v, invstd_v, gamma, beta, out, pool=False, pend=None, eps=1e-5, want_records=False):
    """out = conv1x1(relu(norm(xv))) with the activation applied while the operand is staged; pool: the transition form
    (AvgPool1d(2,2) folded in front of the conv: out has L / 2 positions).  w: the (N, C, 1) torch weight; out: a
    (rows, Lout, N) channel slice (of the next block's buffer, or a plain tensor).  pend = (records, c_first, units,
    units_per_window): the channels [c_first, C) have no table entry yet -- their statistics are merged from the records
    their producer's epilogue wrote, and PUBLISHED to mean_v / invstd_v by this call; want_records: -> (out, records of
    the output) for windows of R rows (units: output positions)."""
    ldx = _pv(xv, 'x')
    rows, l, c = xv.shape
    n = w.shape[0]
    lo = l // 2 if pool else l
    if tuple(w.shape) != (n, c, 1) or not w.is_contiguous() or tuple(out.shape) != (rows, lo, n) or rows % R or (pool and l % 2):
        raise ValueError('conv1x1_bn: unsupported shapes x%s w%s out%s' % (tuple(xv.shape), tuple(w.shape), tuple(out.shape)))
    ldy = _pv(out, 'out')
    ldstat = _sv(mean_v, rows // R, c, 'mean')
    if _sv(invstd_v, rows // R, c, 'invstd') != ldstat:
        raise ValueError('conv1x1_bn: bad statistics slices')
    rec, c_first, units, wu = pend if pend is not None else (None, 0, 0, 0)
    if rec is not None and rec.numel() != _lib.lib().da_stat_records_floats(units, c - c_first):
        raise ValueError('conv1x1_bn: the pending records do not have %d units of %d channels' % (units, c - c_first))
    part = stat_records(rows * lo, n, xv.device) if want_records else None
    _chk(_lib.lib().da_conv1x1_bn(_p(xv), ldx, _p(_f32(w)), _p(out), ldy, rows, R, l, c, n, 1 if pool else 0, _p(mean_v),
                                  _p(invstd_v), ldstat, _p(_f32(gamma)), _p(_f32(beta)), _p(rec), c_first, units, wu, eps,
                                  _p(part), _stream()), 'da_conv1x1_bn')
    return (out, part) if want_records else out


# ------------------------------------------------------------------------------------------------
# pools
# ------------------------------------------------------------------------------------------------
def bn_relu_pool_fwd(y, R, mean, invstd, gamma, beta, pool_mode, out_x3=False, out=None):
    _rlc(y, 'y')
    rows, lin, c = y.shape
    lout = (lin - 1) // 2 + 1
    if out is not None:                                 # a channel slice of a dense block's pitched buffer (float storage)
        if out_x3 or tuple(out.shape) != (rows, lout, c) or ACT != torch.float32:
            raise ValueError('bn_relu_pool_fwd: bad out')
        _chk(_lib.lib().da_bn_relu_pool_fwd(_p(y), c, _p(out), _pv(out, 'out'), rows, R, lin, c, _p(mean), _p(invstd), _p(gamma),
                                            _p(beta), pool_mode, _stream()), 'da_bn_relu_pool_fwd')
        return out
    if out_x3:                      # the pooled map in the x3 format (layer1's k3 s1 convs under conv arithmetic 'f32x3')
        _rlc32(y, 'y')
        out = x3_empty(rows, lout, c, y.device)
        _chk(_lib.lib().da_bn_relu_pool_fwd_x(_p(y), c, _p(out), rows, R, lin, c, _p(mean), _p(invstd), _p(gamma), _p(beta),
                                              pool_mode, _stream()), 'da_bn_relu_pool_fwd_x')
        return out
    out = torch.empty((rows, lout, c), device=y.device, dtype=ACT)
    _chk(_lib.lib().da_bn_relu_pool_fwd(_p(y), c, _p(out), c, rows, R, lin, c, _p(mean), _p(invstd), _p(gamma),
                                        _p(beta), pool_mode, _stream()), 'da_bn_relu_pool_fwd')
    return out


def stem_fused_ok(x2d, w, R):
    """Whether the recomputing stem kernels take this stem: the default one (one input channel, k7 s2 p3, float storage,
    even length, 32 / 64 / 128 channels, a window of R rows that the backward's row pairs / quads divide).  Either activation
    storage type: with bf16 storage only the pooled map and its gradient are bf16 -- the recomputed conv output never is."""
    lin = x2d.shape[-1]
    ok = (x2d.dim() == 2 or x2d.shape[1] == 1) and tuple(w.shape[1:]) == (1, 7) and \
        lin % 2 == 0 and w.shape[0] in (32, 64, 128) and x2d.shape[0] % R == 0 and R % max(1, 128 // w.shape[0]) == 0
    if not ok or x2d.shape[0] == 0:
        return ok
    # the statistics kernel stages the raw rows of one chunk in LDS (da_stem_stats_partial: (chunk / Lc + 1) rows of
    # Lin + 12 floats + its fold area, 64 KB at most): with few chunks per window (many windows, or 128 channels) and long
    # rows -- nb 40 x L 512 -- a chunk does not fit, and the stored-map stem takes the shape as before
    p_, chunk = ctypes.c_int(), ctypes.c_int()
    _lib.lib().da_bn_chunks(x2d.shape[0] // R, R * (lin // 2), w.shape[0], ctypes.byref(p_), ctypes.byref(chunk))
    lc = lin // 2
    max_rows = (chunk.value + lc - 1) // lc + 1
    return (max_rows * (lin + 12) + 33 * 32) * 4 <= 64 * 1024


def stem_fused_fwd(x2d, w, R, gamma, beta, pool_mode, eps=1e-5, out_x3=False, out=None):
    """conv k7 s2 p3 -> BatchNorm (per window of R rows) -> ReLU -> pool(3,2,1) of the raw rows (rows, Lin) WITHOUT storing
    the conv output (recomputed in the statistics and in the apply pass: bit for bit stem_conv_fwd + bn_stats +
    bn_relu_pool_fwd).  -> out (rows, Lp, C) float or x3, mean, invstd (W, C)."""
    _f32(x2d, 'x')
    _f32(w, 'w')
    x = x2d.reshape(x2d.shape[0], x2d.shape[-1])
    rows, lin = x.shape
    c = w.shape[0]
    wn = rows // R
    lc = lin // 2
    lp = (lc - 1) // 2 + 1
    L = _lib.lib()
    part = _bn_ws(wn, R * lc, c, x.device)
    _chk(L.da_stem_stats_partial(_p(x), _p(w), rows, R, lin, c, _p(part), _stream()), 'da_stem_stats_partial')
    mean = torch.empty((wn, c), device=x.device, dtype=torch.float32)
    invstd = torch.empty((wn, c), device=x.device, dtype=torch.float32)
    _chk(L.da_bn_stats_merge(_p(part), wn, R * lc, c, eps, _p(mean), _p(invstd), _stream()), 'da_bn_stats_merge')
    ldo = c
    if out is not None:                                # a channel slice of a dense block's pitched buffer
        if out_x3 or tuple(out.shape) != (rows, lp, c):
            raise ValueError('stem_fused_fwd: bad out')
        ldo = _pv(out, 'out')
    else:
        if out_x3 and ACT != torch.float32:
            raise ValueError('stem_fused_fwd: the x3 format needs float storage')
        out = x3_empty(rows, lp, c, x.device) if out_x3 else torch.empty((rows, lp, c), device=x.device, dtype=ACT)
    _chk(L.da_stem_bn_relu_pool_fwd(_p(x), _p(w), _p(out), ldo, rows, R, lin, c, _p(mean), _p(invstd), _p(gamma), _p(beta),
                                    pool_mode, 1 if out_x3 else 0, _stream()), 'da_stem_bn_relu_pool_fwd')
    return out, mean, invstd


def stem_fused_bwd(dout, x2d, w, R, mean, invstd, gamma, beta, pool_mode, dw=None, accumulate=False, defer=False):
    """Backward of stem_fused_fwd from dout (rows, Lp, C) and the raw rows: -> dw (C, 1, 7) (+= into ``dw`` when
    accumulate), ds (2, W, C) = the BatchNorm's window sums (bn_param_grad_multi folds them into dgamma / dbeta).
    defer: the last fold of the weight gradient is left to the caller -> ((partial, nblk, n), ds) for step_tail_multi(stem=)."""
    ldd = _pv(dout, 'dout') if ACT == torch.float32 else _rlc(dout, 'dout').shape[2]
    x = x2d.reshape(x2d.shape[0], x2d.shape[-1])
    rows, lin = x.shape
    c = w.shape[0]
    if dout.shape[2] != c:
        raise ValueError('stem_fused_bwd: dout must have the stem\'s %d channels' % c)
    if dw is None and not defer:
        if accumulate:
            raise ValueError('accumulate needs dw')
        dw = torch.empty((c, 1, 7), device=x.device, dtype=torch.float32)
    L = _lib.lib()
    ds = torch.empty((2, rows // R, c), device=x.device, dtype=torch.float32)
    ws = torch.empty((L.da_stem_bwd_workspace(rows, c) // 4,), device=x.device, dtype=torch.float32)
    _chk(L.da_stem_bwd(_p(dout), ldd, _p(x), _p(w), rows, R, lin, c, _p(mean), _p(invstd), _p(gamma), _p(beta), pool_mode, _p(ds),
                       None if defer else _p(dw), 1 if accumulate else 0, _p(ws), _stream()), 'da_stem_bwd')
    if defer:
        off, nblk = ctypes.c_size_t(), ctypes.c_int()
        _chk(L.da_stem_bwd_partials(rows, R, c, ctypes.byref(off), ctypes.byref(nblk)), 'da_stem_bwd_partials')
        return (ws[off.value:], nblk.value, c * 7), ds
    return dw, ds


def pool_bwd(dout, y, R, mean, invstd, gamma, beta, pool_mode):
    ldd = _pv(dout, 'dout') if ACT == torch.float32 else _rlc(dout, 'dout').shape[2]
    _rlc(y, 'y')
    rows, lin, c = y.shape
    dz = torch.empty_like(y)
    _chk(_lib.lib().da_pool_bwd(_p(dout), ldd, _p(y), c, _p(dz), c, rows, R, lin, c, _p(mean), _p(invstd), _p(gamma),
                                _p(beta), pool_mode, _stream()), 'da_pool_bwd')
    return dz


def _rlc32(t, name='tensor'):
    if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 3 and t.is_contiguous()):
        raise ValueError('%s must be a contiguous float32 CUDA (rows, L, C) tensor, got %s %s %s' %
                         (name, tuple(t.shape), t.dtype, t.device))
    return t


def global_avgpool_fwd(x):
    """AvgPool1d(L, stride 1) on an L-long map + view: x (rows, L, C) in the activation storage type -> (rows, C)
    float32 features (the activation -> feature boundary)."""
    _rlc(x, 'x')
    rows, lin, c = x.shape
    feat = torch.empty((rows, c), device=x.device, dtype=torch.float32)
    _chk(_lib.lib().da_global_avgpool_fwd(_p(x), c, _p(feat), rows, lin, c, _stream()), 'da_global_avgpool_fwd')
    return feat


def global_avgpool_bwd(dfeat, lin):
    """dfeat (rows, C) float32 -> dx (rows, L, C) in the activation storage type."""
    _f32(dfeat, 'dfeat')
    rows, c = dfeat.shape
    dx = torch.empty((rows, lin, c), device=dfeat.device, dtype=ACT)
    _chk(_lib.lib().da_global_avgpool_bwd(_p(dfeat), _p(dx), c, rows, lin, c, _stream()), 'da_global_avgpool_bwd')
    return dx


def avgpool_fwd(x, k):
    """float32 tensors only (pools inside fp32 networks and over features); the activation -> feature boundary is
    global_avgpool_fwd."""
    _rlc32(x, 'x')
    rows, lin, c = x.shape
    out = torch.empty((rows, lin // k, c), device=x.device, dtype=torch.float32)
    _chk(_lib.lib().da_avgpool_fwd(_p(x), c, _p(out), c, rows, lin, k, c, _stream()), 'da_avgpool_fwd')
    return out


def avgpool_bwd(dout, lin, k):
    _rlc32(dout, 'dout')
    rows, lout, c = dout.shape
    dx = torch.empty((rows, lin, c), device=dout.device, dtype=torch.float32)
    _chk(_lib.lib().da_avgpool_bwd(_p(dout), c, _p(dx), c, rows, lin, k, c, _stream()), 'da_avgpool_bwd')
    return dx


def avgpool_slide_fwd(x, k):
    """AvgPool1d(k, stride=1) + ``view(rows, -1)``: (rows, L, C) -> (rows, C * (L - k + 1)), channel slowest."""
    _rlc(x, 'x')
    rows, lin, c = x.shape
    feat = torch.empty((rows, c * (lin - k + 1)), device=x.device, dtype=torch.float32)
    _chk(_lib.lib().da_avgpool_slide_fwd(_p(x), c, _p(feat), rows, lin, k, c, _stream()), 'da_avgpool_slide_fwd')
    return feat


def avgpool_slide_bwd(dfeat, lin, k, c):
    _f32(dfeat, 'dfeat')
    rows = dfeat.shape[0]
    if dfeat.shape[1] != c * (lin - k + 1):
        raise ValueError('dfeat must be (rows, C * (L - k + 1))')
    dx = torch.empty((rows, lin, c), device=dfeat.device, dtype=ACT)
    _chk(_lib.lib().da_avgpool_slide_bwd(_p(dfeat), _p(dx), c, rows, lin, k, c, _stream()), 'da_avgpool_slide_bwd')
    return dx


# ------------------------------------------------------------------------------------------------
# head / loss
# ------------------------------------------------------------------------------------------------
def linear2_fwd(flat, w, bias):
    _f32(flat, 'flat')
    b, k = flat.shape
    if tuple(w.shape) != (2, k):
        raise ValueError('linear_final weight must be (2, %d), got %s' % (k, tuple(w.shape)))
    logits = torch.empty((b, 2), device=flat.device, dtype=torch.float32)
    _chk(_lib.lib().da_linear2_fwd(_p(flat), _p(w), _p(bias), _p(logits), b, k, _stream()), 'da_linear2_fwd')
    return logits


def linear2_bwd(dlogits, flat, w, need_input=True, dw=None, dbias=None, accumulate=False):
    b, k = flat.shape
    dflat = torch.empty_like(flat) if need_input else None
    if dw is None:
        dw = torch.empty_like(w)
        dbias = torch.empty((2,), device=w.device, dtype=torch.float32)
    _chk(_lib.lib().da_linear2_bwd(_p(dlogits), _p(flat), _p(w), _p(dflat), _p(dw), _p(dbias), b, k,
                                   1 if accumulate else 0, _stream()), 'da_linear2_bwd')
    return dflat, dw, dbias


def head_fwd(xmap, w, bias, target, R, finish=False):
    """The head chain of CNNLinearNetwork, forward: xmap (B * R, L, F) the breath block's last map (activation storage type),
    w (2, R * F), bias (2,), target (B, 2) -> flat (B, R * F), part (B, G, 2) the shares of the two dot products of the G blocks a window is pooled by,
    logits (B, 2), loss (1,) -- the last two filled only with ``finish`` (forward-only callers; the training path gets them
    from head_bwd)."""
    _rlc(xmap, 'xmap')
    rows, l, f = xmap.shape
    b = rows // R
    if rows % R or tuple(w.shape) != (2, R * f) or tuple(target.shape) != (b, 2):
        raise ValueError('head_fwd: shapes x%s w%s target%s' % (tuple(xmap.shape), tuple(w.shape), tuple(target.shape)))
    mk = lambda *shape: torch.empty(shape, device=xmap.device, dtype=torch.float32)
    L = _lib.lib()
    flat, part, logits, loss = mk(b, R * f), mk(b, L.da_head_groups(R, f), 2), mk(b, 2), mk(1)
    _chk(L.da_head_fwd(_p(xmap), f, _p(_f32(w)), _p(_f32(bias)), _p(_f32(target)), _p(flat), _p(part), _p(logits), _p(loss), b, R,
                       l, f, 1 if finish else 0, _stream()), 'da_head_fwd')
    return flat, part, logits, loss


def head_bwd(part, bias, target, flat, w, logits, loss, R, l, dw=None, dbias=None, accumulate=False, gscale=1.0):
    """The loss and the backward of head_fwd: fills logits (B, 2) and loss (1,); -> dx (B * R, l, F) in the activation storage
    type, dw (2, R * F), dbias (2,) (+= into the given ones with accumulate)."""
    b, k = flat.shape
    f = k // R
    dx = torch.empty((b * R, l, f), device=flat.device, dtype=ACT)
    dlogits = torch.empty((b, 2), device=flat.device, dtype=torch.float32)
    terms = torch.empty((b,), device=flat.device, dtype=torch.float32)
    if dw is None:
        if accumulate:
            raise ValueError('accumulate needs dw / dbias')
        dw, dbias = torch.empty_like(w), torch.empty((2,), device=w.device, dtype=torch.float32)
    _chk(_lib.lib().da_head_bwd(_p(part), _p(_f32(bias)), _p(_f32(target)), _p(flat), _p(_f32(w)), _p(dx), f, _p(logits), _p(dlogits),
                                _p(terms), _p(dw), _p(dbias), _p(loss), b, R, l, f, gscale, 1 if accumulate else 0, _stream()),
         'da_head_bwd')
    return dx, dw, dbias


def head_flat_fwd(feat, w, bias, target, R, finish=False):
    """head_fwd on features that are pooled already (bn_fwd_pool): feat (B * R, F) float -> flat, part, logits, loss."""
    _f32(feat, 'feat')
    rows, f = feat.shape
    b = rows // R
    if rows % R or tuple(w.shape) != (2, R * f) or tuple(target.shape) != (b, 2):
        raise ValueError('head_flat_fwd: shapes feat%s w%s target%s' % (tuple(feat.shape), tuple(w.shape), tuple(target.shape)))
    mk = lambda *shape: torch.empty(shape, device=feat.device, dtype=torch.float32)
    L = _lib.lib()
    flat, part, logits, loss = mk(b, R * f), mk(b, L.da_head_groups(R, f), 2), mk(b, 2), mk(1)
    _chk(L.da_head_flat_fwd(_p(feat), _p(_f32(w)), _p(_f32(bias)), _p(_f32(target)), _p(flat), _p(part), _p(logits), _p(loss), b, R,
                            f, 1 if finish else 0, _stream()), 'da_head_flat_fwd')
    return flat, part, logits, loss


def head_flat_bwd(part, bias, target, flat, w, logits, loss, R, dw=None, dbias=None, accumulate=False, gscale=1.0):
    """head_bwd for head_flat_fwd: -> dfeat (B * R, F) float (the gradient of the pooled features), dw, dbias."""
    b, k = flat.shape
    f = k // R
    dfeat = torch.empty((b * R, f), device=flat.device, dtype=torch.float32)
    dlogits = torch.empty((b, 2), device=flat.device, dtype=torch.float32)
    terms = torch.empty((b,), device=flat.device, dtype=torch.float32)
    if dw is None:
        if accumulate:
            raise ValueError('accumulate needs dw / dbias')
        dw, dbias = torch.empty_like(w), torch.empty((2,), device=w.device, dtype=torch.float32)
    _chk(_lib.lib().da_head_flat_bwd(_p(part), _p(_f32(bias)), _p(_f32(target)), _p(flat), _p(_f32(w)), _p(dfeat), _p(logits),
                                     _p(dlogits), _p(terms), _p(dw), _p(dbias), _p(loss), b, R, f, gscale, 1 if accumulate else 0,
                                     _stream()), 'da_head_flat_bwd')
    return dfeat, dw, dbias


def bce_logits(logits, target, want_grad=True, gscale=1.0):
    """-> loss (1,), dlogits (same shape as logits) or None."""
    _f32(logits, 'logits')
    _f32(target, 'target')
    if logits.shape != target.shape:
        raise ValueError('BCE: logits %s vs target %s' % (tuple(logits.shape), tuple(target.shape)))
    loss = torch.empty((1,), device=logits.device, dtype=torch.float32)
    d = torch.empty_like(logits) if want_grad else None
    _chk(_lib.lib().da_bce_logits(_p(logits), _p(target), logits.numel(), gscale, _p(loss), _p(d), _stream()),
         'da_bce_logits')
    return loss, d


# ------------------------------------------------------------------------------------------------
# optimiser / misc
# ------------------------------------------------------------------------------------------------
def clamp_sgd_nesterov_(p, g, buf, lr, momentum, weight_decay, clip, first, gscale=1.0):
    _chk(_lib.lib().da_clamp_sgd_nesterov(_p(p), _p(g), _p(buf), p.numel(), lr, momentum, weight_decay,
                                          clip if clip else 0.0, gscale, 1 if first else 0, _stream()),
         'da_clamp_sgd_nesterov')


def clamp_adam_(p, g, m, v, lr, step, clip, beta1=0.9, beta2=0.999, eps=1e-8, gscale=1.0):
    _chk(_lib.lib().da_clamp_adam(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, step,
                                  clip if clip else 0.0, gscale, _stream()), 'da_clamp_adam')


def clamp_adam_dev_(p, g, m, v, lr, step_dev, clip, beta1=0.9, beta2=0.999, eps=1e-8, gscale=1.0):
    """clamp_adam_ with the step count in a device int64 tensor (incremented by the call): graph-capturable."""
    if not (step_dev.is_cuda and step_dev.dtype == torch.int64 and step_dev.numel() == 1):
        raise ValueError('step_dev must be a one-element int64 CUDA tensor')
    _chk(_lib.lib().da_clamp_adam_dev(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, _p(step_dev),
                                      clip if clip else 0.0, gscale, _stream()), 'da_clamp_adam_dev')


def concat2(a, b, drop=None):
    """cat([a, b], channels); drop = (seed, salt, p): dropout (the mask of dropout(b, ...)) on the b half on the way."""
    _rlc(a, 'a')
    _rlc(b, 'b')
    rows, l, c1 = a.shape
    c2 = b.shape[2]
    out = torch.empty((rows, l, c1 + c2), device=a.device, dtype=torch.float32)
    if drop is None:
        _chk(_lib.lib().da_concat2(_p(a), c1, c1, _p(b), c2, c2, _p(out), c1 + c2, rows * l, _stream()), 'da_concat2')
    else:
        seed, salt, p = drop
        _chk(_lib.lib().da_concat2_dropout(_p(a), c1, c1, _p(b), c2, c2, _p(out), c1 + c2, rows * l, _p(seed), salt, p,
                                           _stream()), 'da_concat2_dropout')
    return out


def slice_channels(src, off, c, out=None, accumulate=False, drop=None):
    """out (rows,L,c) (+)= src[:, :, off:off+c]; drop = (seed, salt, p): out = dropout(slice) with the mask of
    dropout() on a contiguous (rows,L,c) tensor (not with accumulate)."""
    _rlc(src, 'src')
    rows, l, cs = src.shape
    if out is None:
        out = torch.empty((rows, l, c), device=src.device, dtype=torch.float32)
    if drop is None:
        _chk(_lib.lib().da_slice_copy(_p(src), cs, off, _p(out), c, c, rows * l, 1 if accumulate else 0, _stream()),
             'da_slice_copy')
    else:
        if accumulate:
            raise ValueError('slice_channels: drop and accumulate are exclusive')
        seed, salt, p = drop
        _chk(_lib.lib().da_slice_dropout(_p(src), cs, off, _p(out), c, c, rows * l, _p(seed), salt, p, _stream()),
             'da_slice_dropout')
    return out


def dropout(x, seed, salt, p):
    """seed: int64 CUDA tensor of one element (device-resident so the step stays graph-capturable)."""
    y = torch.empty_like(x)
    _chk(_lib.lib().da_dropout(_p(x), _p(y), x.numel(), _p(seed), salt, p, _stream()), 'da_dropout')
    return y


# ------------------------------------------------------------------------------------------------
# device-resident window store
# ------------------------------------------------------------------------------------------------
def gather_normalize(tiles, idx, mu, std, out=None):
    """tiles (N, ...) float64 CUDA raw windows, idx (B,) int64 CUDA -> (B, ...) float32 = float((x-mu)/std).
    The kernel reads tiles[idx[b]] unchecked: idx must already be known to lie in [0, N) (DeviceTileStore.batch
    validates host indices before the upload and builds device-side index lists only from validated ones)."""
    if not (tiles.is_cuda and tiles.dtype == torch.float64 and tiles.is_contiguous()):
        raise ValueError('tiles must be a contiguous float64 CUDA tensor')
    if not (idx.is_cuda and idx.dtype == torch.int64 and idx.is_contiguous()):
        raise ValueError('idx must be a contiguous int64 CUDA tensor')
    b = idx.numel()
    elems = tiles[0].numel()
    shape = (b,) + tuple(tiles.shape[1:])
    if out is None:
        out = torch.empty(shape, device=tiles.device, dtype=torch.float32)
    elif not (out.is_cuda and out.dtype == torch.float32 and out.is_contiguous() and tuple(out.shape) == shape):
        raise ValueError('gather_normalize: out must be a contiguous float32 CUDA tensor of shape %s, got %s %s' %
                         (shape, tuple(out.shape), out.dtype))
    if isinstance(mu, (tuple, list)):          # per-channel factors: (N, NB, C, L) windows with FFT channels
        if tiles.dim() != 4 or len(mu) != tiles.shape[2] or len(std) != tiles.shape[2] or tiles.shape[2] > 4:
            raise ValueError('gather_normalize: one (mu, std) per channel of (N, NB, C <= 4, L) tiles expected')
        c = tiles.shape[2]
        dbl = ctypes.c_double * c
        _chk(_lib.lib().da_gather_normalize_ch(_p(tiles), _p(idx), dbl(*[float(v) for v in mu]), dbl(*[float(v) for v in std]),
                                               _p(out), b, tiles.shape[1], c, tiles.shape[3], _stream()),
             'da_gather_normalize_ch')
        return out
    _chk(_lib.lib().da_gather_normalize(_p(tiles), _p(idx), float(mu), float(std), _p(out), b, elems, _stream()),
         'da_gather_normalize')
    return out


def gather_rows(src, idx, out=None):
    """src (N, W) float32 CUDA, idx (B,) int64 CUDA -> (B, W)."""
    _f32(src, 'src')
    if not (idx.is_cuda and idx.dtype == torch.int64 and idx.is_contiguous()):
        raise ValueError('idx must be a contiguous int64 CUDA tensor')
    b, width = idx.numel(), src.shape[1]
    if out is None:
        out = torch.empty((b, width), device=src.device, dtype=torch.float32)
    elif not (out.is_cuda and out.dtype == torch.float32 and out.is_contiguous() and tuple(out.shape) == (b, width)):
        raise ValueError('gather_rows: out must be a contiguous float32 CUDA tensor of shape %s' % ((b, width),))
    _chk(_lib.lib().da_gather_rows(_p(src), _p(idx), _p(out), b, width, _stream()), 'da_gather_rows')
    return out


def window_median_fwd(feat, nb):
    """feat (B*NB, F) -> (median (B, F), idx (B, F) int32): lower median over the NB rows of every window."""
    _f32(feat, 'feat')
    rows, f = feat.shape
    if rows % nb or nb > 64:
        raise ValueError('window_median: rows %d not a multiple of NB %d (<= 64)' % (rows, nb))
    b = rows // nb
    out = torch.empty((b, f), device=feat.device, dtype=torch.float32)
    idx = torch.empty((b, f), device=feat.device, dtype=torch.int32)
    _chk(_lib.lib().da_window_median_fwd(_p(feat), f, b, nb, f, _p(out), _p(idx), _stream()), 'da_window_median_fwd')
    return out, idx


def window_median_bwd(dout, idx, nb):
    _f32(dout, 'dout')
    b, f = dout.shape
    dx = torch.empty((b * nb, f), device=dout.device, dtype=torch.float32)
    _chk(_lib.lib().da_window_median_bwd(_p(dout), _p(idx), b, nb, f, _p(dx), f, _stream()), 'da_window_median_bwd')
    return dx


def lstm_fwd(gx, whh, bih, bhh, h0=None, c0=None):
    """gx (B,T,4H) = x W_ih^T -> hs, cs (B,T,H), gates (B,T,4H), hT, cT (B,H): the recurrence of nn.LSTM (1 layer)."""
    _f32(gx, 'gx')
    b, t, g4 = gx.shape
    h = g4 // 4
    mk = lambda *shape: torch.empty(shape, device=gx.device, dtype=torch.float32)
    hs, cs, gates, ht, ct = mk(b, t, h), mk(b, t, h), mk(b, t, g4), mk(b, h), mk(b, h)
    _chk(_lib.lib().da_lstm_fwd(_p(gx), _p(whh), _p(bih), _p(bhh), _p(h0), _p(c0), _p(hs), _p(cs), _p(gates), _p(ht), _p(ct),
                                b, t, h, _stream()), 'da_lstm_fwd')
    return hs, cs, gates, ht, ct


def lstm_bwd(dh_all, whh, hs, cs, gates, h0=None, c0=None):
    """-> dgates (B,T,4H), dwhh_part (B,4H,H)."""
    _f32(dh_all, 'dh_all')
    b, t, h = dh_all.shape
    dgates = torch.empty((b, t, 4 * h), device=dh_all.device, dtype=torch.float32)
    part = torch.empty((b, 4 * h, h), device=dh_all.device, dtype=torch.float32)
    _chk(_lib.lib().da_lstm_bwd(_p(dh_all), _p(whh), _p(hs), _p(cs), _p(gates), _p(h0), _p(c0), _p(dgates), _p(part),
                                b, t, h, _stream()), 'da_lstm_bwd')
    return dgates, part


def reduce_rows(m, out=None, accumulate=False):
    """column sums of m (rows, ...) -> (...), fixed order; accumulate adds into out."""
    _f32(m, 'm')
    rows = m.shape[0]
    n = m.numel() // max(rows, 1)
    if out is None:
        out = torch.empty(m.shape[1:], device=m.device, dtype=torch.float32)
    _chk(_lib.lib().da_reduce_rows(_p(m), rows, n, _p(out), 1 if accumulate else 0, _stream()), 'da_reduce_rows')
    return out


def vote_counts(logits, group, votes, want_pred=True):
    """logits (B,2) f32, group (B,) int64 patient slot per window, votes (P,2) int32 accumulated in place.
    -> pred (B,) int32 window predictions (argmax, class 0 on ties)."""
    _f32(logits, 'logits')
    if not (group.is_cuda and group.dtype == torch.int64 and votes.is_cuda and votes.dtype == torch.int32
            and votes.is_contiguous() and votes.shape[1] == 2):
        raise ValueError('group must be int64 CUDA, votes (P,2) int32 CUDA')
    b = logits.shape[0]
    pred = torch.empty((b,), device=logits.device, dtype=torch.int32) if want_pred else None
    _chk(_lib.lib().da_vote_counts(_p(logits), _p(group.contiguous()), b, votes.shape[0], _p(votes), _p(pred), _stream()),
         'da_vote_counts')
    return pred
