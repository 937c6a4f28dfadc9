"""CPU oracle (TEST INFRASTRUCTURE ONLY) -- numpy restatement of the deepards cnn_linear hot path.

This module restates, in plain numpy (float64 by default), the arithmetic the reference executes
through stock ``torch.nn`` modules on the path named by BASELINE.json's ``north_star``.  It is the
*checker* for the HIP kernels in ``deepards_amd/csrc``: only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it.  The product path never does.

Pinned by: ``tests/golden/*.npz`` (logits / loss / grads / post-step params captured from the real
reference imported in the build container, see ``oracle/make_golden.py``).  The reference's own tests
hold no golden vectors for this path so the goldens come from that import.

Layout here is the reference's own: activations ``(rows, C, L)`` ("NCL"), conv weights
``(C_out, C_in, k)``.  All functions are pure.

Reference citations (relative to /root/reference/deepards):
  * models/resnet.py:5-8,11-40,81-163      conv2x2 / BasicBlock / ResNet.forward
  * models/densenet.py:18-44,68-81,83-194   _DenseLayer / _Transition / DenseNet.forward
  * models/torch_cnn_linear_network.py:92-113  CNNLinearNetwork.forward (per-window loop)
  * train_ards_detector.py:161-173,416-422,474-476,526-532  loss / optimiser / clamp hooks
"""
import numpy as np

EPS = 1e-5


# --------------------------------------------------------------------------------------------
# leaf ops (each one == the torch.nn op the reference reaches)
# --------------------------------------------------------------------------------------------
def conv1d_fwd(x, w, stride=1, pad=0):
    """nn.Conv1d(bias=False).  x (N,Ci,L), w (Co,Ci,k) -> (N,Co,Lo).  resnet.py:5-8,86-87."""
    n, ci, l = x.shape
    co, ci2, k = w.shape
    assert ci == ci2
    lo = (l + 2 * pad - k) // stride + 1
    xp = np.zeros((n, ci, l + 2 * pad), dtype=x.dtype)
    xp[:, :, pad:pad + l] = x
    y = np.zeros((n, co, lo), dtype=np.result_type(x, w))
    for t in range(k):
        xs = xp[:, :, t:t + (lo - 1) * stride + 1:stride]          # (N,Ci,Lo)
        y += np.einsum('oc,ncl->nol', w[:, :, t], xs, optimize=True)
    return y


def conv1d_bwd(x, w, dy, stride=1, pad=0, need_dx=True):
    """Gradients of conv1d_fwd: returns (dx, dw)."""
    n, ci, l = x.shape
    co, _, k = w.shape
    lo = dy.shape[2]
    xp = np.zeros((n, ci, l + 2 * pad), dtype=x.dtype)
    xp[:, :, pad:pad + l] = x
    dxp = np.zeros_like(xp, dtype=np.result_type(x, w))
    dw = np.zeros_like(w)
    for t in range(k):
        sl = slice(t, t + (lo - 1) * stride + 1, stride)
        dw[:, :, t] = np.einsum('nol,ncl->oc', dy, xp[:, :, sl], optimize=True)
        if need_dx:
            dxp[:, :, sl] += np.einsum('oc,nol->ncl', w[:, :, t], dy, optimize=True)
    dx = dxp[:, :, pad:pad + l] if need_dx else None
    return dx, dw


def bn_window_fwd(x, gamma, beta, rows_per_window, eps=EPS):
    """Train-mode nn.BatchNorm1d applied one window at a time (torch_cnn_linear_network.py:108-113
    calls breath_block(x[i]) per window, so the batch statistics are over (rows_per_window, L) per
    channel *per window*; SURVEY.md finding 3).  Returns y, (mean, invstd) with shape (W, C)."""
    n, c, l = x.shape
    w = n // rows_per_window
    xv = x.reshape(w, rows_per_window, c, l)
    mean = xv.mean(axis=(1, 3))                                     # (W,C)
    var = xv.var(axis=(1, 3))                                       # biased
    invstd = 1.0 / np.sqrt(var + eps)
    xhat = (xv - mean[:, None, :, None]) * invstd[:, None, :, None]
    y = xhat * gamma[None, None, :, None] + beta[None, None, :, None]
    return y.reshape(n, c, l), (mean, invstd)


def bn_window_bwd(x, gamma, stats, dy, rows_per_window):
    """Backward of bn_window_fwd.  Returns dx, dgamma, dbeta (param grads summed over windows)."""
    n, c, l = x.shape
    w = n // rows_per_window
    mean, invstd = stats
    xv = x.reshape(w, rows_per_window, c, l)
    dv = dy.reshape(w, rows_per_window, c, l)
    xhat = (xv - mean[:, None, :, None]) * invstd[:, None, :, None]
    cnt = rows_per_window * l
    s1 = dv.sum(axis=(1, 3))                                        # (W,C)
    s2 = (dv * xhat).sum(axis=(1, 3))
    dx = (gamma * invstd)[:, None, :, None] * (dv - s1[:, None, :, None] / cnt
                                               - xhat * s2[:, None, :, None] / cnt)
    return dx.reshape(n, c, l), s2.sum(axis=0), s1.sum(axis=0)


def bn_running_update(running_mean, running_var, stats, count, momentum=0.1, eps=EPS):
    """Sequential per-window running-stat update (SURVEY.md finding 5): unbiased var, momentum .1."""
    mean, invstd = stats
    var_b = 1.0 / (invstd ** 2) - eps
    rm, rv = running_mean.copy(), running_var.copy()
    for i in range(mean.shape[0]):
        rm = (1 - momentum) * rm + momentum * mean[i]
        rv = (1 - momentum) * rv + momentum * var_b[i] * count / (count - 1)
    return rm, rv


def relu(x):
    return np.maximum(x, 0)


def maxpool3s2p1_fwd(x):
    """nn.MaxPool1d(3,2,1) (resnet.py:100-102, densenet.py:123).  Returns y and argmax index
    (first max wins, as ATen)."""
    n, c, l = x.shape
    lo = (l + 2 - 3) // 2 + 1
    xp = np.full((n, c, l + 2), -np.inf, dtype=x.dtype)
    xp[:, :, 1:l + 1] = x
    cand = np.stack([xp[:, :, t:t + (lo - 1) * 2 + 1:2] for t in range(3)], axis=-1)  # (N,C,Lo,3)
    arg = cand.argmax(axis=-1)
    y = np.take_along_axis(cand, arg[..., None], axis=-1)[..., 0]
    idx = np.arange(lo)[None, None, :] * 2 - 1 + arg                # position in un-padded x
    return y, idx


def maxpool3s2p1_bwd(dy, idx, l):
    n, c, lo = dy.shape
    dx = np.zeros((n, c, l), dtype=dy.dtype)
    nn_, cc = np.meshgrid(np.arange(n), np.arange(c), indexing='ij')
    for j in range(lo):
        np.add.at(dx, (nn_, cc, idx[:, :, j]), dy[:, :, j])
    return dx


def avgpool_fwd(x, k, stride):
    """nn.AvgPool1d(k, stride) no padding (resnet.py:112, densenet.py:79,167)."""
    n, c, l = x.shape
    lo = (l - k) // stride + 1
    y = np.zeros((n, c, lo), dtype=x.dtype)
    for t in range(k):
        y += x[:, :, t:t + (lo - 1) * stride + 1:stride]
    return y / k


def avgpool_bwd(dy, k, stride, l):
    n, c, lo = dy.shape
    dx = np.zeros((n, c, l), dtype=dy.dtype)
    for t in range(k):
        dx[:, :, t:t + (lo - 1) * stride + 1:stride] += dy / k
    return dx


def avgpool3s2p1_fwd(x):
    """nn.AvgPool1d(3,2,1) (resnet first_pool_type='avg', resnet.py:103-104; count_include_pad)."""
    n, c, l = x.shape
    xp = np.zeros((n, c, l + 2), dtype=x.dtype)
    xp[:, :, 1:l + 1] = x
    return avgpool_fwd(xp, 3, 2)


def avgpool3s2p1_bwd(dy, l):
    return avgpool_bwd(dy, 3, 2, l + 2)[:, :, 1:l + 1]


def linear_fwd(x, w, b):
    return x @ w.T + b


def bce_with_logits(x, t):
    """torch.nn.BCEWithLogitsLoss() mean reduction (train_ards_detector.py:530,929-930).
    Returns loss, dloss/dx."""
    loss = np.maximum(x, 0) - x * t + np.log1p(np.exp(-np.abs(x)))
    sig = 1.0 / (1.0 + np.exp(-x))
    return loss.mean(), (sig - t) / x.size


def clamp_grad(g, clip):
    """register_hook(lambda g: torch.clamp(g, -clip, clip)) (train_ards_detector.py:474-476)."""
    return np.clip(g, -clip, clip)


def sgd_nesterov_step(p, g, buf, lr=1e-3, momentum=0.9, wd=1e-4, first=False):
    """torch.optim.SGD(momentum=.9, weight_decay, nesterov=True) (train_ards_detector.py:421).
    Returns new p, new buf."""
    g = g + wd * p
    buf = g.copy() if first or buf is None else momentum * buf + g
    g = g + momentum * buf
    return p - lr * g, buf


def adam_step(p, g, m, v, step, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam(lr) defaults (train_ards_detector.py:419).  step counts from 1."""
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    mhat = m / (1 - b1 ** step)
    vhat = v / (1 - b2 ** step)
    return p - lr * mhat / (np.sqrt(vhat) + eps), m, v


# --------------------------------------------------------------------------------------------
# network assembly: a tiny tape so forward and backward stay in one place
# --------------------------------------------------------------------------------------------
class _Tape(object):
    """Records closures; backward replays them in reverse.  Values are numpy arrays."""

    def __init__(self, params, rows_per_window):
        self.p = params
        self.g = {}
        self.R = rows_per_window
        self.stats = {}
        self.bf16_storage = False    # BASELINE configs C3 / C5 (build-side, not in the reference): every activation and
                                     # activation gradient that crosses a kernel boundary on the device is rounded to bf16
        self.decisions = {}          # name -> dict(kind, ...): every ReLU mask / max-pool argmax of the forward, mutable
        self.order = []              # decision names in forward order

    def acc(self, name, g):
        self.g[name] = self.g.get(name, 0) + g

    def rs(self, a):
        """A tensor as the device stores it between two kernels: bf16-rounded under bf16 storage, else unchanged."""
        return round_bf16(a) if self.bf16_storage else a


def round_bf16(a):
    """Round-to-nearest-even to bfloat16 precision (the operand rounding of v_cvt_pk_bf16_f32), returned as float64."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    r = ((u + np.uint32(0x7fff) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xffff0000)).view(np.float32)
    return r.astype(np.float64)


def _conv(t, x, name, stride, pad, need_dx=True):
    w = t.p[name]
    # BASELINE config C3 arithmetic (build-side, not in the reference): the k3 s1 p1 / k3 s2 p1 / k1 s2 p0 convs with
    # channel counts that are multiples of 64 (stride 2: even lengths) see bf16-rounded operands in the forward, the
    # data gradient and the weight gradient; sums and everything else stay exact here (fp32 on the device)
    on = getattr(t, 'bf16_convs', False) and w.shape[0] % 64 == 0 and w.shape[1] % 64 == 0
    k = w.shape[2]
    bf_w = on and ((k == 3 and pad == 1 and stride in (1, 2)) or (k == 1 and pad == 0 and stride == 2)) and \
        (stride == 1 or x.shape[2] % 2 == 0)
    bf_fd = bf_w
    y = conv1d_fwd(round_bf16(x), round_bf16(w), stride, pad) if bf_fd else conv1d_fwd(x, w, stride, pad)
    y = t.rs(y)

    def bwd(dy):
        dx, dw = conv1d_bwd(x, w, dy, stride, pad, need_dx)
        if bf_fd:                                     # dx from rounded (dy, w)
            dx, _ = conv1d_bwd(x, round_bf16(w), round_bf16(dy), stride, pad, need_dx)
        if bf_w:                                      # dw from rounded (dy, x)
            _, dw = conv1d_bwd(round_bf16(x), w, round_bf16(dy), stride, pad, False)
        t.acc(name, dw)
        return t.rs(dx) if dx is not None else dx
    return y, bwd


def _bn(t, x, prefix, store=False):
    """store: the BatchNorm output itself is written to memory (the downsample branch); otherwise it is consumed inside
    the same kernel by the ReLU / residual add that follows and only THAT result is stored (and rounded, bf16 storage)."""
    gname, bname = prefix + '.weight', prefix + '.bias'
    y, st = bn_window_fwd(x, t.p[gname], t.p[bname], t.R)
    t.stats[prefix] = (st, t.R * x.shape[2])
    if store:
        y = t.rs(y)

    def bwd(dy):
        dx, dg, db = bn_window_bwd(x, t.p[gname], st, dy, t.R)
        t.acc(gname, dg)
        t.acc(bname, db)
        return t.rs(dx)
    return y, bwd


def _relu(t, x, name, store=True):
    """ReLU whose backward mask lives on the tape (t.decisions[name]['mask']) and is read when the backward RUNS, so a
    test can re-run the backward under another admissible decision of an element whose pre-activation is within fp32
    noise of zero (``rebackward``).  store: the result is written to memory (rounded under bf16 storage)."""
    y = relu(x)
    d = t.decisions[name] = dict(kind='relu', pre=x, mask=y > 0)
    t.order.append(name)
    return (t.rs(y) if store else y), (lambda dy: dy * d['mask'])


def _maxpool(t, x, name):
    y, idx = maxpool3s2p1_fwd(x)
    d = t.decisions[name] = dict(kind='maxpool', x=x, idx=idx)
    t.order.append(name)
    return t.rs(y), (lambda dy: t.rs(maxpool3s2p1_bwd(dy, d['idx'], x.shape[2])))


def _stem(t, x, conv, bn, pool_type='max'):
    if t.bf16_storage and getattr(t, 'stem_recomputed', False):
        # the device's recomputing stem (deepards_amd/csrc/stem_pool.hip: stem_bn_relu_pool_fwd_kernel / stem_bwd_kernel) stores
        # nothing at the stem's resolution: only the pooled map is rounded, and the backward keeps fp32 up to the weights
        t.bf16_storage = False
        try:
            y3, bwd = _stem(t, x, conv, bn, pool_type)
        finally:
            t.bf16_storage = True

        def bwd_unrounded(d):
            t.bf16_storage = False
            try:
                return bwd(d)
            finally:
                t.bf16_storage = True
        return round_bf16(y3), bwd_unrounded
    y0, b_conv = _conv(t, x, conv, 2, 3, need_dx=False)
    y1, b_bn = _bn(t, y0, bn)
    y2, b_relu = _relu(t, y1, bn + '.relu', store=False)       # relu -> pool inside one kernel
    if pool_type == 'max':
        y3, b_pool = _maxpool(t, y2, bn + '.maxpool')
    else:
        y3 = t.rs(avgpool3s2p1_fwd(y2))
        b_pool = lambda d: t.rs(avgpool3s2p1_bwd(d, y2.shape[2]))
    return y3, (lambda d: b_conv(b_bn(b_relu(b_pool(d)))))


def _stem_double(t, x, prefix, pool_type='max'):
    """ResNet.forward with double_conv_first (resnet.py:144-153): conv1_alt (k3 s1 p1) -> bn1 -> conv2 (k7 s2 p3) -> bn2
    -> relu -> first_pool; no ReLU between bn1 and conv2."""
    ya, b_ca = _conv(t, x, prefix + 'conv1_alt.weight', 1, 1, need_dx=False)
    h, b_bn1 = _bn(t, ya, prefix + 'bn1', store=True)
    y2, b_c2 = _conv(t, h, prefix + 'conv2.weight', 2, 3)
    y3, b_bn2 = _bn(t, y2, prefix + 'bn2')
    y4, b_relu = _relu(t, y3, prefix + 'bn2.relu', store=False)
    if pool_type == 'max':
        y5, b_pool = _maxpool(t, y4, prefix + 'bn2.maxpool')
    else:
        y5 = t.rs(avgpool3s2p1_fwd(y4))
        b_pool = lambda d: t.rs(avgpool3s2p1_bwd(d, y4.shape[2]))
    return y5, (lambda d: b_ca(b_bn1(b_c2(b_bn2(b_relu(b_pool(d)))))))


RESNET_LAYERS = {'resnet18': (2, 2, 2, 2), 'resnet34': (3, 4, 6, 3)}              # resnet.py:166-187
DENSENET_BLOCKS = {'densenet18': (2, 2, 2, 2), 'densenet121': (6, 12, 24, 16),    # densenet.py:223-275 (growth 32)
                   'densenet169': (6, 12, 32, 32), 'densenet201': (6, 12, 48, 32)}


def resnet18_features(t, x, prefix='breath_block.', first_pool_type='max', layers=(2, 2, 2, 2), double_conv_first=False):
    """ResNet.forward (resnet.py:141-163) with BasicBlock (resnet.py:24-40); layers [2,2,2,2] = resnet18,
    [3,4,6,3] = resnet34."""
    if double_conv_first:
        h, b_stem = _stem_double(t, x, prefix, first_pool_type)
    else:
        h, b_stem = _stem(t, x, prefix + 'conv1.weight', prefix + 'bn1', first_pool_type)
    backs = [b_stem]
    inpl = 64
    for li, planes in enumerate([64, 128, 256, 512]):
        for bi in range(layers[li]):
            stride = 2 if (li > 0 and bi == 0) else 1
            bp = '%slayer%d.%d.' % (prefix, li + 1, bi)
            xin = h
            o, b1 = _conv(t, xin, bp + 'conv1.weight', stride, 1)
            o, b2 = _bn(t, o, bp + 'bn1')
            o, b3 = _relu(t, o, bp + 'relu1')
            o, b4 = _conv(t, o, bp + 'conv2.weight', 1, 1)
            o, b5 = _bn(t, o, bp + 'bn2')
            if stride != 1 or inpl != planes:
                r, d1 = _conv(t, xin, bp + 'downsample.0.weight', stride, 0)
                r, d2 = _bn(t, r, bp + 'downsample.1', store=True)
                b_res = (lambda d1, d2: (lambda d: d1(d2(d))))(d1, d2)
            else:
                r = xin
                b_res = lambda d: d
            h, b6 = _relu(t, o + r, bp + 'relu2')

            def blk_bwd(d, b1=b1, b2=b2, b3=b3, b4=b4, b5=b5, b6=b6, b_res=b_res):
                dz = b6(d)
                return t.rs(b1(b2(b3(b4(b5(dz))))) + b_res(dz))
            backs.append(blk_bwd)
            inpl = planes
    feat = avgpool_fwd(h, 7, 1)
    lh = h.shape[2]
    n = x.shape[0]
    out = feat.reshape(n, -1)

    def bwd(dout):
        d = t.rs(avgpool_bwd(dout.reshape(feat.shape), 7, 1, lh))
        for b in reversed(backs):
            d = b(d)
        return d
    return out, bwd


def densenet18_features(t, x, prefix='breath_block.', drop_masks=None, block_config=(2, 2, 2, 2)):
    """DenseNet.forward (densenet.py:179-189): stem, 4 blocks x 2 _DenseLayer (densenet.py:18-41,
    pre-activation BN->ReLU->conv1x1->BN->ReLU->conv3 -> dropout -> cat), _Transition
    (densenet.py:68-79), norm5 -> relu -> AvgPool1d(7,1).  Dropout is OFF unless explicit
    keep-masks (already scaled by 1/(1-p)) are passed: drop_masks[(block, layer)] (N,32,L)."""
    fp = prefix + 'features.'
    h, b_stem = _stem(t, x, fp + 'conv0.weight', fp + 'norm0', 'max')
    backs = [b_stem]
    for bi in range(1, 5):
        for li in range(1, block_config[bi - 1] + 1):
            lp = '%sdenseblock%d.denselayer%d.' % (fp, bi, li)
            xin = h
            o, b1 = _bn(t, xin, lp + 'norm1')
            o, b2 = _relu(t, o, lp + 'relu1')
            o, b3 = _conv(t, o, lp + 'conv1.weight', 1, 0)
            o, b4 = _bn(t, o, lp + 'norm2')
            o, b5 = _relu(t, o, lp + 'relu2')
            o, b6 = _conv(t, o, lp + 'conv2.weight', 1, 1)
            mask = None if drop_masks is None else drop_masks.get((bi, li))
            if mask is not None:
                o = o * mask
            cin = xin.shape[1]
            h = np.concatenate([xin, o], axis=1)

            def lay_bwd(d, b1=b1, b2=b2, b3=b3, b4=b4, b5=b5, b6=b6, cin=cin, mask=mask):
                dnew = d[:, cin:]
                if mask is not None:
                    dnew = dnew * mask
                return d[:, :cin] + b1(b2(b3(b4(b5(b6(dnew))))))
            backs.append(lay_bwd)
        if bi != 4:
            tp = '%stransition%d.' % (fp, bi)
            o, b1 = _bn(t, h, tp + 'norm')
            o, b2 = _relu(t, o, tp + 'relu')
            o, b3 = _conv(t, o, tp + 'conv.weight', 1, 0)
            lin = o.shape[2]
            h = avgpool_fwd(o, 2, 2)
            backs.append((lambda b1, b2, b3, lin: (lambda d: b1(b2(b3(avgpool_bwd(d, 2, 2, lin))))))(b1, b2, b3, lin))
    o, b1 = _bn(t, h, fp + 'norm5')
    o, b2 = _relu(t, o, fp + 'relu5')
    lh = o.shape[2]
    feat = avgpool_fwd(o, 7, 1)
    n = x.shape[0]
    out = feat.reshape(n, -1)

    def bwd(dout):
        d = b1(b2(avgpool_bwd(dout.reshape(feat.shape), 7, 1, lh)))
        for b in reversed(backs):
            d = b(d)
        return d
    return out, bwd


def ambiguous_decisions(t, tol):
    """Activation decisions of the forward that an fp32 implementation may legitimately take the other way: ReLU
    elements with |pre-activation| < tol, max-pool windows whose best and second-best candidates (at different
    positions, best > 0) are closer than tol.  -> [(name, flat index, margin)] in forward order."""
    out = []
    for name in t.order:
        d = t.decisions[name]
        if d['kind'] == 'relu':
            for i in np.flatnonzero(np.abs(d['pre']) < tol):
                out.append((name, int(i), float(abs(d['pre'].flat[i]))))
        else:
            x = d['x']
            n, c, l = x.shape
            lo = d['idx'].shape[2]
            xp = np.full((n, c, l + 2), -np.inf)
            xp[:, :, 1:l + 1] = x
            cand = np.stack([xp[:, :, k:k + (lo - 1) * 2 + 1:2] for k in range(3)], axis=-1)
            srt = np.sort(cand, axis=-1)
            gap = srt[..., 2] - srt[..., 1]
            for i in np.flatnonzero((gap < tol) & (srt[..., 2] > 0)):
                out.append((name, int(i), float(gap.flat[i])))
    return out


def _apply_flip(t, name, i):
    d = t.decisions[name]
    if d['kind'] == 'relu':
        d['mask'] = d['mask'].copy()
        d['mask'].flat[i] = ~d['mask'].flat[i]
    else:                                                     # route the gradient to the runner-up position instead
        x = d['x']
        n, c, l = x.shape
        lo = d['idx'].shape[2]
        ni, ci, j = np.unravel_index(i, (n, c, lo))
        best = int(d['idx'][ni, ci, j])
        pos = [q for q in (2 * j - 1, 2 * j, 2 * j + 1) if 0 <= q < l and q != best]
        other = max(pos, key=lambda q: x[ni, ci, q])
        d['idx'] = d['idx'].copy()
        d['idx'][ni, ci, j] = other


HEADS = ('linear', 'to_mean', 'compr_to_rf', 'single_breath', 'double_linear', 'lstm')


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def lstm_fwd(x, w_ih, w_hh, b_ih, b_hh, h0=None, c0=None):
    """nn.LSTM(F, H, num_layers=1, batch_first=True): x (B,T,F) -> h (B,T,H), (hT, cT), tape.  Gate order i, f, g, o."""
    b, t, _ = x.shape
    hd = w_hh.shape[1]
    h = np.zeros((b, hd)) if h0 is None else h0
    c = np.zeros((b, hd)) if c0 is None else c0
    hs, tape = [], []
    for s in range(t):
        z = x[:, s] @ w_ih.T + b_ih + h @ w_hh.T + b_hh
        i, f, g, o = _sigmoid(z[:, :hd]), _sigmoid(z[:, hd:2 * hd]), np.tanh(z[:, 2 * hd:3 * hd]), _sigmoid(z[:, 3 * hd:])
        c_new = f * c + i * g
        tc = np.tanh(c_new)
        h_new = o * tc
        tape.append((h, c, i, f, g, o, tc))
        h, c = h_new, c_new
        hs.append(h)
    return np.stack(hs, axis=1), (h, c), tape


def lstm_bwd(x, w_ih, w_hh, tape, dh_all):
    """-> dx (B,T,F), dw_ih, dw_hh, db (= db_ih = db_hh); no gradient into the initial state (the reference detaches it,
    train_ards_detector.py:848)."""
    b, t, _ = x.shape
    hd = w_hh.shape[1]
    dx = np.zeros_like(x)
    dw_ih, dw_hh, db = np.zeros_like(w_ih), np.zeros_like(w_hh), np.zeros(4 * hd)
    dh_next, dc_next = np.zeros((b, hd)), np.zeros((b, hd))
    for s in range(t - 1, -1, -1):
        h_prev, c_prev, i, f, g, o, tc = tape[s]
        dh = dh_all[:, s] + dh_next
        do = dh * tc * o * (1 - o)
        dc = dh * o * (1 - tc * tc) + dc_next
        di = dc * g * i * (1 - i)
        df = dc * c_prev * f * (1 - f)
        dg = dc * i * (1 - g * g)
        dz = np.concatenate([di, df, dg, do], axis=1)
        dx[:, s] = dz @ w_ih
        dw_ih += dz.T @ x[:, s]
        dw_hh += dz.T @ h_prev
        db += dz.sum(axis=0)
        dh_next = dz @ w_hh
        dc_next = dc * f
    return dx, dw_ih, dw_hh, db


def cnn_linear_forward_backward(params, x, target, backbone='resnet18', n_sub_batches=20,
                                first_pool_type='max', drop_masks=None, need_grads=True, head='linear',
                                bf16_convs=False, bf16_storage=False, double_conv_first=False, stem_recomputed=True):
    """CNNLinearNetwork.forward over a batch (torch_cnn_linear_network.py:104-113) + BCE loss
    (train_ards_detector.py:929-930) + backward.  x (B,NB,C,224); target (B,2) one-hot.
    params: dict name -> ndarray with the reference's state_dict keys.
    Returns dict(logits, loss, grads{name}, feat, stats).

    head selects the sibling networks on the same breath block (torch_cnn_linear_network.py:7-89):
      'to_mean'        CNNLinearToMean:              Linear(F,2)(mean over the NB breaths)              -> (B,2)
      'compr_to_rf'    CNNLinearComprToRF:           Linear(F,2)(torch.median over NB = LOWER median)   -> (B,2)
      'single_breath'  CNNSingleBreathLinearNetwork: Linear(F,2) per breath                             -> (B,NB,2)
      'double_linear'  CNNDoubleLinearNetwork:       Linear(2 NB,2)(flatten(Linear(F,2) per breath))    -> (B,2)
      'lstm'           CNNLSTMNetwork (torch_cnn_lstm_combo.py:6-50, zero initial state, no metadata):
                                                     Linear(H,2)(LSTM over the NB breath features)       -> (B,NB,2)
    The loss of a (B,NB,2) output repeats the window target over the breaths (PerBreathClassifierMixin.calc_loss,
    train_ards_detector.py:540-543)."""
    if x.shape[-1] != 224:
        raise Exception('input breaths must have sequence length of 224')
    if head not in HEADS:
        raise ValueError(head)
    b, nb, c, l = x.shape
    t = _Tape(params, nb)
    t.bf16_convs = bf16_convs
    t.bf16_storage = bf16_storage            # resnets only (the device has no bf16-storage DenseNet)
    t.stem_recomputed = stem_recomputed      # (bf16 storage only: which of the device's two stems is mirrored, see _stem)
    rows = x.reshape(b * nb, c, l)
    if backbone in RESNET_LAYERS:
        feat, fbwd = resnet18_features(t, rows, first_pool_type=first_pool_type, layers=RESNET_LAYERS[backbone],
                                       double_conv_first=double_conv_first)
    elif backbone in DENSENET_BLOCKS:
        feat, fbwd = densenet18_features(t, rows, drop_masks=drop_masks, block_config=DENSENET_BLOCKS[backbone])
    else:
        raise ValueError(backbone)
    w, bias = params['linear_final.weight'], params['linear_final.bias']
    f3 = feat.reshape(b, nb, -1)
    med_idx = None
    if head == 'linear':
        flat = feat.reshape(b, -1)                                   # view(-1) of (NB,F) per window
    elif head == 'to_mean':
        flat = f3.mean(axis=1)
    elif head == 'compr_to_rf':
        med_idx = np.argsort(f3, axis=1, kind='stable')[:, (nb - 1) // 2, :]        # lower median (torch.median)
        flat = np.take_along_axis(f3, med_idx[:, None, :], axis=1)[:, 0, :]
    elif head == 'single_breath':
        flat = feat                                                  # (B*NB, F)
    elif head == 'lstm':
        lw = [params['lstm.' + k] for k in ('weight_ih_l0', 'weight_hh_l0', 'bias_ih_l0', 'bias_hh_l0')]
        hseq, (h_t, c_t), ltape = lstm_fwd(f3, *lw)
        flat = hseq.reshape(b * nb, -1)
    else:
        wi, bi = params['linear_intermediate.weight'], params['linear_intermediate.bias']
        inter = linear_fwd(feat, wi, bi)                             # (B*NB, 2)
        flat = inter.reshape(b, -1)                                  # .view(-1): breath-major, class-minor
    logits = linear_fwd(flat, w, bias)
    if head in ('single_breath', 'lstm'):
        logits = logits.reshape(b, nb, 2)
    out = dict(logits=logits, feat=feat, stats=t.stats)
    if head == 'lstm':
        out['hx'], out['cx'] = h_t, c_t
    if target is None:
        return out
    if head in ('single_breath', 'lstm'):
        loss, dl = bce_with_logits(logits.reshape(b * nb, 2), np.repeat(target, nb, axis=0))
    else:
        loss, dl = bce_with_logits(logits, target)
    out['loss'] = loss

    def run_backward():
        t.g = {}
        t.acc('linear_final.weight', dl.T @ flat)
        t.acc('linear_final.bias', dl.sum(axis=0))
        dflat = dl @ w
        if head == 'linear':
            dfeat = dflat.reshape(feat.shape)
        elif head == 'to_mean':
            dfeat = np.repeat(dflat[:, None, :] / nb, nb, axis=1).reshape(feat.shape)
        elif head == 'compr_to_rf':
            d3 = np.zeros_like(f3)
            np.put_along_axis(d3, med_idx[:, None, :], dflat[:, None, :], axis=1)
            dfeat = d3.reshape(feat.shape)
        elif head == 'single_breath':
            dfeat = dflat
        elif head == 'lstm':
            dx3, dwi, dwh, dbb = lstm_bwd(f3, lw[0], lw[1], ltape, dflat.reshape(b, nb, -1))
            t.acc('lstm.weight_ih_l0', dwi)
            t.acc('lstm.weight_hh_l0', dwh)
            t.acc('lstm.bias_ih_l0', dbb)
            t.acc('lstm.bias_hh_l0', dbb)
            dfeat = dx3.reshape(feat.shape)
        else:
            dinter = dflat.reshape(b * nb, 2)
            t.acc('linear_intermediate.weight', dinter.T @ feat)
            t.acc('linear_intermediate.bias', dinter.sum(axis=0))
            dfeat = dinter @ wi
        fbwd(dfeat)
        return t.g

    def rebackward(flips):
        """Gradients with the listed decisions [(name, flat index)] taken the OTHER way (forward values unchanged: the
        elements are within fp32 noise of the decision boundary).  The tape's decisions are restored afterwards."""
        saved = {}
        for name, i in flips:
            d = t.decisions[name]
            key = 'mask' if d['kind'] == 'relu' else 'idx'
            saved.setdefault(name, (key, d[key]))
            _apply_flip(t, name, i)
        try:
            return dict(run_backward())
        finally:
            for name, (key, val) in saved.items():
                t.decisions[name][key] = val

    if need_grads:
        out['grads'] = dict(run_backward())
        out['tape'] = t
        out['rebackward'] = rebackward
    return out
