"""GPU parity of every block-level autograd Function (deepards_amd/functional.py) against the same unit
stated with stock torch ops on the CPU in float64 (oracle/torch_ref.py's vocabulary), per-window BN:
outputs, input gradients and every parameter gradient."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def Fn():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    import deepards_amd.functional as fn
    return fn


class _BN(object):
    """stand-in for an nn.BatchNorm1d without running stats"""
    track_running_stats = False
    momentum = 0.1
    eps = 1e-5


def bnw(x, g, b, R):
    """per-window train-mode BN on (rows, C, L) float64 CPU tensors"""
    outs = [F.batch_norm(x[i:i + R], None, None, g, b, True, 0.1, 1e-5) for i in range(0, x.shape[0], R)]
    return torch.cat(outs)


def to_rlc(t):
    return t.detach().permute(0, 2, 1).contiguous().float().cuda().requires_grad_(True)


def from_rlc(t):
    return t.detach().cpu().double().permute(0, 2, 1)


def leaf(rng, *shape, scale=1.0, shift=0.0):
    return (torch.from_numpy(rng.standard_normal(shape) * scale + shift)).requires_grad_(True)


def cmp(name, got, ref, tol=2e-5):
    got, ref = got.detach().cpu().double().numpy(), ref.detach().numpy()
    err = np.abs(got - ref).max()
    scale = 1.0 + np.abs(ref).max()
    assert err <= tol * scale, '%s: err %.3e scale %.3e' % (name, err, scale)


def dev(p):
    return p.detach().float().cuda().requires_grad_(True)


@pytest.mark.parametrize('cin,planes,stride,L,rows', [(64, 64, 1, 56, 40), (64, 128, 2, 56, 40), (128, 256, 2, 28, 40),
                                                      (256, 256, 1, 14, 40), (256, 512, 2, 14, 60), (512, 512, 1, 7, 40)])
def test_basic_block(Fn, cin, planes, stride, L, rows):
    rng = np.random.default_rng(cin + planes + L)
    R = 20
    x = leaf(rng, rows, cin, L)
    x.data.clamp_(min=0)                                   # block inputs are post-ReLU
    w1 = leaf(rng, planes, cin, 3, scale=np.sqrt(2.0 / (3 * planes)))
    w2 = leaf(rng, planes, planes, 3, scale=np.sqrt(2.0 / (3 * planes)))
    g1, b1 = leaf(rng, planes, scale=0.2, shift=1.0), leaf(rng, planes, scale=0.2)
    g2, b2 = leaf(rng, planes, scale=0.2, shift=1.0), leaf(rng, planes, scale=0.2)
    ds = stride != 1 or cin != planes
    if ds:
        wd = leaf(rng, planes, cin, 1, scale=np.sqrt(2.0 / planes))
        gd, bd = leaf(rng, planes, scale=0.2, shift=1.0), leaf(rng, planes, scale=0.2)
    o = F.relu(bnw(F.conv1d(x, w1, None, stride, 1), g1, b1, R))
    o = bnw(F.conv1d(o, w2, None, 1, 1), g2, b2, R)
    r = bnw(F.conv1d(x, wd, None, stride, 0), gd, bd, R) if ds else x
    out = F.relu(o + r)
    dout = torch.from_numpy(rng.standard_normal(tuple(out.shape)))
    out.backward(dout)

    params = [w1, g1, b1, w2, g2, b2] + ([wd, gd, bd] if ds else [])
    dp = [dev(p) for p in params]
    xd = to_rlc(x)
    st = Fn.BNState(_BN())
    args = [xd] + dp + ([None, None, None] if not ds else []) + [stride, R, st, st, st if ds else None]
    outd = Fn.BasicBlockFunction.apply(*args)
    outd.backward(dout.permute(0, 2, 1).contiguous().float().cuda())
    cmp('out', from_rlc(outd), out)
    cmp('dx', from_rlc(xd.grad), x.grad)
    for name, pd, p in zip(['w1', 'g1', 'b1', 'w2', 'g2', 'b2', 'wd', 'gd', 'bd'], dp, params):
        cmp('d' + name, pd.grad, p.grad, tol=5e-5)


@pytest.mark.parametrize('pool', ['max', 'avg'])
def test_stem(Fn, pool):
    rng = np.random.default_rng(11)
    rows, R = 40, 20
    x = torch.from_numpy(rng.standard_normal((rows, 1, 224)))
    w = leaf(rng, 64, 1, 7, scale=0.3)
    g, b = leaf(rng, 64, scale=0.2, shift=1.0), leaf(rng, 64, scale=0.2)
    z = F.relu(bnw(F.conv1d(x, w, None, 2, 3), g, b, R))
    out = F.max_pool1d(z, 3, 2, 1) if pool == 'max' else F.avg_pool1d(z, 3, 2, 1)
    dout = torch.from_numpy(rng.standard_normal(tuple(out.shape)))
    out.backward(dout)
    wd_, gd_, bd_ = dev(w), dev(g), dev(b)
    outd = Fn.StemFunction.apply(x[:, 0, :].float().cuda().contiguous(), wd_, gd_, bd_, R,
                                 Fn.POOL_MAX if pool == 'max' else Fn.POOL_AVG, Fn.BNState(_BN()))
    outd.backward(dout.permute(0, 2, 1).contiguous().float().cuda())
    cmp('out', from_rlc(outd), out)
    cmp('dw', wd_.grad, w.grad, tol=5e-5)
    cmp('dg', gd_.grad, g.grad, tol=5e-5)
    cmp('db', bd_.grad, b.grad, tol=5e-5)


@pytest.mark.parametrize('cin,L', [(64, 56), (96, 56), (96, 7), (64, 14)])
def test_dense_layer(Fn, cin, L):
    rng = np.random.default_rng(cin + L)
    rows, R = 40, 20
    x = leaf(rng, rows, cin, L)
    g1, b1 = leaf(rng, cin, scale=0.2, shift=1.0), leaf(rng, cin, scale=0.2)
    w1 = leaf(rng, 128, cin, 1, scale=np.sqrt(2.0 / 128))
    g2, b2 = leaf(rng, 128, scale=0.2, shift=1.0), leaf(rng, 128, scale=0.2)
    w2 = leaf(rng, 32, 128, 3, scale=np.sqrt(2.0 / 96))
    o = F.conv1d(F.relu(bnw(x, g1, b1, R)), w1)
    o = F.conv1d(F.relu(bnw(o, g2, b2, R)), w2, None, 1, 1)
    out = torch.cat([x, o], 1)
    dout = torch.from_numpy(rng.standard_normal(tuple(out.shape)))
    out.backward(dout)
    params = [g1, b1, w1, g2, b2, w2]
    dp = [dev(p) for p in params]
    xd = to_rlc(x)
    st = Fn.BNState(_BN())
    outd = Fn.DenseLayerFunction.apply(xd, *dp, R, st, st, 0.0, None, 1)
    outd.backward(dout.permute(0, 2, 1).contiguous().float().cuda())
    cmp('out', from_rlc(outd), out)
    cmp('dx', from_rlc(xd.grad), x.grad)
    for name, pd, p in zip(['g1', 'b1', 'w1', 'g2', 'b2', 'w2'], dp, params):
        cmp('d' + name, pd.grad, p.grad, tol=5e-5)


def test_transition_norm_pool_head(Fn):
    rng = np.random.default_rng(21)
    rows, R, cin, L = 40, 20, 128, 14
    x = leaf(rng, rows, cin, L)
    g, b = leaf(rng, cin, scale=0.2, shift=1.0), leaf(rng, cin, scale=0.2)
    w = leaf(rng, 64, cin, 1, scale=np.sqrt(2.0 / 64))
    g5, b5 = leaf(rng, 64, scale=0.2, shift=1.0), leaf(rng, 64, scale=0.2)
    wl = leaf(rng, 2, 64 * 20, scale=0.05)
    bl = leaf(rng, 2, scale=0.05)
    h = F.avg_pool1d(F.conv1d(F.relu(bnw(x, g, b, R)), w), 2, 2)          # transition -> (rows,64,7)
    h = F.relu(bnw(h, g5, b5, R))                                         # norm5 + relu
    feat = F.avg_pool1d(h, 7, 1).flatten(1)                               # (rows, 64)
    logits = F.linear(feat.reshape(2, -1), wl, bl)
    tgt = torch.tensor([[1.0, 0.0], [0.0, 1.0]], dtype=torch.float64)
    loss = torch.nn.BCEWithLogitsLoss()(logits, tgt)
    loss.backward()
    params = [g, b, w, g5, b5, wl, bl]
    dg, db, dw, dg5, db5, dwl, dbl = [dev(p) for p in params]
    xd = to_rlc(x)
    st = Fn.BNState(_BN())
    hd = Fn.TransitionFunction.apply(xd, dg, db, dw, R, st)
    hd = Fn.NormReluFunction.apply(hd, dg5, db5, R, st)
    fd = Fn.GlobalAvgPoolFunction.apply(hd)
    ld = Fn.Linear2Function.apply(fd.view(2, -1), dwl, dbl)
    lossd = Fn.bce_with_logits(ld, tgt.float().cuda())
    lossd.backward()
    cmp('logits', ld, logits)
    assert abs(float(lossd) - float(loss)) < 1e-6
    cmp('dx', from_rlc(xd.grad), x.grad)
    for name, pd, p in zip(['g', 'b', 'w', 'g5', 'b5', 'wl', 'bl'], [dg, db, dw, dg5, db5, dwl, dbl], params):
        cmp('d' + name, pd.grad, p.grad, tol=5e-5)


@pytest.mark.parametrize('backbone', ['resnet18', 'densenet18'])
def test_fused_head_chain_against_the_oracle_and_the_six_launch_chain(backbone):
    """functional.HeadLossFunction (global average pool + view(-1) + linear_final + BCEWithLogitsLoss and their backward in
    two launches) against numpy (fp64) on the map, and -- through the trainer -- against the six-launch chain it replaces:
    same losses, same parameters after three steps up to fp32 summation order."""
    import numpy as np
    import deepards_amd.functional as Fn
    import deepards_amd.models as M
    import deepards_amd.train as T
    from oracle import np_ref
    rng = np.random.RandomState(3)
    B, R, L, F = 5, 20, 7, 128
    xm = rng.randn(B * R, L, F)
    w, bias = rng.randn(2, R * F) * 0.02, rng.randn(2) * 0.1
    t = np.zeros((B, 2)); t[np.arange(B), rng.randint(0, 2, B)] = 1
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a).astype(np.float32)).cuda()
    xt, wt, bt = cu(xm).requires_grad_(True), cu(w).requires_grad_(True), cu(bias).requires_grad_(True)
    loss, logits = Fn.head_loss(xt, wt, bt, cu(t), R)
    loss.backward()
    flat = xm.mean(axis=1).reshape(B, R * F)
    lg = flat @ w.T + bias
    loss_ref, dl = np_ref.bce_with_logits(lg, t)
    assert np.abs(logits.cpu().numpy() - lg).max() < 2e-6 * (1 + np.abs(lg).max())
    assert abs(float(loss) - loss_ref) < 2e-6
    dflat = dl @ w
    dx_ref = np.repeat(dflat.reshape(B * R, 1, F), L, axis=1) / L
    assert np.abs(xt.grad.cpu().numpy() - dx_ref).max() < 2e-6 * (1 + np.abs(dx_ref).max())
    assert np.abs(wt.grad.cpu().numpy() - dl.T @ flat).max() < 2e-6 * (1 + np.abs(dl.T @ flat).max())
    assert np.abs(bt.grad.cpu().numpy() - dl.sum(axis=0)).max() < 2e-6
    # a window's share is the same whether it is trained alone or in a batch (the data-parallel step relies on it): logits
    # bit for bit, dx / dW / dbias up to the factor 1 / B (exact for the powers of two the data-parallel step divides by)
    def alone(i):
        xa = cu(xm[i * R:(i + 1) * R]).requires_grad_(True)
        wa, ba = cu(w).requires_grad_(True), cu(bias).requires_grad_(True)
        la, lga = Fn.head_loss(xa, wa, ba, cu(t[i:i + 1]), R)
        la.backward()
        return lga, xa.grad, wa.grad, ba.grad
    parts = [alone(i) for i in range(B)]
    for i, (lga, dxa, _, _) in enumerate(parts):
        assert torch.equal(lga[0], logits[i])
        assert float((dxa / B - xt.grad[i * R:(i + 1) * R]).abs().max()) <= 2e-7 * float(xt.grad.abs().max())   # (B = 5: 1 / B is not exact)
    if backbone == 'resnet18':          # (B = 5: the sum of five shares in another order -- one rounding)
        dw_sum = sum(p_[2] for p_ in parts) / B
        assert float((dw_sum - wt.grad).abs().max()) <= 2e-7 * float(wt.grad.abs().max())
    # forward only (no_grad): the loss comes from the forward kernels
    with torch.no_grad():
        l2, lg2 = Fn.head_loss(xt, wt, bt, cu(t), R)
    assert abs(float(l2) - loss_ref) < 2e-6 and torch.equal(lg2, logits)

    def run(fused):
        old = T._FUSED_HEAD
        T._FUSED_HEAD = fused
        try:
            torch.manual_seed(5)
            bb = M.resnet18() if backbone == 'resnet18' else M.densenet18(drop_rate=0.0)
            m = M.CNNLinearNetwork(bb, 20, 0).cuda()
            tr = T.HotPathTrainer(m, use_graph=True)
            g = torch.Generator().manual_seed(1)
            x = torch.randn(4, 20, 1, 224, generator=g).cuda()
            tt = torch.zeros(4, 2).cuda(); tt[:, 0] = 1
            ls = [float(tr.train_step(x, tt)) for _ in range(3)]
            ts = tr.test_step(x, tt)
            return ls, tr.bucket.p.clone(), float(ts[0]), ts[2].clone()
        finally:
            T._FUSED_HEAD = old
    la, pa, ta, preda = run(True)
    lb, pb, tb, predb = run(False)
    # (three SGD steps apart: the two chains sum the loss and dW in another order, the steps compound the last-bit differences)
    assert max(abs(a - b) for a, b in zip(la, lb)) < 1e-5, (la, lb)
    assert float((pa - pb).abs().max()) < 5e-5 and abs(ta - tb) < 1e-5 and torch.equal(preda, predb)
