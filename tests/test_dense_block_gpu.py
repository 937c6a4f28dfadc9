"""The dense block as one design (deepards_amd/functional.py DenseBlockFunction; csrc: conv1x1_bn_kernel, the wgrad operand
forms, bn_bwd_fused_kernel<EXT>, the Winograd kernel's dropout epilogue) against oracle/np_ref.py (fp64) -- reference
models/densenet.py:18-44 (_DenseLayer), :46-66 (_DenseBlock), :68-81 (_Transition).  Kernel tolerances are those of the
kernels these replace (tests/test_hip_ops_gpu.py: 2e-6 ... 2e-5 of the scale)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pytestmark = pytest.mark.gpu

from oracle import np_ref  # noqa: E402


@pytest.fixture(scope='module')
def H():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from deepards_amd import hip_ops
    return hip_ops


def rlc(a):
    return torch.from_numpy(np.ascontiguousarray(a.transpose(0, 2, 1)).astype(np.float32)).cuda()


def ncl(t):
    return t.detach().cpu().numpy().astype(np.float64).transpose(0, 2, 1)


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a).astype(np.float32)).cuda()


def close(got, ref, tol=2e-6, name=''):
    got = np.asarray(got, dtype=np.float64)
    scale = 1.0 + np.abs(ref).max()
    err = np.abs(got - ref).max()
    assert err <= tol * scale, '%s: max err %.3e (scale %.3e)' % (name, err, scale)
    return err


def pitched(a, cb, off=0):
    """numpy (N, C, L) -> a (N, L, C) channel slice at ``off`` of a fresh NaN-filled (N, L, cb) CUDA buffer: anything a
    kernel reads outside its slice poisons its result."""
    n, c, l = a.shape
    buf = torch.full((n, l, cb), float('nan'), device='cuda')
    buf[:, :, off:off + c] = rlc(a)
    return buf, buf[:, :, off:off + c]


def stat_tables(w, cb):
    t = torch.full((2, w, cb), float('nan'), device='cuda')
    return t[0], t[1]


@pytest.mark.parametrize('rows,R,L,C,N,pool', [(40, 20, 56, 96, 128, False), (40, 20, 56, 64, 128, False),
                                               (60, 20, 28, 128, 64, True), (40, 20, 14, 128, 64, True),
                                               (40, 20, 7, 96, 128, False), (25, 5, 56, 128, 128, False),
                                               (1280, 20, 14, 96, 128, False)])
def test_conv1x1_bn_forward_against_the_oracle(H, rows, R, L, C, N, pool):
    """statistics into the pitched table, then conv1x1(relu(norm(x))) with the activation applied while staging; the
    transition form is checked against the REFERENCE's order of operations (conv, then AvgPool1d(2,2): densenet.py:72-79)."""
    rng = np.random.RandomState(rows + L + C)
    cb = 160
    x = rng.randn(rows, C, L) * 1.5 + rng.randn(1, C, 1)
    w = rng.randn(N, C, 1) / np.sqrt(C)
    gamma, beta = rng.rand(C) + 0.5, rng.randn(C) * 0.3
    gamma[1] = -0.8
    buf, xv = pitched(x, cb)
    mean_t, invstd_t = stat_tables(rows // R, cb)
    H.bn_stats_fused(xv, R, mean_t[:, :C], invstd_t[:, :C])
    h_ref, st = np_ref.bn_window_fwd(x, gamma, beta, R)
    h_ref = np_ref.relu(h_ref)
    close(mean_t[:, :C].cpu().numpy(), st[0], name='mean')
    assert np.abs(invstd_t[:, :C].cpu().numpy() / st[1] - 1).max() < 2e-5
    y_ref = np_ref.conv1d_fwd(h_ref, w, 1, 0)
    if pool:
        y_ref = np_ref.avgpool_fwd(y_ref, 2, 2)
    lo = L // 2 if pool else L
    out = torch.full((rows, lo, N + 32), float('nan'), device='cuda')
    H.conv1x1_bn(xv, cu(w), R, mean_t[:, :C], invstd_t[:, :C], cu(gamma), cu(beta), out[:, :, :N], pool=pool)
    close(ncl(out[:, :, :N]), y_ref, tol=5e-6, name='conv1x1_bn')
    assert torch.isnan(out[:, :, N:]).all() and torch.isnan(buf[:, :, C:]).all()        # nothing written beside the slices
    h = H.bn_relu_ss(xv, R, mean_t[:, :C], invstd_t[:, :C], cu(gamma), cu(beta))
    close(ncl(h), h_ref, tol=5e-6, name='bn_relu_ss')


@pytest.mark.parametrize('rows,R,L,C,half,relu,with_add,drop', [(40, 20, 56, 96, False, 1, True, True),
                                                                (40, 20, 56, 64, False, 1, True, False),
                                                                (60, 20, 28, 128, True, 1, False, True),
                                                                (40, 20, 7, 128, False, 2, False, True),
                                                                (40, 20, 7, 128, False, 0, False, False),
                                                                (1280, 20, 14, 96, False, 1, True, True)])
def test_bn_bwd_ss_against_the_oracle(H, rows, R, L, C, half, relu, with_add, drop):
    """BatchNorm(+ReLU) backward on a pitched buffer with the in-place pass-through accumulation, the half-resolution
    upstream gradient of a pooled transition and the dropout mask on the last 32 channels of dx."""
    rng = np.random.RandomState(rows + L + C + 7)
    cb, G = 160, 32
    x = rng.randn(rows, C, L) + rng.randn(1, C, 1)
    gamma, beta = rng.rand(C) + 0.5, rng.randn(C) * 0.3
    z, st = np_ref.bn_window_fwd(x, gamma, beta, R)
    buf, xv = pitched(x, cb)
    mean_t, invstd_t = stat_tables(rows // R, cb)
    H.bn_stats_fused(xv, R, mean_t[:, :C], invstd_t[:, :C])
    out_fwd = None
    if relu == 2:                      # a forward through da_bn_fwd: its statistics ((W, C) contiguous), the sign of its output
        out_fwd, m_, i_ = H.bn_fwd(xv.contiguous(), R, cu(gamma), cu(beta), relu=True)
        mean_v, invstd_v = m_, i_
    else:
        mean_v, invstd_v = mean_t[:, :C], invstd_t[:, :C]
    ld = L // 2 if half else L
    dout = rng.randn(rows, C, ld)
    g_full = np.repeat(dout, 2, axis=2) * 0.5 if half else dout
    # elements on the ReLU's edge would make this a test of decisions: none may sit within fp32 noise of it
    edge = np.abs(z) < 1e-5
    g = g_full * (z > 0) if relu else g_full
    dx_ref, dgamma_ref, dbeta_ref = np_ref.bn_window_bwd(x, gamma, st, g, R)
    add = rng.randn(rows, C, L) if with_add else None
    if with_add:
        dx_ref = dx_ref + add
    dbuf = torch.full((rows, L, cb), float('nan'), device='cuda')
    dxv = dbuf[:, :, :C]
    if with_add:
        dxv.copy_(rlc(add))
    seed = torch.tensor([0x1234567], dtype=torch.int64, device='cuda')
    keep = None
    if drop:
        keep = H.dropout(torch.ones(rows, L, G, device='cuda'), seed, 5, 0.2)       # the mask of the contiguous (rows, L, G) tensor
        dx_ref[:, C - G:, :] *= ncl(keep)
    ds = H.bn_bwd_ss(rlc(dout), xv, R, mean_v, invstd_v, cu(gamma), cu(beta), relu, dxv, add=dxv if with_add else None,
                     half_dout=half, drop=(seed, 5, 0.2, G) if drop else None, out=out_fwd)
    got = ncl(dxv)
    scale = 1.0 + np.abs(dx_ref).max()
    # an edge element flips its own gradient and shifts its window's sums by one term of ~Wn: compare away from such windows
    bad_w = edge.reshape(rows // R, R, C, L).any(axis=(1, 3)) if relu else np.zeros((rows // R, C), bool)
    ok = ~np.repeat(bad_w, R, axis=0)[:, :, None] & np.ones_like(edge)
    assert np.abs((got - dx_ref) * ok).max() <= 2e-5 * scale, np.abs((got - dx_ref) * ok).max()
    assert ok.mean() > 0.95
    dg, db = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
    H.bn_param_grad_multi([(ds, dg, db)], accumulate=False)
    okc = ~bad_w.any(axis=0)
    close(dg.cpu().numpy()[okc], dgamma_ref[okc], tol=2e-5, name='dgamma')
    close(db.cpu().numpy()[okc], dbeta_ref[okc], tol=2e-5, name='dbeta')
    assert torch.isnan(dbuf[:, :, C:]).all()


@pytest.mark.parametrize('rows,R,L,C,N,half', [(40, 20, 56, 96, 128, False), (40, 20, 56, 64, 128, False),
                                               (60, 20, 28, 128, 64, True), (40, 20, 7, 128, 128, False),
                                               (640, 20, 14, 96, 128, False), (1280, 20, 56, 128, 64, True)])
def test_wgrad_with_the_recomputed_activation_against_the_oracle(H, rows, R, L, C, N, half):
    """dW of a 1x1 conv whose input relu(norm(x)) was never stored: recomputed while the weight-gradient kernel stages x;
    dy_half: the transition form (dy at half resolution, AvgPool1d(2,2) folded in front of the conv)."""
    rng = np.random.RandomState(rows + L + C + 13)
    cb = 160
    x = rng.randn(rows, C, L) + rng.randn(1, C, 1)
    gamma, beta = rng.rand(C) + 0.5, rng.randn(C) * 0.3
    h_ref, _ = np_ref.bn_window_fwd(x, gamma, beta, R)
    h_ref = np_ref.relu(h_ref)
    ld = L // 2 if half else L
    dy = rng.randn(rows, N, ld)
    if half:       # y = avgpool2(conv(h)) -> dconv = upsample(dy) / 2
        dy_full = np.repeat(dy, 2, axis=2) * 0.5
    else:
        dy_full = dy
    _, dw_ref = np_ref.conv1d_bwd(h_ref, np.zeros((N, C, 1)), dy_full, 1, 0, need_dx=False)
    _, xv = pitched(x, cb)
    mean_t, invstd_t = stat_tables(rows // R, cb)
    H.bn_stats_fused(xv, R, mean_t[:, :C], invstd_t[:, :C])
    dyb, dyv = pitched(dy, N + 32)
    extra = {'xform': (mean_t[:, :C], invstd_t[:, :C], cu(gamma), cu(beta), R)}
    if half:
        extra['dy_half'] = True
    (slab,) = H.conv_wgrad_multi([(dyv, xv, 1, 1, 0, extra)])
    dw = torch.zeros(N, C, 1, device='cuda')
    H.wgrad_reduce_multi([(slab, dw)], accumulate=False)
    close(dw.cpu().numpy(), dw_ref, tol=2e-5, name='dW')
    # and beside ordinary jobs in one call (the operand forms run in launches of their own)
    x2, dy2 = rng.randn(20, 64, 56), rng.randn(20, 64, 56)
    slabs = H.conv_wgrad_multi([(rlc(dy2), rlc(x2), 1, 1, 0), (dyv, xv, 1, 1, 0, extra)])
    dw2, dwb = torch.zeros(64, 64, 1, device='cuda'), torch.zeros(N, C, 1, device='cuda')
    H.wgrad_reduce_multi([(slabs[0], dw2), (slabs[1], dwb)], accumulate=False)
    close(dw2.cpu().numpy(), np_ref.conv1d_bwd(x2, np.zeros((64, 64, 1)), dy2, 1, 0, need_dx=False)[1], tol=2e-5)
    assert torch.equal(dwb, dw)


def test_one_launch_for_all_tile_shapes_and_the_batch_plan(H):
    """The dense-block weight gradients of a step: (a) all tile shapes in ONE launch (conv_wgrad_any_kernel) == one launch per
    tile shape, bit for bit (da_debug_set(7, 0)); (b) planned as a batch (da_conv_wgrad_multi_reduce with 8 or more such jobs:
    fewer slabs per job, the count reported back) == the oracle, and the chained reductions equal the reduction launch."""
    from deepards_amd import _lib
    rng = np.random.RandomState(3)
    R, cb = 20, 160
    jobs, refs = [], []
    for rows, L, C, N, k in [(40, 56, 96, 128, 1), (40, 56, 64, 128, 1), (40, 28, 128, 64, 1), (40, 7, 128, 128, 1), (40, 56, 128, 32, 3),
                             (60, 28, 128, 32, 3), (40, 14, 96, 128, 1), (40, 14, 128, 32, 3), (1280, 56, 96, 128, 1), (640, 14, 96, 128, 1)]:
        x = rng.randn(rows, C, L) + rng.randn(1, C, 1)
        gamma, beta = rng.rand(C) + 0.5, rng.randn(C) * 0.3
        h_ref = np_ref.relu(np_ref.bn_window_fwd(x, gamma, beta, R)[0])
        dy = rng.randn(rows, N, L)
        refs.append(np_ref.conv1d_bwd(h_ref, np.zeros((N, C, k)), dy, 1, k // 2, need_dx=False)[1])
        _, xv = pitched(x, cb)
        mean_t, invstd_t = stat_tables(rows // R, cb)
        H.bn_stats_fused(xv, R, mean_t[:, :C], invstd_t[:, :C])
        _, dyv = pitched(dy, N + 32)
        jobs.append((dyv, xv, k, 1, k // 2, {'xform': (mean_t[:, :C], invstd_t[:, :C], cu(gamma), cu(beta), R)}))
    def run(**kw):
        dws = [torch.zeros(r.shape, device='cuda') for r in refs]
        if kw:
            slabs, reduced = H.conv_wgrad_multi(jobs, dws=dws, accumulate=False)
            H.wgrad_reduce_multi([(sl, dw) for sl, dw, r in zip(slabs, dws, reduced) if not r], accumulate=False)
        else:
            slabs = H.conv_wgrad_multi(jobs)
            H.wgrad_reduce_multi(list(zip(slabs, dws)), accumulate=False)
        return dws, [sl[1] for sl in slabs]
    a, sa = run()
    _lib.lib().da_debug_set(7, 0)
    try:
        b, sb = run()
    finally:
        _lib.lib().da_debug_set(7, 1)
    assert sa == sb
    for n, (x, y, r) in enumerate(zip(a, b, refs)):
        assert torch.equal(x, y), 'job %d' % n
        close(x.cpu().numpy(), r, tol=2e-5, name='dW %d' % n)
    c, sc = run(batch=True)
    assert all(q <= p for p, q in zip(sa, sc)) and sum(sc) < sum(sa), (sa, sc)
    for n, (x, r) in enumerate(zip(c, refs)):
        close(x.cpu().numpy(), r, tol=2e-5, name='batched dW %d' % n)


@pytest.mark.parametrize('rows,L', [(40, 56), (40, 7), (300, 28), (1280, 14)])
def test_winograd_growth_conv_with_dropout_in_the_epilogue(H, rows, L):
    """The growth conv (128 -> 32, k3) writing at a channel offset of a pitched buffer with F.dropout applied in its
    epilogue == the plain kernel's output times the keep mask of H.dropout, bit for bit; and against the oracle."""
    rng = np.random.RandomState(rows + L)
    x, w = rng.randn(rows, 128, L), rng.randn(32, 128, 3) * 0.05
    u = H.wino_weights(cu(w))
    xt = rlc(x)
    plain = H.conv3_winograd(xt, u)
    close(ncl(plain), np_ref.conv1d_fwd(x, w, 1, 1), tol=5e-6, name='growth conv')
    seed = torch.tensor([77], dtype=torch.int64, device='cuda')
    buf = torch.full((rows, L, 160), float('nan'), device='cuda')
    H.conv3_winograd(xt, u, out=buf[:, :, 96:128], drop=(seed, 3, 0.2))
    assert torch.equal(buf[:, :, 96:128], H.dropout(plain, seed, 3, 0.2))
    assert torch.isnan(buf[:, :, :96]).all() and torch.isnan(buf[:, :, 128:]).all()
    frac = float((buf[:, :, 96:128] == 0).float().mean())
    assert 0.17 < frac < 0.23
    # the data gradient reads a pitched slice too
    dy = rng.randn(rows, 32, L)
    dbuf, dyv = pitched(dy, 160, 96)
    ud = H.wino_weights(cu(w), transpose=True)
    close(ncl(H.conv3_winograd(dyv, ud)), np_ref.conv1d_bwd(x, w, dy, 1, 1)[0], tol=5e-6, name='growth conv dgrad')


def _densenet_run(block_path, B=3, drop=0.0, seed=5, taps=False):
    import deepards_amd.functional as F_
    import deepards_amd.models as M
    old = F_._DENSE_BLOCK
    F_._DENSE_BLOCK = block_path
    try:
        torch.manual_seed(seed)
        m = M.CNNLinearNetwork(M.densenet18(drop_rate=drop), 20, 0).cuda()
        g = torch.Generator().manual_seed(seed)
        x = torch.randn(B, 20, 1, 224, generator=g).cuda()
        t = torch.zeros(B, 2).cuda()
        t[:, 1] = 1
        m.train()
        logits = m(x, None)
        loss = F_.bce_with_logits(logits, t)
        loss.backward()
        return logits.detach().clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
    finally:
        F_._DENSE_BLOCK = old


@pytest.mark.parametrize('drop', [0.0, 0.2])
def test_block_function_matches_the_per_layer_functions(H, drop):
    """The whole network through F_.DenseBlockFunction against the per-layer Functions it replaces (which the goldens
    pinned in rounds 1-3): same logits, same gradients up to fp32 summation order -- also with dropout ON: both draw the
    same counter-based masks."""
    la, ga = _densenet_run(True, drop=drop)
    lb, gb = _densenet_run(False, drop=drop)
    assert float((la - lb).abs().max()) < 2e-5, float((la - lb).abs().max())
    assert set(ga) == set(gb)
    worst = 0.0
    for k in ga:
        num = float((ga[k] - gb[k]).norm())
        den = max(float(gb[k].norm()), 1e-4 * np.sqrt(gb[k].numel()))
        worst = max(worst, num / den)
    # (a ReLU element whose pre-activation sits within fp32 noise of zero may go either way in the two forms: the bound is
    # what ONE such element moves a late layer's gradient by; the golden tests compare under matched decisions)
    assert worst < 3e-2, worst


@pytest.mark.parametrize('rows,R,L,drop', [(40, 20, 56, 0.2), (60, 20, 28, 0.0), (40, 20, 14, 0.2), (40, 20, 7, 0.0),
                                            (1280, 20, 14, 0.2), (1280, 20, 7, 0.2)])
def test_statistics_records_from_the_growth_conv_epilogue(H, rows, R, L, drop):
    """The growth conv hands the per-window statistics of its new channels over as records (count, mean, centred M2 per
    64-pair tile and window slot); the next 1x1 conv merges them in its prologue, publishes them to the block's table and
    normalises with them: table == numpy statistics of what the conv wrote, conv output == oracle."""
    rng = np.random.RandomState(rows + L + 3)
    cb, C0, G, N = 128, 64, 32, 128
    x0 = rng.randn(rows, C0, L) + rng.randn(1, C0, 1)
    h2 = np.abs(rng.randn(rows, 128, L))
    w2 = rng.randn(G, 128, 3) * 0.05
    buf, x0v = pitched(x0, cb)
    mean_t, invstd_t = stat_tables(rows // R, cb)
    H.bn_stats_fused(x0v, R, mean_t[:, :C0], invstd_t[:, :C0])
    seed = torch.tensor([4242], dtype=torch.int64, device='cuda')
    new, rec = H.conv3_winograd(rlc(h2), H.wino_weights(cu(w2)), out=buf[:, :, C0:C0 + G],
                                drop=(seed, 2, drop) if drop else None, stats_R=R)
    xall = ncl(buf[:, :, :C0 + G])                                  # what the kernels wrote, as float64
    assert np.isfinite(xall).all()
    C = C0 + G
    w1 = rng.randn(N, C, 1) / np.sqrt(C)
    gamma, beta = rng.rand(C) + 0.5, rng.randn(C) * 0.3
    y = torch.empty(rows, L, N, device='cuda')
    pl = (L + 1) // 2
    H.conv1x1_bn(buf[:, :, :C], cu(w1), R, mean_t[:, :C], invstd_t[:, :C], cu(gamma), cu(beta), y,
                 pend=(rec, C0, rows * pl, R * pl))
    h_ref, st = np_ref.bn_window_fwd(xall, gamma, beta, R)
    close(mean_t[:, C0:C].cpu().numpy(), st[0][:, C0:], tol=2e-6, name='published mean')
    assert np.abs(invstd_t[:, C0:C].cpu().numpy() / st[1][:, C0:] - 1).max() < 2e-5
    close(ncl(y), np_ref.conv1d_fwd(np_ref.relu(h_ref), w1, 1, 0), tol=5e-6, name='conv1x1 on merged statistics')
    assert torch.isnan(mean_t[:, C:]).all()


@pytest.mark.parametrize('rows,R,L', [(40, 20, 56), (60, 20, 28), (1280, 20, 14)])
def test_statistics_records_from_the_transition_conv_epilogue(H, rows, R, L):
    """The pooled transition conv writes the next block's first channels AND their statistics records; the next block's
    first 1x1 conv (every input channel pending) merges and publishes them."""
    rng = np.random.RandomState(rows + L + 5)
    C, N, cb2 = 128, 64, 128
    x = rng.randn(rows, C, L) + rng.randn(1, C, 1)
    wt = rng.randn(N, C, 1) / np.sqrt(C)
    gamma, beta = rng.rand(C) + 0.5, rng.randn(C) * 0.3
    _, xv = pitched(x, C)
    mean_t, invstd_t = stat_tables(rows // R, C)
    H.bn_stats_fused(xv, R, mean_t, invstd_t)
    nxt = torch.full((rows, L // 2, cb2), float('nan'), device='cuda')
    _, rec = H.conv1x1_bn(xv, cu(wt), R, mean_t, invstd_t, cu(gamma), cu(beta), nxt[:, :, :N], pool=True, want_records=True)
    x2 = ncl(nxt[:, :, :N])
    h_ref, _ = np_ref.bn_window_fwd(x, gamma, beta, R)
    close(x2, np_ref.avgpool_fwd(np_ref.conv1d_fwd(np_ref.relu(h_ref), wt, 1, 0), 2, 2), tol=5e-6, name='transition')
    m2_t, i2_t = stat_tables(rows // R, cb2)
    w1 = rng.randn(128, N, 1) / np.sqrt(N)
    g2, b2 = rng.rand(N) + 0.5, rng.randn(N) * 0.3
    y = torch.empty(rows, L // 2, 128, device='cuda')
    H.conv1x1_bn(nxt[:, :, :N], cu(w1), R, m2_t[:, :N], i2_t[:, :N], cu(g2), cu(b2), y, pend=(rec, 0, rows * (L // 2), R * (L // 2)))
    h2_ref, st2 = np_ref.bn_window_fwd(x2, g2, b2, R)
    close(m2_t[:, :N].cpu().numpy(), st2[0], tol=2e-6, name='published mean')
    assert np.abs(i2_t[:, :N].cpu().numpy() / st2[1] - 1).max() < 2e-5
    close(ncl(y), np_ref.conv1d_fwd(np_ref.relu(h2_ref), w1, 1, 0), tol=5e-6, name='first conv of the next block')


@pytest.mark.parametrize('rows,R,L,drop', [(40, 20, 56, 0.2), (60, 20, 28, 0.0), (40, 20, 14, 0.2), (40, 20, 7, 0.0),
                                            (1280, 20, 28, 0.2), (1280, 20, 7, 0.0)])
def test_growth_conv_with_norm2_applied_while_staging(H, rows, R, L, drop):
    """norm2 -> relu2 -> conv2 (+dropout) as one kernel: the 1x1 conv hands the statistics of its output over as records,
    the Winograd growth conv merges them, normalises while it stages and publishes mean / invstd; the weight gradient of
    conv2 recomputes relu(norm2(y1)) the same way (3-tap operand form), the BatchNorm backward takes its decisions."""
    rng = np.random.RandomState(rows + L + 11)
    C0, mid, G, cb = 64, 128, 32, 128
    x0 = rng.randn(rows, C0, L) + rng.randn(1, C0, 1)
    w1 = rng.randn(mid, C0, 1) / np.sqrt(C0)
    w2 = rng.randn(G, mid, 3) * 0.08
    g1, b1 = rng.rand(C0) + 0.5, rng.randn(C0) * 0.3
    g2, b2 = rng.rand(mid) + 0.5, rng.randn(mid) * 0.3
    buf, x0v = pitched(x0, cb)
    mean_t, invstd_t = stat_tables(rows // R, cb)
    H.bn_stats_fused(x0v, R, mean_t[:, :C0], invstd_t[:, :C0])
    y1 = torch.empty(rows, L, mid, device='cuda')
    _, rec1 = H.conv1x1_bn(x0v, cu(w1), R, mean_t[:, :C0], invstd_t[:, :C0], cu(g1), cu(b1), y1, want_records=True)
    y1n = ncl(y1)
    m2 = torch.full((rows // R, mid), float('nan'), device='cuda')
    i2 = torch.full_like(m2, float('nan'))
    seed = torch.tensor([99], dtype=torch.int64, device='cuda')
    new = buf[:, :, C0:C0 + G]
    _, rec2 = H.conv3_winograd_bn(y1, H.wino_weights(cu(w2)), R, rec1, m2, i2, cu(g2), cu(b2), new, drop=(seed, 4, drop) if drop else None,
                                  want_records=True)
    z2, st2 = np_ref.bn_window_fwd(y1n, g2, b2, R)
    close(m2.cpu().numpy(), st2[0], tol=2e-6, name='published mean of y1')
    assert np.abs(i2.cpu().numpy() / st2[1] - 1).max() < 2e-5
    h2 = np_ref.relu(z2)
    ref = np_ref.conv1d_fwd(h2, w2, 1, 1)
    if drop:
        ref = ref * ncl(H.dropout(torch.ones(rows, L, G, device='cuda'), seed, 4, drop))
    close(ncl(new), ref, tol=1e-5, name='norm2 + growth conv')
    # the records of the new channels feed the next 1x1 conv as before
    C = C0 + G
    w3 = rng.randn(128, C, 1) / np.sqrt(C)
    g3, b3 = rng.rand(C) + 0.5, rng.randn(C) * 0.3
    y3 = torch.empty(rows, L, 128, device='cuda')
    pl = (L + 1) // 2
    H.conv1x1_bn(buf[:, :, :C], cu(w3), R, mean_t[:, :C], invstd_t[:, :C], cu(g3), cu(b3), y3, pend=(rec2, C0, rows * pl, R * pl))
    h3, st3 = np_ref.bn_window_fwd(ncl(buf[:, :, :C]), g3, b3, R)
    close(mean_t[:, C0:C].cpu().numpy(), st3[0][:, C0:], tol=2e-6, name='published mean of the new channels')
    close(ncl(y3), np_ref.conv1d_fwd(np_ref.relu(h3), w3, 1, 0), tol=5e-6, name='next 1x1 conv')
    # backward pieces: dW of conv2 from (dnew, y1) with relu(norm2(y1)) recomputed; norm2 backward in the same decision form
    dnew = rng.randn(rows, G, L)
    _, dw_ref = np_ref.conv1d_bwd(h2, w2, dnew, 1, 1, need_dx=False)
    _, dnv = pitched(dnew, cb, C0)
    (slab,) = H.conv_wgrad_multi([(dnv, y1, 3, 1, 1, {'xform': (m2, i2, cu(g2), cu(b2), R)})])
    dw = torch.zeros(G, mid, 3, device='cuda')
    H.wgrad_reduce_multi([(slab, dw)], accumulate=False)
    close(dw.cpu().numpy(), dw_ref, tol=2e-5, name='dW of the growth conv')
