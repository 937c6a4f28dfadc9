"""GPU parity: every C-ABI kernel against the numpy oracle (oracle/np_ref.py, float64) on seeded inputs.

Tolerances are for fp32 kernels against an fp64 oracle: the error of a length-K fp32 dot product is
~1e-7 * sum|a*b|, so results are compared with atol scaled by the magnitude of the data."""
import numpy as np
import pytest
import torch

from oracle import np_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def H():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from deepards_amd import hip_ops
    return hip_ops


def rlc(a):
    """numpy (N,C,L) -> cuda (N,L,C) float32"""
    return torch.from_numpy(np.ascontiguousarray(a.transpose(0, 2, 1)).astype(np.float32)).cuda()


def ncl(t):
    return t.detach().cpu().numpy().astype(np.float64).transpose(0, 2, 1)


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a).astype(np.float32)).cuda()


def close(got, ref, tol=2e-6, name=''):
    """max|got-ref| <= tol * (1 + max|ref|) * sqrt-ish slack"""
    got = np.asarray(got, dtype=np.float64)
    scale = 1.0 + np.abs(ref).max()
    err = np.abs(got - ref).max()
    assert err <= tol * scale, '%s: max err %.3e (scale %.3e)' % (name, err, scale)
    return err


CONV_CASES = [
    # ci, co, k, stride, pad, L, rows
    (64, 64, 3, 1, 1, 56, 40),
    (64, 128, 3, 2, 1, 56, 40),
    (64, 128, 1, 2, 0, 56, 40),
    (128, 128, 3, 1, 1, 28, 23),
    (128, 256, 3, 2, 1, 28, 40),
    (256, 512, 3, 2, 1, 14, 60),
    (256, 512, 1, 2, 0, 14, 60),
    (512, 512, 3, 1, 1, 7, 40),
    (96, 128, 1, 1, 0, 56, 20),
    (128, 32, 3, 1, 1, 28, 20),
    (128, 64, 1, 1, 0, 14, 20),
    (64, 64, 3, 1, 1, 56, 1),
    (128, 32, 3, 1, 1, 9, 5),
    (256, 256, 3, 1, 1, 14, 40),
    (96, 128, 1, 1, 0, 7, 40),
    (128, 32, 3, 1, 1, 7, 40),
    (128, 128, 3, 1, 1, 28, 40),
    (128, 64, 1, 1, 0, 56, 40),
    # more than 256 64x64 tiles with a partly filled last round: the tail runs as split-K half tiles
    (64, 64, 3, 1, 1, 56, 300),
    (128, 128, 3, 1, 1, 28, 300),
    (64, 128, 3, 2, 1, 56, 300),
    (128, 64, 1, 1, 0, 14, 1200),
    (256, 256, 3, 1, 1, 14, 150),
]


def test_conv_wgrad_multi_matches_single_jobs(H):
    """Every weight gradient of a step in one launch per tile shape (da_conv_wgrad_multi: more jobs than one 24-entry
    table holds) == the oracle; direct jobs are bit-identical to the one-job launches (same plan, same slabs), the
    k3 s1 p1 jobs with 64-multiple channels run in Winograd F(2,3) form (even and odd L, 1 row, > 256 tiles)."""
    rng = np.random.default_rng(77)
    jobs, refs, targets, singles = [], [], [], []
    for n, (ci, co, k, stride, pad, L, rows) in enumerate(CONV_CASES + CONV_CASES[:12]):
        x = rng.standard_normal((rows, ci, L))
        lo = (L + 2 * pad - k) // stride + 1
        dy = rng.standard_normal((rows, co, lo))
        w = np.zeros((co, ci, k))
        refs.append(np_ref.conv1d_bwd(x, w, dy, stride, pad)[1])
        xt, dyt = rlc(x), rlc(dy)
        jobs.append((dyt, xt, k, stride, pad))
        targets.append(torch.zeros(co, ci, k, device='cuda'))
        singles.append(torch.zeros(co, ci, k, device='cuda'))
        H.wgrad_reduce_multi([(H.conv_wgrad(dyt, xt, k, stride, pad, defer=True), singles[-1])], accumulate=True)
    slabs = H.conv_wgrad_multi(jobs)
    H.wgrad_reduce_multi(list(zip(slabs, targets)), accumulate=True)
    nw = 0
    for n, (t, r, s1, (dyt, xt, k, stride, pad)) in enumerate(zip(targets, refs, singles, jobs)):
        close(t.cpu().numpy(), r, tol=3e-6, name='job %d' % n)
        wino = H.WINOGRAD_WGRAD and k == 3 and stride == 1 and pad == 1 and dyt.shape[2] % 64 == 0 and xt.shape[2] % 64 == 0
        nw += bool(wino)
        if not wino:        # same plan, same slabs as the one-job launch; the Winograd form is a different summation
            assert torch.equal(t, s1), 'job %d differs from the one-job launch' % n
    assert nw >= 8 or not H.WINOGRAD_WGRAD
    assert H.conv_wgrad_multi([]) == []


@pytest.mark.parametrize('accumulate', [True, False])
def test_chained_slab_reductions_equal_the_reduction_launch(H, accumulate):
    """da_conv_wgrad_multi_reduce: every launch of the call folds, as its first blocks, the slabs the launch before it wrote;
    the gradients are bit for bit those of da_conv_wgrad_multi + da_wgrad_reduce_multi, the jobs of the last launch (and jobs
    without a destination) are handed back, more than WGRAD_PRE_MAX jobs of one launch too."""
    rng = np.random.default_rng(5)
    cases = [(512, 512, 3, 1, 1, 7, 40)] * 2 + [(64, 64, 3, 1, 1, 56, 20)] * 17 + [(128, 128, 3, 1, 1, 28, 20),
             (64, 128, 3, 2, 1, 56, 20), (64, 128, 1, 2, 0, 56, 20), (128, 32, 3, 1, 1, 7, 40), (96, 128, 1, 1, 0, 7, 40)]
    jobs, a, b = [], [], []
    for ci, co, k, stride, pad, L, rows in cases:
        lo = (L + 2 * pad - k) // stride + 1
        jobs.append((rlc(rng.standard_normal((rows, co, lo))), rlc(rng.standard_normal((rows, ci, L))), k, stride, pad))
        init = torch.from_numpy(rng.standard_normal((co, ci, k)).astype(np.float32)).cuda()
        a.append(init.clone())
        b.append(init.clone())
    a[3] = None                                     # a job without a destination: slabs only
    slabs, reduced = H.conv_wgrad_multi(jobs, dws=a, accumulate=accumulate)
    assert not reduced[3] and not all(reduced) and sum(reduced) >= 17, reduced
    assert not reduced[-1] or not reduced[-2]       # (the last launch's jobs are the caller's)
    a[3] = b[3].clone()
    H.wgrad_reduce_multi([(sl, dw) for sl, dw, r in zip(slabs, a, reduced) if not r], accumulate=accumulate)
    H.wgrad_reduce_multi(list(zip(H.conv_wgrad_multi(jobs), b)), accumulate=accumulate)
    for n, (x, y) in enumerate(zip(a, b)):
        assert torch.equal(x, y), 'job %d' % n
    assert H.conv_wgrad_multi([], dws=[]) == ([], [])


def test_last_round_of_the_winograd_weight_gradients_runs_half_blocks(H):
    """The F(2,3) weight gradients of the resnet18 step at B = 64 (13 jobs, 1 232 blocks on 1 024 slots) through the chained
    call: the jobs of the partly filled last round (the four 64-channel ones) run with half the pairs per split -- twice the
    slabs, reported back -- and agree with the plain call's sums to rounding; the others are bit-identical."""
    g = torch.Generator().manual_seed(9)
    mk = lambda *sh: torch.randn(*sh, generator=g).cuda()
    shapes = [(256, 14)] * 3 + [(128, 28)] * 3 + [(64, 56)] * 4
    jobs = [(mk(1280, l, c), mk(1280, l, c), 3, 1, 1) for c, l in shapes]
    a = [torch.zeros(c, c, 3, device='cuda') for c, _ in shapes]
    b = [torch.zeros(c, c, 3, device='cuda') for c, _ in shapes]
    slabs, reduced = H.conv_wgrad_multi(jobs, dws=a, accumulate=False)
    H.wgrad_reduce_multi([(sl, dw) for sl, dw, r in zip(slabs, a, reduced) if not r], accumulate=False)
    plain = H.conv_wgrad_multi(jobs)
    H.wgrad_reduce_multi(list(zip(plain, b)), accumulate=False)
    for n, ((c, l), x, y, s1, s0) in enumerate(zip(shapes, a, b, slabs, plain)):
        if c == 64:
            assert s1[1] == 2 * s0[1], (n, s1[1], s0[1])
            assert float((x - y).abs().max()) <= 2e-6 * float(y.abs().max()), n
        else:
            assert s1[1] == s0[1] and torch.equal(x, y), n


@pytest.mark.parametrize('n_red,n_bn,with_stem', [(3, 4, True), (3, 0, True), (33, 4, True), (3, 25, False), (0, 3, True), (2, 2, False)])
def test_step_tail_launch_equals_its_parts(H, n_red, n_bn, with_stem):
    """da_step_tail_multi (slab reductions + BatchNorm dgamma / dbeta folds + running-statistics updates + the stem's
    weight-gradient fold in ONE launch) == the separate entry points, bit for bit -- also where the library falls back to
    separate launches (more than 32 reductions / 24 BatchNorms, no reductions at all)."""
    g = torch.Generator().manual_seed(n_red * 100 + n_bn)
    mk = lambda *sh: torch.randn(*sh, generator=g).cuda()
    def build():
        red = []
        for i in range(n_red):
            co, ci, k, splits = (64, 32, 3, 5) if i % 2 else (32, 64, 1, 9)
            red.append(((mk(splits * k * co * ci), splits, k, co, ci), mk(co, ci, k)))
        pg, run = [], []
        for i in range(n_bn):
            c, w = (64, 7) if i % 2 else (96, 64)
            pg.append((mk(2, w, c), mk(c), mk(c)))
            run.append((mk(w, c), torch.rand(w, c, generator=g).cuda() + 0.5, 140, mk(c), torch.rand(c, generator=g).cuda() + 0.5,
                        torch.zeros((), dtype=torch.int64, device='cuda'), 0.1, 1e-5))
        stem = (mk(40, 448), 40, 448, mk(64, 1, 7)) if with_stem else None
        return red, pg, run, stem
    g.manual_seed(5)
    red_a, pg_a, run_a, stem_a = build()
    g.manual_seed(5)
    red_b, pg_b, run_b, stem_b = build()
    H.step_tail_multi(red_a, pg_a, run_a, accumulate=True, stem=stem_a)
    H.wgrad_reduce_multi(red_b, accumulate=True)
    H.bn_param_grad_multi(pg_b, accumulate=True)
    H.bn_running_multi(run_b)
    if stem_b is not None:
        from deepards_amd import _lib
        H._chk(_lib.lib().da_stem_wgrad_reduce(H._p(stem_b[0]), stem_b[1], stem_b[2], H._p(stem_b[3]), 1, H._stream()), 'reduce')
    for (_, da), (_, db) in zip(red_a, red_b):
        assert torch.equal(da, db)
    for (_, ga, ba), (_, gb, bb) in zip(pg_a, pg_b):
        assert torch.equal(ga, gb) and torch.equal(ba, bb)
    for ra, rb in zip(run_a, run_b):
        assert torch.equal(ra[3], rb[3]) and torch.equal(ra[4], rb[4]) and int(ra[5]) == int(rb[5]) > 0
    if stem_a is not None:
        assert torch.equal(stem_a[3], stem_b[3])


@pytest.mark.parametrize('C,L,R,W', [(128, 28, 20, 64), (256, 14, 20, 3), (512, 7, 20, 5), (64, 56, 20, 2)])
def test_batchnorm_pair_launches_equal_the_single_ones(H, C, L, R, W):
    """A stride-2 block entry's two independent BatchNorms in one launch (da_bn_fwd_pair: bn1 + ReLU | the downsample's;
    da_bn_bwd_pair: bn2 | the downsample's from the same masked gradient) == the single launches, bit for bit (the same
    body on the same geometry; oracle parity of those: test_bn_*)."""
    rows = R * W
    g = torch.Generator().manual_seed(C + L + W)
    mk = lambda *sh: torch.randn(*sh, generator=g).cuda()
    y1, yd, y2, res_in = mk(rows, L, C), mk(rows, L, C), mk(rows, L, C), mk(rows, L, C)
    g1, b1, gd, bd, g2, b2 = (torch.rand(C, generator=g).cuda() + 0.5 for _ in range(6))
    assert H.bn_single_pass(W, R * L, C)
    (res, md, idd, _), (h1, m1, i1, _) = H.bn_fwd_pair([(yd, gd, bd, False, None, False), (y1, g1, b1, True, None, False)], R)
    res_s, md_s, id_s = H.bn_fwd(yd, R, gd, bd, relu=False)
    h1_s, m1_s, i1_s = H.bn_fwd(y1, R, g1, b1, relu=True)
    for a, b in ((res, res_s), (md, md_s), (idd, id_s), (h1, h1_s), (m1, m1_s), (i1, i1_s)):
        assert torch.equal(a, b)
    out, m2, i2, mask = H.bn_fwd(y2, R, g2, b2, relu=True, res=res, want_mask=True)
    dout = mk(rows, L, C)
    (dy2, ds2), (dyd, dsd) = H.bn_bwd_pair(dout, [(y2, m2, i2, g2, b2, None), (yd, md, idd, gd, bd, None)], R, mask)
    dy2_s, _, _, g_s, ds2_s = H.bn_bwd(dout, y2, R, m2, i2, g2, b2, 2, out=out, want_g=True, mask=mask, defer_param_grads=True)
    dyd_s, _, _, _, dsd_s = H.bn_bwd(g_s, yd, R, md, idd, gd, bd, 0, defer_param_grads=True)
    for a, b in ((dy2, dy2_s), (ds2, ds2_s), (dyd, dyd_s), (dsd, dsd_s)):
        assert torch.equal(a, b)


@pytest.mark.parametrize('ci,co,L,rows', [(512, 512, 7, 40), (512, 512, 14, 23), (512, 1024, 7, 33), (512, 512, 5, 9),
                                          (512, 512, 1, 70), (512, 512, 4, 64), (512, 512, 8, 1), (1024, 512, 9, 300)])
def test_conv3_wgrad_winograd4(H, ci, co, L, rows):
    """The F(4,3) weight-gradient form (k3 s1 p1, both channel counts >= hip_ops.WINO4_WGRAD_MIN_C; K over output quads:
    lengths that are / are not multiples of 4, one quad per row, more quads than one split) vs the oracle; the transform
    constants reach 8 and 1/24, so the bound is the F(4,3) forward's (4e-6 of the scale; F(2,3) form: 3e-6) -- measured in
    the log -- and the result must agree with the F(2,3) form of the same job to the sum of the two."""
    if not H.WINOGRAD_WGRAD:
        pytest.skip('direct kernels selected')
    rng = np.random.default_rng(ci + co + L + rows)
    x = rng.standard_normal((rows, ci, L))
    dy = rng.standard_normal((rows, co, L))
    dw_ref = np_ref.conv1d_bwd(x, np.zeros((co, ci, 3)), dy, 1, 1)[1]
    xt, dyt = rlc(x), rlc(dy)

    def run():
        dw = torch.zeros(co, ci, 3, device='cuda')
        H.wgrad_reduce_multi(list(zip(H.conv_wgrad_multi([(dyt, xt, 3, 1, 1)]), [dw])), accumulate=True)
        return dw
    dw4 = run()
    old = H.WINO4_WGRAD_MIN_C
    H.WINO4_WGRAD_MIN_C = 1 << 30
    try:
        dw2 = run()
    finally:
        H.WINO4_WGRAD_MIN_C = old
    e4 = close(dw4.cpu().numpy(), dw_ref, tol=4e-6, name='F(4,3) wgrad')
    e2 = close(dw2.cpu().numpy(), dw_ref, tol=3e-6, name='F(2,3) wgrad')
    print('wgrad errors F(4,3) %.2e  F(2,3) %.2e  scale %.2e' % (e4, e2, 1 + np.abs(dw_ref).max()))
    assert not torch.equal(dw4, dw2)                   # (the two forms did run)
    dw4b = run()
    assert torch.equal(dw4, dw4b)                      # deterministic


@pytest.mark.parametrize('ci,co,k,stride,pad,L,rows', CONV_CASES)
def test_conv_fwd_dgrad_wgrad(H, ci, co, k, stride, pad, L, rows):
    rng = np.random.default_rng(ci * 1000 + co + k + L)
    x = rng.standard_normal((rows, ci, L))
    w = rng.standard_normal((co, ci, k)) * np.sqrt(2.0 / (k * co))
    y_ref = np_ref.conv1d_fwd(x, w, stride, pad)
    dy = rng.standard_normal(y_ref.shape)
    dx_ref, dw_ref = np_ref.conv1d_bwd(x, w, dy, stride, pad)

    xt, wt, dyt = rlc(x), cu(w), rlc(dy)
    wf, wd = H.repack_weight(wt, True, True)
    assert torch.equal(wf, wt.permute(2, 0, 1).contiguous())
    assert torch.equal(wd, wt.permute(2, 1, 0).contiguous())
    y = H.conv_fwd(xt, wf, stride, pad)
    close(ncl(y), y_ref, name='fwd')
    dx = H.conv_dgrad(dyt, wd, stride, pad, L)
    close(ncl(dx), dx_ref, name='dgrad')
    dw = H.conv_wgrad(dyt, xt, k, stride, pad)
    close(dw.cpu().numpy(), dw_ref, tol=3e-6, name='wgrad')
    # accumulate forms
    base = rng.standard_normal(dx_ref.shape)
    bt = rlc(base)
    H.conv_dgrad(dyt, wd, stride, pad, L, out=bt, accumulate=True)
    close(ncl(bt), base + dx_ref, name='dgrad+acc')
    dw2 = dw.clone()
    H.conv_wgrad(dyt, xt, k, stride, pad, out=dw2, accumulate=True)
    close(dw2.cpu().numpy(), 2 * dw_ref, tol=3e-6, name='wgrad+acc')
    # deferred slab reduction (batched) and batched repack
    dw3 = dw.clone()
    H.wgrad_reduce_multi([(H.conv_wgrad(dyt, xt, k, stride, pad, defer=True), dw3)], accumulate=True)
    close(dw3.cpu().numpy(), 2 * dw_ref, tol=3e-6, name='wgrad deferred')
    (wf2, wd2, uf2, ud2), = H.repack_multi([wt])
    assert torch.equal(wf2, wf) and torch.equal(wd2, wd) and uf2 is None and ud2 is None


@pytest.mark.parametrize('ci,co,L,rows', [(64, 64, 56, 40), (128, 128, 28, 23), (256, 256, 14, 40), (512, 512, 7, 40),
                                          (128, 32, 56, 20), (64, 64, 56, 1), (128, 32, 9, 5), (32, 32, 1, 7),
                                          (64, 64, 56, 300), (32, 64, 2, 33), (128, 128, 28, 300), (64, 64, 57, 300),
                                          (64, 128, 7, 1040)])
def test_conv3_winograd(H, ci, co, L, rows):
    """Winograd F(2,3) k3 s1 p1 conv: forward, data gradient (transposed taps) and accumulate form vs the oracle
    (the 300 / 150 / 1290-row cases have more than 256 tiles: their last round runs as split-K half tiles)."""
    rng = np.random.default_rng(ci + co + L + rows)
    x = rng.standard_normal((rows, ci, L))
    w = rng.standard_normal((co, ci, 3)) * np.sqrt(2.0 / (3 * co))
    y_ref = np_ref.conv1d_fwd(x, w, 1, 1)
    dy = rng.standard_normal(y_ref.shape)
    dx_ref, _ = np_ref.conv1d_bwd(x, w, dy, 1, 1)
    xt, wt, dyt = rlc(x), cu(w), rlc(dy)
    uf, ud = H.wino_weights(wt), H.wino_weights(wt, transpose=True)
    g = w.astype(np.float32).astype(np.float64)
    u_ref = np.stack([g[:, :, 0], (g[:, :, 0] + g[:, :, 1] + g[:, :, 2]) / 2, (g[:, :, 0] - g[:, :, 1] + g[:, :, 2]) / 2, g[:, :, 2]])
    close(uf.cpu().numpy(), u_ref, tol=1e-6, name='taps fwd')
    close(ud.cpu().numpy(), np.stack([g[:, :, 2].T, u_ref[1].T, u_ref[2].T, g[:, :, 0].T]), tol=1e-6, name='taps dgrad')
    (wf2, wd2, uf2, ud2), = H.repack_multi([wt], [True])          # the batched repack emits the same taps
    assert wf2 is None and wd2 is None and torch.equal(uf2, uf) and torch.equal(ud2, ud)
    close(ncl(H.conv3_winograd(xt, uf)), y_ref, tol=4e-6, name='winograd fwd')
    close(ncl(H.conv3_winograd(dyt, ud)), dx_ref, tol=4e-6, name='winograd dgrad')
    base = rng.standard_normal(dx_ref.shape)
    bt = rlc(base)
    H.conv3_winograd(dyt, ud, out=bt, accumulate=True)
    close(ncl(bt), base + dx_ref, tol=4e-6, name='winograd dgrad+acc')


@pytest.mark.parametrize('ksteps', [16, 32])
@pytest.mark.parametrize('ci,co,L,rows', [(512, 512, 7, 40), (256, 256, 14, 40), (64, 64, 56, 20), (128, 32, 9, 5),
                                          (32, 32, 1, 7), (32, 64, 2, 33), (32, 32, 3, 4), (64, 32, 5, 9),
                                          (512, 512, 7, 300), (64, 64, 57, 300), (96, 64, 6, 1040)])
def test_conv3_winograd_f43(H, ci, co, L, rows, ksteps):
    """Winograd F(4,3) (six contractions per output quad): taps (stand-alone kernel and batched repack), forward, data
    gradient and accumulate form vs the oracle, lengths that leave 1, 2 or 3 outputs in the last quad, launches whose
    last round runs as half tiles; both K-step variants of the kernel."""
    from deepards_amd import _lib
    _lib.lib().da_wino_debug_tail(3 if ksteps == 16 else 2)
    try:
        rng = np.random.default_rng(ci + co + L + rows)
        x = rng.standard_normal((rows, ci, L))
        w = rng.standard_normal((co, ci, 3)) * np.sqrt(2.0 / (3 * co))
        y_ref = np_ref.conv1d_fwd(x, w, 1, 1)
        dy = rng.standard_normal(y_ref.shape)
        dx_ref, _ = np_ref.conv1d_bwd(x, w, dy, 1, 1)
        xt, wt, dyt = rlc(x), cu(w), rlc(dy)
        uf, ud = H.wino_weights(wt, points=6), H.wino_weights(wt, transpose=True, points=6)
        g0, g1, g2 = (w.astype(np.float32).astype(np.float64)[:, :, t] for t in range(3))
        taps = lambda a, b, c: np.stack([a / 4, -(a + b + c) / 6, -(a - b + c) / 6, a / 24 + b / 12 + c / 6,
                                         a / 24 - b / 12 + c / 6, c])
        close(uf.cpu().numpy(), taps(g0, g1, g2), tol=1e-6, name='F(4,3) taps fwd')
        close(ud.cpu().numpy(), taps(g2.T, g1.T, g0.T), tol=1e-6, name='F(4,3) taps dgrad')
        (wf2, wd2, uf2, ud2), = H.repack_multi([wt], [6])
        assert wf2 is None and wd2 is None and torch.equal(uf2, uf) and torch.equal(ud2, ud)
        close(ncl(H.conv3_winograd(xt, uf)), y_ref, tol=2e-5, name='F(4,3) fwd')
        close(ncl(H.conv3_winograd(dyt, ud)), dx_ref, tol=2e-5, name='F(4,3) dgrad')
        base = rng.standard_normal(dx_ref.shape)
        bt = rlc(base)
        H.conv3_winograd(dyt, ud, out=bt, accumulate=True)
        close(ncl(bt), base + dx_ref, tol=2e-5, name='F(4,3) dgrad+acc')
    finally:
        _lib.lib().da_wino_debug_tail(3)


@pytest.mark.parametrize('ci,co,L,rows', [(64, 64, 56, 40), (128, 128, 28, 23), (512, 512, 7, 40), (32, 64, 1, 7),
                                          (96, 128, 2, 33), (64, 192, 3, 5), (64, 64, 57, 9), (256, 64, 14, 1),
                                          (64, 64, 56, 1280)])
def test_conv3_bf16(H, ci, co, L, rows):
    """bf16-MFMA k3 s1 p1 conv (BASELINE config C3 arithmetic): tap packs == torch's round-to-nearest-even bfloat16;
    forward, data gradient and accumulate form == the fp64 convolution of the bf16-ROUNDED operands to fp32 rounding
    (the kernel's only approximation is the operand rounding), and within bf16's 2^-8 of the unrounded one."""
    rng = np.random.default_rng(ci + co + L + rows)
    x = rng.standard_normal((rows, ci, L))
    w = rng.standard_normal((co, ci, 3)) * np.sqrt(2.0 / (3 * co))
    dy = rng.standard_normal((rows, co, L))
    xt, wt, dyt = rlc(x), cu(w), rlc(dy)
    wf, wd = H.pack_conv3_bf16(wt)
    assert torch.equal(wf, wt.permute(2, 0, 1).contiguous().bfloat16())
    assert torch.equal(wd, wt.flip(2).permute(2, 1, 0).contiguous().bfloat16())
    rb = np_ref.round_bf16
    assert np.array_equal(rb(w), torch.from_numpy(w.astype(np.float32)).bfloat16().double().numpy())
    y_b = np_ref.conv1d_fwd(rb(x), rb(w), 1, 1)
    dx_b, _ = np_ref.conv1d_bwd(rb(x), rb(w), rb(dy), 1, 1)
    close(ncl(H.conv3_bf16(xt, wf)), y_b, tol=3e-6, name='bf16 fwd vs rounded operands')
    if ci % 64 == 0:                                 # the data gradient's output channels are the conv's inputs
        close(ncl(H.conv3_bf16(dyt, wd)), dx_b, tol=3e-6, name='bf16 dgrad vs rounded operands')
        base = rng.standard_normal(dx_b.shape)
        bt = rlc(base)
        H.conv3_bf16(dyt, wd, out=bt, accumulate=True)
        close(ncl(bt), base.astype(np.float32).astype(np.float64) + dx_b, tol=3e-6, name='bf16 dgrad+acc')
    else:
        with pytest.raises(ValueError):
            H.conv3_bf16(dyt, wd)
    y_ref = np_ref.conv1d_fwd(x, w, 1, 1)
    close(ncl(H.conv3_bf16(xt, wf)), y_ref, tol=8e-3, name='bf16 fwd vs exact')
    (a, b, uf, ud), = H.repack_multi([wt], [16])
    assert a is None and b is None and torch.equal(uf, wf) and torch.equal(ud, wd)
    with pytest.raises(ValueError):
        H.conv3_bf16(xt, wf[:, :32])                                # N must be a multiple of 64, taps contiguous


@pytest.mark.parametrize('ci,co,L,rows', [(64, 64, 56, 40), (128, 128, 28, 23), (512, 512, 7, 40), (64, 128, 1, 7),
                                          (128, 64, 2, 33), (64, 192, 3, 5), (64, 64, 57, 9), (256, 64, 14, 1),
                                          (64, 64, 56, 300), (192, 128, 9, 130)])
def test_conv3_wgrad_bf16(H, ci, co, L, rows):
    """bf16-operand weight gradient of the k3 s1 p1 conv (ds_read_b64_tr_b16 fragments, K over padded positions):
    slabs + shared reduction == the fp64 weight gradient of the bf16-ROUNDED (dy, x) to fp32 rounding, within bf16's
    2^-8 of the exact one; batched with a Winograd-form and a direct job in one call; accumulate form."""
    rng = np.random.default_rng(ci + co + L + rows)
    x = rng.standard_normal((rows, ci, L))
    dy = rng.standard_normal((rows, co, L))
    rb = np_ref.round_bf16
    w0 = np.zeros((co, ci, 3))
    _, dw_b = np_ref.conv1d_bwd(rb(x), w0, rb(dy), 1, 1, need_dx=False)
    _, dw_ref = np_ref.conv1d_bwd(x, w0, dy, 1, 1, need_dx=False)
    xt, dyt = rlc(x), rlc(dy)
    H.WGRAD_BF16 = True
    try:
        (slab,) = H.conv_wgrad_multi([(dyt, xt, 3, 1, 1)])
        dw = torch.zeros(co, ci, 3, device='cuda')
        H.wgrad_reduce_multi([(slab, dw)], accumulate=False)
        close(dw.cpu().numpy(), dw_b, tol=3e-6, name='bf16 wgrad vs rounded operands')
        close(dw.cpu().numpy(), dw_ref, tol=1.5e-2, name='bf16 wgrad vs exact')
        H.wgrad_reduce_multi([(slab, dw)], accumulate=True)
        close(dw.cpu().numpy(), 2 * dw_b, tol=3e-6, name='bf16 wgrad accumulate')
        # mixed batch: the stride-2 forms (k3 s2 p1 block head, k1 s2 downsample) ride along, bf16 too
        if L % 2 == 0 and L >= 2:
            dy2 = rng.standard_normal((rows, co, L // 2))
            _, dw2_b = np_ref.conv1d_bwd(rb(x), w0, rb(dy2), 2, 1, need_dx=False)
            _, dw1_b = np_ref.conv1d_bwd(rb(x), np.zeros((co, ci, 1)), rb(dy2), 2, 0, need_dx=False)
            s1, s2, s3 = H.conv_wgrad_multi([(dyt, xt, 3, 1, 1), (rlc(dy2), xt, 3, 2, 1), (rlc(dy2), xt, 1, 2, 0)])
            d1, d2 = torch.zeros(co, ci, 3, device='cuda'), torch.zeros(co, ci, 3, device='cuda')
            d3 = torch.zeros(co, ci, 1, device='cuda')
            H.wgrad_reduce_multi([(s1, d1), (s2, d2), (s3, d3)], accumulate=False)
            close(d1.cpu().numpy(), dw_b, tol=3e-6, name='bf16 wgrad in a mixed batch')
            close(d2.cpu().numpy(), dw2_b, tol=3e-6, name='bf16 k3 s2 wgrad')
            close(d3.cpu().numpy(), dw1_b, tol=3e-6, name='bf16 k1 s2 wgrad')
        elif L >= 3:                                  # odd input length: the stride-2 job stays on the fp32 kernel
            lo = (L - 1) // 2 + 1
            dy2 = rng.standard_normal((rows, co, lo))
            _, dw2_ref = np_ref.conv1d_bwd(x, w0, dy2, 2, 1, need_dx=False)
            (s2,) = H.conv_wgrad_multi([(rlc(dy2), xt, 3, 2, 1)])
            d2 = torch.zeros(co, ci, 3, device='cuda')
            H.wgrad_reduce_multi([(s2, d2)], accumulate=False)
            close(d2.cpu().numpy(), dw2_ref, tol=3e-6, name='direct wgrad for an odd length')
    finally:
        H.WGRAD_BF16 = False
    (slab32,) = H.conv_wgrad_multi([(dyt, xt, 3, 1, 1)])             # the flag is off again: fp32 Winograd form
    dw32 = torch.zeros(co, ci, 3, device='cuda')
    H.wgrad_reduce_multi([(slab32, dw32)], accumulate=False)
    close(dw32.cpu().numpy(), dw_ref, tol=4e-6, name='fp32 wgrad after the flag')


@pytest.mark.parametrize('ci,co,L,rows', [(64, 128, 56, 40), (128, 256, 28, 23), (256, 512, 14, 40), (64, 64, 2, 7),
                                          (128, 64, 6, 33), (64, 128, 56, 300), (192, 128, 10, 130)])
def test_stride2_convs_bf16(H, ci, co, L, rows):
    """Stride-2 block head (k3 s2 p1) and 1x1 downsample (k1 s2 p0) with bf16 operands: forward and data gradient
    (even positions: one tap; odd positions: two taps of neighbouring outputs; accumulate form) == the fp64 results
    on the bf16-rounded operands to fp32 rounding; bf16 packs for K = 1 and K = 3 from the batched repack."""
    rng = np.random.default_rng(ci + co + L + rows)
    rb = np_ref.round_bf16
    x = rng.standard_normal((rows, ci, L))
    xt = rlc(x)
    for k, pad in ((3, 1), (1, 0)):
        w = rng.standard_normal((co, ci, k)) * np.sqrt(2.0 / (k * co))
        wt = cu(w)
        (a, b, wf16, wd16), = H.repack_multi([wt], [16])
        assert a is None and b is None and tuple(wf16.shape) == (k, co, ci) and tuple(wd16.shape) == (k, ci, co)
        assert torch.equal(wf16, wt.permute(2, 0, 1).contiguous().bfloat16())
        assert torch.equal(wd16, wt.flip(2).permute(2, 1, 0).contiguous().bfloat16())
        y_b = np_ref.conv1d_fwd(rb(x), rb(w), 2, pad)
        close(ncl(H.conv_fwd_bf16_s2(xt, wf16)), y_b, tol=3e-6, name='bf16 s2 fwd k%d' % k)
        dy = rng.standard_normal(y_b.shape)
        dx_b, _ = np_ref.conv1d_bwd(rb(x), rb(w), rb(dy), 2, pad)
        dyt = rlc(dy)
        close(ncl(H.conv_dgrad_bf16_s2(dyt, wd16, L)), dx_b, tol=3e-6, name='bf16 s2 dgrad k%d' % k)
        base = rng.standard_normal(dx_b.shape)
        bt = rlc(base)
        H.conv_dgrad_bf16_s2(dyt, wd16, L, out=bt, accumulate=True)
        close(ncl(bt), base.astype(np.float32).astype(np.float64) + dx_b, tol=3e-6, name='bf16 s2 dgrad+acc k%d' % k)
        junk = torch.full_like(bt, 7.0)                            # non-accumulating form overwrites every position
        H.conv_dgrad_bf16_s2(dyt, wd16, L, out=junk, accumulate=False)
        close(ncl(junk), dx_b, tol=3e-6, name='bf16 s2 dgrad overwrite k%d' % k)
    with pytest.raises(ValueError):
        H.conv_fwd_bf16_s2(rlc(rng.standard_normal((2, ci, 7))), wf16)          # odd length
    # the block head and its downsample in one launch == the two single launches, bit for bit
    w3, w1 = cu(rng.standard_normal((co, ci, 3)) * 0.05), cu(rng.standard_normal((co, ci, 1)) * 0.1)
    (_, _, f3, _), (_, _, f1, _) = H.repack_multi([w3, w1], [16, 16])
    y3, y1 = H.conv_fwd_bf16_s2(xt, f3, f1)
    assert torch.equal(y3, H.conv_fwd_bf16_s2(xt, f3)) and torch.equal(y1, H.conv_fwd_bf16_s2(xt, f1))


@pytest.mark.parametrize('ci,co,L,rows', [(64, 128, 56, 40), (128, 256, 28, 23), (256, 512, 14, 300), (64, 128, 56, 300)])
def test_stride2_block_head_shared_launches(H, ci, co, L, rows):
    """The k3 s2 p1 conv and the k1 s2 downsample of a block: forward pair in one launch (da_conv_gemm_multi), the
    data gradient of both in two (odd positions + downsample's even positions, then the conv's even positions) ==
    the oracle and the one-problem launches (the last round's split-K half tiles fall on different tiles)."""
    rng = np.random.default_rng(ci + co + L + rows)
    x = rng.standard_normal((rows, ci, L))
    w1 = rng.standard_normal((co, ci, 3)) * np.sqrt(2.0 / (3 * co))
    wd = rng.standard_normal((co, ci, 1)) * np.sqrt(2.0 / co)
    y1_ref, yd_ref = np_ref.conv1d_fwd(x, w1, 2, 1), np_ref.conv1d_fwd(x, wd, 2, 0)
    dy1, dyd = rng.standard_normal(y1_ref.shape), rng.standard_normal(yd_ref.shape)
    dx_ref = np_ref.conv1d_bwd(x, w1, dy1, 2, 1)[0] + np_ref.conv1d_bwd(x, wd, dyd, 2, 0)[0]
    xt = rlc(x)
    wf1, wd1 = H.repack_weight(cu(w1), True, True)
    wfd, wdd = H.repack_weight(cu(wd), True, True)
    y1, yd = H.conv_fwd_multi([(xt, wf1, 2, 1), (xt, wfd, 2, 0)])
    close(ncl(y1), y1_ref, name='pair fwd conv')
    close(ncl(yd), yd_ref, name='pair fwd downsample')
    close(ncl(y1), ncl(H.conv_fwd(xt, wf1, 2, 1)).astype(np.float64), tol=1e-6, name='pair vs single fwd')
    dx = H.conv_dgrad_s2_pair(rlc(dy1), wd1, rlc(dyd), wdd, L)
    close(ncl(dx), dx_ref, name='pair dgrad')
    one = H.conv_dgrad(rlc(dy1), wd1, 2, 1, L)
    H.conv_dgrad(rlc(dyd), wdd, 2, 0, L, out=one, accumulate=True)
    close(ncl(dx), ncl(one).astype(np.float64), tol=1e-6, name='pair vs sequential')


def test_conv_mfma_layout_identity(H):
    """A = identity-like input with an ASYMMETRIC weight: catches a transposed C/D or A/B map."""
    ci = co = 32
    x = np.zeros((1, ci, 40))
    for l in range(32):
        x[0, l, l] = 1.0                      # position l carries unit channel l
    w = (np.arange(co)[:, None] * 100.0 + np.arange(ci)[None, :])[:, :, None]   # w[co][ci] asymmetric
    y = H.conv_fwd(rlc(x), H.repack_weight(cu(w))[0], 1, 0)
    ref = np_ref.conv1d_fwd(x, w, 1, 0)
    assert np.array_equal(ncl(y), ref)


@pytest.mark.parametrize('rows,L', [(40, 224), (3, 224), (20, 64)])
def test_stem_conv(H, rows, L):
    rng = np.random.default_rng(rows + L)
    x = rng.standard_normal((rows, 1, L))
    w = rng.standard_normal((64, 1, 7)) * 0.2
    y_ref = np_ref.conv1d_fwd(x, w, 2, 3)
    xt = cu(x[:, 0, :])
    y = H.stem_conv_fwd(xt, cu(w))
    close(ncl(y), y_ref, name='stem fwd')
    dy = rng.standard_normal(y_ref.shape)
    _, dw_ref = np_ref.conv1d_bwd(x, w, dy, 2, 3, need_dx=False)
    dw = H.stem_conv_wgrad(rlc(dy), xt)
    close(dw.cpu().numpy(), dw_ref, tol=3e-6, name='stem wgrad')


@pytest.mark.parametrize('C,L,R,W', [(64, 56, 20, 3), (128, 28, 20, 2), (512, 7, 20, 2), (96, 56, 20, 1),
                                     (64, 112, 20, 2), (32, 5, 3, 4), (256, 14, 20, 2), (32, 64, 20, 1),
                                     (512, 7, 20, 16), (64, 150, 20, 1)])
@pytest.mark.parametrize('two_stage', [False, True])
def test_bn_fwd_bwd(H, C, L, R, W, two_stage):
    """da_bn_fwd / da_bn_bwd take a single-pass register-resident kernel for Wn <= 1280 and the two-stage kernels
    above that; two_stage forces the latter so that both are checked on every shape."""
    H.bn_debug_two_stage(two_stage)
    try:
        _bn_fwd_bwd(H, C, L, R, W)
    finally:
        H.bn_debug_two_stage(False)


@pytest.mark.parametrize('C,L,R,W', [(64, 56, 20, 4), (128, 28, 20, 3), (32, 9, 7, 5)])
def test_bn_small_channel_groups(H, C, L, R, W):
    """The single-pass kernels with 8-channel blocks (tuning knob DA_BN_BLOCKS / da_bn_debug_target_blocks(n): aim for n
    blocks per launch; measured slower than the default 256 at B=64, kept for other batch sizes) against the oracle."""
    from deepards_amd import _lib
    _lib.lib().da_bn_debug_target_blocks(1 << 20)
    try:
        _bn_fwd_bwd(H, C, L, R, W)
    finally:
        _lib.lib().da_bn_debug_target_blocks(256)


def _bn_fwd_bwd(H, C, L, R, W):
    rng = np.random.default_rng(C + L + R)
    rows = W * R
    x = rng.standard_normal((rows, C, L)) * rng.uniform(0.2, 3, (1, C, 1)) + rng.uniform(-4, 4, (1, C, 1))
    gamma = rng.uniform(0.5, 1.5, C)
    beta = rng.standard_normal(C) * 0.3
    res = rng.standard_normal((rows, C, L))
    y_ref, st = np_ref.bn_window_fwd(x, gamma, beta, R)
    xt, gt, bt = rlc(x), cu(gamma), cu(beta)
    mean, invstd = H.bn_stats(xt, R)
    close(mean.cpu().numpy(), st[0], name='mean')
    close(invstd.cpu().numpy(), st[1], tol=5e-6, name='invstd')
    # plain, relu, relu+residual
    close(ncl(H.bn_apply(xt, R, mean, invstd, gt, bt, relu=False)), y_ref, tol=5e-6, name='bn')
    close(ncl(H.bn_apply(xt, R, mean, invstd, gt, bt, relu=True)), np.maximum(y_ref, 0), tol=5e-6, name='bn relu')
    out_ref = np.maximum(y_ref + res, 0)
    out = H.bn_apply(xt, R, mean, invstd, gt, bt, relu=True, res=rlc(res))
    close(ncl(out), out_ref, tol=5e-6, name='bn relu res')
    # backward, three mask modes
    dout = rng.standard_normal((rows, C, L))
    dt = rlc(dout)
    for mode, z in ((0, None), (1, y_ref), (2, y_ref + res)):
        g_ref = dout if z is None else dout * (z > 0)
        dx_ref, dg_ref, db_ref = np_ref.bn_window_bwd(x, gamma, st, g_ref, R)
        dx, dg, db, g, ds = H.bn_bwd(dt, xt, R, mean, invstd, gt, bt, mode, out=out if mode == 2 else None, want_g=True)
        # the ReLU decision of an element with |z| ~ fp32 rounding is undefined: leave those out (a handful at most)
        sure = np.ones_like(dout, dtype=bool) if z is None else np.abs(z) > 1e-5
        assert sure.mean() > 0.9999
        close(ncl(dx) * sure, dx_ref * sure, tol=2e-5, name='bn dx mode %d' % mode)
        close(ncl(g) * sure, g_ref * sure, tol=1e-6, name='bn g mode %d' % mode)
        slack = (np.abs(dout) * ~sure).sum(axis=(0, 2))            # what undecidable elements may contribute
        assert np.all(np.abs(ds[0].sum(0).cpu().numpy() - db_ref) <= 2e-5 * (1 + np.abs(db_ref).max()) + slack), 'ds1'
        xhat_abs = np.abs((y_ref - beta[None, :, None]) / gamma[None, :, None])
        slack_g = (np.abs(dout) * ~sure * xhat_abs).sum(axis=(0, 2))
        assert np.all(np.abs(dg.cpu().numpy() - dg_ref) <= 2e-5 * (1 + np.abs(dg_ref).max()) + slack_g), 'dgamma'
        assert np.all(np.abs(db.cpu().numpy() - db_ref) <= 2e-5 * (1 + np.abs(db_ref).max()) + slack), 'dbeta' 
    # ReLU decisions as a bit mask (single-pass geometry only): same gradients as mask_mode 2 without reading `out`
    o4, m4, i4, mk = H.bn_fwd(xt, R, gt, bt, relu=True, res=rlc(res), want_mask=True)
    close(ncl(o4), out_ref, tol=5e-6, name='bn_fwd(want_mask) out')
    if mk is not None:        # None: this shape / mode takes the two-stage kernels
        a = H.bn_bwd(dt, xt, R, m4, i4, gt, bt, 2, out=o4, want_g=True, defer_param_grads=True)
        b = H.bn_bwd(dt, xt, R, m4, i4, gt, bt, 2, want_g=True, defer_param_grads=True, mask=mk)
        assert torch.equal(a[0], b[0]) and torch.equal(a[3], b[3]) and torch.equal(a[4], b[4])
    # pass-through gradient added in the same pass (a concatenation's backward, densenet.py:41)
    extra = rng.standard_normal((rows, C + 32, L))
    et = rlc(extra)
    dxa, _, _, _, _ = H.bn_bwd(dt, xt, R, mean, invstd, gt, bt, 1, defer_param_grads=True, add=(et, 32))
    g1_ref = dout * (y_ref > 0)
    dx1_ref = np_ref.bn_window_bwd(x, gamma, st, g1_ref, R)[0]
    s1 = np.abs(y_ref) > 1e-5
    close(ncl(dxa) * s1, (dx1_ref + extra[:, 32:32 + C]) * s1, tol=2e-5, name='bn dx + add')
    # in-place form used by the block functions: dx aliases dout
    g_ref = dout * (y_ref > 0)
    dx_ref, dg_ref, db_ref = np_ref.bn_window_bwd(x, gamma, st, g_ref, R)
    sure = np.abs(y_ref) > 1e-5
    d2 = dt.clone()
    _, _, _, _, ds = H.bn_bwd(d2, xt, R, mean, invstd, gt, bt, 1, dx=d2, defer_param_grads=True)
    close(ncl(d2) * sure, dx_ref * sure, tol=2e-5, name='bn dx in-place')
    # deferred, batched parameter-gradient fold (accumulating form)
    dg2, db2 = torch.ones(C, device='cuda'), torch.ones(C, device='cuda')
    H.bn_param_grad_multi([(ds, dg2, db2)], accumulate=True)
    unsure = ~sure
    sl_b = (np.abs(dout) * unsure).sum(axis=(0, 2))
    sl_g = (np.abs(dout) * unsure * np.abs((y_ref - beta[None, :, None]) / gamma[None, :, None])).sum(axis=(0, 2))
    assert np.all(np.abs(dg2.cpu().numpy() - dg_ref - 1) <= 2e-5 * (1 + np.abs(dg_ref).max()) + sl_g), 'deferred dgamma'
    assert np.all(np.abs(db2.cpu().numpy() - db_ref - 1) <= 2e-5 * (1 + np.abs(db_ref).max()) + sl_b), 'deferred dbeta' 
    # statistics + normalisation in one call
    for relu_, res_, ref_ in ((False, None, y_ref), (True, None, np.maximum(y_ref, 0)), (True, res, out_ref)):
        o3, m3, i3 = H.bn_fwd(xt, R, gt, bt, relu=relu_, res=None if res_ is None else rlc(res_))
        close(ncl(o3), ref_, tol=5e-6, name='bn_fwd out')
        close(m3.cpu().numpy(), st[0], name='bn_fwd mean')
        close(i3.cpu().numpy(), st[1], tol=5e-6, name='bn_fwd invstd')
    # statistics through the fused consumer: bn_apply merges the chunk records and publishes mean/invstd
    part = H.bn_stats_partial(xt, R)
    m2_, i2_ = torch.empty_like(mean), torch.empty_like(invstd)
    o2 = H.bn_apply(xt, R, m2_, i2_, gt, bt, relu=True, part=part)
    close(ncl(o2), np.maximum(y_ref, 0), tol=5e-6, name='bn apply(part)')
    close(m2_.cpu().numpy(), st[0], name='mean via apply')
    close(i2_.cpu().numpy(), st[1], tol=5e-6, name='invstd via apply')
    # running stats (sequential per-window momentum updates)
    rm0, rv0 = rng.standard_normal(C), rng.uniform(0.5, 2, C)
    rm, rv = cu(rm0), cu(rv0)
    nbt = torch.zeros(1, dtype=torch.int64, device='cuda')
    H.bn_stats(xt, R, running_mean=rm, running_var=rv, num_batches_tracked=nbt)
    assert int(nbt) == W
    # oracle closed form starts from given buffers
    m_ref, v_ref = rm0.copy(), rv0.copy()
    var_b = 1.0 / st[1] ** 2 - 1e-5
    n = R * L
    for i in range(W):
        m_ref = 0.9 * m_ref + 0.1 * st[0][i]
        v_ref = 0.9 * v_ref + 0.1 * var_b[i] * n / (n - 1)
    close(rm.cpu().numpy(), m_ref, tol=5e-6, name='running mean')
    close(rv.cpu().numpy(), v_ref, tol=2e-5, name='running var')


@pytest.mark.parametrize('mode', [0, 1])
def test_stem_bn_relu_pool(H, mode):
    rng = np.random.default_rng(5 + mode)
    rows, C, L, R = 40, 64, 112, 20
    y0 = rng.standard_normal((rows, C, L))
    gamma, beta = rng.uniform(0.5, 1.5, C), rng.standard_normal(C) * 0.3
    z, st = np_ref.bn_window_fwd(y0, gamma, beta, R)
    z = np.maximum(z, 0)
    if mode == 0:
        out_ref, idx = np_ref.maxpool3s2p1_fwd(z)
    else:
        out_ref = np_ref.avgpool3s2p1_fwd(z)
    yt, gt, bt = rlc(y0), cu(gamma), cu(beta)
    mean, invstd = H.bn_stats(yt, R)
    out = H.bn_relu_pool_fwd(yt, R, mean, invstd, gt, bt, mode)
    close(ncl(out), out_ref, tol=5e-6, name='pool fwd')
    dout = rng.standard_normal(out_ref.shape)
    dz_ref = np_ref.maxpool3s2p1_bwd(dout, idx, L) if mode == 0 else np_ref.avgpool3s2p1_bwd(dout, L)
    dz = H.pool_bwd(rlc(dout), yt, R, mean, invstd, gt, bt, mode)
    # ties (both zero after ReLU) route differently but are masked by the ReLU afterwards
    close(ncl(dz) * (z > 0), dz_ref * (z > 0), tol=1e-6, name='pool bwd')


def test_avgpools(H):
    rng = np.random.default_rng(9)
    x = rng.standard_normal((40, 128, 28))
    close(ncl(H.avgpool_fwd(rlc(x), 2)), np_ref.avgpool_fwd(x, 2, 2), name='avg2')
    d = rng.standard_normal((40, 128, 14))
    close(ncl(H.avgpool_bwd(rlc(d), 28, 2)), np_ref.avgpool_bwd(d, 2, 2, 28), name='avg2 bwd')
    x7 = rng.standard_normal((40, 512, 7))
    close(ncl(H.avgpool_fwd(rlc(x7), 7)), np_ref.avgpool_fwd(x7, 7, 1), name='avg7')
    d7 = rng.standard_normal((40, 512, 1))
    close(ncl(H.avgpool_bwd(rlc(d7), 7, 7)), np_ref.avgpool_bwd(d7, 7, 1, 7), name='avg7 bwd')


@pytest.mark.parametrize('B,K', [(4, 10240), (7, 2560), (64, 10240)])
def test_head_and_loss(H, B, K):
    rng = np.random.default_rng(B + K)
    flat = rng.standard_normal((B, K))
    w = rng.uniform(-1, 1, (2, K)) / np.sqrt(K)
    bias = rng.uniform(-0.1, 0.1, 2)
    tgt = np.zeros((B, 2))
    tgt[np.arange(B), rng.integers(0, 2, B)] = 1
    logits_ref = np_ref.linear_fwd(flat, w, bias)
    loss_ref, dl_ref = np_ref.bce_with_logits(logits_ref, tgt)
    ft, wt, bt = cu(flat), cu(w), cu(bias)
    logits = H.linear2_fwd(ft, wt, bt)
    close(logits.cpu().numpy(), logits_ref, name='logits')
    loss, dl = H.bce_logits(logits, cu(tgt))
    assert abs(float(loss) - loss_ref) < 1e-6
    close(dl.cpu().numpy(), dl_ref, tol=1e-6, name='dlogits')
    dflat, dw, db = H.linear2_bwd(dl, ft, wt)
    close(dflat.cpu().numpy(), dl_ref @ w, tol=1e-6, name='dflat')
    close(dw.cpu().numpy(), dl_ref.T @ flat, tol=2e-6, name='dW')
    close(db.cpu().numpy(), dl_ref.sum(0), tol=1e-6, name='dbias')


def test_bce_extremes(H):
    x = np.array([[-40.0, 40.0], [0.0, 1e-4], [88.0, -88.0], [15.0, -15.0]])
    t = np.array([[0.0, 1.0], [1.0, 0.0], [0.0, 1.0], [1.0, 0.0]])
    loss_ref, d_ref = np_ref.bce_with_logits(x, t)
    loss, d = H.bce_logits(cu(x), cu(t))
    assert abs(float(loss) - loss_ref) < 1e-5 * (1 + abs(loss_ref))
    close(d.cpu().numpy(), d_ref, tol=1e-6, name='bce extreme grads')


def test_optimizers(H):
    rng = np.random.default_rng(3)
    n = 100003
    p0 = rng.standard_normal(n) * 0.1
    p_ref = p0.copy()
    buf_ref = None
    pt = cu(p0)
    buf = torch.empty_like(pt)
    for step in range(3):
        g = rng.standard_normal(n) * 0.02            # about a third of the entries exceed the clip
        gc = np_ref.clamp_grad(g, 0.01)
        p_ref, buf_ref = np_ref.sgd_nesterov_step(p_ref, gc, buf_ref, first=(step == 0))
        H.clamp_sgd_nesterov_(pt, cu(g), buf, 1e-3, 0.9, 1e-4, 0.01, step == 0)
    close(pt.cpu().numpy(), p_ref, tol=1e-6, name='sgd')
    # gscale: clamp AFTER the 1/world scaling
    pt2, buf2 = cu(p0), torch.empty(n, device='cuda')
    g = rng.standard_normal(n) * 0.05
    H.clamp_sgd_nesterov_(pt2, cu(g * 4), buf2, 1e-3, 0.9, 1e-4, 0.01, True, gscale=0.25)
    pr, _ = np_ref.sgd_nesterov_step(p0, np_ref.clamp_grad(g, 0.01), None, first=True)
    close(pt2.cpu().numpy(), pr, tol=1e-6, name='sgd gscale')
    # adam
    pa, m_ref, v_ref = p0.copy(), 0.0, 0.0
    pt3 = cu(p0)
    m, v = torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda')
    for step in range(1, 4):
        g = rng.standard_normal(n) * 0.02
        pa, m_ref, v_ref = np_ref.adam_step(pa, np_ref.clamp_grad(g, 0.01), m_ref, v_ref, step)
        H.clamp_adam_(pt3, cu(g), m, v, 1e-3, step, 0.01)
    close(pt3.cpu().numpy(), pa, tol=2e-6, name='adam')


def test_concat_slice_dropout(H):
    rng = np.random.default_rng(4)
    a, b = rng.standard_normal((20, 64, 28)), rng.standard_normal((20, 32, 28))
    cat = H.concat2(rlc(a), rlc(b))
    assert np.array_equal(ncl(cat), np.concatenate([a, b], 1).astype(np.float32).astype(np.float64))
    s = H.slice_channels(cat, 64, 32)
    assert torch.equal(s, rlc(b))
    acc = rlc(a)
    H.slice_channels(cat, 0, 64, out=acc, accumulate=True)
    close(ncl(acc), 2 * a, name='slice acc')
    x = torch.ones(1 << 20, device='cuda')
    seed = torch.tensor([12345], dtype=torch.int64, device='cuda')
    y1 = H.dropout(x, seed, 3, 0.2)
    y2 = H.dropout(x, seed, 3, 0.2)
    y3 = H.dropout(x, seed, 4, 0.2)
    assert torch.equal(y1, y2) and not torch.equal(y1, y3)
    keep = (y1 > 0).float().mean().item()
    assert abs(keep - 0.8) < 0.005
    assert torch.allclose(y1[y1 > 0], torch.tensor(1.25, device='cuda'))
    # the fused forms carry exactly the mask of dropout() on the contiguous new-feature tensor
    bt = rlc(b)
    catd = H.concat2(rlc(a), bt, drop=(seed, 7, 0.2))
    assert torch.equal(catd[:, :, :64], rlc(a)) and torch.equal(catd[:, :, 64:], H.dropout(bt, seed, 7, 0.2))
    sd = H.slice_channels(cat, 64, 32, drop=(seed, 7, 0.2))
    assert torch.equal(sd, H.dropout(bt, seed, 7, 0.2))


@pytest.mark.parametrize('B,NB,F', [(3, 20, 128), (64, 20, 512), (2, 7, 32), (5, 1, 64), (4, 64, 16)])
def test_window_median(H, B, NB, F):
    """Lower median over the NB breath rows (torch.median semantics, models/torch_cnn_linear_network.py:47) and its
    gradient; ties (post-ReLU zeros) go to the earliest row."""
    rng = np.random.default_rng(B + NB + F)
    x = np.maximum(rng.standard_normal((B, NB, F)), -0.3).astype(np.float32)
    x[x == -0.3] = 0.0                                              # plenty of exact ties
    xt = cu(x.reshape(B * NB, F))
    med, idx = H.window_median_fwd(xt, NB)
    order = np.argsort(x, axis=1, kind='stable')[:, (NB - 1) // 2, :]
    ref = np.take_along_axis(x, order[:, None, :], axis=1)[:, 0, :]
    assert np.array_equal(med.cpu().numpy(), ref)
    assert np.array_equal(idx.cpu().numpy(), order)
    tm = torch.median(torch.from_numpy(x), dim=1)[0].numpy()
    assert np.array_equal(med.cpu().numpy(), tm)
    dout = rng.standard_normal((B, F)).astype(np.float32)
    dx = H.window_median_bwd(cu(dout), idx, NB).cpu().numpy().reshape(B, NB, F)
    dref = np.zeros_like(x)
    np.put_along_axis(dref, order[:, None, :], dout[:, None, :], axis=1)
    assert np.array_equal(dx, dref)


@pytest.mark.parametrize('B,T,F,Hd', [(3, 20, 128, 16), (64, 20, 512, 16), (2, 5, 128, 8), (4, 7, 64, 64)])
def test_lstm_recurrence(H, B, T, F, Hd):
    """da_lstm_fwd / da_lstm_bwd (+ the GEMMs around them, as LSTMFunction composes them) vs the numpy LSTM of the
    oracle (pinned to nn.LSTM through the reference goldens): states, outputs, all gradients; with and without an
    initial state."""
    from deepards_amd.functional import LSTMFunction
    rng = np.random.default_rng(B + T + F + Hd)
    x = rng.standard_normal((B, T, F))
    k = 1.0 / np.sqrt(Hd)
    w_ih, w_hh = rng.uniform(-k, k, (4 * Hd, F)), rng.uniform(-k, k, (4 * Hd, Hd))
    b_ih, b_hh = rng.uniform(-k, k, 4 * Hd), rng.uniform(-k, k, 4 * Hd)
    dh = rng.standard_normal((B, T, Hd))
    for init in (False, True):
        h0 = rng.standard_normal((B, Hd)) if init else None
        c0 = rng.standard_normal((B, Hd)) if init else None
        hs_ref, (ht_ref, ct_ref), tape = np_ref.lstm_fwd(x, w_ih, w_hh, b_ih, b_hh, h0, c0)
        dx_ref, dwi_ref, dwh_ref, db_ref = np_ref.lstm_bwd(x, w_ih, w_hh, tape, dh)
        feat = cu(x.reshape(B * T, F)).requires_grad_(True)
        ps = [cu(a).requires_grad_(True) for a in (w_ih, w_hh, b_ih, b_hh)]
        hs, ht, ct = LSTMFunction.apply(feat, *ps, T, None if h0 is None else cu(h0), None if c0 is None else cu(c0))
        close(hs.detach().cpu().numpy(), hs_ref, tol=3e-6, name='lstm hs')
        close(ht.detach().cpu().numpy()[0], ht_ref, tol=3e-6, name='lstm hT')
        close(ct.detach().cpu().numpy()[0], ct_ref, tol=3e-6, name='lstm cT')
        hs.backward(cu(dh))
        close(feat.grad.cpu().numpy().reshape(B, T, F), dx_ref, tol=5e-6, name='lstm dx')
        close(ps[0].grad.cpu().numpy(), dwi_ref, tol=5e-6, name='lstm dW_ih')
        close(ps[1].grad.cpu().numpy(), dwh_ref, tol=5e-6, name='lstm dW_hh')
        close(ps[2].grad.cpu().numpy(), db_ref, tol=5e-6, name='lstm db_ih')
        close(ps[3].grad.cpu().numpy(), db_ref, tol=5e-6, name='lstm db_hh')


def test_bad_arguments_are_refused(H):
    x = torch.zeros(4, 8, 48, device='cuda')               # C = 48 is not a multiple of 32
    with pytest.raises(ValueError):
        H.conv_fwd(x, torch.zeros(3, 32, 48, device='cuda'), 1, 1)
    with pytest.raises(ValueError):
        H.bn_stats(torch.zeros(5, 8, 32, device='cuda'), 2)  # rows not a multiple of R
    with pytest.raises(ValueError):
        H.conv_fwd(torch.zeros(4, 8, 32), torch.zeros(3, 32, 32), 1, 1)   # CPU tensors
