"""Conv arithmetic 'f32x3' with PRE-SPLIT operands (deepards_amd/csrc/conv_x3p.hip, the x3 activation format of
csrc/common.h): fp32-equivalent convolutions on the bf16 matrix cores whose operands their producers already stored as
exact three-term bf16 splits.

What is pinned here: (1) the x3 format is lossless -- merge(split(x)) == x bit for bit, the weight packs' three terms add
up to the fp32 weight exactly; (2) the conv kernel (full tiles, the half tiles of the last round, ragged edges, sequences
shorter than the taps, accumulate) against fp64 at the fp32 kernels' own error level; (3) the BatchNorm / pool kernels' x3
store forms equal their float forms BIT FOR BIT after merging (the x3 pipeline changes how a value is stored, never the
value); (4) the weight-gradient kernel on x3 operands against fp64.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
LOG = os.path.join(os.path.dirname(os.path.dirname(__file__)), 'gpurun_out', 'parity_model.log')


def log(*a):
    os.makedirs(os.path.dirname(LOG), exist_ok=True)
    with open(LOG, 'a') as f:
        f.write(' '.join(str(x) for x in a) + '\n')


@pytest.fixture(scope='module')
def H():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from deepards_amd import hip_ops
    return hip_ops


def _err(y, ref):
    return float((y.double() - ref).abs().max() / ref.abs().max())


def test_x3_format_is_lossless(H):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(7, 13, 64, generator=g)
    x[0] *= 1e-25                       # small (|x| >= 2^-109: below that the third term underflows), huge, zeros, sparse significands
    x[1] *= 1e30
    x[2, :, :32] = 0.0
    x[3, :, 0] = 1.0 + 2.0 ** -20
    x[4] = torch.round(x[4] * 8) / 8
    xd = x.cuda()
    s = H.x3_split(xd)
    assert H.is_x3(s) and tuple(s.shape) == (7, 13, 4, 3, 16)
    assert torch.equal(H.x3_merge(s), xd)
    # the planes ARE h = bf16(x), m = bf16(x - h), l = bf16(x - h - m)
    h = xd.bfloat16()
    m = (xd - h.float()).bfloat16()
    l = (xd - h.float() - m.float()).bfloat16()
    v = s.view(7, 13, 4, 3, 16)
    for plane, want in enumerate((h, m, l)):
        assert torch.equal(v[:, :, :, plane, :].reshape(7, 13, 64), want)


def _pack49(H, w):
    return H.repack_multi([w], [49])[0][2:]


def test_chunked_weight_pack_terms_add_up_exactly(H):
    torch.manual_seed(3)
    w = torch.randn(128, 64, 3, device='cuda') * 0.1
    uf, ud = _pack49(H, w)
    assert tuple(uf.shape) == (2, 4, 18, 64, 8) and tuple(ud.shape) == (1, 8, 18, 64, 8)
    # element (t, n, c), term s at [n/64][c/16][(t*2 + (n%64)/32)*3 + s][(c%16/8)*32 + n%32][c%8]
    f = uf.float().view(2, 4, 3, 2, 3, 2, 32, 8)            # nt, cg, t, nb, s, kg, n32, c8
    tot = f.sum(dim=4)                                      # h + m + l  (fp32 add of three bf16: exact here? check in fp64)
    tot = uf.double().view(2, 4, 3, 2, 3, 2, 32, 8).sum(dim=4)
    got = tot.permute(2, 0, 3, 5, 1, 4, 6).reshape(3, 128, 64)     # [t][nt, nb, n32][cg, kg, c8]
    assert torch.equal(got, w.double().permute(2, 0, 1))
    d = ud.double().view(1, 8, 3, 2, 3, 2, 32, 8).sum(dim=4).permute(2, 0, 3, 5, 1, 4, 6).reshape(3, 64, 128)
    assert torch.equal(d, w.double().permute(2, 1, 0).flip(0))    # data gradient: channels swapped, taps reversed


@pytest.mark.parametrize('rows,L,ci,co', [(40, 56, 64, 64), (37, 7, 128, 64), (3, 1, 64, 128), (5, 2, 64, 64),
                                           (20, 14, 256, 256), (19, 28, 64, 192), (1280, 7, 512, 512), (1280, 56, 64, 64),
                                           (1281, 14, 256, 256)])
def test_conv3_x3p_forward_and_data_gradient_against_fp64(H, rows, L, ci, co):
    """Forward (Uf pack) and data gradient (Ud pack): max error below 3e-6 of the output scale for K up to 1536 and never
    above 1.5x the native fp32 direct kernel's on the same data; B = 64 shapes exercise full tiles + the half tiles of the
    last round (560 tiles = 512 + 48), the small ones half tiles only; ragged M, L shorter than the taps, accumulate."""
    torch.manual_seed(rows * 131 + L)
    x = torch.randn(rows, L, ci, device='cuda')
    w = torch.randn(co, ci, 3, device='cuda') * (2.0 / (3 * co)) ** 0.5
    ref = torch.nn.functional.conv1d(x.double().transpose(1, 2), w.double(), padding=1).transpose(1, 2)
    uf, ud = _pack49(H, w)
    x3 = H.x3_split(x)
    y = H.conv3_x3p(x3, uf)
    e = _err(y, ref)
    wdir, _ = H.repack_weight(w, True, True)
    e32 = _err(H.conv_fwd(x, wdir, 1, 1), ref)
    log('conv3_x3p %s: err vs fp64 %.2e (fp32 direct kernel %.2e)' % ((rows, L, ci, co), e, e32))
    assert e < 3e-6 and e <= 1.5 * e32 + 2e-7
    base = torch.randn_like(y)
    acc = base.clone()
    H.conv3_x3p(x3, uf, out=acc, accumulate=True)
    assert _err(acc - base, ref) < 3e-6
    dy = torch.randn(rows, L, co, device='cuda')
    dref = torch.nn.functional.conv_transpose1d(dy.double().transpose(1, 2), w.double(), padding=1).transpose(1, 2)
    assert _err(H.conv3_x3p(H.x3_split(dy), ud), dref) < 3e-6


@pytest.mark.parametrize('W,L,c', [(3, 56, 64), (2, 28, 128), (5, 7, 512), (64, 14, 256)])
def test_batchnorm_and_pool_x3_store_forms_equal_the_float_forms(H, W, L, c):
    """bn_fwd_x / bn_bwd_x / bn_relu_pool_fwd(out_x3): the SAME kernels as bn_fwd / bn_bwd / bn_relu_pool_fwd with the
    store (and the residual load) going through the x3 format -- merged results are bit-identical, statistics too."""
    R = 20
    rows = W * R
    torch.manual_seed(W * 7 + L)
    x = torch.randn(rows, L, c, device='cuda') * 2 + 0.3
    res = torch.randn(rows, L, c, device='cuda')
    gamma, beta = torch.rand(c, device='cuda') + 0.5, torch.randn(c, device='cuda') * 0.1
    assert H.bn_x3_ok(rows, L, c, R)
    for relu, r in ((True, None), (False, None), (True, res)):
        want, m0, i0, mask0 = H.bn_fwd(x, R, gamma, beta, relu=relu, res=r, want_mask=True) if relu else \
            H.bn_fwd(x, R, gamma, beta, relu=relu, res=r) + (None,)
        for res_in in ((None,) if r is None else (r, H.x3_split(r))):
            for out_x3 in (True, False):
                got = H.bn_fwd_x(x, R, gamma, beta, relu=relu, res=res_in, want_mask=relu, out_x3=out_x3)
                o = H.x3_merge(got[0]) if out_x3 else got[0]
                assert torch.equal(o, want) and torch.equal(got[1], m0) and torch.equal(got[2], i0)
                if relu:
                    assert torch.equal(got[3], mask0)
    out, mean, invstd, mask = H.bn_fwd(x, R, gamma, beta, relu=True, res=res, want_mask=True)
    dout = torch.randn(rows, L, c, device='cuda')
    dx0, _, _, g0, ds0 = H.bn_bwd(dout, x, R, mean, invstd, gamma, beta, 2, want_g=True, defer_param_grads=True, mask=mask)
    for dx_x3 in (True, False):
        dx, g, ds = H.bn_bwd_x(dout, x, R, mean, invstd, gamma, beta, 3, want_g=True, mask=mask, dx_x3=dx_x3)
        assert torch.equal(H.x3_merge(dx) if dx_x3 else dx, dx0) and torch.equal(g, g0) and torch.equal(ds, ds0)
    out1, mean1, invstd1 = H.bn_fwd(x, R, gamma, beta, relu=True)
    dx1, _, _, _, ds1 = H.bn_bwd(dout, x, R, mean1, invstd1, gamma, beta, 1, defer_param_grads=True)
    dx, g, ds = H.bn_bwd_x(dout, x, R, mean1, invstd1, gamma, beta, 1)
    assert g is None and torch.equal(H.x3_merge(dx), dx1) and torch.equal(ds, ds1)
    if L % 2 == 0 and c == 64:
        y = torch.randn(rows, 2 * L, c, device='cuda')
        ms, isd = H.bn_stats(y, R, 1e-5)
        for mode in (0, 1):
            want = H.bn_relu_pool_fwd(y, R, ms, isd, gamma, beta, mode)
            assert torch.equal(H.x3_merge(H.bn_relu_pool_fwd(y, R, ms, isd, gamma, beta, mode, out_x3=True)), want)


@pytest.mark.parametrize('ci,co,L,rows', [(64, 64, 56, 40), (128, 64, 28, 60), (64, 128, 7, 37), (512, 512, 7, 1280)])
def test_weight_gradient_on_x3_operands_against_fp64(H, ci, co, L, rows):
    """dW of the k3 s1 p1 conv from x3 operands (job code 49, wgrad_x3p_multi_kernel): error vs fp64 at the fp32
    kernels' level; the slabs go through the shared reduction."""
    torch.manual_seed(ci + L)
    x = torch.randn(rows, L, ci, device='cuda')
    dy = torch.randn(rows, L, co, device='cuda') * 1e-3
    w = torch.zeros(co, ci, 3, device='cuda', dtype=torch.float64, requires_grad=True)
    yy = torch.nn.functional.conv1d(x.double().transpose(1, 2), w, padding=1)
    (ref,) = torch.autograd.grad(yy, w, dy.double().transpose(1, 2))
    (slab,) = H.conv_wgrad_multi([(H.x3_split(dy), H.x3_split(x), 3, 1, 1)])
    dw = torch.zeros(co, ci, 3, device='cuda')
    H.wgrad_reduce_multi([(slab, dw)], accumulate=False)
    (slab32,) = H.conv_wgrad_multi([(dy, x, 3, 1, 1)])
    dw32 = torch.zeros(co, ci, 3, device='cuda')
    H.wgrad_reduce_multi([(slab32, dw32)], accumulate=False)
    e, e32 = _err(dw, ref), _err(dw32, ref)
    log('wgrad on x3 operands %s: err vs fp64 %.2e (fp32 kernels %.2e)' % ((ci, co, L, rows), e, e32))
    assert e < 5e-6


class arithmetic(object):
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        from deepards_amd import functional as F_
        self.prev = F_.conv_dtype()
        F_.set_conv_dtype(self.name)

    def __exit__(self, *exc):
        from deepards_amd import functional as F_
        F_.set_conv_dtype(self.prev)


def _goldens(kind):
    import test_model_gpu as TM
    return [p for p in getattr(TM, kind) if 'resnet18' in os.path.basename(p)]


@pytest.mark.parametrize('path', _goldens('GOLD'), ids=[os.path.basename(p)[:-4] for p in _goldens('GOLD')])
def test_reference_goldens_hold_under_f32x3p(path):
    """The reference-golden parity test of tests/test_model_gpu.py, UNCHANGED (logits 1e-4, loss 1e-5, decision-matched
    gradients 1e-4, flip caps), with the x3 flow: every k3 s1 conv (forward, data gradient, weight gradient) on
    pre-split operands, BatchNorm / pool kernels storing the x3 format."""
    import test_model_gpu as TM
    import deepards_amd.models as M
    with arithmetic('f32x3p'):
        TM.test_logits_and_grads_match_reference_golden(M, path)


@pytest.mark.parametrize('tag,opt,use_graph', [('resnet18_b2_randn', 'sgd', True), ('resnet18_b2_active', 'sgd', True),
                                               ('resnet18_b2_active', 'adam', True), ('resnet18_b2_randn', 'sgd', False)])
def test_reference_trajectories_hold_under_f32x3p(tag, opt, use_graph):
    """The 3-step optimiser trajectories of the reference (tests/test_model_gpu.py, unchanged bounds) through the captured
    step with the x3 flow."""
    import test_model_gpu as TM
    import deepards_amd.models as M
    with arithmetic('f32x3p'):
        TM.test_trainer_trajectory_matches_reference(M, tag, opt, use_graph)


def test_x3_flow_is_what_runs_and_matches_the_float_path_closely():
    """Under 'f32x3p' the block inputs / hidden activations really are x3 tensors (no silent fall-back to the fp32
    kernels), B = 64 (full tiles + the half tiles of the last round) agrees with the default fp32 path to 2e-5 on the
    logits, and option paths (avg pool, double_conv_first, resnet34) run."""
    import deepards_amd.models as M
    from deepards_amd import functional as F_, hip_ops as H_
    from oracle.weights import seeded_batch
    x, t = seeded_batch(64, 20, 4)
    xt = torch.from_numpy(x).cuda()
    torch.manual_seed(5)
    m = M.CNNLinearNetwork(M.resnet18(), 20, 0).cuda().train()
    with torch.no_grad():
        want = m(xt, None)
    seen, s2seen, wjobs = [], [], []
    orig, orig_f, orig_d, orig_w = H_.conv3_x3p, H_.conv_x3p_s2_fwd, H_.conv_x3p_s2_dgrad, H_.conv_wgrad_multi

    def spy(x3, wpk, out=None, accumulate=False):
        seen.append(tuple(x3.shape))
        return orig(x3, wpk, out=out, accumulate=accumulate)

    def spy_f(x3, w1pk, wdpk):
        s2seen.append(('fwd', tuple(x3.shape)))
        return orig_f(x3, w1pk, wdpk)

    def spy_d(dy1, w1pk, dyd, wdpk, out=None):
        s2seen.append(('dgrad', tuple(dy1.shape)))
        return orig_d(dy1, w1pk, dyd, wdpk, out=out)

    def spy_w(jobs):
        wjobs.extend((H_.is_x3(j[0]) and H_.is_x3(j[1]), j[2], j[3]) for j in jobs)
        return orig_w(jobs)
    with arithmetic('f32x3p'):
        H_.conv3_x3p, H_.conv_x3p_s2_fwd, H_.conv_x3p_s2_dgrad, H_.conv_wgrad_multi = spy, spy_f, spy_d, spy_w
        try:
            out = m(xt, None)
            out.sum().backward()
        finally:
            H_.conv3_x3p, H_.conv_x3p_s2_fwd, H_.conv_x3p_s2_dgrad, H_.conv_wgrad_multi = orig, orig_f, orig_d, orig_w
        assert len(seen) == 26 and all(len(s_) == 5 for s_ in seen)       # 13 k3 s1 convs: forward + data gradient
        # the three stride-2 block entries: one forward and one data-gradient launch each, all on x3 operands ...
        assert sorted(k for k, _ in s2seen) == ['dgrad'] * 3 + ['fwd'] * 3
        # ... and every residual-block weight gradient (13 k3 s1, 3 k3 s2, 3 1x1 s2) as an x3 job
        assert sorted((k, s_) for x3job, k, s_ in wjobs if x3job) == [(1, 2)] * 3 + [(3, 1)] * 13 + [(3, 2)] * 3
        assert not any(k <= 3 and not x3job for x3job, k, s_ in wjobs)
        assert float((out.detach() - want).abs().max()) < 2e-5
        for bb in (M.resnet18(first_pool_type='avg'), M.resnet18(double_conv_first=True), M.resnet34()):
            mm = M.CNNLinearNetwork(bb, 20, 0).cuda().train()
            o = mm(xt[:4], None)
            o.sum().backward()
            assert torch.isfinite(o).all()
    assert F_.conv_dtype() == 'f32'


S2_SHAPES = [(64, 128, 56, 40), (128, 256, 28, 40), (256, 512, 14, 40), (64, 64, 8, 3), (128, 64, 6, 77), (256, 512, 14, 1280)]


@pytest.mark.gpu
@pytest.mark.parametrize('ci,co,lin,rows', S2_SHAPES)
def test_conv_x3p_s2_pair_matches_fp64(H, ci, co, lin, rows):
    """The stride-2 block entry on x3 operands (conv_x3p_s2_kernel): the k3 s2 p1 conv and the 1x1 s2 downsample conv of
    one launch, and their summed data gradient, against the fp64 convolutions (reference models/resnet.py:16-19,123-131);
    bound: the fp32 direct kernels' own error scale."""
    torch.manual_seed(ci + lin)
    x = torch.randn(rows, lin, ci, device='cuda')
    w1 = torch.randn(co, ci, 3, device='cuda') * (2.0 / (3 * ci)) ** 0.5
    wd = torch.randn(co, ci, 1, device='cuda') * (2.0 / ci) ** 0.5
    (_, _, uf1, ud1), (_, _, ufd, udd) = H.repack_multi([w1, wd], [49, 49])
    y1, yd = H.conv_x3p_s2_fwd(H.x3_split(x), uf1, ufd)
    xd = x.double().permute(0, 2, 1)
    r1 = torch.nn.functional.conv1d(xd, w1.double(), stride=2, padding=1).permute(0, 2, 1)
    rd = torch.nn.functional.conv1d(xd, wd.double(), stride=2).permute(0, 2, 1)
    e1 = float((y1.double() - r1).abs().max() / r1.abs().max())
    ed = float((yd.double() - rd).abs().max() / rd.abs().max())
    dy1, dyd = torch.randn_like(y1) * 1e-2, torch.randn_like(yd) * 1e-2
    dx = H.conv_x3p_s2_dgrad(H.x3_split(dy1), ud1, H.x3_split(dyd), udd)
    rdx = torch.nn.functional.conv_transpose1d(dy1.double().permute(0, 2, 1), w1.double(), stride=2, padding=1, output_padding=1) + \
        torch.nn.functional.conv_transpose1d(dyd.double().permute(0, 2, 1), wd.double(), stride=2, output_padding=1)
    rdx = rdx.permute(0, 2, 1)
    ex = float((dx.double() - rdx).abs().max() / rdx.abs().max())
    log('x3p s2 pair %s: fwd k3 %.2e, 1x1 %.2e, dgrad %.2e vs fp64' % ((ci, co, lin, rows), e1, ed, ex))
    assert e1 < 3e-6 and ed < 3e-6 and ex < 3e-6


@pytest.mark.parametrize('ci,co,lin,rows', [(64, 128, 56, 40), (128, 256, 28, 33), (256, 512, 14, 64), (64, 64, 6, 5)])
def test_wgrad_on_x3_operands_stride_2(H, ci, co, lin, rows):
    """dW of the k3 s2 p1 conv and of the 1x1 s2 downsample conv from x3 operands (job code 49, modes 1 / 2 of
    wgrad_x3p_multi_kernel), one multi launch: error vs fp64 at the fp32 kernels' scale (reference models/resnet.py:16-19,123-131)."""
    torch.manual_seed(lin)
    x = torch.randn(rows, lin, ci, device='cuda')
    dy1 = torch.randn(rows, lin // 2, co, device='cuda') * 1e-2
    dyd = torch.randn(rows, lin // 2, co, device='cuda') * 1e-2
    x3 = H.x3_split(x)
    s1, sd = H.conv_wgrad_multi([(H.x3_split(dy1), x3, 3, 2, 1), (H.x3_split(dyd), x3, 1, 2, 0)])
    dw1 = torch.empty(co, ci, 3, device='cuda'); dwd = torch.empty(co, ci, 1, device='cuda')
    H.wgrad_reduce_multi([(s1, dw1), (sd, dwd)], accumulate=False)
    xp = torch.nn.functional.pad(x.double(), (0, 0, 1, 1))
    r1 = torch.stack([torch.einsum('rjn,rjc->nc', dy1.double(), xp[:, t:t + lin:2][:, :lin // 2]) for t in range(3)], 2)
    rd = torch.einsum('rjn,rjc->nc', dyd.double(), x.double()[:, 0::2])[:, :, None]
    e1, ed = _err(dw1, r1), _err(dwd, rd)
    log('wgrad on x3 operands, stride 2 %s: k3 %.2e, 1x1 %.2e vs fp64' % ((ci, co, lin, rows), e1, ed))
    assert e1 < 3e-6 and ed < 3e-6


def test_x3p_captured_training_is_reproducible_and_tracks_fp32():
    """60 captured SGD steps of cnn_linear + resnet18 at B = 64 under 'f32x3p' (LDS-DMA k3 s1 kernel, stride-2 pair kernels,
    x3 weight gradients: every hand-synchronised ring in the arithmetic): two runs from the same seed agree BIT FOR BIT in
    every loss and every parameter (a missed wait shows up here), and the trajectory stays within 5e-5 in the loss of the
    default fp32 arithmetic's (measured 1.3e-5 over 120 steps)."""
    import numpy as np
    import deepards_amd.models as M
    from deepards_amd.train import HotPathTrainer
    from oracle.weights import seeded_batch
    x, t = seeded_batch(64, 20, 4)
    xt, tt = torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()

    def run(name):
        with arithmetic(name):
            torch.manual_seed(7)
            m = M.CNNLinearNetwork(M.resnet18(), 20, 0).cuda()
            tr = HotPathTrainer(m, optimizer='sgd', use_graph=True)
            losses = np.array([float(tr.train_step(xt, tt)) for _ in range(60)])
            params = torch.cat([q.detach().reshape(-1) for q in m.parameters()]).cpu().numpy()
            tr.release_graphs()
        return losses, params
    la, pa = run('f32x3p')
    lb, pb = run('f32x3p')
    lf, _ = run('f32')
    assert np.array_equal(la, lb) and np.array_equal(pa, pb)
    log('x3p 60 captured steps: reproducible; max loss difference to fp32 %.2e' % np.abs(la - lf).max())
    assert np.abs(la - lf).max() < 5e-5
