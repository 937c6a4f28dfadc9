"""The recomputing default stem (csrc/stem_pool.hip stem_bn_relu_pool_fwd_kernel / stem_bwd_kernel, csrc/bn.hip
stem_stats_partial_kernel): conv k7 s2 p3 -> BatchNorm -> ReLU -> pool(3,2,1) from the RAW rows, the conv output never
stored (reference models/resnet.py:86-87,100-104,141-153, models/densenet.py:118-124).  Checked against the kernels it
replaces (stem_conv_fwd + bn_stats + bn_relu_pool_fwd; pool_bwd + bn_bwd + stem_conv_wgrad), which the oracle tests pin."""
import os
import sys
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def H():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from deepards_amd import hip_ops
    return hip_ops


@pytest.mark.parametrize('rows,R,lin,c,pool_mode', [(1280, 20, 224, 64, 0), (40, 20, 224, 64, 1), (60, 20, 224, 64, 0),
                                                    (80, 40, 512, 64, 0), (24, 4, 30, 32, 0), (20, 20, 224, 128, 0)])
def test_fused_stem_matches_the_stored_map_path(H, rows, R, lin, c, pool_mode):
    torch.manual_seed(rows + lin)
    x = torch.randn(rows, lin, device='cuda')
    x[1] = 0.0                                      # a silent row: ties in the pool, ReLU at exactly beta
    x[2, ::3] = 0.5
    w = torch.randn(c, 1, 7, device='cuda') * 0.4
    gamma = torch.rand(c, device='cuda') + 0.5
    beta = torch.randn(c, device='cuda') * 0.3
    gamma[3] = -0.7                                 # a negative scale: the pool's maximum sits at the smallest y
    # the path it replaces
    y0 = H.stem_conv_fwd(x, w)
    mean0, invstd0 = H.bn_stats(y0, R, 1e-5)
    out0 = H.bn_relu_pool_fwd(y0, R, mean0, invstd0, gamma, beta, pool_mode)
    out, mean, invstd = H.stem_fused_fwd(x, w, R, gamma, beta, pool_mode)
    assert torch.equal(mean, mean0) and torch.equal(invstd, invstd0)           # same chunk records, same merge
    assert torch.equal(out, out0)                                              # bit for bit
    if c % 16 == 0:
        out3, _, _ = H.stem_fused_fwd(x, w, R, gamma, beta, pool_mode, out_x3=True)
        assert torch.equal(out3, H.bn_relu_pool_fwd(y0, R, mean0, invstd0, gamma, beta, pool_mode, out_x3=True))
    dout = torch.randn_like(out0)
    dz = H.pool_bwd(dout, y0, R, mean0, invstd0, gamma, beta, pool_mode)
    dy0, dg0, db0, _, _ = H.bn_bwd(dz, y0, R, mean0, invstd0, gamma, beta, 1, dx=dz)
    dw0 = H.stem_conv_wgrad(dy0, x)
    dw, ds = H.stem_fused_bwd(dout, x, w, R, mean, invstd, gamma, beta, pool_mode)
    dg = torch.zeros(c, device='cuda'); db = torch.zeros(c, device='cuda')
    H.bn_param_grad_multi([(ds, dg, db)], accumulate=False)
    scale = float(dw0.abs().max())
    assert float((dw - dw0).abs().max()) < 2e-5 * scale, (float((dw - dw0).abs().max()), scale)
    assert float((dg - dg0).abs().max()) < 2e-5 * float(dg0.abs().max())
    assert float((db - db0).abs().max()) < 2e-5 * float(db0.abs().max())
    acc = dw0.clone()
    H.stem_fused_bwd(dout, x, w, R, mean, invstd, gamma, beta, pool_mode, dw=acc, accumulate=True)
    assert float((acc - 2 * dw0).abs().max()) < 4e-5 * scale


@pytest.mark.parametrize('rows,R,lin,c,pool_mode', [(40, 20, 224, 64, 0), (40, 20, 224, 64, 1), (80, 40, 512, 64, 0),
                                                    (24, 4, 30, 32, 0), (20, 20, 224, 128, 0)])
def test_fused_stem_against_the_numpy_oracle(H, rows, R, lin, c, pool_mode):
    """The default stem DIRECTLY against oracle/np_ref.py (fp64): conv k7 s2 p3 -> per-window BatchNorm -> ReLU ->
    Max/AvgPool1d(3,2,1) forward, and the backward's dW / dgamma / dbeta (reference models/resnet.py:86-87,100-104,
    141-153).  Tolerances are those of the kernels it replaced (tests/test_hip_ops_gpu.py: 2e-6 ... 2e-5 of the scale).
    Inputs keep every max-pool / ReLU decision away from its boundary by more than fp32 noise (random data, no ties), so
    the fp64 backward takes the decisions the fp32 kernels take; the tie / silent-row cases are the bit-for-bit test above."""
    import numpy as np
    from oracle import np_ref
    rng = np.random.RandomState(rows * 7 + lin + c)
    x = rng.randn(rows, 1, lin)
    w = rng.randn(c, 1, 7) * 0.4
    gamma = rng.rand(c) + 0.5
    beta = rng.randn(c) * 0.3
    gamma[3] = -0.7
    y0 = np_ref.conv1d_fwd(x, w, 2, 3)
    z, st = np_ref.bn_window_fwd(y0, gamma, beta, R)
    a = np_ref.relu(z)
    if pool_mode == 0:
        out_ref, idx = np_ref.maxpool3s2p1_fwd(a)
    else:
        out_ref = np_ref.avgpool3s2p1_fwd(a)
    cu = lambda v: torch.from_numpy(np.ascontiguousarray(v).astype(np.float32)).cuda()
    xt, wt, gt, bt = cu(x[:, 0, :]), cu(w), cu(gamma), cu(beta)
    out, mean, invstd = H.stem_fused_fwd(xt, wt, R, gt, bt, pool_mode)
    got = out.cpu().numpy().astype(np.float64).transpose(0, 2, 1)
    scale = 1.0 + np.abs(out_ref).max()
    assert np.abs(got - out_ref).max() <= 5e-6 * scale, np.abs(got - out_ref).max()
    assert np.abs(mean.cpu().numpy() - st[0]).max() <= 2e-6 * (1.0 + np.abs(st[0]).max())
    assert np.abs(invstd.cpu().numpy() / st[1] - 1.0).max() <= 2e-5
    # backward.  A pool window that holds an ambiguous decision -- a pre-activation within 1e-5 of the ReLU's edge, or (max
    # pool) two largest candidates within 1e-4 of each other -- gets NO upstream gradient: nothing then flows through an
    # element whose decision fp32 and fp64 may take differently, and the comparison is about values, not decisions.
    dout = rng.randn(*out_ref.shape)
    n_, c_, l_ = a.shape
    lo = out_ref.shape[2]
    win = lambda v, fill: np.stack([np.pad(v, ((0, 0), (0, 0), (1, 1)), constant_values=fill)[:, :, t:t + (lo - 1) * 2 + 1:2]
                                    for t in range(3)], axis=-1)                      # (N, C, Lo, 3)
    bad = (win(np.abs(z), 1.0) <= 1e-5).any(axis=-1)
    if pool_mode == 0:
        cand = np.sort(win(a, -np.inf), axis=-1)
        bad |= ((cand[..., 2] - cand[..., 1]) <= 1e-4) & (cand[..., 2] > 0)
    dout = dout * ~bad
    da = np_ref.maxpool3s2p1_bwd(dout, idx, l_) if pool_mode == 0 else np_ref.avgpool3s2p1_bwd(dout, l_)
    dz = da * (z > 0)
    dy0, dgamma_ref, dbeta_ref = np_ref.bn_window_bwd(y0, gamma, st, dz, R)
    _, dw_ref = np_ref.conv1d_bwd(x, w, dy0, 2, 3, need_dx=False)
    assert bad.mean() < 0.01
    dw, ds = H.stem_fused_bwd(cu(dout.transpose(0, 2, 1)), xt, wt, R, mean, invstd, gt, bt, pool_mode)
    dg = torch.zeros(c, device='cuda'); db = torch.zeros(c, device='cuda')
    H.bn_param_grad_multi([(ds, dg, db)], accumulate=False)
    for name, got_, ref_ in (('dW', dw.cpu().numpy().astype(np.float64), dw_ref),
                             ('dgamma', dg.cpu().numpy().astype(np.float64), dgamma_ref),
                             ('dbeta', db.cpu().numpy().astype(np.float64), dbeta_ref)):
        err = np.abs(got_ - ref_).max()
        assert err <= 2e-5 * (1.0 + np.abs(ref_).max()), '%s: max err %.3e (scale %.3e)' % (name, err, np.abs(ref_).max())
