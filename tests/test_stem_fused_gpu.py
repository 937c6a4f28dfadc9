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
