"""CPU: pin the numpy oracle (oracle/np_ref.py) to the golden vectors captured from the real
reference (oracle/make_golden.py).  float64 oracle vs float64 reference: tight tolerances."""
import glob
import os
import numpy as np
import pytest

from oracle import np_ref
from oracle.weights import seeded_params, digest, DEAD_RESNET_PARAMS

GOLD = sorted(p for p in glob.glob(os.path.join(os.path.dirname(__file__), 'golden', '*net18_*.npz'))
              if not os.path.basename(p).startswith(('head_', 'opt_')))
OPT_GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), 'golden', 'opt_*.npz')))
HEAD_GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), 'golden', 'head_*.npz')))
BB_GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), 'golden', 'bb_*.npz')))


def _load(path):
    z = np.load(path, allow_pickle=False)
    return {k: z[k] for k in z.files}


@pytest.mark.parametrize('path', GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_np_oracle_matches_reference(path):
    g = _load(path)
    backbone = str(g['backbone'])
    params = {k: v.astype(np.float64)
              for k, v in seeded_params(backbone, int(g['seed']), bn_bias_shift=float(g['bn_bias_shift'])).items()}
    out = np_ref.cnn_linear_forward_backward(params, g['x'].astype(np.float64), g['target'].astype(np.float64),
                                             backbone=backbone, first_pool_type=str(g['first_pool_type']))
    np.testing.assert_allclose(out['logits'], g['logits64'], rtol=0, atol=1e-10)
    assert abs(out['loss'] - float(g['loss64'])) < 1e-12
    checked = 0
    for k in g:
        if k.startswith('grad64/'):
            name = k[len('grad64/'):]
            np.testing.assert_allclose(digest(out['grads'][name]), g[k], rtol=1e-8, atol=1e-9, err_msg=name)
            checked += 1
    dead = [n for n in DEAD_RESNET_PARAMS if n in params]
    assert checked == len(params) - (len(dead) if backbone == 'resnet18' else 0)
    # the fp32 reference sits within 1e-5 of the fp64 one: the 1e-4 parity budget is meaningful
    assert np.abs(g['logits32'] - g['logits64']).max() < 1e-5


@pytest.mark.parametrize('path', HEAD_GOLD, ids=[os.path.basename(p)[:-4] for p in HEAD_GOLD])
def test_np_oracle_sibling_heads_match_reference(path):
    """CNNLinearToMean / CNNLinearComprToRF / CNNSingleBreathLinearNetwork / CNNDoubleLinearNetwork
    (torch_cnn_linear_network.py:7-89) restated in the oracle vs the reference classes (oracle/make_golden_heads.py)."""
    g = _load(path)
    backbone, head = str(g['backbone']), str(g['head'])
    params = {k: v.astype(np.float64) for k, v in
              seeded_params(backbone, int(g['seed']), bn_bias_shift=float(g['bn_bias_shift']), head=head).items()}
    out = np_ref.cnn_linear_forward_backward(params, g['x'].astype(np.float64), g['target'].astype(np.float64),
                                             backbone=backbone, first_pool_type=str(g['first_pool_type']), head=head)
    assert out['logits'].shape == g['logits64'].shape
    np.testing.assert_allclose(out['logits'], g['logits64'], rtol=0, atol=1e-10)
    assert abs(out['loss'] - float(g['loss64'])) < 1e-12
    checked = 0
    for k in g:
        if k.startswith('grad64/'):
            name = k[len('grad64/'):]
            np.testing.assert_allclose(digest(out['grads'][name]), g[k], rtol=1e-8, atol=1e-9, err_msg=name)
            checked += 1
    assert checked >= 60 and 'linear_final.weight' in out['grads']
    if head == 'double_linear':
        assert 'linear_intermediate.weight' in out['grads']
    if head == 'lstm':
        np.testing.assert_allclose(out['hx'], g['hx64'][0], rtol=0, atol=1e-12)
        np.testing.assert_allclose(out['cx'], g['cx64'][0], rtol=0, atol=1e-12)
        assert 'lstm.weight_hh_l0' in out['grads']


def opt_reference(g):
    """The oracle run for a constructor-option golden (oracle/make_golden_options.py)."""
    backbone = str(g['backbone'])
    params = {k: v.astype(np.float64) for k, v in seeded_params(backbone, int(g['seed']), bn_bias_shift=float(g['bn_bias_shift']),
                                                                   in_ch=int(g['in_ch'])).items()}
    out = np_ref.cnn_linear_forward_backward(params, g['x'].astype(np.float64), g['target'].astype(np.float64),
                                             backbone=backbone, first_pool_type=str(g['first_pool_type']),
                                             double_conv_first=bool(g['opt_double_conv_first']) if 'opt_double_conv_first' in g else False)
    return params, out


@pytest.mark.parametrize('path', OPT_GOLD, ids=[os.path.basename(p)[:-4] for p in OPT_GOLD])
def test_np_oracle_constructor_options_match_reference(path):
    """densenet18(with_fft / only_fft / fft_real_only) (densenet.py:109-115: conv0 with 3 / 2 / 2 / 1 input channels) and
    resnet18(double_conv_first=True) (resnet.py:90-96,142-149) in the oracle vs the reference classes: logits, loss, every
    gradient; with double_conv_first conv1 is the dead parameter and conv1_alt / conv2 / bn2 are live."""
    g = _load(path)
    params, out = opt_reference(g)
    np.testing.assert_allclose(out['logits'], g['logits64'], rtol=0, atol=1e-10)
    assert abs(out['loss'] - float(g['loss64'])) < 1e-12
    names = [k[len('grad64/'):] for k in g if k.startswith('grad64/')]
    for name in names:
        np.testing.assert_allclose(digest(out['grads'][name]), g['grad64/' + name], rtol=1e-8, atol=1e-9, err_msg=name)
    assert sorted(names) == sorted(out['grads'])
    if 'opt_double_conv_first' in g:
        assert 'breath_block.conv1.weight' not in names and 'breath_block.conv2.weight' in names
        assert 'breath_block.conv1_alt.weight' in names and 'breath_block.bn2.bias' in names
    else:
        assert params['breath_block.features.conv0.weight'].shape[1] == g['x'].shape[2] == int(g['in_ch'])


def test_perform_fft_restates_the_reference_call_for_call():
    """tiles.perform_fft vs the literal expression of ARDSRawDataset._perform_fft (dataset.py:1330-1341) -- including the
    axes-less fftshift that also rolls the sub-batch rows by NB // 2 -- and the per-channel scaling factors."""
    from deepards_amd.tiles import perform_fft, scaling_factors_for_indices
    rng = np.random.RandomState(3)
    w = rng.randn(5, 20, 1, 224) * 20
    full = perform_fft(w, add_fft=True)
    assert full.shape == (5, 20, 3, 224)
    for i in range(5):
        trans = np.fft.fftshift(np.fft.fft(w[i], axis=-1))
        assert np.array_equal(full[i], np.concatenate([w[i], trans.real, trans.imag], axis=1))
        # the spectrum rows sit 10 rows away from the flow rows they came from (fftshift over every axis)
        spec = np.fft.fftshift(np.fft.fft(w[i], axis=-1), axes=-1)
        assert np.allclose(full[i][:, 1], np.roll(spec.real[:, 0], 10, axis=0))
    assert perform_fft(w, only_fft=True).shape == (5, 20, 2, 224)
    assert perform_fft(w, add_fft=True, fft_real_only=True).shape == (5, 20, 2, 224)
    assert perform_fft(w, only_fft=True, fft_real_only=True).shape == (5, 20, 1, 224)
    assert perform_fft(w) is not None and np.array_equal(perform_fft(w), w)
    mu, std = scaling_factors_for_indices(full, [0, 2, 3])
    sel = full[[0, 2, 3]]
    assert mu.shape == (3,) and np.allclose(mu, sel.mean(axis=(0, 1, 3))) and np.allclose(std, sel.std(axis=(0, 1, 3)))


@pytest.mark.parametrize('path', BB_GOLD, ids=[os.path.basename(p)[:-4] for p in BB_GOLD])
def test_np_oracle_other_backbones_match_reference(path):
    """resnet34 (models/resnet.py:178) and densenet121 (models/densenet.py:234) through the same oracle blocks vs the
    reference classes (oracle/make_golden_backbones.py)."""
    g = _load(path)
    backbone = str(g['backbone'])
    params = {k: v.astype(np.float64) for k, v in seeded_params(backbone, int(g['seed'])).items()}
    out = np_ref.cnn_linear_forward_backward(params, g['x'].astype(np.float64), g['target'].astype(np.float64), backbone=backbone)
    np.testing.assert_allclose(out['logits'], g['logits64'], rtol=0, atol=1e-10)
    assert abs(out['loss'] - float(g['loss64'])) < 1e-12
    checked = 0
    for k in g:
        if k.startswith('grad64/'):
            np.testing.assert_allclose(digest(out['grads'][k[7:]], 24), g[k], rtol=1e-8, atol=1e-9, err_msg=k)
            checked += 1
    assert checked > 100


def test_np_oracle_sgd_trajectory():
    g = _load([p for p in GOLD if 'resnet18_b2_randn' in p][0])
    params = {k: v.astype(np.float64) for k, v in seeded_params('resnet18', 0).items()}
    x, t = g['x'].astype(np.float64), g['target'].astype(np.float64)
    bufs = {}
    losses = []
    for step in range(3):
        out = np_ref.cnn_linear_forward_backward(params, x, t, backbone='resnet18')
        losses.append(out['loss'])
        for n in out['grads']:
            gr = np_ref.clamp_grad(out['grads'][n], 0.01)
            params[n], bufs[n] = np_ref.sgd_nesterov_step(params[n], gr, bufs.get(n), first=(step == 0))
    np.testing.assert_allclose(losses, g['sgd_losses64'], rtol=0, atol=1e-10)
    for k in g:
        if k.startswith('sgd_p64/'):
            name = k[len('sgd_p64/'):]
            np.testing.assert_allclose(digest(params[name]), g[k], rtol=1e-9, atol=1e-12, err_msg=name)


def test_np_oracle_adam_trajectory():
    g = _load([p for p in GOLD if 'densenet18_b2_randn' in p][0])
    params = {k: v.astype(np.float64) for k, v in seeded_params('densenet18', 0).items()}
    x, t = g['x'].astype(np.float64), g['target'].astype(np.float64)
    m, v = {}, {}
    for step in range(3):
        out = np_ref.cnn_linear_forward_backward(params, x, t, backbone='densenet18')
        for n in out['grads']:
            gr = np_ref.clamp_grad(out['grads'][n], 0.01)
            params[n], m[n], v[n] = np_ref.adam_step(params[n], gr, m.get(n, 0.0), v.get(n, 0.0), step + 1)
    for k in g:
        if k.startswith('adam_p64/'):
            name = k[len('adam_p64/'):]
            np.testing.assert_allclose(digest(params[name]), g[k], rtol=1e-7, atol=1e-10, err_msg=name)


def test_running_stats_closed_form():
    g = _load([p for p in GOLD if 'resnet18_b2_randn' in p][0])
    params = {k: v.astype(np.float64) for k, v in seeded_params('resnet18', 0).items()}
    out = np_ref.cnn_linear_forward_backward(params, g['x'].astype(np.float64), None, backbone='resnet18')
    for short, full in (('bn1', 'breath_block.bn1'), ('layer4.1.bn2', 'breath_block.layer4.1.bn2')):
        st, cnt = out['stats'][full]
        c = st[0].shape[1]
        rm, rv = np_ref.bn_running_update(np.zeros(c), np.ones(c), st, cnt)
        # the golden ran forward twice: model(x) for the whole batch and breath_block(x[0]) -> 3 updates
        st0 = (st[0][:1], st[1][:1])
        rm, rv = np_ref.bn_running_update(rm, rv, st0, cnt)
        np.testing.assert_allclose(rm, g['rm64/' + short], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(rv, g['rv64/' + short], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize('path', GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_torch_oracle_matches_reference(path):
    """oracle/torch_ref.py (the CPU-baseline path) against the fp32 reference goldens."""
    import torch
    from oracle import torch_ref
    g = _load(path)
    backbone = str(g['backbone'])
    p = {k: torch.from_numpy(v)
         for k, v in seeded_params(backbone, int(g['seed']), bn_bias_shift=float(g['bn_bias_shift'])).items()}
    out = torch_ref.cnn_linear(p, torch.from_numpy(g['x']), backbone, str(g['first_pool_type']))
    assert np.abs(out.detach().numpy() - g['logits32']).max() < 2e-6 * max(1.0, np.abs(g['logits32']).max())


def test_torch_oracle_sgd_matches_reference():
    import torch
    from oracle import torch_ref
    g = _load([p for p in GOLD if 'resnet18_b2_randn' in p][0])
    p = {k: torch.from_numpy(v) for k, v in seeded_params('resnet18', 0).items()}
    tr = torch_ref.CpuReferenceTrainer(p, 'resnet18')
    x, t = torch.from_numpy(g['x']), torch.from_numpy(g['target'])
    losses = [float(tr.step(x, t)) for _ in range(3)]
    assert np.abs(np.array(losses) - g['sgd_losses32']).max() < 2e-6


@pytest.mark.skipif(not os.path.isdir('/root/reference/deepards/models'), reason='reference only in the build container')
def test_torch_oracle_bitwise_vs_reference_import():
    """In the build container only: same ATen ops in the same order => bit-identical logits."""
    import sys
    import torch
    from oracle import torch_ref
    sys.path.insert(0, '/root/reference')
    try:
        from deepards.models.resnet import resnet18
        from deepards.models.densenet import densenet18
        from deepards.models.torch_cnn_linear_network import CNNLinearNetwork
    finally:
        sys.path.remove('/root/reference')
    x = torch.from_numpy(_load(GOLD[0])['x'])
    for backbone, ctor in (('resnet18', lambda: resnet18()), ('densenet18', lambda: densenet18(drop_rate=0))):
        ref = CNNLinearNetwork(ctor(), 20, 0)
        p = {k: torch.from_numpy(v) for k, v in seeded_params(backbone, 5).items()}
        ref.load_state_dict(p, strict=False)
        with torch.no_grad():
            a = ref.train()(x, None)
            b = torch_ref.cnn_linear(p, x, backbone)
        assert torch.equal(a, b), backbone


def test_reference_fixture_windows_through_both_oracles():
    """The reference's own pickled fixture (deepards/tests/test_dataset.pkl, arrays extracted without
    unpickling by oracle/extract_fixture.py): z-scored with the fixture's scaling factors exactly as
    ARDSRawDataset.__getitem__ does (dataset.py:1364,1379), through the numpy and the torch oracle."""
    import torch
    from oracle import torch_ref
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'test_dataset_windows.npz'))
    x = (z['x'] - float(z['mu'])) / float(z['std'])                     # float64, like the reference
    assert x.shape == (20, 20, 1, 224) and z['target'].sum() == 20 and z['target'][:, 1].sum() == 15
    xb, tb = x[:4], z['target'][:4].astype(np.float64)
    for backbone in ('resnet18', 'densenet18'):
        p64 = {k: v.astype(np.float64) for k, v in seeded_params(backbone, 3).items()}
        out = np_ref.cnn_linear_forward_backward(p64, xb, tb, backbone=backbone, need_grads=False)
        pt = {k: torch.from_numpy(v) for k, v in seeded_params(backbone, 3).items()}
        lt = torch_ref.cnn_linear(pt, torch.from_numpy(xb).float(), backbone).detach().numpy()
        assert np.abs(lt - out['logits']).max() < 1e-5


def test_round_bf16_is_round_to_nearest_even():
    """np_ref.round_bf16 (the operand rounding of the bf16 conv arithmetic) == torch's float32 -> bfloat16 conversion,
    bit for bit, including ties, negative values and values that round up into the next exponent."""
    import torch
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(20000) * 10.0 ** rng.integers(-6, 6, 20000),
                        np.array([0.0, -0.0, 1.0, 1.00390625, 1.0078125, 1.01171875, -1.00390625, 255.5, 3.3895e38,
                                  1e-30, 65280.0, 65408.0])]).astype(np.float32)
    got = np_ref.round_bf16(x)
    want = torch.from_numpy(x).bfloat16().double().numpy()
    assert np.array_equal(got, want)
    assert np_ref.round_bf16(np.float32(1.00390625)) == 1.0 and np_ref.round_bf16(np.float32(1.01171875)) == 1.015625   # ties to even



@pytest.mark.parametrize('tag', ['densenet18_b4_flow', 'resnet18_b4_flow'])
def test_decision_matching_explains_the_reference_fp32_gradients(tag):
    """The yardstick of the GPU gradient tests, pinned on the reference's OWN fp32 path (oracle/torch_ref.py in float32,
    bit-identical to the imported reference): its gradients differ from the fp64 ones by 3.1e-2 / 6e-3 rel-l2 on the
    flow goldens, and agree to < 1e-5 once the exact gradients are taken under the decisions the fp32 run made -- two or
    three ReLU elements whose pre-activation is within 2e-6 of zero (tests/tools/decision_match.py)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tools'))
    from decision_match import decision_matched_gradients
    from oracle import torch_ref as T
    import torch
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', tag + '.npz'), allow_pickle=False)
    backbone, pool = str(g['backbone']), str(g['first_pool_type'])
    p = seeded_params(backbone, int(g['seed']), bn_bias_shift=float(g['bn_bias_shift']))
    p32 = {k: torch.from_numpy(v).float().requires_grad_(True) for k, v in p.items()}
    out = T.cnn_linear(p32, torch.from_numpy(g['x']).float(), backbone, first_pool_type=pool)
    torch.nn.BCEWithLogitsLoss()(out, torch.from_numpy(g['target']).float()).backward()
    ours = {k: v.grad.numpy().astype(np.float64) for k, v in p32.items() if v.grad is not None}
    ref = np_ref.cnn_linear_forward_backward({k: v.astype(np.float64) for k, v in p.items()}, g['x'].astype(np.float64),
                                             g['target'].astype(np.float64), backbone=backbone, first_pool_type=pool)
    rl2 = lambda a, b: float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))
    before = max(rl2(ours[n], ref['grads'][n]) for n in ours)
    got, flips, ncand = decision_matched_gradients(ref, ours, tag, log=lambda *a: None)
    after = max(rl2(ours[n], got[n]) for n in ours)
    assert before > 3e-3 and after < 1e-5, (before, after)
    assert 1 <= len(flips) <= 4 and all(m < 3e-6 for _, _, m in flips), flips
    again = ref['rebackward']([])                             # the tape is restored after every re-run
    assert all(np.array_equal(again[n], ref['grads'][n]) for n in again)
