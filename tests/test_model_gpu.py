"""GPU parity of the whole hot path against (a) the golden vectors captured from the real reference
(tests/golden, oracle/make_golden.py) and (b) the numpy oracle on fresh seeded inputs.

Bar (BASELINE.json north_star): logits / gradients within 1e-4 (fp32) of the reference CPU path."""
import glob
import os
import sys

import numpy as np
import pytest
import torch

from oracle import np_ref
from oracle.weights import seeded_params, seeded_batch, digest, DEAD_RESNET_PARAMS

pytestmark = pytest.mark.gpu
GOLD = sorted(p for p in glob.glob(os.path.join(os.path.dirname(__file__), 'golden', '*net18_*.npz'))
              if not os.path.basename(p).startswith(('head_', 'opt_')))
OPT_GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), 'golden', 'opt_*.npz')))
HEAD_GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), 'golden', 'head_*.npz')))
BB_GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), 'golden', 'bb_*.npz')))
LOG = os.path.join(os.path.dirname(os.path.dirname(__file__)), 'gpurun_out', 'parity_model.log')


def log(*a):
    os.makedirs(os.path.dirname(LOG), exist_ok=True)
    with open(LOG, 'a') as f:
        f.write(' '.join(str(x) for x in a) + '\n')


@pytest.fixture(scope='module')
def M():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    import deepards_amd.models as models
    return models


def build(M, backbone, seed, first_pool_type='max', drop_rate=0.0, shift=0.0):
    if backbone == 'resnet18':
        bb = M.resnet18(first_pool_type=first_pool_type)
    else:
        bb = M.densenet18(drop_rate=drop_rate)
    model = M.CNNLinearNetwork(bb, 20, 0)
    sd = {k: torch.from_numpy(v) for k, v in seeded_params(backbone, seed, bn_bias_shift=shift).items()}
    missing = model.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys
    return model.cuda().train()


def _gold(path):
    z = np.load(path, allow_pickle=False)
    return {k: z[k] for k in z.files}


def rel_l2(a, b):
    nb = float(np.linalg.norm(b))
    return float(np.linalg.norm(a - b) / (nb if nb > 1e-9 else 1.0))      # ~zero references: absolute


sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tools'))
from decision_match import decision_matched_gradients as _dmg, assert_gradients_match as _agm, feature_reference      # noqa: E402


def decision_matched_gradients(ref, ours, log_tag=''):
    return _dmg(ref, ours, log_tag, log=log)


def assert_gradients_match(ref, ours, tag='', strict=False, max_flips=12, allow=None, taps=None):
    """1e-4 against the oracle's exact gradients under the decisions this run took (tests/tools/decision_match.py)."""
    return _agm(ref, ours, tag, strict=strict, max_flips=max_flips, log=log, allow=allow, taps=taps)


class tapped(object):
    """with tapped() as taps: out = model(x) -- the block Functions record their post-ReLU activations (the decisions this
    forward takes, deepards_amd.functional.DECISION_TAP) for assert_gradients_match(taps=taps)."""

    def __enter__(self):
        from deepards_amd import functional as F_
        F_.DECISION_TAP = []
        return F_.DECISION_TAP

    def __exit__(self, *exc):
        from deepards_amd import functional as F_
        F_.DECISION_TAP = None


def grads64(model, ref):
    return {n: p.grad.cpu().numpy().astype(np.float64) for n, p in model.named_parameters()
            if p.grad is not None and n in ref['grads']}


@pytest.mark.parametrize('path', GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_logits_and_grads_match_reference_golden(M, path):
    """Logits: 1e-4 absolute (north star), every golden, against the reference's fp64 capture.
    Gradients, every parameter, every golden: within 1e-4 (rel-l2 <= 1e-4, or max abs err <= 1e-4 * max(1, max|ref|))
    of the EXACT gradients -- the numpy oracle's, which tests/test_oracle_golden.py pins to the reference's captured
    fp64 gradients at 1e-10 -- under the activation decisions this run took (``decision_matched_gradients``: an fp32
    forward legitimately takes the other branch of a ReLU whose pre-activation is within ~1e-6 of zero; the flips
    adopted are logged, each must be an element the oracle itself lists as within 3e-5 of the boundary, and there may
    be only a handful).  Since round 3 the ReLU decisions behind the stem are not searched for but EXPORTED by the run
    (functional.DECISION_TAP -> decision_match.hip_relu_flips); only the stem's fused ReLU / max-pool is matched by
    pursuit.  '*_active' goldens (every ReLU active, avg first pool) have no decision to flip.
    The reference's own fp32 capture (grad32/) is held to the same yardstick for scale: its rel-l2 against grad64/ is
    logged next to ours."""
    g = _gold(path)
    backbone = str(g['backbone'])
    shift = float(g['bn_bias_shift'])
    strict = shift > 0
    model = build(M, backbone, int(g['seed']), str(g['first_pool_type']), shift=shift)
    x = torch.from_numpy(g['x']).cuda()
    t = torch.from_numpy(g['target']).cuda()
    from deepards_amd.functional import bce_with_logits
    with tapped() as taps:
        out = model(x, None)
    loss = bce_with_logits(out, t)
    loss.backward()
    logits = out.detach().cpu().numpy().astype(np.float64)
    err = np.abs(logits - g['logits64']).max()
    ref32 = np.abs(g['logits32'] - g['logits64']).max()
    log(os.path.basename(path), 'logits max|hip-ref64| %.3e  (ref32-ref64 %.3e)  loss %.8f vs %.8f' %
        (err, ref32, float(loss), float(g['loss64'])))
    assert err < 1e-4
    assert abs(float(loss) - float(g['loss64'])) < 1e-5
    params64 = {k: v.astype(np.float64) for k, v in seeded_params(backbone, int(g['seed']), bn_bias_shift=shift).items()}
    ref = np_ref.cnn_linear_forward_backward(params64, g['x'].astype(np.float64), g['target'].astype(np.float64),
                                             backbone=backbone, first_pool_type=str(g['first_pool_type']))
    ours = {}
    for n, p in model.named_parameters():
        key = 'grad64/' + n
        if key not in g:
            assert p.grad is None, n
            continue
        ours[n] = p.grad.cpu().numpy().astype(np.float64)
        d = digest(ref['grads'][n])                                     # the oracle IS the reference's fp64 capture
        body = slice(None) if p.numel() <= 1024 else slice(0, -3)
        assert np.abs(d - g[key])[body].max() <= 1e-9 * max(1.0, np.abs(g[key][body]).max()), n
    from decision_match import hip_relu_flips, is_stem_decision
    known = hip_relu_flips(ref['tape'], taps, log=log, tag=os.path.basename(path))
    matched, flips, ncand = _dmg(ref, ours, os.path.basename(path), log=log, known=known, only=is_stem_decision)
    if strict:
        assert not [f for f in flips if not f[0].endswith('.maxpool')], flips
    assert len(flips) <= 12, flips
    worst, bad = 0.0, []
    for n in ours:
        abs_err = float(np.abs(ours[n] - matched[n]).max())
        rl2 = rel_l2(ours[n], matched[n])
        rl2_exact = rel_l2(ours[n], ref['grads'][n])
        body = slice(None) if ours[n].size <= 1024 else slice(0, -3)
        ref32_rl2 = rel_l2(g['grad32/' + n][body], g['grad64/' + n][body])
        worst = max(worst, rl2)
        log('   grad %-60s max abs err %.3e rel-l2 %.3e (no matching: %.3e; reference fp32 vs fp64: %.3e)' %
            (n, abs_err, rl2, rl2_exact, ref32_rl2))
        scale = max(1.0, float(np.abs(matched[n]).max()))
        if not (rl2 <= 1e-4 or abs_err <= 1e-4 * scale):
            bad.append((n, abs_err, rl2))
    log('   worst grad rel-l2 %.3e with %d flips of %d candidates' % (worst, len(flips), ncand))
    assert not bad, bad


def build_head(M, head, backbone, seed, first_pool_type='max', shift=0.0):
    bb = M.resnet18(first_pool_type=first_pool_type) if backbone == 'resnet18' else M.densenet18(drop_rate=0.0)
    model = {'to_mean': lambda: M.CNNLinearToMean(bb), 'compr_to_rf': lambda: M.CNNLinearComprToRF(bb),
             'single_breath': lambda: M.CNNSingleBreathLinearNetwork(bb),
             'double_linear': lambda: M.CNNDoubleLinearNetwork(bb, 20, 0),
             'lstm': lambda: M.CNNLSTMNetwork(bb, 0, False, 16)}[head]()
    sd = {k: torch.from_numpy(v) for k, v in seeded_params(backbone, seed, bn_bias_shift=shift, head=head).items()}
    missing = model.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys
    return model.cuda().train()


@pytest.mark.parametrize('path', HEAD_GOLD, ids=[os.path.basename(p)[:-4] for p in HEAD_GOLD])
def test_sibling_heads_match_reference_golden(M, path):
    """CNNLinearToMean / CNNLinearComprToRF / CNNSingleBreathLinearNetwork / CNNDoubleLinearNetwork
    (reference models/torch_cnn_linear_network.py:7-89) and CNNLSTMNetwork (torch_cnn_lstm_combo.py:6-50; goldens from
    the reference classes,
    oracle/make_golden_heads.py): logits 1e-4 absolute; gradients by the suite's one yardstick -- the oracle (pinned to
    the golden's digests here at 1e-9) under the decisions this run took, 1e-4, no flips on the '_active' goldens.  The
    median head keeps one decision the tape does not carry (WHICH breath is the median: fp32 / fp64 ties), so its
    parameters may fall back to rel-l2 5e-2 -- the only place left where that bound is used.  One trainer step runs."""
    from deepards_amd.functional import bce_with_logits
    from deepards_amd.train import HotPathTrainer, _loss_operands
    g = _gold(path)
    backbone, head, shift = str(g['backbone']), str(g['head']), float(g['bn_bias_shift'])
    strict = shift > 0
    model = build_head(M, head, backbone, int(g['seed']), str(g['first_pool_type']), shift)
    x = torch.from_numpy(g['x']).cuda()
    t = torch.from_numpy(g['target']).cuda()
    with tapped() as taps:
        out = model(x, None)
    if head == 'lstm':                                   # (logits, (hx, cx)), zero initial state
        out, (hx, cx) = out
        assert np.abs(hx.detach().cpu().numpy() - g['hx64']).max() < 1e-5
        assert np.abs(cx.detach().cpu().numpy() - g['cx64']).max() < 1e-5
    assert tuple(out.shape) == g['logits64'].shape
    loss = bce_with_logits(*[o.view(-1, 2) if i == 0 else o for i, o in enumerate(_loss_operands(out, t))])
    loss.backward()
    err = np.abs(out.detach().cpu().numpy().astype(np.float64) - g['logits64']).max()
    log(os.path.basename(path), 'logits max|hip-ref64| %.3e loss %.8f vs %.8f' % (err, float(loss), float(g['loss64'])))
    assert err < 1e-4 and abs(float(loss) - float(g['loss64'])) < 1e-5
    params64 = {k: v.astype(np.float64) for k, v in seeded_params(backbone, int(g['seed']), bn_bias_shift=shift,
                                                                   head=head).items()}
    ref = np_ref.cnn_linear_forward_backward(params64, g['x'].astype(np.float64), g['target'].astype(np.float64),
                                             backbone=backbone, first_pool_type=str(g['first_pool_type']), head=head)
    ours = {}
    for n, p in model.named_parameters():
        key = 'grad64/' + n
        if key not in g:
            assert p.grad is None, n
            continue
        ours[n] = p.grad.cpu().numpy().astype(np.float64)
        body = slice(None) if p.numel() <= 1024 else slice(0, -3)       # the oracle IS the reference's fp64 capture
        assert np.abs(digest(ref['grads'][n]) - g[key])[body].max() <= 1e-9 * max(1.0, np.abs(g[key][body]).max()), n
    median_slack = (lambda n: rel_l2(ours[n], ref['grads'][n]) <= 5e-2) if head == 'compr_to_rf' else None
    assert_gradients_match(ref, ours, os.path.basename(path), strict=strict and head != 'compr_to_rf', allow=median_slack, taps=taps)
    if strict and backbone == 'resnet18':                 # the trainer drives every head (per-breath loss included)
        tr = HotPathTrainer(model, use_graph=True)
        l = [float(tr.train_step(x, t)) for _ in range(3)]
        assert all(np.isfinite(l)) and abs(l[0] - float(g['loss64'])) < 1e-4
        lt, lg, pred = tr.test_step(x, t)
        assert pred.shape == tuple(g['logits64'].shape[:-1])


@pytest.mark.parametrize('path', BB_GOLD, ids=[os.path.basename(p)[:-4] for p in BB_GOLD])
def test_other_backbones_match_reference_golden(M, path):
    """resnet34 and densenet121 (the other BasicBlock / growth-32 base networks of the reference) on the same kernels:
    logits 1e-4, loss, gradients 1e-4 under matched decisions (the oracle pinned to the golden's digests), a few trainer
    steps."""
    from deepards_amd.functional import bce_with_logits
    from deepards_amd.train import HotPathTrainer
    g = _gold(path)
    name = str(g['backbone'])
    bb = M.base_networks[name]() if name.startswith('resnet') else M.base_networks[name](drop_rate=0.0)
    assert bb.network_name == name
    model = M.CNNLinearNetwork(bb, 20, 0)
    sd = {k: torch.from_numpy(v) for k, v in seeded_params(name, int(g['seed'])).items()}
    assert not model.load_state_dict(sd, strict=False).unexpected_keys
    model = model.cuda().train()
    x, t = torch.from_numpy(g['x']).cuda(), torch.from_numpy(g['target']).cuda()
    with tapped() as taps:
        out = model(x, None)
    loss = bce_with_logits(out, t)
    loss.backward()
    err = np.abs(out.detach().cpu().numpy() - g['logits64']).max()
    log(name, 'logits err %.3e loss %.7f vs %.7f' % (err, float(loss), float(g['loss64'])))
    assert err < 1e-4 and abs(float(loss) - float(g['loss64'])) < 1e-5
    ref = np_ref.cnn_linear_forward_backward({k: v.astype(np.float64) for k, v in seeded_params(name, int(g['seed'])).items()},
                                             g['x'].astype(np.float64), g['target'].astype(np.float64), backbone=name)
    ours = {}
    for n, p in model.named_parameters():
        key = 'grad64/' + n
        if key not in g:
            assert p.grad is None, n
            continue
        ours[n] = p.grad.cpu().numpy().astype(np.float64)
        body = slice(None) if p.numel() <= 1024 else slice(0, -3)
        assert np.abs(digest(ref['grads'][n], 24) - g[key])[body].max() <= 1e-9 * max(1.0, np.abs(g[key][body]).max()), n
    assert_gradients_match(ref, ours, name, max_flips=96, taps=taps)       # up to 7x the layers of the 18s: more decisions near zero
    tr = HotPathTrainer(model, use_graph=True)
    assert all(np.isfinite(float(tr.train_step(x, t))) for _ in range(3))


@pytest.mark.parametrize('path', OPT_GOLD, ids=[os.path.basename(p)[:-4] for p in OPT_GOLD])
def test_constructor_options_match_reference_golden(M, path):
    """The constructor options of SURVEY 8b that round 2 refused: densenet18(with_fft / only_fft / fft_real_only)
    (models/densenet.py:109-115: conv0 on 3 / 2 / 2 / 1 input channels, stem kernels da_stem_conv_*_g) and
    resnet18(double_conv_first=True) (models/resnet.py:90-96,142-149: conv1_alt -> bn1 -> conv2 k7 s2 -> bn2).  Goldens from
    the reference classes (oracle/make_golden_options.py): logits 1e-4, loss 1e-5, every gradient 1e-4 under matched
    decisions, the live / dead parameter sets, a 3-step SGD trajectory through the captured step."""
    from deepards_amd.functional import bce_with_logits
    from deepards_amd.train import HotPathTrainer
    g = _gold(path)
    backbone, shift, in_ch = str(g['backbone']), float(g['bn_bias_shift']), int(g['in_ch'])
    double = bool(g['opt_double_conv_first']) if 'opt_double_conv_first' in g else False

    def make():
        if backbone == 'resnet18':
            bb = M.resnet18(first_pool_type=str(g['first_pool_type']), double_conv_first=double)
        else:
            bb = M.densenet18(drop_rate=0.0, with_fft=bool(g.get('opt_with_fft', 0)), only_fft=bool(g.get('opt_only_fft', 0)),
                              fft_real_only=bool(g.get('opt_fft_real_only', 0)))
        model = M.CNNLinearNetwork(bb, 20, 0)
        sd = {k: torch.from_numpy(v) for k, v in seeded_params(backbone, int(g['seed']), bn_bias_shift=shift, in_ch=in_ch).items()}
        assert not model.load_state_dict(sd, strict=False).unexpected_keys
        return model.cuda().train()
    model = make()
    if backbone != 'resnet18':
        assert model.breath_block.features.conv0.in_channels == in_ch == g['x'].shape[2]
    x, t = torch.from_numpy(g['x']).cuda(), torch.from_numpy(g['target']).cuda()
    with tapped() as taps:
        out = model(x, None)
    loss = bce_with_logits(out, t)
    loss.backward()
    err = np.abs(out.detach().cpu().numpy().astype(np.float64) - g['logits64']).max()
    log(os.path.basename(path), 'logits max|hip-ref64| %.3e loss %.8f vs %.8f' % (err, float(loss), float(g['loss64'])))
    assert err < 1e-4 and abs(float(loss) - float(g['loss64'])) < 1e-5
    params64 = {k: v.astype(np.float64) for k, v in seeded_params(backbone, int(g['seed']), bn_bias_shift=shift, in_ch=in_ch).items()}
    ref = np_ref.cnn_linear_forward_backward(params64, g['x'].astype(np.float64), g['target'].astype(np.float64),
                                             backbone=backbone, first_pool_type=str(g['first_pool_type']),
                                             double_conv_first=double)
    ours = {}
    for n, p in model.named_parameters():
        if 'grad64/' + n not in g:
            assert p.grad is None, n                        # the dead parameters of this configuration
            continue
        ours[n] = p.grad.cpu().numpy().astype(np.float64)
        body = slice(None) if p.numel() <= 1024 else slice(0, -3)
        assert np.abs(digest(ref['grads'][n]) - g['grad64/' + n])[body].max() <= 1e-9 * max(1.0, np.abs(g['grad64/' + n][body]).max()), n
    if double:
        assert 'breath_block.conv1.weight' not in ours and 'breath_block.conv2.weight' in ours
    assert_gradients_match(ref, ours, os.path.basename(path), strict=shift > 0, taps=taps)
    # three SGD-Nesterov steps with the clamp, through the captured step (steps 2, 3 replay the graph)
    tr = HotPathTrainer(make(), optimizer='sgd', use_graph=True)
    losses = [float(tr.train_step(x, t)) for _ in range(3)]
    assert np.abs(np.array(losses) - g['sgd_losses64']).max() < 2e-5 * max(1.0, np.abs(g['sgd_losses64']).max())
    p_tol = 2e-5 if shift > 0 else 1.5e-4                 # as test_trainer_trajectory_matches_reference (flips move clamped entries)
    for n, p in tr.model.named_parameters():
        key = 'sgd_p64/' + n
        if key not in g:
            continue
        body = slice(None) if p.numel() <= 1024 else slice(0, -3)
        assert np.abs(digest(p.detach().cpu().numpy()) - g[key])[body].max() < p_tol, n


def test_fft_channels_through_the_tile_store_and_cli(M, tmp_path):
    """--with-fft end to end: the fixture dataset gets its spectrum channels at ingest (tiles.perform_fft), per-channel
    scaling factors per fold, the device gather normalises every channel with its own factors (da_gather_normalize_ch,
    bit-identical to the float64 host expression of dataset.py:1379 + .float()), and densenet18(with_fft) trains on it
    through the CLI."""
    from deepards_amd import ingest, train_ards_detector as T
    from deepards_amd.tiles import perform_fft, scaling_factors_for_indices
    gold = os.path.join(os.path.dirname(__file__), 'golden', 'test_dataset.npz')
    ds = ingest.load_npz(gold).with_fft(add_fft=True)
    assert ds.windows.shape == (20, 20, 3, 224) and ds.scaling_factors == {}
    store = ds.to_store('cuda')
    mu, std = scaling_factors_for_indices(ds.windows)
    assert store.mu == tuple(mu.tolist()) and store.std == tuple(std.tolist())
    xb, tb = store.batch([3, 0, 19])
    want = ((ds.windows[[3, 0, 19]] - mu.reshape(1, 1, 3, 1)) / std.reshape(1, 1, 3, 1)).astype(np.float32)
    assert np.array_equal(xb.cpu().numpy(), want)
    cls, res = T.main(['--cuda-no-dp', '--train-from-pickle', gold, '--kfolds', '2', '-e', '1', '-b', '4', '--with-fft',
                       '--base-network', 'densenet18', '--seed', '3'])
    assert cls.model.breath_block.features.conv0.in_channels == 3
    assert all(np.isfinite(v).all() for v in (res.patient_results[(1, 1)]['pred_frac'],))
    assert np.isfinite(res.patient_results[(0, 1)]['mean_loss'])


def test_window_independence_and_breath_block_call(M):
    """model(x)[i] == model(x[i:i+1])[0] (SURVEY finding 3) and breath_block(x[i]) as the reference calls it."""
    model = build(M, 'resnet18', 0)
    x, _ = seeded_batch(4, 20, 3)
    xt = torch.from_numpy(x).cuda()
    with torch.no_grad():
        full = model(xt, None)
        for i in range(4):
            one = model(xt[i:i + 1], None)
            assert torch.equal(full[i], one[0])
        feat = model.breath_block(xt[0])                    # (20, 512) exactly like the reference call
        g = _gold([p for p in GOLD if 'resnet18_b2_randn' in p][0])
    assert feat.shape == (20, 512)
    model2 = build(M, 'resnet18', 0)
    with torch.no_grad():
        f0 = model2.breath_block(torch.from_numpy(g['x']).cuda()[0]).cpu().numpy()
    assert np.abs(digest(f0, 256) - g['feat0_64'])[:-3].max() < 1e-4


def test_sequence_length_check(M):
    model = build(M, 'densenet18', 0)
    with pytest.raises(Exception, match='sequence length of 224'):
        model(torch.zeros(2, 20, 1, 200, device='cuda'), None)
    with pytest.raises(IndexError, match='index 0 is out of bounds'):       # the reference's x[0] on an empty batch
        model(torch.zeros(0, 20, 1, 224, device='cuda'), None)
    for nb in (1, 3):                                                       # degenerate sub-batch counts still run
        m = M.CNNLinearNetwork(M.resnet18(), nb, 0).cuda().train()
        out = m(torch.randn(1, nb, 1, 224, device='cuda'), None)
        out.sum().backward()
        assert tuple(out.shape) == (1, 2) and all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)
    cpu_model = M.CNNLinearNetwork(M.resnet18(), 20, 0)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        cpu_model(torch.zeros(2, 20, 1, 224), None)


def test_resnet_running_stats(M):
    g = _gold([p for p in GOLD if 'resnet18_b2_randn' in p][0])
    model = build(M, 'resnet18', 0)
    x = torch.from_numpy(g['x']).cuda()
    with torch.no_grad():
        model(x, None)
        model.breath_block(x[0])
    bb = model.breath_block
    for short, bn in (('bn1', bb.bn1), ('layer4.1.bn2', bb.layer4[1].bn2)):
        assert np.abs(bn.running_mean.cpu().numpy() - g['rm64/' + short]).max() < 1e-5
        assert np.abs(bn.running_var.cpu().numpy() - g['rv64/' + short]).max() < 1e-5
        assert int(bn.num_batches_tracked) == 3


@pytest.mark.parametrize('tag,opt,use_graph', [('resnet18_b2_randn', 'sgd', False), ('resnet18_b2_randn', 'sgd', True),
                                               ('densenet18_b2_randn', 'sgd', True), ('densenet18_b2_randn', 'adam', False),
                                               ('resnet18_b2_randn', 'adam', False), ('resnet18_b2_active', 'sgd', True),
                                               ('densenet18_b2_active', 'sgd', True), ('resnet18_b2_active', 'adam', False),
                                               ('resnet18_b2_active', 'adam', True), ('densenet18_b2_randn', 'adam', True)])
def test_trainer_trajectory_matches_reference(M, tag, opt, use_graph):
    """3 optimiser steps (clamp +-0.01, SGD-Nesterov wd 1e-4 / Adam): losses and parameters against the
    reference trajectory.  '*_active' goldens have no activation decision to flip -> strict bounds;
    the others see ReLU flips (see the gradient test) -> bounds derived from the update rule:
    a clamped gradient entry of opposite sign moves a weight by lr*(1+momentum)*0.02 = 3.8e-5 per SGD
    step; Adam turns the sign of ANY near-zero gradient entry into +-lr per step."""
    from deepards_amd.train import HotPathTrainer
    g = _gold([p for p in GOLD if tag in p][0])
    backbone, shift = str(g['backbone']), float(g['bn_bias_shift'])
    strict = shift > 0
    model = build(M, backbone, int(g['seed']), str(g['first_pool_type']), shift=shift)
    tr = HotPathTrainer(model, optimizer=opt, use_graph=use_graph)
    x = torch.from_numpy(g['x']).cuda()
    t = torch.from_numpy(g['target']).cuda()
    losses = [float(tr.train_step(x, t)) for _ in range(3)]
    ref = g['%s_losses64' % opt]
    log(tag, opt, 'graph' if use_graph else 'eager', 'losses', losses, 'ref', ref.tolist())
    loss_tol = 2e-5 if (strict or opt == 'sgd') else 2e-3
    assert np.abs(np.array(losses) - ref).max() < loss_tol * max(1.0, np.abs(ref).max())
    p_tol = {('sgd', True): 2e-5, ('sgd', False): 1.5e-4, ('adam', True): 6.5e-3, ('adam', False): 6.5e-3}[(opt, strict)]
    worst = 0.0
    init = seeded_params(backbone, int(g['seed']), bn_bias_shift=shift)
    for n, p in model.named_parameters():
        key = '%s_p64/%s' % (opt, n)
        if key not in g:
            assert n in DEAD_RESNET_PARAMS
            # the reference's optimiser skips parameters without a gradient: untouched here too
            assert np.array_equal(p.detach().cpu().numpy(), init[n])
            continue
        d = digest(p.detach().cpu().numpy())
        body = slice(None) if p.numel() <= 1024 else slice(0, -3)
        e = np.abs(d - g[key])[body].max()
        worst = max(worst, e)
        assert e < p_tol, (n, e)
    log('   worst param abs err %.3e' % worst)
    with torch.no_grad():
        after = model(x, None).cpu().numpy()
    if opt == 'sgd':
        ref_after = g['sgd_logits_after64']
        assert np.abs(after - ref_after).max() < (1e-4 if strict else 1e-3) * max(1.0, np.abs(ref_after).max())


def test_fresh_inputs_vs_numpy_oracle(M):
    """Seeded inputs the goldens do not cover (B=3, flow-like waveform) straight against the oracle."""
    for backbone in ('resnet18', 'densenet18'):
        model = build(M, backbone, 7)
        x, t = seeded_batch(3, 20, 11, 'flow')
        params = {k: v.astype(np.float64) for k, v in seeded_params(backbone, 7).items()}
        ref = np_ref.cnn_linear_forward_backward(params, x.astype(np.float64), t.astype(np.float64), backbone=backbone)
        from deepards_amd.functional import bce_with_logits
        with tapped() as taps:
            out = model(torch.from_numpy(x).cuda(), None)
        bce_with_logits(out, torch.from_numpy(t).cuda()).backward()
        err = np.abs(out.detach().cpu().numpy() - ref['logits']).max()
        log(backbone, 'fresh B=3 flow: logits err %.3e' % err)
        assert err < 1e-4
        assert_gradients_match(ref, grads64(model, ref), backbone + ' fresh B=3 flow', taps=taps)


@pytest.mark.parametrize('nb', [40, 8])
def test_other_sub_batch_counts_vs_numpy_oracle(M, nb):
    """n_sub_batches other than 20 (BASELINE config C5 runs nb40; BatchNorm windows of 40 rows go through the 16-channel
    single-pass kernels or the two-stage path, the head is F*NB wide): logits and loss against the oracle."""
    from deepards_amd.functional import bce_with_logits
    from deepards_amd.train import HotPathTrainer
    for backbone in ('resnet18', 'densenet18'):
        bb = M.resnet18() if backbone == 'resnet18' else M.densenet18(drop_rate=0.0)
        model = M.CNNLinearNetwork(bb, nb, 0)
        p32 = seeded_params(backbone, 9, n_sub_batches=nb)
        assert not model.load_state_dict({k: torch.from_numpy(v) for k, v in p32.items()}, strict=False).unexpected_keys
        model = model.cuda().train()
        x, t = seeded_batch(3, nb, 5, 'flow')
        ref = np_ref.cnn_linear_forward_backward({k: v.astype(np.float64) for k, v in p32.items()}, x.astype(np.float64),
                                                 t.astype(np.float64), backbone=backbone, n_sub_batches=nb, need_grads=False)
        xt, tt = torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()
        out = model(xt, None)
        loss = bce_with_logits(out, tt)
        loss.backward()
        err = np.abs(out.detach().cpu().numpy() - ref['logits']).max()
        log(backbone, 'nb=%d: logits err %.3e loss %.7f vs %.7f' % (nb, err, float(loss), ref['loss']))
        assert err < 1e-4 and abs(float(loss) - ref['loss']) < 1e-5
        tr = HotPathTrainer(model, use_graph=True)
        assert all(np.isfinite(float(tr.train_step(xt, tt))) for _ in range(3))


@pytest.mark.parametrize('backbone', ['resnet18', 'densenet18'])
def test_long_sequences_breath_block_vs_numpy_oracle(M, backbone):
    """BASELINE config C5's tile shape (nb 40, seq_len 512): ``breath_block((40, 1, 512))`` gives a 16-long final map,
    AvgPool1d(7, 1) leaves 10 positions and ``view(N, -1)`` flattens channel-major (resnet.py:159-160,
    densenet.py:183-184) -> (40, F * 10).  Features (1e-4) and every weight gradient (1e-4 under matched decisions)
    against the oracle."""
    nb, L = 40, 512
    bb = M.resnet18() if backbone == 'resnet18' else M.densenet18(drop_rate=0.0)
    model = M.CNNLinearNetwork(bb, nb, 0)
    p32 = seeded_params(backbone, 4, n_sub_batches=nb)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in p32.items()}, strict=False)
    model = model.cuda().train()
    rng = np.random.RandomState(12)
    x = rng.randn(nb, 1, L).astype(np.float32)
    F = bb.n_out_filters
    w = rng.randn(nb, F * 10) / np.sqrt(nb * F * 10)
    fr = feature_reference({k: v.astype(np.float64) for k, v in p32.items()}, nb, x.astype(np.float64), w, backbone)
    ref = fr['feat']
    assert ref.shape == (nb, F * 10)
    with tapped() as taps:
        feat = model.breath_block(torch.from_numpy(x).cuda())
    assert tuple(feat.shape) == (nb, F * 10)
    err = np.abs(feat.detach().cpu().numpy() - ref).max()
    log(backbone, 'nb=40 L=512 features err %.3e' % err)
    assert err < 1e-4
    (feat * torch.from_numpy(w.astype(np.float32)).cuda()).sum().backward()
    assert_gradients_match(fr, grads64(model, fr), backbone + ' nb=40 L=512', max_flips=24, taps=taps)    # 4.6x the elements of a (20, 224) tile


def test_long_sequences_bf16_convs(M):
    """BASELINE config C5's tile shape AND arithmetic (resnet18, nb 40, seq_len 512, bf16 operands): features against the
    oracle with the same operand rounding (5e-3, see test_bf16_conv_arithmetic_resnet18) and finite gradients."""
    from deepards_amd import functional as F_
    nb, L = 40, 512
    p32 = seeded_params('resnet18', 4, n_sub_batches=nb)
    rng = np.random.RandomState(13)
    x = rng.randn(nb, 1, L).astype(np.float32)
    t = np_ref._Tape({k: v.astype(np.float64) for k, v in p32.items()}, nb)
    t.bf16_convs = True
    ref, _ = np_ref.resnet18_features(t, x.astype(np.float64))
    exact, _ = np_ref.resnet18_features(np_ref._Tape({k: v.astype(np.float64) for k, v in p32.items()}, nb),
                                        x.astype(np.float64))
    try:
        F_.set_conv_dtype('bf16')
        model = M.CNNLinearNetwork(M.resnet18(), nb, 0)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in p32.items()}, strict=False)
        model = model.cuda().train()
        feat = model.breath_block(torch.from_numpy(x).cuda())
        got = feat.detach().cpu().numpy()
        err, drift = np.abs(got - ref).max(), np.abs(got - exact).max()
        r2, d2 = rel_l2(got, ref), rel_l2(got, exact)
        log('resnet18 bf16 nb=40 L=512 features: max err %.3e / rel L2 %.3e vs same-rounding oracle, %.3e / %.3e vs '
            'exact (scale %.2f)' % (err, r2, drift, d2, np.abs(ref).max()))
        # 204 800 feature values behind 17 rounding layers: the rounding-flip noise is bounded in rel L2 (measured 6e-3
        # against the same-rounding oracle, 1.4e-2 against the exact one) and 2 % of the scale at the worst element
        assert tuple(feat.shape) == (nb, 5120) and r2 < 1.5e-2 and r2 < 0.8 * d2 and err < 2e-2 * max(1.0, np.abs(ref).max())
        feat.square().mean().backward()
        assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
    finally:
        F_.set_conv_dtype('f32')


def test_densenet_dropout_active_and_scaled(M):
    """drop_rate 0.2 is active in train mode (and the reference never leaves train mode): outputs
    differ run to run, and the test step still works under no_grad."""
    from deepards_amd.train import HotPathTrainer
    model = build(M, 'densenet18', 0, drop_rate=0.2)
    x, t = seeded_batch(2, 20, 0)
    xt, tt = torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()
    with torch.no_grad():
        a = model(xt, None)
        b = model(xt, None)
    assert not torch.equal(a, b)
    tr = HotPathTrainer(model, use_graph=True)
    l = [float(tr.train_step(xt, tt)) for _ in range(4)]
    assert all(np.isfinite(l))
    loss, logits, pred = tr.test_step(xt, tt)
    assert pred.shape == (2,) and np.isfinite(float(loss))


def test_reference_pickled_fixture_windows(M):
    """BASELINE config C1's inputs: the 20 windows of the reference's own tests/test_dataset.pkl (extracted
    without unpickling, oracle/extract_fixture.py), normalised like ARDSRawDataset.__getitem__ (float64
    (x-mu)/std, then .float() as train_ards_detector.py:150-152): logits of all 20 windows and the loss against
    the numpy oracle within 1e-4, for both backbones; and the argmax predictions of the test epoch."""
    from deepards_amd.train import HotPathTrainer
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'test_dataset_windows.npz'))
    x64 = (z['x'] - float(z['mu'])) / float(z['std'])
    tgt = z['target'].astype(np.float64)
    for backbone in ('resnet18', 'densenet18'):
        params = {k: v.astype(np.float64) for k, v in seeded_params(backbone, 3).items()}
        ref = np_ref.cnn_linear_forward_backward(params, x64.astype(np.float32).astype(np.float64), tgt, backbone=backbone,
                                                 need_grads=False)
        model = build(M, backbone, 3)
        tr = HotPathTrainer(model, use_graph=False)
        xt = torch.from_numpy(x64).float().cuda()
        tt = torch.from_numpy(tgt).float().cuda()
        loss, logits, pred = tr.test_step(xt, tt)
        err = np.abs(logits.cpu().numpy() - ref['logits']).max()
        log(backbone, 'reference fixture (20 windows): logits err %.3e loss %.7f vs %.7f' % (err, float(loss), ref['loss']))
        assert err < 1e-4 and abs(float(loss) - ref['loss']) < 1e-5
        sure = np.abs(ref['logits'][:, 0] - ref['logits'][:, 1]) > 1e-3
        assert np.array_equal(pred.cpu().numpy()[sure], ref['logits'].argmax(-1)[sure])


def test_device_tile_store_matches_reference_getitem():
    """Gather + normalise kernel vs the reference's host path: float64 (x-mu)/std then .float() -- bit-exact;
    k-fold index map and odd-batch clipping as in dataset.py:765-772 / train_ards_detector.py:482-494."""
    from deepards_amd.data import DeviceTileStore
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'test_dataset_windows.npz'))
    store = DeviceTileStore(z['x'], z['target'], float(z['mu']), float(z['std']))
    ref = ((z['x'] - float(z['mu'])) / float(z['std'])).astype(np.float32)
    x, t = store.batch([3, 0, 19, 7])
    assert np.array_equal(x.cpu().numpy(), ref[[3, 0, 19, 7]])
    assert np.array_equal(t.cpu().numpy(), z['target'][[3, 0, 19, 7]])
    store.set_kfold_indexes([10, 11, 12, 13, 14])
    x, t = store.batch([4, 0])
    assert np.array_equal(x.cpu().numpy(), ref[[14, 10]])
    sizes = [len(i) for i, _, _ in store.epoch(2, shuffle=True, generator=torch.Generator().manual_seed(0))]
    assert sizes == [2, 2] and len(store) == 5          # the odd last batch of 1 is clipped away


def test_device_tile_store_from_tiled_breaths_with_derived_scaling():
    """Ingest path end to end: breaths -> (20, 1, 224) windows (deepards_amd.tiles, dataset.py:1021-1081) -> store whose
    factors come from the train fold (dataset.py:627-649) -> batches == float64 (x - mu) / std cast to float32."""
    from deepards_amd.data import DeviceTileStore
    from deepards_amd.tiles import tile_patient, scaling_factors_for_indices
    rng = np.random.default_rng(3)
    wins = np.concatenate([tile_patient([(rng.standard_normal(int(rng.integers(60, 220))) * 25 + 2, i)
                                         for i in range(150)])[0] for _ in range(3)])
    tg = np.eye(2, dtype=np.float32)[rng.integers(0, 2, len(wins))]
    train = list(range(0, len(wins), 2))
    store = DeviceTileStore.with_derived_scaling(wins, tg, train)
    mu, std = scaling_factors_for_indices(wins, train)
    assert store.mu == mu[0] and store.std == std[0]
    x, t = store.batch([1, 0, len(wins) - 1])
    ref = ((wins - mu[0]) / std[0]).astype(np.float32)
    assert np.array_equal(x.cpu().numpy(), ref[[1, 0, len(wins) - 1]])
    assert np.array_equal(t.cpu().numpy(), tg[[1, 0, len(wins) - 1]])


def test_test_step_graph_replay_matches_eager(M):
    """test_step captured per batch shape (packs once per step, batched small kernels) == the eager form, bit for
    bit, including the BatchNorm running statistics a train-mode forward moves (run_test_epoch never calls eval())."""
    from deepards_amd.train import HotPathTrainer
    x, t = seeded_batch(4, 20, 5)
    xt, tt = torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()
    ma, mb = build(M, 'resnet18', 1), build(M, 'resnet18', 1)
    ta, tb = HotPathTrainer(ma, use_graph=True), HotPathTrainer(mb, use_graph=False)
    for rep in range(3):
        la, ga, pa = ta.test_step(xt + rep, tt)
        lb, gb, pb = tb.test_step(xt + rep, tt)
        assert torch.equal(ga, gb) and torch.equal(pa, pb) and torch.equal(la, lb)
    sa, sb = ma.state_dict(), mb.state_dict()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    assert int(sa['breath_block.bn1.num_batches_tracked']) == 3 * 4
    la, ga, pa = ta.test_step(xt[:2], tt[:2])            # another shape: second graph
    lb, gb, pb = tb.test_step(xt[:2], tt[:2])
    assert torch.equal(ga, gb)


def test_gradcam_surface_on_densenet(M):
    """The explainer surface of SURVEY 8f row 4: gradcam.py:38-108 calls ``model.breath_block.features(x)`` (norm5
    output, (N, C, L), no ReLU), hooks its gradient, then relu -> ``breath_block.avgpool`` -> view(-1) ->
    ``linear_final`` and backpropagates the one-hot class score.  Same calls on this build vs the reference modules
    (oracle/make_golden_gradcam.py); a whole-module torch.save / torch.load round trip keeps working too."""
    import io
    import torch.nn.functional as Fn
    g = _gold(os.path.join(os.path.dirname(__file__), 'golden', 'gradcam_densenet18.npz'))
    model = build(M, 'densenet18', int(g['seed']))
    buf = io.BytesIO()
    torch.save(model, buf)                                   # train_ards_detector.py:364,374 pickles whole modules
    buf.seek(0)
    model = torch.load(buf, weights_only=False)
    x = torch.from_numpy(g['x']).float().cuda()
    grads = {}
    conv = model.breath_block.features(x)
    conv.register_hook(lambda gr: grads.__setitem__('g', gr))
    h = Fn.relu(conv)
    h = model.breath_block.avgpool(h).view(-1)
    out = model.linear_final(h).unsqueeze(0)
    target = int(out.argmax())
    assert target == int(g['target64'])
    one_hot = torch.zeros((1, 2), device='cuda')
    one_hot[0, target] = 1
    model.zero_grad()
    torch.sum(one_hot * out).backward()
    assert tuple(conv.shape) == g['conv64'].shape == (20, 128, 7)
    e_conv = np.abs(conv.detach().cpu().numpy() - g['conv64']).max()
    e_out = np.abs(out.detach().cpu().numpy() - g['out64']).max()
    gr = grads['g'].cpu().numpy()
    e_grad = np.abs(gr - g['grad64']).max()
    log('gradcam surface: conv err %.3e out err %.3e guided-gradient err %.3e (scale %.3e)' %
        (e_conv, e_out, e_grad, np.abs(g['grad64']).max()))
    assert e_conv < 1e-4 * max(1.0, np.abs(g['conv64']).max()) and e_out < 1e-4
    assert e_grad < 1e-4 * np.abs(g['grad64']).max()
    # the pooled path the training step uses is the same function
    feat = model.breath_block(x)
    ref = model.breath_block.avgpool(Fn.relu(model.breath_block.features(x))).view(20, -1)
    assert torch.allclose(feat, ref, atol=1e-6)


def test_in_place_batches_skip_the_input_copies(M):
    """DeviceTileStore.batch(out=trainer.static_batch()) + train_step on those buffers == train_step on fresh tensors."""
    from deepards_amd.data import DeviceTileStore
    from deepards_amd.train import HotPathTrainer
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'test_dataset_windows.npz'))
    store = DeviceTileStore(z['x'], z['target'], float(z['mu']), float(z['std']))
    ma, mb = build(M, 'resnet18', 2), build(M, 'resnet18', 2)
    ta, tb = HotPathTrainer(ma, use_graph=True), HotPathTrainer(mb, use_graph=True)
    order = [np.arange(0, 8), np.arange(8, 16), np.arange(4, 12), np.arange(12, 20)]
    for n, idx in enumerate(order):
        x, t = store.batch(idx)
        la = float(ta.train_step(x, t))
        if tb.static_batch() is None:
            lb = float(tb.train_step(x, t))
        else:
            xs, ts = store.batch(idx, out=tb.static_batch())
            assert xs is tb.static_batch()[0] and ts is tb.static_batch()[1]
            lb = float(tb.train_step(xs, ts))
        assert la == lb, (n, la, lb)
    for (k, a), (_, b) in zip(ma.state_dict().items(), mb.state_dict().items()):
        assert torch.equal(a, b), k


def test_train_epoch_from_store(M):
    """run_train_epoch_from_store == the same batches pushed through train_step by hand (in-place batch buffers after
    the capture, odd last batch clipped like clip_odd_batch_sizes)."""
    from deepards_amd.data import DeviceTileStore
    from deepards_amd.train import HotPathTrainer, run_train_epoch_from_store
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'test_dataset_windows.npz'))
    store = DeviceTileStore(z['x'], z['target'], float(z['mu']), float(z['std']))
    ma, mb = build(M, 'resnet18', 2), build(M, 'resnet18', 2)
    ta, tb = HotPathTrainer(ma, use_graph=True), HotPathTrainer(mb, use_graph=True)
    la = []
    for ep in range(2):
        la += [float(l) for l in run_train_epoch_from_store(ta, store, batch_size=6, shuffle=False)]
    lb = []
    for ep in range(2):
        for s0 in range(0, 20, 6):
            idx = np.arange(s0, min(20, s0 + 6))
            if len(idx) % 2:
                idx = idx[:-1]
            x, t = store.batch(idx)
            lb.append(float(tb.train_step(x, t)))
    assert la == lb and len(la) == 8
    for (k, a), (_, b) in zip(ma.state_dict().items(), mb.state_dict().items()):
        assert torch.equal(a, b), k


def test_test_epoch_votes_on_device(M):
    """Window argmax + per-patient vote table (metrics.py:572-604) computed on the device vs numpy."""
    from deepards_amd.data import DeviceTileStore
    from deepards_amd.train import HotPathTrainer, run_test_epoch
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'test_dataset_windows.npz'))
    store = DeviceTileStore(z['x'], z['target'], float(z['mu']), float(z['std']))
    slot = np.arange(20) % 6                      # 6 synthetic patients (ids are not exported from the fixture)
    model = build(M, 'densenet18', 3)
    tr = HotPathTrainer(model, use_graph=False)
    res = run_test_epoch(tr, store, slot, batch_size=8)
    with torch.no_grad():
        x, _ = store.batch(np.arange(20))
        logits = torch.cat([model(x[i:i + 8], None) for i in range(0, 20, 8)]).cpu().numpy()
    pred = (logits[:, 1] > logits[:, 0]).astype(int)
    assert np.array_equal(res['window_pred'], pred)
    votes = np.zeros((6, 2), dtype=int)
    np.add.at(votes, (slot, pred), 1)
    assert np.array_equal(res['votes'], votes)
    assert np.array_equal(res['prediction'], votes.argmax(1))
    assert np.allclose(res['pred_frac'], votes[:, 1] / votes.sum(1))


def test_data_parallel_step_structure_on_one_gpu(M):
    """The world_size > 1 step (graph: zero-grad/forward/backward -> eager all-reduce of the flat bucket ->
    graph: 1/W scale + clamp + SGD) exercised on ONE GPU with a stand-in all-reduce that sums two identical
    ranks (g -> 2g): with gscale = 1/2 it must reproduce the single-GPU trajectory bit for bit."""
    from deepards_amd.train import HotPathTrainer, FlatBucket
    g = _gold([p for p in GOLD if 'resnet18_b2_randn' in p][0])
    x = torch.from_numpy(g['x']).cuda()
    t = torch.from_numpy(g['target']).cuda()
    ref_model = build(M, 'resnet18', 0)
    ref_tr = HotPathTrainer(ref_model, use_graph=True)
    dp_model = build(M, 'resnet18', 0)
    dp_tr = HotPathTrainer(dp_model, use_graph=True, world_size=2, rank=0)
    dp_tr._synced = True                                # stand-in all-reduce, no process group: skip the broadcast
    orig = FlatBucket.allreduce
    FlatBucket.allreduce = lambda self, group=None: self.g.mul_(2.0)      # sum over 2 identical ranks
    try:
        for _ in range(4):
            l_ref = float(ref_tr.train_step(x, t))
            l_dp = float(dp_tr.train_step(x, t))
            assert l_ref == l_dp
    finally:
        FlatBucket.allreduce = orig
    for (n, p), (_, q) in zip(ref_model.named_parameters(), dp_model.named_parameters()):
        assert torch.equal(p, q), n


def _run_dp_children(tmp_path, gold_path, mode, world=2, graph=True):
    """`world` FRESH child processes (tests/tools/dp_child.py), all on cuda:0, gradients over gloo."""
    import subprocess
    import sys
    port = 29500 + (os.getpid() * 7 + hash(mode) % 97) % 3000
    procs, outs = [], []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   DP_CHILD_GRAPH='1' if graph else '0')
        out = str(tmp_path / ('dp_%s_rank%d.npz' % (mode, r)))
        outs.append(out)
        procs.append(subprocess.Popen([sys.executable, os.path.join(os.path.dirname(__file__), 'tools', 'dp_child.py'),
                                       gold_path, out, mode], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for p_ in procs:
        try:
            o, _ = p_.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o.decode(errors='replace'))
    for r, p_ in enumerate(procs):
        assert p_.returncode == 0, 'rank %d failed:\n%s' % (r, logs[r][-4000:])
    return [dict(np.load(o, allow_pickle=False)) for o in outs]


def test_data_parallel_two_processes_trajectory(M, tmp_path):
    """Config C4's step for real: two processes (one model replica each, both on this GPU, gloo all-reduce of the flat
    bucket), rank r trains on window r of the golden's B=2 batch.  Rank 1 starts from a DIFFERENT random initialisation:
    sync_replicas() must replace it with rank 0's.  Checked: (1) the ranks end with bit-identical parameters;
    (2) the mean of the rank losses and the parameters follow the reference's B=2 trajectory (golden) within the
    bounds of the single-process trajectory test; (3) they equal this build's own single-process B=2 run -- losses to
    2e-6, parameters to 2e-5.  Not bit-for-bit, and not to rounding either: the single process sums the two windows'
    weight-gradient contributions inside one split-K reduction while the data-parallel step adds two rank totals, and the
    BatchNorm / stem-statistics block geometry depends on how many windows a process holds (1 vs 2).  Those are other
    fp32 summation orders; from the second step on the two runs' parameters differ in their last bits, and an element
    whose pre-activation is within that noise of zero takes the other ReLU / max-pool branch in one of them (the
    ambiguity DESIGN.md 2 describes: ~1 element in 1e7 per step, each moving the gradients in front of it by up to ~1e-2
    relative).  Measured after three steps: 1e-6 ... 9.6e-6 (breath_block.conv1.weight) -- which elements flip depends on
    the last bits of everything, e.g. on the summation order inside the head kernels.  What IS exact is checked exactly:
    the replicas, and the head chain across batch sizes (test_functions_gpu: a window's logits / dx / dW shares are
    bit-identical whether it is trained alone or with another).
    """
    from deepards_amd.train import HotPathTrainer
    path = [p for p in GOLD if 'resnet18_b2_randn' in p][0]
    g = _gold(path)
    r0, r1 = _run_dp_children(tmp_path, path, 'traj')
    assert int(r0['allreduce_calls']) == int(r1['allreduce_calls']) == 3
    names = [k for k in r0 if k.startswith('p/')]
    for k in names:
        assert np.array_equal(r0[k], r1[k]), k                                     # replicas stay bit-identical
    losses = (r0['losses'] + r1['losses']) / 2                                     # mean of equal-shard means
    ref = g['sgd_losses64']
    log('dp2 losses', losses.tolist(), 'ref', ref.tolist())
    assert np.abs(losses - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())
    model = build(M, 'resnet18', int(g['seed']), str(g['first_pool_type']))
    tr = HotPathTrainer(model, optimizer='sgd', use_graph=True)
    x, t = torch.from_numpy(g['x']).cuda(), torch.from_numpy(g['target']).cuda()
    single = [float(tr.train_step(x, t)) for _ in range(3)]
    assert np.abs(losses - np.array(single)).max() < 2e-6
    worst, worst_name = 0.0, ''
    for n, p in model.named_parameters():
        a, b = r0['p/' + n], p.detach().cpu().numpy()
        key = 'sgd_p64/' + n
        if key not in g:
            continue                                    # dead parameters: rank 0's initial values everywhere
        if float(np.abs(a - b).max()) > worst:
            worst, worst_name = float(np.abs(a - b).max()), n
        d = digest(a)
        body = slice(None) if a.size <= 1024 else slice(0, -3)
        assert np.abs(d - g[key])[body].max() < 1.5e-4, n
    log('dp2 vs single-process params max abs diff %.3e (%s)' % (worst, worst_name))
    assert worst < 2e-5, (worst_name, worst)
    # dead parameters and buffers were broadcast too: rank 1 holds rank 0's values
    for n in DEAD_RESNET_PARAMS:
        assert np.array_equal(r0['p/' + n], r1['p/' + n]), n


def test_data_parallel_two_processes_epochs_from_store(tmp_path):
    """Two ranks, two shuffled epochs over the 20 fixture windows with a GLOBAL batch of 6 and NO seed anywhere (each
    rank even has a different global RNG state): the ranks must shard one shared permutation -- per step the two
    index lists are disjoint halves of one batch, an epoch covers 18 + 2 windows (the tail batch of 2 is legal, 1 per
    rank), the tail shape gets its own graph (2 graphs, no recapture), and the replicas end bit-identical."""
    path = [p for p in GOLD if 'densenet18_b2_randn' in p][0]
    r0, r1 = _run_dp_children(tmp_path, path, 'epoch')
    n = int(r0['n_steps'])
    assert n == int(r1['n_steps']) == 8                                              # 2 epochs x (3 full + 1 tail)
    for ep in range(2):
        seen = []
        for i in range(4 * ep, 4 * ep + 4):
            a, b = r0['idx%d' % i], r1['idx%d' % i]
            assert len(a) == len(b) == (3 if i % 4 < 3 else 1)
            assert not set(a.tolist()) & set(b.tolist())
            seen += a.tolist() + b.tolist()
        assert sorted(seen) == list(range(20))                                      # one permutation, fully covered
    assert int(r0['n_graphs']) == int(r1['n_graphs']) == 2
    assert int(r0['allreduce_calls']) == int(r1['allreduce_calls']) == 8
    for k in r0:
        if k.startswith('p/'):
            assert np.array_equal(r0[k], r1[k]), k
    assert np.isfinite(r0['losses']).all() and np.isfinite(r1['losses']).all()


def test_identity_blocks_hand_back_a_two_term_input_gradient(M, monkeypatch):
    """resnet18's identity blocks behind another block return conv1-dgrad(dy1) and leave the identity term beside it; the bn2
    backward in front sums the two while loading them (BasicBlockFunction split_dx): parameters, momentum and losses of 3
    captured fp32 steps are bit for bit those of the accumulating data-gradient convs; eager steps too; a term nobody picks up
    is an error, not a silently dropped gradient."""
    import deepards_amd.models.resnet as RN
    import deepards_amd.functional as F_
    from deepards_amd.train import HotPathTrainer
    x = torch.randn(4, 20, 1, 224, device='cuda')
    t = torch.zeros(4, 2, device='cuda')
    t[:2, 0] = 1
    t[2:, 1] = 1

    def run(split, graph):
        monkeypatch.setattr(RN, '_SPLIT_DX', split)
        tr = HotPathTrainer(build(M, 'resnet18', 5), use_graph=graph)
        out = [tr.train_step(x, t).clone() for _ in range(3)] + [tr.bucket.p.clone(), tr.state['buf'].clone()]
        tr.release_graphs()
        return out
    for graph in (True, False):
        for n, (a, b) in enumerate(zip(run(True, graph), run(False, graph))):
            assert torch.equal(a, b), (graph, n)
    with F_.training_step():
        F_._STEP['dout2'][123] = x
        with pytest.raises(RuntimeError, match='two-term'):
            F_.flush_backward()


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_last_block_pools_for_the_head(M, dtype, monkeypatch):
    """resnet18's last block hands the head its POOLED output (BasicBlockFunction pool_out; the block's map is never stored):
    losses, logits, parameters and momentum after 3 captured steps, and a forward-only step, are bit for bit those of the
    stored-map form (models.resnet._FUSED_TAIL = False)."""
    import deepards_amd.models.resnet as RN
    import deepards_amd.functional as F_
    from deepards_amd.train import HotPathTrainer
    x = torch.randn(4, 20, 1, 224, device='cuda')
    t = torch.zeros(4, 2, device='cuda')
    t[:2, 0] = 1
    t[2:, 1] = 1
    F_.set_conv_dtype(dtype)
    if dtype == 'bf16':
        F_.set_storage_dtype('bf16')
    try:
        def run(fused):
            monkeypatch.setattr(RN, '_FUSED_TAIL', fused)
            tr = HotPathTrainer(build(M, 'resnet18', 5), use_graph=True)
            losses = [tr.train_step(x, t).clone() for _ in range(3)]
            tl, tlog, _ = tr.test_step(x, t)
            out = losses + [tl.clone(), tlog.clone(), tr.bucket.p.clone(), tr.state['buf'].clone()]
            tr.release_graphs()
            return out
        for n, (a, b) in enumerate(zip(run(True), run(False))):
            assert torch.equal(a, b), n
    finally:
        F_.set_conv_dtype('f32')
        F_.set_storage_dtype('f32')


@pytest.mark.parametrize('backbone,drop', [('resnet18', 0.0), ('densenet18', 0.2)])
def test_captured_step_without_the_gradient_zero_fill(M, backbone, drop, monkeypatch):
    """A captured step whose every gradient destination has ONE writer with an overwrite form runs without the zero-fill of
    the gradient bucket (functional._OV, HotPathTrainer.grad_overwrite): parameters, momentum and the gradient bucket are bit
    for bit those of the zero-fill + accumulate form over 4 steps and two batch shapes; a writer that can only accumulate
    (the six-launch head chain) keeps the zero-fill form."""
    import deepards_amd.train as T
    from deepards_amd.train import HotPathTrainer
    x = torch.randn(6, 20, 1, 224, device='cuda')
    t = torch.zeros(6, 2, device='cuda')
    t[:3, 0] = 1
    t[3:, 1] = 1

    def run(overwrite):
        monkeypatch.setattr(T, '_GRAD_OVERWRITE', overwrite)
        tr = HotPathTrainer(build(M, backbone, 11, drop_rate=drop), use_graph=True)
        for n in (6, 6, 4, 6, 4):
            tr.train_step(x[:n].contiguous(), t[:n].contiguous())
        torch.cuda.synchronize()
        out = (tr.grad_overwrite, tr.bucket.p.clone(), tr.bucket.g.clone(), tr.state['buf'].clone())
        tr.release_graphs()
        return out
    a, b = run(True), run(False)
    assert a[0] and not b[0]
    for u, v in zip(a[1:], b[1:]):
        assert torch.equal(u, v)
    monkeypatch.setattr(T, '_GRAD_OVERWRITE', True)
    monkeypatch.setattr(T, '_FUSED_HEAD', False)
    tr = HotPathTrainer(build(M, backbone, 11, drop_rate=drop), use_graph=True)
    for _ in range(3):
        loss = tr.train_step(x, t)
    assert not tr.grad_overwrite and np.isfinite(float(loss))
    tr.release_graphs()


def test_recapture_on_shape_change_with_garbage_pending(M):
    """Pins the capture-window ownership rule (deepards_amd.train._capture_graph): cyclic garbage
    that owns captured graphs and device tensors is pending when a NEW shape is captured; it must be finalised before
    the window (gc.collect) and the collector must be off inside it.  Run once; passes by not aborting and by leaving
    the collector in the state it was found."""
    import gc
    from deepards_amd.train import HotPathTrainer
    x = torch.randn(4, 20, 1, 224, device='cuda')
    t = torch.zeros(4, 2, device='cuda')
    t[:, 0] = 1

    class Cycle(object):
        pass
    was = gc.isenabled()
    gc.disable()                                        # let the garbage below stay pending until _capture_graph
    try:
        for _ in range(2):
            junk = HotPathTrainer(build(M, 'densenet18', 7), use_graph=True)
            for _ in range(2):
                junk.train_step(x, t)
            junk.test_step(x, t)
            c = Cycle()
            c.self, c.trainer, c.blob = c, junk, torch.empty(1 << 20, device='cuda')
            del c, junk                                 # unreachable, but only a collection can free it
    finally:
        if was:
            gc.enable()
    tr = HotPathTrainer(build(M, 'densenet18', 8), use_graph=True)
    for n in (4, 4, 2, 2, 4):                           # capture B=4, then a second shape B=2, then replay both
        loss = tr.train_step(x[:n].contiguous(), t[:n].contiguous())
    assert np.isfinite(float(loss)) and len(tr._graphs) == 2
    assert gc.isenabled() == was
    tr.release_graphs()
    assert not tr._graphs and not tr._test_graphs


def test_driver_mirror_train_and_test_on_the_fixture():
    """``network_map[args.network](args).train_and_test()`` (train_ards_detector.py:1589-1590) on the reference
    fixture's windows == the same epochs driven by hand through HotPathTrainer, loss for loss, and a test epoch whose
    patient votes add up."""
    from deepards_amd import train_ards_detector as T
    from deepards_amd.data import DeviceTileStore
    from deepards_amd.train import HotPathTrainer, run_train_epoch_from_store
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'test_dataset_windows.npz'))
    mk = lambda: DeviceTileStore(z['x'], z['target'], float(z['mu']), float(z['std']))
    slot = torch.from_numpy(z['patient_slot'].astype(np.int64)) if 'patient_slot' in z.files else \
        torch.arange(20, dtype=torch.int64) % 6
    args = T.make_args(base_network='resnet18', epochs=2, batch_size=4, clip_grad=True, seed=5, train_store=mk(),
                       test_store=mk(), test_patient_slot=slot)
    cls = T.network_map[args.network](args)
    res = cls.train_and_test()
    got = res.get_meter('loss', 0)
    assert len(got) == 10 and all(np.isfinite(got))                 # 2 epochs x 5 batches of 4 windows
    # by hand: same seed -> same init, same shuffles
    torch.manual_seed(5)
    import deepards_amd.models as M
    model = M.CNNLinearNetwork(M.resnet18(initial_planes=64, first_pool_type='max', double_conv_first=False), 20, 0).cuda()
    tr = HotPathTrainer(model, optimizer='sgd', learning_rate=0.001, weight_decay=0.0001, clip_grad=True, clip_val=0.01)
    store, want = mk(), []
    for epoch in (1, 2):
        g = torch.Generator().manual_seed(5 + epoch)
        want += [float(l) for l in run_train_epoch_from_store(tr, store, batch_size=4, shuffle=True, generator=g)]
    assert got == want
    for (_, p), (_, q) in zip(cls.model.named_parameters(), model.named_parameters()):
        assert torch.equal(p, q)
    r = res.patient_results[(0, 2)]
    assert r['votes'].sum() == 20 and r['votes'].shape == (int(slot.max()) + 1, 2) and np.isfinite(r['mean_loss'])
    assert len(cls.preds) == 20 and sorted(cls.pred_idx) == list(range(20))
    # eager-style entry point of the batch loop
    x, t = store.batch([0, 1])
    l = cls.handle_train_optimization(cls.optimizer, None, t, x, 0, 1, 0, 3, cls.model)
    assert np.isfinite(float(l)) and len(res.get_meter('loss_epoch_3', 0)) == 1


def test_driver_mirror_per_breath_model_votes_per_breath():
    """cnn_single_breath_linear (PerBreathClassifierMixin, train_ards_detector.py:539-555, 959-964): (B, NB, 2) outputs,
    the loss repeats the window target over the breaths, every breath is a prediction and a patient vote."""
    from deepards_amd import train_ards_detector as T
    from deepards_amd.data import DeviceTileStore
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'test_dataset_windows.npz'))
    mk = lambda: DeviceTileStore(z['x'], z['target'], float(z['mu']), float(z['std']))
    slot = torch.arange(20, dtype=torch.int64) % 5
    args = T.make_args(network='cnn_single_breath_linear', base_network='densenet18', epochs=1, batch_size=4, seed=1,
                       train_store=mk(), test_store=mk(), test_patient_slot=slot)
    cls = T.network_map[args.network](args)
    res = cls.train_and_test()
    assert len(res.get_meter('loss', 0)) == 5 and all(np.isfinite(res.get_meter('loss', 0)))
    r = res.patient_results[(0, 1)]
    assert r['votes'].sum() == 20 * 20 and (r['votes'].sum(axis=1) == 4 * 20).all()
    assert len(cls.preds) == 400 and sorted(set(cls.pred_idx)) == list(range(20))
    out = cls.model(mk().batch([0, 1, 2])[0], None)
    assert tuple(out.shape) == (3, 20, 2)
    t = torch.from_numpy(z['target'][:3]).float().cuda()
    ref = torch.nn.functional.binary_cross_entropy_with_logits(out, t.unsqueeze(1).repeat(1, 20, 1))
    assert abs(float(cls.calc_loss(out, t, None)) - float(ref)) < 1e-6


def test_bf16_conv_arithmetic_resnet18(M):
    """BASELINE config C3 (resnet18-1D, bf16): with conv dtype 'bf16' the k3 s1 convs' forward and data gradient round
    their operands to bf16 (fp32 sums, fp32 storage, fp32 weight gradients and statistics).  Parity is stated against
    the oracle run with THE SAME operand rounding (np_ref bf16_convs=True).  Rounding to 8 significant bits is a
    discontinuity like a ReLU, only everywhere: an fp32-vs-fp64 difference of 1e-6 moves ~3e-4 of the elements to the
    neighbouring bf16 value, so the tolerances of this arithmetic, stated here since north_star gives none, are logits
    and loss within 5e-3 of the same-rounding oracle (measured 1e-3) and 5e-2 of the exact one (measured 1e-2),
    parameter gradients within 50 % relative L2 (measured 14-22 %, the stem and first-stage weights).  Training converges; the default dtype is untouched afterwards."""
    from deepards_amd import functional as F_
    from deepards_amd.functional import bce_with_logits
    from deepards_amd.train import HotPathTrainer
    assert F_.conv_dtype() == 'f32'
    x, t = seeded_batch(3, 20, 21, 'flow')
    p32 = seeded_params('resnet18', 6)
    p64 = {k: v.astype(np.float64) for k, v in p32.items()}
    ref = np_ref.cnn_linear_forward_backward(p64, x.astype(np.float64), t.astype(np.float64), bf16_convs=True)
    exact = np_ref.cnn_linear_forward_backward(p64, x.astype(np.float64), t.astype(np.float64), need_grads=False)
    xt, tt = torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()
    try:
        F_.set_conv_dtype('bf16')
        model = build(M, 'resnet18', 6)
        out = model(xt, None)
        got = out.detach().cpu().numpy()
        err, drift = np.abs(got - ref['logits']).max(), np.abs(got - exact['logits']).max()
        log('resnet18 bf16 convs: logits vs same-rounding oracle %.3e, vs exact oracle %.3e' % (err, drift))
        assert err < 5e-3 and 1e-5 < drift < 5e-2      # bf16 really ran; same-rounding oracle is 5-10x closer
        loss = bce_with_logits(out, tt)
        assert abs(float(loss) - ref['loss']) < 5e-3
        loss.backward()
        worst = 0.0
        for n, p in model.named_parameters():
            if n in ref['grads']:
                worst = max(worst, rel_l2(p.grad.cpu().numpy().astype(np.float64), ref['grads'][n]))
        log('resnet18 bf16 convs: worst parameter-gradient rel L2 vs same-rounding oracle %.3e' % worst)
        assert worst < 0.5                            # measured 0.14-0.22 (stem / first-stage weights, behind the most flips)
        tr = HotPathTrainer(build(M, 'resnet18', 6), use_graph=True)
        losses = [float(tr.train_step(xt, tt)) for _ in range(12)]
        assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    finally:
        F_.set_conv_dtype('f32')
    with torch.no_grad():
        assert np.abs(build(M, 'resnet18', 6)(xt, None).cpu().numpy() - exact['logits']).max() < 1e-4


def test_driver_mirror_kfolds_on_device_store():
    """BASELINE config C4's fold loop (train_ards_detector.py:317-338, dataset.py:765-830): patient-wise stratified folds
    on the device tile store, each fold with the scaling factors of ITS train windows, a fresh model per fold, the test
    epoch over the fold's test patients only."""
    from deepards_amd import train_ards_detector as T
    from deepards_amd.data import DeviceTileStore
    from deepards_amd.tiles import scaling_factors_for_indices
    rng = np.random.default_rng(8)
    n_pat, per = 8, 3
    pts = np.repeat(np.arange(n_pat), per)
    label_of = np.arange(n_pat) % 2
    wins = rng.standard_normal((n_pat * per, 20, 1, 224)) * (20 + 2 * pts)[:, None, None, None] + pts[:, None, None, None]
    tg = np.eye(2, dtype=np.float32)[label_of[pts]]
    train = DeviceTileStore(wins, tg, 0.0, 1.0).enable_kfolds(pts, 2)
    test = train.make_test_store_if_kfold()
    assert test.tiles.data_ptr() == train.tiles.data_ptr()
    args = T.make_args(base_network='densenet18', epochs=1, batch_size=4, kfolds=2, seed=2, train_store=train,
                       test_store=test, test_patient_slot=torch.from_numpy(pts))
    cls = T.network_map[args.network](args)
    res = cls.train_and_test()
    for k in (0, 1):
        tr_idx = train.get_kfold_indexes_for_fold(k)
        te_idx = test.get_kfold_indexes_for_fold(k)
        assert not set(pts[tr_idx]) & set(pts[te_idx]) and len(tr_idx) + len(te_idx) == n_pat * per
        mu, std = scaling_factors_for_indices(wins, tr_idx)
        assert train.scaling_factors[k] == (float(mu[0]), float(std[0])) == test.scaling_factors[k]
        assert len(res.get_meter('loss', k)) == len(tr_idx) // 4
        r = res.patient_results[(k, 1)]
        assert r['votes'].sum() == len(te_idx)
        assert set(np.nonzero(r['votes'].sum(axis=1))[0].tolist()) == set(pts[te_idx].tolist())
    assert train.scaling_factors[0] != train.scaling_factors[1]
    x, _ = test.batch([0])                                  # fold 1 is the current one: its factors normalise
    mu, std = train.scaling_factors[1]
    ref = ((wins[test.get_kfold_indexes_for_fold(1)[0]] - mu) / std).astype(np.float32)
    assert np.array_equal(x.cpu().numpy()[0], ref)


def test_cli_main_runs_config_c1_on_the_ingested_fixture(tmp_path):
    """``python -m deepards_amd.train_ards_detector -co <the C1 experiment file> --train-from-pickle <fixture> ...``
    (train_ards_detector.py:1579-1590): configuration merge, dataset ingest (the .npz `python -m deepards_amd.ingest`
    made from the reference's pickled fixture), patient-wise k-folds with minority oversampling, a fresh model per fold,
    per-patient votes over the fixture's 12 (anonymised) patients, reference-style checkpoint names."""
    from deepards_amd import train_ards_detector as T
    from deepards_amd import checkpoint as C
    from deepards_amd import ingest
    gold = os.path.join(os.path.dirname(__file__), 'golden', 'test_dataset.npz')
    exp = os.path.join(os.path.dirname(os.path.dirname(__file__)), 'deepards_amd', 'experiment_files',
                       'unpadded_centered_nb20_cnn_linear.yml')
    cls, res = T.main(['-co', exp, '--train-from-pickle', gold, '--kfolds', '2', '-e', '2', '-b', '4', '--base-network',
                       'resnet18', '--seed', '5', '--save-model', 'runs/c1.pth', '--save-model-per-epoch',
                       '--saved-models-dir', str(tmp_path)])
    a = cls.args
    assert (a.network, a.clip_grad, a.cuda_no_dp, a.oversample_minority, a.kfolds, a.batch_size) == \
        ('cnn_linear', True, True, True, 2, 4)                   # experiment file + CLI over defaults.yml
    ds = ingest.load_npz(gold)
    labels = ds.targets.argmax(1)
    all_test = []
    for fold in (0, 1):
        losses = res.get_meter('loss', fold)
        assert len(losses) >= 4 and np.isfinite(losses).all()
        r = res.patient_results[(fold, 2)]
        assert r['votes'].shape == (12, 2)                      # the fixture's 12 patients
        test_windows = sorted(set(r['window_abs_index'].tolist()))
        voted = set(np.nonzero(r['votes'].sum(axis=1))[0].tolist())
        assert voted == set(ds.patient_slot[test_windows].tolist())
        assert r['votes'].sum() == len(test_windows) == len(r['window_pred'])
        for p in voted:                                         # every patient's votes = its windows' predictions
            w = [i for i in test_windows if ds.patient_slot[i] == p]
            pred = {int(i): int(q) for i, q in zip(r['window_abs_index'], r['window_pred'])}
            assert r['votes'][p].tolist() == [sum(pred[i] == 0 for i in w), sum(pred[i] == 1 for i in w)]
        all_test += test_windows
        for ep in (1, 2):
            assert os.path.exists(os.path.join(str(tmp_path), 'c1-epoch%d-fold%d.pth' % (ep, fold)))
        assert os.path.exists(os.path.join(str(tmp_path), 'c1-fold%d.pth' % fold))
    assert sorted(all_test) == list(range(20))                  # the two test folds partition the windows
    assert cls.pred_idx == res.patient_results[(1, 2)]['window_abs_index'].tolist()     # absolute indices (obs_idx)
    # the train fold was oversampled to class parity (dataset.py:561-573)
    tr = cls.args.train_store if cls.args.train_store is not None else None
    assert tr is None                                           # stores came from the pickle, not from the caller
    # the last fold's checkpoint is this package's own whole module: loads back bit-identical
    back = C.load_model_weights(os.path.join(str(tmp_path), 'c1-fold1.pth'), lambda: None)
    for (k, p), (_, q) in zip(cls.model.state_dict().items(), back.state_dict().items()):
        assert torch.equal(p.cpu(), q.cpu()), k
    # --no-train --load-checkpoint: inference over the same folds with the saved model (evaluate.py's use)
    cls2, res2 = T.main(['--cuda-no-dp', '--train-from-pickle', gold, '--kfolds', '2', '-e', '1', '-b', '4',
                         '--base-network', 'resnet18', '--no-train', '--only-fold', '1', '--load-checkpoint',
                         os.path.join(str(tmp_path), 'c1-fold1.pth')])
    assert (1, 1) in res2.patient_results and (0, 1) not in res2.patient_results
    assert res2.patient_results[(1, 1)]['votes'].sum() == res.patient_results[(1, 2)]['votes'].sum()


def test_cli_main_holdout_with_test_pickle_and_foreign_base_network(tmp_path):
    """Holdout run (no k-folds): --test-from-pickle gets the TRAIN set's scaling factors (train_ards_detector.py:285);
    --load-base-network takes the breath block out of a checkpoint saved under the REFERENCE's class paths, read
    without unpickling (:383-388)."""
    from deepards_amd import train_ards_detector as T
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), 'tools'))
    from ref_paths import as_reference_classes as _as_reference_classes
    gold = os.path.join(os.path.dirname(__file__), 'golden', 'test_dataset.npz')
    path = str(tmp_path / 'ref_model.pth')

    def save(M):
        torch.manual_seed(11)
        m = M.CNNLinearNetwork(M.densenet18(), 20, 0)
        torch.save(m, path, _use_new_zipfile_serialization=False)          # the pytorch-1.0 format of the reference's env
        return m
    ref = _as_reference_classes(save)
    cls, res = T.main(['--cuda-no-dp', '--train-from-pickle', gold, '--test-from-pickle', gold, '-e', '1', '-b', '6',
                       '--base-network', 'resnet18', '--load-base-network', path, '--freeze-base-network', '--seed', '1'])
    assert cls.model.breath_block.network_name == 'densenet18'             # the file's backbone, not --base-network
    r = res.patient_results[(0, 1)]
    assert r['votes'].sum() == 20 and r['votes'].shape == (12, 2)
    # frozen base network: only linear_final moved (get_base_network :411-413)
    for (k, p), (_, q) in zip(cls.model.breath_block.state_dict().items(), ref.breath_block.state_dict().items()):
        assert torch.equal(p.cpu(), q), k
    assert len(res.get_meter('loss', 0)) == 4                              # 20 windows, batches of 6: 6 + 6 + 6 + 2


# ---- the bench shape (BASELINE configs[1]: B = 64 windows of (20, 1, 224)) --------------------------------------------
def _set_fast_paths(on):
    """Toggle every fast path at once: Winograd forward / data / weight gradients, split-K tail tiles, paired stride-2
    launches, ReLU bit masks -- off = the plain direct kernels."""
    from deepards_amd import _lib, functional as F_, hip_ops as H_
    F_._WINOGRAD, F_._PAIR_S2, F_._BN_MASK, H_.WINOGRAD_WGRAD = on, on, on, on
    _lib.lib().da_debug_set(3, 1 if on else 0)
    _lib.lib().da_wino_debug_tail(1 if on else 0)


@pytest.mark.parametrize('backbone', ['resnet18', 'densenet18'])
def test_bench_shape_b64_windows_vs_oracle_and_independence(M, backbone):
    """At the size the metric is quoted on every batched path is live (split-K tail tiles, F(4,3), XCD-chunked block
    order, 16-channel BatchNorm blocks).  (i) window independence: model(x64)[i] vs model(x64[i:i+1])[0] for all 64
    windows.  Not bit-for-bit BY DESIGN: the tile a window's rows land in (full tile or split-K half tile) and the
    BatchNorm block geometry (32- or 16-channel blocks, chosen from the window count) change the fp32 summation ORDER
    with the batch size, never the operands -- bound 4e-6, three orders below the north-star tolerance.  (ii) the
    logits of 4 windows spread over the batch, and their loss terms, against the numpy oracle run on those 4 windows
    alone (windows are independent, SURVEY finding 3) within 1e-4."""
    model = build(M, backbone, 4)
    x, t = seeded_batch(64, 20, 64)
    xt = torch.from_numpy(x).cuda()
    with torch.no_grad():
        full = model(xt, None).cpu().numpy().astype(np.float64)
        single = np.concatenate([model(xt[i:i + 1], None).cpu().numpy() for i in range(64)]).astype(np.float64)
    dev = np.abs(full - single).max()
    log(backbone, 'B=64 window independence: max |model(x64)[i] - model(x64[i:i+1])[0]| = %.3e' % dev)
    assert dev < 4e-6 * max(1.0, np.abs(full).max())
    pick = [0, 21, 42, 63]
    params = {k: v.astype(np.float64) for k, v in seeded_params(backbone, 4).items()}
    ref = np_ref.cnn_linear_forward_backward(params, x[pick].astype(np.float64), t[pick].astype(np.float64),
                                             backbone=backbone, need_grads=False)
    err = np.abs(full[pick] - ref['logits']).max()
    log(backbone, 'B=64: logits of windows %s vs oracle: %.3e' % (pick, err))
    assert err < 1e-4
    bce = lambda z, y: np.maximum(z, 0) - z * y + np.log1p(np.exp(-np.abs(z)))
    assert np.abs(bce(full[pick], t[pick]) - bce(ref['logits'], t[pick])).max() < 1e-5


@pytest.mark.parametrize('backbone', ['resnet18', 'densenet18'])
def test_bench_shape_b64_gradients_are_the_mean_of_its_shards(M, backbone):
    """Linearity at full size (the oracle would need minutes for 64 windows): the loss is a mean over windows and
    BatchNorm never crosses windows, so the B=64 gradient is the mean of the gradients of its eight 8-window shards.
    The B=64 run takes the batched paths (one weight-gradient launch over 1280 rows with split-K slabs, tail tiles),
    the shards other tile shapes and split counts: agreement pins the batching logic, not the arithmetic (that is the
    goldens' job).  Run on the 'active' parameters (BN beta += 6: every ReLU active; avg first pool for the ResNet), so
    that no activation decision can differ between the two runs -- on generic parameters a ReLU element within 1e-6 of
    zero flips between ANY two fp32 summation orders and moves upstream gradients by 1e-3...3e-2 (see the golden test).
    Bound: rel-l2 1e-4 per parameter (measured 2e-5: summation order on gradients that are sums of large cancelling
    terms with beta = 6; a flipped decision would show as 1e-3 or more); the DenseNet stem (MaxPool1d is part of the architecture: near-ties between pool
    candidates remain) is logged only."""
    from deepards_amd.functional import bce_with_logits
    x, t = seeded_batch(64, 20, 65)
    xt, tt = torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()

    def grads(xs, ts):
        model = build(M, backbone, 4, first_pool_type='avg', shift=6.0)
        bce_with_logits(model(xs, None), ts).backward()
        return {n: p.grad.double() for n, p in model.named_parameters() if p.grad is not None}
    full = grads(xt, tt)
    acc = None
    for s0 in range(0, 64, 8):
        g = grads(xt[s0:s0 + 8].contiguous(), tt[s0:s0 + 8].contiguous())
        acc = g if acc is None else {n: acc[n] + g[n] for n in g}
    worst = 0.0
    for n in full:
        r = float((full[n] - acc[n] / 8).norm() / (full[n].norm() + 1e-30))
        if float(full[n].norm()) < 1e-9:                  # analytically zero gradients (a conv in front of a BatchNorm
            assert float((acc[n] / 8).norm()) < 1e-6, n   # with every ReLU active): both runs must say ~0
            continue
        if backbone == 'densenet18' and ('conv0' in n or 'norm0' in n):
            # MaxPool1d is part of the DenseNet stem: near-ties between pool candidates remain decisions, and with
            # beta = 6 the stem's gamma gradient is a sum of large cancelling terms -- logged, pinned by the goldens
            log(backbone, '   B=64 linearity, stem parameter %s: rel-l2 %.3e (not asserted)' % (n, r))
            continue
        # beta of a BatchNorm whose only consumer is (ReLU, all active ->) conv -> BatchNorm has an analytically ZERO
        # gradient (the next BatchNorm removes any per-channel constant): both runs hold rounding noise there, judged
        # absolutely
        abs_err = float((full[n] - acc[n] / 8).abs().max())
        if abs_err < 2e-7:
            continue
        worst = max(worst, r)
        assert r < 1e-4, (n, r, abs_err)
    log(backbone, 'B=64 gradient vs mean of 8 shard gradients: worst rel-l2 %.3e' % worst)


@pytest.mark.parametrize('backbone', ['resnet18', 'densenet18'])
def test_fast_paths_against_the_direct_kernels(M, backbone):
    """Second check, not the only one: the fast paths (Winograd F(2,3) / F(4,3) forward, data and weight gradients,
    split-K tail tiles, paired stride-2 launches, ReLU bit masks) against the plain direct kernels of the same build,
    whole model, at batch sizes around the tile / round boundaries including the bench shape.  Logits within 2e-5;
    parameter gradients within 1e-4 rel-l2 unless an activation decision differs between the two runs (then a looser
    3e-2: both runs are fp32, neither is exact -- exactness is pinned by the goldens above)."""
    from deepards_amd.functional import bce_with_logits
    res = {}
    try:
        for fast in (True, False):
            _set_fast_paths(fast)
            for b in (1, 3, 16, 64, 65):
                model = build(M, backbone, 1)
                x, t = seeded_batch(b, 20, b)
                o = model(torch.from_numpy(x).cuda(), None)
                bce_with_logits(o, torch.from_numpy(t).cuda()).backward()
                res[(fast, b)] = (o.detach().double().cpu().numpy(),
                                  {n: p.grad.double().cpu().numpy() for n, p in model.named_parameters() if p.grad is not None})
    finally:
        _set_fast_paths(True)
    for b in (1, 3, 16, 64, 65):
        (la, ga), (lb, gb) = res[(True, b)], res[(False, b)]
        e = np.abs(la - lb).max()
        worst = max(rel_l2(ga[n], gb[n]) for n in ga)
        tight = sum(rel_l2(ga[n], gb[n]) < 1e-4 for n in ga)
        log(backbone, 'fast vs direct kernels B=%d: logits %.2e, worst grad rel-l2 %.2e (%d of %d parameters < 1e-4)' %
            (b, e, worst, tight, len(ga)))
        assert e < 2e-5 and worst < 3e-2
        assert rel_l2(ga['linear_final.weight'], gb['linear_final.weight']) < 1e-5      # behind no decision at all


def test_densenet_dropout_on_against_the_oracle_with_the_device_masks(M):
    """DenseNet with drop_rate 0.2 ACTIVE (the reference never leaves train mode, SURVEY finding 4) against
    np_ref.densenet18_features(drop_masks=...): the device's counter-based keep masks are exported by running the same
    generator (da_dropout, same seed / salt / rate) over a tensor of ones, handed to the oracle as explicit masks, and
    logits / loss / every parameter gradient must agree like the dropout-free goldens do (1e-4; decision-matched)."""
    from deepards_amd import hip_ops as H_
    from deepards_amd.functional import bce_with_logits
    model = build(M, 'densenet18', 2, drop_rate=0.2)
    feats = model.breath_block.features
    x, t = seeded_batch(3, 20, 9)
    xt, tt = torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()
    seed_used = feats._drop_seed.clone() + (0x9E3779B97F4A7C15 >> 1)        # forward bumps the seed, then uses it
    out = model(xt, None)
    assert torch.equal(feats._drop_seed, seed_used)
    loss = bce_with_logits(out, tt)
    loss.backward()
    masks, salt = {}, 0
    for bi, l in ((1, 56), (2, 28), (3, 14), (4, 7)):
        for li in (1, 2):
            salt += 1
            m = H_.dropout(torch.ones(60, l, 32, device='cuda'), seed_used, salt, 0.2)      # RLC keep mask / (1 - p)
            masks[(bi, li)] = m.permute(0, 2, 1).cpu().numpy().astype(np.float64)           # oracle layout (N, C, L)
            vals = np.unique(masks[(bi, li)])
            assert set(np.round(vals, 6).tolist()) <= {0.0, 1.25}
    kept = np.mean([m.mean() / 1.25 for m in masks.values()])
    assert abs(kept - 0.8) < 0.01                                                            # the rate is the rate
    params = {k: v.astype(np.float64) for k, v in seeded_params('densenet18', 2).items()}
    ref = np_ref.cnn_linear_forward_backward(params, x.astype(np.float64), t.astype(np.float64), backbone='densenet18',
                                             drop_masks=masks)
    err = np.abs(out.detach().cpu().numpy() - ref['logits']).max()
    log('densenet18 dropout ON vs oracle with the device masks: logits %.3e loss %.8f vs %.8f' % (err, float(loss), ref['loss']))
    assert err < 1e-4 and abs(float(loss) - ref['loss']) < 1e-5
    ours = {n: p.grad.cpu().numpy().astype(np.float64) for n, p in model.named_parameters() if p.grad is not None}
    matched, flips, _ = decision_matched_gradients(ref, ours, 'densenet18 dropout on')
    assert len(flips) <= 6
    for n in ours:
        assert rel_l2(ours[n], matched[n]) <= 1e-4 or np.abs(ours[n] - matched[n]).max() <= 1e-4, n
    nodrop = np_ref.cnn_linear_forward_backward(params, x.astype(np.float64), t.astype(np.float64), backbone='densenet18',
                                                need_grads=False)
    assert np.abs(nodrop['logits'] - ref['logits']).max() > 1e-3                             # the masks do matter


def test_rccl_accepts_the_data_parallel_collectives_on_one_gpu():
    """RCCL itself (backend 'nccl'), as far as ONE GPU can show it: a fresh process with a world-size-1 process group
    issues every collective of the data-parallel path -- the fp32 / int64 broadcasts of sync_replicas, the seed broadcast
    of shared_generator, the sum all-reduce of the flat gradient bucket between the two captured graphs, bench.py's
    float64 MAX all-reduce and barrier (scripts/rccl_world1_probe.py).  No data crosses GPUs at world size 1; the
    2-process tests above carry the real exchange over gloo.  What stays unmeasured until a multi-GPU node runs it is
    RCCL's inter-GPU transport."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(29600 + os.getpid() % 300))
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, 'scripts', 'rccl_world1_probe.py')], env=env, capture_output=True,
                       text=True, timeout=300)
    assert p.returncode == 0 and 'rccl world-1 probe ok' in p.stdout, (p.stdout[-2000:], p.stderr[-2000:])


def test_evaluate_reruns_saved_models_per_fold(tmp_path):
    """deepards/evaluate.py:15-49 on the hot path: models saved per fold by a k-fold training run are pushed over their
    fold's test patients again (`python -m deepards_amd.evaluate -co <evaluate config>`), one "epoch" per listed model;
    the per-patient rows reproduce the training run's own last test epoch vote for vote, and the per-fold table holds
    patient accuracy and the AUC of the ARDS vote share."""
    from deepards_amd import evaluate as E
    from deepards_amd import train_ards_detector as T
    gold = os.path.join(os.path.dirname(__file__), 'golden', 'test_dataset.npz')
    cls, res = T.main(['--cuda-no-dp', '--train-from-pickle', gold, '--kfolds', '2', '-e', '1', '-b', '4',
                       '--base-network', 'densenet18', '--seed', '3', '--save-model', 'ev.pth',
                       '--saved-models-dir', str(tmp_path)])
    cfg = tmp_path / 'evaluate.yml'
    cfg.write_text('cuda_no_dp: true\nkfolds: 2\nbatch_size: 4\nnetwork: cnn_linear\nbase_network: densenet18\n'
                   'oversample: false\ntrain_from_pickle: %s\nexperiment_name: ev\nseed: 3\n'
                   'models:\n  0:\n   - ev-fold0.pth\n  1:\n   - ev-fold1.pth\n   - ev-fold1.pth\n' % gold)
    ecls, rows, table = E.main(['-co', str(cfg), '--saved-models-dir', str(tmp_path)])
    assert [t[0] for t in table] == [0, 1] and all(0.0 <= t[1] <= 1.0 for t in table)
    assert all(np.isnan(t[2]) or 0.0 <= t[2] <= 1.0 for t in table)
    assert {(r[0], r[1]) for r in rows} == {(0, 0), (1, 0), (1, 1)}          # fold 1 lists two models = two "epochs"
    f0 = sorted(r[2] for r in rows if (r[0], r[1]) == (0, 0))
    f1 = sorted(r[2] for r in rows if (r[0], r[1]) == (1, 0))
    assert sorted(f0 + f1) == list(range(12)) and f1 == sorted(r[2] for r in rows if (r[0], r[1]) == (1, 1))
    # dropout is active in the test epoch (the reference never calls eval()), so votes are compared on the loss-free
    # part: every test patient of a fold appears once per model, with its true class
    z = np.load(gold)
    truth = {int(p): int(z['target'][i].argmax()) for i, p in enumerate(z['patient_slot'])}
    assert all(r[3] == truth[r[2]] for r in rows)
    assert ecls.args.oversample_minority is False                            # the legacy `oversample` key was honoured


@pytest.mark.parametrize('backbone', ['resnet18', 'densenet18'])
def test_folds_in_flight_reproduce_the_sequential_fold_loop(tmp_path, backbone):
    """--folds-in-flight 2: two k-folds side by side on one GPU (own store view, model, captured step and stream each,
    batches walked round-robin) give every fold the losses, votes and weights of the one-after-the-other loop, bit for
    bit -- with minority oversampling and per-epoch re-draws on (each fold owns its sampler), and with densenet18's
    dropout active (its seed is a module buffer: the stream-placement trial snapshots and restores it)."""
    from deepards_amd import train_ards_detector as T
    gold = os.path.join(os.path.dirname(__file__), 'golden', 'test_dataset.npz')
    exp = os.path.join(os.path.dirname(os.path.dirname(__file__)), 'deepards_amd', 'experiment_files',
                       'unpadded_centered_nb20_cnn_linear.yml')
    runs = {}
    for flight in (1, 2, 3):
        argv = ['-co', exp, '--train-from-pickle', gold, '--kfolds', '3', '-e', '2', '-b', '4', '--base-network', backbone,
                '--seed', '9', '--reshuffle-oversample-per-epoch', '--save-model', 'runs/f.pth', '--saved-models-dir',
                str(tmp_path / ('flight%d' % flight)), '--folds-in-flight', str(flight)]
        cls, res = T.main(argv)
        runs[flight] = (cls, res)
    _, r1 = runs[1]
    for flight in (2, 3):
        _, r = runs[flight]
        for fold in range(3):
            assert r.get_meter("loss", fold) == r1.get_meter("loss", fold), (flight, fold)
            for ep in (1, 2):
                a, b = r.patient_results[(fold, ep)], r1.patient_results[(fold, ep)]
                assert np.array_equal(a['votes'], b['votes']) and np.array_equal(a['window_pred'], b['window_pred'])
                assert a['mean_loss'] == b['mean_loss']
            ma = torch.load(str(tmp_path / ('flight%d' % flight) / ('f-fold%d.pth' % fold)), weights_only=False)
            mb = torch.load(str(tmp_path / 'flight1' / ('f-fold%d.pth' % fold)), weights_only=False)
            for (k, p), (_, q) in zip(ma.state_dict().items(), mb.state_dict().items()):
                assert torch.equal(p, q), (flight, fold, k)


def test_folds_in_flight_eager_steps(tmp_path):
    """--folds-in-flight with --no-graph (eager launches from two streams, no captured step, no placement trial):
    still the sequential loop's numbers."""
    from deepards_amd import train_ards_detector as T
    gold = os.path.join(os.path.dirname(__file__), 'golden', 'test_dataset.npz')
    out = {}
    for flight in (1, 2):
        cls, res = T.main(['--cuda-no-dp', '--train-from-pickle', gold, '--kfolds', '2', '-e', '1', '-b', '6', '--base-network',
                           'resnet18', '--seed', '4', '--clip-grad', '--no-graph', '--folds-in-flight', str(flight)])
        out[flight] = [res.get_meter('loss', f) for f in (0, 1)] + [res.patient_results[(f, 1)]['votes'].tolist() for f in (0, 1)]
    assert out[1] == out[2]


@pytest.mark.parametrize('backbone', ['resnet18', 'densenet18'])
def test_trainer_snapshot_restore_is_traceless(M, backbone):
    """HotPathTrainer.snapshot / restore (what the stream-placement trial of --folds-in-flight relies on): steps taken
    after a snapshot leave no trace once it is restored -- the next steps repeat bit for bit (parameters, momentum,
    BatchNorm running statistics, densenet's dropout seed, static batches)."""
    from deepards_amd.train import HotPathTrainer
    x, t = seeded_batch(4, 20, 3)
    xt, tt = torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()
    model = build(M, backbone, 2, drop_rate=0.2) if backbone == 'densenet18' else build(M, backbone, 2)
    tr = HotPathTrainer(model, use_graph=True)
    for _ in range(3):
        tr.train_step(xt, tt)
    snap = tr.snapshot()

    def three():
        losses = [float(tr.train_step(xt + 0.1 * i, tt)) for i in range(3)]
        return losses, {k: v.clone() for k, v in model.state_dict().items()}, tr.steps
    a = three()
    tr.restore(snap)
    b = three()
    assert a[0] == b[0] and a[2] == b[2]
    for k in a[1]:
        assert torch.equal(a[1][k], b[1][k]), k


def test_fold_groups_on_the_hip_path(tmp_path):
    """BASELINE config C4 (k folds, each data-parallel over a group of GPUs) on the HIP path, as far as one GPU goes: FOUR
    fresh processes share cuda:0 over gloo as 2 fold groups x 2 data-parallel ranks (--fold-groups 2, 4 folds: group 0
    trains folds 0 and 2, group 1 folds 1 and 3, every step's gradients reduced inside the group).  Every rank must end
    with every fold's patient results, and they must be those of the SAME folds trained one after the other by ONE
    data-parallel pair (two processes, no fold groups): bit for bit -- a fold's numbers may not depend on which group ran it
    or on what ran beside it.  (The single-process fold loop draws another epoch permutation -- a data-parallel group shares
    ONE drawn by its leader -- so it is a different, equally valid trajectory: only the test windows per patient and the size
    of the losses are compared with it.)"""
    import subprocess
    import sys
    from deepards_amd import train_ards_detector as T
    gold = os.path.join(os.path.dirname(__file__), 'golden', 'test_dataset.npz')
    argv = ['--cuda-no-dp', '--train-from-pickle', gold, '--kfolds', '4', '-e', '1', '-b', '4', '--base-network', 'resnet18',
            '--seed', '11', '--clip-grad']
    _, ref = T.main(argv)                                         # one process, folds one after the other

    def children(world, extra, tag):
        port = 31500 + (os.getpid() * 11 + world * 17) % 3000
        procs, outs = [], []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
            out = str(tmp_path / ('fg_%s_rank%d.npz' % (tag, r)))
            outs.append(out)
            procs.append(subprocess.Popen([sys.executable, os.path.join(os.path.dirname(__file__), 'tools', 'fold_group_child.py'), out] +
                                          argv + extra, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
        logs = []
        for p_ in procs:
            try:
                o, _ = p_.communicate(timeout=420)
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                raise
            logs.append(o.decode(errors='replace'))
        for r, p_ in enumerate(procs):
            assert p_.returncode == 0, 'rank %d failed:\n%s' % (r, logs[r][-4000:])
        return [dict(np.load(o, allow_pickle=False)) for o in outs]

    pair = children(2, [], 'pair')                                # one data-parallel pair, folds one after the other
    groups = children(4, ['--fold-groups', '2'], 'groups')
    keys = sorted(k for k in pair[0])
    assert len([k for k in keys if k.startswith('votes/')]) == len(ref.patient_results) == 4
    for got in pair[1:] + groups:
        assert sorted(got) == keys
        for k in keys:
            assert np.array_equal(got[k], pair[0][k]), k
    for (fold, ep), r_ in ref.patient_results.items():            # the same test windows per patient as the one-process loop
        assert np.array_equal(pair[0]['votes/%d/%d' % (fold, ep)].sum(axis=1), np.asarray(r_['votes']).sum(axis=1)), fold
        assert abs(float(pair[0]['loss/%d/%d' % (fold, ep)]) - float(r_['mean_loss'])) < 0.2, fold
