"""CPU tests of the drop-in plumbing around the hot path: configuration merge and command line (reference config.py:6-22,
train_ards_detector.py:1439-1590), the dataset pickle read WITHOUT unpickling (dataset.py:706-763 wire format), the
k-fold / oversampling index plumbing (dataset.py:561-582,765-830) and checkpoints in the reference's formats
(train_ards_detector.py:355-388,468-469)."""
import io
import os
import pickle
import sys
import types

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden')
REF_PKL = '/root/reference/deepards/tests/test_dataset.pkl'


# ---- configuration / command line -----------------------------------------------------------------------------------
def test_configuration_merge_precedence_and_reference_values(tmp_path):
    """defaults.yml < override file < command line; store_true switches stay None ("not given") until a file or a
    flag sets them; the defaults are the reference's VALUES (defaults.yml:12-33,66-70)."""
    from deepards_amd import train_ards_detector as T
    from deepards_amd.config import Configuration
    parse = lambda argv: Configuration(T.build_parser().parse_args(argv), T.BUILD_DEFAULTS)
    a = parse([])
    assert (a.network, a.epochs, a.batch_size, a.base_network, a.loader_threads, a.initial_planes,
            a.resnet_first_pool_type, a.optimizer, a.dataset_type, a.learning_rate, a.n_sub_batches, a.weight_decay,
            a.loss_func, a.clip_val, a.cuda_device, a.oversample_all_factor, a.undersample_factor, a.train_pt_frac,
            a.time_series_hidden_units, a.stop_thresh, a.stop_after_epoch) == \
        ('cnn_linear', 10, 16, 'densenet18', 0, 64, 'max', 'sgd', 'unpadded_centered_sequences', 0.001, 20, 0.0001,
         'bce', 0.01, 0, 1.0, -1, 1.0, 16, 1.5, 1)
    assert a.clip_grad is None and a.cuda is None and a.cuda_no_dp is None and a.kfolds is None and a.seed is None
    assert a.use_graph is True
    exp = os.path.join(ROOT, 'deepards_amd', 'experiment_files', 'unpadded_centered_nb20_cnn_linear.yml')
    b = parse(['-co', exp])
    assert (b.kfolds, b.clip_grad, b.cuda_no_dp, b.oversample_minority, b.random_kfold, b.batch_size, b.epochs) == \
        (5, True, True, True, False, 16, 10)                     # the reference's experiment file for config C1
    over = tmp_path / 'o.yml'
    over.write_text('batch_size: 32\nlearning_rate: 0.01\nclip_grad: true\nbase_network: resnet18\n')
    c = parse(['-co', str(over), '-b', '8', '--optimizer', 'adam', '--no-graph'])
    assert (c.batch_size, c.learning_rate, c.clip_grad, c.base_network, c.optimizer, c.use_graph) == \
        (8, 0.01, True, 'resnet18', 'adam', False)               # CLI beats the file, the file beats defaults.yml
    c.network = 'cnn_linear_to_mean'                              # main() writes through (train_ards_detector.py:1583-1584)
    assert c.network == 'cnn_linear_to_mean' and 'network' in c
    with pytest.raises(AttributeError):
        c.not_a_key


def test_parser_keeps_reference_flag_names_and_refuses_out_of_scope_flags():
    from deepards_amd import train_ards_detector as T
    p = T.build_parser()
    ns = p.parse_args(['-co', 'x.yml', '-n', 'cnn_linear', '-e', '3', '-p', 'tr.pkl', '--test-from-pickle', 'te.pkl',
                       '--cuda-no-dp', '--cuda-device', '2', '-b', '4', '--base-network', 'resnet18', '-nb', '20',
                       '--kfolds', '5', '-rip', '64', '-rfpt', 'avg', '--optimizer', 'sgd', '-dt',
                       'unpadded_centered_sequences', '-lr', '0.01', '--save-model', 'm.pth', '--save-model-per-epoch',
                       '--load-base-network', 'b.pth', '--load-checkpoint', 'c.pth', '--no-train', '-wd', '0.1', '-loss',
                       'bce', '--oversample-minority', '--oversample-all-factor', '1.5', '--reshuffle-oversample-per-epoch',
                       '--freeze-base-network', '--clip-grad', '--clip-val', '0.02', '--only-fold', '1',
                       '--saved-models-dir', 'd', '--random-kfold', '--unshuffled', '--debug', '--no-test-after-epochs'])
    assert (ns.config_override, ns.train_from_pickle, ns.test_from_pickle, ns.cuda_no_dp, ns.cuda_device, ns.kfolds,
            ns.resnet_first_pool_type, ns.save_model_per_epoch, ns.clip_val, ns.only_fold, ns.oversample_all_factor) == \
        ('x.yml', 'tr.pkl', 'te.pkl', True, 2, 5, 'avg', True, 0.02, 1, 1.5)
    assert all(v is None for k, v in vars(p.parse_args([])).items()), 'every parser default must be None'
    for flag in ('--transforms', '--butter-low', '-usf', '--load-siamese', '--plot-dtw-with-disease'):
        with pytest.raises(SystemExit, match='outside the accelerated'):
            T.main([flag, '1'])
    with pytest.raises(Exception, match='Must specify a filename'):
        T.main(['--save-model-per-epoch', '--cuda-no-dp'])


# ---- the dataset pickle, read without unpickling ----------------------------------------------------------------------
def test_ingested_fixture_equals_the_golden_windows_and_keeps_the_patients():
    """tests/golden/test_dataset.npz is `python -m deepards_amd.ingest <reference fixture> <npz>` (anonymised): the same
    20 windows / targets / factors as the opcode-walk export of round 1, plus 12 patient slots and the window hours."""
    from deepards_amd import ingest
    ds = ingest.load_npz(os.path.join(GOLD, 'test_dataset.npz'))
    old = np.load(os.path.join(GOLD, 'test_dataset_windows.npz'))
    assert np.array_equal(ds.windows, old['x']) and np.array_equal(ds.targets, old['target'])
    mu, std = ds.scaling_factors[None]
    assert float(mu[0]) == float(old['mu']) == 2.0560646853765587 and float(std[0]) == float(old['std'])
    assert ds.n_patients == 12 and len(ds) == 20 and ds.patients is None             # anonymised export
    assert ds.patient_slot.tolist() == [0, 1, 2, 2, 3, 3, 4, 5, 5, 6, 7, 8, 8, 8, 9, 9, 9, 9, 10, 11]
    assert ds.hours.shape == (20, 20) and np.isfinite(ds.hours).all() and (np.diff(ds.hours, axis=1) >= 0).all()
    assert (ds.n_sub_batches, ds.dataset_type, ds.train, ds.total_kfolds) == (20, 'unpadded_centered_sequences', False, None)
    # an export of the round-1 file (x / target / mu / std only) still loads
    ds0 = ingest.load_npz(os.path.join(GOLD, 'test_dataset_windows.npz'))
    assert np.array_equal(ds0.windows, ds.windows) and float(ds0.scaling_factors[None][0][0]) == float(mu[0])


@pytest.mark.skipif(not os.path.exists(REF_PKL), reason='build container only: the reference fixture does not travel')
def test_reference_fixture_pickle_parses_inertly_to_the_committed_npz(tmp_path):
    from deepards_amd import ingest
    ds = ingest.read_ards_dataset(REF_PKL)
    gold = ingest.load_npz(os.path.join(GOLD, 'test_dataset.npz'))
    assert np.array_equal(ds.windows, gold.windows) and np.array_equal(ds.targets, gold.targets)
    assert np.array_equal(ds.patient_slot, gold.patient_slot) and np.array_equal(ds.hours, gold.hours)
    assert len(set(ds.patients.tolist())) == 12
    out = ds.save_npz(str(tmp_path / 'f.npz'))
    assert 'patients' not in np.load(out).files                                      # ids never leave by default
    assert 'patients' in np.load(ds.save_npz(str(tmp_path / 'g.npz'), anonymise=False)).files


def _fake_reference_dataset(protocol, object_ids):
    """A pickle with the shape of a py3-era k-fold ``ARDSRawDataset``: class path deepards.dataset.ARDSRawDataset,
    all_sequences of [patient, (NB, 1, L) float64, target, hours], (NB, C, L) scaling-factor broadcasts per fold,
    kfold_patient_splits of patient-id arrays (object dtype, as pandas' .unique() gives, or fixed-width str)."""
    mod = types.ModuleType('deepards.dataset')
    pkg = types.ModuleType('deepards')

    class ARDSRawDataset(object):
        pass
    ARDSRawDataset.__module__ = 'deepards.dataset'
    ARDSRawDataset.__qualname__ = 'ARDSRawDataset'
    mod.ARDSRawDataset = ARDSRawDataset
    rng = np.random.RandomState(3)
    pts = ['%04dRPI%02d' % (i, i) for i in range(6)]
    seqs = []
    for w in range(14):
        p = pts[w % 6]
        seqs.append([p, rng.randn(5, 1, 32), np.array([0., 1.]) if (w % 6) < 3 else np.array([1., 0.]),
                     [float(w) + 0.01 * k for k in range(5)]])
    d = ARDSRawDataset()
    d.all_sequences = seqs
    d.n_sub_batches, d.dataset_type, d.train, d.total_kfolds, d.kfold_num = 5, 'unpadded_centered_sequences', True, 2, 0
    mk = lambda v: np.full((5, 1, 32), v)
    d.scaling_factors = {0: (mk(0.25), mk(1.5)), 1: (mk(-0.5), mk(2.0))}
    arr = (lambda ids: np.array(ids, dtype=object)) if object_ids else (lambda ids: np.array(ids))
    d.kfold_patient_splits = {0: {'train': arr(pts[:2] + pts[3:5]), 'test': arr([pts[2], pts[5]])},
                              1: {'train': arr([pts[0], pts[2], pts[3], pts[5]]), 'test': arr([pts[1], pts[4]])}}
    d.cohort = {'not': 'looked at'}
    saved = {k: sys.modules.get(k) for k in ('deepards', 'deepards.dataset')}
    sys.modules['deepards'], sys.modules['deepards.dataset'] = pkg, mod
    try:
        blob = pickle.dumps(d, protocol=protocol)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return blob, seqs, pts


@pytest.mark.parametrize('protocol,object_ids', [(2, True), (4, True), (4, False), (5, True)])
def test_kfold_dataset_pickles_of_every_protocol_parse_inertly(protocol, object_ids, tmp_path):
    from deepards_amd import ingest
    blob, seqs, pts = _fake_reference_dataset(protocol, object_ids)
    assert 'deepards' not in sys.modules                      # the class the pickle names does NOT exist at parse time
    ds = ingest.dataset_from_tree(ingest.parse_pickle(blob))
    assert np.array_equal(ds.windows, np.stack([s[1] for s in seqs]))
    assert np.array_equal(ds.targets, np.stack([s[2] for s in seqs]).astype(np.float32))
    assert ds.patients.tolist() == [s[0] for s in seqs] and ds.n_patients == 6
    assert np.allclose(ds.hours, np.array([s[3] for s in seqs]))
    assert ds.total_kfolds == 2 and ds.train is True
    assert {k: (float(m[0]), float(s[0])) for k, (m, s) in ds.scaling_factors.items()} == {0: (0.25, 1.5), 1: (-0.5, 2.0)}
    assert ds.kfold_patient_splits[0]['test'].tolist() == [2, 5] and ds.kfold_patient_splits[1]['train'].tolist() == [0, 2, 3, 5]
    p = ds.save_npz(str(tmp_path / 'k.npz'))
    back = ingest.load_npz(p)
    assert np.array_equal(back.windows, ds.windows) and back.total_kfolds == 2
    assert back.kfold_patient_splits[1]['test'].tolist() == [1, 4]
    assert {k: float(m[0]) for k, (m, s) in back.scaling_factors.items()} == {0: 0.25, 1: -0.5}
    # the k-fold plumbing of the store keeps the pickled splits and factors (from_pickle does too)
    store = back.to_store(device='cpu')
    store.set_kfold_indexes_for_fold(1)
    assert (store.mu, store.std) == (-0.5, 2.0)
    want = [i for p_ in (0, 2, 3, 5) for i in range(14) if i % 6 == p_]
    assert store.kfold_indexes.tolist() == want
    test = store.make_test_store_if_kfold()
    test.set_kfold_indexes_for_fold(1)
    assert sorted(test.kfold_indexes.tolist()) == sorted(i for i in range(14) if i % 6 in (1, 4))
    assert (test.mu, test.std) == (-0.5, 2.0)                 # test windows are scaled with the TRAIN fold's factors


def test_inert_parser_executes_nothing(tmp_path):
    """A pickle whose reduce callables would run code under pickle.loads is parsed to inert nodes; nothing is imported
    or called (the marker file is never created, the named module is never imported)."""
    from deepards_amd import ingest
    marker = tmp_path / 'pwned'

    class Evil(object):
        def __reduce__(self):
            return (os.system, ('touch %s' % marker,))

    class Evil2(object):
        def __reduce__(self):
            return (__import__, ('deepards_amd_should_not_be_imported_xyz',))
    blob = pickle.dumps({'a': Evil(), 'b': [Evil2(), np.arange(3.0)]}, protocol=2)
    tree = ingest.parse_pickle(blob)
    assert not marker.exists() and 'deepards_amd_should_not_be_imported_xyz' not in sys.modules
    a = tree['a']
    assert isinstance(a, ingest.Obj) and a.func.name == 'system' and 'touch' in ingest._text(a.args[0])
    assert np.array_equal(ingest.resolve(tree['b'][1]), np.arange(3.0))
    # object arrays carry pickled objects: refused as array payloads
    obj_arr = pickle.dumps(np.array([Evil(), None], dtype=object), protocol=2)
    with pytest.raises(ingest.PickleFormatError):
        ingest.resolve(ingest.parse_pickle(obj_arr))
    assert not marker.exists()
    with pytest.raises(ingest.PickleFormatError):
        ingest.parse_pickle(b'\x80\x02P1\n.')                 # persistent ids are refused for datasets
    with pytest.raises(ValueError, match='out-of-date'):
        ingest.dataset_from_tree(ingest.parse_pickle(pickle.dumps({'x': 1}, protocol=2)))


# ---- k-fold + oversampling index plumbing ------------------------------------------------------------------------------
def test_random_over_sample_rules():
    """imblearn's RandomOverSampler as the reference uses it (dataset.py:571-572,580-581), restated (parity unpinned:
    the package is not installable here): originals first, then per class in label order n_extra draws with replacement
    from that class's positions."""
    from deepards_amd.tiles import random_over_sample
    x = np.arange(100, 110)
    y = np.array([0, 0, 0, 1, 1, 1, 1, 1, 1, 1])
    r = random_over_sample(x, y, None, np.random.RandomState(0))
    assert r[:10].tolist() == x.tolist() and len(r) == 14 and set(r[10:].tolist()) <= {100, 101, 102}
    rs = np.random.RandomState(0)
    assert r[10:].tolist() == (100 + rs.randint(0, 3, 4)).tolist()              # the published draw: randint(0, n_class, n_extra)
    r2 = random_over_sample(x, y, {0: 6, 1: 14}, np.random.RandomState(1))
    assert len(r2) == 20 and (np.isin(r2[10:13], [100, 101, 102])).all() and (r2[13:] >= 103).all()
    assert random_over_sample(x, y, {0: 3, 1: 7}).tolist() == x.tolist()         # nothing to add
    with pytest.raises(ValueError):
        random_over_sample(x, y, {0: 2, 1: 7})                                   # cannot shrink
    bal = random_over_sample(np.arange(6), np.array([0, 1, 0, 1, 0, 1]))         # balanced: unchanged
    assert bal.tolist() == list(range(6))


def test_store_kfold_and_oversampling_index_plumbing_on_the_fixture():
    """set_kfold_indexes_for_fold (dataset.py:765-772) on the fixture's 12 patients: patient-wise stratified folds, the
    fold's factors from its train windows, then oversampling of the minority class (only for the TRAIN store, only with
    k-folds: dataset.py:563-567)."""
    from deepards_amd import ingest
    from deepards_amd.tiles import scaling_factors_for_indices
    ds = ingest.load_npz(os.path.join(GOLD, 'test_dataset.npz'))
    ds.total_kfolds, ds.train = 2, True
    store = ds.to_store(device='cpu')
    labels = ds.targets.argmax(1)
    test = store.make_test_store_if_kfold()
    store.oversample_minority = True
    store.sampling_rng = np.random.RandomState(4)
    seen_test = []
    for k in range(2):
        store.set_kfold_indexes_for_fold(k)
        test.set_kfold_indexes_for_fold(k)
        tr, te = store.kfold_indexes.numpy(), test.kfold_indexes.numpy()
        assert not set(ds.patient_slot[tr].tolist()) & set(ds.patient_slot[te].tolist())     # patient-wise split
        assert sorted(set(tr.tolist()) | set(te.tolist())) == list(range(20))
        n0, n1 = int((labels[tr] == 0).sum()), int((labels[tr] == 1).sum())
        assert n0 == n1                                                                      # minority re-drawn up to parity
        base = np.unique(tr)
        assert tr[:len(base)].tolist() == sorted(base.tolist(), key=tr.tolist().index)       # originals first
        mu, std = scaling_factors_for_indices(ds.windows, tr[:len(base)].tolist())            # the fold's train windows, patient by patient
        assert (store.mu, store.std) == (float(mu[0]), float(std[0])) == (test.mu, test.std)
        assert len(np.unique(te)) == len(te)                                                 # the test fold is never oversampled
        seen_test += te.tolist()
    assert sorted(seen_test) == list(range(20))
    before = store.kfold_indexes.clone()
    store.set_oversampling_indices()                       # --reshuffle-oversample-per-epoch (:350-351): grows again from the current list
    assert len(store.kfold_indexes) >= len(before)
    hold = ingest.load_npz(os.path.join(GOLD, 'test_dataset.npz')).to_store(device='cpu')
    hold.train, hold.oversample_minority = True, True
    with pytest.raises(NotImplementedError, match='holdout'):
        hold.set_oversampling_indices()
    with pytest.raises(IndexError):
        hold.set_kfold_indexes([0, 25])
    with pytest.raises(IndexError):
        hold.batch([20])                                   # bounds are checked before anything reaches the gather kernel


def test_export_without_patients_refuses_patient_wise_use_and_device_indices_are_tagged(tmp_path):
    """an older .npz without patient_slot used to get invented patients (arange % 6), so 'patient-wise'
    folds put one real patient's windows in train AND test; and batch_from_device forwarded any int64 device tensor to
    the gather kernels, which read tiles[idx] unchecked."""
    from deepards_amd import ingest
    ds = ingest.load_npz(os.path.join(GOLD, 'test_dataset_windows.npz'))       # the older export: x, target, mu, std
    assert ds.patient_slot is None and ds.n_patients == 0
    store = ds.to_store(device='cpu')                                           # plain training over windows: fine
    assert store.patient_slot is None and len(store) == 20
    ds.total_kfolds = 2
    with pytest.raises(ValueError, match='patient'):
        ds.to_store(device='cpu')
    out = ds.save_npz(str(tmp_path / 'again.npz'))
    assert 'patient_slot' not in np.load(out).files
    # oversample_all_factor on a fold that lacks one class: the absent class is left out of the request
    full = ingest.load_npz(os.path.join(GOLD, 'test_dataset.npz'))
    full.total_kfolds, full.train = 2, True
    st = full.to_store(device='cpu')
    st.set_kfold_indexes_for_fold(0)
    ards_only = [i for i in st.kfold_indexes.tolist() if full.targets[i, 1] == 1]
    st.set_kfold_indexes(ards_only)
    st.oversample_all_factor = 2.0
    st.set_oversampling_indices()
    assert len(st.kfold_indexes) == 2 * len(ards_only)
    # batch_from_device accepts slices of what device_indices checked, nothing else (needs a device: shape of the rule
    # is tested with the storage registry directly)
    checked = store.device_indices(torch.arange(6))
    assert checked.untyped_storage().data_ptr() in store._checked_idx
    with pytest.raises(IndexError):
        store.device_indices(torch.tensor([0, 20]))


# ---- checkpoints ---------------------------------------------------------------------------------------------------------
sys.path.insert(0, os.path.join(ROOT, 'tests', 'tools'))
from ref_paths import as_reference_classes as _as_reference_classes      # noqa: E402


@pytest.mark.parametrize('legacy', [False, True])
@pytest.mark.parametrize('wrap', [False, True])
def test_foreign_whole_module_checkpoints_are_read_without_unpickling(tmp_path, legacy, wrap):
    """A whole module saved under the reference's class paths (zip format, or the pytorch<=1.5 legacy format its
    environment pins; optionally wrapped in nn.DataParallel, train_ards_detector.py:96,385-386) -> state_dict with
    nn.Module.state_dict's keys and values, class / backbone names; the file is never unpickled."""
    from deepards_amd import checkpoint as C
    path = str(tmp_path / 'ref.pth')

    def save(M):
        torch.manual_seed(3)
        m = M.CNNLinearNetwork(M.densenet18(), 20, 0)
        m.breath_block.features.norm0.weight.data.uniform_(0.5, 1.5)
        torch.save(torch.nn.DataParallel(m) if wrap else m, path, _use_new_zipfile_serialization=not legacy)
        return m
    m = _as_reference_classes(save)
    assert C.checkpoint_kind(path) == 'foreign'
    info = C.read_module_checkpoint(path)
    sd = m.state_dict()
    assert list(info['state_dict']) == list(sd) and all(torch.equal(info['state_dict'][k], sd[k]) for k in sd)
    assert info['class_name'] == 'deepards.models.torch_cnn_linear_network.CNNLinearNetwork'
    assert info['network_name'] == 'densenet18' and info['breath_block_class'] == 'deepards.models.densenet.DenseNet'
    # --load-checkpoint: a fresh model of the configured architecture receives the weights
    import deepards_amd.models as M
    got = C.load_model_weights(path, lambda: M.CNNLinearNetwork(M.densenet18(), 20, 0))
    assert all(torch.equal(a, b) for a, b in zip(got.state_dict().values(), sd.values()))
    # --load-base-network: the breath block named by the file, whatever --base-network says
    bb = C.load_base_network(path, M.base_networks, {'base_network': 'resnet18'})
    assert bb.network_name == 'densenet18'
    assert torch.equal(bb.features.norm0.weight, m.breath_block.features.norm0.weight)


def test_own_and_state_dict_checkpoints_and_reference_file_names(tmp_path):
    from deepards_amd import checkpoint as C
    import deepards_amd.models as M
    torch.manual_seed(1)
    m = M.CNNLinearNetwork(M.resnet18(), 20, 0)
    own, sdp = str(tmp_path / 'own.pth'), str(tmp_path / 'sd.pth')
    torch.save(m, own)
    torch.save(m.state_dict(), sdp)
    assert C.checkpoint_kind(own) == 'own' and C.checkpoint_kind(sdp) == 'state_dict'
    a = C.load_model_weights(own, lambda: None)
    b = C.load_model_weights(sdp, lambda: M.CNNLinearNetwork(M.resnet18(), 20, 0))
    for x in (a, b):
        assert all(torch.equal(p, q) for p, q in zip(x.state_dict().values(), m.state_dict().values()))
    assert C.load_base_network(own, M.base_networks).network_name == 'resnet18'
    bb = C.load_base_network(sdp, M.base_networks, {'base_network': 'resnet18'})
    assert torch.equal(bb.layer4[1].conv2.weight, m.breath_block.layer4[1].conv2.weight)
    with pytest.raises(RuntimeError):                        # a densenet file into a resnet model: strict
        torch.save(M.CNNLinearNetwork(M.densenet18(), 20, 0).state_dict(), sdp)
        C.load_model_weights(sdp, lambda: M.CNNLinearNetwork(M.resnet18(), 20, 0))
    # train_ards_detector.py:355-374
    assert C.model_save_path('runs/m.pth', 'saved', 1, 0) == os.path.join('saved', 'm.pth')
    assert C.model_save_path('runs/m.pth', 'saved', 5, 3) == os.path.join('saved', 'm-fold3.pth')
    assert C.model_save_path('m.pth', 'saved', 5, 3, epoch_num=2) == os.path.join('saved', 'm-epoch2-fold3.pth')
    assert C.model_save_path('m.pth', 'saved', 1, 0, epoch_num=7) == os.path.join('saved', 'm-epoch7.pth')


class _Boom(object):
    """Pickles as a call of os.system: what a crafted checkpoint would smuggle in."""

    def __init__(self, marker):
        self.marker = marker

    def __reduce__(self):
        return (os.system, ('touch %s' % self.marker,))


@pytest.mark.parametrize('where', ['attribute', 'buffer_dict'])
def test_spoofed_own_checkpoint_executes_nothing(tmp_path, where):
    """A file whose pickle HEAD names a deepards_amd.models class (so checkpoint_kind says 'own') but whose body
    carries a REDUCE of os.system must be refused before anything runs (the 'own' branch used to be a
    plain torch.load(weights_only=False))."""
    import pickle
    from deepards_amd import checkpoint as C
    import deepards_amd.models as M
    marker = str(tmp_path / 'pwned')
    m = M.CNNLinearNetwork(M.resnet18(), 20, 0)
    if where == 'attribute':
        m.payload = _Boom(marker)
    else:
        m.breath_block._buffers['payload'] = _Boom(marker)
    path = str(tmp_path / 'spoof.pth')
    torch.save(m, path)
    assert C.checkpoint_kind(path) == 'own'
    for load in (lambda: C.load_model_weights(path, lambda: None), lambda: C.load_base_network(path, M.base_networks)):
        with pytest.raises(pickle.UnpicklingError):
            load()
    assert not os.path.exists(marker)


@pytest.mark.skipif(not os.path.exists('/root/reference/deepards/models/resnet.py'),
                    reason='build container only: needs the reference models importable')
@pytest.mark.parametrize('legacy', [False, True])
def test_checkpoint_pickled_by_the_real_reference_classes(tmp_path, legacy):
    """The same with the REAL reference classes (imported from /root/reference, allowed: SURVEY 8c): their whole-module
    pickle is read inertly and its weights load into this package's model, key for key."""
    from deepards_amd import checkpoint as C
    import deepards_amd.models as M
    sys.path.insert(0, '/root/reference')
    try:
        from deepards.models.resnet import resnet18
        from deepards.models.torch_cnn_linear_network import CNNLinearNetwork
        torch.manual_seed(0)
        ref = CNNLinearNetwork(resnet18(), 20, 0)
        path = str(tmp_path / 'r.pth')
        torch.save(ref, path, _use_new_zipfile_serialization=not legacy)
    finally:
        sys.path.remove('/root/reference')
        for k in [k for k in sys.modules if k == 'deepards' or k.startswith('deepards.')]:
            del sys.modules[k]
    mine = C.load_model_weights(path, lambda: M.CNNLinearNetwork(M.resnet18(), 20, 0))
    assert list(mine.state_dict()) == list(ref.state_dict()) and len(mine.state_dict()) == 129
    assert all(torch.equal(a, b) for a, b in zip(mine.state_dict().values(), ref.state_dict().values()))


def test_saved_module_carries_no_trainer_gradient_views(tmp_path):
    """a trainer hangs ``_da_grad`` (a view into its flat gradient bucket) on every live Parameter;
    ``BaseTraining._save`` must not pickle them (the file doubled in size and the loaded Parameters pointed
    functional._tgt at a dead buffer), the saving model keeps them, and ``load_own_module`` strips them from older files."""
    import os
    from deepards_amd import checkpoint as C
    from deepards_amd.train_ards_detector import BaseTraining
    import deepards_amd.models as M
    torch.manual_seed(2)
    m = M.CNNLinearNetwork(M.densenet18(), 20, 0)
    bucket = torch.zeros(sum(p.numel() for p in m.parameters()))
    off = 0
    for p in m.parameters():
        p._da_grad = bucket[off:off + p.numel()].view(p.shape)
        off += p.numel()
    plain = str(tmp_path / 'plain.pth')
    torch.save(m, plain)                                       # what round 3 wrote: the views ride along
    saver = BaseTraining.__new__(BaseTraining)
    saver._dp_override = (1, 0, None)
    stripped = str(tmp_path / 'stripped.pth')
    saver._save(m, stripped)
    assert all(hasattr(p, '_da_grad') for p in m.parameters())                 # the live model keeps its destinations
    assert os.path.getsize(stripped) < 0.7 * os.path.getsize(plain)
    for path in (plain, stripped):
        loaded = C.load_own_module(path)
        assert not any(hasattr(p, '_da_grad') for p in loaded.parameters())
        assert all(torch.equal(p, q) for p, q in zip(loaded.state_dict().values(), m.state_dict().values()))
