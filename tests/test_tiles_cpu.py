"""Host-side tile forming and scaling factors: the rules of dataset.py:1021-1081,
1279-1288, 1308-1321 and 627-649 on hand-built breaths (parity unpinned: the reference's dataset module does not
import here and its fixture holds finished windows only)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from deepards_amd.tiles import (UnpaddedCenteredTiler, kfold_patient_splits, patient_map_to_loc,   # noqa: E402
                                scaling_factors_for_indices, should_drop_frame, tile_patient)


def breath(n, start):
    return np.arange(start, start + n, dtype=np.float64)


def test_rows_concatenate_truncate_and_restart_on_a_fresh_breath():
    t = UnpaddedCenteredTiler(n_sub_batches=2, seq_len=10)
    # MIN_BREATH_SAMPLES is 21, so use long breaths with a short row: every breath overfills a row
    assert t.add_breath(breath(25, 0), 1) is None               # row 0 = samples 0..9 of breath 1, rest discarded
    w = t.add_breath(breath(30, 100), 2)                        # row 1 = samples 100..109 of breath 2
    assert w is not None
    win, hours = w
    assert win.shape == (2, 1, 10) and win.dtype == np.float64
    assert np.array_equal(win[0, 0], np.arange(0, 10)) and np.array_equal(win[1, 0], np.arange(100, 110))
    assert t.batch_arr == [] and t.breath_arr == []


def test_whole_breaths_fill_a_row_before_the_truncated_one():
    t = UnpaddedCenteredTiler(n_sub_batches=1, seq_len=60)
    assert t.add_breath(breath(25, 0), 1) is None               # 25 < 60: whole breath kept
    assert t.add_breath(breath(25, 100), 2) is None             # 50 < 60
    win, _ = t.add_breath(breath(25, 200), 3)                   # 75 >= 60: first 10 samples of breath 3, rest dropped
    assert np.array_equal(win[0, 0], np.concatenate([np.arange(0, 25), np.arange(100, 125), np.arange(200, 210)]))
    assert t.add_breath(breath(22, 300), 4) is None             # the next row starts with breath 4, not breath 3's tail
    assert t.breath_arr[:3] == [300.0, 301.0, 302.0]


def test_exact_fit_closes_the_row():
    t = UnpaddedCenteredTiler(n_sub_batches=1, seq_len=50)
    assert t.add_breath(breath(25, 0), 1) is None
    win, _ = t.add_breath(breath(25, 100), 2)                   # 25 + 25 == 50 is not < 50: row closes
    assert win.shape == (1, 1, 50) and win[0, 0, 25] == 100.0


def test_short_breaths_are_skipped_and_do_not_count_as_vent_bns():
    t = UnpaddedCenteredTiler(n_sub_batches=1, seq_len=30)
    assert t.add_breath(breath(20, 0), 1) is None and t.breath_arr == [] and t.seq_vent_bns == []
    assert t.add_breath(breath(21, 0), 2) is None and len(t.breath_arr) == 21


def test_vent_bn_gap_rule():
    assert not should_drop_frame([1, 2, 3, 4], 20)
    assert not should_drop_frame([1, 2, 13, 14], 20)            # 10 missing == threshold int(20 * .5): kept
    assert should_drop_frame([1, 2, 14, 15], 20)                # 11 missing: dropped
    assert not should_drop_frame([65535, 0, 1], 20)             # the 16-bit counter wrapping is not a gap
    t = UnpaddedCenteredTiler(n_sub_batches=2, seq_len=10)
    t.add_breath(breath(25, 0), 1)
    assert t.add_breath(breath(25, 50), 40) is None and t.frames_dropped == 1
    assert t.batch_arr == [] and t.breath_arr == [] and t.seq_vent_bns == []


def test_new_patient_forgets_the_partial_window():
    t = UnpaddedCenteredTiler(n_sub_batches=2, seq_len=10)
    t.add_breath(breath(25, 0), 1)
    t.new_patient()
    assert t.batch_arr == [] and t.breath_arr == []
    wins, dropped = tile_patient([(breath(25, 10 * i), i) for i in range(5)], n_sub_batches=2, seq_len=10)
    assert wins.shape == (2, 2, 1, 10) and dropped == 0         # 5 rows -> 2 windows, the fifth row stays open


def test_default_geometry_is_the_hot_path_tile():
    rng = np.random.default_rng(0)
    breaths = [(rng.standard_normal(int(rng.integers(60, 220))), i) for i in range(200)]
    wins, dropped = tile_patient(breaths)
    assert wins.shape[1:] == (20, 1, 224) and wins.shape[0] >= 2 and dropped == 0


def test_scaling_factors_are_the_population_moments_of_the_fold():
    rng = np.random.default_rng(1)
    wins = rng.standard_normal((7, 20, 1, 224)) * 28.0 + 2.0
    mu, std = scaling_factors_for_indices(wins)
    assert mu.shape == (1,) and abs(mu[0] - wins.mean()) < 1e-12 and abs(std[0] - wins.std()) < 1e-10
    mu2, std2 = scaling_factors_for_indices(wins, [0, 3, 4])
    sel = wins[[0, 3, 4]]
    assert abs(mu2[0] - sel.mean()) < 1e-12 and abs(std2[0] - sel.std()) < 1e-10
    two = rng.standard_normal((3, 20, 2, 224)) * np.array([1.0, 5.0]).reshape(1, 1, 2, 1)
    mu3, std3 = scaling_factors_for_indices(list(two))
    assert mu3.shape == (2,) and np.allclose(std3, two.transpose(2, 0, 1, 3).reshape(2, -1).std(axis=1))
    with pytest.raises(ValueError):
        scaling_factors_for_indices(wins, [])


def test_fixture_windows_normalise_with_their_fold_factors():
    """The reference fixture's windows with factors derived from themselves: zero mean, unit variance after (x-mu)/std."""
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'test_dataset_windows.npz'))
    mu, std = scaling_factors_for_indices(z['x'])
    n = (z['x'] - mu[0]) / std[0]
    assert abs(n.mean()) < 1e-12 and abs(n.std() - 1.0) < 1e-12


def test_kfold_patient_splits_are_patientwise_and_stratified():
    """set_kfold_patient_splits (dataset.py:774-791): folds split PATIENTS (never windows of one patient), stratified by
    pathology, non-ARDS patients listed first; deterministic without random_kfold."""
    rng = np.random.default_rng(5)
    pts = np.repeat(np.arange(20), rng.integers(3, 9, 20))            # 20 patients, 3..8 windows each
    rng.shuffle(pts)
    label_of = (np.arange(20) % 2 == 0).astype(int)                   # 10 ARDS, 10 other
    labels = label_of[pts]
    sp = kfold_patient_splits(pts, labels, 5)
    assert sorted(sp) == [0, 1, 2, 3, 4]
    seen_test = []
    for k in range(5):
        tr, te = set(sp[k]['train'].tolist()), set(sp[k]['test'].tolist())
        assert not (tr & te) and len(tr | te) == 20 and len(te) == 4
        assert sum(label_of[p] for p in te) == 2                      # 2 ARDS + 2 other in every test fold
        seen_test += sorted(te)
        loc = patient_map_to_loc(pts, sp[k]['test'])
        assert set(pts[loc].tolist()) == te and len(loc) == int(np.isin(pts, list(te)).sum())
    assert sorted(seen_test) == list(range(20))                       # every patient tested exactly once
    sp2 = kfold_patient_splits(pts, labels, 5)
    assert all(np.array_equal(sp[k]['test'], sp2[k]['test']) for k in range(5))
