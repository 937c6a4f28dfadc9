"""Host-side tile forming and scaling factors -- ingest-time numpy, no GPU work.

What a "(sub_batch, 1, seq_len) waveform tile" is in the ``unpadded_centered_sequences`` dataset
(reference deepards/dataset.py:1021-1081 ``get_unpadded_sequences_dataset`` with the processing function
``_unpadded_centered_processing`` :1279-1288 and the frame filter ``_should_we_drop_frame`` :1308-1321):

* whole breaths (50 Hz flow samples) are concatenated into a row until the row holds ``seq_len`` samples; the breath
  that crosses the boundary is truncated, THE REST OF IT IS DISCARDED, and the next row starts with the next breath;
* breaths shorter than 21 samples are skipped (:1044-1045);
* ``n_sub_batches`` consecutive rows make one window ``(NB, 1, seq_len)``; the window is dropped when the ventilator
  breath numbers of the breaths that went into it have more than ``int(NB * 0.5)`` gaps in total (unless the gap is
  the 16-bit counter wrapping), and the collection restarts empty either way;
* a new patient restarts the row and the window.

``scaling_factors_for_indices`` restates ``_get_scaling_factors_for_indices`` (:627-649): per-channel mean and
POPULATION standard deviation over the chosen windows, float64, two passes (the second pass centres on the first's
mean).  The reference broadcasts both to (NB, C, seq_len) arrays; ``DeviceTileStore`` keeps the per-channel scalars
and folds ``(x - mu) / std`` into the batch gather.

Neither function is pinned by a reference fixture: ``dataset.py`` does not import here (ventmap, imblearn, ...), and
``tests/test_dataset.pkl`` holds finished windows and the TRAIN fold's factors, not the raw breaths -- parity unpinned;
the tests pin the rules above on hand-built breaths.
"""
import numpy as np

MIN_BREATH_SAMPLES = 21          # dataset.py:1044
VENT_BN_FRAC_MISSING = 0.5       # dataset.py:393


def scaling_factors_for_indices(windows, indices=None):
    """windows: (N, NB, C, L) array-like (or a list of (NB, C, L) arrays); indices: the windows of the fold (None: all).
    Returns (mu, std) as float64 arrays of shape (C,) -- dataset.py:627-649."""
    idx = list(range(len(windows)) if indices is None else indices)
    if not idx:
        raise ValueError('no windows to derive scaling factors from')
    chans = np.asarray(windows[idx[0]]).shape[1]
    mean_sum = np.zeros(chans, dtype=np.float64)
    std_sum = np.zeros(chans, dtype=np.float64)
    obs_count = 0
    for i in idx:
        obs = np.asarray(windows[i], dtype=np.float64)
        obs_count += obs.shape[0] * obs.shape[-1]
        mean_sum += obs.sum(axis=-1).sum(axis=0)
    mu = mean_sum / obs_count
    for i in idx:
        obs = np.asarray(windows[i], dtype=np.float64)
        std_sum += ((obs - mu.reshape(1, chans, 1)) ** 2).sum(axis=-1).sum(axis=0)
    return mu, np.sqrt(std_sum / obs_count)


def perform_fft(windows, add_fft=False, only_fft=False, fft_real_only=False):
    """``ARDSRawDataset._perform_fft`` (dataset.py:1330-1341) on (N, NB, C, L) windows: per window
    ``trans = np.fft.fftshift(np.fft.fft(seq, axis=-1))``, channels ``[trans.real]`` (fft_real_only) or
    ``[trans.real, trans.imag]`` appended to the flow channel (add_fft) or replacing it (only_fft) along axis 1.
    Restated call for call -- INCLUDING that ``fftshift`` is given no ``axes`` there, so it rolls every axis of the
    (NB, C, L) window: the spectrum's rows end up NB // 2 sub-batch rows away from the flow rows they came from (and the
    single-channel axis is unmoved).  That is what the reference trains on, so it is what this returns.  Same numpy
    functions as the reference calls; dataset.py itself does not import here, so parity is unpinned beyond that."""
    w = np.asarray(windows, dtype=np.float64)
    if not add_fft and not only_fft:
        return w
    if w.ndim != 4:
        raise ValueError('windows must be (N, NB, C, L)')
    out = []
    for seq in w:
        trans = np.fft.fftshift(np.fft.fft(seq, axis=-1))
        fft_chans = [trans.real] if fft_real_only else [trans.real, trans.imag]
        out.append(np.concatenate(([seq] if add_fft else []) + fft_chans, axis=1))
    return np.ascontiguousarray(np.stack(out))


def should_drop_frame(seq_vent_bns, n_sub_batches, frac_missing=VENT_BN_FRAC_MISSING):
    """The vent-BN continuity rule of ``_should_we_drop_frame`` (:1308-1321; the optional autocorrelation filter
    ``drop_if_under_r2`` is off by default and not restated)."""
    bns = np.asarray(seq_vent_bns)
    if bns.size < 2:
        return False
    missing = int(np.abs(bns[:-1] + 1 - bns[1:]).sum())
    thresh = int(n_sub_batches * frac_missing)
    if missing > thresh and not abs(missing - 2 ** 16) <= thresh:
        return True
    return False


class UnpaddedCenteredTiler(object):
    """Streams breaths of ONE patient at a time into ``(NB, 1, seq_len)`` float64 windows.

    ``add_breath(flow, vent_bn, seq_hour)`` returns a finished window ``(window, hours)`` or None; ``new_patient()``
    forgets the partial row / window (dataset.py:1028-1035).  ``frames_dropped`` counts the windows the vent-BN rule
    rejected."""

    def __init__(self, n_sub_batches=20, seq_len=224):
        self.n_sub_batches, self.seq_len = int(n_sub_batches), int(seq_len)
        self.frames_dropped = 0
        self.new_patient()

    def new_patient(self):
        self.batch_arr, self.breath_arr, self.seq_vent_bns, self.batch_seq_hours = [], [], [], []

    def _process(self, flow, seq_hour):
        # _unpadded_centered_processing, dataset.py:1279-1288
        if len(flow) + len(self.breath_arr) < self.seq_len:
            self.breath_arr.extend(flow)
        else:
            remaining = self.seq_len - len(self.breath_arr)
            self.breath_arr.extend(flow[:remaining])
            self.batch_arr.append(np.array(self.breath_arr, dtype=np.float64))
            self.batch_seq_hours.append(seq_hour)
            self.breath_arr = []

    def add_breath(self, flow, vent_bn, seq_hour=0.0):
        flow = list(flow)
        if len(flow) < MIN_BREATH_SAMPLES:
            return None
        self.seq_vent_bns.append(vent_bn)
        self._process(flow, seq_hour)
        out = None
        if len(self.batch_arr) == self.n_sub_batches:
            raw = np.array(self.batch_arr)
            drop = should_drop_frame(self.seq_vent_bns, self.n_sub_batches)
            hours = self.batch_seq_hours
            self.batch_arr, self.seq_vent_bns, self.batch_seq_hours = [], [], []
            if drop:
                self.frames_dropped += 1
                self.breath_arr = []
                return None
            out = (raw.reshape(self.n_sub_batches, 1, self.seq_len), hours)
        if len(self.batch_arr) > 0 and self.breath_arr == []:      # dataset.py:1080-1081
            self.batch_seq_hours.append(seq_hour)
        return out


def tile_patient(breaths, n_sub_batches=20, seq_len=224):
    """breaths: iterable of (flow, vent_bn) or (flow, vent_bn, seq_hour) of one patient, in time order.
    Returns (windows (W, NB, 1, seq_len) float64, frames_dropped)."""
    t = UnpaddedCenteredTiler(n_sub_batches, seq_len)
    wins = []
    for b in breaths:
        r = t.add_breath(*b)
        if r is not None:
            wins.append(r[0])
    w = np.stack(wins) if wins else np.zeros((0, n_sub_batches, 1, seq_len))
    return w, t.frames_dropped


def kfold_patient_splits(patients, labels, total_kfolds, random_kfold=False):
    """Patient-wise stratified folds, ``set_kfold_patient_splits`` (dataset.py:774-791): the patients in order of first
    appearance, non-ARDS (label 0) first and ARDS (1) after them, ``StratifiedKFold(n_splits, shuffle=random_kfold)``
    over (patient, pathology).  patients: (N,) ids per window; labels: (N,) window labels (argmax of the one-hot target).
    Returns {fold: {'train': patient ids, 'test': patient ids}}."""
    from sklearn.model_selection import StratifiedKFold          # the reference's own splitter (dataset.py:787)
    patients, labels = np.asarray(patients), np.asarray(labels)
    if patients.shape != labels.shape or patients.ndim != 1:
        raise ValueError('patients and labels must be (N,) arrays')

    def uniq(mask):                                               # pandas .unique(): order of first appearance
        seen, out = set(), []
        for p in patients[mask].tolist():
            if p not in seen:
                seen.add(p)
                out.append(p)
        return out
    other, ards = uniq(labels == 0), uniq(labels == 1)
    all_patients = np.array(other + ards)
    patho = [0] * len(other) + [1] * len(ards)
    splits = {}
    kf = StratifiedKFold(n_splits=total_kfolds, shuffle=random_kfold)
    for k, (tr, te) in enumerate(kf.split(all_patients, patho)):
        splits[k] = {'train': all_patients[tr], 'test': all_patients[te]}
    return splits


def patient_map_to_loc(patients, selected):
    """Window indices of the selected patients, patient by patient (``_patient_map_to_loc``, dataset.py:811-820)."""
    patients = np.asarray(patients)
    locs = []
    for pt in selected:
        locs.extend(np.nonzero(patients == pt)[0].tolist())
    return locs


def random_over_sample(x, y, sampling_strategy=None, rng=None):
    """``imblearn.over_sampling.RandomOverSampler(sampling_strategy).fit_resample(x.reshape(-1, 1), y)[0].ravel()``
    as the reference calls it (dataset.py:571-572, 580-581), restated from the published algorithm of imbalanced-learn
    0.4.x (``RandomOverSampler._fit_resample``): the output is the input followed, class by class in ascending label
    order, by ``n_extra`` items of that class drawn with replacement -- ``rng.randint(0, n_class, n_extra)`` indexes the
    class's positions in input order.  ``sampling_strategy`` None ('auto' = 'not majority'): every class but the
    largest grows to the largest's size; a dict {label: wanted count}: each class grows to its count (never shrinks).
    ``rng``: np.random.RandomState; None uses numpy's global one like ``random_state=None`` does.
    The package is not installable here, so this restatement is NOT pinned against it (parity unpinned); the tests pin
    the rules above."""
    x, y = np.asarray(x), np.asarray(y)
    if x.shape != y.shape or x.ndim != 1:
        raise ValueError('x and y must be (n,) arrays')
    rng = np.random if rng is None else rng
    classes, counts = np.unique(y, return_counts=True)
    stats = dict(zip(classes.tolist(), counts.tolist()))
    if sampling_strategy is None:
        n_max = max(stats.values())
        majority = max(stats, key=stats.get)
        extra = {c: n_max - n for c, n in stats.items() if c != majority}
    else:
        extra = {}
        for c, want in sampling_strategy.items():
            if c not in stats:
                raise ValueError('class %r is not in y' % (c,))
            if want < stats[c]:
                raise ValueError('over-sampling cannot shrink class %r from %d to %d' % (c, stats[c], want))
            extra[c] = want - stats[c]
    idx = np.arange(len(x))
    for c in sorted(extra):
        where = np.flatnonzero(y == c)
        draw = rng.randint(low=0, high=stats[c], size=extra[c])
        idx = np.append(idx, where[draw])
    return x[idx]
