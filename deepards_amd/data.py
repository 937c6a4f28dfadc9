"""Device-resident window store.

The reference feeds every step through ``ARDSRawDataset.__getitem__`` (dataset.py:1343-1404: k-fold relative ->
absolute index, ``(data - mu) / std`` in float64), default_collate, a float64->float32 cast and a synchronous
H2D copy (train_ards_detector.py:144-152, ``num_workers=0``).  On MI355X the whole dataset fits in HBM many
times over (a 24 h recording is ~36 KB per window), so the raw float64 windows are uploaded ONCE and a batch is
one HIP kernel: gather by index fused with the normalisation and the cast -- bit-identical values, no host work
per step.  Index plumbing (k-fold index lists, oversampling, shuffling) stays plain host/torch code.
"""
import torch

from . import hip_ops as H


class DeviceTileStore(object):
    def __init__(self, windows, targets, mu, std, device='cuda'):
        """windows: (N, NB, C, L) float64 array-like of RAW (un-normalised) windows -- C = 1: flow; C = 2 / 3: with the
        spectrum channels of ``tiles.perform_fft`` (dataset.py:1330-1341); targets (N, 2) one-hot; mu, std: this
        fold's scaling factors (dataset.py:627-649): scalars for one channel, one per channel otherwise."""
        w = torch.as_tensor(windows, dtype=torch.float64)
        if w.dim() != 4 or not 1 <= w.shape[2] <= 4:
            raise ValueError('windows must be (N, NB, C <= 4, L)')
        self.tiles = w.contiguous().to(device)
        self.targets = torch.as_tensor(targets, dtype=torch.float32).contiguous().to(device)
        if self.targets.shape != (w.shape[0], 2):
            raise ValueError('targets must be (N, 2) one-hot')
        self.mu, self.std = self._factors(mu, w.shape[2]), self._factors(std, w.shape[2])
        self.kfold_indexes = None                     # absolute indices of the current fold (dataset.py:765-772)
        self.patients = self.total_kfolds = self.kfold_patient_splits = self.scaling_factors = None
        self.kfold_num = None
        self.train = True
        # sampling knobs of ARDSRawDataset (dataset.py:360-372); only the oversampling ones are on the hot path
        self.oversample_minority = False
        self.oversample_all_factor = 1.0
        self.undersample_factor = -1
        self.train_patient_fraction = 1.0
        self.sampling_rng = None                      # np.random.RandomState for the oversampler; None: numpy's global RNG
        self.hours = None                             # (N, NB) seq_hours of the windows when ingested from a pickle
        self.patient_slot = None                      # (N,) patient slot per window when ingested from a pickle

    @staticmethod
    def _factors(v, chans):
        """A fold's mu (or std) as the gather kernel takes it: a float for one-channel windows, a tuple per channel."""
        import numpy as np
        a = np.ravel(np.asarray(v, dtype=np.float64))
        if a.size != chans:
            raise ValueError('%d scaling factors for %d channels' % (a.size, chans))
        return float(a[0]) if chans == 1 else tuple(float(q) for q in a)

    # ---- k-fold plumbing of ARDSRawDataset (dataset.py:651-670, 672-700, 765-830) --------------------------------
    def enable_kfolds(self, patients, total_kfolds, train=True, random_kfold=False, splits=None, scaling_factors=None):
        """Patient-wise stratified folds over this store's windows; per-fold scaling factors from each fold's TRAIN
        windows (``derive_scaling_factors``).  ``set_kfold_indexes_for_fold(k)`` then selects fold k's train (or, with
        train=False, test) windows and that fold's factors.  ``splits`` / ``scaling_factors``: the ones a dataset
        pickle already carries are kept instead of derived, as ``from_pickle`` does (``set_kfold_patient_splits`` only
        fills an empty dict, dataset.py:774-781; factors are re-derived only for new FFT channels, :757-762)."""
        from .tiles import kfold_patient_splits, patient_map_to_loc, scaling_factors_for_indices
        import numpy as np
        self.patients = np.asarray(patients)
        if self.patients.shape != (self.tiles.shape[0],):
            raise ValueError('one patient id per window expected')
        labels = self.targets.argmax(dim=1).cpu().numpy()
        self.total_kfolds, self.train = int(total_kfolds), bool(train)
        if splits:
            self.kfold_patient_splits = {int(k): {'train': np.asarray(v['train']), 'test': np.asarray(v['test'])}
                                         for k, v in splits.items()}
            if sorted(self.kfold_patient_splits) != list(range(self.total_kfolds)):
                raise ValueError('the given patient splits do not cover folds 0..%d' % (self.total_kfolds - 1))
        else:
            self.kfold_patient_splits = kfold_patient_splits(self.patients, labels, self.total_kfolds, random_kfold)
        if scaling_factors:
            c = self.tiles.shape[2]
            self.scaling_factors = {int(k): (self._factors(m, c), self._factors(s_, c)) for k, (m, s_) in scaling_factors.items()}
        else:
            host = self.tiles.cpu().numpy()
            self.scaling_factors = {}
            for k, sp in self.kfold_patient_splits.items():
                mu, std = scaling_factors_for_indices(host, patient_map_to_loc(self.patients, sp['train']))
                self.scaling_factors[k] = (self._factors(mu, len(mu)), self._factors(std, len(std)))
        return self

    def make_test_store_if_kfold(self):
        """``ARDSRawDataset.make_test_dataset_if_kfold`` (dataset.py:672-700): the same windows, splits and TRAIN-fold
        scaling factors, serving the test patients of each fold (no copy of the device tiles)."""
        if self.total_kfolds is None:
            raise ValueError('enable_kfolds first')
        other = object.__new__(DeviceTileStore)
        other.__dict__.update(self.__dict__)
        other.train, other.kfold_indexes = False, None
        other.oversample_minority, other.oversample_all_factor = False, 1.0      # dataset.py:689-690
        return other

    def get_kfold_indexes_for_fold(self, kfold_num):
        from .tiles import patient_map_to_loc
        sp = self.kfold_patient_splits[kfold_num]
        return patient_map_to_loc(self.patients, sp['train'] if self.train else sp['test'])

    def set_kfold_indexes_for_fold(self, kfold_num):
        if self.total_kfolds is None:
            raise ValueError('enable_kfolds first')
        self.kfold_num = kfold_num
        self.set_kfold_indexes(self.get_kfold_indexes_for_fold(kfold_num))
        self.mu, self.std = self.scaling_factors[kfold_num]
        # dataset.py:768-772: fractional patients, undersampling, then oversampling
        if self.train_patient_fraction != 1.0:
            raise NotImplementedError('train_pt_frac != 1.0 is outside the accelerated hot path')
        if self.undersample_factor != -1:
            raise NotImplementedError('patient-level undersampling is outside the accelerated hot path')
        self.set_oversampling_indices()

    def set_oversampling_indices(self):
        """``ARDSRawDataset.set_oversampling_indices`` (dataset.py:561-582) on the current fold's index list: with
        ``oversample_minority`` the minority class's windows are re-drawn with replacement until the classes are even;
        with ``oversample_all_factor`` > 1 both classes grow to ``int(n_class * factor)``.  The reference delegates to
        ``imblearn.over_sampling.RandomOverSampler`` (imbalanced-learn, pinned 0.4.3 in environment-py2.yml, absent
        here); ``tiles.random_over_sample`` restates its published algorithm -- parity unpinned."""
        from .tiles import random_over_sample
        if not self.train:
            return                                               # "Cannot oversample with testing set"
        if self.oversample_minority and not self.total_kfolds:
            raise NotImplementedError('We havent implemented oversampling for holdout sets yet')
        if not self.oversample_minority and not self.oversample_all_factor > 1.0:
            return
        if self.kfold_indexes is None:
            raise ValueError('set_kfold_indexes_for_fold first')
        labels = self.targets.argmax(dim=1).cpu().numpy()
        x = self.kfold_indexes.cpu().numpy()
        if self.oversample_minority:
            x = random_over_sample(x, labels[x], None, self.sampling_rng)
        if self.oversample_all_factor > 1.0:
            y = labels[x]
            # (a fold that lacks one class: that class is left out of the request instead of asking for 0 items of a
            #  class that is not there -- imbalanced-learn raises its own error for the reference in that case)
            want = {c: int((y == c).sum() * self.oversample_all_factor) for c in (0, 1) if (y == c).any()}
            x = random_over_sample(x, y, want, self.sampling_rng)
        self.set_kfold_indexes(x)

    @classmethod
    def with_derived_scaling(cls, windows, targets, indices=None, device='cuda'):
        """Store whose (mu, std) are derived from the windows `indices` (the train fold; None: all) the way
        ``derive_scaling_factors`` / ``_get_scaling_factors_for_indices`` do (dataset.py:627-673)."""
        from .tiles import scaling_factors_for_indices
        mu, std = scaling_factors_for_indices(windows, indices)
        return cls(windows, targets, mu, std, device=device)

    def __len__(self):
        return self.tiles.shape[0] if self.kfold_indexes is None else len(self.kfold_indexes)

    def set_kfold_indexes(self, indexes):
        """Relative -> absolute index map of the current fold (reference: set_kfold_indexes_for_fold)."""
        if indexes is None:
            self.kfold_indexes = None
            return
        idx = torch.as_tensor(indexes, dtype=torch.int64).reshape(-1)
        if idx.numel() and (int(idx.min()) < 0 or int(idx.max()) >= self.tiles.shape[0]):
            raise IndexError('fold index out of range [0, %d)' % self.tiles.shape[0])
        self.kfold_indexes = idx.to(self.tiles.device)

    def batch(self, rel_idx, out=None):
        """(inputs (B, NB, 1, L) float32, targets (B, 2) float32) for fold-relative indices; out = (x, t) buffers to
        fill in place (HotPathTrainer.static_batch(): the captured step then needs no input copies)."""
        idx = torch.as_tensor(rel_idx, dtype=torch.int64)
        if idx.dim() != 1:
            raise ValueError('rel_idx must be a 1-D index list')
        # the gather kernels read tiles[idx] unchecked: bounds are enforced here, on the host copy when there is one
        # (one min/max over <= batch_size integers) and by a device-side check otherwise
        if idx.numel() and (int(idx.min()) < 0 or int(idx.max()) >= len(self)):
            raise IndexError('window index out of range [0, %d)' % len(self))
        idx = idx.to(self.tiles.device).contiguous()
        if self.kfold_indexes is not None:
            idx = self.kfold_indexes[idx].contiguous()
        ox, ot = out if out is not None else (None, None)
        return H.gather_normalize(self.tiles, idx, self.mu, self.std, out=ox), H.gather_rows(self.targets, idx, out=ot)

    def device_indices(self, rel_idx):
        """Absolute window indices ON THE DEVICE for a whole list of fold-relative indices (an epoch's permutation): one
        bounds check and one upload per epoch instead of one per batch -- a per-batch host-to-device copy of the
        indices makes the host wait on the stream every step.  Slices of the result go to ``batch_from_device``."""
        idx = torch.as_tensor(rel_idx, dtype=torch.int64)
        if idx.dim() != 1:
            raise ValueError('rel_idx must be a 1-D index list')
        if idx.numel() and (int(idx.min()) < 0 or int(idx.max()) >= len(self)):
            raise IndexError('window index out of range [0, %d)' % len(self))
        idx = idx.to(self.tiles.device)
        if self.kfold_indexes is not None:
            idx = self.kfold_indexes[idx]
        idx = idx.contiguous()
        # remember the buffer: batch_from_device only accepts slices of index tensors that were checked here.  STRONG
        # references (the last few epochs' lists, a few KB each): callers keep only slices, and under inference_mode a
        # slice does not keep its base alive -- a weak reference would refuse valid slices there.
        live = getattr(self, '_checked_idx', None)
        if live is None:
            live = self._checked_idx = {}
        while len(live) >= 8:
            del live[next(iter(live))]
        live[idx.untyped_storage().data_ptr()] = idx
        return idx

    def batch_from_device(self, abs_idx, out=None):
        """``batch`` for a contiguous int64 DEVICE tensor of absolute indices: a slice of what ``device_indices`` returned
        (bounds-checked there once per epoch).  The gather kernels read ``tiles[idx]`` unchecked, so any other tensor is
        refused: the check is on the storage the slice lives in."""
        if not (abs_idx.is_cuda and abs_idx.dtype == torch.int64 and abs_idx.dim() == 1 and abs_idx.is_contiguous()):
            raise ValueError('batch_from_device: a contiguous 1-D int64 device tensor from device_indices() expected')
        ref = getattr(self, '_checked_idx', {}).get(abs_idx.untyped_storage().data_ptr()) if abs_idx.numel() else True
        if ref is None:
            raise ValueError('batch_from_device: indices must be (a slice of) a tensor returned by device_indices(), which '
                             'checks them against the store; use batch() for anything else')
        ox, ot = out if out is not None else (None, None)
        return (H.gather_normalize(self.tiles, abs_idx, self.mu, self.std, out=ox),
                H.gather_rows(self.targets, abs_idx, out=ot))

    def epoch(self, batch_size, shuffle=True, generator=None, drop_odd=True):
        """Iterate one epoch like DataLoader(batch_size, shuffle) + clip_odd_batch_sizes (:146-147,482-494)."""
        n = len(self)
        order = torch.randperm(n, generator=generator) if shuffle else torch.arange(n)
        for s in range(0, n, batch_size):
            idx = order[s:s + batch_size]
            if drop_odd and batch_size != 1 and len(idx) % 2 == 1:
                idx = idx[:-1]
            if len(idx) == 0:
                continue
            x, t = self.batch(idx)
            yield idx, x, t
