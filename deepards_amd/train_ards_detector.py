"""Host mirror of the reference training driver for the cnn_linear path: same command line, configuration merge,
class / method / argument names as ``deepards/train_ards_detector.py`` (``build_parser`` / ``main`` :1439-1590,
``network_map`` :1410-1436, ``base_networks`` :45-69, ``BaseTraining`` :73-512, ``PatientClassifierMixin`` :514-537,
``CNNLinearModel`` :925-939, values of ``defaults.yml``), so that

    python -m deepards_amd.train_ards_detector -co experiment_files/unpadded_centered_nb20_cnn_linear.yml \\
           --train-from-pickle <dataset.pkl | dataset.npz> [--base-network resnet18] ...

or, from Python,

    cls = network_map[args.network](args)        # train_ards_detector.py:1589-1590
    cls.train_and_test()

runs the hot path on MI355X.  What differs, and why:

* the dataset objects are :class:`deepards_amd.data.DeviceTileStore` (windows resident in HBM) instead of
  ``ARDSRawDataset`` + ``DataLoader``; a "loader" is ``(store, batch_size, shuffle)``.  ``--train-from-pickle`` /
  ``--test-from-pickle`` take the reference's dataset pickle -- read WITHOUT unpickling by ``deepards_amd.ingest`` -- or
  the ``.npz`` that module exports; callers may also hand stores in (``args.train_store`` / ``args.test_store``).
  Building a dataset from raw ventilator files (``-dp``) needs ventmap / the cohort tree and is out of scope;
* ``get_optimizer`` returns a :class:`deepards_amd.train.HotPathTrainer`: the clamp hooks of ``get_model`` (:474-476),
  SGD-Nesterov / Adam (:416-422) and ``zero_grad`` are one fused kernel at the end of the captured step, so
  ``handle_train_optimization`` is a single ``train_step``;
* multi-GPU is one process per GPU under ``torch.distributed`` (launch with ``python -m torch.distributed.run``), not
  ``nn.DataParallel`` (:96): every rank takes its window shard of each batch (``deepards_amd.train``);
* results: the loss meters and the per-patient vote aggregation of ``DeepARDSResults`` (metrics.py:142-153,572-604) are
  kept on the device and read back once per epoch (``self.results`` is a small dict-of-lists recorder, not the
  reference's reporting / plotting class, which is out of scope).

There is no CPU path: constructing a model class with neither ``--cuda`` nor ``--cuda-no-dp`` raises.
"""
import argparse
import os

import torch

from . import models as M
from .config import Configuration
from .train import HotPathTrainer, run_test_epoch, run_train_epoch_from_store, test_epoch_steps

base_networks = M.base_networks
saved_models_default_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'saved_models')

# Knobs this build adds below defaults.yml (the reference has none of them) + the store_true switches of build_parser,
# which must not live in defaults.yml (defaults.yml:9) and read as off / None when nobody set them.
BUILD_DEFAULTS = dict(
    train_store=None, test_store=None, test_patient_slot=None, use_graph=True, seed=None, conv_dtype=None, act_dtype=None,
    cuda=None, cuda_no_dp=None, no_print_progress=None, print_progress=None, no_test_after_epochs=None, debug=None,
    save_model_per_epoch=None, no_train=None, resnet_double_conv=None, bm_to_linear=None, unshuffled=None,
    oversample_minority=None, reshuffle_oversample_per_epoch=None, freeze_base_network=None, stop_on_loss=None,
    clip_grad=None, with_fft=None, only_fft=None, fft_real_only=None, random_kfold=None, bootstrap=None,
    kfolds=None, only_fold=None, load_checkpoint=None, load_base_network=None, save_model=None, saved_models_dir=None,
    train_from_pickle=None, train_to_pickle=None, test_from_pickle=None, test_to_pickle=None,
    experiment_name='deepards_amd', config_override=None, folds_in_flight=None, fold_groups=None,
)

# make_args(): the merged view with every reference default, for callers that build ``args`` in Python
DEFAULTS = dict(
    network='cnn_linear', epochs=10, batch_size=16, base_network='densenet18', loader_threads=0,
    initial_planes=64, resnet_first_pool_type='max', resnet_double_conv=False,
    optimizer='sgd', dataset_type='unpadded_centered_sequences', learning_rate=0.001, n_sub_batches=20,
    weight_decay=0.0001, loss_func='bce', clip_grad=False, clip_val=0.01, time_series_hidden_units=16,
    with_fft=False, only_fft=False, fft_real_only=False, freeze_base_network=False,
    kfolds=None, bootstrap=False, random_kfold=False, only_fold=None, unshuffled=False, no_train=False,
    no_test_after_epochs=False, debug=False, cuda=True, cuda_no_dp=False, cuda_device=0, load_checkpoint=None,
    load_base_network=None, save_model=None, save_model_per_epoch=False, saved_models_dir=None,
    no_print_progress=True, print_progress=False, experiment_name='deepards_amd',
    oversample_minority=False, oversample_all_factor=1.0, reshuffle_oversample_per_epoch=False,
    undersample_factor=-1, train_pt_frac=1.0, train_from_pickle=None, test_from_pickle=None, train_to_pickle=None,
    test_to_pickle=None, stop_on_loss=False, stop_thresh=1.5, stop_after_epoch=1,
    train_store=None, test_store=None, test_patient_slot=None, use_graph=True, seed=None, conv_dtype=None, act_dtype=None,
)


def make_args(**overrides):
    """An ``args`` namespace with the reference's attribute names (``Configuration`` of config.py merges CLI,
    overrides file and defaults.yml the same way: later wins)."""
    unknown = set(overrides) - set(DEFAULTS)
    if unknown:
        raise TypeError('unknown arguments: %s' % sorted(unknown))
    d = dict(DEFAULTS)
    d.update(overrides)
    return argparse.Namespace(**d)


class Results(object):
    """The slice of ``DeepARDSResults`` the training loop writes: named meters per fold and the patient vote tables."""

    def __init__(self):
        self.meters = {}
        self.patient_results = {}

    def update_meter(self, name, fold_num, value):
        self.meters.setdefault((name, fold_num), []).append(value)

    def get_meter(self, name, fold_num):
        vals = self.meters.get((name, fold_num), [])
        return [float(v) for v in vals]                       # host sync only when somebody looks


def _flag(args, name, default=False):
    v = getattr(args, name, default)
    return default if v is None else v


class BaseTraining(object):
    clip_odd_batches = False

    def __init__(self, args):
        self.args = args
        legacy = getattr(args, 'oversample', None)
        if legacy is not None:
            args.oversample_minority = legacy             # older configuration files say `oversample` (:80-83)
        if not (_flag(args, 'cuda') or _flag(args, 'cuda_no_dp')):
            raise RuntimeError('deepards_amd runs the hot path on an MI355X only: pass --cuda or --cuda-no-dp (no CPU fallback)')
        if not torch.cuda.is_available():
            raise RuntimeError('no HIP device visible; there is no CPU fallback')
        self.device = torch.device('cuda', args.cuda_device if _flag(args, 'cuda_no_dp') else torch.cuda.current_device())
        self.cuda_wrapper = lambda x: x.to(self.device)
        self.model_cuda_wrapper = lambda x: x.to(self.device)   # one process per GPU; nn.DataParallel (:96) is not used
        self.set_loss_criterion()
        self.n_metadata_inputs = 9 if args.dataset_type == 'padded_breath_by_breath_with_flow_time_features' else 0
        if _flag(args, 'unshuffled') and args.batch_size > 1:
            raise Exception('Currently we can only run unshuffled runs with a batch size of 1!')
        if _flag(args, 'bootstrap'):
            raise NotImplementedError('--bootstrap (80/20 patient resampling, dataset.py:792-807) is outside the hot path')
        if _flag(args, 'stop_on_loss'):
            raise NotImplementedError('--stop-on-loss drops into an interactive shell in the reference (:155-157)')
        self.n_kfolds = args.kfolds if args.kfolds else 1
        self._prev_dtypes = None
        if _flag(args, 'conv_dtype', None):
            from . import functional as F_
            # the arithmetic / storage switches are process-wide: remember what was set so that restore_dtypes() (called
            # when train_and_test finishes, and by evaluate.main) leaves a later run in this process what it had before
            self._prev_dtypes = (F_.conv_dtype(), F_.storage_dtype())
            F_.set_conv_dtype(args.conv_dtype)
            # BASELINE's bf16 configs: bf16 storage too where the network has bf16 kernels throughout (the ResNets)
            if args.conv_dtype == 'bf16' and str(args.base_network).startswith('resnet') and \
                    getattr(args, 'act_dtype', None) != 'f32':
                F_.set_storage_dtype('bf16')
        self.results = Results()
        self.preds, self.pred_idx = [], []

    # ---- model / optimizer ---------------------------------------------------------------------------------------
    def _base_network_kwargs(self):
        a = self.args
        return dict(resnet_kwargs=dict(initial_planes=a.initial_planes, first_pool_type=a.resnet_first_pool_type,
                                       double_conv_first=_flag(a, 'resnet_double_conv')),
                    densenet_kwargs=dict(with_fft=_flag(a, 'with_fft'), only_fft=_flag(a, 'only_fft'),
                                         fft_real_only=_flag(a, 'fft_real_only')),
                    base_network=a.base_network)

    def get_base_network(self):
        """:380-414 for the backbones this package builds.  ``--load-base-network`` takes the ``breath_block`` of a
        saved model -- a whole module pickled by this package, by the reference (read without unpickling) or a
        state_dict (``deepards_amd.checkpoint``)."""
        a = self.args
        kw = self._base_network_kwargs()
        if a.load_base_network:
            from .checkpoint import load_base_network
            base_network = load_base_network(a.load_base_network, base_networks, kw)
        elif a.base_network.startswith('resnet'):
            base_network = base_networks[a.base_network](**kw['resnet_kwargs'])
        else:
            base_network = base_networks[a.base_network](**kw['densenet_kwargs'])
        if _flag(a, 'freeze_base_network'):
            for p in base_network.parameters():
                p.requires_grad = False
        return base_network

    def get_model(self):
        """:467-477.  The +-clip_val clamp the reference registers as a hook on every trainable parameter is applied by
        the fused optimizer kernel of the trainer ``get_optimizer`` returns (after the gradient all-reduce when data
        parallel), so no hooks are registered here.  Under data parallelism the replicas are made identical by the
        trainer's rank-0 broadcast before the first update, whatever each rank's seed was."""
        if self.args.load_checkpoint:
            from .checkpoint import load_model_weights
            model = load_model_weights(self.args.load_checkpoint, lambda: self.get_network(self.get_base_network()))
        else:
            if self.args.seed is not None:
                torch.manual_seed(self.args.seed)
            model = self.get_network(self.get_base_network())
        return self.model_cuda_wrapper(model)

    def get_optimizer(self, model, world_size=1, rank=0, process_group=None):
        """:416-422: Adam(lr) or SGD(lr, momentum .9, weight_decay, nesterov) + the clamp of get_model, as the trainer
        that owns the captured step."""
        a = self.args
        if a.optimizer not in ('adam', 'sgd'):
            raise ValueError('optimizer must be adam or sgd')
        return HotPathTrainer(model, optimizer=a.optimizer, learning_rate=a.learning_rate, weight_decay=a.weight_decay,
                              clip_grad=bool(_flag(a, 'clip_grad')), clip_val=a.clip_val, world_size=world_size, rank=rank,
                              process_group=process_group, use_graph=_flag(a, 'use_graph', True))

    # ---- data ----------------------------------------------------------------------------------------------------
    def get_base_datasets(self):
        """:189-315 for prepared datasets: stores handed in by the caller, or ``--train-from-pickle`` (+ optional
        ``--test-from-pickle``) read by ``deepards_amd.ingest`` (the reference's pickle, parsed without executing it, or
        its .npz export).  K-fold runs take the test patients of each fold from the train dataset
        (``make_test_dataset_if_kfold`` :272-273); a holdout test pickle gets the TRAIN set's scaling factors (:285)."""
        a = self.args
        if a.train_store is not None and a.test_store is not None:
            return a.train_store, a.test_store
        if not a.train_from_pickle:
            raise ValueError('no dataset: pass --train-from-pickle <dataset.pkl|.npz> (or args.train_store / args.test_store); '
                             'building one from raw ventilator files (--data-path) is outside the accelerated path')
        from .ingest import load_dataset
        fft = dict(add_fft=bool(_flag(a, 'with_fft')), only_fft=bool(_flag(a, 'only_fft')),
                   fft_real_only=bool(_flag(a, 'fft_real_only')))
        ds = load_dataset(a.train_from_pickle).with_fft(**fft)            # dataset.py:743-762
        if a.kfolds is not None and ds.total_kfolds is None:
            ds.total_kfolds = a.kfolds                            # an unfolded pickle split now (:126-133 reads args.kfolds)
        ds.train = True
        train = ds.to_store(self.device, random_kfold=bool(_flag(a, 'random_kfold')))
        train.oversample_minority = bool(_flag(a, 'oversample_minority'))
        train.oversample_all_factor = float(getattr(a, 'oversample_all_factor', 1.0) or 1.0)
        train.undersample_factor = getattr(a, 'undersample_factor', -1)
        train.train_patient_fraction = getattr(a, 'train_pt_frac', 1.0)
        if a.seed is not None:
            import numpy as np
            train.sampling_rng = np.random.RandomState(a.seed)
        self.n_sub_batches = ds.n_sub_batches
        if a.train_to_pickle:
            ds.save_npz(a.train_to_pickle)
        if not a.test_from_pickle and a.kfolds is not None:
            test = train.make_test_store_if_kfold()
        elif a.test_from_pickle:
            tds = load_dataset(a.test_from_pickle).with_fft(**fft)
            tds.train = False
            test = tds.to_store(self.device)
            test.mu, test.std = train.mu, train.std               # test_dataset.scaling_factors = train_dataset's (:285)
            test.scaling_factors = train.scaling_factors
            if a.test_to_pickle:
                tds.save_npz(a.test_to_pickle)
        else:
            raise ValueError('a holdout run needs --test-from-pickle (or --kfolds)')
        if a.test_patient_slot is None and test.patient_slot is not None:
            a.test_patient_slot = torch.as_tensor(test.patient_slot, dtype=torch.int64)
        return train, test

    def get_splits(self):
        """:317-338: per fold, (train_dataset, train_loader, test_dataset, test_loader); a loader is the tuple
        (store, batch_size, shuffle) the epoch functions iterate on the device."""
        train_dataset, test_dataset = self.get_base_datasets()
        for i in range(self.n_kfolds):
            if self.args.kfolds is not None:
                self._seed_fold_sampler(train_dataset, i)
                for ds in (train_dataset, test_dataset):
                    if hasattr(ds, 'set_kfold_indexes_for_fold'):
                        ds.set_kfold_indexes_for_fold(i)
            shuffle = not _flag(self.args, 'unshuffled')
            yield (train_dataset, (train_dataset, self.args.batch_size, shuffle),
                   test_dataset, (test_dataset, self.args.batch_size, shuffle))

    def _seed_fold_sampler(self, store, fold_num):
        """Each fold's oversampling draws come from a generator of their own (seeded from --seed and the fold; numpy's
        global RNG without a seed, like the reference's RandomOverSampler()): a fold then draws the same windows
        whether the folds run one after the other or side by side (--folds-in-flight)."""
        if hasattr(store, 'sampling_rng') and self.args.seed is not None:
            import numpy as np
            store.sampling_rng = np.random.RandomState(self.args.seed + 7919 * (fold_num + 1))

    # ---- epochs --------------------------------------------------------------------------------------------------
    def run_train_epoch(self, model, train_loader, optimizer, epoch_num, fold_num):
        """:139-159.  ``optimizer`` is the trainer from get_optimizer; every batch is one gather kernel + one graph
        replay; the per-batch losses stay on the device until a meter is read.  The permutation comes from the seed when
        one is given; data-parallel ranks always shard ONE permutation (rank 0's seed is broadcast per epoch)."""
        store, batch_size, shuffle = train_loader
        if optimizer.model is not model:
            raise ValueError('optimizer was built for another model')
        gen = None
        if self.args.seed is not None:
            gen = torch.Generator().manual_seed(self.args.seed + 1000 * fold_num + epoch_num)
        for loss in run_train_epoch_from_store(optimizer, store, batch_size=batch_size, shuffle=shuffle, generator=gen):
            self.results.update_meter('loss_epoch_{}'.format(epoch_num), fold_num, loss)
            self.results.update_meter('loss', fold_num, loss)
            if _flag(self.args, 'debug'):
                break

    def handle_train_optimization(self, optimizer, outputs, target, inputs, fold_num, total_batches, batch_idx,
                                  epoch_num, model):
        """:161-173.  loss + backward + clamp + step + zero_grad are one captured step; ``outputs`` (a forward the
        caller already ran) is not needed and ignored."""
        loss = optimizer.train_step(inputs, target)
        self.results.update_meter('loss_epoch_{}'.format(epoch_num), fold_num, loss.clone())
        return loss

    def run_test_epoch(self, epoch_num, model, test_dataset, test_loader, fold_num, optimizer=None, _steps_only=False):
        """:424-465 + record_final_epoch_testing_results (:519-524): no_grad forward with train-mode modules (the
        reference never calls eval()), loss meter, window argmax, per-patient votes -- reduced on the device."""
        store, batch_size, shuffle = test_loader
        trainer = optimizer if optimizer is not None else HotPathTrainer(model, use_graph=_flag(self.args, 'use_graph', True))
        trainer.clip_odd_batches = self.clip_odd_batches
        slot = self.args.test_patient_slot
        if slot is None:
            slot = torch.zeros(store.tiles.shape[0], dtype=torch.int64)
        gen = None                                            # the test loader shuffles too unless --unshuffled (:333-338)
        if shuffle and self.args.seed is not None:
            gen = torch.Generator().manual_seed(self.args.seed + 1000 * fold_num + epoch_num + 500009)
        steps = test_epoch_steps(trainer, store, slot, batch_size=batch_size, shuffle=shuffle, generator=gen)
        if _steps_only:                                       # folds in flight: the caller walks the steps and finishes
            return steps
        for _ in steps:
            pass
        if optimizer is None:
            trainer.release_graphs()                          # a throw-away trainer: free its graphs here, not in a GC pass
        return self._finish_test_epoch(steps, epoch_num, fold_num)

    def _finish_test_epoch(self, steps, epoch_num, fold_num):
        res = steps.result()
        self.preds, self.pred_idx = res['window_pred'].tolist(), res['window_abs_index'].tolist()   # obs_idx is absolute
        self.results.update_meter('test_loss', fold_num, res['mean_loss'])
        self.results.patient_results[(fold_num, epoch_num)] = res
        return res

    def restore_dtypes(self):
        """Put back the process-wide conv arithmetic / activation storage this run replaced (--conv-dtype)."""
        if self._prev_dtypes is not None:
            from . import functional as F_
            conv, storage = self._prev_dtypes
            F_.set_conv_dtype(conv)
            if storage != F_.storage_dtype():
                F_.set_storage_dtype(storage)
            self._prev_dtypes = None

    def train_and_test(self):
        """:340-378 without plotting: fold loop, epoch loop, per-epoch / per-fold whole-module saves under the
        reference's file names (``deepards_amd.checkpoint.model_save_path``)."""
        try:
            return self._train_and_test()
        finally:
            self.restore_dtypes()

    def _train_and_test(self):
        from .checkpoint import model_save_path
        a = self.args
        saved_models_dir = a.saved_models_dir if getattr(a, 'saved_models_dir', None) else saved_models_default_dir
        n_flight = getattr(a, 'folds_in_flight', None)
        if n_flight is None:                             # default: up to 5 folds side by side on a single GPU (same results)
            n_flight = min(5, self.n_kfolds) if self._data_parallel()[0] == 1 else 1
        n_flight = int(n_flight)
        if n_flight > 1 and self.n_kfolds > 1 and a.kfolds is not None:
            return self._train_and_test_folds_in_flight(n_flight, saved_models_dir)
        my_folds = None
        n_groups = int(getattr(a, 'fold_groups', None) or 1)
        if n_groups > 1:
            # config C4: fold groups x data-parallel sub-groups (deepards_amd.train.fold_group_layout)
            if self._data_parallel()[0] == 1:
                raise ValueError('--fold-groups needs torch.distributed (python -m torch.distributed.run ...)')
            from .train import make_fold_groups
            group, gworld, grank, my_folds = make_fold_groups(n_groups, self.n_kfolds)
            self._dp_override = (gworld, grank, group)
        for fold_num, (train_dataset, train_loader, test_dataset, test_loader) in enumerate(self.get_splits()):
            if (a.only_fold and fold_num != a.only_fold) or (my_folds is not None and fold_num not in my_folds):
                continue
            model = self.get_model()
            optimizer = self.get_optimizer(model, *self._data_parallel())
            for epoch_num in range(1, a.epochs + 1):
                if not _flag(a, 'no_train'):
                    self.run_train_epoch(model, train_loader, optimizer, epoch_num, fold_num)
                if _flag(a, 'reshuffle_oversample_per_epoch'):
                    train_loader[0].set_oversampling_indices()                       # :350-351
                if not _flag(a, 'no_test_after_epochs') or epoch_num == a.epochs - 1:
                    self.run_test_epoch(epoch_num, model, test_dataset, test_loader, fold_num, optimizer=optimizer)
                if _flag(a, 'save_model_per_epoch'):
                    self._save(model, model_save_path(a.save_model, saved_models_dir, self.n_kfolds, fold_num, epoch_num))
            if a.save_model:
                self._save(model, model_save_path(a.save_model, saved_models_dir, self.n_kfolds, fold_num))
            optimizer.release_graphs()                   # this fold's captured steps go NOW, not in some later gc pass (the
            self.model, self.optimizer = model, optimizer    # trainer stays usable: it re-captures on its next step)
        if my_folds is not None:                         # every rank ends with every fold's patient results
            from .train import gather_fold_results
            self.results.patient_results = gather_fold_results(self.results.patient_results, self._data_parallel()[1] == 0)
        return self.results

    def _train_and_test_folds_in_flight(self, n_flight, saved_models_dir):
        """The fold loop of ``train_and_test`` with ``n_flight`` folds side by side on this GPU: every fold has its own
        view of the tile store (fold indices, scaling factors, oversampling draws), model, trainer (captured step) and
        HIP stream; the host walks the folds' batches round-robin, so the step kernels of different folds overlap on the
        chip (independent launches fill the ramp / drain / latency gaps a B <= 64 step leaves: +12 % aggregate at B = 64
        with two folds, +17 % at B = 16 with four, scripts/two_fold_probe.py).  Within a fold nothing changes -- same
        kernels in the same order on one stream -- so its losses, predictions and weights are those of the sequential
        loop bit for bit (tests/test_model_gpu.py)."""
        import copy
        from .checkpoint import model_save_path
        from .train import concurrent_streams, epoch_shards_on_device, place_replicas_on_streams, shared_generator
        a = self.args
        if self._data_parallel()[0] > 1:
            raise NotImplementedError('--folds-in-flight with data parallelism: give every rank group its own folds instead')
        train_dataset, test_dataset = self.get_base_datasets()
        folds = [f for f in range(self.n_kfolds) if not (a.only_fold and f != a.only_fold)]
        shuffle = not _flag(a, 'unshuffled')
        for g0 in range(0, len(folds), n_flight):
            ctx = []
            streams = concurrent_streams(len(folds[g0:g0 + n_flight]))     # on different hardware queues (measured)
            for fold_num in folds[g0:g0 + n_flight]:
                tr_ds, te_ds = copy.copy(train_dataset), copy.copy(test_dataset)
                self._seed_fold_sampler(tr_ds, fold_num)
                for ds in (tr_ds, te_ds):
                    ds.set_kfold_indexes_for_fold(fold_num)
                stream = streams[len(ctx)]
                with torch.cuda.stream(stream):
                    model = self.get_model()
                    optimizer = self.get_optimizer(model, 1, 0, None)
                ctx.append((fold_num, stream, model, optimizer, tr_ds, te_ds))
            torch.cuda.synchronize()
            placed = not _flag(a, 'use_graph', True)
            for epoch_num in range(1, a.epochs + 1):
                if not _flag(a, 'no_train'):
                    plans = []
                    for fold_num, stream, model, optimizer, tr_ds, te_ds in ctx:
                        gen = None
                        if a.seed is not None:
                            gen = torch.Generator().manual_seed(a.seed + 1000 * fold_num + epoch_num)
                        if shuffle:
                            gen = shared_generator(optimizer, gen)
                        with torch.cuda.stream(stream):      # the index upload is ordered before the fold's gathers
                            plans.append(epoch_shards_on_device(tr_ds, a.batch_size, shuffle, gen, 1, 0))
                    for b in range(max(len(p) for p in plans)):
                        if not placed and all(c[3].static_batch() is not None for c in ctx) and len(ctx) > 1:
                            # every fold has its captured step: measure which streams let them overlap (restores all state)
                            streams = place_replicas_on_streams([c[3] for c in ctx])
                            ctx = [(c[0], s) + c[2:] for c, s in zip(ctx, streams)]
                            placed = True
                        for (fold_num, stream, model, optimizer, tr_ds, te_ds), plan in zip(ctx, plans):
                            if b >= len(plan) or (_flag(a, 'debug') and b > 0):
                                continue
                            idx = plan[b]
                            with torch.cuda.stream(stream):
                                static = optimizer.static_batch(len(idx))
                                x, t = tr_ds.batch_from_device(idx, out=static) if static is not None else tr_ds.batch_from_device(idx)
                                loss = optimizer.train_step(x, t).clone()
                            self.results.update_meter('loss_epoch_{}'.format(epoch_num), fold_num, loss)
                            self.results.update_meter('loss', fold_num, loss)
                    torch.cuda.synchronize()                 # the meters are read from the default stream
                for fold_num, stream, model, optimizer, tr_ds, te_ds in ctx:
                    if _flag(a, 'reshuffle_oversample_per_epoch'):
                        tr_ds.set_oversampling_indices()
                if not _flag(a, 'no_test_after_epochs') or epoch_num == a.epochs - 1:
                    walks = []                           # the folds' test epochs, one step of each in turn
                    for fold_num, stream, model, optimizer, tr_ds, te_ds in ctx:
                        with torch.cuda.stream(stream):
                            walks.append(self.run_test_epoch(epoch_num, model, te_ds, (te_ds, a.batch_size, shuffle), fold_num,
                                                             optimizer=optimizer, _steps_only=True))
                    live = list(range(len(ctx)))
                    while live:
                        for i in list(live):
                            with torch.cuda.stream(ctx[i][1]):
                                if next(walks[i], None) is None:
                                    live.remove(i)
                    torch.cuda.synchronize()
                    for (fold_num, stream, model, optimizer, tr_ds, te_ds), steps in zip(ctx, walks):
                        self._finish_test_epoch(steps, epoch_num, fold_num)
                for fold_num, stream, model, optimizer, tr_ds, te_ds in ctx:
                    if _flag(a, 'save_model_per_epoch'):
                        self._save(model, model_save_path(a.save_model, saved_models_dir, self.n_kfolds, fold_num, epoch_num))
            torch.cuda.synchronize()
            for fold_num, stream, model, optimizer, tr_ds, te_ds in ctx:
                if a.save_model:
                    self._save(model, model_save_path(a.save_model, saved_models_dir, self.n_kfolds, fold_num))
                self.model, self.optimizer = model, optimizer
            self.fold_models = getattr(self, 'fold_models', {})
            self.fold_models.update({c[0]: c[2] for c in ctx})
            for c in ctx:                                # the group's captured steps are released here, explicitly, before
                c[3].release_graphs()                    # the next group captures its own (ownership rule, train._capture_graph)
        if folds:                                        # like the sequential loop, leave the caller's stores at the last fold
            self._seed_fold_sampler(train_dataset, folds[-1])
            for ds in (train_dataset, test_dataset):
                ds.set_kfold_indexes_for_fold(folds[-1])
        return self.results

    def _save(self, model, path):
        """``torch.save(model, model_path)`` (:364,374): the whole module; under data parallelism the replicas' BatchNorm
        running statistics (updated from each rank's own shard) are averaged first, then rank 0 of the group writes."""
        world, rank, group = self._data_parallel()
        if world > 1:
            from .train import average_replica_buffers
            average_replica_buffers(model, world, group)
        if rank != 0:
            return
        os.makedirs(os.path.dirname(path) or '.', exist_ok=True)
        # a trainer hangs its gradient destinations on the Parameters (``p._da_grad``: views into its flat bucket); pickled
        # along they would double the file and, in the loaded model, point functional._tgt at a dead buffer
        stash = [(q, q.__dict__.pop('_da_grad')) for q in model.parameters() if '_da_grad' in q.__dict__]
        try:
            torch.save(model, path)
        finally:
            for q, g in stash:
                q._da_grad = g

    _dp_override = None

    def _data_parallel(self):
        """(world_size, rank, group): one process per GPU under torch.distributed (RCCL) instead of nn.DataParallel (:96);
        every rank takes its window shard of each batch, gradients meet in one all-reduce (deepards_amd.train).  With
        --fold-groups the triple describes this rank's data-parallel SUB-group."""
        import torch.distributed as dist
        if self._dp_override is not None:
            return self._dp_override
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(), dist.get_rank(), None
        return 1, 0, None

    def clip_odd_batch_sizes(self, obs_idx, seq, metadata, target):
        from .train import clip_odd_batch_sizes
        return clip_odd_batch_sizes(obs_idx, seq, metadata, target)

    def transform_obs_idx(self, obs_idx, outputs):
        return obs_idx


class PatientClassifierMixin(object):
    def set_loss_criterion(self):
        """:526-532: BCEWithLogitsLoss (mean over B*2) -- ``da_bce_logits`` inside the trainer's step; the other
        loss_func choices of the reference are out of scope."""
        if self.args.loss_func != 'bce':
            raise NotImplementedError('loss_func %r: only bce is on the hot path' % self.args.loss_func)
        from .functional import bce_with_logits
        self.criterion = bce_with_logits


class CNNLinearModel(BaseTraining, PatientClassifierMixin):
    def __init__(self, args):
        super(CNNLinearModel, self).__init__(args)

    def calc_loss(self, outputs, target, inputs):
        return self.criterion(outputs, target)

    def _process_test_batch_results(self, outputs, target, inputs, fold_num):
        return outputs.argmax(dim=-1).cpu().tolist()

    def get_network(self, base_network):
        return M.CNNLinearNetwork(base_network, self.args.n_sub_batches, self.n_metadata_inputs)


class CNNDoubleLinearModel(CNNLinearModel):
    def get_network(self, base_network):
        return M.CNNDoubleLinearNetwork(base_network, self.args.n_sub_batches, self.n_metadata_inputs)


class CNNLinearToMeanModel(CNNLinearModel):
    def get_network(self, base_network):
        return M.CNNLinearToMean(base_network)


class CNNLinearComprToRFModel(CNNLinearModel):
    def get_network(self, base_network):
        return M.CNNLinearComprToRF(base_network)


class PerBreathClassifierMixin(object):
    """:539-555: the window target repeated over the breaths for the loss (the trainer's step does that for (B, NB, 2)
    outputs), one prediction and one patient vote per breath."""

    def calc_loss(self, outputs, target, inputs):
        if self.args.batch_size > 1:
            target = target.unsqueeze(1)
        return self.criterion(outputs, target.repeat((1, outputs.shape[1], 1)))

    def _process_test_batch_results(self, outputs, target, inputs, fold_num):
        return outputs.argmax(dim=-1).cpu().view(-1).tolist()

    def transform_obs_idx(self, obs_idx, outputs):
        return obs_idx.reshape((outputs.shape[0], 1)).repeat((1, outputs.shape[1])).view(-1)


class CNNSingleBreathLinearModel(PerBreathClassifierMixin, BaseTraining, PatientClassifierMixin):
    def __init__(self, args):
        super(CNNSingleBreathLinearModel, self).__init__(args)

    def get_network(self, base_network):
        return M.CNNSingleBreathLinearNetwork(base_network)


network_map = {
    'cnn_linear': CNNLinearModel,
    'cnn_single_breath_linear': CNNSingleBreathLinearModel,
    'cnn_double_linear': CNNDoubleLinearModel,
    'cnn_linear_to_mean': CNNLinearToMeanModel,
    'cnn_linear_compr_to_rf': CNNLinearComprToRFModel,
}

# flags of the reference's parser that steer code outside the hot path: recognised so that the error says why
OUT_OF_SCOPE_FLAGS = (
    '--transforms', '-tp', '--transform-probability', '--use-i', '-r2', '--drop-if-under-r2', '--drop-i-lim', '--drop-e-lim',
    '--truncate-e-lim', '--butter-low', '--butter-high', '--post-hoc-downsampling', '--fft-filtering-low',
    '--fft-filtering-high', '--load-siamese', '--valpha', '--conf-beta', '--fl-gamma', '--fl-alpha', '--transformer-blocks',
    '--plot-untiled-disease-evol', '--plot-tiled-disease-evol', '--plot-dtw-with-disease', '--plot-pt-dtw-by-minute',
    '--perform-dtw-preprocessing', '--n-warm-epochs', '-pse', '--push-start-epoch', '--push-every-n', '--n-push-iters',
    '--clust-lambda', '--sep-lambda', '-vse', '--viz-start-epoch', '--viz-every-n', '--prototype-results-dir',
    '--prototype-fname-prefix', '-np', '--n-prototypes', '-ic', '--incorrect-strength', '--average-linear-layer', '--use-l1',
    '-2dt', '--two-dim-transforms', '-bks', '--block-kernel-size', '--multitask-epochs', '--row-mix', '-usf',
    '--undersample-factor', '-usdf', '--undersample-std-factor', '--train-pt-frac', '--final-validation',
    '--holdout-set-type', '--downsample-factor', '-lc', '--loss-calc', '--bm-to-linear',
)


def build_parser():
    """The reference's parser (:1439-1576) for the flags the hot path reads: same names, short forms, types and
    all-None defaults (so that ``Configuration`` can tell "not given" from a value).  Flags that only steer code
    outside the hot path are refused by ``main`` with the reason (``OUT_OF_SCOPE_FLAGS``)."""
    parser = argparse.ArgumentParser(prog='deepards_amd.train_ards_detector')

    def true_false_flag(flag, help):
        return parser.add_argument(flag, action='store_true', help=help, default=None)
    parser.add_argument('-co', '--config-override', help='path to yml file that overrides elements of defaults.yml')
    parser.add_argument('-dp', '--data-path', help='kept for the configuration files; raw-file ingestion is out of scope')
    parser.add_argument('-en', '--experiment-num', type=int)
    parser.add_argument('-c', '--cohort-file')
    parser.add_argument('-n', '--network', choices=list(network_map.keys()))
    parser.add_argument('-e', '--epochs', type=int)
    parser.add_argument('-p', '--train-from-pickle', help='ARDSRawDataset pickle (read without unpickling) or its .npz export')
    parser.add_argument('--train-to-pickle', help='write the ingested train dataset as .npz')
    parser.add_argument('--test-from-pickle')
    parser.add_argument('--test-to-pickle')
    true_false_flag('--cuda', 'run on the GPU(s): one process per GPU under torch.distributed.run')
    true_false_flag('--cuda-no-dp', 'run on the single GPU --cuda-device')
    parser.add_argument('-b', '--batch-size', type=int)
    parser.add_argument('--base-network', choices=sorted(base_networks))
    parser.add_argument('-nb', '--n-sub-batches', type=int)
    true_false_flag('--no-print-progress', '')
    parser.add_argument('--kfolds', type=int)
    parser.add_argument('-rip', '--initial-planes', type=int)
    parser.add_argument('-rfpt', '--resnet-first-pool-type', choices=['max', 'avg'])
    true_false_flag('--no-test-after-epochs', '')
    true_false_flag('--debug', 'debug code and dont train')
    parser.add_argument('--optimizer', choices=['adam', 'sgd'])
    parser.add_argument('-dt', '--dataset-type', choices=['unpadded_centered_sequences',
                                                          'padded_breath_by_breath_with_flow_time_features'])
    parser.add_argument('-lr', '--learning-rate', type=float)
    parser.add_argument('--loader-threads', type=int, help='accepted and ignored: batches are gathered on the device')
    parser.add_argument('--save-model', help='save the model to a specific file')
    true_false_flag('--save-model-per-epoch', 'save the model at the end of each epoch')
    parser.add_argument('--load-base-network', help='load base network only from a saved model')
    parser.add_argument('--load-checkpoint', help='load a checkpoint of the model for further training or inference')
    true_false_flag('--no-train', 'Dont train model, just evaluate for inference')
    true_false_flag('--resnet-double-conv', '')
    parser.add_argument('-exp', '--experiment-name')
    parser.add_argument('-wd', '--weight-decay', type=float)
    parser.add_argument('-loss', '--loss-func', choices=['bce'])
    parser.add_argument('--time-series-hidden-units', type=int)
    true_false_flag('--unshuffled', 'dont shuffle data')
    true_false_flag('--oversample-minority', '')
    parser.add_argument('--oversample-all-factor', type=float)
    true_false_flag('--reshuffle-oversample-per-epoch', '')
    true_false_flag('--freeze-base-network', '')
    true_false_flag('--stop-on-loss', '')
    parser.add_argument('--stop-thresh', type=float)
    parser.add_argument('--stop-after-epoch', type=int)
    true_false_flag('--clip-grad', '')
    parser.add_argument('--clip-val', type=float)
    parser.add_argument('--cuda-device', type=int, help='number of cuda device you want to use')
    parser.add_argument('--only-fold', type=int, default=None, help='only run specific fold')
    parser.add_argument('--saved-models-dir', help='directory to save models')
    true_false_flag('--print-progress', '')
    true_false_flag('--with-fft', '')
    true_false_flag('--only-fft', '')
    true_false_flag('--fft-real-only', '')
    true_false_flag('--random-kfold', 'perform a random kfold splitting.')
    true_false_flag('--bootstrap', '')
    # this build's own switches
    parser.add_argument('--seed', type=int, help='seed of the initialisation, the shuffles and the oversampler')
    parser.add_argument('--no-graph', dest='use_graph', action='store_false', default=None, help='run the step eagerly')
    parser.add_argument('--conv-dtype', choices=['f32', 'bf16', 'f32x3p'], help='arithmetic of the residual-block convs')
    parser.add_argument('--folds-in-flight', type=int, help='k-folds trained side by side on this GPU, each on its own stream '
                        '(same per-fold results as one after the other; a B <= 64 step leaves the chip partly idle).  '
                        'Default: min(5, kfolds) on one GPU, 1 under data parallelism; 1 = the sequential loop')
    parser.add_argument('--fold-groups', type=int, help='under torch.distributed: split the ranks into this many groups; group g '
                        'trains folds g, g + G, ... data-parallel over its own ranks (BASELINE config C4: 5 folds over 4 GPUs '
                        'as e.g. 2 groups x 2 ranks).  Default 1: every fold over all ranks')
    parser.add_argument('--act-dtype', choices=['f32', 'bf16'], help='activation storage under --conv-dtype bf16 (default bf16 for ResNets)')
    return parser


def main(argv=None):
    """:1579-1590.  Under ``python -m torch.distributed.run --nproc-per-node N`` every process takes the GPU
    LOCAL_RANK and joins the RCCL process group before the model class is built."""
    import sys
    argv = sys.argv[1:] if argv is None else list(argv)
    for a in argv:
        if a.split('=')[0] in OUT_OF_SCOPE_FLAGS:
            raise SystemExit('%s steers code outside the accelerated cnn_linear hot path (SURVEY.md section 2) and is not '
                             'accepted by this build' % a.split('=')[0])
    args = Configuration(build_parser().parse_args(argv), BUILD_DEFAULTS)
    if args.save_model_per_epoch and not args.save_model:
        raise Exception('Must specify a filename to save your model using --save-model')
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world > 1:
        import torch.distributed as dist
        if not dist.is_initialized():                    # (a caller may bring its own group: the one-GPU rehearsals use gloo)
            local = int(os.environ.get('LOCAL_RANK', '0'))
            torch.cuda.set_device(local)
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    cls = network_map[args.network](args)
    results = cls.train_and_test()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return cls, results


if __name__ == "__main__":
    main()
