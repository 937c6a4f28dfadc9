"""Host mirror of the reference training driver for the cnn_linear path: same class / method / argument names as
``deepards/train_ards_detector.py`` (``network_map`` :1410-1436, ``base_networks`` :45-69, ``BaseTraining`` :73-512,
``PatientClassifierMixin`` :514-537, ``CNNLinearModel`` :925-939, defaults of ``defaults.yml``), so that

    cls = network_map[args.network](args)        # train_ards_detector.py:1589-1590
    cls.train_and_test()

runs the hot path on MI355X.  What differs, and why:

* the dataset objects are :class:`deepards_amd.data.DeviceTileStore` (windows resident in HBM) instead of
  ``ARDSRawDataset`` + ``DataLoader``; a "loader" is ``(store, batch_size, shuffle)``.  ``get_base_datasets`` therefore
  takes the stores from ``args.train_store`` / ``args.test_store`` (build them with ``deepards_amd.tiles`` from raw
  breaths, or from exported window arrays) -- unpickling the reference's dataset pickles needs the reference package;
* ``get_optimizer`` returns a :class:`deepards_amd.train.HotPathTrainer`: the clamp hooks of ``get_model`` (:474-476),
  SGD-Nesterov / Adam (:416-422) and ``zero_grad`` are one fused kernel at the end of the captured step, so
  ``handle_train_optimization`` is a single ``train_step``;
* results: the loss meters and the per-patient vote aggregation of ``DeepARDSResults`` (metrics.py:142-153,572-604) are
  kept on the device and read back once per epoch (``self.results`` is a small dict-of-lists recorder, not the
  reference's reporting / plotting class, which is out of scope).

There is no CPU path: constructing a model class with ``args.cuda`` false raises.
"""
import argparse

import torch

from . import models as M
from .train import HotPathTrainer, run_test_epoch, run_train_epoch_from_store

base_networks = M.base_networks

# defaults.yml (values the hot path reads) + the store_true flags of build_parser, all False / None by default
DEFAULTS = dict(
    network='cnn_linear', epochs=10, batch_size=16, base_network='densenet18', loader_threads=0,
    initial_planes=64, resnet_first_pool_type='max', resnet_double_conv=False,
    optimizer='sgd', dataset_type='unpadded_centered_sequences', learning_rate=0.001, n_sub_batches=20,
    weight_decay=0.0001, loss_func='bce', clip_grad=False, clip_val=0.01,
    with_fft=False, only_fft=False, fft_real_only=False, freeze_base_network=False,
    kfolds=None, bootstrap=False, only_fold=None, unshuffled=False, no_train=False, no_test_after_epochs=False,
    debug=False, cuda=True, cuda_no_dp=False, cuda_device=0, load_checkpoint=None, load_base_network=None,
    save_model=None, no_print_progress=True, print_progress=False, experiment_name='deepards_amd',
    train_store=None, test_store=None, test_patient_slot=None, use_graph=True, seed=None,
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


class BaseTraining(object):
    clip_odd_batches = False

    def __init__(self, args):
        self.args = args
        if not (args.cuda or args.cuda_no_dp):
            raise RuntimeError('deepards_amd runs the hot path on an MI355X only: pass cuda=True (no CPU fallback)')
        if not torch.cuda.is_available():
            raise RuntimeError('no HIP device visible; there is no CPU fallback')
        self.device = torch.device('cuda', args.cuda_device if args.cuda_no_dp else torch.cuda.current_device())
        self.cuda_wrapper = lambda x: x.to(self.device)
        self.model_cuda_wrapper = lambda x: x.to(self.device)   # one process per GPU; nn.DataParallel (:96) is not used
        self.set_loss_criterion()
        self.n_metadata_inputs = 9 if args.dataset_type == 'padded_breath_by_breath_with_flow_time_features' else 0
        if args.unshuffled and args.batch_size > 1:
            raise Exception('Currently we can only run unshuffled runs with a batch size of 1!')
        self.n_kfolds = 1 if (args.bootstrap or not args.kfolds) else args.kfolds
        self.results = Results()
        self.preds, self.pred_idx = [], []

    # ---- model / optimizer ---------------------------------------------------------------------------------------
    def get_base_network(self):
        """:380-414 for the backbones this package builds."""
        a = self.args
        ctor = base_networks[a.base_network]
        if a.load_base_network:
            saved = torch.load(a.load_base_network, weights_only=False)
            base_network = saved.breath_block
        elif a.base_network.startswith('resnet'):
            base_network = ctor(initial_planes=a.initial_planes, first_pool_type=a.resnet_first_pool_type,
                                double_conv_first=a.resnet_double_conv)
        else:
            base_network = ctor(with_fft=a.with_fft, only_fft=a.only_fft, fft_real_only=a.fft_real_only)
        if a.freeze_base_network:
            for p in base_network.parameters():
                p.requires_grad = False
        return base_network

    def get_model(self):
        """:467-477.  The +-clip_val clamp the reference registers as a hook on every trainable parameter is applied by
        the fused optimizer kernel of the trainer ``get_optimizer`` returns (after the gradient all-reduce when data
        parallel), so no hooks are registered here."""
        if self.args.load_checkpoint:
            model = torch.load(self.args.load_checkpoint, weights_only=False)
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
                              clip_grad=bool(a.clip_grad), clip_val=a.clip_val, world_size=world_size, rank=rank,
                              process_group=process_group, use_graph=a.use_graph)

    # ---- data ----------------------------------------------------------------------------------------------------
    def get_base_datasets(self):
        if self.args.train_store is None or self.args.test_store is None:
            raise ValueError('args.train_store / args.test_store (DeviceTileStore) are required')
        return self.args.train_store, self.args.test_store

    def get_splits(self):
        """:317-338: per fold, (train_dataset, train_loader, test_dataset, test_loader); a loader is the tuple
        (store, batch_size, shuffle) the epoch functions iterate on the device."""
        train_dataset, test_dataset = self.get_base_datasets()
        for i in range(self.n_kfolds):
            if self.args.kfolds is not None or self.args.bootstrap:
                for ds in (train_dataset, test_dataset):
                    if hasattr(ds, 'set_kfold_indexes_for_fold'):
                        ds.set_kfold_indexes_for_fold(i)
            shuffle = not self.args.unshuffled
            yield (train_dataset, (train_dataset, self.args.batch_size, shuffle),
                   test_dataset, (test_dataset, self.args.batch_size, shuffle))

    # ---- epochs --------------------------------------------------------------------------------------------------
    def run_train_epoch(self, model, train_loader, optimizer, epoch_num, fold_num):
        """:139-159.  ``optimizer`` is the trainer from get_optimizer; every batch is one gather kernel + one graph
        replay; the per-batch losses stay on the device until a meter is read."""
        store, batch_size, shuffle = train_loader
        if optimizer.model is not model:
            raise ValueError('optimizer was built for another model')
        gen = None
        if self.args.seed is not None:
            gen = torch.Generator().manual_seed(self.args.seed + 1000 * fold_num + epoch_num)
        for loss in run_train_epoch_from_store(optimizer, store, batch_size=batch_size, shuffle=shuffle, generator=gen):
            self.results.update_meter('loss_epoch_{}'.format(epoch_num), fold_num, loss)
            self.results.update_meter('loss', fold_num, loss)
            if self.args.debug:
                break

    def handle_train_optimization(self, optimizer, outputs, target, inputs, fold_num, total_batches, batch_idx,
                                  epoch_num, model):
        """:161-173.  loss + backward + clamp + step + zero_grad are one captured step; ``outputs`` (a forward the
        caller already ran) is not needed and ignored."""
        loss = optimizer.train_step(inputs, target)
        self.results.update_meter('loss_epoch_{}'.format(epoch_num), fold_num, loss.clone())
        return loss

    def run_test_epoch(self, epoch_num, model, test_dataset, test_loader, fold_num, optimizer=None):
        """:424-465 + record_final_epoch_testing_results (:519-524): no_grad forward with train-mode modules (the
        reference never calls eval()), loss meter, window argmax, per-patient votes -- reduced on the device."""
        store, batch_size, _ = test_loader
        trainer = optimizer if optimizer is not None else HotPathTrainer(model, use_graph=self.args.use_graph)
        trainer.clip_odd_batches = self.clip_odd_batches
        slot = self.args.test_patient_slot
        if slot is None:
            slot = torch.zeros(store.tiles.shape[0], dtype=torch.int64)
        res = run_test_epoch(trainer, store, slot, batch_size=batch_size)
        if optimizer is None:
            trainer.release_graphs()                          # a throw-away trainer: free its graphs here, not in a GC pass
        self.preds, self.pred_idx = res['window_pred'].tolist(), res['window_abs_index'].tolist()   # obs_idx is absolute
        self.results.update_meter('test_loss', fold_num, res['mean_loss'])
        self.results.patient_results[(fold_num, epoch_num)] = res
        return res

    def train_and_test(self):
        """:340-378 without checkpoint-per-epoch and plotting."""
        for fold_num, (train_dataset, train_loader, test_dataset, test_loader) in enumerate(self.get_splits()):
            if self.args.only_fold and fold_num != self.args.only_fold:
                continue
            model = self.get_model()
            optimizer = self.get_optimizer(model, *self._data_parallel())
            for epoch_num in range(1, self.args.epochs + 1):
                if not self.args.no_train:
                    self.run_train_epoch(model, train_loader, optimizer, epoch_num, fold_num)
                if not self.args.no_test_after_epochs or epoch_num == self.args.epochs - 1:
                    self.run_test_epoch(epoch_num, model, test_dataset, test_loader, fold_num, optimizer=optimizer)
            if self.args.save_model:
                torch.save(model, self.args.save_model if self.n_kfolds == 1 else
                           '%s-fold%d.pth' % (self.args.save_model.rsplit('.pth', 1)[0], fold_num))
            self.model, self.optimizer = model, optimizer
        return self.results

    @staticmethod
    def _data_parallel():
        """(world_size, rank, group): one process per GPU under torch.distributed (RCCL) instead of nn.DataParallel (:96);
        every rank takes its window shard of each batch, gradients meet in one all-reduce (deepards_amd.train)."""
        import torch.distributed as dist
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
