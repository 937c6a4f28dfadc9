"""Host side of the training / test hot loop.

Mirrors, for the cnn_linear path only, what the reference does in
``train_ards_detector.py``: ``BaseTraining.run_train_epoch`` (:139-159),
``handle_train_optimization`` (:161-173), ``get_optimizer`` (:416-422), the clamp hooks of
``get_model`` (:474-476), ``run_test_epoch`` (:424-465), ``clip_odd_batch_sizes`` (:482-494) and
``CNNLinearModel.calc_loss/_process_test_batch_results`` (:929-936).

MI355X-first differences (results identical, see tests):
  * parameters / gradients / momentum live in three flat fp32 buffers; clamp + weight decay +
    Nesterov momentum (or Adam) is ONE fused kernel over the flat buffer instead of a hook and an
    optimizer loop per parameter;
  * the whole step (zero-grad, forward, BCE, backward, update) is captured once into a hipGraph and
    replayed: ~400 kernel launches per step cost one graph launch;
  * data parallelism is one process per GPU: each rank takes an equal shard of the windows of the
    batch, gradients are summed with a single RCCL all-reduce of the flat bucket over xGMI, and the
    1/world scale and the +-clip clamp are applied AFTER the reduction (SURVEY.md finding 7), inside
    the optimizer kernel.  BatchNorm never crosses windows, so no SyncBN is needed (finding 3).
  * no per-step device->host sync: losses stay on the device until asked for.
"""
import contextlib
import gc
import math
import os

import torch

from . import functional as F_
from . import hip_ops as H


@contextlib.contextmanager
def _capture_graph(graph, _ctx=None):
    """``torch.cuda.graph(graph)`` with the cyclic garbage collector flushed before and held off during the capture,
    and its previous state restored afterwards (also when the capture raises).

    Why ("capture window"): between hipStreamBeginCapture and EndCapture the HIP runtime
    refuses everything that is not a stream-ordered enqueue (hipFree, hipEventQuery, hipGraphExecDestroy of a graph
    whose pool is live, ...), and PyTorch turns such a refusal inside a destructor into a process abort.  A
    collection that happens to trigger inside the window finalises whatever cyclic garbage is pending -- dropped
    trainers with their captured graphs, static outputs, events -- i.e. runs exactly those calls.  Ownership rule that
    makes the window safe: (1) nothing this package owns is RELEASED inside a capture (graphs are cached per shape for
    the trainer's lifetime and dropped only by ``release_graphs()``; the code inside the window creates objects but
    drops none that existed before it), (2) pending cyclic garbage is finalised by ``gc.collect()`` BEFORE the window,
    (3) the collector stays off inside it.  Refcount frees inside the window are then only those of tensors allocated
    inside it, which the caching allocator's capture-aware pool handles."""
    ctx = torch.cuda.graph if _ctx is None else _ctx
    gc.collect()
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        with ctx(graph):
            yield
    finally:
        if was_enabled:
            gc.enable()


def graph_branch_count(graph):
    """Number of nodes of a captured ``torch.cuda.CUDAGraph(keep_graph=True)`` with more than one successor or predecessor
    -- 0 for the single chain every captured step of this package is meant to be; None when the topology cannot be read.
    (hipGraphGetEdges on the raw graph handle; this stack's hipGraphDebugDotPrint writes an empty file.)"""
    import ctypes
    try:
        raw = graph.raw_cuda_graph()
        hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so'))
        ne = ctypes.c_size_t(0)
        if hip.hipGraphGetEdges(ctypes.c_void_p(raw), None, None, ctypes.byref(ne)) != 0:
            return None
        if ne.value == 0:
            return 0
        fr, to = (ctypes.c_void_p * ne.value)(), (ctypes.c_void_p * ne.value)()
        if hip.hipGraphGetEdges(ctypes.c_void_p(raw), fr, to, ctypes.byref(ne)) != 0:
            return None
    except Exception:                                    # noqa: BLE001
        return None
    succ, pred = {}, {}
    for a_, b_ in set(zip(list(fr), list(to))):
        succ[a_] = succ.get(a_, 0) + 1
        pred[b_] = pred.get(b_, 0) + 1
    return sum(1 for v in succ.values() if v > 1) + sum(1 for v in pred.values() if v > 1)


def clip_odd_batch_sizes(obs_idx, seq, metadata, target):
    """train_ards_detector.py:482-494 -- drop the last item of an odd batch."""
    if seq.shape[0] % 2 == 1:
        n = seq.shape[0] - 1
        return obs_idx[:n], seq[:n], metadata[:n], target[:n]
    return obs_idx, seq, metadata, target


def shard_windows(n_windows, world_size, rank):
    """Equal contiguous shards of the batch's windows: rank r owns [r*B/W, (r+1)*B/W).  Equal sizes
    make the mean of the rank-local BCE means equal the global mean.  Batches reach this
    already trimmed by ``legal_batch_len`` (the epoch iterators do it), so a remainder is a caller bug."""
    if n_windows % world_size:
        raise ValueError('batch of %d windows does not split evenly over %d ranks' % (n_windows, world_size))
    per = n_windows // world_size
    return slice(rank * per, (rank + 1) * per)


def batch_multiple(batch_size, world_size=1):
    """Windows per batch must be a multiple of this: 2 from clip_odd_batch_sizes (train_ards_detector.py:146-147,
    482-494; not applied when batch_size == 1), times the ranks that share the batch -> lcm(2, world)."""
    m = 1 if batch_size == 1 else 2
    return m * world_size // math.gcd(m, world_size)


def legal_batch_len(n, batch_size, world_size=1):
    """Length a batch of n windows is trimmed to: the reference drops the last item of an odd batch; under data
    parallelism the (tail) batch is trimmed to a multiple of lcm(2, world) so that every rank gets the same number
    of windows (N=70, batch 16, world 4: the tail of 6 trains on 4 windows instead of raising)."""
    m = batch_multiple(batch_size, world_size)
    return n - n % m


def fold_group_layout(world_size, rank, n_groups, n_folds):
    """BASELINE config C4 ("5-fold data-parallel over 4 GPUs") as fold groups x data-parallel sub-groups: the ranks are
    dealt to ``n_groups`` contiguous groups of world_size / n_groups ranks; group g trains the folds g, g + n_groups, ...
    data-parallel over ITS ranks while the other groups train theirs.  n_groups = 1: every fold over all ranks (one after
    the other); n_groups = world_size: every rank trains its own folds (what the reference's
    scripts/main/run_non_pretraining_experiments.py:17-25 does with one process per GPU).
    -> (rank lists of all groups, this rank's group, the folds of that group)."""
    if n_groups < 1 or world_size % n_groups:
        raise ValueError('%d fold groups do not divide %d ranks' % (n_groups, world_size))
    per = world_size // n_groups
    groups = [list(range(g * per, (g + 1) * per)) for g in range(n_groups)]
    mine = rank // per
    return groups, mine, [f for f in range(n_folds) if f % n_groups == mine]


def make_fold_groups(n_groups, n_folds):
    """The process groups of ``fold_group_layout`` (every rank creates every group, in the same order, as
    ``dist.new_group`` requires) -> (this rank's sub-group, its size, this rank's rank in it, its folds)."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    groups, mine, folds = fold_group_layout(world, rank, n_groups, n_folds)
    pgs = [dist.new_group(ranks=g) for g in groups]
    return pgs[mine], len(groups[mine]), rank - groups[mine][0], folds


def gather_fold_results(patient_results, is_group_leader):
    """Every rank ends with every fold's patient results: the leaders of the fold groups contribute theirs (the other
    ranks of a group hold the same numbers), merged over the WORLD group."""
    import torch.distributed as dist
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, patient_results if is_group_leader else {})
    merged = {}
    for p in parts:
        merged.update(p)
    return merged


def average_replica_buffers(model, world_size, group=None):
    """Data parallel: the ResNet running statistics are updated from each rank's OWN window shard, so they drift apart
    between replicas (parameters do not: identical reduced gradients).  Nothing on this path reads them (the reference never
    calls eval(), SURVEY finding 4), but a checkpoint should not depend on which rank wrote it: before a save every floating
    buffer is replaced by its mean over the ranks, integer buffers (num_batches_tracked: equal shard sizes, so
    equal counts) by rank 0's.  One flat all-reduce per dtype."""
    if world_size == 1:
        return
    import torch.distributed as dist
    floats = [b for b in model.buffers() if b.is_floating_point()]
    if floats:
        flat = torch.cat([b.detach().reshape(-1).float() for b in floats])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat /= world_size
        off = 0
        with torch.no_grad():
            for b in floats:
                n = b.numel()
                b.copy_(flat[off:off + n].view(b.shape).to(b.dtype))
                off += n
    ints = [b for b in model.buffers() if not b.is_floating_point()]
    src = dist.get_global_rank(group, 0) if group is not None else 0
    for b in ints:
        dist.broadcast(b, src=src, group=group)


_GRAD_OVERWRITE = True    # False (tests): every captured step zero-fills the gradient bucket and its writers accumulate
_FUSED_HEAD = os.environ.get('DA_FUSED_HEAD', '1') != '0'      # 0: the six-launch head chain (A/B and the sibling heads' path)


def _logits(out):
    """CNNLSTMNetwork returns (logits, (hx, cx)); every batch starts from a zero state here (the reference's stateful
    per-patient carry, train_ards_detector.py:845-849, is the caller's loop: pass hx_cx to the model yourself)."""
    return out[0] if isinstance(out, tuple) else out


def _loss_operands(logits, target):
    """(logits, target) as (n, 2) pairs for the BCE kernel.  Per-breath outputs (B, NB, 2) repeat the window target
    over the breaths (PerBreathClassifierMixin.calc_loss, train_ards_detector.py:540-543)."""
    if logits.dim() == 3:
        target = target.unsqueeze(1).expand(-1, logits.shape[1], -1)
    return logits.reshape(-1, 2).contiguous(), target.reshape(-1, 2).contiguous()


class FlatBucket(object):
    """Flat fp32 views over the live parameters: p (weights), g (gradients)."""

    def __init__(self, params):
        self.params = list(params)
        align = 64                                       # floats: every parameter starts 256-B aligned (float4 loads)
        self.offsets = []
        off = 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + align - 1) // align * align
        self.numel = off
        dev = self.params[0].device
        self.p = torch.zeros(self.numel, device=dev, dtype=torch.float32)
        self.g = torch.zeros(self.numel, device=dev, dtype=torch.float32)
        for p, off in zip(self.params, self.offsets):
            n = p.numel()
            self.p[off:off + n].copy_(p.data.reshape(-1))
            if p.grad is not None:
                self.g[off:off + n].copy_(p.grad.reshape(-1))
            p.data = self.p[off:off + n].view(p.shape)
            p.grad = self.g[off:off + n].view(p.shape)
            p._da_grad = p.grad                          # backward kernels accumulate straight into the bucket

    def zero_grad(self):
        self.g.zero_()

    def allreduce(self, group=None):
        import torch.distributed as dist
        dist.all_reduce(self.g, op=dist.ReduceOp.SUM, group=group)


class HotPathTrainer(object):
    """One model replica on one GPU.  ``train_step(inputs, target)`` == one iteration of the
    reference's batch loop; ``test_step`` == one iteration of run_test_epoch (train-mode modules,
    no_grad -- the reference never calls model.eval(), SURVEY.md finding 4)."""

    def __init__(self, model, optimizer='sgd', learning_rate=1e-3, weight_decay=1e-4, momentum=0.9,
                 clip_grad=True, clip_val=0.01, world_size=1, rank=0, process_group=None, use_graph=True):
        if optimizer not in ('sgd', 'adam'):
            raise ValueError('optimizer must be sgd or adam')
        self.model = model
        self.optimizer = optimizer
        self.lr, self.wd, self.momentum = learning_rate, weight_decay, momentum
        self.clip = clip_val if clip_grad else 0.0
        self.world_size, self.rank, self.group = world_size, rank, process_group
        self.use_graph = use_graph
        self.bucket = None
        self.state = {}
        self.steps = 0
        self._graphs = {}                  # input shape -> captured step (a tail batch keeps its own graph: no recapture)
        self._graph = None                 # the entry train_step used last: (graph, static, static_out, graph_opt)
        self._static = None
        self._test_graphs = {}
        self.last_loss = None
        self.last_logits = None
        self.allreduce_calls = 0
        self._synced = False
        self._graph_opt_shared = None      # the captured update (data parallel, two-graph form): independent of the batch shape
        # data parallel: DA_DP_CAPTURE_ALLREDUCE=1 tries to capture the all-reduce INSIDE the step graph (opt-in, see _capture)
        self._capture_allreduce = os.environ.get('DA_DP_CAPTURE_ALLREDUCE', '0') == '1'
        self.allreduce_in_graph = False
        self.grad_overwrite = False         # the form of the step being captured (set per capture)

    # ---- replicas ----------------------------------------------------------------------------
    def sync_replicas(self):
        """Data parallel only: rank 0's parameters and buffers (BatchNorm running statistics, the DenseNet dropout
        seed) replace every other rank's before the first update, in one flat broadcast per dtype.  The reference's
        ``nn.DataParallel`` (train_ards_detector.py:96) replicates module 0 every step; with one process per GPU the
        replicas only stay identical if they START identical -- ``get_model`` seeds only when a seed is given, so
        without this every rank would train its own initialisation and share nothing but gradients."""
        if self._synced or self.world_size == 1:
            self._synced = True
            return
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError('world_size %d needs an initialised torch.distributed process group' % self.world_size)
        tensors = [p.data for p in self.model.parameters()] + [b.data for b in self.model.buffers()]
        by_dtype = {}
        for t in tensors:
            by_dtype.setdefault(t.dtype, []).append(t)
        src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
        for dtype in sorted(by_dtype, key=str):
            ts = by_dtype[dtype]
            flat = torch.cat([t.reshape(-1) for t in ts])
            dist.broadcast(flat, src=src, group=self.group)
            off = 0
            for t in ts:
                n = t.numel()
                t.copy_(flat[off:off + n].view(t.shape))
                off += n
        self._synced = True

    def _allreduce(self):
        self.bucket.allreduce(self.group)
        self.allreduce_calls += 1

    # ---- eager pieces ------------------------------------------------------------------------
    def _forward_backward(self, inputs, target, zero_grad=False):
        """zero_grad: clear the flat gradient bucket first (optimizer.zero_grad(), train_ards_detector.py:164)."""
        if zero_grad:
            self.bucket.zero_grad()
        with F_.training_step(self.model):               # weights are constant within one step
            fused = self.model.forward_loss(inputs, target) if _FUSED_HEAD and hasattr(self.model, 'forward_loss') else None
            if fused is not None:                        # CNNLinearNetwork: pool + linear + loss (+ their backward) in 2 launches
                loss, logits = fused
                F_.flush_forward(defer=True)        # (the running-statistics updates ride on the tail launch)
                torch.autograd.backward(loss, grad_tensors=self._one(loss))
                F_.flush_backward()
                return loss, logits
            logits = _logits(self.model(inputs, None))
            F_.flush_forward(defer=True)                 # BN running-statistics updates: on flush_backward's launch
            lg, tg = _loss_operands(logits.detach(), target)
            loss, dlogits = H.bce_logits(lg, tg, want_grad=True)
            logits.backward(dlogits.view(logits.shape))
            F_.flush_backward()                          # batched dgamma/dbeta folds + wgrad slab reductions
        return loss, logits.detach()

    def _one(self, like):
        """A constant 1 of the loss's shape (autograd would launch a fill kernel for an implicit one every step)."""
        one = getattr(self, '_one_t', None)
        if one is None or one.device != like.device or one.shape != like.shape:
            one = self._one_t = torch.ones_like(like)
        return one

    def _optimizer_step(self):
        b = self.bucket
        gscale = 1.0 / self.world_size
        if self.optimizer == 'sgd':
            first = 'buf' not in self.state
            if first:
                self.state['buf'] = torch.empty_like(b.p)
            H.clamp_sgd_nesterov_(b.p, b.g, self.state['buf'], self.lr, self.momentum, self.wd, self.clip, first,
                                  gscale)
        else:
            if 'm' not in self.state:
                self.state['m'] = torch.zeros_like(b.p)
                self.state['v'] = torch.zeros_like(b.p)
                self.state['t'] = torch.zeros(1, dtype=torch.int64, device=b.p.device)   # step count on the device
            H.clamp_adam_dev_(b.p, b.g, self.state['m'], self.state['v'], self.lr, self.state['t'], self.clip,
                              gscale=gscale)

    def _first_step(self, inputs, target):
        """Eager step that discovers the live parameters (the reference's optimiser skips
        parameters whose grad is None -- resnet's conv1_alt/conv2/bn2, SURVEY.md finding 6)."""
        model = self.model
        model.train()
        self.sync_replicas()
        for p in model.parameters():
            p.grad = None
        loss, logits = self._forward_backward(inputs, target)
        live = [p for p in model.parameters() if p.requires_grad and p.grad is not None]
        self.bucket = FlatBucket(live)
        if self.world_size > 1:
            self._allreduce()
        self._optimizer_step()
        return loss, logits

    def _eager_step(self, inputs, target):
        loss, logits = self._forward_backward(inputs, target, zero_grad=True)
        if self.world_size > 1:
            self._allreduce()
        self._optimizer_step()
        return loss, logits

    # ---- graph replay ------------------------------------------------------------------------
    def _capture(self, inputs, target):
        """Capture the step for this batch shape.  Every shape keeps its own graph (an epoch's tail batch has another
        shape than the full ones: it is captured once, not twice per epoch)."""
        static = (inputs.clone(), target.clone())
        # Warm the caching allocator on a side stream with a forward+backward whose side effects
        # (BN running stats, dropout seed) are rolled back, so capture adds no training step.
        saved = [b.clone() for b in self.model.buffers()]
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        F_._OV['ok'] = True
        with torch.cuda.stream(s):
            self._forward_backward(*static, zero_grad=True)
            for b, c in zip(self.model.buffers(), saved):
                b.copy_(c)
        torch.cuda.current_stream().wait_stream(s)
        # Did every gradient destination get exactly one write, by a writer with an overwrite form (functional._OV)?  Then
        # the captured step needs no zero-fill of the gradient bucket: its writers overwrite (6 us a step at B = 64).
        self.grad_overwrite = _GRAD_OVERWRITE and F_._OV['ok']
        try:
            F_.grad_overwrite(self.grad_overwrite)
            return self._capture_forms(inputs, static)
        finally:
            F_.grad_overwrite(False)

    def _capture_forms(self, inputs, static):
        graph, static_out, graph_opt = None, None, None
        if self.world_size > 1 and self._capture_allreduce and self._collectives_capturable():
            # Data parallel, OPT-IN form (DA_DP_CAPTURE_ALLREDUCE=1): ONE graph holds backward | all-reduce | update.  RCCL
            # collectives are stream-capturable; inside the graph the exchange needs no host round trip and no extra
            # launches per step.  The default stays the two-graph form (backward | eager all-reduce | update) until a
            # world >= 2 RCCL run of this form exists: so far only a world-1 group has carried it.  Three guards:
            #   * any refusal of the capture selects the two-graph form;
            #   * the captured graph must be ONE chain -- a process group that enqueues on its own stream would come
            #     back as a forked branch, and a hipGraphExec with parallel branches owns streams that die with another
            #     such exec (the segfault of DESIGN.md section 5); a graph with a branch is dropped, not replayed;
            #   * the ranks AGREE on the form (MIN over the group of "my capture is good"): one rank replaying the
            #     single graph while its peer waits in an eager all-reduce would deadlock the exchange.
            ok, why = True, ''
            try:
                graph = torch.cuda.CUDAGraph(keep_graph=True)    # (keeps the hipGraph: its topology is read below)
                with _capture_graph(graph):
                    static_out = self._eager_whole_step(*static)
                branches = graph_branch_count(graph)
                graph.instantiate()
                if branches:
                    ok, why = False, 'the capture holds %d forked branch(es)' % branches
                elif branches is None:
                    import warnings
                    warnings.warn('could not read the captured step graph\'s topology: single-chain form not verified')
            except Exception as e:                       # noqa: BLE001 -- any refusal selects the fallback
                ok, why = False, '%s: %s' % (type(e).__name__, e)
                torch.cuda.synchronize()
            agreed = self._all_ranks_agree(ok)
            if not agreed:
                import warnings
                warnings.warn('all-reduce not captured into the step graph (%s); using the two-graph form (backward | eager '
                              'all-reduce | update)' % (why or 'another rank could not'))
                self._capture_allreduce, graph, static_out = False, None, None
            self.allreduce_in_graph = agreed
        if graph is None:
            graph = torch.cuda.CUDAGraph()
            with _capture_graph(graph):
                static_out = self._eager_single_gpu_parts(*static)
        if self.world_size > 1 and not self.allreduce_in_graph:
            graph_opt = self._graph_opt_shared
            if graph_opt is None:                        # the update does not depend on the batch shape: one graph
                graph_opt = torch.cuda.CUDAGraph()
                with _capture_graph(graph_opt):
                    self._optimizer_step()
                self._graph_opt_shared = graph_opt
        ent = self._graphs[tuple(inputs.shape)] = (graph, static, static_out, graph_opt)
        return ent

    def _all_ranks_agree(self, ok):
        """MIN over the trainer's process group of ``ok``: every rank takes the same form of the captured step."""
        import torch.distributed as dist
        flag = torch.tensor([1 if ok else 0], device=self.bucket.p.device, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        return bool(int(flag.item()))

    def _collectives_capturable(self):
        """Only RCCL ('nccl') collectives are stream operations a hipGraph can hold; a gloo all-reduce synchronises the
        host, and a capture it invalidates cannot be retried in the same process -- so the backend decides, not a trial."""
        import torch.distributed as dist
        try:
            return dist.is_initialized() and dist.get_backend(self.group) == 'nccl'
        except Exception:                                # noqa: BLE001
            return False

    def _eager_whole_step(self, inputs, target):
        """zero-grad, forward, loss, backward, gradient all-reduce, update: the data-parallel step as one capturable chain."""
        loss, logits = self._forward_backward(inputs, target, zero_grad=not self.grad_overwrite)
        self.bucket.allreduce(self.group)
        self._optimizer_step()
        return loss, logits

    def _eager_single_gpu_parts(self, inputs, target):
        loss, logits = self._forward_backward(inputs, target, zero_grad=not self.grad_overwrite)
        if self.world_size == 1:
            self._optimizer_step()
        return loss, logits

    # ---- public ------------------------------------------------------------------------------
    def train_step(self, inputs, target):
        """inputs (B_local, NB, 1, 224) float32 CUDA, target (B_local, 2) one-hot float32 CUDA.
        Returns the device-resident loss of this rank's shard: with use_graph it is the captured step's own output
        buffer, overwritten by the next step -- ``.clone()`` it to keep it (the epoch functions do)."""
        if not inputs.is_cuda:
            raise RuntimeError('HotPathTrainer needs CUDA (MI355X) tensors; there is no CPU fallback')
        if not self.model.training:                       # (a recursive walk over every module: 0.26 ms a step when unconditional)
            self.model.train()
        if self.bucket is None:
            loss, logits = self._first_step(inputs, target)
        elif not self.use_graph:
            loss, logits = self._eager_step(inputs, target)
        else:
            ent = self._graphs.get(tuple(inputs.shape))
            if ent is None:
                # one eager warm step already happened (_first_step); capture now
                ent = self._capture(inputs, target)
            graph, static, static_out, graph_opt = self._graph = ent
            self._static = static
            if inputs is not static[0]:                  # batches built in place (static_batch()) skip the copies
                static[0].copy_(inputs)
            if target is not static[1]:
                static[1].copy_(target)
            graph.replay()
            if self.world_size > 1:
                if graph_opt is None:                    # the all-reduce is a node of the step graph
                    self.allreduce_calls += 1
                else:
                    self._allreduce()
                    graph_opt.replay()
            loss, logits = static_out
        self.steps += 1
        self.last_loss, self.last_logits = loss, logits
        return loss

    def snapshot(self):
        """Everything a train / test step changes, copied: parameters (the flat bucket), optimizer state, module buffers
        (BatchNorm running statistics, dropout seeds), the static batch of every captured graph and the counters.
        ``restore`` puts it back bit for bit -- replays in between (e.g. to time stream placements) leave no trace."""
        torch.cuda.synchronize()
        snap = {'steps': self.steps, 'allreduce_calls': getattr(self, 'allreduce_calls', 0),
                'buffers': [b.detach().clone() for b in self.model.buffers()],
                'state': {k: v.detach().clone() for k, v in self.state.items()},
                'static': {k: tuple(t.clone() for t in e[1]) for k, e in self._graphs.items()}}
        if self.bucket is not None:
            snap['p'], snap['g'] = self.bucket.p.clone(), self.bucket.g.clone()
        else:
            snap['params'] = [q.detach().clone() for q in self.model.parameters()]
        return snap

    def restore(self, snap):
        torch.cuda.synchronize()
        with torch.no_grad():
            if self.bucket is not None:
                self.bucket.p.copy_(snap['p'])
                self.bucket.g.copy_(snap['g'])
            else:
                for q, v in zip(self.model.parameters(), snap['params']):
                    q.copy_(v)
            for b, v in zip(self.model.buffers(), snap['buffers']):
                b.copy_(v)
            for k, v in snap['state'].items():
                self.state[k].copy_(v)
            for k, st in snap['static'].items():
                for dst, src in zip(self._graphs[k][1], st):
                    dst.copy_(src)
        self.steps = snap['steps']
        if hasattr(self, 'allreduce_calls'):
            self.allreduce_calls = snap['allreduce_calls']
        torch.cuda.synchronize()

    def release_graphs(self):
        """Drop every captured graph and its static buffers NOW (outside any capture): call before discarding a
        trainer so that its graphs are not left to a later garbage-collection pass."""
        torch.cuda.synchronize()
        self._graphs.clear()
        self._test_graphs.clear()
        self._graph = self._static = self._graph_opt_shared = None
        self.last_loss = self.last_logits = None

    def static_batch(self, n_windows=None):
        """(inputs, target) buffers the captured training step reads (None before the capture): those of the graph for
        batches of ``n_windows`` windows, or of the step run last.  A producer that writes the next batch into them
        (DeviceTileStore.batch(..., out=...)) and passes them to train_step saves the two device copies per step."""
        if n_windows is None:
            return self._static
        for shape, ent in self._graphs.items():
            if shape[0] == n_windows:
                return ent[1]
        return None

    def _test_forward(self, inputs, target):
        with torch.no_grad(), F_.training_step(self.model):      # packs / Winograd taps once, batched small kernels
            fused = self.model.forward_loss(inputs, target) if _FUSED_HEAD and hasattr(self.model, 'forward_loss') else None
            if fused is not None:
                loss, logits = fused
                F_.flush_forward()
                return loss, logits, logits.argmax(dim=-1)
            logits = _logits(self.model(inputs, None))
            F_.flush_forward()                                   # train-mode forward: BN running statistics do move
            loss, _ = H.bce_logits(*_loss_operands(logits, target), want_grad=False)
        return loss, logits, logits.argmax(dim=-1)

    def test_step(self, inputs, target):
        """run_test_epoch body: no_grad forward with train-mode modules, loss, argmax predictions.  With use_graph
        the step is captured once per batch shape and replayed; the returned tensors are copies, safe to hold across
        calls (the next replay overwrites the graph's own output buffers)."""
        if not inputs.is_cuda:
            raise RuntimeError('HotPathTrainer needs CUDA (MI355X) tensors; there is no CPU fallback')
        if not self.model.training:                       # (a recursive walk over every module: 0.26 ms a step when unconditional)
            self.model.train()
        if not self.use_graph:
            return self._test_forward(inputs, target)
        key = (tuple(inputs.shape), tuple(target.shape))
        ent = self._test_graphs.get(key)
        if ent is None:
            static = (inputs.clone(), target.clone())
            saved = [b.clone() for b in self.model.buffers()]
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):                           # allocator warm-up, side effects rolled back
                self._test_forward(*static)
                for b, c in zip(self.model.buffers(), saved):
                    b.copy_(c)
            torch.cuda.current_stream().wait_stream(s)
            g = torch.cuda.CUDAGraph()
            with _capture_graph(g):
                out = self._test_forward(*static)
            ent = self._test_graphs[key] = (g, static, out)
        g, static, out = ent
        static[0].copy_(inputs)
        static[1].copy_(target)
        g.replay()
        return tuple(o.clone() for o in out)                     # never hand out the graph's own buffers


def run_train_epoch(trainer, loader, batch_size=None):
    """BaseTraining.run_train_epoch (:139-159) over an iterable of (obs_idx, seq, metadata, target) -- the hand-over
    of the reference (host batches from a DataLoader).  Every rank must iterate the SAME batches (same loader seed);
    each takes its window shard.  Returns the list of device-resident per-batch losses (no per-step host sync)."""
    losses = []
    dev = next(trainer.model.parameters()).device
    for obs_idx, seq, metadata, target in loader:
        n = legal_batch_len(seq.shape[0], batch_size, trainer.world_size)
        if n == 0:
            continue
        sl = shard_windows(n, trainer.world_size, trainer.rank)
        inputs = seq[:n][sl].float().to(dev, non_blocking=True)
        tgt = target[:n][sl].float().to(dev, non_blocking=True)
        losses.append(trainer.train_step(inputs, tgt).clone())
    return losses


def place_replicas_on_streams(trainers, rounds=3, candidates=10):
    """Streams for ``trainers`` (each with a captured train step) so that their replays overlap best: HIP deals streams
    -- the ones a graph launch runs on and the ones a graph's forked branches use internally -- onto a few hardware
    queues, two streams on one queue run one after the other, and which streams share a queue depends on everything the
    process created before.  So the placement is MEASURED with the real thing: every trainer's state is snapshotted,
    candidate streams are tried one trainer at a time with a few replays of all the steps, the fastest placement is
    kept, and the snapshots are restored bit for bit (the trial leaves no trace in parameters, optimizer state,
    running statistics or static batches)."""
    import time
    snaps = [tr.snapshot() for tr in trainers]
    statics = [tr.static_batch() for tr in trainers]
    if any(st is None for st in statics):
        raise ValueError('place_replicas_on_streams: every trainer needs a captured step (run two steps first)')

    def rounds_time(streams):
        for tr, sn in zip(trainers, snaps):             # every trial from the same state: same kernels, same data
            tr.restore(sn)
        t0 = time.perf_counter()
        for _ in range(rounds):
            for tr, st, s in zip(trainers, statics, streams):
                with torch.cuda.stream(s):
                    tr.train_step(st[0], st[1])
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    streams = [torch.cuda.Stream() for _ in trainers]
    rounds_time(streams)                                 # warm
    best = rounds_time(streams)
    for i in range(1, len(trainers)):                    # trainer 0 keeps its stream; the others try the candidates in turn
        for _ in range(candidates):
            trial = list(streams)
            trial[i] = torch.cuda.Stream()
            t = rounds_time(trial)
            if t < best * 0.98:
                best, streams = t, trial
    for tr, sn in zip(trainers, snaps):
        tr.restore(sn)
    return streams


def concurrent_streams(n, spin_cycles=1500000, candidates=None):
    """``n`` HIP streams whose work really overlaps.  HIP deals streams onto a small number of hardware queues
    (GPU_MAX_HW_QUEUES, 4 by default) and two streams on ONE queue run their kernels one after the other -- which pair
    shares a queue depends on every stream the process made before.  Measured, not guessed: a one-block spin kernel
    (torch.cuda._sleep) on two candidates takes one spin when they overlap and two when they do not.  Falls back to
    plain new streams when fewer than ``n`` overlapping ones are found."""
    import time
    cands = [torch.cuda.Stream() for _ in range(candidates or 3 * n + 2)]

    def spin(streams):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in streams:
            with torch.cuda.stream(s):
                torch.cuda._sleep(spin_cycles)
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    spin(cands[:1])
    one = min(spin(cands[:1]) for _ in range(3))
    chosen = [cands[0]]
    for c in cands[1:]:
        if len(chosen) == n:
            break
        if all(spin([c, s]) < 1.5 * one for s in chosen):
            chosen.append(c)
    for c in cands:
        if len(chosen) == n:
            break
        if c not in chosen:
            chosen.append(c)
    return chosen


def shared_generator(trainer, generator=None):
    """The generator an epoch's permutation is drawn from.  One GPU: the caller's (None = torch's global RNG, like a
    DataLoader).  Data parallel: rank 0 draws a seed (from its generator or its global RNG), broadcasts it, and every
    rank seeds a fresh generator with it -- the ranks shard ONE permutation even when they were started with different
    seeds or none (``generator=None`` on every rank would give every rank its own shuffle)."""
    if trainer.world_size == 1:
        return generator
    import torch.distributed as dist
    seed = torch.randint(0, 2 ** 62, (1,), generator=generator, dtype=torch.int64)
    dev = next(trainer.model.parameters()).device
    backend = dist.get_backend(trainer.group)
    t = seed.to(dev) if backend == 'nccl' else seed
    src = dist.get_global_rank(trainer.group, 0) if trainer.group is not None else 0
    dist.broadcast(t, src=src, group=trainer.group)
    return torch.Generator().manual_seed(int(t.item()))


def run_train_epoch_from_store(trainer, store, batch_size=16, shuffle=True, generator=None):
    """One training epoch straight from a DeviceTileStore: the DataLoader / collate / cast / H2D of
    train_ards_detector.py:139-152 is one gather+normalise kernel per batch, written IN PLACE into the buffers the
    captured step reads once they exist (batches of the captured shape), so a steady-state step is: gather kernel,
    graph replay.  ``batch_size`` is the GLOBAL batch (what the reference's DataParallel scatters): each rank takes
    its window shard of every batch of one shared permutation.  Returns the device-resident per-batch losses."""
    losses = []
    if shuffle:
        generator = shared_generator(trainer, generator)
    for mine in epoch_shards_on_device(store, batch_size, shuffle, generator, trainer.world_size, trainer.rank):
        static = trainer.static_batch(len(mine))
        x, t = store.batch_from_device(mine, out=static) if static is not None else store.batch_from_device(mine)
        losses.append(trainer.train_step(x, t).clone())
    return losses


def epoch_shards_on_device(store, batch_size, shuffle, generator, world_size=1, rank=0):
    """This rank's window shard of every batch of the epoch, as slices of ONE device tensor of absolute indices (the
    epoch's batches of ``_epoch_indices`` concatenated, uploaded once): no host-to-device copy and no host wait per step."""
    batches = [idx for idx, _, _ in _epoch_indices(store, batch_size, shuffle, generator, world_size)]
    if not batches:
        return []
    mine = [idx[shard_windows(len(idx), world_size, rank)] for idx in batches]
    dev = store.device_indices(torch.cat(mine))
    out, s = [], 0
    for m in mine:
        out.append(dev[s:s + len(m)])
        s += len(m)
    return out


def _epoch_indices(store, batch_size, shuffle, generator, world_size=1):
    """Batches of fold-relative indices like DataLoader(batch_size, shuffle) + clip_odd_batch_sizes (:146-147,482-494),
    each trimmed to a multiple of lcm(2, world) windows (``legal_batch_len``)."""
    n = len(store)
    if batch_size == 1 and world_size > 1:
        raise ValueError('batch_size 1 cannot be sharded over %d ranks' % world_size)
    order = torch.randperm(n, generator=generator) if shuffle else torch.arange(n)
    for s in range(0, n, batch_size):
        idx = order[s:s + batch_size]
        idx = idx[:legal_batch_len(len(idx), batch_size, world_size)]
        if len(idx):
            yield idx, None, None


def run_test_epoch(trainer, store, patient_slot, batch_size=16, shuffle=False, generator=None):
    """BaseTraining.run_test_epoch (:424-465) + record_final_epoch_testing_results (:519-524) with the reductions on
    the device: no_grad forward in train mode, BCE loss, window argmax, per-patient vote table.  ``store`` is a
    DeviceTileStore, ``patient_slot`` an int64 tensor (len(store),) mapping every window to a patient slot
    0..P-1 (indexed by the ABSOLUTE window index, like the reference's ground-truth frame).  One host sync at the end.
    Returns dict(votes (P,2), pred_frac (P,), prediction (P,), window_pred, window_index (fold-relative),
    window_abs_index (index into all windows = the reference's obs_idx, dataset.py:1349-1350,1404), mean_loss) mirroring metrics.py:572-604: pred_frac = ARDS votes / all votes, prediction = argmax of the votes."""
    steps = test_epoch_steps(trainer, store, patient_slot, batch_size, shuffle=shuffle, generator=generator)
    for _ in steps:
        pass
    return steps.result()


class test_epoch_steps(object):
    """``run_test_epoch`` as an iterator: every ``next`` enqueues ONE test step (gather, forward, loss, vote kernel) on
    the current stream and returns without a host sync; ``result()`` after the last one reads the reductions back.
    The fold loop with folds in flight walks several of these round-robin, each under its own stream.

    ``shuffle``: the reference builds the TEST DataLoader with ``shuffle=True`` too unless ``--unshuffled``
    (train_ards_detector.py:333-338; evaluate.py: ``DataLoader(test_dataset, 16, True)``), so its test batches mix
    patients; with shuffle the epoch walks one permutation (``generator``, or torch's global RNG like a DataLoader; under
    data parallelism rank 0's draw is shared).  BatchNorm statistics are per WINDOW on this path (SURVEY finding 3), so the
    batch composition does not change a window's prediction beyond fp32 summation order; what the order does decide is
    the order of ``window_pred`` / ``window_index`` (the reference's ``preds`` / ``pred_idx`` lists), the sequence of the
    ResNet running-statistics updates and which dropout draw a DenseNet window gets.  ``window_abs_index`` keys every
    result by the absolute window, so the per-patient votes are order-independent."""

    def __init__(self, trainer, store, patient_slot, batch_size=16, shuffle=False, generator=None):
        self.trainer, self.store = trainer, store
        dev = store.tiles.device
        host_slot = torch.as_tensor(patient_slot, dtype=torch.int64)
        n_pat = int(host_slot.max()) + 1 if not host_slot.is_cuda else int(host_slot.max().item()) + 1
        self.slot = host_slot.to(dev)
        self.votes = torch.zeros((n_pat, 2), dtype=torch.int32, device=dev)
        self.preds, self.losses, self.order, self.absolute = [], [], [], []
        # the epoch's batches (DataLoader(batch_size, shuffle) order + clip_odd_batch_sizes when the model asks for
        # it); their absolute indices are uploaded ONCE, the steps take device slices: no host-to-device copy per step
        n, drop_odd = len(store), trainer_clip_odd_batches(trainer)
        if shuffle:
            order = torch.randperm(n, generator=shared_generator(trainer, generator))
        else:
            order = torch.arange(n)
        self.rel = []
        for s0 in range(0, n, batch_size):
            idx = order[s0:min(n, s0 + batch_size)]
            if drop_odd and batch_size != 1 and len(idx) % 2 == 1:
                idx = idx[:-1]
            if len(idx):
                self.rel.append(idx)
        self.abs_dev = store.device_indices(torch.cat(self.rel)) if self.rel else None
        self.pos, self.i = 0, 0

    def __iter__(self):
        return self

    def __next__(self):
        if self.i >= len(self.rel):
            raise StopIteration
        idx = self.rel[self.i]
        self.i += 1
        gidx = self.abs_dev[self.pos:self.pos + len(idx)]
        self.pos += len(idx)
        x, t = self.store.batch_from_device(gidx)
        loss, logits, _ = self.trainer.test_step(x, t)
        grp = self.slot[gidx]
        if logits.dim() == 3:                       # per-breath heads: every breath votes for its window's patient
            nb = logits.shape[1]                    # (PerBreathClassifierMixin, train_ards_detector.py:548-555)
            grp = grp.repeat_interleave(nb)
            idx = idx.repeat_interleave(nb)
            gidx = gidx.repeat_interleave(nb)
            logits = logits.reshape(-1, 2)
        self.preds.append(H.vote_counts(logits.contiguous(), grp, self.votes))
        self.losses.append(loss.reshape(1))
        self.order.append(idx)
        self.absolute.append(gidx)
        return self.i

    def result(self):
        v = self.votes.cpu().numpy()
        tot = v.sum(axis=1)
        return dict(votes=v, pred_frac=v[:, 1] / tot.clip(min=1), prediction=v.argmax(axis=1),
                    window_pred=torch.cat(self.preds).cpu().numpy(), window_index=torch.cat(self.order).numpy(),
                    window_abs_index=torch.cat(self.absolute).cpu().numpy(),
                    mean_loss=float(torch.cat(self.losses).mean()))


def trainer_clip_odd_batches(trainer):
    """BaseTraining.clip_odd_batches is False for the cnn_linear model: test batches keep their odd item (:449-450)."""
    return getattr(trainer, 'clip_odd_batches', False)
