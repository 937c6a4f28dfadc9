"""CPU tests (no GPU): the C-ABI library loads and exports every symbol the header declares, host-side
logic of the training loop, the model surface (state_dict keys / attributes of the reference), and
the world_size-2 data-parallel path over gloo."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_loads_and_exports_header_symbols():
    from deepards_amd import _lib
    _lib.build()
    lib = _lib.lib()
    syms = _lib.header_symbols()
    assert len(syms) >= 20
    assert set(syms) == set(_lib.SIGNATURES)
    for s in syms:
        assert hasattr(lib, s), s
    assert lib.da_version() >= 100
    # pure host helpers of the ABI (no GPU needed)
    assert lib.da_conv_wgrad_workspace(1280, 56, 64, 64, 3) % (3 * 64 * 64 * 4) == 0
    assert lib.da_stem_wgrad_workspace(1280, 64) == 512 * 64 * 7 * 4


def test_abi_structs_agree_between_header_library_and_binding(tmp_path):
    """The descriptor structs exist three times: in the public header, in the library's sources and as ctypes mirrors.
    sizeof of each must agree: header compiled by gcc, da_abi_sizes() of the built library, ctypes.sizeof."""
    import ctypes
    import subprocess
    from deepards_amd import _lib
    src = tmp_path / 'abi.c'
    src.write_text('#include <stdio.h>\n#include "deepards_hip.h"\nint main(void) { printf("%zu %zu %zu %zu %zu %zu\\n", '
                   'sizeof(da_wgrad_job), sizeof(da_conv_job), sizeof(da_wgrad_reduce_desc), sizeof(da_repack_desc), '
                   'sizeof(da_bn_running_desc), sizeof(da_bn_pgrad_desc)); return 0; }\n')
    exe = tmp_path / 'abi'
    subprocess.check_call(['gcc', '-std=c99', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)])
    header = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    out = (ctypes.c_int * 6)()
    _lib.lib().da_abi_sizes(out)
    binding = [ctypes.sizeof(c) for c in (_lib.WgradJob, _lib.ConvJob, _lib.WgradReduceDesc, _lib.RepackDesc,
                                          _lib.BnRunningDesc, _lib.BnPgradDesc)]
    assert header == list(out) == binding, (header, list(out), binding)


def test_product_path_refuses_cpu_tensors():
    import deepards_amd.models as M
    from deepards_amd import hip_ops as H
    model = M.CNNLinearNetwork(M.resnet18(), 20, 0)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        model(torch.zeros(2, 20, 1, 224), None)
    with pytest.raises(ValueError):
        H.bn_stats(torch.zeros(20, 56, 64), 20)
    # the product package never imports the oracle
    for name, mod in list(sys.modules.items()):
        if name.startswith('deepards_amd') and mod is not None:
            src = getattr(mod, '__file__', None)
            if src and src.endswith('.py'):
                assert 'oracle' not in open(src).read().replace('oracle/', '').replace('the oracle', ''), name


def test_model_surface_matches_reference_inventory():
    import deepards_amd.models as M
    from oracle.weights import param_spec
    for backbone, ctor, nkeys, nout in (('resnet18', M.resnet18, 129, 512), ('densenet18', M.densenet18, 64, 128)):
        bb = ctor()
        assert bb.network_name == backbone and bb.n_out_filters == nout
        model = M.CNNLinearNetwork(bb, 20, 0)
        assert model.seq_size == 224 and model.breath_block is bb
        assert tuple(model.linear_final.weight.shape) == (2, nout * 20)
        assert [n for n, _ in model.named_parameters()] == [s[0] for s in param_spec(backbone)]
        assert [tuple(p.shape) for _, p in model.named_parameters()] == [tuple(s[1]) for s in param_spec(backbone)]
        assert len(model.state_dict()) == nkeys
    d = M.densenet18()
    assert hasattr(d, 'features') and hasattr(d, 'avgpool') and hasattr(d, 'forward_no_pool')
    ks, st, pd = d.conv_info()
    assert len(ks) == len(st) == len(pd) == 24 and ks[:2] == [7, 3]
    assert all(not m.track_running_stats for m in d.modules() if isinstance(m, torch.nn.BatchNorm1d))
    assert M.base_networks['resnet18'] is M.resnet18
    with pytest.raises(Exception, match='sequence length of 224'):
        M.CNNLinearNetwork(M.resnet18(), 20, 0)(torch.zeros(1, 20, 1, 100), None)
    # reference init: conv ~ N(0, sqrt(2/(k*C_out))), BN gamma 1 / beta 0 (resnet.py:115-121)
    r = M.resnet18()
    w = r.layer3[0].conv1.weight
    assert abs(float(w.std()) - np.sqrt(2.0 / (3 * 256))) < 2e-3
    assert float(r.bn1.weight.min()) == 1.0 and float(r.bn1.bias.abs().max()) == 0.0


def test_host_helpers():
    from deepards_amd.train import clip_odd_batch_sizes, shard_windows
    idx, seq, meta, tgt = torch.arange(5), torch.zeros(5, 20, 1, 224), torch.zeros(5), torch.zeros(5, 2)
    a, b, c, d = clip_odd_batch_sizes(idx, seq, meta, tgt)
    assert a.shape[0] == b.shape[0] == c.shape[0] == d.shape[0] == 4
    a, b, c, d = clip_odd_batch_sizes(idx[:4], seq[:4], meta[:4], tgt[:4])
    assert b.shape[0] == 4
    assert shard_windows(64, 4, 1) == slice(16, 32)
    assert [shard_windows(8, 2, r) for r in range(2)] == [slice(0, 4), slice(4, 8)]
    with pytest.raises(ValueError):
        shard_windows(6, 4, 0)                       # untrimmed batches are a caller bug (epoch iterators trim first)


def test_epoch_batches_are_legal_for_every_world_size():
    """Every batch an epoch yields splits evenly over the ranks and keeps the reference's even-batch rule
    (train_ards_detector.py:146-147,482-494): N=70, batch 16, world 4 has a tail of 6 -> trimmed to 4 (1 per rank)
    instead of raising; the shards of a batch partition it."""
    from deepards_amd.train import _epoch_indices, shard_windows, legal_batch_len, batch_multiple

    class Store(object):
        def __init__(self, n):
            self.n = n

        def __len__(self):
            return self.n
    assert [batch_multiple(16, w) for w in (1, 2, 3, 4, 8)] == [2, 2, 6, 4, 8] and batch_multiple(1, 1) == 1
    assert legal_batch_len(6, 16, 4) == 4 and legal_batch_len(7, 16, 1) == 6 and legal_batch_len(1, 1, 1) == 1
    for n, bs in ((70, 16), (20, 6), (1000, 64), (9, 16)):
        for world in (1, 2, 4, 8):
            g = torch.Generator().manual_seed(3)
            seen = []
            for idx, _, _ in _epoch_indices(Store(n), bs, True, g, world):
                assert len(idx) % 2 == 0 and len(idx) % world == 0 and 0 < len(idx) <= bs
                parts = [idx[shard_windows(len(idx), world, r)] for r in range(world)]
                assert len({len(p) for p in parts}) == 1
                assert torch.equal(torch.cat(parts), idx)
                seen += idx.tolist()
            assert len(set(seen)) == len(seen)
            full, tail = divmod(n, bs)
            assert len(seen) == full * legal_batch_len(bs, bs, world) + legal_batch_len(tail, bs, world)
    sizes = [len(i) for i, _, _ in _epoch_indices(Store(70), 16, False, None, 4)]
    assert sizes == [16, 16, 16, 16, 4]
    with pytest.raises(ValueError):
        list(_epoch_indices(Store(10), 1, False, None, 2))


def test_capture_guard_holds_the_collector_off_and_restores_it():
    """deepards_amd.train._capture_graph: pending garbage is collected BEFORE the capture window opens, the cyclic
    collector is disabled inside it, and its previous state comes back afterwards -- also when the capture raises,
    and also when the collector was already off."""
    import contextlib
    import gc
    import weakref
    from deepards_amd.train import _capture_graph

    class Node(object):
        pass
    events = []

    @contextlib.contextmanager
    def fake_capture(graph):
        events.append(('enter', gc.isenabled(), ref() is None))
        yield
        events.append(('exit', gc.isenabled()))
    was = gc.isenabled()
    try:
        for state in (True, False):
            gc.enable() if state else gc.disable()
            n = Node()
            n.me = n
            ref = weakref.ref(n)
            del n                                              # cyclic garbage pending at the time of the capture
            events.clear()
            with _capture_graph(object(), _ctx=fake_capture):
                assert not gc.isenabled()
            assert events == [('enter', False, True), ('exit', False)]    # collected before the window, off inside
            assert gc.isenabled() == state
            with pytest.raises(KeyError):
                with _capture_graph(object(), _ctx=fake_capture):
                    raise KeyError('capture failed')
            assert gc.isenabled() == state
    finally:
        gc.enable() if was else gc.disable()


def _sync_worker(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from deepards_amd.train import HotPathTrainer, shared_generator, _epoch_indices, shard_windows
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import deepards_amd.models as M
    torch.manual_seed(100 + rank)                              # every rank its own initialisation
    model = M.CNNLinearNetwork(M.resnet18(), 20, 0)
    model.breath_block.bn1.running_mean.fill_(float(rank + 1))
    before = model.linear_final.weight.detach().clone()
    tr = HotPathTrainer(model, world_size=world, rank=rank)
    tr.sync_replicas()
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()] +
                     [b.detach().reshape(-1).float() for b in model.buffers()])

    class Store(object):
        def __len__(self):
            return 70
    torch.manual_seed(5000 + 17 * rank)                        # ... and its own global RNG
    perms = []
    for gen in (None, torch.Generator().manual_seed(rank)):    # no generator / a different generator per rank
        g = shared_generator(tr, gen)
        perms.append(torch.cat([i[shard_windows(len(i), world, rank)] for i, _, _ in _epoch_indices(Store(), 16, True, g, world)]))
        full = torch.cat([i for i, _, _ in _epoch_indices(Store(), 16, True, shared_generator(tr, gen), world)])
        perms.append(full)
    q.put((rank, flat.numpy(), [p.numpy() for p in perms], bool(torch.equal(before, model.linear_final.weight))))
    dist.barrier()
    dist.destroy_process_group()


def test_replica_sync_and_shared_permutation_gloo_world2():
    """HotPathTrainer.sync_replicas + shared_generator over gloo, world 2 (the host side of config C4): ranks that
    were initialised differently hold rank 0's parameters AND buffers afterwards; with no seed / different seeds
    they still draw ONE permutation per epoch and shard it disjointly."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_sync_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, f0, perms0, same0), (_, f1, perms1, same1) = res
    assert np.array_equal(f0, f1)                              # parameters and buffers identical after the sync
    assert same0 and not same1                                 # rank 0 kept its weights, rank 1 was overwritten
    for k in (0, 2):                                           # shards of one permutation: disjoint, equal sizes
        assert len(perms0[k]) == len(perms1[k]) and not set(perms0[k].tolist()) & set(perms1[k].tolist())
    for k in (1, 3):                                           # the full permutation every rank iterates is the same
        assert np.array_equal(perms0[k], perms1[k])
    assert not np.array_equal(perms0[1], perms0[3])            # a fresh seed per epoch


def _dp_worker(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from oracle import np_ref
    from oracle.weights import seeded_params, seeded_batch
    from deepards_amd.train import FlatBucket, shard_windows
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    params64 = {k: v.astype(np.float64) for k, v in seeded_params('densenet18', 0).items()}
    x, t = seeded_batch(world * 1, 20, 5)
    sl = shard_windows(x.shape[0], world, rank)
    out = np_ref.cnn_linear_forward_backward(params64, x[sl].astype(np.float64), t[sl].astype(np.float64),
                                             backbone='densenet18')
    names = sorted(out['grads'])
    tens = [torch.nn.Parameter(torch.from_numpy(params64[n].copy())) for n in names]
    for p, n in zip(tens, names):
        p.grad = torch.from_numpy(np.ascontiguousarray(out['grads'][n]))
    # FlatBucket is fp32 on the device of the params; here fp32 CPU
    tens32 = [torch.nn.Parameter(p.detach().float()) for p in tens]
    for p32, p in zip(tens32, tens):
        p32.grad = p.grad.float()
    bucket = FlatBucket(tens32)
    bucket.allreduce()
    gflat = (bucket.g / world).numpy().astype(np.float64)
    assert all(off % 64 == 0 for off in bucket.offsets)          # 256-B aligned segments
    # clamp AFTER the reduce, then the optimiser step -- identical on every rank
    newp, gs = [], []
    for n, p, off in zip(names, tens32, bucket.offsets):
        k = p.numel()
        gs.append(gflat[off:off + k])
        gi = np_ref.clamp_grad(gflat[off:off + k].reshape(p.shape), 0.01)
        pn, _ = np_ref.sgd_nesterov_step(params64[n], gi, None, first=True)
        newp.append(pn.ravel())
    q.put((rank, np.concatenate(gs), np.concatenate(newp), names))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_gloo_world2_matches_full_batch():
    """Sharding windows over 2 ranks + sum-all-reduce of the flat bucket + 1/W scale reproduces the
    full-batch gradient (BN never crosses windows; loss is a mean over equal shards)."""
    import torch.multiprocessing as mp
    from oracle import np_ref
    from oracle.weights import seeded_params, seeded_batch
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    res.sort(key=lambda r: r[0])
    (_, g0, p0, names), (_, g1, p1, _) = res
    assert np.array_equal(g0, g1) and np.array_equal(p0, p1)          # replicas stay bit-identical
    params64 = {k: v.astype(np.float64) for k, v in seeded_params('densenet18', 0).items()}
    x, t = seeded_batch(2, 20, 5)
    full = np_ref.cnn_linear_forward_backward(params64, x.astype(np.float64), t.astype(np.float64), backbone='densenet18')
    ref = np.concatenate([full['grads'][n].ravel() for n in names])
    assert np.abs(g0 - ref).max() < 1e-6 * (1 + np.abs(ref).max())


def _fold_group_worker(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from deepards_amd.train import FlatBucket, HotPathTrainer, gather_fold_results, make_fold_groups, shared_generator
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    group, gworld, grank, folds = make_fold_groups(2, 5)               # 2 fold groups x 2 ranks, 5 folds (config C4's shape)
    # the gradient exchange stays inside the sub-group
    p = torch.nn.Parameter(torch.zeros(70))
    p.grad = torch.full((70,), float(rank + 1))
    bucket = FlatBucket([p])
    bucket.allreduce(group)
    # replica sync + one shared permutation per sub-group
    import deepards_amd.models as M
    torch.manual_seed(100 + rank)
    model = M.CNNLinearNetwork(M.densenet18(), 20, 0)
    tr = HotPathTrainer(model, world_size=gworld, rank=grank, process_group=group)
    tr.sync_replicas()
    w = model.linear_final.weight.detach().clone()
    torch.manual_seed(900 + rank)
    perm = torch.randperm(12, generator=shared_generator(tr, None))
    # checkpoint-time averaging of the replicas' running statistics inside the sub-group
    from deepards_amd.train import average_replica_buffers
    bn = torch.nn.BatchNorm1d(4)
    bn.running_mean.fill_(float(rank))
    bn.num_batches_tracked.fill_(7 + rank)
    average_replica_buffers(bn, gworld, group)
    avg = (float(bn.running_mean[0]), int(bn.num_batches_tracked))
    res = {(f, 1): {'votes': np.full((3, 2), 10 * f + grank)} for f in folds}
    merged = gather_fold_results(res, grank == 0)
    q.put((rank, gworld, grank, folds, float(bucket.g[0]), w.numpy(), perm.numpy(),
           sorted(merged), {k: int(v['votes'][0, 0]) for k, v in merged.items()}, avg))
    dist.barrier()
    dist.destroy_process_group()


def test_fold_groups_times_data_parallel_subgroups_gloo_world4():
    """BASELINE config C4's shape on the host side: 4 ranks as 2 fold groups x 2 data-parallel ranks
    (train.fold_group_layout / make_fold_groups, --fold-groups): folds dealt round-robin to the groups, the gradient
    all-reduce, the replica sync and the shared permutation stay inside a sub-group, and every rank ends with every fold's
    patient results (gather_fold_results takes them from the group leaders)."""
    import torch.multiprocessing as mp
    from deepards_amd.train import fold_group_layout
    assert fold_group_layout(4, 3, 2, 5) == ([[0, 1], [2, 3]], 1, [1, 3])
    assert fold_group_layout(4, 0, 1, 5)[2] == [0, 1, 2, 3, 4] and fold_group_layout(4, 2, 4, 5) == ([[0], [1], [2], [3]], 2, [2])
    with pytest.raises(ValueError):
        fold_group_layout(4, 0, 3, 5)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_fold_group_worker, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [(r[1], r[2]) for r in res] == [(2, 0), (2, 1), (2, 0), (2, 1)]
    assert [r[3] for r in res] == [[0, 2, 4], [0, 2, 4], [1, 3], [1, 3]]
    assert [r[4] for r in res] == [3.0, 3.0, 7.0, 7.0]                   # 1 + 2 and 3 + 4: sums within the sub-groups
    assert np.array_equal(res[0][5], res[1][5]) and np.array_equal(res[2][5], res[3][5])      # each group holds ITS leader's weights
    assert not np.array_equal(res[0][5], res[2][5])
    assert np.array_equal(res[0][6], res[1][6]) and np.array_equal(res[2][6], res[3][6])      # one permutation per group
    assert [r[9] for r in res] == [(0.5, 7), (0.5, 7), (2.5, 9), (2.5, 9)]   # running stats: group mean; counters: the leader's
    for r in res:                                                         # every rank: all five folds, the leaders' numbers
        assert r[7] == [(f, 1) for f in range(5)]
        assert r[8] == {(f, 1): 10 * f for f in range(5)}


def test_driver_mirror_names_and_no_cpu_path():
    """deepards_amd.train_ards_detector keeps the reference driver's names (train_ards_detector.py:45-69, 73-512,
    925-939, 1410-1436) and refuses to run without a GPU."""
    from deepards_amd import train_ards_detector as T
    assert set(T.network_map) <= {'cnn_lstm', 'cnn_linear', 'cnn_double_linear', 'cnn_single_breath_linear',
                                  'cnn_linear_compr_to_rf', 'cnn_linear_to_mean'}
    assert T.network_map['cnn_linear'] is T.CNNLinearModel and 'resnet18' in T.base_networks
    for name in ('run_train_epoch', 'handle_train_optimization', 'get_splits', 'train_and_test', 'get_base_network',
                 'get_optimizer', 'run_test_epoch', 'get_model', 'clip_odd_batch_sizes', 'set_loss_criterion',
                 'calc_loss', 'get_network', '_process_test_batch_results'):
        assert callable(getattr(T.CNNLinearModel, name)), name
    args = T.make_args()
    assert (args.network, args.base_network, args.batch_size, args.optimizer, args.learning_rate, args.n_sub_batches,
            args.weight_decay, args.clip_val, args.epochs) == ('cnn_linear', 'densenet18', 16, 'sgd', 0.001, 20, 0.0001,
                                                               0.01, 10)                      # defaults.yml
    with pytest.raises(TypeError):
        T.make_args(not_an_argument=1)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        T.CNNLinearModel(T.make_args(cuda=False))
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match='no CPU fallback'):
            T.CNNLinearModel(T.make_args())
