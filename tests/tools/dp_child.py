"""One rank of the data-parallel rehearsal used by tests/test_model_gpu.py::test_data_parallel_two_processes_*.

Launched as a FRESH process per rank (never re-exec'd from a process that touched the GPU):

    RANK=r WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=p python tests/tools/dp_child.py <golden.npz> <out.npz> <mode>

Both ranks share cuda:0; gradients travel through gloo (RCCL refuses two ranks on one device).  Every rank starts
from a DIFFERENT initialisation (rank 0: the golden's seeded weights, rank 1: torch.manual_seed(1234 + rank) random
init) -- HotPathTrainer.sync_replicas must make them identical before the first update.

mode 'traj':   3 SGD steps on the golden batch (B windows, rank r takes its shard) -> losses, final parameters
mode 'epoch':  2 shuffled epochs from a DeviceTileStore (20 fixture windows, global batch 6, seed None on every rank)
               -> the indices every step trained on, final parameters
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    gold_path, out_path, mode = sys.argv[1:4]
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import deepards_amd.models as M
    from deepards_amd.train import HotPathTrainer, shard_windows, run_train_epoch_from_store
    from oracle.weights import seeded_params

    g = np.load(gold_path, allow_pickle=False)
    backbone = str(g['backbone'])
    torch.manual_seed(1234 + rank)                                   # different random init per rank
    bb = M.resnet18(first_pool_type=str(g['first_pool_type'])) if backbone == 'resnet18' else M.densenet18(drop_rate=0.0)
    model = M.CNNLinearNetwork(bb, 20, 0)
    if rank == 0:                                                    # only rank 0 holds the golden's weights
        sd = {k: torch.from_numpy(v) for k, v in seeded_params(backbone, int(g['seed']),
                                                              bn_bias_shift=float(g['bn_bias_shift'])).items()}
        model.load_state_dict(sd, strict=False)
    model = model.cuda().train()
    use_graph = os.environ.get('DP_CHILD_GRAPH', '1') == '1'
    tr = HotPathTrainer(model, optimizer='sgd', world_size=world, rank=rank, use_graph=use_graph)
    out = {}
    if mode == 'traj':
        x = torch.from_numpy(g['x']).cuda()
        t = torch.from_numpy(g['target']).cuda()
        sl = shard_windows(x.shape[0], world, rank)
        losses = [float(tr.train_step(x[sl].contiguous(), t[sl].contiguous())) for _ in range(3)]
        out['losses'] = np.array(losses)
    else:
        from deepards_amd.data import DeviceTileStore
        z = np.load(os.path.join(ROOT, 'tests', 'golden', 'test_dataset_windows.npz'))
        store = DeviceTileStore(z['x'], z['target'], float(z['mu']), float(z['std']))
        seen = []
        orig = store.batch_from_device                               # what run_train_epoch_from_store gathers with

        def spy(abs_idx, out=None):                                  # no folds here: absolute == relative indices
            seen.append(np.asarray(abs_idx.cpu()).copy())
            return orig(abs_idx, out=out)
        store.batch_from_device = spy
        torch.manual_seed(777 + 13 * rank)                           # different global RNG per rank, generator=None
        losses = []
        for _ in range(2):
            losses += [float(l) for l in run_train_epoch_from_store(tr, store, batch_size=6, shuffle=True, generator=None)]
        out['losses'] = np.array(losses)
        out['n_steps'] = np.array(len(seen))
        for i, s in enumerate(seen):
            out['idx%d' % i] = s
        out['n_graphs'] = np.array(len(tr._graphs))
    torch.cuda.synchronize()
    for n, p in model.named_parameters():
        out['p/' + n] = p.detach().cpu().numpy()
    for n, b in model.named_buffers():
        if 'num_batches' not in n:
            out['b/' + n] = b.detach().cpu().numpy()
    out['allreduce_calls'] = np.array(tr.allreduce_calls)
    np.savez(out_path, **out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
