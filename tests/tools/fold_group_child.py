"""One rank of the --fold-groups rehearsal on ONE GPU (tests/test_model_gpu.py::test_fold_groups_on_the_hip_path): a FRESH
process per rank, all on cuda:0, gloo between them (RCCL refuses several ranks on one device):

    RANK=r WORLD_SIZE=4 MASTER_ADDR=127.0.0.1 MASTER_PORT=p python tests/tools/fold_group_child.py <out.npz> <argv...>

runs deepards_amd.train_ards_detector.main(argv) -- the reference's k-fold protocol with the folds dealt to fold groups and
every fold trained data-parallel inside its group -- and writes every fold's patient votes / losses as this rank ends up
holding them."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    out_path, argv = sys.argv[1], sys.argv[2:]
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from deepards_amd import train_ards_detector as T
    cls, res = T.main(argv)
    out = {}
    for (fold, ep), r in res.patient_results.items():
        out['votes/%d/%d' % (fold, ep)] = np.asarray(r['votes'])
        out['pred/%d/%d' % (fold, ep)] = np.asarray(r['window_pred'])
        out['loss/%d/%d' % (fold, ep)] = np.asarray(r['mean_loss'], dtype=np.float64)
    np.savez(out_path, **out)


if __name__ == '__main__':
    main()
