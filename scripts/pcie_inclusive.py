"""PCIe-inclusive rate: the reference's hand-over (host float64 batches from a DataLoader, train_ards_detector.py:144-152)
through run_train_epoch (cast + H2D per batch) vs the device-resident store."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import deepards_amd.models as M
from deepards_amd.train import HotPathTrainer, run_train_epoch, run_train_epoch_from_store
from deepards_amd.data import DeviceTileStore
torch.manual_seed(0)
torch.set_num_threads(min(16, os.cpu_count() or 1))     # the box's cgroup has 16 CPUs behind a 256-CPU affinity mask
B, NBATCH = 64, 60
model = M.CNNLinearNetwork(M.resnet18(), 20, 0).cuda()
tr = HotPathTrainer(model)
rng = np.random.default_rng(0)
tiles = rng.standard_normal((B * NBATCH, 20, 1, 224)) * 28 + 2          # raw float64 windows
tgt = np.zeros((B * NBATCH, 2)); tgt[np.arange(B * NBATCH), rng.integers(0, 2, B * NBATCH)] = 1
host = [(torch.arange(B), torch.from_numpy((tiles[i * B:(i + 1) * B] - 2.0) / 28.0), None, torch.from_numpy(tgt[i * B:(i + 1) * B]))
        for i in range(NBATCH)]                                          # what the DataLoader yields: float64 CPU tensors
run_train_epoch(tr, host[:4])
torch.cuda.synchronize(); t0 = time.perf_counter()
run_train_epoch(tr, host)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print('host-fed (float64 -> .float() -> H2D per batch): %.3f ms/step, %.0f breath-seq/s' % (dt / NBATCH * 1e3, B * 20 * NBATCH / dt))
store = DeviceTileStore(tiles, tgt, 2.0, 28.0)
run_train_epoch_from_store(tr, store, batch_size=B, shuffle=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
run_train_epoch_from_store(tr, store, batch_size=B, shuffle=True)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print('device tile store (gather+normalise kernel in place): %.3f ms/step, %.0f breath-seq/s' % (dt / NBATCH * 1e3, B * 20 * NBATCH / dt))
