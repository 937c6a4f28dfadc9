"""The CLI's k-fold loop on a synthetic dataset (N windows of (20, 1, 224), 40 patients), folds one after the other vs
--folds-in-flight 2 / 4: wall time of the whole run (training + test epochs) and per-fold results equal or not.
usage: python scripts/folds_in_flight_bench.py [N] [batch] [backbone] [kfolds]"""
import contextlib, io, os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from deepards_amd import train_ards_detector as T

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
BATCH = sys.argv[2] if len(sys.argv) > 2 else '64'
BB = sys.argv[3] if len(sys.argv) > 3 else 'resnet18'
KF = int(sys.argv[4]) if len(sys.argv) > 4 else 4
rng = np.random.RandomState(0)
x = (rng.randn(N, 20, 1, 224) * 28 + 2).astype(np.float32)
lab = rng.randint(0, 2, N)
tgt = np.zeros((N, 2), np.float32); tgt[np.arange(N), lab] = 1
path = os.path.join(tempfile.mkdtemp(), 'synthetic.npz')
np.savez(path, x=x, target=tgt, patient_slot=np.arange(N) % 40, hours=np.zeros((N, 20)), n_sub_batches=20,
         dataset_type='unpadded_centered_sequences', train=True, total_kfolds=-1, mu=np.float64(2.0), std=np.float64(28.0))
res = {}
for flight in (1, 2, KF, 1):
    argv = ['--cuda-no-dp', '--train-from-pickle', path, '--kfolds', str(KF), '-e', '2', '-b', BATCH, '--base-network', BB,
            '--seed', '3', '--clip-grad', '--folds-in-flight', str(flight), '--no-test-after-epochs']
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        cls, r = T.main(argv)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    losses = [r.get_meter('loss', f) for f in range(KF)]
    steps = sum(len(l) for l in losses)
    same = bool(res) and all(a == b for a, b in zip(losses, res[1][1]))
    res.setdefault(flight, (dt, losses))
    print('%s, %d folds in flight %%d: %%.2f s for %%d steps of B=%%s' % (BB, KF) % (flight, dt, steps, BATCH) + ' (%.0f breath-seq/s incl. set-up / captures); losses equal to the '
          'sequential run: %s' % (steps * int(BATCH) * 20 / dt, same if flight != 1 or same else '-'), flush=True)
    continue
    print('folds in flight %d: %.2f s for %d steps of B=%s (%.0f breath-seq/s incl. set-up / captures); losses equal to the '
          'sequential run: %s' % (flight, dt, steps, BATCH, steps * int(BATCH) * 20 / dt, same if flight != 1 or same else '-'), flush=True)
