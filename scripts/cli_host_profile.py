"""cProfile of the CLI's sequential fold loop on a synthetic dataset (host-side cost per step).
usage: python scripts/cli_host_profile.py [N] [batch]"""
import contextlib, cProfile, io, os, pstats, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from deepards_amd import train_ards_detector as T

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
BATCH = sys.argv[2] if len(sys.argv) > 2 else '16'
rng = np.random.RandomState(0)
x = (rng.randn(N, 20, 1, 224) * 28 + 2).astype(np.float32)
lab = rng.randint(0, 2, N)
tgt = np.zeros((N, 2), np.float32); tgt[np.arange(N), lab] = 1
path = os.path.join(tempfile.mkdtemp(), 'synthetic.npz')
np.savez(path, x=x, target=tgt, patient_slot=np.arange(N) % 40, hours=np.zeros((N, 20)), n_sub_batches=20,
         dataset_type='unpadded_centered_sequences', train=True, total_kfolds=-1, mu=np.float64(2.0), std=np.float64(28.0))
argv = ['--cuda-no-dp', '--train-from-pickle', path, '--kfolds', '2', '-e', '2', '-b', BATCH, '--base-network', 'resnet18',
        '--seed', '3', '--clip-grad', '--no-test-after-epochs']
pr = cProfile.Profile()
with contextlib.redirect_stdout(io.StringIO()):
    pr.enable()
    T.main(argv)
    pr.disable()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(28)
