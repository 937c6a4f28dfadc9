"""Extract the arrays of the reference's test fixture WITHOUT unpickling it (build container only).

    python oracle/extract_fixture.py     # -> tests/golden/test_dataset_windows.npz

``/root/reference/deepards/tests/test_dataset.pkl`` is a pickled ``ARDSRawDataset`` (protocol 2).  Pickles
that ship inside the reference must not be loaded with anything that executes them, so this script only walks
the opcode stream with ``pickletools.genops`` (a pure parser: it constructs no objects and calls nothing) and
copies the raw numpy payloads: the 20 windows are the twenty 35 840-byte strings (20 x 1 x 224 float64), each
followed by its 16-byte one-hot target (2 x float64).  Patient identifiers are NOT exported; windows keep only
their order.  Scaling factors are the values SURVEY.md 8c records for this fixture.
"""
import os
import pickletools
import numpy as np

SRC = '/root/reference/deepards/tests/test_dataset.pkl'
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden',
                   'test_dataset_windows.npz')


def raw_payload(data, pos, name):
    if name == 'BINSTRING':                       # 'T' + uint32 length + bytes
        n = int.from_bytes(data[pos + 1:pos + 5], 'little')
        return data[pos + 5:pos + 5 + n]
    if name == 'SHORT_BINSTRING':                 # 'U' + uint8 length + bytes
        n = data[pos + 1]
        return data[pos + 2:pos + 2 + n]
    raise ValueError(name)


def main():
    data = open(SRC, 'rb').read()
    ops = [(op.name, pos) for op, arg, pos in pickletools.genops(data)]
    windows, targets = [], []
    for i, (name, pos) in enumerate(ops):
        if name != 'BINSTRING':
            continue
        raw = raw_payload(data, pos, name)
        if len(raw) != 20 * 224 * 8:
            continue
        windows.append(np.frombuffer(raw, dtype='<f8').reshape(20, 1, 224))
        for name2, pos2 in ops[i + 1:i + 200]:    # the next 16-byte string is this window's target
            if name2 == 'SHORT_BINSTRING':
                r2 = raw_payload(data, pos2, name2)
                if len(r2) == 16:
                    targets.append(np.frombuffer(r2, dtype='<f8').copy())
                    break
    x = np.stack(windows)
    t = np.stack(targets)
    assert x.shape == (20, 20, 1, 224) and t.shape == (20, 2), (x.shape, t.shape)
    assert set(map(tuple, t.tolist())) <= {(0, 1), (1, 0)}, t
    assert np.isfinite(x).all() and abs(x.mean() - 2.056) < 1.0 and abs(x.std() - 28.08) < 1.0   # ~ the fixture's scaling factors
    np.savez_compressed(OUT, x=x, target=t.astype(np.float32), mu=2.0560646853765587, std=28.08296533428954)
    print('windows', x.shape, 'range', x.min(), x.max(), 'ARDS', int(t[:, 1].sum()), 'non-ARDS', int(t[:, 0].sum()),
          os.path.getsize(OUT), 'B')


if __name__ == '__main__':
    main()
