"""Read an ``ARDSRawDataset`` pickle WITHOUT executing it -> plain arrays for the device tile store.

The reference's wire format for a prepared dataset is a pickle of the whole ``ARDSRawDataset`` object
(``pd.to_pickle(self, to_pickle)`` dataset.py:540-541, read back by ``ARDSRawDataset.from_pickle`` dataset.py:706-763
through ``pd.read_pickle``).  Unpickling runs whatever the file says (it imports ``deepards.dataset``, pandas
internals, numpy reconstructors ...), needs the reference package importable, and pickles shipped inside the reference
must not be loaded with anything that executes them.  This module therefore does what ``pickle.Unpickler`` does with
the stack and the memo, but **never imports or calls anything named by the file**: ``GLOBAL`` pushes an inert
``Global(module, name)`` marker, ``REDUCE`` / ``NEWOBJ`` / ``BUILD`` record "this callable, these arguments, this
state" in an inert ``Obj`` node.  Afterwards OUR code recognises the handful of shapes it understands -- numpy
``_reconstruct`` + ``__setstate__`` tuples, numpy scalars, py2 ``str`` payloads -- and turns exactly those into arrays
with ``np.frombuffer``.  Anything else (the pandas cohort frame, sklearn objects ...) stays an inert node nobody looks
at.  Object-dtype arrays are refused (their payload would be pickled Python objects).

    ds = read_ards_dataset('unpadded_centered_sequences-nb20-kfold.pkl')     # inert parse, nothing executed
    ds.save_npz('train.npz')                                                  # travels to the GPU box
    store = load_npz('train.npz').to_store()                                  # DeviceTileStore (+ k-fold plumbing)

What is taken from the object's state (dataset.py:384-421, 1343-1404): ``all_sequences`` -- a list of
``[patient_id, data (NB, C, L) float64, target (2,), seq_hours]`` (4 items; 5 with a metadata vector before the
target) --, ``scaling_factors`` ``{fold or None: (mu, std)}`` (scalars in old pickles, (NB, C, L) broadcasts in new ones),
``n_sub_batches``, ``dataset_type``, ``train``, ``total_kfolds``, ``kfold_patient_splits``.  Patient identifiers are
replaced by slots 0..P-1 in order of first appearance when exported with ``anonymise=True`` (the default).

CLI (run in the build container, the .npz is what gets committed / shipped):
    python -m deepards_amd.ingest <dataset.pkl> <out.npz> [--keep-patient-ids]
"""
import struct
import sys

import numpy as np


class Global(object):
    """An unresolved ``module.name`` reference of the pickle stream.  Never imported."""
    __slots__ = ('module', 'name')

    def __init__(self, module, name):
        self.module, self.name = module, name

    def __repr__(self):
        return 'Global(%s.%s)' % (self.module, self.name)

    def is_(self, modules, name):
        return self.name == name and self.module in modules


class Obj(object):
    """An object the stream asked to construct: ``func(*args)`` (REDUCE), ``cls.__new__(cls, *args)`` (NEWOBJ / OBJ /
    INST), then ``state`` (BUILD), appended items / set items.  Nothing is constructed."""
    __slots__ = ('how', 'func', 'args', 'kwargs', 'state', 'items', 'pairs')

    def __init__(self, how, func, args, kwargs=None):
        self.how, self.func, self.args, self.kwargs = how, func, args, kwargs
        self.state, self.items, self.pairs = None, [], []

    def __repr__(self):
        return 'Obj(%s %r)' % (self.how, self.func)


class _Mark(object):
    pass


class PickleFormatError(ValueError):
    pass


_MARK = _Mark()


def parse_pickle(data, start=0, persistent=False, return_end=False):
    """Interpret the opcode stream of ``data`` (bytes / memoryview) into plain containers, numbers, bytes / str
    payloads and inert ``Global`` / ``Obj`` nodes.  Protocols 0-5 opcodes that carry data are understood; extension
    registry codes and out-of-band buffers are refused, persistent ids too unless ``persistent`` (torch checkpoints
    name their storages that way): then a persistent id becomes the inert node ``Obj('persid', None, id_tuple)``.
    Python-2 ``str`` payloads come back as ``bytes`` (zero-copy slices of ``data`` for the long ones).
    ``start``: byte offset of the pickle inside ``data``; ``return_end``: -> (value, offset after STOP)."""
    buf = memoryview(data)
    n = len(buf)
    pos = start
    stack, memo = [], {}

    def need(k):
        if pos + k > n:
            raise PickleFormatError('truncated pickle at byte %d' % pos)

    def pop_mark():
        for i in range(len(stack) - 1, -1, -1):
            if stack[i] is _MARK:
                items = stack[i + 1:]
                del stack[i:]
                return items
        raise PickleFormatError('MARK not found')

    def readline():
        nonlocal pos
        end = pos
        while end < n and buf[end] != 0x0a:
            end += 1
        if end >= n:
            raise PickleFormatError('unterminated line argument')
        line = bytes(buf[pos:end])
        pos = end + 1
        return line

    def counted(fmt, size, as_bytes=True, encoding=None):
        nonlocal pos
        need(size)
        (k,) = struct.unpack_from(fmt, buf, pos)
        pos += size
        need(k)
        raw = buf[pos:pos + k]
        pos += k
        if encoding:
            return bytes(raw).decode(encoding, 'surrogatepass')
        return raw if k >= 4096 else bytes(raw)              # long payloads stay zero-copy views

    def setitems(target, pairs):
        if isinstance(target, dict):
            for k_, v_ in pairs:
                target[_hashable(k_)] = v_
        elif isinstance(target, Obj):
            target.pairs.extend(pairs)
        else:
            raise PickleFormatError('SETITEM on %r' % type(target))

    def append(target, items):
        if isinstance(target, list):
            target.extend(items)
        elif isinstance(target, Obj):
            target.items.extend(items)
        else:
            raise PickleFormatError('APPEND on %r' % type(target))

    while True:
        need(1)
        op = buf[pos]
        pos += 1
        c = chr(op)
        if c == '.':                                          # STOP
            if len(stack) != 1:
                raise PickleFormatError('stack holds %d items at STOP' % len(stack))
            return (stack[0], pos) if return_end else stack[0]
        elif c == 'Q' and persistent:                         # BINPERSID: the id stays an inert node
            stack.append(Obj('persid', None, stack.pop()))
        elif c == 'P' and persistent:                         # PERSID (text)
            stack.append(Obj('persid', None, readline().decode('utf-8')))
        elif op == 0x80:                                      # PROTO
            need(1)
            if buf[pos] > 5:
                raise PickleFormatError('pickle protocol %d' % buf[pos])
            pos += 1
        elif op == 0x95:                                      # FRAME
            need(8)
            pos += 8
        elif c == '(':
            stack.append(_MARK)
        elif c == 'N':
            stack.append(None)
        elif op == 0x88:
            stack.append(True)
        elif op == 0x89:
            stack.append(False)
        elif c == 'K':
            need(1)
            stack.append(buf[pos])
            pos += 1
        elif c == 'M':
            need(2)
            stack.append(struct.unpack_from('<H', buf, pos)[0])
            pos += 2
        elif c == 'J':
            need(4)
            stack.append(struct.unpack_from('<i', buf, pos)[0])
            pos += 4
        elif c in 'IL':                                       # text INT / LONG (protocol 0)
            line = readline().rstrip(b'L')
            stack.append(True if line == b'01' else False if line == b'00' else int(line))
        elif op == 0x8a or op == 0x8b:                        # LONG1 / LONG4
            size = 1 if op == 0x8a else 4
            need(size)
            k = buf[pos] if size == 1 else struct.unpack_from('<i', buf, pos)[0]
            pos += size
            need(k)
            stack.append(int.from_bytes(bytes(buf[pos:pos + k]), 'little', signed=True))
            pos += k
        elif c == 'G':
            need(8)
            stack.append(struct.unpack_from('>d', buf, pos)[0])
            pos += 8
        elif c == 'F':
            stack.append(float(readline()))
        elif c == 'U':                                        # SHORT_BINSTRING (py2 str)
            stack.append(counted('<B', 1))
        elif c == 'T':                                        # BINSTRING (py2 str)
            stack.append(counted('<i', 4))
        elif c == 'S':                                        # STRING (protocol 0, repr-quoted)
            line = readline()
            if len(line) < 2 or line[0] != line[-1] or line[:1] not in (b'"', b"'"):
                raise PickleFormatError('bad STRING argument')
            stack.append(line[1:-1].decode('unicode_escape').encode('latin-1'))
        elif c == 'C':                                        # SHORT_BINBYTES
            stack.append(counted('<B', 1))
        elif c == 'B':                                        # BINBYTES
            stack.append(counted('<I', 4))
        elif op == 0x8e:                                      # BINBYTES8
            stack.append(counted('<Q', 8))
        elif op == 0x96:                                      # BYTEARRAY8
            stack.append(counted('<Q', 8))
        elif op == 0x8c:                                      # SHORT_BINUNICODE
            stack.append(counted('<B', 1, encoding='utf-8'))
        elif c == 'X':                                        # BINUNICODE
            stack.append(counted('<I', 4, encoding='utf-8'))
        elif op == 0x8d:                                      # BINUNICODE8
            stack.append(counted('<Q', 8, encoding='utf-8'))
        elif c == 'V':                                        # UNICODE (protocol 0)
            stack.append(readline().decode('raw-unicode-escape'))
        elif c == ')':
            stack.append(())
        elif c == ']':
            stack.append([])
        elif c == '}':
            stack.append({})
        elif op == 0x8f:                                      # EMPTY_SET
            stack.append(set())
        elif op == 0x85:
            stack[-1:] = [(stack[-1],)]
        elif op == 0x86:
            stack[-2:] = [(stack[-2], stack[-1])]
        elif op == 0x87:
            stack[-3:] = [(stack[-3], stack[-2], stack[-1])]
        elif c == 't':
            stack.append(tuple(pop_mark()))
        elif c == 'l':
            stack.append(list(pop_mark()))
        elif c == 'd':
            items = pop_mark()
            d = {}
            for i in range(0, len(items), 2):
                d[_hashable(items[i])] = items[i + 1]
            stack.append(d)
        elif op == 0x91:                                      # FROZENSET
            stack.append(frozenset(_hashable(v) for v in pop_mark()))
        elif op == 0x90:                                      # ADDITEMS
            items = pop_mark()
            stack[-1].update(_hashable(v) for v in items)
        elif c == 'a':
            v = stack.pop()
            append(stack[-1], [v])
        elif c == 'e':
            items = pop_mark()
            append(stack[-1], items)
        elif c == 's':
            v = stack.pop()
            k = stack.pop()
            setitems(stack[-1], [(k, v)])
        elif c == 'u':
            items = pop_mark()
            setitems(stack[-1], [(items[i], items[i + 1]) for i in range(0, len(items), 2)])
        elif c == 'c':                                        # GLOBAL: two text lines; NOT imported
            stack.append(Global(readline().decode('utf-8'), readline().decode('utf-8')))
        elif op == 0x93:                                      # STACK_GLOBAL
            name = stack.pop()
            module = stack.pop()
            stack.append(Global(_text(module), _text(name)))
        elif c == 'R':                                        # REDUCE: recorded, NOT called
            args = stack.pop()
            func = stack.pop()
            stack.append(Obj('reduce', func, args))
        elif op == 0x81:                                      # NEWOBJ
            args = stack.pop()
            cls = stack.pop()
            stack.append(Obj('new', cls, args))
        elif op == 0x92:                                      # NEWOBJ_EX
            kwargs = stack.pop()
            args = stack.pop()
            cls = stack.pop()
            stack.append(Obj('new', cls, args, kwargs))
        elif c == 'o':                                        # OBJ
            items = pop_mark()
            stack.append(Obj('new', items[0], tuple(items[1:])))
        elif c == 'i':                                        # INST
            g = Global(readline().decode('utf-8'), readline().decode('utf-8'))
            stack.append(Obj('new', g, tuple(pop_mark())))
        elif c == 'b':                                        # BUILD: state recorded, __setstate__ NOT called
            state = stack.pop()
            tgt = stack[-1]
            if not isinstance(tgt, Obj):
                raise PickleFormatError('BUILD on %r' % type(tgt))
            tgt.state = state
        elif c == 'q':
            need(1)
            memo[buf[pos]] = stack[-1]
            pos += 1
        elif c == 'r':
            need(4)
            memo[struct.unpack_from('<I', buf, pos)[0]] = stack[-1]
            pos += 4
        elif c == 'p':
            memo[int(readline())] = stack[-1]
        elif op == 0x94:                                      # MEMOIZE
            memo[len(memo)] = stack[-1]
        elif c == 'h':
            need(1)
            stack.append(memo[buf[pos]])
            pos += 1
        elif c == 'j':
            need(4)
            stack.append(memo[struct.unpack_from('<I', buf, pos)[0]])
            pos += 4
        elif c == 'g':
            stack.append(memo[int(readline())])
        elif c == '0':
            stack.pop()
        elif c == '2':
            stack.append(stack[-1])
        elif c == '1':
            pop_mark()
        else:
            raise PickleFormatError('opcode %r at byte %d is not data (persistent ids, extension codes and out-of-band '
                                    'buffers are refused)' % (c, pos - 1))


def _text(v):
    if isinstance(v, (bytes, memoryview)):
        return bytes(v).decode('latin-1')
    return str(v)


def _hashable(k):
    if isinstance(k, memoryview):
        return bytes(k)
    if isinstance(k, list):
        return tuple(_hashable(v) for v in k)
    if isinstance(k, Obj):                                    # e.g. a numpy scalar used as a dict key
        v = resolve(k)
        return v.item() if isinstance(v, np.generic) else id(k)
    return k


# ---- recognising the numpy shapes ------------------------------------------------------------------------------------
_NP_MULTIARRAY = ('numpy.core.multiarray', 'numpy._core.multiarray', 'numpy.core._multiarray_umath',
                  'numpy._core._multiarray_umath')
_SAFE_KINDS = 'biufcSUV?'                                     # never 'O': object arrays hold pickled Python objects


def _dtype_of(node):
    """np.dtype from ``numpy.dtype(code, align, copy)`` + its state ``(version, byteorder, ...)``; plain types only."""
    if not (isinstance(node, Obj) and isinstance(node.func, Global) and node.func.is_(('numpy',), 'dtype')):
        raise PickleFormatError('not a numpy dtype: %r' % (node,))
    code = _text(node.args[0])
    dt = np.dtype(code)
    if dt.kind not in _SAFE_KINDS or dt.kind == 'V' or dt.hasobject:
        raise PickleFormatError('dtype %r is refused (object / structured payloads are not plain data)' % code)
    st = node.state
    if st is not None:
        order = _text(st[1])
        if dt.kind in 'SU' and len(st) >= 7 and isinstance(st[5], int) and st[5] > 0:
            dt = np.dtype('%s%d' % (dt.kind, st[5] // (4 if dt.kind == 'U' else 1)))
        if order in '<>':
            dt = dt.newbyteorder(order)
    return dt


def _payload_bytes(raw):
    """The byte string of an array / scalar payload: bytes, a zero-copy view, a protocol-2 text payload (latin-1), or
    the ``_codecs.encode(text, 'latin1')`` reduce that Python 3 writes for bytes at protocol <= 2."""
    if isinstance(raw, (bytes, memoryview, bytearray)):
        return raw
    if isinstance(raw, str):
        return raw.encode('latin-1')
    if isinstance(raw, Obj) and isinstance(raw.func, Global) and raw.func.is_(('_codecs',), 'encode') and \
            len(raw.args) == 2 and isinstance(raw.args[0], str) and _text(raw.args[1]).replace('-', '').lower() == 'latin1':
        return raw.args[0].encode('latin-1')
    raise PickleFormatError('array payload is not a byte string (object array?)')


def resolve(node):
    """Turn a numpy ``_reconstruct`` / ``scalar`` node into an ndarray / numpy scalar; other nodes come back as they
    are.  Arrays are built with ``np.frombuffer`` over the pickle's own bytes (read-only, zero-copy when large)."""
    if not isinstance(node, Obj) or not isinstance(node.func, Global):
        return node
    f = node.func
    if f.is_(_NP_MULTIARRAY, '_reconstruct'):
        st = node.state
        if not (isinstance(st, tuple) and len(st) == 5):
            raise PickleFormatError('ndarray without a (version, shape, dtype, fortran, data) state')
        _, shape, dt_node, fortran, raw = st
        dt = _dtype_of(dt_node)
        arr = np.frombuffer(_payload_bytes(raw), dtype=dt)
        shape = tuple(int(s) for s in shape)
        if int(np.prod(shape, dtype=np.int64)) != arr.size:
            raise PickleFormatError('ndarray payload size does not match its shape')
        return arr.reshape(shape, order='F' if fortran else 'C')
    if f.is_(_NP_MULTIARRAY, 'scalar'):
        dt = _dtype_of(node.args[0])
        return np.frombuffer(bytes(_payload_bytes(node.args[1])), dtype=dt)[0]
    if f.name == '_frombuffer' and f.module in ('numpy.core.numeric', 'numpy._core.numeric'):     # protocol 5, in-band
        raw, dt_node, shape, order = node.args
        arr = np.frombuffer(_payload_bytes(raw), dtype=_dtype_of(dt_node))
        return arr.reshape(tuple(int(v) for v in shape), order=_text(order))
    return node


def _state_dict(obj):
    st = obj.state
    if isinstance(st, tuple) and len(st) == 2 and isinstance(st[0], dict):       # (dict, slots) form
        st = st[0]
    if not isinstance(st, dict):
        raise PickleFormatError('object state is not a dict')
    return {_text(k) if isinstance(k, (bytes, memoryview)) else k: v for k, v in st.items()}


def _plain(v):
    """Scalars / small containers of the object's state as plain Python values."""
    v = resolve(v)
    if isinstance(v, np.generic):
        return v.item()
    if isinstance(v, (bytes, memoryview)):
        return _text(v)
    return v


# ---- the dataset ------------------------------------------------------------------------------------------------------
class IngestedDataset(object):
    """Plain arrays of one ``ARDSRawDataset``.

    windows (N, NB, C, L) float64 RAW flow (un-normalised), targets (N, 2) float32 one-hot, patients (N,) str ids (or
    None when loaded from an anonymised export), patient_slot (N,) int64 (first-appearance order), hours (N, NB)
    float64 (NaN padded), metadata (N, M) float64 or None, scaling_factors {fold|None: (mu (C,), std (C,))},
    kfold_patient_splits {fold: {'train': slots, 'test': slots}} or None, plus n_sub_batches / dataset_type / train /
    total_kfolds."""

    def __init__(self, windows, targets, patients, hours, scaling_factors, n_sub_batches, dataset_type, train=True,
                 total_kfolds=None, kfold_patient_splits=None, metadata=None, patient_slot=None):
        self.windows = np.ascontiguousarray(windows, dtype=np.float64)
        self.targets = np.ascontiguousarray(targets, dtype=np.float32)
        if self.windows.ndim != 4 or self.targets.shape != (self.windows.shape[0], 2):
            raise ValueError('windows must be (N, NB, C, L) and targets (N, 2)')
        self.patients = None if patients is None else np.asarray(patients).astype(str)
        if patient_slot is None and self.patients is not None:
            slots, seen = [], {}
            for p in self.patients.tolist():
                slots.append(seen.setdefault(p, len(seen)))
            patient_slot = np.array(slots, dtype=np.int64)
        # None: an older export without patient information -- usable for plain training / a holdout test over windows,
        # refused for anything patient-wise (k-folds, vote tables) instead of inventing groups
        self.patient_slot = None if patient_slot is None else np.asarray(patient_slot, dtype=np.int64)
        self.hours = hours
        self.metadata = metadata
        self.scaling_factors = scaling_factors
        self.n_sub_batches, self.dataset_type, self.train = int(n_sub_batches), dataset_type, bool(train)
        self.total_kfolds = None if total_kfolds is None else int(total_kfolds)
        self.kfold_patient_splits = kfold_patient_splits

    def __len__(self):
        return self.windows.shape[0]

    @property
    def n_patients(self):
        if self.patient_slot is None:
            return 0
        return int(self.patient_slot.max()) + 1 if len(self.patient_slot) else 0

    def save_npz(self, path, anonymise=True):
        """One .npz with everything a training run needs (loads with allow_pickle=False).  anonymise: patient
        identifiers are dropped, only their slots (order of first appearance) are written."""
        out = dict(x=self.windows, target=self.targets, hours=self.hours,
                   n_sub_batches=self.n_sub_batches, dataset_type=str(self.dataset_type), train=self.train,
                   total_kfolds=-1 if self.total_kfolds is None else self.total_kfolds)
        if self.patient_slot is not None:
            out['patient_slot'] = self.patient_slot
        for k, (mu, std) in self.scaling_factors.items():
            tag = 'none' if k is None else str(int(k))
            out['mu/' + tag], out['std/' + tag] = np.asarray(mu, dtype=np.float64), np.asarray(std, dtype=np.float64)
        if None in self.scaling_factors:                         # the fixture's names, kept for older readers
            out['mu'], out['std'] = (np.float64(np.ravel(v)[0]) for v in self.scaling_factors[None])
        if self.metadata is not None:
            out['metadata'] = self.metadata
        if self.kfold_patient_splits:
            for k, sp in self.kfold_patient_splits.items():
                out['split/%d/train' % k] = np.asarray(sp['train'], dtype=np.int64)
                out['split/%d/test' % k] = np.asarray(sp['test'], dtype=np.int64)
        if not anonymise and self.patients is not None:
            out['patients'] = self.patients.astype('U')
        np.savez_compressed(path, **out)
        return path

    def with_fft(self, add_fft=False, only_fft=False, fft_real_only=False):
        """What ``ARDSRawDataset.from_pickle`` does with --with-fft / --only-fft / --fft-real-only (dataset.py:743-762):
        a dataset that does not carry spectrum channels yet gets them (``_perform_fft``) and its scaling factors are
        dropped so that they are derived again per channel (``derive_scaling_factors``; ``to_store`` does that);
        one that already has them (C > 1), or no FFT option, is returned unchanged."""
        if not (add_fft or only_fft) or self.windows.shape[2] != 1:
            return self
        from .tiles import perform_fft
        self.windows = perform_fft(self.windows, add_fft, only_fft, fft_real_only)
        self.scaling_factors = {}
        return self

    def to_store(self, device='cuda', fold=None, random_kfold=False):
        """DeviceTileStore over these windows with the scaling factors of ``fold`` (None: the holdout factors, or
        derived from all windows when the pickle holds none).  K-fold datasets: ``enable_kfolds`` is applied with the
        pickled patient splits / factors when present (``from_pickle`` keeps them too, dataset.py:740-741)."""
        from .data import DeviceTileStore
        chans = self.windows.shape[2]
        key = fold if fold in self.scaling_factors else None
        if key in self.scaling_factors:
            mu, std = (np.ravel(v)[:chans] for v in self.scaling_factors[key])
            store = DeviceTileStore(self.windows, self.targets, mu, std, device=device)
        else:
            store = DeviceTileStore.with_derived_scaling(self.windows, self.targets, device=device)
        store.hours = self.hours
        store.patient_slot = self.patient_slot
        store.train = self.train
        if self.total_kfolds is not None:
            if self.patient_slot is None:
                raise ValueError('k-fold splits are patient-wise (dataset.py:765-830) and this dataset file carries no '
                                 'patient information (an export without patient_slot): re-export it with '
                                 'python -m deepards_amd.ingest <dataset.pkl> <out.npz>')
            factors = {k: (np.ravel(m)[:chans], np.ravel(s)[:chans]) for k, (m, s) in self.scaling_factors.items()
                       if k is not None} or None
            store.enable_kfolds(self.patient_slot, self.total_kfolds, train=self.train, random_kfold=random_kfold,
                                splits=self.kfold_patient_splits, scaling_factors=factors)
        return store


def _channel_scalars(v):
    """mu / std of the pickle: a scalar (old pickles) or the (NB, C, L) broadcast (dataset.py:641,648) -> (C,)."""
    v = resolve(v)
    a = np.asarray(v, dtype=np.float64)
    if a.ndim == 0:
        return a.reshape(1)
    if a.ndim == 3:
        return np.ascontiguousarray(a[0, :, 0])
    return a.reshape(-1)


def dataset_from_tree(root):
    """IngestedDataset from the inert tree of an ``ARDSRawDataset`` pickle (``parse_pickle`` output)."""
    if not (isinstance(root, Obj) and isinstance(root.func, Global) and root.func.name == 'ARDSRawDataset'):
        cls = root.args[0] if isinstance(root, Obj) and root.args and isinstance(root.args[0], Global) else None
        if not (cls is not None and cls.name == 'ARDSRawDataset'):          # copy_reg._reconstructor(cls, object, None)
            raise ValueError('The pickle file you have specified is out-of-date. Please re-process your dataset and '
                             'save the new pickled dataset.')                # the reference's message, dataset.py:727
    st = _state_dict(root)
    seqs = st.get('all_sequences')
    if not isinstance(seqs, list) or not seqs:
        raise ValueError('the dataset holds no sequences')
    windows, targets, patients, hours, metas = [], [], [], [], []
    for seq in seqs:
        seq = list(seq)
        if len(seq) == 4:
            pt, data, target, hrs = seq
            meta = None
        elif len(seq) == 5:
            pt, data, meta, target, hrs = seq
        else:
            raise NotImplementedError('sequences of %d items (dataset.py:1359-1361) are outside the hot path' % len(seq))
        windows.append(resolve(data))
        targets.append(np.asarray(resolve(target), dtype=np.float64))
        patients.append(_plain(pt))
        hours.append([float(_plain(h)) for h in (hrs if isinstance(hrs, (list, tuple)) else [hrs])])
        if meta is not None:
            metas.append(np.asarray(resolve(meta), dtype=np.float64).reshape(-1))
    x = np.stack(windows)
    nb = x.shape[1]
    hr = np.full((len(hours), nb), np.nan)
    for i, h in enumerate(hours):
        hr[i, :min(nb, len(h))] = h[:nb]
    factors = {}
    sf = st.get('scaling_factors')
    if isinstance(sf, dict):
        for k, v in sf.items():
            k = _plain(k)
            mu, std = v
            factors[None if k is None else int(k)] = (_channel_scalars(mu), _channel_scalars(std))
    slots, seen = [], {}
    for p in patients:
        slots.append(seen.setdefault(p, len(seen)))
    splits = None
    ks = st.get('kfold_patient_splits')
    if isinstance(ks, dict) and ks:
        splits = {}
        for k, sp in ks.items():
            splits[int(_plain(k))] = {
                part: np.array([seen[_plain(p)] for p in _iter_array(sp[part if part in sp else part.encode()])],
                               dtype=np.int64)
                for part in ('train', 'test')}
    total = _plain(st.get('total_kfolds'))
    return IngestedDataset(x, np.stack(targets), patients, hr, factors, _plain(st.get('n_sub_batches', nb)),
                           _plain(st.get('dataset_type', 'unpadded_centered_sequences')),
                           train=bool(_plain(st.get('train', True))), total_kfolds=total, kfold_patient_splits=splits,
                           metadata=np.stack(metas) if metas else None, patient_slot=np.array(slots, dtype=np.int64))


def _iter_array(v):
    """Items of a patient-id container: a str / bytes ndarray, a list, or an OBJECT-dtype ndarray (pandas'
    ``.unique()`` of strings, dataset.py:779-781), whose pickled payload is a plain list of str -- read as that list."""
    if isinstance(v, Obj) and isinstance(v.state, tuple) and len(v.state) == 5 and isinstance(v.state[4], list):
        return v.state[4]
    v = resolve(v)
    if isinstance(v, np.ndarray):
        return v.tolist()
    if isinstance(v, Obj):
        raise PickleFormatError('patient list in an unknown container')
    return list(v)


def read_ards_dataset(path):
    """Parse the pickle at ``path`` inertly (nothing in it is imported or called) -> IngestedDataset."""
    with open(path, 'rb') as f:
        data = f.read()
    return dataset_from_tree(parse_pickle(data))


def load_npz(path):
    """IngestedDataset from ``save_npz`` output (or from the older fixture export holding x / target / mu / std)."""
    z = np.load(path, allow_pickle=False)
    files = set(z.files)
    factors = {}
    for k in files:
        if k.startswith('mu/'):
            tag = k[3:]
            factors[None if tag == 'none' else int(tag)] = (z[k], z['std/' + tag])
    if not factors and 'mu' in files:
        factors[None] = (np.reshape(z['mu'], (1,)), np.reshape(z['std'], (1,)))
    n = z['x'].shape[0]
    splits = {}
    for k in files:
        if k.startswith('split/') and k.endswith('/train'):
            f = int(k.split('/')[1])
            splits[f] = {'train': z[k], 'test': z['split/%d/test' % f]}
    total = int(z['total_kfolds']) if 'total_kfolds' in files else -1
    return IngestedDataset(
        z['x'], z['target'], z['patients'] if 'patients' in files else None,
        z['hours'] if 'hours' in files else np.full((n, z['x'].shape[1]), np.nan), factors,
        int(z['n_sub_batches']) if 'n_sub_batches' in files else z['x'].shape[1],
        str(z['dataset_type']) if 'dataset_type' in files else 'unpadded_centered_sequences',
        train=bool(z['train']) if 'train' in files else True, total_kfolds=None if total < 0 else total,
        kfold_patient_splits=splits or None, metadata=z['metadata'] if 'metadata' in files else None,
        patient_slot=z['patient_slot'] if 'patient_slot' in files else None)


def load_dataset(path):
    """``--train-from-pickle`` / ``--test-from-pickle`` accept the reference's pickle (parsed inertly) or the .npz."""
    return load_npz(path) if str(path).endswith('.npz') else read_ards_dataset(path)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    keep = '--keep-patient-ids' in argv
    argv = [a for a in argv if not a.startswith('--')]
    if len(argv) != 2:
        raise SystemExit(__doc__.split('CLI')[1])
    ds = read_ards_dataset(argv[0])
    ds.save_npz(argv[1], anonymise=not keep)
    print('%d windows %s, %d patients, %d ARDS / %d other, factors %s -> %s' % (
        len(ds), ds.windows.shape[1:], ds.n_patients, int(ds.targets[:, 1].sum()), int(ds.targets[:, 0].sum()),
        {k: (float(m[0]), float(s[0])) for k, (m, s) in ds.scaling_factors.items()}, argv[1]))


if __name__ == '__main__':
    main()
