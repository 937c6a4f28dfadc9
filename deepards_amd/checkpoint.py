"""Checkpoints: what ``--save-model`` writes and what ``--load-checkpoint`` / ``--load-base-network`` accept.

The reference saves WHOLE MODULES -- ``torch.save(model, path)`` per fold and optionally per epoch
(train_ards_detector.py:355-374) -- and loads them back with ``torch.load`` (:383-388, :468-469), which unpickles:
the file names its classes (``deepards.models.resnet.ResNet`` ...) and ``torch.load`` imports and runs them.  A file
written by the reference therefore (a) cannot be unpickled here (those classes are not this package's) and (b) must
not be: foreign checkpoints are only ever read with loaders that execute nothing.  Three kinds of file are accepted:

* a ``state_dict`` (tensors in an OrderedDict)           -> ``torch.load(..., weights_only=True)``;
* a whole module pickled by the REFERENCE (or anybody)   -> ``read_module_checkpoint``: the pickle stream is walked
  by the inert parser of ``deepards_amd.ingest`` (nothing imported, nothing called), the tensors are rebuilt from the
  storages the stream points at (zip ``data/<key>`` entries, or the raw blobs that follow a pytorch<=1.5 "legacy"
  file -- the reference's environment pins pytorch 1.0), the module tree is flattened to a ``state_dict`` with
  ``nn.Module.state_dict``'s naming, ``nn.DataParallel`` wrappers are unwrapped like train_ards_detector.py:385-386;
* a whole module pickled by THIS package                 -> recognised by the class path ``deepards_amd.`` at the head of its
  pickle -- which anybody can write there, so these files are NOT unpickled freely either: ``load_own_module`` reads them
  with torch's restricted unpickler (``weights_only=True``) and an allowlist of exactly the ``nn.Module`` classes of
  ``deepards_amd.models`` and ``torch.nn``.  A file whose head says ``deepards_amd.models.X`` but whose body carries any
  other global (``os.system`` in a REDUCE, ...) is refused with ``pickle.UnpicklingError`` before anything runs.

``state_dict`` keys are the reference's, so the extracted weights load straight into this package's models.
"""
import io
import os
import struct
import zipfile
from collections import OrderedDict

import numpy as np
import torch

from .ingest import Global, Obj, PickleFormatError, parse_pickle, _text

LEGACY_MAGIC = 0x1950a86a20f9469cfc6c

_STORAGE_DTYPES = {
    'FloatStorage': np.float32, 'DoubleStorage': np.float64, 'HalfStorage': np.float16, 'LongStorage': np.int64,
    'IntStorage': np.int32, 'ShortStorage': np.int16, 'CharStorage': np.int8, 'ByteStorage': np.uint8,
    'BoolStorage': np.bool_,
}


class _Storages(object):
    """key -> raw little-endian bytes of a storage."""

    def __init__(self):
        self.blobs = {}

    def array(self, key, dtype, numel):
        raw = self.blobs[key]
        return np.frombuffer(raw, dtype=dtype, count=numel)


def _read_zip(path):
    with zipfile.ZipFile(path) as z:
        names = z.namelist()
        pkl = [n for n in names if n.endswith('/data.pkl') or n == 'data.pkl']
        if not pkl:
            raise PickleFormatError('no data.pkl in %s' % path)
        prefix = pkl[0][:-len('data.pkl')]
        order = [n for n in names if n == prefix + 'byteorder']
        if order and z.read(order[0]).strip() != b'little':
            raise PickleFormatError('big-endian checkpoints are not supported')
        tree = parse_pickle(z.read(pkl[0]), persistent=True)
        st = _Storages()
        for n in names:
            if n.startswith(prefix + 'data/'):
                st.blobs[n[len(prefix) + 5:]] = z.read(n)
    return tree, st


def _read_legacy(data):
    magic, pos = parse_pickle(data, 0, return_end=True)
    if magic != LEGACY_MAGIC:
        raise PickleFormatError('not a torch checkpoint (bad magic number)')
    _, pos = parse_pickle(data, pos, return_end=True)                 # protocol version
    _, pos = parse_pickle(data, pos, return_end=True)                 # system info
    tree, pos = parse_pickle(data, pos, persistent=True, return_end=True)
    keys, pos = parse_pickle(data, pos, return_end=True)
    st = _Storages()
    sizes = _storage_sizes(tree)
    for key in keys:
        key = _text(key)
        (numel,) = struct.unpack_from('<q', data, pos)
        pos += 8
        nbytes = numel * sizes[key]
        st.blobs[key] = bytes(data[pos:pos + nbytes])
        pos += nbytes
    return tree, st


def _walk(node, fn, seen=None):
    seen = set() if seen is None else seen
    if id(node) in seen:
        return
    seen.add(id(node))
    fn(node)
    if isinstance(node, Obj):
        for sub in (node.args, node.kwargs, node.state, node.items, [v for kv in node.pairs for v in kv]):
            _walk(sub, fn, seen)
    elif isinstance(node, dict):
        for v in node.values():
            _walk(v, fn, seen)
    elif isinstance(node, (list, tuple)):
        for v in node:
            _walk(v, fn, seen)


def _storage_ref(node):
    """('storage', type Global, key, location, numel[, view]) of a persistent-id node, or None."""
    if isinstance(node, Obj) and node.how == 'persid' and isinstance(node.args, tuple) and node.args and \
            _text(node.args[0]) == 'storage':
        return node.args
    return None


def _storage_dtype(g):
    if isinstance(g, Global) and g.name in _STORAGE_DTYPES:
        return _STORAGE_DTYPES[g.name]
    if isinstance(g, Global) and g.name == 'BFloat16Storage':
        return 'bf16'
    raise PickleFormatError('storage type %r is not supported' % (g,))


def _storage_sizes(tree):
    sizes = {}

    def visit(n):
        ref = _storage_ref(n)
        if ref is not None:
            dt = _storage_dtype(ref[1])
            sizes[_text(ref[2])] = 2 if dt == 'bf16' else np.dtype(dt).itemsize
    _walk(tree, visit)
    return sizes


def read_torch_file(path):
    """-> (inert tree, storages) of a torch.save file, zip or legacy format.  Nothing in the file is executed."""
    if zipfile.is_zipfile(path):
        return _read_zip(path)
    with open(path, 'rb') as f:
        return _read_legacy(f.read())


def _tensor(node, st):
    """torch.Tensor of a ``_rebuild_tensor_v2`` / ``_rebuild_parameter`` node."""
    if not (isinstance(node, Obj) and isinstance(node.func, Global)):
        raise PickleFormatError('not a tensor node: %r' % (node,))
    name = node.func.name
    if name in ('_rebuild_parameter', '_rebuild_parameter_with_state'):
        return _tensor(node.args[0], st)
    if name != '_rebuild_tensor_v2' or node.func.module != 'torch._utils':
        raise PickleFormatError('tensor rebuilt by %r is not supported' % (node.func,))
    ref = _storage_ref(node.args[0])
    if ref is None:
        raise PickleFormatError('tensor without a storage reference')
    offset, size, stride = int(node.args[1]), tuple(int(v) for v in node.args[2]), tuple(int(v) for v in node.args[3])
    dt = _storage_dtype(ref[1])
    numel = int(ref[4])
    if dt == 'bf16':
        flat = torch.from_numpy(st.array(_text(ref[2]), np.int16, numel).copy()).view(torch.bfloat16)
    else:
        flat = torch.from_numpy(st.array(_text(ref[2]), dt, numel).copy())
    return torch.as_strided(flat, size, stride, offset).clone() if size else flat[offset].clone()


def _odict_items(node):
    """(name, value) pairs of an OrderedDict / dict node."""
    if isinstance(node, dict):
        return list(node.items())
    if isinstance(node, Obj):
        pairs = list(node.pairs)
        if not pairs and node.args and isinstance(node.args[0], list):      # OrderedDict([[k, v], ...]) (protocol 2 py2)
            pairs = [tuple(kv) for kv in node.args[0]]
        return [(_text(k) if isinstance(k, (bytes, memoryview)) else k, v) for k, v in pairs]
    if node is None:
        return []
    raise PickleFormatError('not a dict node: %r' % (node,))


def _module_state(node):
    if not (isinstance(node, Obj) and isinstance(node.state, dict)):
        raise PickleFormatError('not a module node: %r' % (node,))
    return {(_text(k) if isinstance(k, (bytes, memoryview)) else k): v for k, v in node.state.items()}


def _class_of(node):
    f = node.func
    if isinstance(f, Obj) and f.how == 'persid' and isinstance(f.args, tuple) and _text(f.args[0]) == 'module':
        f = f.args[1]                                    # legacy files wrap container classes with their source text
    if isinstance(f, Global):
        return '%s.%s' % (f.module, f.name)
    return repr(f)


def _flatten(node, st, prefix, out):
    d = _module_state(node)
    skip = set()
    nps = d.get('_non_persistent_buffers_set')
    if isinstance(nps, Obj) and isinstance(nps.func, Global) and nps.func.name in ('set', 'frozenset'):
        nps = set(nps.args[0]) if nps.args else set()          # protocol 2 writes set([...]) as a reduce
    if isinstance(nps, (set, frozenset)):
        skip = {_text(v) if isinstance(v, (bytes, memoryview)) else v for v in nps}
    for name, v in _odict_items(d.get('_parameters')):
        if v is not None:
            out[prefix + name] = _tensor(v, st)
    for name, v in _odict_items(d.get('_buffers')):
        if v is not None and name not in skip:
            out[prefix + name] = _tensor(v, st)
    for name, v in _odict_items(d.get('_modules')):
        if v is not None:
            _flatten(v, st, prefix + name + '.', out)


def _submodule(node, name):
    for k, v in _odict_items(_module_state(node).get('_modules')):
        if k == name:
            return v
    return None


def read_module_checkpoint(path):
    """Whole-module checkpoint -> dict(state_dict, class_name, network_name, breath_block_class) without unpickling.
    ``nn.DataParallel`` is unwrapped (train_ards_detector.py:385-386)."""
    tree, st = read_torch_file(path)
    if not isinstance(tree, Obj) or not isinstance(tree.state, dict):
        raise PickleFormatError('%s does not hold a pickled nn.Module' % path)
    if _class_of(tree).endswith('.DataParallel'):
        tree = _submodule(tree, 'module')
    sd = OrderedDict()
    _flatten(tree, st, '', sd)
    info = dict(state_dict=sd, class_name=_class_of(tree), network_name=None, breath_block_class=None)
    bb = _submodule(tree, 'breath_block')
    if bb is not None:
        info['breath_block_class'] = _class_of(bb)
        nn_ = _module_state(bb).get('network_name')
        info['network_name'] = None if nn_ is None else _text(nn_) if isinstance(nn_, (bytes, memoryview)) else nn_
    return info


def checkpoint_kind(path):
    """'own' (a module this package pickled), 'foreign' (any other pickled module) or 'state_dict'."""
    tree, _ = read_torch_file(path)
    if isinstance(tree, Obj) and isinstance(tree.state, dict) and ('_modules' in tree.state or b'_modules' in tree.state):
        return 'own' if _class_of(tree).startswith('deepards_amd.') else 'foreign'
    return 'state_dict'


def _own_module_classes():
    """The only classes a whole-module file of this package may name: the nn.Module subclasses defined in
    deepards_amd.models.* and torch.nn's own module classes (containers, Conv1d / BatchNorm1d / Linear / LSTM ... that
    the models hold as parameter containers)."""
    import inspect
    from .models import densenet, resnet, torch_cnn_linear_network
    allow = []
    for mod in (resnet, densenet, torch_cnn_linear_network):
        allow += [c for c in vars(mod).values()
                  if inspect.isclass(c) and issubclass(c, torch.nn.Module) and c.__module__ == mod.__name__]
    allow += [c for c in vars(torch.nn).values() if inspect.isclass(c) and issubclass(c, torch.nn.Module)]
    return allow


def load_own_module(path):
    """A whole module this package saved (``torch.save(model, path)``), through torch's RESTRICTED unpickler: only the
    classes of ``_own_module_classes`` (constructed with ``__new__`` + ``__setstate__``, never called), tensors,
    parameters and plain containers are accepted; any other global in the stream raises ``pickle.UnpicklingError``
    before it is looked up.  The class path at the head of the file is therefore not trusted for anything."""
    with torch.serialization.safe_globals(_own_module_classes()):
        model = torch.load(path, weights_only=True, map_location='cpu')
    for q in model.parameters():                 # (files written before round 4 carry the saving trainer's gradient views)
        q.__dict__.pop('_da_grad', None)
    return model


def load_model_weights(path, build_model):
    """Model for ``--load-checkpoint`` (train_ards_detector.py:468-469).  Own whole-module files go through
    ``load_own_module`` (restricted unpickler + class allowlist); for everything else ``build_model()`` makes a fresh
    model of the configured architecture and the file's weights are loaded into it (strict: the keys are the
    reference's)."""
    kind = checkpoint_kind(path)
    if kind == 'own':
        return load_own_module(path)
    sd = torch.load(path, weights_only=True, map_location='cpu') if kind == 'state_dict' else \
        read_module_checkpoint(path)['state_dict']
    if any(k.startswith('module.') for k in sd):                          # a DataParallel state_dict
        sd = OrderedDict((k[7:] if k.startswith('module.') else k, v) for k, v in sd.items())
    model = build_model()
    model.load_state_dict(sd, strict=True)
    return model


def load_base_network(path, base_networks, build_args=None):
    """``--load-base-network`` (train_ards_detector.py:383-388): the ``breath_block`` of a saved model.  Foreign
    files: a fresh base network of the architecture the file names (``breath_block.network_name``) receives the
    ``breath_block.*`` weights."""
    kind = checkpoint_kind(path)
    if kind == 'own':
        saved = load_own_module(path)
        if isinstance(saved, torch.nn.DataParallel):
            saved = saved.module
        return saved.breath_block
    if kind == 'state_dict':
        sd, name = torch.load(path, weights_only=True, map_location='cpu'), None
    else:
        info = read_module_checkpoint(path)
        sd, name = info['state_dict'], info['network_name']
    sd = OrderedDict((k[7:] if k.startswith('module.') else k, v) for k, v in sd.items())
    bb = OrderedDict((k[len('breath_block.'):], v) for k, v in sd.items() if k.startswith('breath_block.'))
    if not bb:
        raise ValueError('%s holds no breath_block.* weights' % path)
    build_args = build_args or {}
    name = name or build_args.get('base_network')
    if name not in base_networks:
        raise ValueError('base network %r of %s is not built by this package' % (name, path))
    kwargs = build_args.get('resnet_kwargs', {}) if name.startswith('resnet') else build_args.get('densenet_kwargs', {})
    net = base_networks[name](**kwargs)
    net.load_state_dict(bb, strict=True)
    return net


def model_save_path(save_model, saved_models_dir, n_kfolds, fold_num, epoch_num=None):
    """File names of train_ards_detector.py:355-374: ``<dir>/<stem>[-epoch{E}][-fold{K}].pth``."""
    stem = os.path.splitext(save_model)[0]
    if epoch_num is not None:
        stem += '-epoch{}'.format(epoch_num)
    tail = '-fold{}.pth'.format(fold_num) if n_kfolds > 1 else '.pth'
    return os.path.join(saved_models_dir, os.path.basename(stem + tail))
