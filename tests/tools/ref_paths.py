"""Test helper: pickle this package's models under the REFERENCE's class paths."""
import sys
import types


def as_reference_classes(fn):
    """Run fn() while this package's model classes claim the REFERENCE's class paths (deepards.models.*), so that
    torch.save writes what a file saved by the reference names -- without the reference being importable."""
    import deepards_amd.models as M
    import deepards_amd.models.resnet as R
    import deepards_amd.models.densenet as D
    import deepards_amd.models.torch_cnn_linear_network as N
    moved = []
    fakes = {}
    for mod, refname in ((R, 'deepards.models.resnet'), (D, 'deepards.models.densenet'),
                         (N, 'deepards.models.torch_cnn_linear_network')):
        fake = fakes.setdefault(refname, types.ModuleType(refname))
        for name, obj in vars(mod).items():
            if isinstance(obj, type) and obj.__module__ == mod.__name__:
                moved.append((obj, obj.__module__))
                obj.__module__ = refname
                setattr(fake, name, obj)
    fakes['deepards'] = types.ModuleType('deepards')
    fakes['deepards.models'] = types.ModuleType('deepards.models')
    saved = {k: sys.modules.get(k) for k in fakes}
    sys.modules.update(fakes)
    try:
        return fn(M)
    finally:
        for obj, m in moved:
            obj.__module__ = m
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
