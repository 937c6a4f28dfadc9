"""Build + load libdeepards_hip.so (the C-ABI of include/deepards_hip.h) with ctypes.

There is NO fallback: if the library is missing or a symbol cannot be bound the import of any
compute entry point raises.  PyTorch is used only for device memory and streams.
"""
import ctypes
import os
import re
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
LIB_PATH = os.environ.get('DA_LIB_PATH') or os.path.join(_HERE, 'libdeepards_hip.so')   # (override: A/B builds, scripts/)
HEADER = os.path.join(os.path.dirname(_HERE), 'include', 'deepards_hip.h')
SOURCES = ['conv_gemm.hip', 'conv_wino.hip', 'conv_bf16.hip', 'conv_x3p.hip', 'bn.hip', 'stem_pool.hip', 'head_optim.hip']

_P, _I, _F, _Z, _U = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t, ctypes.c_uint
_IP = ctypes.POINTER(ctypes.c_int)


class BnRunningDesc(ctypes.Structure):
    _fields_ = [('mean', _P), ('invstd', _P), ('running_mean', _P), ('running_var', _P), ('num_batches_tracked', _P),
                ('W', _I), ('C', _I), ('Wn', _I), ('eps', _F), ('momentum', _F)]


class BnFwdDesc(ctypes.Structure):
    _fields_ = [('x', ctypes.c_void_p), ('ldx', ctypes.c_int), ('res', ctypes.c_void_p), ('ldr', ctypes.c_int),
                ('out', ctypes.c_void_p), ('ldo', ctypes.c_int), ('mean', ctypes.c_void_p), ('invstd', ctypes.c_void_p),
                ('gamma', ctypes.c_void_p), ('beta', ctypes.c_void_p), ('relu', ctypes.c_int), ('mask', ctypes.c_void_p)]


class BnBwdDesc(ctypes.Structure):
    _fields_ = [('x', ctypes.c_void_p), ('ldx', ctypes.c_int), ('dx', ctypes.c_void_p), ('lddx', ctypes.c_int),
                ('mean', ctypes.c_void_p), ('invstd', ctypes.c_void_p), ('gamma', ctypes.c_void_p), ('beta', ctypes.c_void_p),
                ('ds', ctypes.c_void_p)]


class BnPgradDesc(ctypes.Structure):
    _fields_ = [('s1', _P), ('s2', _P), ('dgamma', _P), ('dbeta', _P), ('W', _I), ('C', _I)]


class WgradReduceDesc(ctypes.Structure):
    _fields_ = [('slab', _P), ('dw', _P), ('splits', _I), ('ntaps', _I), ('N', _I), ('C', _I)]


class WgradJob(ctypes.Structure):
    _fields_ = [('dy', _P), ('x', _P), ('workspace', _P)] + [(n, _I) for n in
               ('rows', 'Lm', 'Ldy', 'lddy', 'N', 'Lx', 'ldx', 'C', 'dy_stride', 'dy_off', 'src_stride', 'ntaps')] + \
               [('src_off', _I * 3), ('winograd', _I), ('xform', _I), ('dy_half', _I), ('Wn', _I), ('ldstat', _I),
                ('mean', _P), ('invstd', _P), ('gamma', _P), ('beta', _P)]


class ConvJob(ctypes.Structure):
    _fields_ = [('x', _P), ('w', _P), ('y', _P)] + [(n, _I) for n in
               ('rows', 'Lm', 'Lsrc', 'ldx', 'C', 'Ldst', 'ldy', 'N', 'dst_stride', 'dst_off', 'src_stride', 'ntaps')] + \
               [('src_off', _I * 3), ('wtap', _I * 3), ('accumulate', _I), ('x2', _P), ('w2', _P), ('tap_split', _I)]


class RepackDesc(ctypes.Structure):
    _fields_ = [('W', _P), ('Wf', _P), ('Wd', _P), ('Uf', _P), ('Ud', _P), ('Co', _I), ('Ci', _I), ('K', _I),
                ('points', _I)]


# name -> (restype, argtypes); must list exactly the symbols the header declares (tests check it)
SIGNATURES = {
    'da_version': (_I, []),
    'da_set_act_dtype': (_I, [_I]),
    'da_get_act_dtype': (_I, []),
    'da_abi_sizes': (None, [ctypes.POINTER(_I)]),
    'da_sizeof_wgrad_reduce_desc': (_I, []),
    'da_sizeof_bn_running_desc': (_I, []),
    'da_sizeof_bn_pgrad_desc': (_I, []),
    'da_hip_runtime_symbol': (_P, []),
    'da_conv_gemm': (_I, [_P, _P, _P] + [_I] * 12 + [_IP, _IP, _I, _P]),
    'da_conv_wgrad_workspace': (_Z, [_I] * 5),
    'da_conv_wgrad': (_I, [_P, _P, _P, _P] + [_I] * 12 + [_IP, _I, _P]),
    'da_debug_set': (_I, [_I, _I]),
    'da_repack_conv_weight': (_I, [_P, _P, _P, _I, _I, _I, _P]),
    'da_stem_conv_fwd': (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    'da_stem_wgrad_workspace': (_Z, [_I, _I]),
    'da_stem_conv_wgrad': (_I, [_P, _I, _P, _P, _P, _I, _I, _I, _I, _P]),
    'da_stem_conv_fwd_g': (_I, [_P, _P, _P] + [_I] * 7 + [_P]),
    'da_stem_wgrad_workspace_g': (_Z, [_I] * 4),
    'da_stem_conv_wgrad_g': (_I, [_P, _I, _P, _P, _P] + [_I] * 7 + [_P]),
    'da_bn_chunks': (None, [_I, _I, _I, _IP, _IP]),
    'da_bn_workspace': (_Z, [_I, _I, _I]),
    'da_bn_stats_partial': (_I, [_P, _I, _I, _I, _I, _P, _P]),
    'da_stem_stats_partial': (_I, [_P, _P, _I, _I, _I, _I, _P, _P]),
    'da_stem_bn_relu_pool_fwd': (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _I, _P]),
    'da_stem_bwd_workspace': (_Z, [_I, _I]),
    'da_stem_bwd': (_I, [_P, _I, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P, _I, _P, _P]),
    'da_bn_stats_merge': (_I, [_P, _I, _I, _I, _F, _P, _P, _P]),
    'da_bn_running_multi': (_I, [ctypes.POINTER(BnRunningDesc), _I, _P]),
    'da_bn_apply': (_I, [_P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _F, _P]),
    'da_bn_fwd_x': (_I, [_P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _F, _P, _I, _I, _P]),
    'da_bn_bwd_x': (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P, _I, _P]),
    'da_bn_relu_pool_fwd_x': (_I, [_P, _I, _P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P]),
    'da_bn_fwd': (_I, [_P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _F, _P, _P]),
    'da_bn_debug_two_stage': (_I, [_I]),
    'da_bn_debug_target_blocks': (_I, [_I]),
    'da_bn_bwd': (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _I, _P]),
    'da_bn_mask_words': (_Z, [_I, _I, _I]),
    'da_bn_fwd_mask': (_I, [_P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _P, _P, _P, _F, _P, _P, _P]),
    'da_bn_bwd_mask': (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P]),
    'da_bn_pool_ok': (_I, [_I, _I, _I, _I]),
    'da_bn_fwd_pool': (_I, [_P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _P, _P, _P, _F, _P, _P]),
    'da_bn_bwd_pool': (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    'da_bn_bwd_add': (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _I, _P, _I, _P]),
    'da_bn_param_grad_multi': (_I, [ctypes.POINTER(BnPgradDesc), _I, _I, _P]),
    'da_bn_fwd_pair': (_I, [ctypes.POINTER(BnFwdDesc), _I, _I, _I, _F, _P]),
    'da_bn_bwd_pair': (_I, [_P, _I, ctypes.POINTER(BnBwdDesc), _I, _I, _I, _P, _P]),
    'da_bn_bwd_pair2': (_I, [_P, _I, _P, _I, ctypes.POINTER(BnBwdDesc), _I, _I, _I, _P, _P]),
    'da_bn_two_ok': (_I, [_I, _I, _I]),
    'da_bn_bwd_mask2': (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    'da_bn_stats_fused': (_I, [_P, _I, _I, _I, _I, _P, _P, _I, _F, _P]),
    'da_bn_relu_ss': (_I, [_P, _I, _P, _I, _I, _I, _I, _P, _P, _I, _P, _P, _P]),
    'da_bn_bwd_ss': (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _P, _I, _P, _P, _I, _I, _P, _U, _F, _I, _P, _P, _I, _P]),
    'da_conv1x1_bn': (_I, [_P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _I, _P, _P, _P, _I, ctypes.c_long, _I, _F, _P, _P]),
    'da_conv3_winograd_drop': (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _U, _F, _P, _I, _P]),
    'da_stat_records_floats': (_Z, [ctypes.c_long, _I]),
    'da_conv3_winograd_bn': (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _F, _P, _U, _F, _P, _P]),
    'da_conv_wgrad_splits': (_I, [_I] * 5),
    'da_conv_wgrad_plan': (_I, [_I] * 6 + [ctypes.POINTER(_I)]),
    'da_conv_gemm_multi': (_I, [ctypes.POINTER(ConvJob), _I, _P]),
    'da_conv3_winograd': (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    'da_conv3_winograd4': (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    'da_conv3_bf16': (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    'da_conv3_bf16_bn': (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _F, _P, _P]),
    'da_pack_conv3_bf16': (_I, [_P, _P, _P, _I, _I, _P]),
    'da_conv3_x3p': (_I, [_P, _P, _P] + [_I] * 6 + [_P]),
    'da_conv_x3p_s2_fwd': (_I, [_P] * 5 + [_I] * 4 + [_P]),
    'da_conv_x3p_s2_dgrad': (_I, [_P] * 5 + [_I] * 4 + [_P]),
    'da_x3_split': (_I, [_P, _I, _P, _Z, _I, _P]),
    'da_x3_merge': (_I, [_P, _P, _I, _Z, _I, _P]),
    'da_conv_bf16_multi': (_I, [ctypes.POINTER(ConvJob), _I, _P]),
    'da_wino_debug_tail': (_I, [_I]),
    'da_wino_debug_pchunk': (_I, [_I]),
    'da_wino_debug_tapmod': (_I, [_I]),
    'da_wino_weights': (_I, [_P, _P, _I, _I, _I, _P]),
    'da_wino4_weights': (_I, [_P, _P, _I, _I, _I, _P]),
    'da_conv_wgrad_multi': (_I, [ctypes.POINTER(WgradJob), _I, _P]),
    'da_conv_wgrad_multi_reduce': (_I, [ctypes.POINTER(WgradJob), _I, ctypes.POINTER(ctypes.c_void_p), _I, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), _P]),
    'da_wgrad_reduce_multi': (_I, [ctypes.POINTER(WgradReduceDesc), _I, _I, _P]),
    'da_step_tail_multi': (_I, [ctypes.POINTER(WgradReduceDesc), _I, ctypes.POINTER(BnPgradDesc), _I, ctypes.POINTER(BnRunningDesc), _I, _P, _I, _I, _P, _I, _P]),
    'da_stem_bwd_partials': (_I, [_I, _I, _I, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(_I)]),
    'da_stem_wgrad_reduce': (_I, [_P, _I, _I, _P, _I, _P]),
    'da_repack_multi': (_I, [ctypes.POINTER(RepackDesc), _I, _P]),
    'da_bn_relu_pool_fwd': (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P]),
    'da_pool_bwd': (_I, [_P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P]),
    'da_avgpool_fwd': (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _P]),
    'da_avgpool_bwd': (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _P]),
    'da_avgpool_slide_fwd': (_I, [_P, _I, _P, _I, _I, _I, _I, _P]),
    'da_avgpool_slide_bwd': (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    'da_global_avgpool_fwd': (_I, [_P, _I, _P, _I, _I, _I, _P]),
    'da_global_avgpool_bwd': (_I, [_P, _P, _I, _I, _I, _I, _P]),
    'da_linear2_fwd': (_I, [_P, _P, _P, _P, _I, _I, _P]),
    'da_bce_logits': (_I, [_P, _P, _I, _F, _P, _P, _P]),
    'da_linear2_bwd': (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    'da_head_groups': (_I, [_I, _I]),
    'da_head_fwd': (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    'da_head_bwd': (_I, [_P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _P]),
    'da_head_flat_fwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    'da_head_flat_bwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _I, _P]),
    'da_clamp_sgd_nesterov': (_I, [_P, _P, _P, _Z, _F, _F, _F, _F, _F, _I, _P]),
    'da_clamp_adam': (_I, [_P, _P, _P, _P, _Z, _F, _F, _F, _F, _I, _F, _F, _P]),
    'da_clamp_adam_dev': (_I, [_P, _P, _P, _P, _Z, _F, _F, _F, _F, _P, _F, _F, _P]),
    'da_gather_normalize': (_I, [_P, _P, ctypes.c_double, ctypes.c_double, _P, _I, _I, _P]),
    'da_gather_normalize_ch': (_I, [_P, _P, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), _P, _I, _I, _I, _I, _P]),
    'da_window_median_fwd': (_I, [_P, _I, _I, _I, _I, _P, _P, _P]),
    'da_window_median_bwd': (_I, [_P, _P, _I, _I, _I, _P, _I, _P]),
    'da_lstm_fwd': (_I, [_P] * 11 + [_I, _I, _I, _P]),
    'da_lstm_bwd': (_I, [_P] * 9 + [_I, _I, _I, _P]),
    'da_reduce_rows': (_I, [_P, _I, _I, _P, _I, _P]),
    'da_gather_rows': (_I, [_P, _P, _P, _I, _I, _P]),
    'da_vote_counts': (_I, [_P, _P, _I, _I, _P, _P, _P]),
    'da_concat2': (_I, [_P, _I, _I, _P, _I, _I, _P, _I, _Z, _P]),
    'da_slice_copy': (_I, [_P, _I, _I, _P, _I, _I, _Z, _I, _P]),
    'da_dropout': (_I, [_P, _P, _Z, _P, _U, _F, _P]),
    'da_concat2_dropout': (_I, [_P, _I, _I, _P, _I, _I, _P, _I, _Z, _P, _U, _F, _P]),
    'da_slice_dropout': (_I, [_P, _I, _I, _P, _I, _I, _Z, _P, _U, _F, _P]),
}


def header_symbols():
    """Function names declared in include/deepards_hip.h."""
    txt = open(HEADER).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(da_[a-z0-9_]+)\s*\(', txt)))


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 every csrc/*.hip into deepards_amd/libdeepards_hip.so (in-tree)."""
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(CSRC, 'common.h')]
    if not force and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(d) for d in deps):
        return LIB_PATH
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    if not os.path.exists(hipcc):
        hipcc = 'hipcc'
    cmd = [hipcc, '-O3', '--offload-arch=gfx950', '-fPIC', '-shared', '-std=c++17', '-o', LIB_PATH] + srcs
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


_lib = None


def lib():
    """The loaded library with argtypes bound.  Raises (never falls back) if it cannot be loaded."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError('libdeepards_hip.so is not built: run `python -c "import __graft_entry__ as g; '
                               'g.build()"` (hipcc --offload-arch=gfx950).  There is no CPU fallback.')
        import torch  # noqa: F401  (first: its bundled HIP runtime must be the one we bind to)
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)          # AttributeError if the symbol is missing: fail loudly
            fn.restype = res
            fn.argtypes = args
        _check_same_runtime(l)
        _lib = l
    return _lib


def _check_same_runtime(l):
    """PyTorch wheels bundle their own libamdhip64.so; this library must bind to the SAME runtime
    (streams, graph capture and ordering are per runtime).  torch is imported first so that its HIP
    runtime sits in the global symbol scope; verify, and refuse to run on a mismatch."""
    import sys
    torch = sys.modules.get('torch')
    if torch is None:
        return
    tlib = os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so')
    if not os.path.exists(tlib):
        return
    theirs = ctypes.cast(ctypes.CDLL(tlib).hipGetLastError, ctypes.c_void_p).value
    ours = l.da_hip_runtime_symbol()
    if ours != theirs:
        raise RuntimeError('libdeepards_hip.so is bound to a different HIP runtime (%#x) than PyTorch (%#x): '
                           'import torch before deepards_amd' % (ours or 0, theirs or 0))
