"""Loader for ``libmsmc_hip.so`` -- the gfx950 kernels behind the C ABI in ``include/msmc_hip.h``.

The library is built in-tree (``msmc-tts_amd/lib/libmsmc_hip.so``) by ``__graft_entry__.build()``.
There is deliberately no fallback: a missing library, a failed launch or a host tensor raises.
``use_library_for_tests`` exists only so that the CPU test-suite can point the *same* Python ops at
the kernel interpreter build (tests/emu); the product never calls it.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_PATH = os.path.normpath(os.path.join(_HERE, '..', '..', 'lib', 'libmsmc_hip.so'))
# A/B HOOK (tools): another build of the same library, e.g. the previous commit's, for a same-box comparison
DEFAULT_PATH = os.environ.get('MSMC_HIP_LIB', DEFAULT_PATH)

_lib = None
_host_pointers_ok = False

_vp, _i, _f, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t

_SIGNATURES = {
    'msmc_backend': (ctypes.c_char_p, []),
    'msmc_abi_version': (_i, []),
    'msmc_stream_create': (_i, [ctypes.POINTER(ctypes.c_void_p)]),
    'msmc_stream_destroy': (_i, [_vp]),
    'msmc_vq_prepare': (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    'msmc_vq_search': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'msmc_vq_shortlist_bytes': (_sz, [_i, _i, _i]),
    'msmc_vq_prepare_shortlist': (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    'msmc_vq_search_shortlist': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'msmc_vq_ema_workspace': (_sz, [_i, _i, _i, _i]),
    'msmc_vq_ema_update': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _f, _f, _vp]),
    'msmc_vq_ema_stats': (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _vp]),
    'msmc_vq_ema_apply': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _f, _vp]),
    'msmc_vq_backward': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
}


MAX_TAPS = 16


class ConvDesc(ctypes.Structure):
    """msmc_conv_desc of include/msmc_hip.h."""
    _fields_ = [('x', _vp), ('w', _vp), ('bias', _vp), ('mask_src', _vp), ('res', _vp), ('res2', _vp), ('out', _vp),
                ('dtype', _i), ('B', _i), ('Hin', _i), ('Win', _i), ('Cin', _i), ('Hout', _i), ('Wout', _i),
                ('Cout', _i), ('QH', _i), ('QW', _i), ('oy0', _i), ('osy', _i), ('ox0', _i), ('osx', _i),
                ('isy', _i), ('isx', _i), ('iy0', _i), ('ix0', _i), ('ntaps', _i),
                ('tap_dy', _i * MAX_TAPS), ('tap_dx', _i * MAX_TAPS), ('tap_w', _i * MAX_TAPS),
                ('pad_mode', _i), ('in_slope', _f), ('mask_slope', _f), ('out_div', _f), ('out_slope', _f),
                ('variant', _i), ('split_shift', _i), ('dw_copies', _i)]


class WgPending(ctypes.Structure):
    """msmc_wg_pending of include/msmc_hip.h."""
    _fields_ = [('ws', _vp), ('mid', _vp), ('dw', _vp), ('db', _vp), ('stride', ctypes.c_long), ('n_dw', ctypes.c_long),
                ('n_db', _i), ('nsplit', _i)]


class WnItem(ctypes.Structure):
    """msmc_wn_item of include/msmc_hip.h."""
    _fields_ = [('v', _vp), ('g', _vp), ('dst1', _vp), ('dst2', _vp), ('inv_norm', _vp), ('dw', _vp), ('gv', _vp),
                ('gg', _vp), ('s1', ctypes.c_long * 3), ('s2', ctypes.c_long * 3), ('A', _i), ('Bc', _i), ('T', _i),
                ('dtype', _i), ('block0', _i), ('nbias', _i), ('db', _vp), ('gb', _vp), ('copies', _i), ('tblock0', _i),
                ('dw_copy_stride', ctypes.c_long), ('db_copy_stride', ctypes.c_long)]


class SpectralOp(ctypes.Structure):
    """msmc_spectral_op of include/msmc_hip.h."""
    _fields_ = [('kind', _i), ('dtype', _i), ('a', _vp), ('b', _vp), ('c', _vp), ('out', _vp),
                ('B', _i), ('L', _i), ('T', _i), ('n_fft', _i), ('NP', _i), ('hop', _i), ('pad', _i),
                ('F', _i), ('CP', _i), ('FP', _i), ('clamp_mode', _i), ('lo', _f), ('R', ctypes.c_long)]


SPECTRAL_MULTI_MAX = 8


class OptTensor(ctypes.Structure):
    """msmc_opt_tensor of include/msmc_hip.h."""
    _fields_ = [('p', _vp), ('g', _vp), ('m', _vp), ('v', _vp), ('n', ctypes.c_long), ('first_chunk', _i), ('pad_', _i)]


MAX_TENSORS = 64


class TensorTable(ctypes.Structure):
    """msmc_tensor_table of include/msmc_hip.h."""
    _fields_ = [('a', _vp * MAX_TENSORS), ('b', _vp * MAX_TENSORS), ('ga', _vp * MAX_TENSORS),
                ('n', ctypes.c_long * MAX_TENSORS), ('count', _i), ('dtype', _i)]


_SIGNATURES.update({
    'msmc_spectral_multi': (_i, [ctypes.POINTER(SpectralOp), _i, _vp]),
    'msmc_window_gather': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, ctypes.c_long, _vp]),
    'msmc_stft_frames_fwd': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'msmc_stft_frames_bwd': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'msmc_spec_mag_fwd': (_i, [_vp, _vp, ctypes.c_long, _i, _i, _i, _f, _i, _vp]),
    'msmc_spec_mag_bwd': (_i, [_vp, _vp, _vp, _vp, ctypes.c_long, _i, _i, _i, _f, _i, _vp]),
    'msmc_mrd_image_fwd': (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    'msmc_mrd_image_bwd': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'msmc_mrd_image_fwd_dt': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'msmc_mrd_image_bwd_dt': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'msmc_wave_fan_fwd': (_i, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_i), _i, _i, _i, _i, _vp]),
    'msmc_wave_fan_bwd': (_i, [ctypes.POINTER(_vp), _i, ctypes.POINTER(_vp), ctypes.POINTER(_i), _i, _vp, _i, _i, _i, _vp]),
    'msmc_log_clamp_fwd': (_i, [_vp, _vp, ctypes.c_long, _f, _vp]),
    'msmc_log_clamp_bwd': (_i, [_vp, _vp, _vp, ctypes.c_long, _f, _vp]),
    'msmc_l1_multi_fwd': (_i, [ctypes.POINTER(TensorTable), _vp, _vp]),
    'msmc_scalar_wsum_fwd': (_i, [ctypes.POINTER(_vp), ctypes.POINTER(_f), _i, _vp, _vp]),
    'msmc_scalar_wsum_bwd': (_i, [_vp, ctypes.POINTER(_f), _i, _vp, _vp]),
    'msmc_loss_multi_parts': (_i, []),
    'msmc_l1_multi_fwd_ws': (_i, [ctypes.POINTER(TensorTable), _vp, _vp, _vp]),
    'msmc_mse_const_multi_fwd_ws': (_i, [ctypes.POINTER(TensorTable), _f, _vp, _vp, _vp]),
    'msmc_l1_multi_bwd': (_i, [ctypes.POINTER(TensorTable), _vp, _vp]),
    'msmc_mse_const_multi_fwd': (_i, [ctypes.POINTER(TensorTable), _f, _vp, _vp]),
    'msmc_mse_const_multi_bwd': (_i, [ctypes.POINTER(TensorTable), _f, _vp, _vp]),
    'msmc_triple_loss': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp]),
    'msmc_masked_mean_parts': (_i, [_i]),
    'msmc_masked_mean_fwd': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    'msmc_masked_mean_bwd': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    'msmc_conv_gather': (_i, [ctypes.POINTER(ConvDesc), _vp]),
    'msmc_conv_gather_group': (_i, [ctypes.POINTER(ConvDesc), _i, _vp]),
    'msmc_conv_wgrad': (_i, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp]),
    'msmc_conv_wgrad_group': (_i, [ctypes.POINTER(ConvDesc), ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp),
                              _i, _vp]),
    'msmc_conv_wgrad_workspace': (_sz, [ctypes.POINTER(ConvDesc), _vp]),
    'msmc_conv_wgrad_ws': (_i, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _sz, _vp]),
    'msmc_conv_wgrad_group_ws': (_i, [ctypes.POINTER(ConvDesc), ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp),
                                 _i, _vp, _sz, _vp]),
    'msmc_conv_wgrad_group_ws4': (_i, [ctypes.POINTER(ConvDesc), ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp),
                                  _i, _vp, _sz, _vp, _i]),
    'msmc_conv_wgrad_defer_begin': (None, [ctypes.POINTER(WgPending), _i]),
    'msmc_conv_wgrad_defer_end': (_i, []),
    'msmc_conv_wgrad_reduce_pending': (_i, [ctypes.POINTER(WgPending), _i, _vp]),
    'msmc_attn_fwd': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _vp, ctypes.c_longlong, _vp]),
    'msmc_attn_bwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _vp, ctypes.c_longlong, _vp]),
    'msmc_wn_prepare_multi': (_i, [_vp, _i, _i, _vp]),
    'msmc_wn_prepare_multi_tiled': (_i, [_vp, _i, _i, _i, _vp]),
    'msmc_wn_tile_blocks': (_i, [_i, _i, _i]),
    'msmc_wn_prepare_multi_tiles': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp]),
    'msmc_wn_backward_multi': (_i, [_vp, _i, _i, _vp]),
    'msmc_wn_backward_multi_acc': (_i, [_vp, _i, _i, _i, _vp]),
    'msmc_wn_backward_multi_rows': (_i, [_vp, _i, _i, _i, _i, _vp]),
    'msmc_reflect_fold_multi': (_i, [ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_i), ctypes.POINTER(_i), ctypes.POINTER(_i),
                                ctypes.POINTER(_i), _i, _i, _f, _i, _vp]),
    'msmc_reflect_fold_multi_res': (_i, [ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_i),
                                    ctypes.POINTER(_i), ctypes.POINTER(_i), ctypes.POINTER(_i), _i, _i, _f, _i, _vp]),
    'msmc_reflect_fold_multi_tap': (_i, [ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_i),
                                    ctypes.POINTER(_i), ctypes.POINTER(_i), ctypes.POINTER(_i), _i, _i, _f, _i, _vp]),
    'msmc_lrelu_bwd_multi': (_i, [ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(ctypes.c_long), _i, _f, _i, _vp]),
    'msmc_colsum': (_i, [_vp, _vp, ctypes.c_long, _i, _i, _vp]),
    'msmc_colsum_workspace': (_sz, [ctypes.c_long, _i]),
    'msmc_colsum_ws': (_i, [_vp, _vp, ctypes.c_long, _i, _i, _i, _vp, _sz, _vp]),
    'msmc_add_ln_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_long, _i, _f, _f, _vp, ctypes.c_longlong,
                        _i, _vp]),
    'msmc_fc_add_ln_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_long, _i, _i, _f, _f, _vp,
                           ctypes.c_longlong, _vp]),
    'msmc_add_ln_bwd_workspace': (_sz, [ctypes.c_long, _i]),
    'msmc_add_ln_param_multi': (_i, [_vp, _i, _vp]),
    'msmc_add_ln_bwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, ctypes.c_long, _i, _f, _vp,
                        ctypes.c_longlong, _i, _i, _vp]),
    'msmc_fft_prologue': (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'msmc_gate_fwd': (_i, [_vp, _vp, ctypes.c_long, _i, _f, _vp, ctypes.c_longlong, _i, _vp]),
    'msmc_gate_bwd': (_i, [_vp, _vp, _vp, ctypes.c_long, _i, _f, _vp, ctypes.c_longlong, _i, _vp]),
    'msmc_sum_n': (_i, [_vp, _vp, _vp, _vp, _vp, ctypes.c_long, _i, _vp]),
    'msmc_dropout_add_fwd': (_i, [_vp, _vp, _vp, ctypes.c_long, _f, _vp, ctypes.c_longlong, _i, _vp]),
    'msmc_dropout_bwd': (_i, [_vp, _vp, ctypes.c_long, _f, _vp, ctypes.c_longlong, _i, _vp]),
    'msmc_row_mask': (_i, [_vp, _i, _vp, _i, _i, _i, _vp]),
    'msmc_tanh_fwd': (_i, [_vp, _vp, ctypes.c_long, _i, _vp]),
    'msmc_tanh_bwd': (_i, [_vp, _vp, _vp, ctypes.c_long, _i, _vp]),
    'msmc_tanh_f32_fwd': (_i, [_vp, _vp, ctypes.c_long, _i, _vp]),
    'msmc_tanh_f32_bwd': (_i, [_vp, _vp, _vp, ctypes.c_long, _i, _vp]),
    'msmc_opt_chunk': (_i, []),
    'msmc_opt_clip_adamw': (_i, [_vp, _i, _i, _f, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _i, _vp]),
    'msmc_lrelu_bwd': (_i, [_vp, _vp, _vp, ctypes.c_long, _f, _i, _vp]),
    'msmc_reflect_fold': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp]),
})

# include/msmc_hip_debug.h: process-global A/B switches, ablation masks and the experimental fused ResBlock unit -- exported by
# the library for tools/ and tests/, not part of the product ABI
_DEBUG_SIGNATURES = {
    'msmc_vq_last_kernel': (ctypes.c_char_p, []),
    'msmc_conv_last_kernel': (ctypes.c_char_p, []),
    'msmc_conv_launch_count': (ctypes.c_long, []),
    'msmc_prof_enable': (None, [_i]),
    'msmc_prof_count': (_i, []),
    'msmc_prof_read': (_i, [_i, ctypes.c_char_p, _i, ctypes.POINTER(ctypes.c_float)]),
    'msmc_vq_set_shortlist_ablate': (None, [_i]),
    'msmc_vq_set_variant': (None, [_i]),
    'msmc_conv_set_grouping': (None, [_i]),
    'msmc_conv_set_pipeline': (None, [_i]),
    'msmc_conv_set_wgrad_split': (None, [_i]),
    'msmc_conv_set_wgrad_generation': (None, [_i]),
    'msmc_conv_set_wgrad4_ablate': (None, [_i]),
    'msmc_conv_set_gather4_grid': (None, [_i]),
    'msmc_conv_set_gather4_grouping': (None, [_i]),
    'msmc_conv_set_gather_generation': (None, [_i]),
    'msmc_conv_set_narrow': (None, [_i]),
    'msmc_conv_set_wgrad_tpw': (None, [_i]),
}


class LnParamItem(ctypes.Structure):
    """msmc_ln_param_item (include/msmc_hip.h)"""
    _fields_ = [('part', ctypes.c_void_p), ('dgamma', ctypes.c_void_p), ('dbeta', ctypes.c_void_p), ('nblocks', ctypes.c_int),
                ('C', ctypes.c_int), ('accumulate', ctypes.c_int), ('reserved', ctypes.c_int)]


def exported_symbols():
    """Names of the product ABI (include/msmc_hip.h) every build of the library must export (checked by the CPU test-suite)."""
    return sorted(_SIGNATURES)


def debug_symbols():
    """Names of include/msmc_hip_debug.h (switches for tools/ and tests/)."""
    return sorted(_DEBUG_SIGNATURES)


def _bind(handle):
    for name, (res, args) in list(_SIGNATURES.items()) + list(_DEBUG_SIGNATURES.items()):
        fn = getattr(handle, name)
        fn.restype = res
        fn.argtypes = args
    return handle


def load(path=None):
    path = DEFAULT_PATH if path is None else path
    if not os.path.isfile(path):
        raise RuntimeError(
            'msmctts_amd: HIP extension %s not found. Build it with `python -c "import __graft_entry__ as g; '
            'g.build()"` (hipcc --offload-arch=gfx950). There is no CPU fallback.' % path)
    return _bind(ctypes.CDLL(path))


def get():
    global _lib
    if _lib is None:
        _lib = load()
    return _lib


def use_library_for_tests(path):
    """TEST HOOK: bind another build of the C ABI (the CPU kernel interpreter, backend 'emu')."""
    global _lib, _host_pointers_ok
    _lib = load(path)
    _host_pointers_ok = _lib.msmc_backend() == b'emu'
    return _lib


def backend():
    return get().msmc_backend().decode()


def ptr(t, dtype=None):
    """Raw pointer of a contiguous tensor; refuses host memory unless the interpreter build is bound."""
    if t is None:
        return None
    if dtype is not None and t.dtype != dtype:
        raise TypeError('expected %s, got %s' % (dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError('msmc HIP ops take contiguous tensors')
    if not t.is_cuda and not _host_pointers_ok:
        raise RuntimeError('msmc HIP ops run on the GPU only (got a %s tensor); there is no CPU path' % t.device)
    return ctypes.c_void_p(t.data_ptr())


def stream(t):
    if t.is_cuda:
        return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
    return ctypes.c_void_p(0)


def check(rc, what):
    if rc != 0:
        raise RuntimeError('%s failed with code %d' % (what, rc))
