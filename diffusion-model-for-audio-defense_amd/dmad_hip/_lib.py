"""ctypes binding of libdmad_hip.so (C ABI: include/dmad.h).

The HIP library is the product path.  There is NO fallback: if the shared object is missing or a
call fails, a DmadError is raised."""
from __future__ import annotations

import ctypes as C
import os

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(_PKG_ROOT, 'libdmad_hip.so')


class DmadError(RuntimeError):
    pass


class DmadConfig(C.Structure):
    """dmad_config of include/dmad.h; positional arguments are the fields AFTER struct_size, which is filled in here."""
    _fields_ = [(n, C.c_int32) for n in (
        'struct_size', 'res_channels', 'skip_channels', 'num_res_layers', 'dilation_cycle', 'embed_dim_in', 'embed_dim_mid',
        'embed_dim_out', 'clip_len', 'max_batch', 'num_classes', 'precision', 'with_classifier', 'recheck_batch', 'half_type',
        'with_wavenet')]

    def __init__(self, *fields, **named):
        if len(fields) < 15:
            named.setdefault('with_wavenet', 1)
        super().__init__(C.sizeof(type(self)), *fields, **named)


_P = C.c_void_p
_SIGNATURES = {
    'dmad_create': (C.c_int, [C.POINTER(DmadConfig), C.POINTER(_P)]),
    'dmad_destroy': (None, [_P]),
    'dmad_last_error': (C.c_char_p, []),
    'dmad_version': (C.c_char_p, []),
    'dmad_last_warning': (C.c_char_p, []),
    'dmad_load_weight': (C.c_int, [_P, C.c_char_p, _P, C.POINTER(C.c_int64), C.c_int32]),
    'dmad_finalize_weights': (C.c_int, [_P]),
    'dmad_wavenet_eps': (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, _P]),
    'dmad_one_shot': (C.c_int, [_P, _P, C.c_int32, C.c_float, C.c_float, C.c_int32, _P, _P]),
    'dmad_ddpm_step': (C.c_int, [_P, _P, C.c_int32, C.c_float, C.c_float, C.c_float, _P, C.c_uint64, C.c_uint64, C.c_int32, _P]),
    'dmad_diffuse': (C.c_int, [_P, _P, C.c_float, C.c_float, _P, C.c_uint64, C.c_uint64, C.c_int32, _P, _P]),
    'dmad_ddpm_purify': (C.c_int, [_P, _P, C.c_int32, C.c_float, C.c_float, _P, _P, _P, C.c_uint64, C.c_uint64, C.c_int32, _P, _P]),
    'dmad_unet_eps': (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, _P]),
    'dmad_unet_eps_tier': (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    'dmad_unet_p_sample': (C.c_int, [_P, _P, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _P, C.c_uint64, C.c_uint64,
                                     C.c_int32, _P, _P]),
    'dmad_mel_db': (C.c_int, [_P, _P, C.c_int32, _P, _P]),
    'dmad_mel_power': (C.c_int, [_P, _P, C.c_int32, _P, _P]),
    'dmad_power_to_db': (C.c_int, [_P, _P, C.c_int64, _P, _P]),
    'dmad_classify': (C.c_int, [_P, _P, C.c_int32, _P, _P]),
    'dmad_classify_tier': (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, _P]),
    'dmad_conv_h16': (C.c_int, [_P, _P, C.c_int32, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                C.c_int32, _P, _P, _P]),
    'dmad_conv_h16_up2': (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P]),
    'dmad_split_f16': (C.c_int, [_P, C.c_int64, _P, _P]),
    'dmad_conv_x3': (C.c_int, [_P, _P, C.c_int32, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                               C.c_int32, _P, _P]),
    'dmad_conv_h16_stats': (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P]),
    'dmad_groupnorm16_apply': (C.c_int, [_P, _P, _P, _P, C.c_int32, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P]),
    'dmad_smooth_votes': (C.c_int, [_P, _P, C.c_float, C.c_float, C.c_int32, C.c_float, C.c_float, C.c_int64, C.c_int32,
                                    C.c_uint64, C.c_uint64, _P, _P, _P, _P, _P]),
    'dmad_set_mode': (C.c_int, [_P, C.c_int32]),
    'dmad_set_waveform_tier': (C.c_int, [_P, C.c_int32]),
    'dmad_set_recheck_margin': (C.c_int, [_P, C.c_float]),
    'dmad_set_recheck_margin2': (C.c_int, [_P, C.c_float]),
    'dmad_recheck_stats': (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int32]),
    'dmad_wavenet_eps_path': (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    'dmad_eval_samples': (C.c_int, [_P, _P, C.c_float, C.c_float, C.c_int32, C.c_float, C.c_float, C.c_uint64, C.c_uint64, _P, _P, C.c_int64,
                                    C.c_int32, _P, _P, _P]),
    'dmad_debug_rounding': (C.c_int, [_P, C.POINTER(C.c_int32)]),
    'dmad_query_logits': (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, _P, _P, _P,
                                    C.c_uint64, C.c_uint64, _P, _P, _P]),
    'dmad_spec_query_logits': (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, _P, _P, _P, _P, _P, C.c_float, C.c_float,
                                         C.c_uint64, C.c_uint64, _P, _P, _P]),
    'dmad_spec_smooth_votes': (C.c_int, [_P, _P, C.c_float, C.c_int32, C.c_float, C.c_float, _P, _P, _P, _P, _P, C.c_float, C.c_float,
                                         C.c_int64, C.c_int32, C.c_uint64, C.c_uint64, _P, _P, _P, _P]),
    'dmad_spec_eval_samples': (C.c_int, [_P, _P, C.c_float, C.c_int32, C.c_float, C.c_float, _P, _P, _P, _P, _P, C.c_float, C.c_float,
                                         C.c_uint64, _P, C.c_int64, C.c_int32, _P, _P, _P]),
    'dmad_set_spec_recheck_margin': (C.c_int, [_P, C.c_float]),
    'dmad_spec_recheck_stats': (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int32]),
    'dmad_set_spec_recheck_margin2': (C.c_int, [_P, C.c_float]),
    'dmad_spec_recheck_stats2': (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int32]),
    'dmad_vote': (C.c_int, [_P, _P, C.c_int32, _P, _P]),
    'dmad_philox_raw': (C.c_int, [_P, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, _P, _P]),
    'dmad_philox_normal': (C.c_int, [_P, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int32, _P, _P]),
    'dmad_time_layer': (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_float), _P]),
    'dmad_device_bytes': (C.c_int64, [_P]),
    'dmad_profile_layers': (C.c_int, [_P, C.c_int32]),
    'dmad_profile_read': (C.c_int, [_P, C.POINTER(C.c_float), C.POINTER(C.c_int32)]),
    'dmad_profile_read_final': (C.c_int, [_P, C.POINTER(C.c_float), C.POINTER(C.c_int32)]),
}
EXPORTS = tuple(_SIGNATURES)

_lib = None


def load():
    """Load (once) and return the ctypes library.  torch must be imported first so that the HIP
    runtime already mapped by torch (same SONAME libamdhip64.so.7) is the one the library binds to."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DmadError('HIP extension missing: %s (build it with `python -c "import __graft_entry__ as g; g.build()"` '
                        'or `make -C diffusion-model-for-audio-defense_amd/csrc`); there is no CPU fallback' % LIB_PATH)
    import torch  # noqa: F401  (maps torch's libamdhip64 first)
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int):
    if rc != 0:
        msg = load().dmad_last_error()
        raise DmadError('dmad call failed (%d): %s' % (rc, msg.decode() if msg else '?'))
