"""ctypes binding of libty1.so (include/ty1.h).  There is no CPU fallback: if the HIP library is
missing or fails to load, importing an op raises."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# TY1_LIB points a developer A/B run at another build of the same library (tools/abl.sh)
LIB_PATH = os.environ.get('TY1_LIB') or os.path.join(_HERE, 'lib', 'libty1.so')

_c = ctypes
_vp, _i, _f, _d, _i64 = _c.c_void_p, _c.c_int, _c.c_float, _c.c_double, _c.c_int64

# name -> argtypes, exactly the declarations of include/ty1.h (tests/test_abi.py checks both ways)
SIGNATURES = {
    'ty1_fold_bn': [_vp, _vp, _vp, _vp, _f, _i, _vp, _vp],
    'ty1_conv3x3_bn_leaky_f32': [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp],
    'ty1_conv3x3_workspace_bytes': [_i, _i, _i, _i, _i, _i, _vp],
    'ty1_conv3x3_bn_leaky_ws_f32': [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp, _c.c_size_t, _vp],
    'ty1_conv3x3_bn_leaky_static_f32': [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp],
    'ty1_static_scratch_bytes': [],
    'ty1_static_scratch_init': [],
    'ty1_conv3x3_workspace_bytes_bf16': [_i, _i, _i, _i, _i, _i, _vp],
    'ty1_conv3x3_bn_leaky_bf16': [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _vp, _c.c_size_t, _vp],
    'ty1_conv3x3_x2_c3_bf16': [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp],
    'ty1_cast_f32_to_bf16': [_vp, _vp, _i64, _vp],
    'ty1_preprocess_u8': [_vp, _i, _i, _i, _vp, _i, _i, _vp],
    'ty1_preprocess_batch_u8': [_vp, _i, _i, _i, _i, _vp, _i, _i, _vp],
    'ty1_dense_splitk_f32': [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    'ty1_dense_splitk_bf16': [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    'ty1_head_activate_f32': [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    'ty1_decode_nms': [_vp, _i, _i, _i, _i, _i, _d, _vp, _vp, _vp, _vp, _vp],
    'ty1_head_decode_nms_f32': [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _d, _vp, _vp, _vp, _vp, _vp],
    'ty1_pack_detections': [_vp, _vp, _vp, _vp, _vp, _i64, _vp],
    'ty1_yolo_head_np_f64': [_vp, _vp, _vp, _i, _i, _i, _i, _vp],
    'ty1_iou_f64': [_vp, _vp, _vp, _vp, _vp, _i64, _vp],
    'ty1_yolo_head_f32': [_vp, _vp, _vp, _i, _i, _i, _i, _vp],
    'ty1_iou_f32': [_vp, _vp, _vp, _vp, _vp, _i64, _vp],
    'ty1_yolo_loss_fwd_bwd': [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp],
    'ty1_head_activate_bwd_f32': [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    'ty1_train_tail_f32': [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp],
    'ty1_dense_wgrad_f32': [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    'ty1_adam_step_f32': [_vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _i64, _vp],
    'ty1_set_reserved_cus': [_i],
    'ty1_get_reserved_cus': [_c.POINTER(_i)],
    'ty1_prof_enable': [_i],
    'ty1_prof_reset': [],
    'ty1_prof_filter': [_c.c_char_p],
    'ty1_prof_stride': [_i],
    'ty1_prof_query': [_c.c_char_p, _c.POINTER(_i), _c.POINTER(_d)],
    'ty1_prof_dump': [_c.c_char_p, _c.c_size_t],
    'ty1_debug_stamp_buffer': [_vp, _i64],
    'ty1_version': [],
}

_LIB = None


class Ty1Error(RuntimeError):
    pass


def lib():
    """Load libty1.so once.  Raises (never falls back) if it is absent."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise Ty1Error(f'{LIB_PATH} not found: build it with `make -C yolo_amd/csrc` '
                           '(or __graft_entry__.build()); there is no CPU fallback')
        L = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = argtypes
            fn.restype = _c.c_size_t if name == 'ty1_static_scratch_bytes' else _i
        L.ty1_last_error.restype = _c.c_char_p
        L.ty1_last_error.argtypes = []
        _LIB = L
    return _LIB


def call(name, *args):
    L = lib()
    rc = getattr(L, name)(*args)
    if rc != 0:
        raise Ty1Error(L.ty1_last_error().decode())


def ptr(t):
    """Device (or host) pointer of a torch tensor / numpy array, None -> NULL."""
    if t is None:
        return None
    if hasattr(t, 'data_ptr'):
        return t.data_ptr()
    return t.ctypes.data


def current_stream():
    import torch
    return torch.cuda.current_stream().cuda_stream
