"""ctypes binding of libyolo3hip.so (C ABI: include/yolo3hip.h).

There is no CPU fallback: importing this module without the built library, or
calling an entry point that fails, raises.  Build with
``make -C object-detection-yolov3_amd/csrc`` (or ``__graft_entry__.build()``).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('Y3_LIB') or os.path.join(_HERE, '_lib', 'libyolo3hip.so')     # Y3_LIB: A/B two builds (tools/)


class HipLibraryMissing(ImportError):
    pass


class HipError(RuntimeError):
    pass


if not os.path.exists(LIB_PATH):
    raise HipLibraryMissing(
        'libyolo3hip.so not found at %s: build it with `make -C object-detection-yolov3_amd/csrc` '
        '(hipcc --offload-arch=gfx950).  There is no CPU fallback.' % LIB_PATH)

lib = C.CDLL(LIB_PATH)


class Tensor(C.Structure):
    """y3_tensor: NHWC view with pixel pitch ld (floats)."""
    _fields_ = [('ptr', C.c_void_p), ('n', C.c_int), ('h', C.c_int), ('w', C.c_int), ('c', C.c_int), ('ld', C.c_int)]


TP = C.POINTER(Tensor)
vp, fp, ip = C.c_void_p, C.c_void_p, C.c_void_p     # device pointers travel as integers
i32, u32, f32, sz = C.c_int, C.c_uint, C.c_float, C.c_size_t

# name -> (restype, argtypes); every symbol declared in include/yolo3hip.h
SIGNATURES = {
    'y3_last_error': (C.c_char_p, []),
    'y3_debug_div': (i32, [i32, i32]),
    'y3_version': (i32, []),
    'y3_conv2d_fwd': (i32, [TP, fp, fp, i32, i32, TP, u32, f32, fp, fp, TP, fp, vp, sz, vp]),
    'y3_conv2d_stats_tiles': (i32, [i32, i32, i32, i32]),
    'y3_conv2d_fwd_workspace': (sz, [i32, i32, i32, i32]),
    'y3_conv2d_dgrad': (i32, [TP, fp, i32, i32, TP, u32, vp, sz, vp]),
    'y3_conv2d_stats_tiles_x': (i32, [i32, i32, i32, i32, u32]),
    'y3_conv2d_fwd_workspace_x': (sz, [i32, i32, i32, i32, u32]),
    'y3_conv2d_x3_ok': (i32, [i32, i32, i32, i32]),
    'y3_conv2d_dgrad_x3_ok': (i32, [TP, i32, i32, TP]),
    'y3_x3_split_weights': (i32, [fp, vp, i32, i32, i32, vp]),
    'y3_x3_split_weights_batched': (i32, [fp, vp, ip, i32, i32, vp]),
    'y3_x3_prepare_weights_batched': (i32, [fp, fp, vp, vp, ip, i32, i32, vp]),
    'y3_conv2d_dgrad_workspace_x': (sz, [TP, i32, i32, TP, u32]),
    'y3_conv2d_dgrad_bn_tiles_x': (i32, [TP, i32, i32, TP, u32]),
    'y3_conv2d_dgrad_workspace': (sz, [TP, i32, i32, TP]),
    'y3_conv2d_dgrad_bn': (i32, [TP, fp, i32, i32, TP, u32, TP, fp, vp, sz, vp]),
    'y3_conv2d_dgrad_bn_tiles': (i32, [TP, i32, i32, TP]),
    'y3_conv2d_wgrad': (i32, [TP, TP, i32, i32, fp, vp, sz, vp]),
    'y3_conv2d_wgrad_workspace': (sz, [TP, TP, i32, i32]),
    'y3_conv2d_wgrad_x': (i32, [TP, TP, i32, i32, fp, u32, vp, sz, vp]),
    'y3_conv2d_wgrad_workspace_x': (sz, [TP, TP, i32, i32, u32]),
    'y3_conv2d_wgrad_x3_ok': (i32, [i32, i32, i32, i32]),
    'y3_transpose_weights': (i32, [fp, fp, i32, i32, i32, vp]),
    'y3_transpose_weights_batched': (i32, [fp, fp, ip, i32, i32, vp]),
    'y3_bn_stats_finalize': (i32, [fp, i32, i32, i32, fp, fp, f32, f32, fp, fp, fp, fp, fp, fp, vp]),
    'y3_bn_fold_inference': (i32, [fp, fp, fp, fp, f32, i32, fp, fp, vp]),
    'y3_bn_fold_inference_batched': (i32, [fp, fp, fp, ip, i32, f32, vp]),
    'y3_bn_apply': (i32, [TP, fp, fp, TP, TP, vp]),
    'y3_bn_bwd_stats': (i32, [TP, TP, TP, i32, fp, fp, fp, f32, fp, fp, fp, fp, vp, sz, vp]),
    'y3_bn_bwd_workspace': (sz, [i32, i32]),
    'y3_bn_bwd_apply': (i32, [TP, TP, fp, f32, TP, vp]),
    'y3_bn_bwd_apply_fanin': (i32, [TP, TP, fp, f32, TP, TP, i32, vp]),
    'y3_bn_bwd_finalize_tiles': (i32, [fp, i32, i32, i32, fp, fp, fp, f32, fp, fp, fp, fp, vp]),
    'y3_upsample_sum2x_fwd': (i32, [TP, TP, vp]),
    'y3_upsample_sum2x_bwd': (i32, [TP, TP, vp]),
    'y3_copy': (i32, [TP, TP, vp]),
    'y3_conv2d_fwd_bf16': (i32, [TP, vp, fp, i32, i32, TP, i32, u32, f32, fp, fp, TP, vp]),
    'y3_conv2d_fwd_bf16_ws': (i32, [TP, vp, fp, i32, i32, TP, i32, u32, f32, fp, fp, TP, vp, sz, vp]),
    'y3_conv2d_fwd_bf16_workspace': (sz, [i32, i32, i32, i32]),
    'y3_f32_to_bf16': (i32, [fp, vp, sz, vp]),
    'y3_conv2d_first_bf16': (i32, [TP, fp, fp, TP, u32, f32, fp, fp, vp]),
    'y3_tile_gather': (i32, [vp, i32, i32, i32, i32, ip, i32, i32, i32, fp, vp]),
    'y3_tile_gather_zscore_nhwc': (i32, [vp, i32, i32, i32, i32, ip, i32, i32, i32, fp, i32, vp, vp]),
    'y3_upsample_sum2x_fwd_bf16': (i32, [TP, TP, vp]),
    'y3_add_inplace': (i32, [TP, TP, vp]),
    'y3_fill': (i32, [fp, sz, f32, vp]),
    'y3_nchw_to_nhwc': (i32, [fp, i32, i32, i32, i32, TP, vp]),
    'y3_nhwc_to_nchw': (i32, [TP, fp, vp]),
    'y3_colsum': (i32, [TP, fp, vp]),
    'y3_decode_fwd': (i32, [TP, i32, C.POINTER(C.c_float), i32, i32, i32, i32, fp, vp]),
    'y3_loss_fwd_bwd': (i32, [TP, fp, C.POINTER(C.c_float), i32, i32, i32, i32, f32, fp, TP, vp, vp]),
    'y3_loss_workspace_bytes': (sz, []),
    'y3_adam_step': (i32, [fp, fp, fp, fp, sz, fp, f32, f32, f32, vp]),
    'y3_nms_per_class': (i32, [fp, i32, i32, i32, f32, f32, f32, f32, f32, ip, ip, fp, i32, vp, sz, vp]),
    'y3_nms_workspace_bytes': (sz, [i32, i32, i32]),
    'y3_nms_single_class': (i32, [fp, i32, f32, ip, ip, fp, vp, sz, vp]),
    'y3_filter_small_boxes': (i32, [fp, i32, i32, f32, ip, ip, vp]),
    'y3_compute_iou': (i32, [fp, fp, i32, i32, fp, vp]),
    'y3_comm_unique_id': (i32, [vp]),
    'y3_comm_init': (i32, [vp, i32, i32, C.POINTER(C.c_void_p)]),
    'y3_allreduce_sum_f32': (i32, [vp, fp, sz, vp]),
    'y3_conv2d_plan': (sz, [i32, i32, i32, i32, C.POINTER(C.c_int)]),
    'y3_conv2d_wgrad_plan': (sz, [i32, i32, i32, i32, C.POINTER(C.c_int)]),
    'y3_conv2d_plan_x': (sz, [i32, i32, i32, i32, u32, C.POINTER(C.c_int)]),
    'y3_conv2d_wgrad_plan_x': (sz, [i32, i32, i32, i32, u32, C.POINTER(C.c_int)]),
    'y3_comm_info': (i32, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    'y3_comm_destroy': (i32, [vp]),
    'y3_zscore': (i32, [fp, fp, i32, sz, vp, vp]),
    'y3_zscore_workspace_bytes': (sz, [i32]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _f = getattr(lib, _name)          # AttributeError if the library lacks a declared symbol
    _f.restype = _res
    _f.argtypes = _args

EPI_LRELU = 1
EPI_ACCUM = 2
BF16_NO_PATCH = 8   # Y3_BF16_NO_PATCH: keep a y3_conv2d_fwd_bf16 launch off the patch kernels (small batches, yolo3hip.h)
CONV_X3 = 4      # Y3_CONV_X3: fp32 arithmetic as three bf16 pieces per operand (conv_x3.hip); the weight operand changes layout


def check(rc, what=''):
    if rc != 0:
        raise HipError('%s failed (%d): %s' % (what or 'libyolo3hip call', rc, lib.y3_last_error().decode()))


def view(t, n, h, w, c, ld=None, offset=0):
    """Tensor struct over a torch CUDA tensor's storage (offset and ld in elements of t's dtype)."""
    return Tensor(t.data_ptr() + t.element_size() * offset, n, h, w, c, c if ld is None else ld)


def float_array(vals):
    arr = (C.c_float * len(vals))(*[float(v) for v in vals])
    return arr
