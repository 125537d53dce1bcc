"""ctypes binding of libsfnative.so (C ABI: include/sfnative.h).

The library is the product: there is no CPU or PyTorch fallback.  `lib()` raises if the shared
object is missing or does not export every symbol the header declares.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SF_LIB_PATH") or os.path.join(_HERE, "libsfnative.so")   # SF_LIB_PATH: experiment builds only

f32p = C.POINTER(C.c_float)
i32p = C.POINTER(C.c_int32)

SF_COEF_STRIDE = 12
SF_ABI_VERSION = 6       # include/sfnative.h: changes whenever a public struct changes layout
SF_PROF_KEYS = 168
PACK_TRANSPOSED, PACK_FOLD_DUP, PACK_INTERLEAVE, PACK_BF16X3, PACK_WINOGRAD = 1, 2, 4, 8, 16      # SF_PACK_* of sfnative.h
ACT = {"none": 0, "lrelu": 1, "relu": 2, "tanh": 3, "sigmoid": 4, "gelu": 5}
SOLVER = {"euler": 0, "midpoint": 1, "rk4": 2}
OP_JUMP, OP_STEP = 0, 1


class ConvW(C.Structure):
    _fields_ = [("w", C.c_void_p), ("scale", C.c_void_p), ("bias", C.c_void_p),
                ("cout", C.c_int32), ("cout_pad", C.c_int32), ("c0", C.c_int32), ("c1", C.c_int32),
                ("cin_pad", C.c_int32), ("kh", C.c_int32), ("kw", C.c_int32), ("dil", C.c_int32),
                ("stride", C.c_int32), ("pad", C.c_int32), ("act", C.c_int32), ("reserved", C.c_int32), ("w_bf16x3", C.c_void_p), ("w_wino", C.c_void_p)]


class GruW(C.Structure):
    _fields_ = [("gates", ConvW), ("cand", ConvW), ("decoder", ConvW)]


class DualW(C.Structure):
    _fields_ = [("gates1", ConvW), ("cand1", ConvW), ("gates2", ConvW), ("cand2", ConvW), ("dec2", ConvW),
                ("tg7", ConvW), ("tgproj", ConvW), ("tg1", ConvW), ("tg3", ConvW),
                ("w_logit", C.c_void_p), ("C", C.c_int32), ("gates1_x", ConvW), ("gates1_s", ConvW), ("tg7_h", ConvW), ("tg7_r", ConvW)]


class BottleW(C.Structure):
    _fields_ = [("c7", ConvW), ("c1", ConvW), ("c3", ConvW), ("proj", ConvW)]


class ResW(C.Structure):
    _fields_ = [("conv1", ConvW), ("conv2", ConvW), ("proj", ConvW)]


class PModelW(C.Structure):
    _fields_ = [("rb0", ResW), ("rb1", ResW), ("se0_fc0", C.c_void_p), ("se0_fc2", C.c_void_p),
                ("se1_fc0", C.c_void_p), ("se1_fc2", C.c_void_p), ("last", ConvW), ("C", C.c_int32)]


class EncoderW(C.Structure):
    _fields_ = [("blocks", ResW * 5), ("last", ConvW), ("C", C.c_int32), ("F", C.c_int32)]


class DecoderW(C.Structure):
    _fields_ = [("first", ConvW), ("blocks", ResW * 5), ("last0", ConvW), ("last1", ConvW),
                ("C", C.c_int32), ("F", C.c_int32)]


class ConvNextW(C.Structure):
    _fields_ = [("dw_w", C.c_void_p), ("dw_b", C.c_void_p), ("ln_w", C.c_void_p), ("ln_b", C.c_void_p),
                ("pw1", ConvW), ("pw2", ConvW), ("C", C.c_int32)]


class DeepLabW(C.Structure):
    _fields_ = [("branch", ConvW * 4), ("pool_w", C.c_void_p), ("pool_scale", C.c_void_p),
                ("pool_bias", C.c_void_p), ("proj_pool_w", C.c_void_p), ("project", ConvW),
                ("conv3", ConvW), ("cls", ConvW), ("C", C.c_int32), ("hid", C.c_int32)]


class BottleneckW(C.Structure):
    _fields_ = [("down", ConvW), ("conv", ConvW), ("up", ConvW), ("proj", ConvW), ("downsample", C.c_int32)]


# SF_STRUCT_* order of sfnative.h: what lib() hands to sf_abi_check
ABI_STRUCTS = (ConvW, GruW, DualW, ResW, PModelW, EncoderW, DecoderW, ConvNextW, DeepLabW, BottleneckW, BottleW)

_vp, _i, _sz = C.c_void_p, C.c_int, C.c_size_t
# name -> (restype, argtypes); every symbol declared in include/sfnative.h
_f3 = C.POINTER(C.c_float * 3)
_i3 = C.POINTER(C.c_int32 * 3)
_f6 = C.POINTER(C.c_float * 6)
SIGNATURES = {
    "sf_version": (_i, []),
    "sf_set_flow_mode": (_i, [_i]),
    "sf_flow_errors": (_i, [_vp]),
    "sf_abi_version": (_i, []),
    "sf_abi_sizeof": (_sz, [_i]),
    "sf_abi_check": (_i, [_i, C.POINTER(_sz), _i]),
    "sf_status_string": (C.c_char_p, [_i]),
    "sf_nchw_to_nhwc": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "sf_nhwc_to_nchw": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "sf_nchw_to_nhwc_strided": (_i, [_vp, _sz, _vp, _sz, _i, _i, _i, _vp]),
    "sf_nhwc_to_nchw_strided": (_i, [_vp, _sz, _vp, _sz, _i, _i, _i, _vp]),
    "sf_conv2d_fwd": (_i, [C.POINTER(ConvW), _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "sf_conv2d_repeat": (_i, [C.POINTER(ConvW), _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "sf_gru_cell_fwd": (_i, [C.POINTER(GruW), _vp, _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "sf_gru_cell_ws_bytes": (_sz, [_i, _i, _i, _i]),
    "sf_gru_ode_cell_fwd": (_i, [C.POINTER(GruW), _vp, _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "sf_spatial_gru_fwd": (_i, [C.POINTER(GruW), _vp, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    "sf_spatial_gru_ws_bytes": (_sz, [_i, _i, _i, _i]),
    "sf_dual_cell_fwd": (_i, [C.POINTER(DualW), _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    "sf_dual_cell_ws_bytes": (_sz, [_i, _i, _i, _i]),
    "sf_trust_mix_fwd": (_i, [C.POINTER(DualW), _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "sf_bottleblock_fwd": (_i, [C.POINTER(BottleW), _vp, _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "sf_bottleblock_ws_bytes": (_sz, [_i, _i, _i, _i]),
    "sf_infer_state_fwd": (_i, [C.POINTER(PModelW), _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "sf_infer_state_ws_bytes": (_sz, [_i, _i, _i, _i]),
    "sf_ode_step_fwd": (_i, [C.POINTER(DualW), C.POINTER(PModelW), _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i,
                             _vp, _sz, _vp]),
    "sf_ode_step_ws_bytes": (_sz, [_i, _i, _i, _i]),
    "sf_nnfo_rollout_fwd": (_i, [C.POINTER(DualW), C.POINTER(DualW), C.POINTER(PModelW), _i, _i, i32p, _i, _vp, _vp,
                                 _vp, _i, i32p, _i, _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "sf_nnfo_rollout_philox_fwd": (_i, [C.POINTER(DualW), C.POINTER(DualW), C.POINTER(PModelW), _i, _i, i32p, _i, _vp, _vp,
                                        _vp, _i, i32p, _i, _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "sf_infer_state_philox_fwd": (_i, [C.POINTER(PModelW), _vp, _vp, _i, _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "sf_nnfo_rollout_ws_bytes": (_sz, [_i, _i, _i, _i]),
    "sf_small_encoder_fwd": (_i, [C.POINTER(EncoderW), _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "sf_small_encoder_ws_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "sf_small_decoder_fwd": (_i, [C.POINTER(DecoderW), _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "sf_small_decoder_ws_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "sf_convnext_block_fwd": (_i, [C.POINTER(ConvNextW), _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "sf_convnext_block_ws_bytes": (_sz, [_i, _i, _i, _i]),
    "sf_deeplab_head_fwd": (_i, [C.POINTER(DeepLabW), _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "sf_deeplab_head_planar_fwd": (_i, [C.POINTER(DeepLabW), _vp, _vp, _i, _i, _i, _i, _sz, _sz, _vp, _sz, _vp]),
    "sf_deeplab_head_ws_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "sf_bottleneck_fwd": (_i, [C.POINTER(BottleneckW), _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "sf_bottleneck_ws_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "sf_dist_head_fwd": (_i, [C.POINTER(ConvW), _vp, _vp, _i, _i, _i, _i, _i, C.c_float, C.c_float, _vp, _sz, _vp]),
    "sf_dist_head_ws_bytes": (_sz, [_i, _i]),
    "sf_conv2d_ex_fwd": (_i, [C.POINTER(ConvW), _vp, _i, _vp, _i, _vp, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "sf_conv2d_ex_ws_bytes": (_sz, []),
    "sf_upsample_bilinear2_add_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "sf_channel_mean_ws_bytes": (_sz, [_i, _i]),
    "sf_channel_mean_fwd": (_i, [_vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "sf_broadcast_channels_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "sf_bev_pool_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "sf_lift_index_ws_bytes": (_sz, [_i, _i]),
    "sf_lift_index_fwd": (_i, [_vp, _i, _i, _f3, _f3, _i3, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sf_lift_index_coords_fwd": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "sf_lift_index_rig_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f3, _f3, _i3, _vp, _vp, _vp, _sz, _vp]),
    "sf_lift_pool_fwd": (_i, [_vp, _vp, _vp, _i, _i, _vp, C.c_float, _vp, _vp]),
    "sf_lift_pool_fused_fwd": (_i, [_vp, _vp, _i, _i, _vp, _vp, _i, _i, _vp, C.c_float, _vp, _vp]),
    "sf_depth_softmax_fwd": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "sf_hard_voxelize_ws_bytes": (_sz, [_i]),
    "sf_logsigmoid_fwd": (_i, [_vp, _vp, _sz, _vp]),
    "sf_dynamic_voxelize_fwd": (_i, [_vp, _i, _i, _f3, _f6, _vp, _vp]),
    "sf_hard_voxelize_fwd": (_i, [_vp, _i, _i, _f3, _f6, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sf_sparse_index_ws_bytes": (_sz, [_i, _i]),
    "sf_sparse_out_sites_fwd": (_i, [_vp, _i, _i, _i3, _i3, _i3, _i3, _vp, _i, _vp, _vp, _sz, _vp]),
    "sf_sparse_table_fwd": (_i, [_vp, _i, _vp, _i, _i, _i3, _i3, _i3, _i3, _i, _vp, _vp, _sz, _vp]),
    "sf_sparse_conv_fwd": (_i, [C.POINTER(ConvW), _vp, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _sz, _vp]),
    "sf_sparse_conv_masked_fwd": (_i, [C.POINTER(ConvW), _vp, _i, _i, _vp, _vp, _i, _vp, _i, _vp, _vp, _sz, _vp]),
    "sf_sparse_to_dense_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "sf_warp_affine_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "sf_confusion_fwd": (_i, [_vp, _vp, C.c_long, _i, _vp, _vp, _vp]),
    "sf_instance_centers_ws_bytes": (_sz, [_i, _i]),
    "sf_instance_centers_fwd": (_i, [_vp, _i, _i, C.c_float, _vp, _i, _vp, _vp, _sz, _vp]),
    "sf_group_pixels_fwd": (_i, [_vp, _i, _vp, _vp, _i, _i, _vp, _vp]),
    "sf_instance_moments_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "sf_confusion_frames_fwd": (_i, [_vp, _vp, C.c_long, _i, _i, _vp, _vp, _vp]),
    "sf_graph_begin": (_i, [_vp]),
    "sf_graph_end": (_i, [_vp, C.POINTER(_vp)]),
    "sf_graph_launch": (_i, [_vp, _vp]),
    "sf_graph_destroy": (_i, [_vp]),
    "sf_event_create": (_i, [C.POINTER(_vp)]),
    "sf_event_record": (_i, [_vp, _vp]),
    "sf_event_elapsed_ms": (_i, [_vp, _vp, C.POINTER(C.c_float)]),
    "sf_event_destroy": (_i, [_vp]),
    "sf_prof_enable": (_i, [_i]),
    "sf_prof_collect": (_i, [i32p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "sf_debug_stamps": (_i, [_vp]),
    "sf_pack_conv_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "sf_debug_occupancy": (_i, [_i]),
    "sf_bn_fold": (_i, [_vp, _vp, _vp, _vp, _vp, C.c_float, _i, _vp, _vp, _vp]),
    "sf_pack_conv": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_float, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, C.POINTER(ConvW), _vp]),
}

# kernel key = tile_config*8 + epilogue  (csrc/conv_igemm.hip launch_conv; configs 10..13: the LDS-DMA kernel conv_glds_kernel)
KERNEL_NAMES = {c * 8 + e: (f"convnext_mlp<{cn[3:]},gelu+residual>" if cn.startswith("mlp") else f"conv_wino<{cn[4:]},{en}>" if cn.startswith("wino") else f"conv_glds<{cn[3:]},{en}>" if cn.startswith("dma") else f"conv_sp<{cn[2:]},{en}>" if cn.startswith("sp") else f"conv_igemm<{cn},{en}>")
                for c, cn in enumerate(("S16x64k4", "L64x64", "LN64x128", "direct16px", "T64x64splitK", "dmaLN128x64", "L64x128", "L128x128w4", "L64x128w8", "L128x128w8",
                                        "dma128x128w8", "dma64x64", "dma64x128w8", "dmaLN64x128", "sp64x32", "dmaT64x64splitK", "wino128x32t", "wino64x64t", "wino64x32t2", "wino64x32t2dil",
                                        "mlp64-256-64"))
                for e, en in enumerate(("affine", "blend", "ln_gelu", "trust", "sample"))}

_LIB = None


def lib():
    """Load (once) and return the ctypes handle; raises RuntimeError when the HIP library is absent."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -m streamingflow_amd.build` "
                "(hipcc --offload-arch=gfx950). streamingflow_amd has no CPU fallback.")
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)     # AttributeError => missing export
            fn.restype, fn.argtypes = res, args
        # ABI guard: this binding's struct layouts against the ones the library was compiled with
        sizes = (_sz * len(ABI_STRUCTS))(*[C.sizeof(t) for t in ABI_STRUCTS])
        if h.sf_abi_check(SF_ABI_VERSION, sizes, len(ABI_STRUCTS)) != 0:
            theirs = [h.sf_abi_sizeof(i) for i in range(len(ABI_STRUCTS))]
            raise RuntimeError(f"{LIB_PATH}: ABI mismatch (library ABI {h.sf_abi_version()}, sizes {theirs}; binding ABI "
                               f"{SF_ABI_VERSION}, sizes {list(sizes)}): rebuild with `python -m streamingflow_amd.build`")
        _LIB = h
    return _LIB


def check(status, what=""):
    if status != 0:
        msg = lib().sf_status_string(status).decode()
        raise RuntimeError(f"libsfnative {what}: {msg} ({status})")
