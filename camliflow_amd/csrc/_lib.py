"""ctypes binding of libcamli_hip.so (prototypes: include/camli_hip.h)."""
import ctypes
import os

from .build import LIB_PATH

_c_float_p = ctypes.c_void_p   # device pointers travel as integers
_c_i64_p = ctypes.c_void_p
_int = ctypes.c_int
_stream = ctypes.c_void_p

# name -> (restype, argtypes); every symbol include/camli_hip.h declares
PROTOTYPES = {
    "camli_version": (_int, []),
    "camli_last_error_string": (ctypes.c_char_p, []),
    "camli_knn": (_int, [_c_float_p, _c_float_p, _c_i64_p, _int, _int, _int, _int, _int, _stream]),
    "camli_knn_prefixes": (_int, [_c_float_p, _c_float_p, ctypes.c_void_p, ctypes.c_void_p, _int, _int, _int, _int, _int, _int, _stream]),
    "camli_knn_prefixes_prior": (_int, [_c_float_p, _c_float_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, _int, _int, _int, _int,
                                        _int, _int, _stream]),
    "camli_fps": (_int, [_c_float_p, _c_i64_p, _int, _int, _int, _stream]),
    "camli_corr2d_fwd": (_int, [_c_float_p, _c_float_p, _c_float_p, _int, _int, _int, _int, _int, _stream]),
    "camli_corr2d_bwd": (_int, [_c_float_p, _c_float_p, _c_float_p, _c_float_p, _c_float_p,
                                _int, _int, _int, _int, _int, _stream]),
    "camli_allpairs_build_fwd": (_int, [_c_float_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, _int, _int, _int, _int,
                                        ctypes.c_float, _stream]),
    "camli_allpairs_build_bwd": (_int, [_c_float_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, _int, _c_float_p,
                                        ctypes.c_void_p, _int, _int, _int, ctypes.c_float, _stream]),
    "camli_allpairs_build_bwd_marked": (_int, [_c_float_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, _int, _c_float_p,
                                               ctypes.c_void_p, _int, _int, _int, ctypes.c_float, ctypes.c_void_p, _stream]),
    "camli_transpose_planes": (_int, [_c_float_p, ctypes.c_int64, ctypes.c_int64, _c_float_p, ctypes.c_int64, ctypes.c_int64, _int,
                                      _int, _int, _stream]),
    "camli_allpairs_build_bwd_workspace_bytes": (ctypes.c_int64, [ctypes.c_void_p, _int, _int, _int, _int]),
    "camli_allpairs_build_bwd_splitk": (_int, [_c_float_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, _int, _c_float_p,
                                               ctypes.c_void_p, _int, _int, _int, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_int64, _stream]),
    "camli_allpairs_lookup_bwd_marked": (_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, _int, _c_float_p,
                                                _c_float_p, _int, _int, _int, _int, ctypes.c_void_p, _stream]),
    "camli_allpairs_clear_marked": (_int, [ctypes.c_void_p, ctypes.c_void_p, _int, ctypes.c_void_p, _int, _int, _stream]),
    "camli_allpairs_lookup_fwd": (_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, _int, _c_float_p,
                                         _c_float_p, _int, _int, _int, _int, _stream]),
    "camli_allpairs_lookup_bwd": (_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, _int, _c_float_p,
                                         _c_float_p, _int, _int, _int, _int, _stream]),
    "camli_pointconv_dw_fwd": (_int, [_c_float_p, _c_float_p, _c_i64_p, _int, _c_float_p, ctypes.c_void_p,
                                      _c_float_p, ctypes.c_void_p, _int, _int, _int, _int, _int, _stream]),
    "camli_pointconv_dw_fwd_kmajor": (_int, [_c_float_p, _c_float_p, ctypes.c_void_p, _c_float_p, ctypes.c_void_p,
                                             _c_float_p, ctypes.c_void_p, _int, _int, _int, _int, _int, _stream]),
    "camli_pointconv_dw_bwd": (_int, [_c_float_p, _c_float_p, _c_float_p, ctypes.c_void_p, _c_float_p, _c_float_p,
                                      _int, _int, _int, _int, _stream]),
    "camli_pointconv_dw_bwd_strided": (_int, [_c_float_p, ctypes.c_int64, _c_float_p, _c_float_p, ctypes.c_void_p, _c_float_p, _c_float_p,
                                      _int, _int, _int, _int, _stream]),
    "camli_pointconv_dw_bwd_ordered": (_int, [_c_float_p, _c_float_p, _c_float_p, ctypes.c_void_p, _c_float_p, _c_float_p,
                                      _int, _int, _int, _int, _stream]),
    "camli_pointconv_dw_expand": (_int, [ctypes.c_void_p, ctypes.c_void_p, _int, _c_float_p,
                                         _int, _int, _int, _int, _int, _stream]),
    "camli_gather_cf_fwd": (_int, [_c_float_p, _c_i64_p, _c_float_p, _int, _int, _int, _int, _stream]),
    "camli_gather_cf_bwd": (_int, [_c_float_p, _c_i64_p, _c_float_p, _int, _int, _int, _int, _stream]),
    "camli_gather_cf_bwd_sorted": (_int, [_c_float_p, ctypes.c_void_p, ctypes.c_void_p, _c_float_p, _int, _int, _int, _int, _stream]),
    "camli_gather_cl_fwd": (_int, [_c_float_p, _c_i64_p, _c_float_p, _int, _int, _int, _int, _stream]),
    "camli_gather_cl_bwd_sorted": (_int, [_c_float_p, ctypes.c_void_p, ctypes.c_void_p, _c_float_p, _int, _int, _int, _int, _stream]),
    "camli_pointconv_mix_bwd_scratch_bytes": (ctypes.c_int64, [_int, _int, _int, _int]),
    "camli_pointconv_mix_bwd_sorted": (_int, [_c_float_p, _c_float_p, _c_float_p, _c_i64_p, _int, ctypes.c_void_p, ctypes.c_void_p,
                                              _c_float_p, _c_float_p, _c_float_p, _int, _int, _int, _int, _int, _int, _stream]),
    "camli_knn_interp_fwd": (_int, [_c_float_p, _c_float_p, _c_float_p, _c_i64_p, _int, _c_float_p,
                                    _int, _int, _int, _int, _int, _stream]),
    "camli_knn_interp_bwd": (_int, [_c_float_p, _c_float_p, _c_float_p, _c_i64_p, _int, _c_float_p,
                                    _int, _int, _int, _int, _int, _stream]),
    "camli_knn_interp_weights": (_int, [_c_float_p, _c_float_p, _c_i64_p, _int, _c_float_p, _int, _int, _int, _int, _stream]),
    "camli_knn_interp_bwd_sorted": (_int, [_c_float_p, _c_float_p, ctypes.c_void_p, ctypes.c_void_p, _c_float_p,
                                           _int, _int, _int, _int, _stream]),
    "camli_knn_interp_bwd_xyz": (_int, [_c_float_p, _c_float_p, _c_float_p, _c_float_p, _c_i64_p, _int, _c_float_p,
                                        _c_float_p, _int, _int, _int, _int, _int, _stream]),
    "camli_corr3d_gather_fwd": (_int, [_c_float_p, _c_float_p, _c_float_p, _c_i64_p, _c_float_p,
                                       _int, _int, _int, _int, _stream]),
    "camli_corr3d_gather_levels_fwd": (_int, [_c_float_p, _c_float_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, _int,
                                              _c_float_p, _int, _int, _int, _int, _stream]),
    "camli_corr3d_gather_levels_bwd": (_int, [_c_float_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, _int,
                                              _int, _int, _int, _int, _stream]),
    "camli_corr3d_gather_bwd": (_int, [_c_float_p, _c_i64_p, _c_float_p, _int, _int, _int, _int, _stream]),
    "camli_pwc3d_pair_fwd": (_int, [_c_float_p, _c_float_p, _c_float_p, _c_i64_p, _c_float_p, _int, _int, _int, _int, _int,
                                    ctypes.c_float, _stream]),
    "camli_pwc3d_pair_bwd": (_int, [_c_float_p, _c_float_p, _c_float_p, _c_float_p, _int, _int, _int, _int, ctypes.c_float, _stream]),
    "camli_ksum_fwd": (_int, [_c_float_p, _c_float_p, _c_float_p, _int, _int, _int, _int, _stream]),
    "camli_ksum_bwd": (_int, [_c_float_p, _c_float_p, _c_float_p, _c_float_p, _c_float_p, _int, _int, _int, _int, _stream]),
    "camli_gather_wsum_fwd": (_int, [_c_float_p, _c_float_p, _c_i64_p, _c_float_p, _int, _int, _int, _int, _int, _stream]),
    "camli_gather_wsum_bwd": (_int, [_c_float_p, _c_float_p, _c_float_p, _c_i64_p, _c_float_p, _c_float_p,
                                     _int, _int, _int, _int, _int, _stream]),
    "camli_pointconv_mix_fwd": (_int, [_c_float_p, _c_float_p, _c_i64_p, _int, _c_float_p,
                                       _int, _int, _int, _int, _int, _int, _stream]),
    "camli_pointconv_mix_bwd": (_int, [_c_float_p, _c_float_p, _c_float_p, _c_i64_p, _int, _c_float_p, _c_float_p,
                                       _int, _int, _int, _int, _int, _int, _stream]),
    "camli_corr3d_mlp_supported": (_int, [_int, _int, _int, _int]),
    "camli_corr3d_mlp_fwd": (_int, [_c_float_p] * 6 + [_int] * 5 + [_stream]),
    "camli_corr3d_mlp_bwd_workspace_bytes": (ctypes.c_int64, [_int, _int]),
    "camli_corr3d_mlp_bwd": (_int, [_c_float_p] * 12 + [_int] * 5 + [_stream]),
    "camli_convex_upsample_fwd": (_int, [_c_float_p, _c_float_p, _c_float_p, _c_float_p, _int, _int, _int, _int,
                                         ctypes.c_float, _stream]),
    "camli_convex_upsample_bwd": (_int, [_c_float_p, _c_float_p, _c_float_p, _c_float_p, _c_float_p, _c_float_p,
                                         _int, _int, _int, _int, ctypes.c_float, _stream]),
    "camli_convex_upsample_rows_fwd": (_int, [_c_float_p, _c_float_p, _c_float_p, _c_float_p, _int, _int, _int, _int, _int,
                                              ctypes.c_float, _stream]),
    "camli_convex_upsample_rows_bwd": (_int, [_c_float_p, _c_float_p, _c_float_p, _c_float_p, _c_float_p, _c_float_p,
                                              _int, _int, _int, _int, _int, ctypes.c_float, _stream]),
    "camli_gru_gates_fwd": (_int, [_c_float_p] * 6 + [_int, _int, _int, _stream]),
    "camli_gru_gates_bwd": (_int, [_c_float_p] * 7 + [_int, _int, _int, _stream]),
    "camli_gru_gates_bwd_strided": (_int, [_c_float_p, ctypes.c_int64, _c_float_p, ctypes.c_int64] + [_c_float_p] * 5 + [_int, _int, _int, _stream]),
    "camli_gru_blend_fwd": (_int, [_c_float_p] * 6 + [_int, _int, _int, _int, _stream]),
    "camli_gru_blend_bwd": (_int, [_c_float_p] * 7 + [_int, _int, _int, _int, _stream]),
    "camli_bias_act_mask_bytes": (ctypes.c_int64, [_int, _int, _int]),
    "camli_bias_act_fwd": (_int, [_c_float_p, _c_float_p, ctypes.c_void_p, _int, _int, _int, _int, _stream]),
    "camli_bias_act_nhwc_mask_bytes": (ctypes.c_int64, [ctypes.c_longlong, _int]),
    "camli_bias_act_nhwc_fwd": (_int, [_c_float_p, _c_float_p, _c_float_p, ctypes.c_void_p, ctypes.c_longlong, _int, _int, _stream]),
    "camli_bias_act_nhwc_bwd_workspace_bytes": (ctypes.c_int64, [ctypes.c_longlong, _int]),
    "camli_bias_act_nhwc_bwd": (_int, [_c_float_p, ctypes.c_void_p, _c_float_p, _c_float_p, _c_float_p, ctypes.c_longlong, _int, _int, _stream]),
    "camli_bias_act_res_fwd": (_int, [_c_float_p, _c_float_p, _c_float_p, ctypes.c_void_p, _int, _int, _int, _int, _stream]),
    "camli_bias_act_bwd": (_int, [_c_float_p, _c_float_p, ctypes.c_void_p, _c_float_p, _c_float_p, _int, _int, _int, _int, _stream]),
    "camli_bias_act_into_fwd": (_int, [_c_float_p, _c_float_p, ctypes.c_void_p, _c_float_p, ctypes.c_int64, _int, _int, _int, _int, _stream]),
    "camli_bias_act_bwd_strided": (_int, [_c_float_p, ctypes.c_int64, _c_float_p, ctypes.c_void_p, _c_float_p, _c_float_p, _int, _int, _int, _int, _stream]),
    "camli_weightnet_fwd": (_int, [_c_float_p, _c_float_p, _c_i64_p, _int] + [_c_float_p] * 7
                            + [_int, _int, _int, _int, _int, _int, _stream]),
    "camli_bilinear_sample_fwd": (_int, [_c_float_p, _c_float_p, _c_float_p, _int, _int, _int, _int, _int, _stream]),
    "camli_gather_scale_fwd": (_int, [_c_float_p, _c_float_p, _c_i64_p, _c_float_p, _int, _int, _int, _int, _stream]),
    "camli_masked_l2_fwd": (_int, [_c_float_p, _c_float_p, _int, _c_float_p, _int, _int, _int, _stream]),
    "camli_masked_l2_bwd": (_int, [_c_float_p, _c_float_p, _int, _c_float_p, _c_float_p, _int, _int, _int, _stream]),
    "camli_ids_flow_fwd": (_int, [_c_float_p] * 7 + [ctypes.c_float] * 5 + [_int, _int, _stream]),
    "camli_ids_flow_bwd": (_int, [_c_float_p] * 7 + [ctypes.c_float] * 5 + [_int, _int, _stream]),
    "camli_persp2paral": (_int, [_c_float_p, _c_float_p, _c_float_p, _c_float_p, _int, _int] + [ctypes.c_float] * 5 + [_stream]),
    "camli_project_pc2image": (_int, [_c_float_p, _c_float_p, _c_float_p, _int, _int, _int] + [ctypes.c_float] * 4 + [_stream]),
    "camli_convcl_fwd": (_int, [_c_float_p, _int, _int, _c_float_p, _int, _int, _c_float_p, _c_float_p, _int, _int, _c_float_p, _int,
                                _int, _int, _int, _int, _int, ctypes.c_char_p, ctypes.c_char_p, _int, _int, _stream]),
    "camli_convcl_gru_gates": (_int, [_c_float_p, _c_float_p, _int, _c_float_p, _c_float_p, _c_float_p, _c_float_p, _c_float_p,
                                      _int, _int, _int, _int, ctypes.c_char_p, ctypes.c_char_p, _stream]),
    "camli_convcl_gru_blend": (_int, [_c_float_p, _c_float_p, _int, _c_float_p, _c_float_p, _c_float_p, _c_float_p, _c_float_p,
                                      _c_float_p, _int, _int, _int, _int, _int, ctypes.c_char_p, ctypes.c_char_p, _stream]),
    "camli_gru_gates_bwd_into": (_int, [_c_float_p, ctypes.c_int64, _c_float_p, ctypes.c_int64, _c_float_p, _c_float_p, _c_float_p,
                                        _c_float_p, _c_float_p, _c_float_p, _int, _int, _int, _stream]),
    "camli_gru_blend_bwd_acc": (_int, [_c_float_p] * 8 + [_int, _int, _int, _int, _stream]),
    "camli_wino1d_workspace_bytes": (ctypes.c_int64, [_int] * 6),
    "camli_wino1d_weights": (_int, [_c_float_p, _c_float_p, _int, _int, _int, _stream]),
    "camli_wino1d_conv": (_int, [_c_float_p, _int, _int, _c_float_p, _int, _int, _c_float_p, _c_float_p, _int, _int, _c_float_p, _int,
                                 _c_float_p, ctypes.c_int64] + [_int] * 7 + [_stream]),
    "camli_wino1d_gru_gates": (_int, [_c_float_p, _c_float_p, _int, _c_float_p, _c_float_p, _c_float_p, _c_float_p, _c_float_p,
                                      _c_float_p, _c_float_p, ctypes.c_int64] + [_int] * 4 + [_stream]),
    "camli_wino1d_gru_blend": (_int, [_c_float_p, _c_float_p, _int, _c_float_p, _c_float_p, _c_float_p, _c_float_p, _c_float_p,
                                      _c_float_p, _int, _c_float_p, _c_float_p, ctypes.c_int64] + [_int] * 4 + [_stream]),
    "camli_wino1d_wrw_workspace_bytes": (ctypes.c_int64, [_int] * 6),
    "camli_wino1d_wrw_reuse": (_int, [_int] * 6),
    "camli_wino1d_wrw": (_int, [_c_float_p, _int, _int, _c_float_p, _int, _int, _c_float_p, _int, _c_float_p, _c_float_p, _c_float_p, ctypes.c_int64]
                         + [_int] * 6 + [_stream]),
    "camli_convcl_wrw_workspace_bytes": (ctypes.c_int64, [_int] * 6),
    "camli_convcl_wrw": (_int, [_c_float_p, _int, _int, _c_float_p, _int, _int, _c_float_p, _int, _c_float_p, ctypes.c_int64,
                                _c_float_p, _int, _int, _int, _int, _int, _int, ctypes.c_char_p, ctypes.c_char_p, _stream]),
    "camli_wino_weight_floats": (ctypes.c_int64, [_int, _int, _int]),
    "camli_wino_weights": (_int, [_c_float_p, _c_float_p, _int, _int, _int, _int, _stream]),
    "camli_wino_workspace_bytes": (ctypes.c_int64, [_int] * 6),
    "camli_wino_mask_bytes": (ctypes.c_int64, [_int] * 4),
    "camli_wino_conv3x3": (_int, [_c_float_p, ctypes.c_int64, ctypes.c_void_p, _c_float_p, _c_float_p, _c_float_p,
                                  ctypes.c_int64, ctypes.c_void_p, _c_float_p, ctypes.c_int64] + [_int] * 8 + [_stream]),
    "camli_wino_wrw_workspace_bytes": (ctypes.c_int64, [_int] * 6),
    "camli_wino_wrw": (_int, [_c_float_p, ctypes.c_int64, _c_float_p, ctypes.c_int64, ctypes.c_void_p, _c_float_p, _c_float_p,
                              _c_float_p, ctypes.c_int64] + [_int] * 8 + [_stream]),
    "camli_conv3x3_co2_fwd": (_int, [_c_float_p] * 4 + [_int] * 4 + [_stream]),
    "camli_conv3x3_co2_bwd_data": (_int, [_c_float_p] * 3 + [_int] * 4 + [_stream]),
    "camli_conv3x3_co2_bwd_weight_workspace_bytes": (ctypes.c_longlong, [_int, _int, _int]),
    "camli_conv3x3_co2_bwd_weight": (_int, [_c_float_p] * 5 + [_int] * 5 + [_stream]),
    "camli_maxpool3x3s2_fwd": (_int, [_c_float_p, _c_float_p, ctypes.c_void_p] + [_int] * 5 + [_stream]),
    "camli_maxpool3x3s2_bwd": (_int, [_c_float_p, ctypes.c_void_p, _c_float_p] + [_int] * 5 + [_stream]),
    "camli_pad_normalize": (_int, [_c_float_p, _c_float_p, _c_float_p, _int, _int, _int, _int, _int, _int,
                                   ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), _stream]),
    "camli_sk_gate_fwd": (_int, [_c_float_p] * 6 + [_int, _int, _int, _stream]),
    "camli_sk_gate_bwd": (_int, [_c_float_p] * 10 + [_int, _int, _int, _stream]),
    "camli_sk_pool_fwd": (_int, [_c_float_p] * 3 + [_int, _int, _int, _stream]),
    "camli_sk_mix_fwd": (_int, [_c_float_p] * 4 + [_int, _int, _int, _stream]),
    "camli_sk_mix_bwd_w": (_int, [_c_float_p] * 4 + [_int, _int, _int, _stream]),
    "camli_sk_mix_bwd_x": (_int, [_c_float_p] * 5 + [_int, _int, _int, _stream]),
    "camli_weightnet_bwd_workspace_bytes": (ctypes.c_int64, [_int]),
    "camli_weightnet_bwd": (_int, [_c_float_p, _c_float_p, _c_i64_p, _int] + [_c_float_p] * 14
                            + [ctypes.c_int64, _int, _int, _int, _int, _int, _int, _stream]),
}

_lib = None


class CamliHipError(RuntimeError):
    pass


def load():
    """Load the shared library once.  Raises if it has not been built: there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CamliHipError(
            "camliflow_amd: %s is missing -- build it with `python -m camliflow_amd.csrc.build` "
            "(or __graft_entry__.build()).  The HIP library is mandatory; there is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError here == header / library mismatch
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(code, what):
    if code != 0:
        msg = load().camli_last_error_string()
        raise CamliHipError("%s failed (%d): %s" % (what, code, msg.decode() if msg else "?"))


# ------------------------------------------------------------------------------------------------
# optional per-kernel timing (HIP events on the launching stream); used by bench.py's roofline leg
# ------------------------------------------------------------------------------------------------
class KernelTimer:
    """Records a (start, end) event pair around every launch routed through ``launch()`` while
    enabled, together with the launch's algorithmic work (bytes for the HBM-bound kernels, pairs /
    point-updates for KNN / FPS).  Events go on torch's current stream, which is the stream the
    kernels are launched on."""

    def __init__(self):
        self.enabled = False
        self.only = None        # optional set of entry-point names to restrict timing to
        self.records = {}
        self.order = []         # entry-point names in launch order (tools/kernel_bench.py pairs them with a kineto trace)
        self._pool = []         # recycled events: creating one costs as much as recording it

    def reset(self):
        self.order = []
        for recs in self.records.values():
            for r in recs:
                self._pool.append(r[0])
                self._pool.append(r[1])
        self.records = {}

    def event(self):
        if self._pool:
            return self._pool.pop()
        import torch
        return torch.cuda.Event(enable_timing=True)

    def summary(self):
        """name -> dict(launches, total_ms, work, unit).  Call after torch.cuda.synchronize()."""
        out = {}
        for name, recs in self.records.items():
            out[name] = {'launches': len(recs), 'total_ms': sum(r[0].elapsed_time(r[1]) for r in recs),
                         'work': float(sum(r[2] for r in recs)), 'unit': recs[0][3],
                         'flop': float(sum(r[4] for r in recs))}
        return out


TIMER = KernelTimer()


_CENSUS = None      # set by cores.runtime when the census is on (avoids an import cycle here)


def launch(name, fn, *args, work=None, flop=0.0):
    """Call a C-ABI entry point, check its status, optionally time it.  ``work`` = (amount, unit) of
    algorithmic work of this launch, ``flop`` its floating-point work for the
    matrix-core kernels; both only evaluated bookkeeping-wise."""
    if _CENSUS is not None:
        _CENSUS[name] += 1
    if TIMER.enabled and (TIMER.only is None or name in TIMER.only):
        start, end = TIMER.event(), TIMER.event()
        start.record()
        code = fn(*args)
        end.record()
        amount, unit = work if work is not None else (0.0, 'B')
        TIMER.records.setdefault(name, []).append((start, end, amount, unit, flop))
        TIMER.order.append(name)
    else:
        code = fn(*args)
    check(code, name)
