"""ctypes binding of liblp_hip.so (C ABI declared in include/lp_hip.h).

There is deliberately NO fallback: if the library is missing or a call fails, an exception is raised."""
import ctypes
import os

import torch  # noqa: F401  -- MUST precede CDLL: torch bundles its own libamdhip64; loading it first makes liblp_hip.so bind
#                              to the same HIP runtime instance (otherwise launches fail with "no ROCm-capable device")

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('LP_LIB_OVERRIDE') or os.path.join(_HERE, 'liblp_hip.so')      # override: ablation builds (probes/)

PREC_BF16 = 0
PREC_BF16X3 = 1
PREC_F16 = 2

_vp, _i, _f, _ll = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_longlong

# name -> (restype, argtypes); must list every symbol of include/lp_hip.h (tests/test_abi.py checks this)
SIGNATURES = {
    'lp_last_error': (ctypes.c_char_p, []),
    'lp_abi_version': (_i, []),
    'lp_pack_weights': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'lp_pack_desc_bytes': (_i, []),
    'lp_pack_weights_batch': (_i, [_vp, _i, _ll, _vp]),
    'lp_pack_pair_desc_bytes': (_i, []),
    'lp_pack_weights_pairs': (_i, [_vp, _i, _ll, _vp]),
    'lp_act_pack': (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _vp, _vp]),
    'lp_amax_slots': (_i, []),
    'lp_amax_slot_stride': (_i, []),
    'lp_amax_blocks': (_i, []),
    'lp_amax_partial': (_i, [_vp, _ll, _vp, _vp]),
    'lp_conv16_fwd': (_i, [_vp] * 9 + [_i] * 11 + [_vp, _vp, _vp, _i, _vp, _ll, _vp, _vp]),
    'lp_conv16_fwd_workspace_bytes': (_ll, [_i] * 5),
    'lp_conv16_fwd_stats': (_i, [_vp] * 9 + [_i] * 11 + [_vp, _vp, _vp, _i, _vp, _ll, _vp, _vp, _ll, _vp, _vp]),
    'lp_conv16_stats_floats': (_ll, [_i] * 4),
    'lp_norm_stats_finalize': (_i, [_vp, _i, _vp, _vp, _i, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    'lp_gconv16_fwd_stats': (_i, [_vp] * 6 + [_i] * 6 + [_vp, _vp, _ll, _vp, _vp]),
    'lp_gconv16_fwd_planes': (_i, [_vp] * 8 + [_i] * 7 + [_vp, _vp, _ll, _vp, _vp]),
    'lp_conv_wgrad_workspace_bytes': (_ll, [_i, _i, _i, _i]),
    'lp_conv16_wgrad': (_i, [_vp] * 6 + [_i] * 9 + [_vp, _i, _vp, _vp, _vp, _vp]),
    'lp_conv_wgrad_dot_blocks': (_i, [_i] * 3),
    'lp_thin_conv_supported': (_i, [_i] * 4),
    'lp_thin_conv_fwd': (_i, [_vp] * 6 + [_i] * 9 + [_vp, _i, _vp]),
    'lp_thin_conv_emits_planes': (_i, [_i] * 5),
    'lp_thin_wgrad_supported': (_i, [_i] * 5),
    'lp_thin_wgrad_has_dbias': (_i, [_i] * 2),
    'lp_thin_wgrad': (_i, [_vp] * 6 + [_i] * 8 + [_vp, _vp]),
    'lp_linear_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'lp_linear_bwd_workspace_bytes': (_ll, [_i, _i, _i]),
    'lp_linear_bwd': (_i, [_vp] * 8 + [_i, _i, _i, _vp]),
    'lp_grid_crop_fwd': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'lp_grid_crop_bwd': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'lp_stem_conv_s2': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'lp_dwconv3x3_fwd': (_i, [_vp] * 5 + [_i] * 5 + [_vp]),
    'lp_affine_res': (_i, [_vp] * 7 + [_ll, _i, _i, _vp]),
    'lp_affine_relu6_mean': (_i, [_vp] * 4 + [_i, _i, _i, _vp]),
    'lp_bn_stats_workspace_bytes': (_ll, [_ll, _i]),
    'lp_pwconv_stat_rows': (_i, [_ll, _i, _i]),
    'lp_pwconv_fwd': (_i, [_vp] * 5 + [_i, _vp, _vp, _vp, _ll, _i, _i, _vp]),
    'lp_dwconv_stat_rows': (_i, [_i] * 5),
    'lp_dwconv3x3_stats_fwd': (_i, [_vp] * 6 + [_i] * 5 + [_vp]),
    'lp_bn_finalize': (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _f, _f, _vp]),
    'lp_bn_stats': (_i, [_vp] * 8 + [_ll, _i, _f, _f, _vp]),
    'lp_instnorm_workspace_bytes': (_ll, [_i, _i, _i]),
    'lp_instnorm_stats': (_i, [_vp, _vp, _vp, _i, _f, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'lp_adain_bwd_workspace_bytes': (_ll, [_i, _i, _i]),
    'lp_adain_relu_bwd': (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    'lp_norm_act_bwd': (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _f, _i, _vp, _vp]),
    'lp_bn_train_stats_workspace_bytes': (_ll, [_ll, _i]),
    'lp_bn_train_stats': (_i, [_vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _i, _vp]),
    'lp_gconv16_fwd': (_i, [_vp] * 6 + [_i] * 6 + [_vp, _vp]),
    'lp_gconv_wgrad_workspace_bytes': (_ll, [_i, _i]),
    'lp_gconv16_wgrad': (_i, [_vp] * 6 + [_i] * 7 + [_vp, _vp]),
    'lp_pack_grouped': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'lp_im2col_planes': (_i, [_vp, _vp, _vp] + [_i] * 8 + [_vp]),
    'lp_bn_relu_maxpool_fwd': (_i, [_vp] * 7 + [_i] * 5 + [_vp]),
    'lp_maxpool_bwd': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'lp_bn_add_act': (_i, [_vp] * 9 + [_ll, _i, _i, _i, _vp]),
    'lp_bn_add_act_planes': (_i, [_vp] * 8 + [_ll, _i, _i, _i, _vp]),
    'lp_bn_add_act16': (_i, [_vp] * 8 + [_ll, _i, _i, _vp]),
    'lp_bn_act16': (_i, [_vp] * 4 + [_ll, _i, _i, _vp]),
    'lp_adain_act16': (_i, [_vp] * 4 + [_i, _ll, _i, _i, _vp]),
    'lp_adain_relu_bwd16': (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    'lp_thin_wgrad16': (_i, [_vp] * 6 + [_i] * 8 + [_vp]),
    'lp_subsample2': (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    'lp_zero_stuff2': (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    'lp_add_strided2': (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    'lp_bn_bwd16_workspace_bytes': (_ll, [_ll, _i]),
    'lp_bn_bwd16': (_i, [_vp] * 14 + [_ll, _i, _i, _f, _i, _i, _vp, _vp]),
    'lp_bn_bwd16_h': (_i, [_vp] * 15 + [_ll, _i, _i, _f, _i, _i, _vp, _vp]),
    'lp_spatial_mean_fwd': (_i, [_vp, _vp, _i, _i, _i, _vp]),
    'lp_spatial_mean_bwd': (_i, [_vp, _vp, _i, _i, _i, _vp]),
    'lp_dwconv3x3_dgrad': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'lp_dwconv3x3_wgrad_workspace_bytes': (_ll, [_i]),
    'lp_dwconv3x3_wgrad': (_i, [_vp] * 6 + [_i] * 5 + [_vp]),
    'lp_sum2x2': (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    'lp_sum2x2_planes': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'lp_proj_score_fwd': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'lp_proj_score_bwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'lp_image_prep_fwd': (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp]),
    'lp_image_prep_bwd': (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    'lp_pool_grad_pack': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _i, _vp, _vp]),
    'lp_adain_relu_bwd_planes': (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp]),
    'lp_head_fwd': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'lp_head_bwd': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    'lp_relu_bwd': (_i, [_vp, _vp, _vp, _ll, _vp]),
    'lp_avgpool2_fwd': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    'lp_avgpool2_fwd16': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'lp_avgpool2_bwd': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    'lp_l1_partial_blocks': (_i, []),
    'lp_l1_fwd': (_i, [_vp, _vp, _vp, _ll, _i, _f, _vp, _vp, _vp]),
    'lp_l1_fwd_b16': (_i, [_vp, _vp, _i, _vp, _ll, _i, _f, _vp, _vp, _vp]),
    'lp_l1_fwd_ab16': (_i, [_vp, _vp, _i, _vp, _ll, _f, _vp, _vp, _vp]),
    'lp_avgpool2_bwd_m16': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    'lp_reflect_border_fwd': (_i, [_vp, _vp] + [_i] * 7 + [_vp, _i, _vp, _vp, _vp]),
    'lp_reflect_border_dgrad': (_i, [_vp] + [_i] * 4 + [_vp, _i, _vp, _vp, _i, _vp, _vp]),
    'lp_reflect_border_wgrad': (_i, [_vp, _vp] + [_i] * 7 + [_vp, _i, _vp, _vp, _vp]),
    'lp_reflect_border_wgrad_workspace_bytes': (_ll, [_i] * 5),
    'lp_l1_bwd': (_i, [_vp, _vp, _vp, _f, _vp, _vp, _ll, _i, _vp, _vp, _vp]),
    'lp_dice_partial_blocks': (_i, []),
    'lp_reduce_dice': (_i, [_vp] * 5 + [_i] * 4 + [_f, _vp]),
    'lp_reduce_dice_bwd': (_i, [_vp] * 5 + [_i] * 4 + [_f, _vp]),
    'lp_reduce_hinge': (_i, [_vp, _vp, _vp, _vp, _i, _vp]),
    'lp_reduce_hinge_bwd': (_i, [_vp] * 7 + [_i, _vp]),
    'lp_mt_desc_bytes': (_i, []),
    'lp_mt_optimizer_step': (_i, [_vp, _i, _ll, _vp, _i, _f, _f, _f, _f, _vp]),
    'lp_mt_ema': (_i, [_vp, _i, _ll, _f, _i, _vp]),
    'lp_sn_desc_bytes': (_i, []),
    'lp_sn_power_iter': (_i, [_vp, _i, _i, _i, _i, _vp]),
    'lp_sn_row_block': (_i, []),
    'lp_sn_grad_apply': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _vp]),
    'lp_sn_apply_desc_bytes': (_i, []),
    'lp_sn_grad_apply_batch': (_i, [_vp, _i, _vp]),
    'lp_sn_embed_grad': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
}

_lib = None


class LpError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LpError(f'{LIB_PATH} is missing: run `python -m latent_pose_reenactment_amd.build` '
                          '(or __graft_entry__.build()); there is no fallback path')
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().lp_last_error()
        raise LpError(f'{what} failed (rc={rc}): {msg.decode() if msg else ""}')
