"""ctypes binding of libsvgpvae_hip.so (include/svgpvae_hip.h).

There is NO CPU fallback: if the shared object is missing or does not export a symbol the
header declares, loading fails loudly.  PyTorch only supplies device memory / streams."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SVGP_LIB_PATH: load another build of the SAME library (the host-ASan build of `make asan`); never a fallback
LIB_PATH = os.environ.get("SVGP_LIB_PATH") or os.path.join(_HERE, "libsvgpvae_hip.so")


class SvgpError(RuntimeError):
    pass


class MnistCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("b", "b_global", "m", "L", "M", "n_obj", "normalize_obj", "clip_qs",
                                         "geco", "train_ip", "train_gp", "train_ov", "b_cap", "clip_pv", "n_pix",
                                         "titsias", "kl_form", "single_stat_block", "gemm_f32",
                                         "split_grad_exchange")] + \
               [(n, C.c_double) for n in ("N_train", "jitter", "kappa_squared", "alpha", "rep_weight")]


PARAM_FIELDS = ("enc_c1_w", "enc_c1_b", "enc_c2_w", "enc_c2_b", "enc_c3_w", "enc_c3_b", "enc_d_w", "enc_d_b",
                "dec_d_w", "dec_d_b", "dec_c1_w", "dec_c1_b", "dec_c2_w", "dec_c2_b", "dec_c3_w", "dec_c3_b",
                "ip", "l_GP", "amplitude", "ov", "n_enc", "n_vae", "n_total")


class ParamLayout(C.Structure):
    _fields_ = [(n, C.c_int64) for n in PARAM_FIELDS]


WS_FIELDS = ("enc_a1", "enc_a2", "enc_a3", "qnet_mu", "qnet_var_raw", "qnet_var", "K", "Kn", "knn",
             "statA", "statA_len", "S", "v", "stat_parts",
             "Ki", "ldK", "Si", "t", "G", "A", "Aji", "mu_hat", "u", "M2", "KL", "q",
             "p_m", "p_v", "e", "d", "eps", "z",
             "dec_h0", "dec_a1", "dec_a2", "recon", "dec_d2", "dec_d1", "dec_dh0", "flags", "dec_weff",
             "zbar", "g_pv", "g_pm", "mvbar",
             "statB", "statB_len", "A2", "ud", "td",
             "Kbar", "fb_part", "Qm", "vbar", "Ssym", "Knbar_part",
             "scr_bm", "scr_mm", "scr_vec", "scr_inv", "scr_bl", "scr_sm",
             "Knbar", "knnbar", "ybar", "s2bar", "d_on",
             "part_dec", "part_enc", "n_part", "part_gp", "part_sums", "n_post",
             "gradC", "gradC_len", "grad", "sums",
             "tit_S2", "tit_v2", "tit_Si", "tit_t", "tit_scal", "xpack", "xpack_len", "total")


class WsLayout(C.Structure):
    _fields_ = [(n, C.c_int64) for n in WS_FIELDS]


class SpritesKcfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("b", "m", "La", "Lc", "n_act", "normalize", "k_se")] + [("rep_weight", C.c_double)]


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n", "Hi", "Wi", "Ci", "Ho", "Wo", "Co", "Hs", "Ws", "sy", "sx", "osy", "osx",
                                         "ooy", "oox", "nt", "act")] + \
               [("oy", C.c_int32 * 16), ("ox", C.c_int32 * 16), ("woff", C.c_int32 * 16)]


class SumJob(C.Structure):
    """svgp_sum_job (include/svgpvae_hip.h): one deferred partial-sum reduction."""
    _fields_ = [("part", C.c_void_p), ("out", C.c_void_p)] + [(n, C.c_int32) for n in ("ng", "len", "stride", "accumulate")]


class PearceBufs(C.Structure):
    """svgp_pearce_bufs (include/svgpvae_hip.h)."""
    _fields_ = [("B", C.c_int32), ("T", C.c_int32), ("n", C.c_int32)] + \
               [(n, C.c_void_p) for n in ("times", "idx", "tmask", "ls_x", "ls_y", "y_x", "y_y", "s2_x", "s2_y", "p_m_x",
                                          "p_m_y", "p_v_x", "p_v_y", "eps_x", "eps_y", "z_x", "z_y", "zbar_x", "zbar_y",
                                          "ybar_x", "ybar_y", "s2bar_x", "s2bar_y", "Ai", "alpha", "lh", "ce", "row_ce",
                                          "dl_part")]


STATE = dict(C_MA=0, LAGRANGE=1, ALPHA=2, ADAM_T=3, LR=4, BETA=5, ELBO=6, RECON_LOSS=7, KL_TERM=8,
             INSIDE_ELBO=9, CE_TERM=10, INSIDE_RECON=11, INSIDE_KL=12, RNG_CTR=13)
STATE_LEN = 16

_P = C.c_void_p
_CFG = C.POINTER(MnistCfg)

# name -> argtypes ; every function returns int (status) except the two noted below
SIGNATURES = {
    "svgp_mnist_param_layout_get": [_CFG, C.POINTER(ParamLayout)],
    "svgp_mnist_ws_layout_get": [_CFG, C.POINTER(WsLayout)],
    "svgp_mnist_encoder_fwd": [_CFG, _P, _P, _P, _P],
    "svgp_kernel_matrix_fwd": [_CFG, _P, _P, _P, _P],
    "svgp_kernel_matrix_xy": [C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, _P, C.c_int, _P, _P, _P, C.c_int, _P, _P],
    "svgp_gp_stats_fwd": [_CFG, _P, _P],
    "svgp_gp_factor_fwd": [_CFG, _P, _P],
    "svgp_gp_posterior_fwd": [_CFG, _P, _P, _P, _P],
    "svgp_gp_posterior_fwd_with_aji": [_CFG, _P, _P, _P, _P],
    "svgp_mnist_decoder_fwd": [_CFG, _P, _P, _P, _P],
    "svgp_mnist_decoder_bwd": [_CFG, _P, _P, _P, _P, _P],
    "svgp_mnist_decoder_bwd_data": [_CFG, _P, _P, _P, _P, _P],
    "svgp_mnist_decoder_fwd_pre": [_CFG, _P, _P, _P, _P],
    "svgp_mnist_decoder_bwd_data_pre": [_CFG, _P, _P, _P, _P, _P],
    "svgp_mnist_decoder_bwd_weights": [_CFG, _P, _P, _P, C.c_int, C.c_int, _P],
    "svgp_gp_stats_bwd": [_CFG, _P, _P, _P],
    "svgp_gp_factor_bwd": [_CFG, _P, _P, _P],
    "svgp_gp_posterior_bwd": [_CFG, _P, _P, _P],
    "svgp_gp_factor_fwd_defer_aji": [_CFG, _P, _P],
    "svgp_gp_factor_fwd_aji_tail": [_CFG, _P, _P],
    "svgp_gp_factor_bwd_early": [_CFG, _P, _P, _P],
    "svgp_gp_factor_bwd_late": [_CFG, _P, _P, _P],
    "svgp_gp_factor_bwd_early_a": [_CFG, _P, _P, _P],
    "svgp_gp_factor_bwd_early_b": [_CFG, _P, _P, _P],
    "svgp_gp_factor_bwd_late_a": [_CFG, _P, _P, _P],
    "svgp_gp_factor_bwd_late_b": [_CFG, _P, _P, _P],
    "svgp_gp_factor_bwd_late_b_channels": [_CFG, _P, _P, _P],
    "svgp_gp_factor_bwd_late_b_kbar": [_CFG, _P, _P, _P],
    "svgp_gp_factor_bwd_late_b_final": [_CFG, _P, _P, _P],
    "svgp_mnist_encoder_kernel_matrix_fwd": [_CFG, _P, _P, _P, _P, _P],
    "svgp_gp_stats_bwd_with_aji": [_CFG, _P, _P, _P],
    "svgp_gp_factor_bwd_nofinal": [_CFG, _P, _P, _P],
    "svgp_gp_factor_bwd_nofinal_wgrad": [_CFG, _P, _P, _P, _P],
    "svgp_gp_stats_factor_bwd_wgrad": [_CFG, _P, _P, _P, _P],
    "svgp_mnist_encoder_bwd_km": [_CFG, _P, _P, _P, _P, _P],
    "svgp_mnist_encoder_bwd_km_sum": [_CFG, _P, _P, _P, _P, _P, _P],
    "svgp_mnist_encoder_bwd_km_regs": [C.POINTER(C.c_int)],
    "svgp_mnist_decoder_bwd_data_pre_aji": [_CFG, _P, _P, _P, _P, _P],
    "svgp_mnist_decoder_bwd_data_aji_regs": [C.POINTER(C.c_int)],
    "svgp_gp_posterior_bwd_rows": [_CFG, _P, _P, _P],
    "svgp_mnist_grad_reduce_part": [_CFG, _P, _P, C.c_int, _P],
    "svgp_gp_posterior_bwd_with_final": [_CFG, _P, _P, _P],
    "svgp_gp_titsias_stats": [_CFG, _P, _P],
    "svgp_gp_titsias_fwd": [_CFG, _P, _P, _P],
    "svgp_gp_titsias_bwd": [_CFG, _P, _P, _P],
    "svgp_kernel_matrix_bwd": [_CFG, _P, _P, _P, _P],
    "svgp_mnist_encoder_bwd": [_CFG, _P, _P, _P, _P],
    "svgp_mnist_grad_reduce": [_CFG, _P, _P],
    "svgp_kernel_matrix_bwd_partials": [_CFG, _P, _P, _P, _P],
    "svgp_mnist_grad_reduce_all": [_CFG, _P, _P, _P],
    "svgp_adam_tf1_step": [C.c_int64, _P, _P, _P, _P, _P, C.c_double, C.c_double, C.c_double, _P],
    "svgp_elbo_finalize": [_CFG, _P, _P, _P],
    "svgp_adam_tf1_finalize": [_CFG, C.c_int64, _P, _P, _P, _P, _P, _P, C.c_double, C.c_double, C.c_double, _P],
    "svgp_elbo_finalize_noadam": [_CFG, _P, _P, _P],
    "svgp_mnist_step_phase": [_CFG, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "svgp_mnist_train_step": [_CFG, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "svgp_stream_features_f32": [_P, C.c_int64, _P, C.c_int, C.c_int, _P, _P, _P],
    "svgp_stream_knm_f32": [_P, C.c_int64, C.c_int, _P, _P, _P, _P],
    "svgp_stream_stats_f32": [C.c_int64, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, C.c_int64, _P],
    "svgp_comm_unique_id": [_P, C.c_int],
    "svgp_comm_init": [_P, C.c_int, C.c_int, C.c_int, C.POINTER(_P)],
    "svgp_comm_destroy": [_P],
    "svgp_allreduce_sum_f64": [_P, _P, C.c_int64, _P],
    "svgp_allreduce_sum_f32": [_P, _P, C.c_int64, _P],
    "svgp_reduce_scatter_sum_f64": [_P, _P, C.c_int64, _P],
    "svgp_allgather_f64": [_P, _P, C.c_int64, _P],
    "svgp_gp_factor_fwd_channels": [_CFG, C.c_int, C.c_int, _P, _P],
    "svgp_gp_factor_bwd_channels": [_CFG, C.c_int, C.c_int, _P, _P, _P],
    "svgp_gp_factor_fwd_channels_part": [_CFG, C.c_int, C.c_int, C.c_int, _P, _P],
    "svgp_gp_factor_bwd_channels_part": [_CFG, C.c_int, C.c_int, C.c_int, _P, _P, _P],
    "svgp_mnist_train_step_dp": [_CFG, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "svgp_comm_group_begin": [_P],
    "svgp_comm_group_end": [_P],
    "svgp_comm_timing": [_P, C.c_int],
    "svgp_comm_timing_read": [_P, _P, C.c_int, C.POINTER(C.c_int)],
    "svgp_sym_pack": [C.c_int, C.c_int, C.c_int, _P, _P, _P],
    "svgp_sym_unpack": [C.c_int, C.c_int, _P, _P, _P],
    "svgp_dgemm_batched": [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, _P, C.c_int, C.c_longlong, _P,
                           C.c_int, C.c_longlong, C.c_double, _P, C.c_int, C.c_longlong, C.c_int, _P],
    "svgp_dgemm_f32c_batched": [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, _P, C.c_int, C.c_longlong, _P,
                                C.c_int, C.c_longlong, C.c_double, _P, C.c_int, C.c_longlong, C.c_int, _P],
    "svgp_sgemm_batched": [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _P, C.c_int, C.c_longlong, _P,
                           C.c_int, C.c_longlong, C.c_float, _P, C.c_int, C.c_longlong, C.c_int, _P],
    "svgp_spd_inverse_batched": [C.c_int, C.c_int, _P, _P, _P, _P],
    "svgp_potrf_batched": [C.c_int, C.c_int, _P, C.c_int, C.c_longlong, _P, _P, _P],
    "svgp_trsm_batched": [C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_longlong, _P, C.c_int, C.c_longlong,
                          C.c_int, _P, _P],
    "svgp_potri_batched": [C.c_int, C.c_int, _P, _P, _P, _P],
    "svgp_lu_inverse": [C.c_int, _P, _P, _P, _P],
    "svgp_dgemm_splitk": [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, _P, C.c_int, _P, C.c_int, C.c_double, _P,
                          C.c_int, _P, C.c_longlong, _P],
    "svgp_conv_taps_fwd": [C.POINTER(ConvDesc), C.c_int, _P, _P, _P, _P, _P],
    "svgp_conv_taps_wgrad": [C.POINTER(ConvDesc), C.c_int, _P, _P, _P, C.c_int, C.c_int, _P, C.c_int, _P],
    "svgp_elu_bwd_bias": [C.c_longlong, C.c_int, _P, _P, _P, _P, _P],
    "svgp_upconv_weights": [C.c_int, C.c_int, _P, _P, _P],
    "svgp_upconv_fold_wgrad": [C.c_int, C.c_int, _P, _P, _P],
    "svgp_conv_taps_wgrad_fused": [C.POINTER(ConvDesc), C.c_int, _P, _P, _P, _P, _P, C.c_int, C.c_int, _P, _P, _P],
    "svgp_conv_taps_wgrad_fused_f32": [C.POINTER(ConvDesc), C.c_int, _P, _P, _P, _P, _P, C.c_int, C.c_int, _P, _P, _P],
    "svgp_conv_taps_fwd_f32": [C.POINTER(ConvDesc), C.c_int, _P, _P, _P, _P, _P],
    "svgp_conv_taps_wgrad_f32": [C.POINTER(ConvDesc), C.c_int, _P, _P, _P, C.c_int, C.c_int, _P, C.c_int, _P],
    "svgp_elu_bwd_bias_f32": [C.c_longlong, C.c_int, _P, _P, _P, _P, _P],
    "svgp_upconv_weights_f32": [C.c_int, C.c_int, _P, _P, _P],
    "svgp_upconv_fold_wgrad_f32": [C.c_int, C.c_int, _P, _P, _P],
    "svgp_avgpool_fwd_f32": [C.c_int, C.c_int, C.c_int, _P, _P, _P],
    "svgp_avgpool_bwd_f32": [C.c_int, C.c_int, C.c_int, _P, _P, _P],
    "svgp_bias_add_f32": [C.c_longlong, C.c_int, _P, _P, _P],
    "svgp_sqerr_fwd_f32": [C.c_longlong, C.c_int, _P, _P, _P, _P],
    "svgp_sqerr_bwd_f32": [C.c_longlong, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P],
    "svgp_sprites_kernel_matrix_fwd": [C.POINTER(SpritesKcfg), _P, _P, _P, _P, _P, _P, _P, _P],
    "svgp_sprites_kernel_matrix_bwd": [C.POINTER(SpritesKcfg)] + [_P] * 12 + [C.c_longlong, _P],
    "svgp_sprites_aux_fwd": [C.c_int, C.c_int, C.c_int, _P, _P, _P, _P],
    "svgp_sprites_aux_bwd": [C.c_int, C.c_int, C.c_int, _P, _P, _P],
    "svgp_avgpool_fwd": [C.c_int, C.c_int, C.c_int, _P, _P, _P],
    "svgp_avgpool_bwd": [C.c_int, C.c_int, C.c_int, _P, _P, _P],
    "svgp_enc_head_fwd": [C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P],
    "svgp_enc_head_bwd": [C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P],
    "svgp_softmax_xent": [C.c_int, C.c_int, _P, _P, _P, _P, _P, _P],
    "svgp_bias_add": [C.c_longlong, C.c_int, _P, _P, _P],
    "svgp_conv_taps_wgrad_fused_jobs": [C.POINTER(ConvDesc), C.c_int, _P, _P, _P, _P, _P, C.c_int, C.c_int, _P, _P,
                                        C.POINTER(SumJob), C.c_int, C.POINTER(C.c_int), _P],
    "svgp_conv_taps_wgrad_fused_jobs_f32": [C.POINTER(ConvDesc), C.c_int, _P, _P, _P, _P, _P, C.c_int, C.c_int, _P, _P,
                                            C.POINTER(SumJob), C.c_int, C.POINTER(C.c_int), _P],
    "svgp_sum_partials_multi": [C.POINTER(SumJob), C.c_int, _P],
    "svgp_sum_partials_multi_f32": [C.POINTER(SumJob), C.c_int, _P],
    "svgp_transpose_taps": [C.c_int, C.c_int, C.c_int, _P, _P, _P],
    "svgp_transpose_taps_f32": [C.c_int, C.c_int, C.c_int, _P, _P, _P],
    "svgp_cast_f64_f32": [C.c_longlong, _P, _P, _P],
    "svgp_cast_f32_f64": [C.c_longlong, _P, _P, _P],
    "svgp_streams_overlap": [_P, _P, C.POINTER(C.c_int)],
    "svgp_side_streams_prepare": [_P],
    "svgp_gauss_cross_entropy": [C.c_longlong, _P, _P, _P, _P, _P, _P],
    "svgp_sqerr_fwd": [C.c_longlong, C.c_int, _P, _P, _P, _P],
    "svgp_sqerr_bwd": [C.c_longlong, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P],
    "svgp_clip_by_value": [C.c_longlong, C.c_double, _P, _P],
    "svgp_se1d_kernel_matrix_fwd": [C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, _P],
    "svgp_se1d_kernel_matrix_bwd": [C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P],
    "svgp_bias_act_fwd": [C.c_longlong, C.c_int, C.c_int, _P, _P, _P],
    "svgp_act_bwd_bias": [C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P],
    "svgp_ball_head_fwd": [C.c_int, C.c_int, C.c_int] + [_P] * 9,
    "svgp_ball_head_bwd": [C.c_int, C.c_int, C.c_int] + [_P] * 8,
    "svgp_ball_pack_z": [C.c_int, C.c_int, _P, _P, _P, _P],
    "svgp_ball_unpack_zbar": [C.c_int, C.c_int, _P, _P, _P, _P],
    "svgp_sigmoid_xent": [C.c_int, C.c_int, C.c_double, _P, _P, _P, _P, _P, _P],
    "svgp_ball_elbo_assemble": [_CFG, _P, _P, _P, _P, _P, _P],
    "svgp_ball_finalize": [C.c_int, C.c_int, C.c_longlong, _P, _P, _P],
    "svgp_state_add": [_P, C.c_int, C.c_double, _P],
    "svgp_pearce_gp_fwd": [C.POINTER(PearceBufs), _P, _P, _P, _P],
    "svgp_pearce_gp_bwd": [C.POINTER(PearceBufs), C.c_double, C.c_int, _P, _P, _P, _P],
    "svgp_pearce_elbo_assemble": [C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "svgp_scale_rows": [C.c_longlong, C.c_int, _P, _P, _P],
    "svgp_svigp_fwd": [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double] + [_P] * 9,
    "svgp_svigp_bwd": [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, _P, _P, _P, _P, _P, _P, C.c_int] + [_P] * 8,
    "svgp_svigp_assemble": [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, _P, _P, C.c_int, _P, _P, _P],
    "svgp_scale_by_device_scalar": [C.c_longlong, _P, _P, _P],
    "svgp_ball_rasterize": [C.c_longlong, C.c_int, C.c_int, C.c_double, _P, _P, _P],
    "svgp_stream_create": [C.POINTER(_P)],
    "svgp_stream_destroy": [_P],
    "svgp_stream_sync": [_P],
    "svgp_graph_begin": [_P],
    "svgp_graph_end": [_P, C.POINTER(_P)],
    "svgp_graph_launch": [_P, _P],
    "svgp_graph_destroy": [_P],
    "svgp_event_create": [C.POINTER(_P)],
    "svgp_event_record": [_P, _P],
    "svgp_event_elapsed_ms": [_P, _P, C.POINTER(C.c_float)],
    "svgp_event_destroy": [_P],
}
class StreamKdesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("d1", C.c_int32), ("d2", C.c_int32), ("normalize", C.c_int32),
                ("n_table", C.c_int32), ("p", C.c_float * 4)]


NON_STATUS = {"svgp_version": ([], C.c_int), "svgp_last_error": ([], C.c_char_p),
              "svgp_struct_sizeof": ([C.c_int], C.c_int),
              "svgp_comm_unique_id_bytes": ([], C.c_int),
              "svgp_sym_packed_elems": ([C.c_int], C.c_int64),
              "svgp_stream_feature_elems": ([C.c_void_p, C.c_int64], C.c_int64),
              "svgp_stream_stats_workspace_elems": ([C.c_int64, C.c_int, C.c_int], C.c_int64),
              "svgp_spd_inverse_workspace_elems": ([C.c_int, C.c_int], C.c_size_t),
              "svgp_potrf_workspace_elems": ([C.c_int, C.c_int], C.c_size_t),
              "svgp_trsm_workspace_elems": ([C.c_int, C.c_int, C.c_int], C.c_size_t),
              "svgp_potri_workspace_elems": ([C.c_int, C.c_int], C.c_size_t),
              "svgp_lu_inverse_workspace_elems": ([C.c_int], C.c_size_t),
              "svgp_act_bwd_bias_scratch_elems": ([C.c_int], C.c_int),
              "svgp_dgemm_splitk_scratch_elems": ([C.c_int, C.c_int, C.c_int], C.c_longlong),
              "svgp_sprites_kernel_bwd_scratch_elems": ([C.POINTER(SpritesKcfg)], C.c_longlong),
              "svgp_svigp_workspace_elems": ([C.c_int, C.c_int, C.c_int], C.c_longlong),
              "svgp_svigp_scale_offset": ([C.c_int, C.c_int, C.c_int], C.c_longlong)}

_lib = None


def load_library(path=None):
    """Loads the HIP library once and checks every declared symbol.  Raises SvgpError when the
    shared object is absent (build it with `python -c "import __graft_entry__ as g; g.build()"`
    or `make -C svgp-vae_amd/csrc`)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise SvgpError(f"{p} not found: the HIP extension is not built; there is no CPU fallback")
    lib = C.CDLL(p)
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise SvgpError(f"{p} does not export {name}") from e
        fn.argtypes = argtypes
        fn.restype = C.c_int
    for name, (argtypes, restype) in NON_STATUS.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = restype
    # the ctypes mirrors must have the layout the library was compiled with (svgp_struct_sizeof)
    for which, cls in enumerate((MnistCfg, ParamLayout, WsLayout, StreamKdesc, ConvDesc, SpritesKcfg, PearceBufs, SumJob)):
        if lib.svgp_struct_sizeof(which) != C.sizeof(cls):
            raise SvgpError(f"{p}: sizeof({cls.__name__}) is {lib.svgp_struct_sizeof(which)} in the library but "
                            f"{C.sizeof(cls)} in the binding; rebuild the library or update _lib.py")
    if path is None:
        _lib = lib
    return lib


def check(rc, lib=None):
    if rc != 0:
        lib = lib or load_library()
        raise SvgpError(f"libsvgpvae_hip status {rc}: {lib.svgp_last_error().decode()}")


def call(name, *args):
    lib = load_library()
    check(getattr(lib, name)(*args), lib)
