"""ctypes binding of libvpmi.so (include/vpmi.h) -- the only compute backend of this package.

There is deliberately NO CPU / PyTorch fallback: if the HIP library is missing or no GPU is
visible, every compute entry point raises.  PyTorch is used for device memory, the current HIP
stream and (later) torch.distributed only.
"""
import ctypes as C
import os
import threading

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.dirname(HERE)
LIB_PATH = os.environ.get('VPMI_LIB') or os.path.join(PKG_ROOT, 'lib', 'libvpmi.so')   # VPMI_LIB: A/B a build

VP_F32, VP_BF16, VP_F32X3, VP_HL32 = 0, 1, 2, 3
VP_OK, VP_EINVAL, VP_ENOMEM, VP_EHIP, VP_EUNSUP, VP_EWORKSPACE = 0, -1, -2, -3, -4, -5
VP_PAD_NONE, VP_PAD_REFLECT, VP_PAD_ZERO = 0, 1, 2
VP_ACT_NONE, VP_ACT_RELU, VP_ACT_SIGMOID, VP_ACT_TANH, VP_ACT_HARDTANH20, VP_ACT_SILU = 0, 1, 2, 3, 4, 5
VP_LOSS_AAM, VP_LOSS_AM, VP_LOSS_ARM, VP_LOSS_CE, VP_LOSS_SUBCENTER = 0, 1, 2, 3, 4
VP_MAX_SE_BLOCKS, VP_MAX_RES2 = 8, 15

c_void_p, c_int, c_float, c_size_t = C.c_void_p, C.c_int, C.c_float, C.c_size_t


class FbankOpts(C.Structure):
    _fields_ = [('sample_rate', c_int), ('n_mels', c_int), ('frame_length_ms', c_float),
                ('frame_shift_ms', c_float), ('preemph', c_float), ('remove_dc', c_int),
                ('low_freq', c_float), ('high_freq', c_float), ('log_floor', c_float)]


class MelOpts(C.Structure):
    _fields_ = [('sample_rate', c_int), ('n_fft', c_int), ('hop_length', c_int), ('win_length', c_int),
                ('n_mels', c_int), ('f_min', c_float), ('f_max', c_float), ('power', c_float),
                ('log_db', c_int), ('amin', c_float), ('ref_value', c_float)]


class Conv1dDesc(C.Structure):
    _fields_ = [('dtype_in', c_int), ('dtype_out', c_int), ('B', c_int), ('T_in', c_int), ('T_out', c_int),
                ('Cin', c_int), ('Cout', c_int), ('KW', c_int), ('dilation', c_int), ('stride', c_int),
                ('pad_left', c_int), ('pad_mode', c_int),
                ('x', c_void_p), ('ldx', c_int), ('xoff', c_int),
                ('w', c_void_p), ('bias', c_void_p), ('rowbias', c_void_p), ('act', c_int),
                ('bn_scale', c_void_p), ('bn_shift', c_void_p), ('act2', c_int),
                ('y', c_void_p), ('ldy', c_int), ('yoff', c_int),
                ('y2', c_void_p), ('ldy2', c_int), ('y2off', c_int), ('ysplit', c_int),
                ('add_in', c_void_p), ('ld_add', c_int), ('add_off', c_int),
                ('aux', c_void_p), ('ld_aux', c_int), ('aux_off', c_int),
                ('psum', c_void_p), ('psumsq', c_void_p),
                ('F_in', c_int), ('F_out', c_int), ('KF', c_int), ('stride_f', c_int), ('pad_f', c_int),
                ('pro_scale', c_void_p), ('pro_shift', c_void_p),
                ('res', c_void_p), ('ld_res', c_int), ('res_off', c_int),
                ('gate', c_void_p), ('gate_len', c_int), ('gate_nseg', c_int), ('mfma_bf16', c_int)]


class Res2TrainDesc(C.Structure):
    _fields_ = [('B', c_int), ('T', c_int), ('C', c_int), ('scale', c_int), ('width', c_int), ('dil', c_int),
                ('momentum', c_float), ('eps', c_float), ('x', c_void_p), ('out', c_void_p),
                ('w', c_void_p * 7), ('bias', c_void_p * 7), ('gamma', c_void_p * 7), ('beta', c_void_p * 7),
                ('run_mean', c_void_p * 7), ('run_var', c_void_p * 7),
                ('z', c_void_p), ('inb', c_void_p), ('dzb', c_void_p), ('stats', c_void_p), ('dvec', c_void_p), ('out_bf16', c_void_p),
                ('x_is_bf16', c_int)]


class TdnnLayer(C.Structure):
    _fields_ = [('w', c_void_p), ('bias', c_void_p), ('bn_scale', c_void_p), ('bn_shift', c_void_p),
                ('cin', c_int), ('cout', c_int), ('kw', c_int), ('dil', c_int), ('w_hl', c_void_p)]


class SeRes2Block(C.Structure):
    _fields_ = [('tdnn1', TdnnLayer), ('res2', TdnnLayer * VP_MAX_RES2), ('tdnn2', TdnnLayer),
                ('se_w1', c_void_p), ('se_b1', c_void_p), ('se_w2', c_void_p), ('se_b2', c_void_p)]


class AspWeights(C.Structure):
    _fields_ = [('tdnn', TdnnLayer), ('w_ctx', c_void_p), ('conv_w', c_void_p), ('conv_b', c_void_p),
                ('C', c_int), ('att', c_int)]


class EcapaWeights(C.Structure):
    _fields_ = [('dtype', c_int), ('feat_dim', c_int), ('embd_dim', c_int), ('n_blocks', c_int),
                ('res2_scale', c_int), ('se_ch', c_int), ('block0', TdnnLayer),
                ('blk', SeRes2Block * VP_MAX_SE_BLOCKS), ('mfa', TdnnLayer), ('asp', AspWeights),
                ('fc_w', c_void_p), ('fc_b', c_void_p)]


class TdnnWeights(C.Structure):
    _fields_ = [('dtype', c_int), ('feat_dim', c_int), ('embd_dim', c_int), ('channels', c_int),
                ('td', TdnnLayer * 5), ('asp', AspWeights), ('lin_w', c_void_p), ('lin_b', c_void_p)]


VP_MAX_CAM_LAYERS, VP_MAX_CAM_BLOCKS = 64, 4


class ResBlock(C.Structure):
    _fields_ = [('conv1', TdnnLayer), ('conv2', TdnnLayer), ('shortcut', TdnnLayer), ('stride', c_int),
                ('has_shortcut', c_int)]


class CamLayer(C.Structure):
    _fields_ = [('bn1_scale', c_void_p), ('bn1_shift', c_void_p), ('linear1', TdnnLayer), ('local', TdnnLayer),
                ('ctx_w1', c_void_p), ('ctx_b1', c_void_p), ('ctx_w2', c_void_p), ('ctx_b2', c_void_p)]


class Transit(C.Structure):
    _fields_ = [('bn_scale', c_void_p), ('bn_shift', c_void_p), ('linear', TdnnLayer)]


class CamppWeights(C.Structure):
    _fields_ = [('dtype', c_int), ('feat_dim', c_int), ('embd_dim', c_int), ('m_channels', c_int),
                ('init_channels', c_int), ('growth', c_int), ('bn_channels', c_int), ('seg_len', c_int),
                ('n_blocks', c_int), ('block_layers', c_int * VP_MAX_CAM_BLOCKS),
                ('fcm1_w', c_void_p), ('fcm1_b', c_void_p), ('fcm1_scale', c_void_p), ('fcm1_shift', c_void_p),
                ('res', ResBlock * 4), ('fcm_conv2', TdnnLayer), ('tdnn', TdnnLayer),
                ('layers', CamLayer * VP_MAX_CAM_LAYERS), ('transit', Transit * VP_MAX_CAM_BLOCKS),
                ('out_bn_scale', c_void_p), ('out_bn_shift', c_void_p), ('dense_w', c_void_p), ('dense_b', c_void_p)]


VP_MAX_RSE_BLOCKS = 32


class RseBlock(C.Structure):
    _fields_ = [('conv1', TdnnLayer), ('conv2', TdnnLayer), ('conv3', TdnnLayer), ('down', TdnnLayer),
                ('se_w1', c_void_p), ('se_b1', c_void_p), ('se_w2', c_void_p), ('se_b2', c_void_p),
                ('stride', c_int), ('has_down', c_int)]


class ResnetSeWeights(C.Structure):
    _fields_ = [('dtype', c_int), ('feat_dim', c_int), ('embd_dim', c_int), ('n_blocks', c_int), ('c1_channels', c_int),
                ('c1_w', c_void_p), ('c1_b', c_void_p), ('c1_scale', c_void_p), ('c1_shift', c_void_p),
                ('blk', RseBlock * VP_MAX_RSE_BLOCKS), ('asp', AspWeights), ('lin_w', c_void_p), ('lin_b', c_void_p)]


VP_MAX_ERE_BLOCKS, VP_MAX_ERE_SCALE = 40, 4


class AffWeights(C.Structure):
    _fields_ = [('c1', TdnnLayer), ('c2', TdnnLayer)]


class EreBlock(C.Structure):
    _fields_ = [('conv1', TdnnLayer), ('convs', TdnnLayer * VP_MAX_ERE_SCALE), ('conv3', TdnnLayer), ('shortcut', TdnnLayer),
                ('fuse', AffWeights * (VP_MAX_ERE_SCALE - 1)), ('stride', c_int), ('has_shortcut', c_int), ('use_aff', c_int),
                ('width', c_int), ('scale', c_int)]


class Eres2netWeights(C.Structure):
    _fields_ = [('dtype', c_int), ('feat_dim', c_int), ('embd_dim', c_int), ('n_blocks', c_int), ('m_channels', c_int),
                ('stage_blocks', c_int * 4), ('first_fuse', c_int), ('c1_w', c_void_p), ('c1_b', c_void_p), ('c1_scale', c_void_p), ('c1_shift', c_void_p),
                ('blk', EreBlock * VP_MAX_ERE_BLOCKS), ('down', TdnnLayer * 3), ('fuse', AffWeights * 3),
                ('seg_w', c_void_p), ('seg_b', c_void_p)]


_PROTOS = {
    'vp_version': (c_int, []),
    'vp_create': (c_void_p, [c_int]),
    'vp_destroy': (None, [c_void_p]),
    'vp_last_error': (C.c_char_p, [c_void_p]),
    'vp_set_margin_table': (c_int, [c_void_p, c_void_p]),
    'vp_conv1d_wgrad_bf16_oik_batched': (c_int, [c_void_p, C.POINTER(Conv1dDesc), c_void_p, c_int, c_void_p, c_int, C.c_longlong, C.c_longlong, c_void_p,
                                         c_size_t, c_void_p]),
    'vp_se_scale_residual_shadow': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p,
                                    c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'vp_moments_finalize_affine': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_int, c_void_p, c_void_p]),
    'vp_se_dense_train_fwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                              c_void_p]),
    'vp_se_dense_train_bwd_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'vp_se_dense_train_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                              c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'vp_scale_rows_bwd_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'vp_scale_rows_bwd_ws_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'vp_time_stats_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'vp_time_stats_ws_f32': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    'vp_col_sums_masked_f32': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, C.c_float, C.c_longlong,
                               c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    'vp_bn_relu_bwd_masked_f32': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  C.c_float, C.c_longlong, c_int, c_void_p, c_int, c_int, c_void_p]),
    'vp_res2_train_workspace_bytes': (c_size_t, [c_int, c_int]),
    'vp_res2_train_fwd': (c_int, [c_void_p, C.POINTER(Res2TrainDesc), c_void_p, c_size_t, c_void_p]),
    'vp_res2_train_bwd': (c_int, [c_void_p, C.POINTER(Res2TrainDesc), c_void_p, c_size_t, c_void_p]),
    'vp_se_scale_residual_z16': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int,
                                 c_int, c_int, c_int, c_void_p]),
    'vp_utt_dot_z16': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    'vp_col_sums_f32_b16_utt': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, C.c_longlong,
                                c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    'vp_bn_relu_bwd_dbias_b16_utt': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                     c_void_p, C.c_longlong, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    'vp_time_stats_bwd_coeffs': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p]),
    'vp_col_sums_f32_b16_ctx': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                C.c_longlong, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    'vp_bn_relu_bwd_dbias_b16_ctx': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                     c_void_p, c_void_p, c_void_p, C.c_longlong, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_size_t,
                                     c_void_p]),
    'vp_prep_weights_bf16': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    'vp_grid_barrier_status': (c_int, [c_void_p]),
    'vp_grid_barrier_reset': (c_int, [c_void_p, c_void_p]),
    'vp_set_grid_barrier_words': (c_int, [c_void_p, c_void_p]),
    'vp_set_grid_reserve_cus': (c_int, [c_void_p, c_int]),
    'vp_occupy_cus': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p]),
    'vp_cosine_aam_tiled_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'vp_cosine_aam_tiled_bwd_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'vp_cosine_aam_tiled_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float, c_float, c_int, c_float,
                                        c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'vp_cosine_aam_tiled_fwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float, c_float, c_int,
                                        c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'vp_fbank_default_opts': (None, [C.POINTER(FbankOpts)]),
    'vp_fbank_num_frames': (c_int, [C.POINTER(FbankOpts), c_int]),
    'vp_fbank_workspace_bytes': (c_size_t, [C.POINTER(FbankOpts), c_int, c_int]),
    'vp_fbank_cmn_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, C.POINTER(FbankOpts), c_void_p,
                                 c_void_p, c_void_p, c_size_t, c_void_p]),
    'vp_fbank_cmn_pcm16': (c_int, [c_void_p, c_void_p, c_float, c_void_p, c_int, c_int, C.POINTER(FbankOpts), c_void_p, c_void_p, c_void_p,
                                   c_size_t, c_void_p]),
    'vp_fbank_cmn_ragged_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, C.POINTER(FbankOpts), c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_size_t, c_void_p]),
    'vp_mel_default_opts': (None, [C.POINTER(MelOpts)]),
    'vp_mel_num_frames': (c_int, [C.POINTER(MelOpts), c_int]),
    'vp_mel_workspace_bytes': (c_size_t, [C.POINTER(MelOpts), c_int, c_int]),
    'vp_melspec_cmn_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, C.POINTER(MelOpts), c_void_p,
                                   c_void_p, c_void_p, c_size_t, c_void_p]),
    'vp_conv1d_tiles_m': (c_int, [c_int, c_int]),
    'vp_conv1d_nseg': (c_int, [c_int]),
    'vp_conv1d_fwd': (c_int, [c_void_p, C.POINTER(Conv1dDesc), c_void_p]),
    'vp_moments_finalize': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_int,
                                    c_void_p, c_void_p]),
    'vp_dense_f32': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int,
                             c_void_p, c_int, c_void_p]),
    'vp_cast_f32_bf16': (c_int, [c_void_p, c_void_p, c_void_p, C.c_longlong, c_void_p]),
    'vp_se_scale_residual': (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int,
                                     c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'vp_asp_softmax_stats': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                     c_float, c_void_p, c_void_p]),
    'vp_ecapa_workspace_bytes': (c_size_t, [C.POINTER(EcapaWeights), c_int, c_int]),
    'vp_ecapa_fwd': (c_int, [c_void_p, C.POINTER(EcapaWeights), c_void_p, c_int, c_int, c_void_p, c_void_p,
                             c_size_t, c_void_p]),
    'vp_tdnn_workspace_bytes': (c_size_t, [C.POINTER(TdnnWeights), c_int, c_int]),
    'vp_tdnn_fwd': (c_int, [c_void_p, C.POINTER(TdnnWeights), c_void_p, c_int, c_int, c_void_p, c_void_p,
                            c_size_t, c_void_p]),
    'vp_campplus_workspace_bytes': (c_size_t, [C.POINTER(CamppWeights), c_int, c_int]),
    'vp_campplus_fwd': (c_int, [c_void_p, C.POINTER(CamppWeights), c_void_p, c_int, c_int, c_void_p, c_void_p,
                                c_size_t, c_void_p]),
    'vp_spec_augment': (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p]),
    'vp_pad_batch': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    'vp_wave_batch_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p,
                                  c_void_p]),
    'vp_speed_perturb_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    'vp_resnetse_workspace_bytes': (c_size_t, [C.POINTER(ResnetSeWeights), c_int, c_int]),
    'vp_resnetse_fwd': (c_int, [c_void_p, C.POINTER(ResnetSeWeights), c_void_p, c_int, c_int, c_void_p, c_void_p,
                                c_size_t, c_void_p]),
    'vp_eres2net_workspace_bytes': (c_size_t, [C.POINTER(Eres2netWeights), c_int, c_int]),
    'vp_eres2net_fwd': (c_int, [c_void_p, C.POINTER(Eres2netWeights), c_void_p, c_int, c_int, c_void_p, c_void_p,
                                c_size_t, c_void_p]),
    'vp_cosine_logits_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'vp_cosine_logits_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p,
                                     c_size_t, c_void_p]),
    'vp_aam_ce_fwd': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_float, c_int,
                              c_void_p, c_void_p, c_void_p]),
    'vp_cosine_aam_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'vp_cosine_aam_ce_fwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float,
                                     c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'vp_cosine_aam_ce_bwd_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'vp_cosine_aam_ce_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float, c_float, c_int,
                                     c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'vp_conv1d_wgrad_workspace_bytes': (c_size_t, [C.POINTER(Conv1dDesc)]),
    'vp_conv1d_wgrad_f32': (c_int, [c_void_p, C.POINTER(Conv1dDesc), c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    'vp_conv1d_wgrad_oik_f32': (c_int, [c_void_p, C.POINTER(Conv1dDesc), c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    'vp_conv_weight_layouts_f32': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'vp_conv2d_weight_layouts_f32': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'vp_col_sums_workspace_bytes': (c_size_t, [C.c_longlong, c_int]),
    'vp_col_sums_f32': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, C.c_longlong, c_int, c_void_p,
                                c_void_p, c_size_t, c_void_p]),
    'vp_bn_train_finalize': (c_int, [c_void_p, c_void_p, c_void_p, c_int, C.c_longlong, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'vp_affine_rows_f32': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, C.c_longlong, c_int, c_void_p, c_int, c_int, c_void_p]),
    'vp_bn_relu_bwd_f32': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, C.c_longlong,
                                   c_int, c_int, c_void_p, c_int, c_void_p]),
    'vp_bn_relu_bwd_dbias_workspace_bytes': (c_size_t, [C.c_longlong, c_int]),
    'vp_bn_relu_bwd_dbias_f32': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, C.c_longlong,
                                         c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    'vp_bn_relu_bwd_dbias_bf16out': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, C.c_longlong,
                                             c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    'vp_conv1d_wgrad_bf16_oik': (c_int, [c_void_p, C.POINTER(Conv1dDesc), c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    'vp_affine_rows_b16_f32': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, C.c_longlong, c_int, c_void_p, c_int, c_int, c_void_p]),
    'vp_col_sums_f32_b16': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, C.c_longlong, c_int, c_void_p,
                                    c_void_p, c_size_t, c_void_p]),
    'vp_bn_relu_bwd_dbias_b16': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, C.c_longlong,
                                         c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    'vp_pack_segments_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    'vp_adamw_step_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, C.c_longlong, c_float, c_float, c_float, c_float,
                                  c_float, c_int, c_float, c_void_p]),
    'vp_momentum_step_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, C.c_longlong, c_float, c_float, c_float, c_int, c_float,
                                     c_void_p]),
    'vp_adam_step_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, C.c_longlong, c_float, c_float, c_float, c_float,
                                 c_float, c_int, c_float, c_void_p]),
    'vp_utt_sums_f32': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'vp_time_stats_f32': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p, c_void_p]),
    'vp_time_stats_bwd_add_f32': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_int, c_void_p, c_int,
                                          c_void_p, c_int, c_void_p]),
    'vp_time_stats_bwd_f32': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_int, c_void_p, c_int, c_void_p]),
    'vp_attn_stats_bwd_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p,
                                      c_void_p, c_int, c_void_p]),
    'vp_asp_softmax_stats_l16': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p]),
    'vp_attn_stats_bwd_e16': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p,
                                       c_void_p, c_int, c_void_p]),
    'vp_affine_rows_f32_b16': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, C.c_longlong, c_int, c_void_p, c_int, c_int, c_void_p]),
    'vp_affine_rows_b16_b16': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, C.c_longlong, c_int, c_void_p, c_int, c_int, c_void_p]),
    'vp_time_stats_bwd_add_x16': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_int, c_void_p, c_int,
                                           c_void_p, c_int, c_void_p]),
    'vp_utt_sums_b16': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'vp_utt_dot_x16': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    'vp_attn_stats_bwd_de16': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p,
                                      c_void_p, c_int, c_void_p]),
    'vp_act_f32': (c_int, [c_void_p, c_int, c_void_p, C.c_longlong, c_void_p, c_void_p]),
    'vp_act_bwd_f32': (c_int, [c_void_p, c_int, c_void_p, c_void_p, C.c_longlong, c_void_p, c_void_p]),
    'vp_zero_insert_2d_f32': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'vp_seg_ctx_f32': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'vp_seg_ctx_bwd_f32': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'vp_seg_scale_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'vp_seg_scale_bwd_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'vp_cam_gate_fwd_f32': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                    c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    'vp_cam_gate_bwd_f32': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                    c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    'vp_cam_gate_wgrad_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'vp_relu_bwd_f32': (c_int, [c_void_p, c_void_p, c_void_p, C.c_longlong, c_void_p, c_void_p]),
    'vp_aff_combine_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, C.c_longlong, c_int, c_void_p, c_void_p]),
    'vp_aff_combine_bwd_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, C.c_longlong, c_void_p, c_void_p, c_void_p, c_void_p]),
    'vp_reflect_fold_f32': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'vp_affine_rows_aux_f32': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, C.c_longlong, c_int, c_void_p, c_int, c_void_p, c_int,
                                       c_void_p, c_int, c_void_p]),
    'vp_reflect_fold_into_f32': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    'vp_utt_dot_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    'vp_scale_shift_rows_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    'vp_scale_rows_bwd_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'vp_aam_ce_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_float, c_int, c_float, c_void_p,
                              c_void_p, c_void_p, c_void_p]),
    'vp_margin_ce_fwd': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_int,
                                 c_void_p, c_void_p, c_void_p]),
    'vp_margin_ce_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_int, c_float,
                                 c_void_p, c_void_p, c_void_p, c_void_p]),
    'vp_sphereface2': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_float, c_int, c_int, c_float,
                               c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'vp_cosine_logits_bwd_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'vp_cosine_logits_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                     c_size_t, c_void_p]),
    'vp_conv256_select': (c_int, [c_int]),
    'vp_cam_block_fwd': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'vp_resblock_c32_fwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    'vp_conv3x3_c32_fwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'vp_pointwise_fwd': (c_int, [c_void_p, C.POINTER(Conv1dDesc), c_void_p]),
    'vp_res2_chain_fwd': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'vp_res2_chain_x3_fwd': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'vp_asp_fused_x3_fwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                    c_float, c_void_p, c_void_p]),
    'vp_asp_utt_fwd': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p]),
    'vp_asp_fused_fwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                 c_float, c_void_p, c_void_p]),
    'vp_se_gate_fwd': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                               c_void_p]),
    'vp_cosine_scores_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'vp_cosine_scores_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p,
                                     c_size_t, c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_PROTOS.keys())

_lib = None
_lock = threading.RLock()      # re-entrant: ctx() -> lib() -> load_library() nest
_ctx = {}


class VpmiError(RuntimeError):
    pass


def load_library():
    """dlopen libvpmi.so and set the prototypes.  Works without a GPU (symbol checks only)."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise VpmiError(f'{LIB_PATH} is missing: build it with `python {PKG_ROOT}/build.py` '
                                '(there is no CPU fallback)')
            lib = C.CDLL(LIB_PATH)
            for name, (res, args) in _PROTOS.items():
                fn = getattr(lib, name)
                fn.restype = res
                fn.argtypes = args
            _lib = lib
    return _lib


def lib():
    return load_library()


def default_device():
    """The current HIP device as a torch.device; raises when none is visible (no CPU fallback)."""
    if not torch.cuda.is_available():
        raise VpmiError('no HIP device visible: the ppvector MI355X engine has no CPU fallback')
    return torch.device('cuda', torch.cuda.current_device())


def ctx(device=None):
    """Per-device vp_ctx; requires a visible GPU."""
    if not torch.cuda.is_available():
        raise VpmiError('no HIP device visible: the ppvector MI355X engine has no CPU fallback')
    if device is None:
        device = torch.cuda.current_device()
    device = torch.device(device).index if not isinstance(device, int) else device
    if device is None:
        device = torch.cuda.current_device()
    library = lib()
    with _lock:
        if device not in _ctx:
            h = library.vp_create(device)
            if not h:
                raise VpmiError(f'vp_create({device}) failed')
            _ctx[device] = h
            # the grid-barrier words of the fused training kernels live in a tensor of OURS (csrc/api.hip: vp_set_grid_barrier_words): the
            # data-parallel step all-reduces the bail-out flag with the gradients and polls it with an asynchronous copy (train/step.py)
            # (never under a stream capture: an allocation + synchronise there would break it; the context then keeps its own words, which
            # work the same on one rank -- only the collective drop across ranks needs the tensor)
            if not torch.cuda.is_current_stream_capturing():
                words = torch.zeros(GRID_WORDS, dtype=torch.int32, device=torch.device('cuda', device))
                torch.cuda.synchronize(device)
                if library.vp_set_grid_barrier_words(h, words.data_ptr()) == 0:
                    _grid_words[device] = words
    return _ctx[device]


GRID_WORDS, FAULT_WORD = 512, 8 * 32 + 1       # csrc/common.h: VP_FAULT_WORD
_grid_words = {}


def grid_words(device=None):
    """The context's grid-barrier words as an int32 tensor (512,) -- element FAULT_WORD is the bail-out flag -- or None."""
    ctx(device)
    if device is None:
        device = torch.cuda.current_device()
    device = torch.device(device).index if not isinstance(device, int) else device
    if device is None:
        device = torch.cuda.current_device()
    return _grid_words.get(device)


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


# Kernels that write parameters or buffers through raw pointers (the flat Adam step, the BatchNorm running statistics of a
# train-mode forward) do not move torch's per-tensor version counters.  Every such writer bumps this epoch; caches of packed
# weights (ppvector/models/engine.py) carry it in their key.
_weights_epoch = 0


def bump_weights_epoch():
    global _weights_epoch
    _weights_epoch += 1


def weights_epoch():
    return _weights_epoch


def check(rc, c=None):
    if rc != 0:
        msg = lib().vp_last_error(c).decode('utf-8', 'replace') if c else ''
        raise VpmiError(f'libvpmi error {rc}: {msg}')


def ptr(t):
    return None if t is None else t.data_ptr()


def dtype_id(torch_dtype):
    if torch_dtype == torch.float32:
        return VP_F32
    if torch_dtype == torch.bfloat16:
        return VP_BF16
    raise VpmiError(f'unsupported dtype {torch_dtype}')


class Workspace:
    """Grow-only device scratch, one per consumer; reused across calls so the steady state
    allocates nothing (hipGraph-friendly)."""

    def __init__(self):
        self.bufs = {}

    def get(self, nbytes, device):
        # one buffer per (device, stream): the same consumer may run on several streams at once (engine.forward_streams' shard
        # tails), and a kernel's scratch must not be shared between launch sequences that overlap
        key = (device, torch.cuda.current_stream(device).cuda_stream if device.type == 'cuda' else 0)
        buf = self.bufs.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
            self.bufs[key] = buf
        return buf
