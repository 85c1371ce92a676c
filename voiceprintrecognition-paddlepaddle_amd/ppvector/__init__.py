"""MI355X-native drop-in for the hot path of ``ppvector`` (yeyupiaoling/VoiceprintRecognition-PaddlePaddle).

Same module paths, class names and call signatures as the reference package for the path
Fbank / MelSpectrogram -> ECAPA-TDNN / TDNN / CAM++ / ResNetSE / ERes2Net -> cosine head -> AAMLoss -> scoring, and the
training step of the same models; compute runs in libvpmi.so
(hand-written HIP for gfx950, include/vpmi.h).  PyTorch is used for device memory and streams.
"""
__version__ = "1.1.1+mi355x.0"

_COMPUTE_DTYPE = 'float32'


def set_compute_dtype(name):
    """Arithmetic of the eval-mode backbones:
    'float32'   exact f32 matrix cores (v_mfma_f32_16x16x4_f32) -- the reference's precision, the default;
    'float32x3' f32 tensors, every conv / GEMM in split precision: both operands become bf16 hi + lo on their way into LDS and are
                contracted as hi*hi + hi*lo + lo*hi on the bf16 matrix cores with f32 accumulation (~2^-16 per product).  Meets the
                reference's 1e-4 cosine-score tolerance at trained weights on every backbone at several times the f32 rate;
    'bfloat16'  bf16 tensors and bf16 MFMA with f32 accumulation and f32 statistics -- the throughput path; at TRAINED weights its
                scores differ from the f32 reference by 2e-3 (ECAPA-TDNN) to 4e-2 (ResNetSE): outside the 1e-4 tolerance."""
    global _COMPUTE_DTYPE
    if name not in ('float32', 'float32x3', 'bfloat16'):
        raise ValueError(f'unsupported compute dtype {name}')
    _COMPUTE_DTYPE = name


def get_compute_dtype():
    return _COMPUTE_DTYPE


_GRAPH_MODE = False


def set_graph_mode(on):
    """Eval-mode backbones replay their launch sequence from a captured HIP graph (one per input shape and dtype) instead of
    issuing ~50-400 launches from the host: pays for the launch-bound models (CAM++: 52 dense layers of small kernels)."""
    global _GRAPH_MODE
    _GRAPH_MODE = bool(on)


def get_graph_mode():
    return _GRAPH_MODE


_FORWARD_STREAMS = 1


def set_forward_streams(n):
    """Eval-mode backbones cut batches of >= 32 n utterances into n shards that run as concurrent launch sequences on n HIP
    streams (bit-identical results; ppvector/models/engine.py: forward_streams)."""
    global _FORWARD_STREAMS
    _FORWARD_STREAMS = max(1, int(n))


def get_forward_streams():
    return _FORWARD_STREAMS


_TRAIN_AMP = False


def set_train_amp(on):
    """Mixed-precision training (the reference's train_conf.enable_amp -> paddle.amp.auto_cast level O1, trainer.py:209-229): the
    convolution GEMMs of the training step -- forward, data gradient, weight gradient -- round their f32 operands to bf16 on the way
    into LDS and run the bf16 matrix cores with f32 accumulation; tensors, BatchNorm statistics, the head / loss, gradients in
    memory and the Adam master weights stay f32.  bf16 has f32's exponent range: no loss scaling (GradScaler) is needed."""
    global _TRAIN_AMP
    _TRAIN_AMP = bool(on)


def get_train_amp():
    return _TRAIN_AMP


_TRAIN_X3 = False


def set_train_x3(on):
    """Split-precision training (no reference counterpart; off by default): tensors, statistics, gradients and master weights stay f32
    exactly as in the default (exact-f32) step, but the convolution GEMMs of the step -- forward, data gradient, weight gradient -- carry
    each f32 operand as bf16 hi + lo and contract hi*hi + hi*lo + lo*hi on the bf16 matrix cores with f32 accumulation (~2^-17 per product):
    f32-grade gradients at a multiple of the f32 matrix-core rate.  Ignored while enable_amp (set_train_amp) is on."""
    global _TRAIN_X3
    _TRAIN_X3 = bool(on)


def get_train_x3():
    return _TRAIN_X3 and not _TRAIN_AMP


_FUSED_GRID_KERNELS = True


def set_fused_grid_kernels(on):
    """The training kernels that meet at an in-kernel grid barrier (csrc/res2_train.hip).  GraphedTrainStep switches them off for the
    rest of the process when a barrier gives up (workgroups not co-resident: another process on the GPU); the per-chunk kernels run
    instead."""
    global _FUSED_GRID_KERNELS
    _FUSED_GRID_KERNELS = bool(on)


def get_fused_grid_kernels():
    return _FUSED_GRID_KERNELS
