"""MI355X-native drop-in for the hot path of ``ppvector`` (yeyupiaoling/VoiceprintRecognition-PaddlePaddle).

Same module paths, class names and call signatures as the reference package for the path
Fbank -> ECAPA-TDNN / TDNN -> cosine head -> AAMLoss -> scoring; compute runs in libvpmi.so
(hand-written HIP for gfx950, include/vpmi.h).  PyTorch is used for device memory and streams.
"""
__version__ = "1.1.1+mi355x.0"

_COMPUTE_DTYPE = 'float32'


def set_compute_dtype(name):
    """'float32' (exact f32 matrix cores; the reference's precision) or 'bfloat16' (bf16 MFMA with
    f32 accumulation and f32 statistics; the throughput path)."""
    global _COMPUTE_DTYPE
    if name not in ('float32', 'bfloat16'):
        raise ValueError(f'unsupported compute dtype {name}')
    _COMPUTE_DTYPE = name


def get_compute_dtype():
    return _COMPUTE_DTYPE
