"""Training step on the MI355X engine (f32): autograd functions over the HIP forward / backward entry points of
include/vpmi.h, an Adam optimiser on a flat buffer, and data-parallel gradient averaging over torch.distributed
(backend "nccl" = RCCL over xGMI).  PyTorch supplies the tape, device memory and the collective; every FLOP of the
layers runs in libvpmi."""
