"""MI355X-native VQA hot path (fusion model + region-feature extractor).

Host side is Python on PyTorch-ROCm (device memory, streams, torch.distributed);
all arithmetic of the hot path runs in hand-written HIP kernels behind the C ABI
of libvqahot.so (include/vqa_hot.h).  There is no CPU fallback: every op raises
if the library is missing or a call fails.
"""
from . import _lib  # noqa: F401
from ._lib import VqaHotError, lib_path  # noqa: F401
