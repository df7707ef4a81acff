"""sequoia-pub_amd: MI355X-native SEQUOIA hot path.

Host side mirrors the reference's Python interfaces (ViS, resnet50.forward_extract,
KMeans, train/evaluate/predict); all arithmetic runs in the hand-written HIP
kernels of ``csrc/`` behind the C ABI declared in ``include/sequoia_hip.h``.
There is no CPU fallback: using a model without the built library and a GPU raises.
"""
import os as _os

# The pipeline keeps up to 8 HIP streams busy (main, two ResNet chains, k-Means, uploads, the library's two helper
# streams, RCCL).  ROCm maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); streams sharing a queue
# serialise -- measured: uploads from pinned host memory stuck behind a ResNet chain, 38 instead of 50 slides/s.
# Must be set before the HIP runtime initialises (first device call), hence here, at import.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

__version__ = "0.1.0"
