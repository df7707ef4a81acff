"""sequoia-pub_amd: MI355X-native SEQUOIA hot path.

Host side mirrors the reference's Python interfaces (ViS, resnet50.forward_extract,
KMeans, train/evaluate/predict); all arithmetic runs in the hand-written HIP
kernels of ``csrc/`` behind the C ABI declared in ``include/sequoia_hip.h``.
There is no CPU fallback: using a model without the built library and a GPU raises.
"""
__version__ = "0.1.0"
