"""facl_amd -- MI355X-native hot path of tangent-T/FACL's contrastive training step.

Host side mirrors the reference's Python interface for this path (same function / class
names, argument meaning and error behaviour); all compute runs in hand-written HIP kernels
behind the C ABI of ``libfacl_hip.so`` (include/facl_hip.h).  There is no CPU or eager
fallback: importing an op without the built library raises.
"""
from ._lib import lib_path, load_library  # noqa: F401

__version__ = "0.1.0"
