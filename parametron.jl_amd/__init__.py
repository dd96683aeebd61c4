"""parametron.jl_amd — MI355X-native implementation of Parametron.jl's parameter-update hot path.

Host-side mirror of the reference API (Model / Variable / Parameter / expression / objective /
constraint / solve!) over the C ABI of libparametron_hip.so (include/parametron_hip.h).
"""
from . import _lib  # noqa: F401
from ._lib import ArgumentError, DimensionMismatch, ErrorException  # noqa: F401
