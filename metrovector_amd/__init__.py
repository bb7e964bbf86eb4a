"""metrovector_amd — MI355X-native brute-force similarity search for MVF files.

Host-side mirror of the reference's reader API (MvfReader / VectorSpace /
Vector) plus the GPU search path that replaces
examples/similarity_search.rs::find_top_k_similar.  The numeric work lives in
the in-tree native libraries (metrovector_amd/csrc); there is no CPU fallback.
"""
from .errors import (BuildError, CorruptedData, DeviceError, DimensionMismatch, IndexOutOfBounds,  # noqa: F401
                     InvalidArgument, InvalidFormat, IoError, MvfError, UnsupportedVersion, VectorSpaceNotFound)

__all__ = ["MvfError"]
