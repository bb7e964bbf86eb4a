"""metrovector_amd — MI355X-native brute-force similarity search for MVF files.

Host-side mirror of the reference's reader API (MvfReader / VectorSpace /
Vector) plus the GPU search path that replaces
examples/similarity_search.rs::find_top_k_similar.  The numeric work lives in
the in-tree native libraries (metrovector_amd/csrc); there is no CPU fallback.
"""
from .errors import (BuildError, CorruptedData, DeviceError, DimensionMismatch, IndexOutOfBounds,  # noqa: F401
                     InvalidArgument, InvalidFormat, IoError, MvfError, UnsupportedVersion, VectorSpaceNotFound)

from .reader import MvfReader, Vector, VectorSlice, VectorSpace  # noqa: F401,E402  (reference: src/reader.rs, src/vectors/*)
from .builder import BuiltMvf, MvfBuilder  # noqa: F401,E402                      (reference: src/builder.rs)
from .gpu import GpuCorpus, SearchResult  # noqa: F401,E402
from .search import ScoredVector, find_top_k_similar, find_top_k_similar_batch, upload_space  # noqa: F401,E402  (examples/similarity_search.rs:140-176)

__all__ = ["MvfError", "MvfReader", "VectorSpace", "Vector", "VectorSlice", "MvfBuilder", "BuiltMvf", "GpuCorpus",
           "SearchResult", "ScoredVector", "find_top_k_similar", "find_top_k_similar_batch", "upload_space"]
