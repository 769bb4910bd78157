"""sshash_amd -- MI355X-native batched k-mer Lookup engine behind SSHash's dictionary API.

This package is a thin ctypes binding over ``libsshash_amd.so`` (C ABI: ``include/sshash_amd.h``),
used by the tests and by ``bench.py``. The product is the shared library: a C++17 host layer
(``csrc/index.cpp``: index construction / persistence mirroring the reference's components) and
hand-written HIP kernels for gfx950 (``csrc/engine.hip``, ``csrc/streaming.hip``). There is no
CPU lookup path here: lookups need a visible HIP device and fail loudly otherwise.

Method names and argument meaning follow ``sshash::dictionary`` (reference
``include/dictionary.hpp:40-82``): ``lookup``, ``is_member``, ``access``,
``streaming_query_from_file``, ``k()``/``m()``/``canonical()``/``num_kmers()``/``num_strings()``.
"""
from ._binding import (  # noqa: F401
    INVALID_U64,
    Dictionary,
    LookupResult,
    SSHashError,
    StreamingQueryReport,
    device_count,
    encode_kmers,
    library_path,
)

__all__ = [
    "INVALID_U64",
    "Dictionary",
    "LookupResult",
    "SSHashError",
    "StreamingQueryReport",
    "device_count",
    "encode_kmers",
    "library_path",
]
