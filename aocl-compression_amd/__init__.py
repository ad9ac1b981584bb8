"""MI355X-native block-compression engine behind the AOCL-Compression API (host-side Python mirror).

The package directory is named ``aocl-compression_amd`` (not importable as written), so the repo root
carries ``aocl_compression_amd.py`` which loads it under the importable name ``aocl_compression_amd``.
"""
from .api import (  # noqa: F401
    LZ4, SNAPPY, ZSTD, Batch, Engine, EngineError, compress_bound, engine, lib_path,
)
from . import corpus  # noqa: F401
