"""sparrowhawk_amd — MI355X-native implementation of the sparrowhawk-asm assembly path.

Host-side mirror of the reference's wasm-bindgen surface (`AssemblyHelper`,
/root/reference/www/src/workers/Assembler.ts:15-39) over the C ABI of libshk_hip.so
(include/shk.h).  The compute lives in hand-written HIP kernels for gfx950; this package holds
no fallback path — without the built library and a HIP device every call fails loudly.
"""
from .helper import AssemblyHelper, ShkError, pack_fastq  # noqa: F401
from . import _lib  # noqa: F401

__all__ = ["AssemblyHelper", "ShkError", "pack_fastq"]
