"""scrooge_amd — MI355X-native GenASM/Scrooge pairwise aligner.

The product is the C-ABI shared library ``libscrooge_amd.so`` (HIP kernels for
gfx950 + C++ host code, sources in ``scrooge_amd/csrc``, interface in
``include/scrooge_amd.h``).  This package is the thin ctypes binding used by
the tests and the benchmark; it mirrors the reference's two library surfaces
(src/genasm_gpu.hpp:7-8).  There is no CPU fallback: importing works without a
GPU, but every compute call raises unless the HIP library and a device exist.
"""
from .api import (Aligner, Alignment, Params, ScroogeError, build_library, library_path,
                  load_library)

__all__ = ["Aligner", "Alignment", "Params", "ScroogeError", "build_library", "library_path",
           "load_library"]
