"""tagdust_amd -- TagDust2's per-read HMM decoding path on MI355X (gfx950).

The product is libtagdust_hip.so (C-ABI in include/tagdust_hip.h; hand-written HIP kernels in
csrc/).  This Python package is a thin ctypes mirror of that ABI for tests and bench.py; it never
falls back to a CPU implementation."""
from .lib import (TagdustHip, TdError, load_library, LIB_PATH, RESULT_DTYPE,  # noqa: F401
                  MODE_GET_LABEL, MODE_GET_PROB, MODE_ARCH_COMP, NUM_COUNTERS, NUM_OUTCOME_SLOTS)
from . import shard  # noqa: F401,E402
