#!/usr/bin/env python3
"""tools/ubench/memcpy2d_repro.hip on the HIP runtime that PyTorch-ROCm bundles: torch is imported
first (as the library's Python tests do), so libmemcpy2d_repro.so binds to torch's libamdhip64
instead of the one under /opt/rocm.  Run plain and as `rocprofv3 --kernel-trace -- python3
tools/memcpy2d_repro.py`."""
import ctypes
import os
import sys

import torch  # noqa: F401  (loads the bundled HIP runtime)

here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "ubench", "libmemcpy2d_repro.so"))
torch.zeros(4, device="cuda").sum().item()  # the runtime is initialised by torch, as in the tests
print("HIP runtime of this process:", torch.version.hip, flush=True)
sys.exit(lib.memcpy2d_repro_run(int(sys.argv[1]) if len(sys.argv) > 1 else 30))
