"""CPU oracle for the sepconv hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  The product operator
(``sstem-restoration_amd/libs/sepconv``) never does; it raises when the HIP
library is missing.

``sepconv_c``     ctypes binding of ``sepconv_oracle.c`` -- the literal C
                  restatement of libs/sepconv/src/SeparableConvolution_kernel.cu
                  (:25-52 forward, :77-112 gradVertical, :115-150 gradHorizontal).
``sepconv_numpy`` an independent float64 restatement (sliding windows + einsum)
                  used only to cross-check ``sepconv_c``.

Pinning status: "parity unpinned" by reference outputs (the reference has no
fixtures/tests for this op, no CPU implementation of it, and its CUDA sources
do not build here); pinned by analytic KATs + the independent restatement.
"""
from . import sepconv_c, sepconv_numpy, warp_numpy  # noqa: F401
