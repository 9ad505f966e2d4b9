"""MI355X-native counterpart of the reference's ``libs/sepconv`` package.

Same import path and operator contract as the reference
(``libs/sepconv/SeparableConvolution.py:11-78``); the native layer underneath is
``csrc/libsstem_hip.so`` (hand-written gfx950 kernels behind the C-ABI declared
in ``include/sstem_sepconv.h``) instead of the cffi/THC ``_cunnex.so``.
"""
