#!/bin/bash
# Developer tool: builds variants of the native library for A/B runs into build_ablate/libsstem_<name>.so
#   tools/build_ablate.sh <name> "<extra hipcc flags>"   e.g.  tools/build_ablate.sh mem "-DSSTEM_ABLATE=12"
# (SSTEM_ABLATE bit mask and the other switches: top of sepconv_kernels.hip).  Select one at run time with
# SSTEM_NATIVE_LIB=<path>.  The other objects come from the product build (run make first).
set -e
cd "$(dirname "$0")/../sstem-restoration_amd/csrc"
OUT=../../build_ablate
mkdir -p $OUT
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function"
/opt/rocm/bin/hipcc $FLAGS $2 -c sepconv_kernels.hip -o $OUT/sepconv_$1.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libsstem_$1.so $OUT/sepconv_$1.o sstem_capi.o conv_kernels.o conv_bf16_kernels.o conv_split_kernels.o convt_kernels.o warp_kernels.o misc_kernels.o norm_kernels.o
