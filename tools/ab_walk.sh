#!/bin/bash
# Developer A/B on one box: the split-fp16 layers with and without the tile-walking stream (SSTEM_SPLIT_WALK), three alternations
cd "$(dirname "$0")/.."
L="8,32,1024,1024,32 8,64,1024,1024,32 8,32,512,512,32 8,64,512,512,64 16,32,256,256,32 2,32,256,256,32"
for rep in 1 2 3; do
  for cfg in "0 32" "4 32" "2 32" "8 32" "4 96"; do
    set -- $cfg
    echo -n "walk=$1 co=$2: "; SSTEM_SPLIT_WALK=$1 SSTEM_SPLIT_WALK_CO=$2 python tools/time_conv.py f16x3 $L
  done
done
