#!/bin/bash
# Developer A/B on one box: the split-fp16 layers with and without the tile-walking stream (SSTEM_SPLIT_WALK = tiles per workgroup, 0 = off), three alternations
cd "$(dirname "$0")/.."
L="8,32,1024,1024,32 8,64,1024,1024,32 8,32,512,512,32 8,6,1024,1024,32 16,32,256,256,32 2,32,256,256,32"
for rep in 1 2 3; do
  for w in 0 2 4 8; do
    echo -n "walk=$w: "; SSTEM_SPLIT_WALK=$w python tools/time_conv.py f16x3 $L
  done
done
