#!/bin/bash
# Developer tool: builds libmaelsim_<tag>.so variants of mk8.hip (LDS queue slots / transaction slots in LDS) for occupancy A/B runs:
#   tools/mk8_variants.sh rq4 -DM8_RQ=4u ; tools/mk8_variants.sh rq4sl0 -DM8_RQ=4u -DM8_SL_N=0u
set -e
cd "$(dirname "$0")/.."
TAG=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c -o /tmp/mk8_$TAG.o maelstrom_amd/csrc/mk8.hip
OBJS=$(ls maelstrom_amd/build/*.o | grep -v "/mk8")
hipcc --offload-arch=gfx950 -shared -fPIC -o maelstrom_amd/libmaelsim_$TAG.so $OBJS /tmp/mk8_$TAG.o -ldl
echo built maelstrom_amd/libmaelsim_$TAG.so
