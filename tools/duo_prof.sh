#!/bin/bash
# Developer tool: builds maelstrom_amd/libmaelsim_<tag>.so = the library with duo.hip compiled -DDUO_PROF (wave-round counts and
# cycle counters of the two round bodies written into msim_inst_meta) plus any extra -D flags; use with MSIM_LIB=...
#   tools/duo_prof.sh [tag [extra hipcc flags...]]      (default tag: prof)
set -e
cd "$(dirname "$0")/.."
TAG=${1:-prof}; shift || true
python -m maelstrom_amd.build > /dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "${@:--DDUO_PROF}" -c -o maelstrom_amd/build/duo_$TAG.o maelstrom_amd/csrc/duo.hip
OBJS=$(ls maelstrom_amd/build/*.o | grep -v "/duo\|/raft4_\|/txn8_")
hipcc --offload-arch=gfx950 -shared -fPIC -o maelstrom_amd/libmaelsim_$TAG.so $OBJS maelstrom_amd/build/duo_$TAG.o -ldl
echo built maelstrom_amd/libmaelsim_$TAG.so
