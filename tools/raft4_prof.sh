#!/bin/bash
# Developer tool: builds maelstrom_amd/libmaelsim_<tag>.so with raft4.hip compiled -DR4_PROF (cycle counters of the round's
# sections written into msim_inst_meta) plus any extra -D flags; use with MSIM_LIB=... tools/raft4_prof_report.py
#   tools/raft4_prof.sh [tag [extra hipcc flags...]]      (default tag: r4prof)
set -e
cd "$(dirname "$0")/.."
TAG=${1:-r4prof}; shift || true
python -m maelstrom_amd.build > /dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DR4_PROF "$@" -c -o maelstrom_amd/build/raft4_$TAG.o maelstrom_amd/csrc/raft4.hip
OBJS=$(ls maelstrom_amd/build/*.o | grep -v "/raft4\|/duo_\|/txn8_")
hipcc --offload-arch=gfx950 -shared -fPIC -o maelstrom_amd/libmaelsim_$TAG.so $OBJS maelstrom_amd/build/raft4_$TAG.o -ldl
echo built maelstrom_amd/libmaelsim_$TAG.so
