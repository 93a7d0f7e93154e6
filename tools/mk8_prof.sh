#!/bin/bash
# Developer tool: builds maelstrom_amd/libmaelsim_<tag>.so with mk8.hip compiled -DM8_PROF (cycle counters of the round's
# sections written into msim_inst_meta); use with tools/mk8_prof_report.py
set -e
cd "$(dirname "$0")/.."
TAG=${1:-m8prof}; shift || true
python -m maelstrom_amd.build > /dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DM8_PROF "$@" -c -o maelstrom_amd/build/mk8_$TAG.o maelstrom_amd/csrc/mk8.hip
OBJS=$(ls maelstrom_amd/build/*.o | grep -v "/mk8\|/duo_\|/raft4_\|/txn8_\|/engine_w")
hipcc --offload-arch=gfx950 -shared -fPIC -o maelstrom_amd/libmaelsim_$TAG.so $OBJS maelstrom_amd/build/mk8_$TAG.o -ldl
echo built maelstrom_amd/libmaelsim_$TAG.so
