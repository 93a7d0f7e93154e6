#!/bin/bash
# Developer tool: builds maelstrom_amd/libmaelsim_<tag>.so with txn8.hip compiled -DT8_PROF (cycle counters of the round's
# sections written into msim_inst_meta); use with tools/txn8_prof_report.py
set -e
cd "$(dirname "$0")/.."
TAG=${1:-t8prof}; shift || true
python -m maelstrom_amd.build > /dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DT8_PROF "$@" -c -o maelstrom_amd/build/txn8_$TAG.o maelstrom_amd/csrc/txn8.hip
OBJS=$(ls maelstrom_amd/build/*.o | grep -v "/txn8\|/duo_\|/raft4_")
hipcc --offload-arch=gfx950 -shared -fPIC -o maelstrom_amd/libmaelsim_$TAG.so $OBJS maelstrom_amd/build/txn8_$TAG.o -ldl
echo built maelstrom_amd/libmaelsim_$TAG.so
