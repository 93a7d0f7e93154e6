#!/bin/bash
# Developer tool: builds maelstrom_amd/libmaelsim_wprof.so = the library with engine.hip compiled -DWIDE_PROF (cycles of a wavefront of
# sim_kernel_wide<> by section of the round, written over msim_net_stats / meta.reserved); use with tools/wide_prof_report.py
set -e
cd "$(dirname "$0")/.."
python -m maelstrom_amd.build > /dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DWIDE_PROF -c -o maelstrom_amd/build/engine_wprof.o maelstrom_amd/csrc/engine.hip
OBJS=$(ls maelstrom_amd/build/*.o | grep -v "/engine\|/duo_\|/raft4_\|/txn8_")
hipcc --offload-arch=gfx950 -shared -fPIC -o maelstrom_amd/libmaelsim_wprof.so $OBJS maelstrom_amd/build/engine_wprof.o -ldl
echo built maelstrom_amd/libmaelsim_wprof.so
