#!/bin/bash
# Developer tool: the product library plus two variants of engine.hip — libmaelsim_wprof.so (-DWIDE_PROF, tools/wide_prof_report.py)
# and libmaelsim_wnt.so (-DWIDE_NT_SPILL: the wide kernel's queue traffic non-temporal) — for A/B runs with MSIM_LIB
set -e
cd "$(dirname "$0")/.."
python -m maelstrom_amd.build > /dev/null
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
hipcc $F -DWIDE_PROF -c -o maelstrom_amd/build/engine_wprof.o maelstrom_amd/csrc/engine.hip &
hipcc $F -DWIDE_NT_SPILL -c -o maelstrom_amd/build/engine_wnt.o maelstrom_amd/csrc/engine.hip &
wait
OBJS=$(ls maelstrom_amd/build/*.o | grep -v "/engine\|/duo_\|/raft4_\|/txn8_")
hipcc --offload-arch=gfx950 -shared -fPIC -o maelstrom_amd/libmaelsim_wprof.so $OBJS maelstrom_amd/build/engine_wprof.o -ldl
hipcc --offload-arch=gfx950 -shared -fPIC -o maelstrom_amd/libmaelsim_wnt.so $OBJS maelstrom_amd/build/engine_wnt.o -ldl
echo built
