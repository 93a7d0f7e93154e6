#!/bin/bash
# Developer tool: builds maelstrom_amd/libmaelsim_prof.so = the library with duo.hip compiled -DDUO_PROF (wave-round counts and
# cycle counters of the two round bodies written into msim_inst_meta); use it with MSIM_LIB=.../libmaelsim_prof.so
set -e
cd "$(dirname "$0")/.."
python -m maelstrom_amd.build > /dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DDUO_PROF -c -o maelstrom_amd/build/duo_prof.o maelstrom_amd/csrc/duo.hip
OBJS=$(ls maelstrom_amd/build/*.o | grep -v "duo.hip.o\|duo_prof.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o maelstrom_amd/libmaelsim_prof.so $OBJS maelstrom_amd/build/duo_prof.o -ldl
echo built maelstrom_amd/libmaelsim_prof.so
