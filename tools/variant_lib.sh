#!/bin/bash
# Developer tool: builds maelstrom_amd/libmaelsim_<tag>.so = the product library with ONE translation unit recompiled with extra flags
# (profiling builds, A/B variants), for runs with MSIM_LIB=maelstrom_amd/libmaelsim_<tag>.so.  The variant objects live outside
# maelstrom_amd/build so that the product build never links them.
#   tools/variant_lib.sh prof   duo.hip         -DDUO_PROF        (tools/duo_prof_report.py; -DDUO_PROF2 / -DDUO_PROF3: duo_prof2_report.py)
#   tools/variant_lib.sh wprof  k_wide_gset.hip -DWIDE_PROF       (tools/wide_prof_report.py)
#   tools/variant_lib.sh m8prof mk8.hip -DM8_PROF | r4prof raft4.hip -DR4_PROF | t8prof txn8.hip -DT8_PROF   (tools/*_prof_report.py)
set -e
cd "$(dirname "$0")/.."
TAG=$1; UNIT=$2; shift 2
python -m maelstrom_amd.build > /dev/null
mkdir -p maelstrom_amd/build/variants
OBJ=maelstrom_amd/build/variants/${UNIT}_$TAG.o
# the product's own flags for this unit (maelstrom_amd/build.py FLAGS + UNIT_FLAGS), so that a variant differs from the product by "$@" only
FLAGS=$(python -c "from maelstrom_amd.build import FLAGS, UNIT_FLAGS; print(*FLAGS, *UNIT_FLAGS.get('$UNIT', []))")
hipcc $FLAGS "$@" -c -o $OBJ maelstrom_amd/csrc/$UNIT
OBJS=$(ls maelstrom_amd/build/*.o | grep -v "/$UNIT.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o maelstrom_amd/libmaelsim_$TAG.so $OBJS $OBJ -ldl
echo built maelstrom_amd/libmaelsim_$TAG.so
