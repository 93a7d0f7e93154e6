#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r2j}; mkdir -p $O
for n in 64 1024 4096; do
  N=$n LAT=100 DIST=exponential timeout 300 python tools/duo_prof_report.py > $O/prof_exp100_$n.txt 2>&1
done
N=4096 LAT=50 DIST=uniform timeout 300 python tools/duo_prof_report.py > $O/prof_uni50.txt 2>&1
cat $O/prof_*.txt
