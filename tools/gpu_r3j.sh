#!/bin/bash
# round 3, call j: wide clusters with the nodes' sets in LDS (SETL): parity, cfg3 / wide broadcast timings with and without; duo profiles
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3j; mkdir -p $O
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "wide" --timeout 600 > $O/wide_parity.log 2>&1; tail -3 $O/wide_parity.log
timeout 400 python -m pytest tests/test_fuzz_gpu.py -m gpu -q -x -k "wide" --timeout 400 > $O/wide_fuzz.log 2>&1; tail -3 $O/wide_fuzz.log
for c in "cfg3 g-set n=100 lat100 exponential" "cfg3 g-set n=100 lat100 exponential p_loss 0.05" "broadcast n=100 grid lat0" "broadcast n=100 grid lat100 exponential"; do
  timeout 300 python tools/bench_configs.py "$c" >> $O/wide_setl.jsonl 2>> $O/wide.err
  MSIM_DEV_FLAGS=0x4000 timeout 300 python tools/bench_configs.py "$c" >> $O/wide_hbm.jsonl 2>> $O/wide.err
done
echo "-- SETL"; cut -c1-330 $O/wide_setl.jsonl; echo "-- HBM sets"; cut -c1-330 $O/wide_hbm.jsonl; tail -3 $O/wide.err
MSIM_LIB=maelstrom_amd/libmaelsim_prof.so timeout 200 python tools/duo_prof_report.py > $O/duo_prof_lat0.txt 2>&1; cat $O/duo_prof_lat0.txt
MSIM_LIB=maelstrom_amd/libmaelsim_p3.so GENERAL=1 timeout 200 python tools/duo_prof2_report.py > $O/duo_prof3_lat0.txt 2>&1; cat $O/duo_prof3_lat0.txt
MSIM_LIB=maelstrom_amd/libmaelsim_p2.so timeout 200 python tools/duo_prof2_report.py > $O/duo_prof2_lat0.txt 2>&1; cat $O/duo_prof2_lat0.txt
MSIM_LIB=maelstrom_amd/libmaelsim_prof.so LAT=100 DIST=exponential timeout 200 python tools/duo_prof_report.py > $O/duo_prof_exp100.txt 2>&1; cat $O/duo_prof_exp100.txt
MSIM_LIB=maelstrom_amd/libmaelsim_p2.so LAT=100 DIST=exponential timeout 200 python tools/duo_prof2_report.py > $O/duo_prof2_exp100.txt 2>&1; cat $O/duo_prof2_exp100.txt
MSIM_LIB=maelstrom_amd/libmaelsim_p3.so GENERAL=1 LAT=100 DIST=exponential timeout 200 python tools/duo_prof2_report.py > $O/duo_prof3_exp100.txt 2>&1; cat $O/duo_prof3_exp100.txt
