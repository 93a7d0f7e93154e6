#!/bin/bash
# Developer tool (GPU box): cfg2's random-latency shapes with bags of 16 / 8 / 4 envelopes per node in LDS (csrc/duo.hip DUO_BAG_N; the smaller the bag,
# the more wavefronts a CU holds and the more envelopes spill to HBM), one batch at a time and with several in flight (tools/cfg2_overlap.py).
OUT=$1; mkdir -p $OUT
for b in 16 8 4; do
  if [ $b = 16 ]; then L=""; else tools/variant_lib.sh bag$b duo.hip -DDUO_BAG_N=$b > /dev/null 2>&1 || { echo "build bag$b failed"; continue; }; L=_bag$b; fi
  par=$(MSIM_LIB=$PWD/maelstrom_amd/libmaelsim$L.so timeout 900 python3 -m pytest tests/test_parity_gpu.py tests/test_bench_shapes_gpu.py -m gpu -q -x -p no:cacheprovider -k "broadcast or latency or headline or spill" 2>&1 | grep -E "passed|failed" | tail -1)
  echo "bag $b parity: $par"
  MSIM_LIB=$PWD/maelstrom_amd/libmaelsim$L.so python3 tools/cfg2_overlap.py --shapes exp100,lat10 --depths 1,2,3,4 | sed "s/^{/{\"bag\": $b, /" >> $OUT/duo_bag_ab.jsonl
done
cut -c1-230 $OUT/duo_bag_ab.jsonl
