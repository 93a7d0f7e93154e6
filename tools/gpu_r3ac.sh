#!/bin/bash
# round 3, call ac: hat8_kernel (sixteen / eight txn-rw-register clusters per wavefront): parity on the device, the two bench shapes against hat_kernel<>
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3ac; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py -k "txn_rw_register" -m gpu -q -x --timeout 800 > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 600 python -m pytest tests/test_rw_check_gpu.py tests/test_bench_shapes_gpu.py -m gpu -q -x --timeout 500 > $O/tests2.log 2>&1; tail -2 $O/tests2.log
for f in 0 0x200; do
  MSIM_DEV_FLAGS=$f timeout 600 python tools/bench_configs.py "txn-rw-register hat n=2 rate100 30s + partitions" "txn-rw-register hat n=5 rate100 30s lat5 + partitions" > $O/hat_$f.jsonl 2> $O/hat_$f.err
  echo "flags $f"; cut -c1-420 $O/hat_$f.jsonl; tail -2 $O/hat_$f.err
done
