#!/bin/bash
# round 3, call d: check_kernel with the data-parallel pairing pass and double-buffered bitmap loads
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3d; mkdir -p $O
timeout 600 python -m pytest tests/test_set_full_batch_gpu.py tests/test_checker_gpu.py tests/test_parity_gpu.py -m gpu -q -x --timeout 600 > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -3
timeout 300 python bench.py --steps 10 --warmup 2 > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json; d=json.load(open("gpurun_out/r3d/bench.json")); print({k:d[k] for k in ("value","ms_per_step","histories_per_sec","kernel_ms")})
PY
