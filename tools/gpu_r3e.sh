#!/bin/bash
# round 3, call e: whole GPU suite after the check_kernel rewrite; headline profile (trace + PMC passes)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3e; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -3
timeout 500 bash tools/profile_headline.sh r3e/headline full > $O/headline_prof.log 2>&1; grep "check_kernel" $O/headline_prof.log | tail -24
