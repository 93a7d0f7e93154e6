#!/bin/bash
# Developer tool (GPU box): the whole GPU suite under fenced slabs (csrc/guard.cpp), one worker process that is replaced when a test faults
# (pytest-xdist); a test during which a byte outside a slab was overwritten FAILS (tests/conftest.py).
OUT=$1; mkdir -p $OUT
for mode in 1 2; do
  MSIM_GUARD=$mode timeout 1200 python3 -m pytest tests -m gpu -q -n 1 -p no:cacheprovider > $OUT/suite_mode$mode.log 2>&1
  echo "suite mode $mode rc=$?"; grep -i "crashed\|fault\|passed\|failed\|msim guard\|did not land" $OUT/suite_mode$mode.log | cut -c1-250 | head -20
done
# the guard must see what it is there for: a deliberate one-word overrun behind the check slab (MSIM_GUARD_SELFTEST, csrc/guard.cpp)
for mode in 1 2 3; do
  MSIM_GUARD=$mode python3 -c "
import ctypes as C
from maelstrom_amd import _abi
lib = _abi.load()
lib.msim_guard_selftest.restype = C.c_int
print('guard selftest, MSIM_GUARD=$mode:', lib.msim_guard_selftest())
" > $OUT/selftest_mode$mode.log 2>&1
  echo "selftest mode $mode rc=$?"; tail -3 $OUT/selftest_mode$mode.log
done
