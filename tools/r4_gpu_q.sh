#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4q
mkdir -p $O
timeout 300 python -m pytest tests/test_order_gpu.py -x -q -k "dense" > $O/tests_dense.txt 2>&1
tail -2 $O/tests_dense.txt
for v in "" dol_notouch dol_ns7 dol_ns7_notouch dolcpp; do
  if [ -n "$v" ]; then export ULTRA_AMD_LIB=ultra_amd/lib/variants/libultra_amd_$v.so; else unset ULTRA_AMD_LIB; fi
  echo "== ${v:-default (4 stages + touch)}"
  timeout 100 python tools/dense_order_probe.py 2>&1 | tail -1
done
unset ULTRA_AMD_LIB
timeout 200 python tools/step_probe.py 2>&1 | tail -2
ULTRA_AMD_LIB=ultra_amd/lib/variants/libultra_amd_dolcpp.so timeout 200 python tools/step_probe.py 2>&1 | tail -2
