#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4p
mkdir -p $O
timeout 300 python -m pytest tests/test_order_gpu.py -x -q -k "dense" > $O/tests_dense.txt 2>&1
tail -3 $O/tests_dense.txt
if grep -q "passed" $O/tests_dense.txt && ! grep -q "failed" $O/tests_dense.txt; then
echo "== asm chain";  timeout 200 python tools/step_probe.py 2>&1 | tail -2
echo "== C++ chain";  ULTRA_AMD_LIB=ultra_amd/lib/variants/libultra_amd_dolcpp.so timeout 200 python tools/step_probe.py 2>&1 | tail -2
echo "== asm chain";  timeout 200 python tools/step_probe.py 2>&1 | tail -2
timeout 100 python tools/dense_order_probe.py 2>&1 | tail -6
ULTRA_AMD_LIB=ultra_amd/lib/variants/libultra_amd_dolcpp.so timeout 100 python tools/dense_order_probe.py 2>&1 | tail -6
timeout 900 python -m pytest tests -m gpu -x -q > $O/tests_all.txt 2>&1
tail -3 $O/tests_all.txt
fi
