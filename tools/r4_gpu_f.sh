#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4f
mkdir -p $O
timeout 900 python -m pytest tests/test_launch_gpu.py tests/test_rspmm_gpu.py -x -q -k "two_ranks or aliased or real_triples or torchrun" > $O/tests_new.txt 2>&1
tail -15 $O/tests_new.txt
timeout 900 python -m pytest tests -m gpu -q > $O/tests_all.txt 2>&1
tail -6 $O/tests_all.txt
