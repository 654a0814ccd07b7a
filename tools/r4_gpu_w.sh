#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4w
mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu -k "pipelin or graph or launch or bench" > $O/tests_pipe.txt 2>&1
tail -3 $O/tests_pipe.txt
bash tools/r4_gpu_g.sh A
