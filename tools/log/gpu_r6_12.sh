#!/bin/bash
# round 6: the whole GPU suite on HEAD, smoke
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
git_head=$(cat .git/HEAD 2>/dev/null)
( time timeout 3000 python -m pytest tests -m gpu -q -x ) > gpurun_out/r6_gpu_tests.txt 2>&1; echo "rc $?" >> gpurun_out/r6_gpu_tests.txt
tail -8 gpurun_out/r6_gpu_tests.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r6_smoke.txt 2>&1; tail -3 gpurun_out/r6_smoke.txt
