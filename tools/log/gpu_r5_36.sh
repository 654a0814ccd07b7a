#!/bin/bash
# After the record-request clamp / padding fix: the sequence that aborted, then the full GPU suite.
O=gpurun_out/r5
mkdir -p $O
timeout 600 python -m pytest tests/test_models_gpu.py tests/test_order_gpu.py -q -x -m gpu > $O/combo.txt 2>&1
tail -2 $O/combo.txt | cut -c1-200
timeout 1200 python -m pytest tests -m gpu -q > $O/r5_gpu_tests.txt 2> $O/r5_gpu_tests.err
tail -3 $O/r5_gpu_tests.txt | cut -c1-300
tail -3 $O/r5_gpu_tests.err | cut -c1-300
