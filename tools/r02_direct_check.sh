#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scenes.py tests/test_gpu_vs_ref.py tests/test_gpu_fullsize.py tests/test_gpu_edge.py tests/test_gpu_cli.py -m gpu -x -q 2>&1 | tail -8
STEPS=8 bash tools/variants_bench.sh c2 64 default
STEPS=2 bash tools/variants_bench.sh c5 512 default
