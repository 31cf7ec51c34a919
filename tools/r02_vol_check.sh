#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_vol.py tests/test_gpu_vs_ref.py tests/test_gpu_fullsize.py tests/test_gpu_cli.py -m gpu -x -q 2>&1 | tail -12
bash tools/variants_bench.sh c4 128 default
VPT_BURN_ROUNDS=0 bash tools/variants_bench.sh c4 128 default
