cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05_run16; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bench_ranks.py -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -8 > $O/pytest.txt; cat $O/pytest.txt
