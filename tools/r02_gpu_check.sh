#!/bin/bash
# GPU box: the whole -m gpu suite, then the default bench line (what the driver runs)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -25
python bench.py > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err; tail -3 gpurun_out/r02_bench_default.err; cat gpurun_out/r02_bench_default.json
