cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scenes.py tests/test_gpu_vs_ref.py tests/test_gpu_fullsize.py tests/test_gpu_edge.py -m gpu -x -q 2>&1 | tail -3
STEPS=8 bash tools/variants_bench.sh c2 64 base default base default
for tm in 40 32 24; do echo "trans_min $tm"; VPT_TRANS_MIN=$tm STEPS=8 bash tools/variants_bench.sh c2 64 default; done
STEPS=3 bash tools/variants_bench.sh c3 256 base default
STEPS=2 bash tools/variants_bench.sh c5 512 base default
