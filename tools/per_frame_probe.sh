#!/bin/bash
# the per-frame call (vpt_render + sync) under different persistent-grid sizes / batching thresholds
cd ${GRAFT_REPO_ROOT:-.}
for bpc in 3 2 1; do
VPT_BLOCKS_PER_CU=$bpc python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --frames 128 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); p=d['per_frame']; print('blocks per CU $bpc: per frame %.4f ms (%s)  batch step %.3f ms' % (p['ms_per_frame'], p['kernels_ms_last_frame'], d['ms_per_step']))"
done
