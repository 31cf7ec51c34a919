cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05_run15; mkdir -p $O
L=$PWD/volumetric-path-tracer_amd
( timeout 900 python -m pytest tests/test_gpu_bench_ranks.py -q -k "single_rank or two_ranks" 2>&1 | grep -v amdgpu.ids | tail -5 ) > $O/pytest.txt
b1() { local name=$1 cfg=$2 spp=$3 steps=$4; shift 4; env "$@" python bench.py --config $cfg --spp $spp --no-cpu-baseline --no-other-configs --no-per-frame --steps $steps --warmup 2 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-16s %s spp%s: %9.1f Msamples/s  step %8.3f ms  raygen %7.3f trace %8.3f tail %7.3f' % ('$name', '$cfg', '$spp', d['value'], d['ms_per_step'], r['raygen_ms_per_step'], r['trace_ms_per_step'], r['tail_resolve_ms_per_step']))"; }
{ for rep in 1 2; do b1 default c2 64 10 VPT_X=1; b1 no-raygen-pushes c2 64 10 VPT_LIB_PATH=$L/libvpt_hip_nopush.so; done; } > $O/ab.txt 2>&1
cat $O/pytest.txt; cat $O/ab.txt
