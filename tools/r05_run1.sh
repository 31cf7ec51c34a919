#!/bin/bash
# round 5, GPU call 1: the whole GPU suite on the new default library, then A/B against the round-4 library and the study variants
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05_run1; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -40 ) > $O/pytest.txt
echo "== A/B" > $O/ab.txt
STEPS=6 bash tools/variants_bench.sh c2 64 r4base default nt nosuninv r4base default >> $O/ab.txt 2>&1
STEPS=4 bash tools/variants_bench.sh c3 64 r4base default nt ntq >> $O/ab.txt 2>&1
STEPS=3 bash tools/variants_bench.sh c5 32 r4base default nt >> $O/ab.txt 2>&1
STEPS=3 bash tools/variants_bench.sh c4 16 r4base default ntq nt >> $O/ab.txt 2>&1
( timeout 600 python tools/c5_parity_probe.py 2 2>&1 | grep -v amdgpu.ids | tail -5 ) > $O/c5_p99.txt
cat $O/pytest.txt | tail -15; cat $O/ab.txt; cat $O/c5_p99.txt
