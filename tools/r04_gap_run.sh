#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04gap; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
(cd /tmp && timeout 300 rocprofv3 --hip-trace --kernel-trace -d $R/$O/kt -o kt -- python $R/bench.py --no-cpu-baseline --no-v1-shaped --no-host-trace --steps 4 > /dev/null 2> $R/$O/err.txt)
ls $O/kt
python tools/gap_api.py $O/kt/kt_results.db 15 > $O/gap_api.txt 2>&1
python tools/trace_gaps.py $O/kt/kt_results.db 25 2 > $O/trace_gaps.txt 2>&1
rm -rf $O/kt
head -50 $O/gap_api.txt
