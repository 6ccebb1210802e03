cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06b; mkdir -p $O
for W in headline keccakw v1; do
  rm -rf /tmp/kt_$W
  timeout 600 rocprofv3 --kernel-trace -d /tmp/kt_$W -o kt -- python $R/tools/prove_loop.py $W --steps 2 > $O/loop_prof_$W.json 2>/dev/null
  python $R/tools/kernel_sequence.py $(find /tmp/kt_$W -name '*_results.db' | head -1) $O/seq_nostats_$W.txt
  python $R/tools/prove_loop.py $W --steps 4 > $O/loop_$W.json 2>/dev/null
  head -14 $O/seq_nostats_$W.txt | cut -c1-150; cat $O/loop_$W.json
done
