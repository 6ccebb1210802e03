cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06b; mkdir -p $O
for L in 16 18; do
  rm -rf /tmp/kt_$L
  timeout 600 rocprofv3 --kernel-trace -d /tmp/kt_$L -o kt -- python $R/tools/prove_loop.py headline --log-rows $L --steps 3 > /dev/null 2>&1
  python $R/tools/kernel_sequence.py $(find /tmp/kt_$L -name '*_results.db' | head -1) $O/seq_small_$L.txt
  python $R/tools/prove_loop.py headline --log-rows $L --steps 20
  NX_HOST_PROF=1 python $R/tools/prove_loop.py headline --log-rows $L --steps 1 2> $O/hp_small_$L.txt > /dev/null
  head -3 $O/seq_small_$L.txt | cut -c1-150
done
