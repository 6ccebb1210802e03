cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/v1
for e in "" "NX_FFT_BATCH=1" "NX_FFT_BATCH=1 NX_FFT_STREAMS=3" "NX_FFT_BATCH=4" "NX_FFT_STREAMS=1"; do
  echo "== env: $e"
  env $e timeout 300 python bench.py --no-cpu-baseline --steps 3 2>/dev/null | tail -1 | python -c "
import sys,json; r=json.loads(sys.stdin.read()); v=r['config_v1_shaped']; print('headline ms', round(r['ms_per_step'],2), 'lde', round(r['roofline']['kernel_ms'],2), '| v1 ms', round(v['ms_per_step'],1), 'stages', v['stages_ms'], 'lde', round(v['roofline']['kernel_ms'],1))"
done
