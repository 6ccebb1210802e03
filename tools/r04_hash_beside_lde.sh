# Does leaf hashing BESIDE the LDE pay once the FFT leaves room for its waves?  (62 % of the LDE's time is data
# movement during which the VALUs are under-used; Blake2s is pure VALU work.)  NX_PIPE_COLS: leaf hashing of finished column groups on the hash
# stream next to the next group's LDE; NX_FFT_LDS_EXTRA: extra dynamic LDS per FFT block = fewer FFT blocks per CU.
# usage: bash tools/r04_hash_beside_lde.sh out.jsonl
out=${1:-gpurun_out/r04_hash_beside_lde.jsonl}
: > "$out"
for cfg in "0:0" "32:0" "64:0" "32:16384" "64:16384" "32:40000" "64:40000" "0:16384" "0:0"; do
  pc=${cfg%%:*}; ex=${cfg#*:}
  NX_PIPE_COLS=$pc NX_FFT_LDS_EXTRA=$ex python bench.py --no-cpu-baseline --no-v1-shaped --no-host-trace --steps 8 --warmup 2 | python -c "
import sys, json
r = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(json.dumps({'NX_PIPE_COLS': $pc, 'NX_FFT_LDS_EXTRA': $ex, 'ms_per_prove': round(r['ms_per_step'], 3), 'commit_ms': r['stages_ms']['commit'], 'lde_kernel_ms': round(r['roofline']['kernel_ms'], 3), 'merkle_kernel_ms': round(r['merkle']['kernel_ms'], 3)}))" >> "$out"
done
cat "$out"
