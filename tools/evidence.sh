#!/bin/bash
# Runs every measurement DESIGN.md quotes outside the headline bench and writes one JSON line each to $1 (default
# gpurun_out/evidence.jsonl); copy the file to profiles/ after a GPU run.  ~3 minutes on an MI355X.
OUT=${1:-gpurun_out/evidence.jsonl}
mkdir -p "$(dirname "$OUT")"; : > "$OUT"
run() { echo "{\"tool\": \"$1\", \"args\": \"${*:2}\", \"result\": $(timeout 400 python tools/$1 "${@:2}" 2>/dev/null | tail -1 || echo null)}" >> "$OUT"; }
run fft_tune.py 20 347 5             # BASELINE config #2: 347 columns x 2^20, LDE (K3+K4) and Merkle commit (K5), GB/s against both ceilings
run fft_tune.py 22 347 3             # the same at the headline height
run cp_bench.py 22                 # recorded constraints: interpreter vs hiprtc JIT, 438 columns x 2^23 rows
run session_bench.py 22            # headline prove through the recorded-AIR session vs nx_prove_synth (same bytes)
run logup_bench.py 22              # logup kernels
run wide_logup_bench.py 20 32      # a wide logup AIR end to end through the session
run upload_bench.py 22 64          # host trace hand-over
run many_components.py             # prover2-shaped statement (55 components)
run concurrent_proves.py 22 2      # throughput mode
for mode in 0 1; do echo "{\"tool\": \"bench.py\", \"args\": \"--hash-mode $mode --no-cpu-baseline --no-v1-shaped\", \"result\": $(timeout 300 python bench.py --hash-mode $mode --no-cpu-baseline --no-v1-shaped 2>/dev/null | tail -1)}" >> "$OUT"; done
echo "{\"tool\": \"bench.py\", \"args\": \"--legacy-synth --no-cpu-baseline\", \"result\": $(timeout 300 python bench.py --legacy-synth --no-cpu-baseline 2>/dev/null | tail -1)}" >> "$OUT"
cat "$OUT" | cut -c 1-400
