for k in 9 10 11; do for lg in 23 24; do
NX_FFT_KMAX=$k python tools/fft_tune.py $lg 32 3 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('kmax $k log $lg', 'lde_ms %.2f'%(r['lde_ms']), 'GB/s %.0f' % r['lde_alg_GBs'])"
done; done
NX_FFT_KMAX=11 python -m pytest tests -m gpu -x -q -k "large_transforms or interpolate" 2>&1 | tail -1
