#!/bin/bash
# Strip kernels with weight ring depths 1 / 2 / 3 at L = 64 and L = 128 (libraries built with -DIDM_STRIP_WD=n).
for L in 64 128; do B=$((L==64?2048:512));
for lib in diffusion-model_amd/isingdiff/libisingdiff.so diffusion-model_amd/isingdiff/abl/lib_swd1.so diffusion-model_amd/isingdiff/abl/lib_swd2.so diffusion-model_amd/isingdiff/libisingdiff.so; do
  ISINGDIFF_LIB=$PWD/$lib timeout -k 10 300 python -u bench.py --L $L --batch $B --T 20 --steps 2 --warmup 1 --no-cpu-baseline 2>gpurun_out/l.err | python -c "import json,sys; d=json.load(sys.stdin); print('$lib'.split('/')[-1], d['config']['L'], round(d['value'],1), round(d['whole_job_mfma_frac'],4), ' '.join(k['name'].split('(')[0][-5:]+'='+str(round(k['avg_us'])) for k in d['kernels']))" || { tail -3 gpurun_out/l.err; exit 1; }
done; done
