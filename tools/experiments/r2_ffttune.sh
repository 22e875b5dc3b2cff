#!/bin/bash
# irfft2 kernel shapes: a tuning library (csrc/idm_fft.hip built with -DIDM_FFT_TUNE) picks a candidate per ISINGDIFF_FFT_CFG.
set -o pipefail
export ISINGDIFF_LIB=$PWD/diffusion-model_amd/isingdiff/abl/lib_ffttune.so
mkdir -p gpurun_out/fft
for c in 0 1 2 3 4; do
  ISINGDIFF_FFT_CFG=$c timeout -k 10 120 python3 -u tools/bench_fft.py 20 32 64 > gpurun_out/fft/cfg$c.json 2> gpurun_out/fft/cfg$c.err || { echo "cfg $c failed"; tail -5 gpurun_out/fft/cfg$c.err; exit 1; }
  echo "cfg $c: $(cat gpurun_out/fft/cfg$c.json)"
done
ISINGDIFF_FFT_CFG=1 timeout -k 10 300 python3 -u -m pytest tests -m gpu -x -q -k "irfft or fft" --timeout 120 --timeout-method thread 2>&1 | tail -3
