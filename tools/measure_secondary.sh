#!/bin/bash
# Measurements of everything besides the headline bench, in one box session: large lattices (kernel lists and
# rocprofv3 kernel stats at L = 64 / 128), the spectral UNet, the dense spectral model, the MNIST bit-flip sampler,
# the irfft2 kernel and the small HBM-bound kernels.  Outputs under gpurun_out/secondary/ (copied to profiles/).
set -o pipefail
out=gpurun_out/secondary; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { name=$1; shift; timeout -k 10 300 "$@" > $out/$name.json 2> $out/$name.err || { echo "FAILED $name"; tail -5 $out/$name.err; exit 1; }; echo "$name done"; }
run bench_L64 python3 -u bench.py --L 64 --batch 8192 --T 20 --no-cpu-baseline --fp32-lattices 0
run bench_L128 python3 -u bench.py --L 128 --batch 2048 --T 20 --no-cpu-baseline --fp32-lattices 0
run spectral python3 -u tools/bench_spectral.py 4096
run dense python3 -u tools/bench_dense.py 65536
run bitflip python3 -u tools/bench_bitflip.py 4096
run fft python3 -u tools/bench_fft.py 20 32 64 128
run elementwise python3 -u tools/bench_elementwise.py
for L in 64 128; do
  b=$([ $L = 64 ] && echo 8192 || echo 2048)
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_L$L -o s -- python3 bench.py --L $L --batch $b --T 20 --no-cpu-baseline --no-profile --fp32-lattices 0 > $out/trace_L$L.log 2>&1 || { tail -5 $out/trace_L$L.log; exit 1; }
  echo "trace L=$L done"
done
ls $out
