#!/bin/bash
# The round's measurements in one box session (about 7 minutes): the headline bench with its rocprofv3 kernel stats and
# HBM counter passes (tools/measure_round.sh), the SQ and vector-memory counter passes of the fused kernels
# (tools/pmc_round.sh, tools/pmc_mem.sh), the full-schedule runs of L = 64 and L = 128 and the spectral / bit-flip /
# dense workloads of bench.py.  Outputs under gpurun_out/measure/, gpurun_out/pmc/, gpurun_out/pmc_mem/ and
# gpurun_out/workloads/; the summaries are copied to profiles/ by hand.
set -o pipefail
bash tools/measure_round.sh || exit 1
bash tools/pmc_round.sh > /dev/null || exit 1
echo "sq passes done"
bash tools/pmc_mem.sh > /dev/null || exit 1
echo "mem pass done"
out=gpurun_out/workloads; mkdir -p $out
run() { name=$1; shift; timeout -k 10 300 python3 -u bench.py "$@" > $out/$name.json 2> $out/$name.err || { echo "FAILED $name"; tail -5 $out/$name.err; exit 1; }; echo "$name done"; }
run bench_L64_T1000 --L 64 --batch 8192 --no-cpu-baseline --fp32-lattices 0
run bench_L128_T1000 --L 128 --batch 2048 --no-cpu-baseline --fp32-lattices 0
run bench_spectral_L32_T1000 --workload spectral
run bench_bitflip_T100 --workload bitflip
run bench_dense_T1000 --workload dense --steps 5
ls $out
