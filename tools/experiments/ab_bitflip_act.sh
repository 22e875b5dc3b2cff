#!/bin/bash
# Bit-flip sampler, bf16 mode: half activations with folded GroupNorms (default) against float32 activations with the separate
# GroupNorm kernels (ISINGDIFF_BITFLIP_ACT=f32), same box: throughput at three batch sizes, then the errors against the float32
# oracle.  bash tools/experiments/ab_bitflip_act.sh
for n in 4096 16384 65536; do
  for act in f16 f32; do
    echo "== $n images, activations $act"
    ISINGDIFF_BITFLIP_ACT=$act timeout -k 10 300 python3 tools/bench_bitflip.py $n bf16 || exit 1
  done
done
for act in f16 f32; do
  echo "== errors against the float32 oracle, activations $act"
  ISINGDIFF_BITFLIP_ACT=$act timeout -k 10 200 python3 tools/experiments/bitflip_errors.py || exit 1
done
