#!/bin/bash
# The split-role fused_B build against the product library: bitwise comparison at several batch sizes, slot timeline, timing.
set -o pipefail
P=diffusion-model_amd/isingdiff
for b in ${CMP_BATCHES:-5 777 4096}; do
  timeout -k 10 200 python -u tools/experiments/compare_libs_batch.py $P/abl/lib_bsplit.so $P/libisingdiff.so $b 3 2>/dev/null || exit 1
done
[ -f $P/abl/lib_TLbsplit.so ] && { ISINGDIFF_LIB=$PWD/$P/abl/lib_TLbsplit.so timeout -k 10 100 python -u tools/timeline3.py 2>/dev/null || exit 1; }
rm -f $P/abl/lib_TLbsplit.so
bash tools/bench_variants.sh
