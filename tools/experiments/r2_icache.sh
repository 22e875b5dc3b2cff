#!/bin/bash
# Instruction-cache counters of the fused kernels: product library (two workgroups per CU) and the one-workgroup build.
set -o pipefail
out=gpurun_out/icache; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
args="bench.py --batch 16384 --L 32 --T 20 --no-cpu-baseline --no-profile --fp32-lattices 0"
for v in product onewg; do
  lib=$PWD/diffusion-model_amd/isingdiff/libisingdiff.so
  [ $v = onewg ] && lib=$PWD/diffusion-model_amd/isingdiff/abl/lib_onewg.so
  export ISINGDIFF_LIB=$lib
  timeout -s KILL 150 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQ_IFETCH SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $out/$v -o s --output-format csv -- python3 $args > $out/$v.log 2>&1 || { tail -5 $out/$v.log; exit 1; }
  echo "== $v"; python3 tools/pmc_summary.py $out/$v/s_counter_collection.csv | grep -A9 "fused2" | cut -c1-160
done
