#!/bin/bash
# Dynamic vector-ALU instruction mix of the fused kernels (rocprofv3 SQ_INSTS_VALU_* categories).
set -o pipefail
out=gpurun_out/pmc; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 -L > $out/counters_list.txt 2>&1 || true
grep -o "SQ_INSTS_VALU[A-Z0-9_]*" $out/counters_list.txt | sort -u | tr '\n' ' '; echo
args="bench.py --batch 16384 --T 20 --no-cpu-baseline --no-profile --fp32-lattices 0"
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_MFMA -d $out/mix -o s --output-format csv -- python3 $args > $out/mix.log 2>&1 || { tail -5 $out/mix.log; exit 1; }
python3 tools/pmc_summary.py $out/mix/s_counter_collection.csv | grep -A9 fused2
