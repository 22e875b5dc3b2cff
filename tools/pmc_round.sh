#!/bin/bash
# SQ / GRBM counter passes for the fused band kernels on a short slice of the bench (16 384 lattices, 20 steps):
# MFMA-busy, vector-ALU, wait and LDS-conflict shares per kernel, and the clock each kernel ran at.
# PMC_CMD replaces the profiled python command (e.g. "tools/bench_fft.py 32 64"), PMC_OUT the output directory.
# Each pass is its own rocprofv3 run (at most 8 SQ counters per pass).  Outputs under gpurun_out/pmc/; the summary
# printed by tools/pmc_report.py is what gets copied to profiles/.
set -o pipefail
out=${PMC_OUT:-gpurun_out/pmc}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
args="${PMC_CMD:-bench.py --batch ${PMC_BATCH:-16384} --L ${PMC_L:-32} --T 20 --no-cpu-baseline --no-profile --fp32-lattices 0}"
pass() { name=$1; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" -d $out/$name -o s --output-format csv -- python3 $args > $out/$name.log 2>&1 || { tail -5 $out/$name.log; exit 1; }
  echo "pass $name done"; }
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
pass b SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
python3 tools/pmc_report.py $out/a/s_counter_collection.csv $out/b/s_counter_collection.csv > $out/report.txt
cat $out/report.txt
