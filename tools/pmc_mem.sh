#!/bin/bash
# Texture-addresser / vector-L1 counter pass for the fused band kernels (the path the weight fragments and the tile
# prefetches take): how busy the address and data units are and how long the L1 stalls on them.  One rocprofv3 run
# (2 TA, 4 TCP, 1 TD, 1 GRBM counter).  Output: gpurun_out/pmc_mem/report.txt (per kernel: counter sums per dispatch).
set -o pipefail
out=${PMC_OUT:-gpurun_out/pmc_mem}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
args="${PMC_CMD:-bench.py --batch ${PMC_BATCH:-16384} --L ${PMC_L:-32} --T 20 --no-cpu-baseline --no-profile --fp32-lattices 0}"
timeout -s KILL 150 rocprofv3 --pmc TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE -d $out/m -o s --output-format csv -- python3 $args > $out/m.log 2>&1 || { tail -5 $out/m.log; exit 1; }
python3 tools/pmc_summary.py $out/m/s_counter_collection.csv > $out/report.txt
cat $out/report.txt
