#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export IDM_FFT32=1
timeout -k 10 120 python3 -u tools/bench_fft.py 20 32 64 128 > gpurun_out/r2/fft.json 2> gpurun_out/r2/fft.err || { tail -5 gpurun_out/r2/fft.err; exit 1; }
cat gpurun_out/r2/fft.json
timeout -s KILL 90 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2/fft_trace -o s -- python3 tools/bench_fft.py 32 > gpurun_out/r2/fft_trace.log 2>&1 || { tail -5 gpurun_out/r2/fft_trace.log; exit 1; }
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS -d gpurun_out/r2/fft_pmc1 -o s --output-format csv -- python3 tools/bench_fft.py 32 > gpurun_out/r2/fft_pmc1.log 2>&1 || { tail -5 gpurun_out/r2/fft_pmc1.log; exit 1; }
timeout -s KILL 90 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM -d gpurun_out/r2/fft_pmc2 -o s --output-format csv -- python3 tools/bench_fft.py 32 > gpurun_out/r2/fft_pmc2.log 2>&1 || { tail -5 gpurun_out/r2/fft_pmc2.log; exit 1; }
ls gpurun_out/r2/fft_trace gpurun_out/r2/fft_pmc1
