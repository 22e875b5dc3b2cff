cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in base w0; do
  if [ $v = base ]; then export ISINGDIFF_LIB=$PWD/diffusion-model_amd/isingdiff/libisingdiff.so; else export ISINGDIFF_LIB=$PWD/diffusion-model_amd/isingdiff/abl/lib_w0.so; fi
  timeout -s KILL 150 rocprofv3 --pmc GRBM_GUI_ACTIVE -d gpurun_out/clk_$v -o s --output-format csv -- python3 bench.py --batch 4096 --T 200 --no-cpu-baseline --no-profile > gpurun_out/clk_$v.log 2>&1 || exit 1
  echo $v; python3 tools/pmc_clock.py gpurun_out/clk_$v/s_counter_collection.csv | grep fused2
done
