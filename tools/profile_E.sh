set -x
out=gpurun_out/r03_prof_E
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $out/trace -o trace -- python3 bench.py --workload E --steps 4 --warmup 2 --no-decode --no-cpu-baseline > $out/trace.log 2>&1
tail -1 $out/trace.log | cut -c1-300
python3 tools/rocprof_summary.py stats $(find $out/trace -name "*.db" | head -1) $out/r03_E_kernel_stats.csv
head -32 $out/r03_E_kernel_stats.csv | cut -c1-200
rm -rf $out/trace
