set -x
out=gpurun_out/r03_gaps
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d $out/trace -o trace -- python3 bench.py --steps 4 --warmup 2 --no-decode --no-cpu-baseline > $out/trace.log 2>&1
python3 tools/rocprof_summary.py gaps $(find $out/trace -name "*.db" | head -1) $out/r03_step_gaps.json
rm -rf $out/trace
