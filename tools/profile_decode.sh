#!/bin/bash
# decode leg only of tools/profile_round.sh (BASELINE configs[3]): kernel trace + the two HBM-side PMC passes of `bench.py --only-decode`
set -x
tag=${1:-r04}
out=gpurun_out/${tag}_prof
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $out/dtrace -o dtrace -- python3 bench.py --only-decode --no-cpu-baseline > $out/dtrace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $out/dfetch -o dfetch -- python3 bench.py --only-decode --no-cpu-baseline > $out/dfetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $out/dwrite -o dwrite -- python3 bench.py --only-decode --no-cpu-baseline > $out/dwrite.log 2>&1
python3 tools/rocprof_summary.py stats $(find $out/dtrace -name "*.db" | head -1) $out/${tag}_decode_kernel_stats.csv
python3 tools/rocprof_summary.py pmc_decode $(find $out/dfetch -name "*.db" | head -1) $(find $out/dwrite -name "*.db" | head -1) $out/${tag}_decode_pmc.json
head -12 $out/${tag}_decode_kernel_stats.csv | cut -c1-170
cat $out/${tag}_decode_pmc.json | head -30
rm -rf $out/dtrace $out/dfetch $out/dwrite
