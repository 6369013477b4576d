#!/bin/bash
# rocprofv3 passes of the headline benchmark (run on the GPU box): kernel trace + stats (bench.py --no-replay: nothing but warm-up and
# timed steps is launched, so AverageNs is an IN-STEP average and calls / step match the PMC file), then the two HBM-side PMC passes
# (separate runs, counters only), summarised into profiles/ form by tools/rocprof_summary.py.  Usage: bash tools/profile_round.sh r02
set -x
tag=${1:-r02}
out=gpurun_out/${tag}_prof
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $out/trace -o trace -- python3 bench.py --steps 6 --warmup 2 --no-replay --no-cpu-baseline > $out/trace.log 2>&1
tail -1 $out/trace.log | cut -c1-200
rocprofv3 --pmc FETCH_SIZE -d $out/fetch -o fetch -- python3 bench.py --steps 2 --warmup 1 --no-replay --no-cpu-baseline > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $out/write -o write -- python3 bench.py --steps 2 --warmup 1 --no-replay --no-cpu-baseline > $out/write.log 2>&1
find $out -name "*.db" | head
python3 tools/rocprof_summary.py stats $(find $out/trace -name "*.db" | head -1) $out/${tag}_kernel_stats.csv
python3 tools/rocprof_summary.py pmc $(find $out/fetch -name "*.db" | head -1) $(find $out/write -name "*.db" | head -1) $out/${tag}_pmc_traffic.json
head -12 $out/${tag}_kernel_stats.csv | cut -c1-160
python3 -c "import json; d=json.load(open('$out/${tag}_pmc_traffic.json')); print(d['step_read_bytes'], d['step_write_bytes'], d['step_bytes'])"
# decode leg (BASELINE configs[3]): kernel trace + the two HBM-side PMC passes of `bench.py --only-decode`
rocprofv3 --kernel-trace --stats -d $out/dtrace -o dtrace -- python3 bench.py --only-decode --no-cpu-baseline > $out/dtrace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $out/dfetch -o dfetch -- python3 bench.py --only-decode --no-cpu-baseline > $out/dfetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $out/dwrite -o dwrite -- python3 bench.py --only-decode --no-cpu-baseline > $out/dwrite.log 2>&1
python3 tools/rocprof_summary.py stats $(find $out/dtrace -name "*.db" | head -1) $out/${tag}_decode_kernel_stats.csv
python3 tools/rocprof_summary.py pmc_decode $(find $out/dfetch -name "*.db" | head -1) $(find $out/dwrite -name "*.db" | head -1) $out/${tag}_decode_pmc.json
head -8 $out/${tag}_decode_kernel_stats.csv | cut -c1-160
rm -rf $out/trace $out/fetch $out/write $out/dtrace $out/dfetch $out/dwrite
