#!/bin/bash
# rocprofv3 SQ counter passes (counters only, one pass each): where the waves of each kernel spend their cycles, for one decode of
# BASELINE configs[3] and for one training step of configs[1].  Usage (GPU box): bash tools/profile_sq.sh r03
set -x
tag=${1:-r03}
out=gpurun_out/${tag}_sq
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CTRS="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU"
rocprofv3 --pmc $CTRS -d $out/dsq -o dsq -- python3 bench.py --only-decode --no-cpu-baseline > $out/dsq.log 2>&1
db=$(find $out/dsq -name "*.db" | head -1)
if [ -z "$db" ]; then
  CTRS="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"
  rocprofv3 --pmc $CTRS -d $out/dsq -o dsq -- python3 bench.py --only-decode --no-cpu-baseline > $out/dsq.log 2>&1
  db=$(find $out/dsq -name "*.db" | head -1)
fi
python3 tools/rocprof_summary.py sq $db $out/${tag}_decode_sq.csv rvq_kernelI6bf16_t && cut -c1-240 $out/${tag}_decode_sq.csv | head -14
rocprofv3 --pmc $CTRS -d $out/tsq -o tsq -- python3 bench.py --steps 2 --warmup 1 --no-decode --no-cpu-baseline > $out/tsq.log 2>&1
db=$(find $out/tsq -name "*.db" | head -1)
python3 tools/rocprof_summary.py sq $db $out/${tag}_step_sq.csv && cut -c1-240 $out/${tag}_step_sq.csv | head -30
rm -rf $out/dsq $out/tsq
