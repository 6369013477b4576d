set -x
tag=${1:-r04}
out=gpurun_out/${tag}_sq_f32
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU -d $out/dsq -o dsq -- python3 tools/decode_probe.py f32 > $out/dsq.log 2>&1
python3 tools/rocprof_summary.py sq $(find $out/dsq -name "*.db" | head -1) $out/${tag}_decode_f32_sq.csv rvq_x2_kernel && cut -c1-230 $out/${tag}_decode_f32_sq.csv | head -24
rm -rf $out/dsq
