#!/bin/bash
# rocprofv3 passes for the bench command (run on the GPU box through gpurun). Summaries land in gpurun_out/prof/.
# Counters are collected in their own passes, without any tracing (see MI355X_MICROARCH.md §rocprofv3 PMC slots).
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=${PROF_OUT:-gpurun_out/prof}
rm -rf $OUT; mkdir -p $OUT
CMD="python bench.py --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline --no-other-configs ${BENCH_ARGS:-}"
echo "== kernel trace + stats: $CMD"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- $CMD > $OUT/kt.log 2>&1
pmc() { # name counters...
  local name=$1; shift
  echo "== pmc $name: $*"
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o $name -- $CMD > $OUT/$name.log 2>&1
}
PASSES=${PASSES:-"fetch write sq1 sq2 grbm"}   # e.g. PASSES="fetch write" for a 1 B-row workload (every pass regenerates the data)
for p in $PASSES; do
  case $p in
    fetch) pmc fetch FETCH_SIZE ;;
    write) pmc write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum ;;
    sq1) pmc sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD ;;
    sq2) pmc sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT ;;
    grbm) pmc grbm GRBM_GUI_ACTIVE ;;
  esac
done
find $OUT -name "*.csv" | head -40
python tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
