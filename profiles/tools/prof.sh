#!/bin/bash
# usage: prof.sh <tag> <bench args...>   -> gpurun_out/prof_<tag>/{stats,fetch,write,sq}
set -o pipefail
R=$GRAFT_REPO_ROOT; TAG=$1; shift
export TMPDIR=/tmp; cd /tmp
OUT=$R/gpurun_out/prof_$TAG; mkdir -p $OUT
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py "$@" --no-cpu-baseline --no-extra > $OUT/bench_stats.log 2>&1 || echo "stats run failed"
timeout -k 10 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py "$@" --no-cpu-baseline --no-extra > $OUT/bench_fetch.log 2>&1 || echo "fetch run failed"
timeout -k 10 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py "$@" --no-cpu-baseline --no-extra > $OUT/bench_write.log 2>&1 || echo "write run failed"
timeout -k 10 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/sq -- python3 $R/bench.py "$@" --no-cpu-baseline --no-extra > $OUT/bench_sq.log 2>&1 || echo "sq run failed"
timeout -k 10 150 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/sq2 -- python3 $R/bench.py "$@" --no-cpu-baseline --no-extra > $OUT/bench_sq2.log 2>&1 || echo "sq2 run failed"
timeout -k 10 150 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/grbm -- python3 $R/bench.py "$@" --no-cpu-baseline --no-extra > $OUT/bench_grbm.log 2>&1 || echo "grbm run failed"
find $OUT -name "*.csv" | head -30
