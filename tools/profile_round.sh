#!/bin/bash
# One gpurun call that produces everything profiles/rNN_<tag>_* is made of (run from the repo root on the GPU box):
#   bash tools/profile_round.sh <tag> [dtype]
# 1. default bench line; 2. serial bench under rocprofv3 --kernel-trace --stats (tile table from a file so that no
# tuning pass sits in the trace); 3. three separate --pmc passes (FETCH_SIZE / WRITE_SIZE / MFMA busy) of the same
# serial command; 4. per-layer table.  Summaries are copied to profiles/ by tools/collect_profiles.py afterwards.
TAG=${1:-x}; DT=${2:-f16x3}
R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O
export HMMR_TILE_CACHE=/tmp/tiles_$TAG.json
cd /tmp && export TMPDIR=/tmp
SER="python $R/bench.py --dtype $DT --serial --only-main --no-cpu-baseline --no-pcie --no-by-config --no-power --sustain 0 --steps 6 --warmup 2"
python $R/bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err      # the driver's command (dtype auto)
$SER > $O/bench_serial.json 2>> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- $SER > $O/bench_serial_under_rocprof.json 2>> $O/rocprof.err
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES"; do
  n=$(echo $c | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$n -- timeout 600 python $R/bench.py --dtype $DT --serial --only-main --no-cpu-baseline --no-pcie --no-by-config --no-power --sustain 0 --steps 2 --warmup 1 > /dev/null 2>> $O/rocprof.err
done
cd $R
python tools/layer_table.py 257 $DT 5 > $O/layer_table_$DT.log 2>&1
# keep the merge small: kernel stats + per-kernel counter sums only
python tools/pmc_summary.py $O $O/pmc_summary.json $DT > $O/pmc_summary.log 2>&1
find $O -name "*kernel_trace.csv" -size +20M -delete
find $O -name "*counter_collection.csv" -delete
ls -la $O $O/prof/* | head -40
cat $O/bench.json; cat $O/layer_table_$DT.log; cat $O/pmc_summary.log
