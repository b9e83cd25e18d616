#!/bin/bash
# First GPU call of the next round (run from the repo root on the GPU box, ~3 min): does the chunk-major K order of the
# 3x3 layers (hmmr_conv_desc_t.k_order = 1) compute the same numbers, and does it pay?
#   bash tools/next_round_first_call.sh > gpurun_out/kcm.log 2>&1
R=$PWD; O=$R/gpurun_out/kcm; mkdir -p $O
echo "== 1. equal to tap-major up to rounding, every tile"
timeout 120 python tools/kcm_check.py 16 2>&1 | grep -v amdgpu.ids
echo "== 2. per-shape timing, tap-major vs chunk-major (tiles 5 7 8; ms / TFLOP/s)"
for k in 0 1; do echo "KORD=$k"; KORD=$k timeout 120 python tools/conv_bench.py 257 bf16x3 5,7,8 conv2 2>&1 | grep "^{"; done
echo "== 3. L2 hit rate and fabric reads of the same launches (one PMC pass each)"
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
for k in 0 1; do
  rm -rf $O/pmc_$k
  KORD=$k timeout 150 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum FETCH_SIZE --output-format csv -d $O/pmc_$k -- \
      python $R/tools/conv_bench.py 257 bf16x3 7,8 conv2 > $O/pmc_$k.log 2>&1
  python - $O/pmc_$k <<'PY'
import csv, glob, sys, collections
fs = sorted(glob.glob(sys.argv[1] + "/*/*counter_collection.csv"))
acc = collections.OrderedDict()
for r in csv.DictReader(open(fs[-1])) if fs else []:
    if "conv_gemm_kernel" not in r["Kernel_Name"]:
        continue
    e = acc.setdefault((r["Kernel_Name"][:90], r["Grid_Size"]), collections.defaultdict(float))
    e[r["Counter_Name"]] += float(r["Counter_Value"]); e["n"] += 1.0 / 3
for (nm, grid), c in acc.items():
    hit, miss = c["TCC_HIT_sum"], c["TCC_MISS_sum"]
    print("%s grid %s: L2 hit %.1f %%, fetch %.0f MB/launch (x2-corrected)" % (nm[-40:], grid, 100 * hit / max(hit + miss, 1), 2 * c["FETCH_SIZE"] * 1024 / 1e6 / max(c["n"], 1)))
PY
done
cd $R
echo "== 4. whole pipeline: parity tests and the bench line with the chunk-major filters"
HMMR_CHUNK_MAJOR=1 timeout 300 python -m pytest tests/test_gpu_bf16x3.py tests/test_gpu_sizes.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
for k in 0 1; do
  HMMR_CHUNK_MAJOR=$k timeout 120 python bench.py --only-main --no-cpu-baseline --no-pcie --steps 20 2>/dev/null | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('HMMR_CHUNK_MAJOR=$k', d['value'], d['ms_per_step'], d['roofline']['resnet_pass_ms'], d.get('e2e_verts_max_abs_err'))"
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +8M -delete
