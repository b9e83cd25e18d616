#!/bin/bash
# PMC study of the unit pair alone (development aid): separate counter passes of tools/pair_check.py for one block / batch / form;
# per-dispatch averages of the pair kernel printed by the python below.   bash tools/pmc_pair_study.sh b3 257 0 <outdir>
R=$PWD; BLK=${1:-b3}; N=${2:-257}; FORM=${3:-0}; O=$R/${4:-gpurun_out/pmc_pair}_$FORM; mkdir -p $O
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
i=0
for c in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
         "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CU_CYCLES" \
         "SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_ANY SQ_WAVES SQ_WAIT_ANY" \
         "SQ_INSTS_SALU SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/p$i -- python $R/tools/pair_check.py $BLK $N $FORM > $O/p$i.log 2>&1
done
cd $R
python - "$O" <<'PY'
import csv, glob, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(root + "/p*/*/*counter_collection.csv")):
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if "unit_pair" not in r["Kernel_Name"]:
            continue
        e = disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"].split("(")[0][:60], "c": collections.defaultdict(float)})
        e["c"][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, e in disp.items():
        for cn, v in e["c"].items():
            acc[e["name"]][cn].append(v)
for name, cs in acc.items():
    print(name)
    for cn, vs in sorted(cs.items()):
        print("   %-28s %16.0f  (%d dispatches)" % (cn, sum(vs) / len(vs), len(vs)))
PY
find $O -name "*.csv" -size +5M -delete
