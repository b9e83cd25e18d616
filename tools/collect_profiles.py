"""Copy the summaries of one tools/profile_round.sh run from gpurun_out/<tag>/ (scratch) into profiles/ (tracked).

    python tools/collect_profiles.py r02a
"""
import glob
import json
import os
import shutil
import subprocess
import sys

tag = sys.argv[1]
src, dst = "gpurun_out/" + tag, "profiles"
pairs = [("bench.json", "bench.json"), ("bench_serial.json", "bench_serial.json"),
         ("bench_serial_under_rocprof.json", "bench_serial_under_rocprof.json"),
         ("pmc_summary.json", "pmc_summary.json")]
for a, b in pairs:
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(dst, "%s_%s" % (tag, b)))
for f in glob.glob(src + "/pmc_*_per_kernel.csv") + glob.glob(src + "/layer_table_*.log"):
    shutil.copy(f, os.path.join(dst, "%s_%s" % (tag, os.path.basename(f))))
# stamp the commit the numbers belong to (the GPU box has no .git: the snapshot that ran there is this working tree)
pm = os.path.join(dst, "%s_pmc_summary.json" % tag)
if os.path.exists(pm):
    js = json.load(open(pm))
    head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    dirty = subprocess.run(["git", "status", "--porcelain", "--", "human_dynamics_amd/csrc", "include"], capture_output=True, text=True).stdout.strip()
    js["commit"] = head + ("+uncommitted kernel edits" if dirty else "")
    json.dump(js, open(pm, "w"), indent=1)
ks = sorted(glob.glob(src + "/prof/*/*kernel_stats.csv"), key=os.path.getmtime)
if ks:
    shutil.copy(ks[-1], os.path.join(dst, "%s_kernel_stats_serial.csv" % tag))
print("\n".join(sorted(glob.glob(dst + "/%s_*" % tag))))
