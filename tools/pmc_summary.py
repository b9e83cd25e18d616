"""Summarise separate rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE / MFMA busy) of bench.py.

    python tools/pmc_summary.py gpurun_out profiles/r01_pmc_summary.json

Corrections (MI355X_MICROARCH.md section HBM, re-calibrated here on stem_repack, whose byte counts
are known): FETCH_SIZE counts 64 B per 128-B request -> x2; WRITE_SIZE x1; both in KiB.
"""
import collections
import csv
import glob
import os
import json
import sys


def agg(root, tag, counter):
    # gpurun merges into existing local directories: take the newest run (highest rocprofv3 pid prefix)
    f = max(glob.glob("%s/pmc_%s/*/*counter_collection.csv" % (root, tag)), key=lambda x: int(os.path.basename(x).split("_")[0]))
    d = collections.defaultdict(lambda: [0, 0.0])
    for x in csv.DictReader(open(f)):
        if x["Counter_Name"] == counter:
            d[x["Kernel_Name"]][0] += 1
            d[x["Kernel_Name"]][1] += float(x["Counter_Value"])
    return d


LPP = int(sys.argv[3]) if len(sys.argv) > 3 else 40      # MFMA launches per ResNet pass (bench.py roofline.kernel)


def main():
    root, out = sys.argv[1], sys.argv[2]
    fe, wr = agg(root, "FETCH_SIZE", "FETCH_SIZE"), agg(root, "WRITE_SIZE", "WRITE_SIZE")
    mf = agg(root, "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES")
    gui = agg(root, "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE")
    res = {"corrections": {"FETCH_SIZE": "x2 x1024 B", "WRITE_SIZE": "x1 x1024 B",
                           "mfma_util": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 XCDs * 1024 SIMDs)"},
           "kernels": {}}
    for k in sorted(fe, key=lambda k: -fe[k][1])[:16]:
        n = fe[k][0]
        rd = 2 * 1024 * fe[k][1] / n
        w = 1024 * wr[k][1] / max(wr[k][0], 1) if k in wr else 0.0
        util = (mf[k][1] / (gui[k][1] / 8.0 * 1024.0)) if k in mf and gui[k][1] > 0 else None
        res["kernels"][k[:110]] = {"launches": n, "hbm_read_MB_per_launch": round(rd / 1e6, 1),
                                   "hbm_write_MB_per_launch": round(w / 1e6, 1),
                                   "mfma_util": None if util is None else round(util, 4)}
    # the ResNet's MFMA launches: the bf16->bf16 instantiations of conv_gemm_kernel, the fused bottleneck tails
    # and the fused stem
    rn = [k for k in fe if "conv_gemm_kernelIDF16bDF16b" in k or "stem_fused_kernel" in k or "bottleneck_tail_kernel" in k]
    n = sum(fe[k][0] for k in rn)
    rd = sum(2 * 1024 * fe[k][1] for k in rn)
    w = sum(1024 * wr[k][1] for k in rn if k in wr)
    busy = sum(mf[k][1] for k in rn if k in mf)
    g = sum(gui[k][1] for k in rn if k in gui)
    res["resnet_conv_gemm"] = {"launches": n, "hbm_bytes_per_pass": round((rd + w) / n * LPP), "launches_per_pass": LPP, "hbm_bytes_per_launch": round((rd + w) / n),
                               "hbm_read_bytes_per_launch": round(rd / n), "hbm_write_bytes_per_launch": round(w / n),
                               "mfma_util": round(busy / (g / 8.0 * 1024.0), 4),
                               "note": "averaged over the ResNet passes of `HMMR_TILE_CACHE=<tuned> bench.py --serial --steps 2 --warmup 1 "
                                       "--no-cpu-baseline --no-pcie` (every pass encodes 257 frames)"}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res["resnet_conv_gemm"], indent=1))


if __name__ == "__main__":
    main()
