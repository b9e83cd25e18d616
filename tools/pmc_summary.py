"""Summarise separate rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE / MFMA busy) of bench.py.

    python tools/pmc_summary.py <dir with pmc_FETCH_SIZE/ pmc_WRITE_SIZE/ pmc_SQ_VALU_MFMA_BUSY_CYCLES/> <out.json> [dtype]

Also writes <dir>/pmc_<COUNTER>_per_kernel.csv (per-kernel launch counts and counter sums), the small files that are
committed under profiles/ -- the raw per-dispatch CSVs stay on the GPU box.

Corrections (MI355X_MICROARCH.md section HBM, re-calibrated in round 1 on stem_repack, whose byte counts are known):
FETCH_SIZE counts 64 B per 128-B request -> x2; WRITE_SIZE x1; both in KiB.
"""
import collections
import csv
import glob
import json
import os
import sys


def agg(root, tag, counter):
    # gpurun merges into existing local directories: take the newest run (highest rocprofv3 pid prefix)
    files = glob.glob("%s/pmc_%s/*/*counter_collection.csv" % (root, tag))
    if not files:
        # the raw per-dispatch files stay on the GPU box: re-summarise from the per-kernel sums a run there left behind
        per = "%s/pmc_%s_per_kernel.csv" % (root, counter)
        d = collections.defaultdict(lambda: [0, 0.0])
        if os.path.exists(per):
            for x in csv.reader(open(per)):
                if x[0] != "kernel":
                    d[x[0]] = [int(x[1]), float(x[2])]
        return d
    f = max(files, key=lambda x: int(os.path.basename(x).split("_")[0]))
    d = collections.defaultdict(lambda: [0, 0.0])
    for x in csv.DictReader(open(f)):
        if x["Counter_Name"] == counter:
            d[x["Kernel_Name"]][0] += 1
            d[x["Kernel_Name"]][1] += float(x["Counter_Value"])
    with open("%s/pmc_%s_per_kernel.csv" % (root, counter), "w") as o:
        o.write("kernel,launches,%s_sum\n" % counter)
        for k in sorted(d, key=lambda k: -d[k][1]):
            o.write('"%s",%d,%.1f\n' % (k, d[k][0], d[k][1]))
    return d


def resnet_kernel(k, dtype):
    """The ResNet's MFMA launches of one operand mode (the IEF GEMMs of the same instantiation ride along: < 1 % of the bytes)."""
    if ("stem_fused" in k or "bottleneck_tail_kernel" in k or "tail_split_kernel" in k or "unit_pair_kernel" in k or "conv3x3_stream_kernel" in k
            or "b1_unit_kernel" in k or "conv1x1_stream_kernel" in k):
        return True
    if "conv3x3_patch_kernel" in k or "conv3x3_pipe_kernel" in k:
        return ("bsplit_t" in k) == (dtype == "f16x3")
    if "conv_gemm_kernel" not in k:
        return False
    if dtype == "bf16":
        return "conv_gemm_kernelIDF16bDF16b" in k or "conv_gemm_kernel<__bf16, __bf16" in k
    if dtype == "f16x3":
        return "8bsplit_tS0_" in k or "conv_gemm_kernel<bsplit_t, bsplit_t" in k
    return "conv_gemm_kernelIffL" in k or "conv_gemm_kernel<float, float" in k


def main():
    root, out = sys.argv[1], sys.argv[2]
    dtype = sys.argv[3] if len(sys.argv) > 3 else "f16x3"
    fe, wr = agg(root, "FETCH_SIZE", "FETCH_SIZE"), agg(root, "WRITE_SIZE", "WRITE_SIZE")
    mf = agg(root, "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES")
    gui = agg(root, "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE")
    res = {"dtype": dtype,
           "corrections": {"FETCH_SIZE": "x2 x1024 B", "WRITE_SIZE": "x1 x1024 B",
                           "mfma_util": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 XCDs * 1024 SIMDs)"},
           "kernels": {}}
    for k in sorted(fe, key=lambda k: -fe[k][1])[:20]:
        n = fe[k][0]
        rd = 2 * 1024 * fe[k][1] / n
        w = 1024 * wr[k][1] / max(wr[k][0], 1) if k in wr else 0.0
        util = (mf[k][1] / (gui[k][1] / 8.0 * 1024.0)) if k in mf and gui[k][1] > 0 else None
        res["kernels"][k[:140]] = {"launches": n, "hbm_read_MB_per_launch": round(rd / 1e6, 1),
                                   "hbm_write_MB_per_launch": round(w / 1e6, 1),
                                   "mfma_util": None if util is None else round(util, 4)}
    passes = sum(fe[k][0] for k in fe if "bn_relu_avgpool_kernel" in k)      # one pool5 launch per ResNet pass
    rn = [k for k in fe if resnet_kernel(k, dtype)]
    n = sum(fe[k][0] for k in rn)
    rd = sum(2 * 1024 * fe[k][1] for k in rn)
    # every counter is its OWN run of the command: if the runs held different numbers of passes (a time-based loop in the command), sums must not be
    # mixed -- scale each kernel's WRITE_SIZE sum to the launch count the FETCH_SIZE run saw (per-launch values are what is stable; round 6)
    wr = {k: [fe[k][0], v[1] * fe[k][0] / v[0]] if (k in fe and v[0]) else v for k, v in wr.items()}
    w = sum(1024 * wr[k][1] for k in rn if k in wr)
    busy = sum(mf[k][1] for k in rn if k in mf)
    g = sum(gui[k][1] for k in rn if k in gui)
    fam = {}
    for name, pred in (("conv_gemm_kernel", lambda k: "conv_gemm_kernel" in k and resnet_kernel(k, dtype)),
                       ("conv3x3_patch_kernel", lambda k: "conv3x3_patch_kernel" in k or "conv3x3_pipe_kernel" in k),
                       ("conv3x3_stream_kernel", lambda k: "conv3x3_stream_kernel" in k),
                       ("conv1x1_stream_kernel", lambda k: "conv1x1_stream_kernel" in k),
                       ("fused_unit_tails", lambda k: "tail_split_kernel" in k or "bottleneck_tail_kernel" in k),
                       ("unit_pair_kernel", lambda k: "unit_pair_kernel" in k),
                       ("b1_unit_kernel", lambda k: "b1_unit_kernel" in k),
                       ("fused_stem", lambda k: "stem_fused" in k)):
        ks = [k for k in rn if pred(k)]
        gg = sum(gui[k][1] for k in ks if k in gui)
        if ks and gg:
            fam[name] = {"launches_per_pass": round(sum(fe[k][0] for k in ks) / max(passes, 1), 2),
                         "mfma_util": round(sum(mf[k][1] for k in ks if k in mf) / (gg / 8.0 * 1024.0), 4),
                         "hbm_bytes_per_pass": round(sum(2 * 1024 * fe[k][1] + (1024 * wr[k][1] if k in wr else 0) for k in ks) / max(passes, 1))}
    if n and passes:
        res["families"] = fam
        res["resnet_conv_gemm"] = {
            "launches": n, "resnet_passes": passes, "launches_per_pass": round(n / passes, 2),
            "hbm_bytes_per_pass": round((rd + w) / passes), "hbm_bytes_per_launch": round((rd + w) / n),
            "hbm_read_bytes_per_launch": round(rd / n), "hbm_write_bytes_per_launch": round(w / n),
            "mfma_util": round(busy / (g / 8.0 * 1024.0), 4) if g else None,
            "note": "averaged over the ResNet passes of `HMMR_TILE_CACHE=<tuned> bench.py --dtype %s --serial --only-main --steps 2 "
                    "--warmup 1 --no-cpu-baseline --no-pcie` (every pass encodes 257 frames; the few IEF launches of the same "
                    "kernel instantiation are included, < 1 %% of the bytes)" % dtype}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res.get("resnet_conv_gemm"), indent=1))


if __name__ == "__main__":
    main()
