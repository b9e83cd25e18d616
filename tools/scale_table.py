"""Scaling table from bench.py lines (one JSON line per N, any order):

    python tools/scale_table.py line_n1.json line_n2.json line_n4.json line_n8.json

weak scaling:   efficiency(N) = value(N) / (N * value(1))
strong scaling: efficiency(N) = value(N) / (N * value(1))   (same video, N times the GPUs)
"""
import json
import sys


def main():
    lines = sorted((json.loads(open(p).read().strip().splitlines()[-1]) for p in sys.argv[1:]), key=lambda d: d["n_gpus"])
    base = next((d for d in lines if d["n_gpus"] == 1), None)
    print("%4s %12s %12s %10s %12s %14s" % ("N", "fps", "fps/GPU", "ms/step", "efficiency", "all_gather_ms"))
    for d in lines:
        eff = d["value"] / (d["n_gpus"] * base["value"]) if base else float("nan")
        print("%4d %12.1f %12.1f %10.3f %12.3f %14s" % (d["n_gpus"], d["value"], d["value"] / d["n_gpus"], d["ms_per_step"],
                                                      eff, d.get("all_gather_ms")))


if __name__ == "__main__":
    main()
