"""A/B of per-layer tile choices inside the network on ONE box (development aid): the shipped table of a batch size with overrides,
run through tools/layer_table.py alternately.

    python tools/tile_ab.py 257 "13:conv3=24,14:conv3=24,15:conv3=24,7:shortcut=25" "13:conv3=26,14:conv3=26,15:conv3=26,7:shortcut=26" [reps]
"""
import json
import os
import subprocess
import sys

sys.path.insert(0, ".")
from human_dynamics_amd import engine as E

n = int(sys.argv[1])
variants = sys.argv[2:4]
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
tabs = json.load(open(E.TILE_TABLES))
base = tabs["2:%d" % n]
rows = sorted({k.split("=")[0] for v in variants for k in v.split(",")})
names = {"conv1": "c1", "conv2": "c2", "conv3": "c3", "shortcut": "sc"}
for rep in range(reps):
    for vi, v in enumerate(variants):
        tab = dict(base)
        for kv in v.split(","):
            k, t = kv.split("=")
            tab[k] = int(t)
        path = "/tmp/tile_ab_%d.json" % vi
        json.dump({"2:%d" % n: tab}, open(path, "w"))
        env = dict(os.environ, HMMR_TILE_CACHE=path, HMMR_AUTOTUNE="0")
        out = subprocess.run([sys.executable, "tools/layer_table.py", str(n), "f16x3", "5"], env=env, capture_output=True, text=True).stdout
        got = {}
        units = ["1.1", "1.2", "1.3", "2.1", "2.2", "2.3", "2.4", "3.1", "3.2", "3.3", "3.4", "3.5", "3.6", "4.1", "4.2", "4.3"]
        for line in out.splitlines():
            p = line.split()
            if len(p) > 3 and p[0] in units:
                got["%d:%s" % (units.index(p[0]), p[1])] = float(p[-4])
            if p and p[0] == "TOTAL":
                got["TOTAL"] = float(p[1])
        print("variant %d rep %d:" % (vi, rep), " ".join("%s %.4f" % (r, got.get(r.split(":")[0] + ":" + names[r.split(":")[1]], -1)) for r in rows),
              "TOTAL %.4f" % got.get("TOTAL", -1), flush=True)
