"""Measure the per-layer conv tiles of the ResNet on this MI355X for the batch sizes the BASELINE configurations produce and
write them in the format `human_dynamics_amd/tile_tables.json` ships (dev aid, run on the GPU box):

    python tools/make_tile_tables.py gpurun_out/tile_tables.json [dtype ...]

Keys "<hmmr_dtype_t>:<frames>" -> {"<unit>:<layer>": tile}.  Sizes: 20-frame windows are below the tuner's floor; 40 (the operand-mode probe, precision.py); 64 / 65
(config 2, FeatureExtractor batches), 128 / 129 (the two concurrent parts of a 256-frame shard + its zero image), 256 / 257
(a 256-frame shard as one pass: the bench's step streams), 512 / 513 and 1024 (config 3 / long device-resident videos); round 6: 96 (the
parts of a 192-frame call) and 160 / 161 (the reference's default Tester.predict, B = 8 x T = 20, one stream).  HMMR_TABLE_SIZES=a,b,... measures other sizes.
The engine uses the nearest size within 30 %, so these cover 32 ... 1330 frames.  Per candidate tile the minimum over five timed passes (the on-line tuner takes two)."""
import json
import os
import sys

import torch

sys.path.insert(0, ".")
out_path = sys.argv[1]
dtypes = sys.argv[2:] or ["f16x3", "bf16", "f32"]
os.environ["HMMR_TILE_TABLE"] = "0"
os.environ["HMMR_AUTOTUNE"] = "force"
from human_dynamics_amd import assets                     # noqa: E402
from human_dynamics_amd.engine import HmmrEngine, DTYPES  # noqa: E402

SIZES = tuple(int(v) for v in os.environ["HMMR_TABLE_SIZES"].split(",")) if os.environ.get("HMMR_TABLE_SIZES") else (40, 64, 65, 96, 128, 129, 160, 161, 256, 257, 512, 513, 1024)
w = assets.make_synthetic_weights(0)
result = {"_comment": "per-layer hmmr_conv_desc_t.tile, measured by tools/make_tile_tables.py on one MI355X; "
                      "keys '<hmmr_dtype_t>:<frames>' -> {'<unit>:<layer>': tile}; tiles never change a result bit"}
frames = torch.rand((max(SIZES), 224, 224, 3), device="cuda") * 2 - 1
for dt in dtypes:
    eng = HmmrEngine(w, None, dtype=dt, device="cuda:0")
    eng.resnet_streams = 1
    for nt in SIZES:
        n_zero = nt % 2                       # the odd sizes are "shard + the zero padding image"
        n = nt - n_zero
        tab = eng._tune_resnet(frames[:n], n, n_zero, reps=6)       # min over five timed passes per candidate tile
        eng._set_tiles(tab)
        result["%d:%d" % (DTYPES[dt], nt)] = {"%d:%s" % k: int(v) for k, v in sorted(tab.items())}
        print(dt, nt, "tuned in %.0f ms" % eng.tune_log[-1][1], flush=True)
    del eng
    torch.cuda.empty_cache()
json.dump(result, open(out_path, "w"), indent=0, sort_keys=True)
print("wrote", out_path)
