"""Development switches of the Python package -- the ONE place it reads the environment.

`libhmmr_hip.so` never reads the environment (include/hmmr_hip.h); its development switches are the fields of
`hmmr_debug_t`, set through `hmmr_set_debug`.  The package above it has A/B switches of its own (launch schedules,
fusions, tuner candidates) that tools/ and a few tests flip per process; every one of them is declared in `FLAGS`
below with its default and meaning, is read through `get()` and nowhere else, and leaves every result bit-identical
unless its line says otherwise.  Product code paths take the same choices as constructor arguments (HmmrEngine,
ShardedPredictor); a flag only overrides the default of the argument it names.
"""
from __future__ import annotations

import os

FLAGS = {
    # name: (default, meaning)
    "LIB_PATH": ("", "load this shared object instead of human_dynamics_amd/libhmmr_hip.so (tools/probe_build.sh)"),
    "STEM": ("", "u[nfused] | f[used]: hmmr_debug_t.stem_route"),
    "STEM_C1": ("1", "0: hmmr_debug_t.stem_no_conv1"),
    "GEMM_PROBE": ("0", "hmmr_debug_t.gemm_probe (only the -DHMMR_GEMM_PROBE development build reads the field; results are garbage there)"),
    "RESNET_CHUNK": ("", "encode the batch in sequential chunks of this many frames (heuristic tiles)"),
    "FUSE_PREACT": ("block1,block2,block3,block4", "blocks whose units pre-activate inside the operand staging of conv1 / shortcut"),
    "FUSE_TAIL": ("1", "0 | 1 | block1 | noconv2 | conv2b1 | nosc | nostride2: which fused bottleneck tails the packer marks"),
    "FUSE_SC": ("1", "0 | 1 | all: shortcut + conv1 as one column-split GEMM (1: blocks 3-4)"),
    "PREACT_FIRST": ("0", "1: a block's first unit fuses its pre-activation too"),
    "FOLD_SC": ("", "0 | 1: conv shortcut folded into conv3's K (default: f16x3 only)"),
    "PATCH_3X3": ("2", "the stride-1 3x3 layers of blocks 2-4 (f16x3): 2 = the one-wave-per-SIMD stream kernel (k_order 2), 1 = the 8-wave patch kernels (k_order 1), 0 = tap-major K order and the im2col gather; the three differ by fp32 accumulation rounding"),
    "B1_STREAM": ("0", "1: the conv2 of block1/unit_1 and unit_2 as a launch of the 3x3 stream kernel (64-channel tiles) instead of inside their fused tails (f16x3; measured equal)"),
    "B1_UNIT": ("1", "0: block1/unit_1 and unit_2 as the round-3 LDS-panel tails with conv2 inside (tap-major conv2) instead of the whole-unit kernel of csrc/b1_unit.hip (f16x3; conv2 chunk-major: the two differ by fp32 accumulation rounding)"),
    "STREAM_1X1": ("1", "0: the conv1 of block 4 and of block2/unit_1 and block3/unit_1's shortcut + conv1 as launches of the 8-wave tiles (csrc/gemm_conv.hip) instead of the two-ring stream kernel of csrc/conv1x1_stream.hip (f16x3; the two differ by fp32 accumulation rounding)"),
    "UNIT_PAIR": ("1", "0 | 1 | block2 | block3: the stride-1 units of blocks 2-3 as register-resident unit pairs (csrc/unit_pair.hip; f16x3)"),
    "AUTOTUNE": ("1", "0: no per-layer tile tuning pass (shipped table / library heuristic only)"),
    "TILE_TABLE": ("1", "0: ignore the shipped tile tables (tile_tables.json), tune or fall back to the heuristic"),
    "TILE_CACHE": ("", "json file the tuned tile tables are read from / written to (profiling runs)"),
    "TUNE_TILES": ("5,6,3,1,2,7,8,11,4", "hmmr_conv_desc_t.tile candidates of the tuner"),
    "RESNET_STREAMS": ("2", "contiguous parts a large batch is encoded as, on concurrent streams"),
    "RESNET_PRIORITY": ("-1", "HIP priority of the ResNet side streams"),
    "TAIL_PRIORITY": ("0", "HIP priority of the tail stream (dist.ShardedPredictor)"),
    "STEP_BUFFERS": ("2", "calls in flight of a pipelined ShardedPredictor (record buffers, and encode streams with STEP_STREAMS): 2, or more to try"),
    "STEP_STREAMS": ("1", "0: keep consecutive steps' ResNet passes on one stream (dist.ShardedPredictor)"),
    "STREAM_POISON": ("", "1: evaluation/streaming.py fills a chunk's record buffer with NaN before its tail writes it (a row nobody wrote, or a download that ran early, shows on the host)"),
    "STREAM_TRACE": ("", "1: evaluation/streaming.py prints the host and device timeline of a call"),
}


def get(name):
    """Value of development switch HMMR_<name> (a string), or its declared default."""
    default = FLAGS[name][0]
    return os.environ.get("HMMR_" + name, default)


def is_set(name):
    return ("HMMR_" + name) in os.environ and name in FLAGS
