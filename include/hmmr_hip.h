/* libhmmr_hip.so -- C ABI of the MI355X (gfx950) HMMR inference hot path.
 *
 * Drop-in boundary for akanazawa/human_dynamics `Tester.predict`
 * (src/evaluation/tester.py:229-258).  The reference has no FFI of its own:
 * the path sits behind one `sess.run` of a TF-1.8 graph.  Each entry point
 * below replaces one op-group of that graph (citations are relative to the
 * reference tree) and is what a ctypes binding on the reference side would
 * call.
 *
 * Conventions
 *   - every pointer that carries tensor data is a DEVICE pointer (HBM);
 *     structs themselves live in host memory;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *     all work is enqueued asynchronously on it, nothing is allocated or
 *     synchronised inside a call: scratch comes from the caller through
 *     (`ws`, `ws_bytes`) sized by the matching *_workspace_bytes();
 *   - return 0 on success, <0 on error; hmmr_last_error() describes the last
 *     failure of the calling thread;
 *   - activations are NHWC / row-major, fp32 at every stage boundary.  Inside
 *     the ResNet / temporal / IEF stages the GEMM operand type is selected by
 *     `dtype` (HMMR_F32 = exact-fp32 MFMA, HMMR_BF16 = bf16 MFMA with fp32
 *     accumulate).  The SMPL stage is always fp32.
 */
#ifndef HMMR_HIP_H
#define HMMR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HMMR_ABI_VERSION 19      /* 19: hmmr_clock_probe (measurement aid); 18: HMMR_FLAG_NAN, hmmr_debug_t.pair_form (the wave-specialised unit pair), the C-side packers hmmr_pack_*; 17: k_order = 2 on a 1x1 filter (the two-ring stream kernel of csrc/conv1x1_stream.hip, tiles 22 .. 26), hmmr_conv1x1_stream_bytes; 16: hmmr_tail_desc_t.unit_stream, hmmr_resnet_unit_t.unit_stream (the whole-unit kernel of block 1, csrc/b1_unit.hip), hmmr_b1_unit_stream_bytes, hmmr_debug_t.pair_min_pixels / pair_two_tile_min / launch counters, hmmr_resnet_unit_t.conv1_frag (block1/unit_1's conv1 inside the split stem); 15: k_order = 2 (the one-wave-per-SIMD 3x3 stream kernel, tiles 12 .. 18), hmmr_conv3x3_stream_bytes; 14: hmmr_tail_desc_t.pair_stream / c_xp, hmmr_resnet_unit_t.pair_stream (the register-resident unit pair of blocks 2-3), hmmr_run_flags; 13: hmmr_conv_desc_t.batch (grouped launches); 12: k_order = 1, 3x3 SAME convolutions out of an LDS-resident input patch (tiles 9, 10, 11) */

/* HMMR_F16X3: "split" tensors -- every group of 8 consecutive channels is 32 bytes, [hi x8][lo x8] with
 * hi = fp16(x), lo = fp16(x - hi), x clamped to +-65504 (4 bytes per element, 22 mantissa bits while lo is a normal
 * fp16, an absolute 6e-8 below that); GEMMs on them issue three fp16 MFMAs per operand pair (lo*hi + hi*lo + hi*hi,
 * fp32 accumulate; v_mfma_f32_32x32x16_f16 keeps fp16 subnormals).  Filter banks of this dtype are expected with every
 * output-channel row scaled by a power of two to a maximum in [2^13, 2^14) and the inverse factor in the layer's `scale`
 * (packing.row_pow2) -- unscaled filters work too, with fewer bits for small values.  The parity-grade throughput mode:
 * fp32-class operands at the 16-bit MFMA rate / 3.  (Rounds 1-2 used bf16 halves: 16-17 bits.) */
enum { HMMR_F32 = 0, HMMR_BF16 = 1, HMMR_F16X3 = 2 };

int hmmr_abi_version(void);
const char* hmmr_last_error(void);

/* Sticky run flags of the CURRENT device, OR-ed over everything launched on it since the last clear.  HMMR_FLAG_SATURATED: a value
 * beyond the fp16 range (+-65504, or +-inf) reached a split (HMMR_F16X3) store and was clamped -- the results of that call are not
 * the network's; run it with fp32 operands instead (the Python mirror does: Tester.precision["saturated"]).  Synchronises with the
 * device (hipDeviceSynchronize, then a 4-byte copy per translation unit): call it where the results are read, not per launch.
 * clear != 0 resets the flags.
 * HMMR_FLAG_NAN (round 6, together with HMMR_FLAG_SATURATED): the value was a NaN -- a NaN pixel, weight or constant.  The clamp
 * (v_med3_f32) turns a NaN into a finite number, so without this flag it would leave as a plausible result.  Every split store keeps
 * its running maximum with gfx950's NaN-propagating v_maximum3_f32 and tests !(max <= 65504); the split stem applies the same test to
 * the fp32 pixels it reads. */
#define HMMR_FLAG_SATURATED 1u
#define HMMR_FLAG_NAN 2u
int hmmr_run_flags(unsigned* flags, int clear);

/* Development switches for A/B measurements and tests.  Process-wide; all zero = the product defaults.  No
 * switch changes a result beyond what its comment says; the library never reads the environment. */
typedef struct hmmr_debug_s {
    int stem_route;        /* 0: default (fused stem kernel for bf16 / f16x3, re-pack + GEMM + pool for f32);
                              1: always the three-kernel route; 2: always the fused kernel */
    int stem_no_conv1;     /* 1: the fused bf16 stem leaves block1/unit_1's conv1 to its own launch */
    int gemm_probe;        /* read only by the -DHMMR_GEMM_PROBE development build (tools/probe_build.sh): the GEMM K loop
                              drops its MFMAs (1), its operand loads after the first stage (2) or its barriers (4), to see
                              which of the three bounds a shape; results are garbage then.  The product build ignores it. */
    int smpl_blend_mfma;   /* the SMPL blend-shape product [m,218] x [218,3 x 6890]: 0 = the default, split-fp16 operands on the
                              matrix cores (smpl_verts_split_kernel; needs hmmr_smpl_consts_t.dirs_split, else form 2);
                              1 = exact fp32 on the matrix cores (smpl_verts_mfma_kernel, v_mfma_f32_32x32x2_f32);
                              2 = the packed-FMA vector form (smpl_verts_kernel).  1 and 2 agree to one fp32 ulp, 0 with
                              them to ~1e-7 m */
    int ief_no_group;      /* 1: hmmr_ief_fwd runs the delta regressors one after the other instead of as grouped launches (same bits) */
    int reserved[3];
    int pair_min_pixels;   /* hmmr_resnet50_fwd runs a register-resident unit pair (csrc/unit_pair.hip) as the two launches it replaces when
                              the unit has fewer pixels than this (same bits, faster for short batches): 0 = the default (12000 in block 2, 14000 in block 3),
                              1 = always the pair kernel, INT_MAX = never */
    int pair_two_tile_min; /* the pair kernel of the block-2 shapes runs as PERSISTENT workgroups (several 128-pixel tiles each) when the launch has at least
                              this many tiles (same bits; one workgroup per CU then walks tiles b, b + grid, ...): 0 = the default (512 = two rounds of workgroups), 1 = always, INT_MAX = never */
    int pair_form;         /* (ABI 18) the unit pairs with a shortcut tensor (blocks 2-3): 0 = the default, the one-wave-per-SIMD form of round 4;
                              2 = the wave-specialised form of round 6 (two waves per SIMD: conv3 + the trunk epilogue in one, conv1' + all
                              memory traffic in the other).  Same bits; measured equal */
} hmmr_debug_t;
void hmmr_set_debug(const hmmr_debug_t* d);     /* NULL = defaults */
void hmmr_get_debug(hmmr_debug_t* d);

/* Launch counters of the fused-unit kernels, process-wide, since the last clear: how often hmmr_bottleneck_tail / hmmr_conv_gemm ended in
 * each of them.  For tests that must know WHICH kernel a schedule ran (a bit-identity test of two schedules that silently take the same
 * kernels proves nothing). */
typedef struct {
    unsigned long long unit_pair;       /* csrc/unit_pair.hip */
    unsigned long long b1_unit;         /* csrc/b1_unit.hip */
    unsigned long long tail_split;      /* csrc/bottleneck_split.hip (LDS-panel tails) */
    unsigned long long conv3x3_stream;  /* csrc/conv3x3_stream.hip */
    unsigned long long conv1x1_stream;  /* csrc/conv1x1_stream.hip (ABI 17) */
} hmmr_launch_counts_t;
void hmmr_launch_counts(hmmr_launch_counts_t* out, int clear);

/* ------------------------------------------------------------------------- *
 * Generic implicit-GEMM convolution / fully-connected building block.
 * out[m, n] = epilogue( sum_k A[m, k] * W[n, k] ), m = (img, oy, ox) over an
 * NHWC input gathered on the fly, k = (ky, kx, ci).  Replaces every
 * slim.conv2d / tf.contrib.layers.conv2d / slim.fully_connected on the path
 * (src/models.py:65-74 via slim resnet_v2; :102-113; :173-184; :209-221).
 * Exposed so each layer shape can be parity-tested in isolation.
 * ------------------------------------------------------------------------- */
typedef struct {
    /* operands */
    const void* in;        /* activations, dtype in_dtype                        */
    const void* w;         /* [cout_pad][K], K = kh*kw*cin contiguous, in_dtype;
                              cout_pad = cout rounded up to 128                   */
    const float* scale;    /* [cout_pad] or NULL (=1): multiplies the accumulator */
    const float* shift;    /* [cout_pad] or NULL (=0): bias / folded BN shift     */
    const void* res;       /* residual added before the ReLU, out_dtype, or NULL  */
    void* out;             /* [M, ldo] out_dtype, may be NULL if out2 is set      */
    void* out2;            /* optional second output relu(out*scale2+shift2) of the value as STORED in `out`'s
                              dtype (= what a consumer applying pro_scale/pro_shift to `out` computes), or NULL */
    const float* scale2;
    const float* shift2;
    int in_dtype;          /* HMMR_F32 / HMMR_BF16 / HMMR_F16X3 */
    int out_dtype;
    /* input geometry (element strides) */
    int n_img, hin, win, cin;          /* cin: channels per tap, power of two, >= 32B/elt */
    int64_t in_img_stride;
    int in_row_stride, in_px_stride;
    /* filter geometry */
    int kh, kw, sy, sx, py, px;
    int ho, wo;
    int cout;
    int ldo;               /* row stride of out / out2 in elements (multiple of 8) */
    /* residual addressing: flat rows of stride ldr, or strided pixels */
    int ldr;
    int res_strided;       /* 1: res[img*res_img_stride + oy*res_row_stride + ox*res_px_stride + n] */
    int64_t res_img_stride;
    int res_row_stride, res_px_stride;
    int relu;              /* ReLU on `out` after the residual add */
    int tile;              /* 0 = auto; 4-wave tiles: 1 = 128x128, 2 = 128x64, 3 = 64x64;
                              8-wave tiles: 5 = 128x128, 6 = 128x64 (two workgroups per CU, 2 LDS stages);
                              7 = 256x128, 8 = 128x256 (one workgroup per CU: 3-stage LDS ring, ping-pong wave groups);
                              k_order 1 only: 9 = 256x128, 10 = 128x256 (the same structure, A operand out of an input patch),
                              11 = 256x128 without a load segment (two fragment sets per wave, one barrier per K step);
                              k_order 2 only (csrc/conv3x3_stream.hip; 4 waves, one per SIMD, 128 output channels per tile): 12 = 448 pixels
                              (wave tile 7 x 2 accumulators of 32 x 32), 13 = 256 (4 x 2), 14 = 512 (8 x 2), 15 = 384 (6 x 2), 16 = 320 (5 x 2);
                              17 = 512, 18 = 384 pixels with the waves splitting the pixels (4 x 4 / 3 x 4); 19 = 640, 20 = 512 pixels x 64 channels
                              (cout = 64, images up to 56 pixels wide); 21 = 224 pixels, every wave all of them and 32 channels (7 x 1; split only);
                              k_order 2 with a 1x1 filter only (csrc/conv1x1_stream.hip; 4 waves, one per SIMD, 128 output channels per tile, both
                              operands through LDS rings): 22 = 224 pixels (7 x 1 accumulators per wave, every wave 32 of the channels),
                              23 = 256 pixels (8 x 1), 24 = 448 pixels (7 x 2, waves 2 x 2), 25 = 256 pixels (4 x 2, waves 2 x 2),
                              26 = 256 pixels (4 x 2) with TWO workgroups per CU (rings 3 deep, 256 registers per wave) */
    /* split-K (for GEMMs with few output tiles and a long K): split_k > 1 slices K into that many
     * contiguous ranges, each range leaves an fp32 partial plane in `ws`, and a second launch adds the
     * planes IN SLICE ORDER and applies the epilogue.  The caller fixes split_k per layer (never from
     * the batch size), so results stay independent of batch composition.  0/1 = off. */
    int split_k;
    void* ws;              /* >= hmmr_conv_splitk_workspace_bytes(M, cout, split_k) */
    size_t ws_bytes;
    /* fused pre-activation of the INPUT: A[m, k] = relu(in * pro_scale[ci] + pro_shift[ci]) applied
     * while the operand is staged (slim bottleneck_v2 `preact` BN + ReLU, consumer side).  [cin]
     * floats each, or both NULL.  Only for 1x1, un-padded convolutions. */
    const float* pro_scale;
    const float* pro_shift;
    /* column split: output channels [n_split, cout) go to out_b (row stride ldo_b, channel n - n_split)
     * with ReLU flag relu_b instead of `out` / `relu` -- two convolutions over the same input as ONE
     * GEMM (the conv shortcut and conv1 of a block's first bottleneck unit).  n_split % 8 == 0; not
     * with out2, res or split_k.  out_b == NULL: off. */
    void* out_b;
    int ldo_b, n_split, relu_b;
    /* second operand source: K = cin + cin2, the first cin elements of a row of A come from `in`, the next cin2 from
     * `in2` -- two 1x1 convolutions over two tensors of the same pixel grid summed in ONE accumulator (a unit's conv3
     * and its conv shortcut: `w` = [W3 | Wsc] along K, `shift` = the sum of the two biases; the shortcut tensor is
     * never stored).  Both tensors dense [M][cin] / [M][cin2] (1x1, stride 1, no padding), cin and cin2 multiples of
     * the 128-byte K step, no pro_scale, no split_k.  in2 == NULL: off. */
    const void* in2;
    int cin2;
    /* order of K inside a filter row.  0: (tap, channel) -- k = (ky*kw + kx)*cin + ci.
     * 1: chunk-major -- k = ((ci / E)*9 + ky*3 + kx)*E + ci % E, E = the elements of one 128-byte K step (32 for a
     * split tensor).  For 3x3 / stride 1 / pad 1 convolutions over a dense NHWC tensor only, and it selects another
     * kernel: a workgroup keeps the 128-byte channel chunk of every input pixel its tile touches (a PATCH: the tile's
     * pixels and a halo of win + 1 pixels on either side) in LDS and reads the nine taps of that chunk as nine shifted
     * fragment sets (out-of-image taps read a zero row), so an input
     * line leaves L2 once per chunk instead of once per tap (tiles 9 / 10 / 11; scale/shift/relu epilogue only: no res,
     * out2, out_b, pro_scale, in2, split_k).  The sum over k is the same set of products in another order: results
     * differ from k_order 0 by fp32 rounding of the accumulation only.
     * 2: `w` is not a matrix but the FILTER STREAM of packing.pack_conv3x3_stream (hmmr_conv3x3_stream_bytes(cin, cout) bytes):
     * per 128 output channels, K steps kt = (ci / 16) * 9 + ky*3 + kx of 8 KB = 4 row blocks x (hi plane | lo plane) of MFMA
     * A-operand fragments (lane = 32 * (k half) + row; 8 halves = W[row][16 (ci / 16) + 8 half .. + 7] of that tap), rows scaled
     * like every split filter bank.  Same convolutions and epilogue as k_order 1; split (f16x3) tensors, cin % 32 == 0 -- or bf16 tensors (K steps of 32
     * channels `(ci / 32) * 9 + tap`, the two planes = the two 16-wide MFMA chunks, no row scaling), cin % 64 == 0 --,
     * cout % 128 == 0 and win <= 28 (tiles 12 .. 18, 21) or cout = 64 and win <= 56 (tiles 19 / 20) (csrc/conv3x3_stream.hip).  Every tile produces the same bits.
     * 2 with kh = kw = 1 (round 5; csrc/conv1x1_stream.hip, tiles 22 .. 26): `w` is the stream of packing.pack_conv1x1_stream
     * (hmmr_conv1x1_stream_bytes(cin + cin2, cout) bytes: K steps of 16 input channels, the same 8 KB per 128 output channels as one tap above).  A
     * stride-1 1x1 convolution over a dense [M][cin] split tensor, cin % 16 == 0 (at least 64: the rings are 4-6 K steps deep), cout % 128 == 0,
     * scale and shift given; no pro_scale, split_k, batch.  Two epilogues: scale / shift / relu with the out_b column split (n_split % 128 == 0), or
     * -- with any of res / out2 / in2, tiles 24 .. 26 -- the conv3 form: in2 continues K (cin2 % 16 == 0; not together with res), res is a dense
     * shortcut tensor with ldr == ldo, out2 = relu(scale2 * stored(out) + shift2) exactly as k_order 0 defines it.
     * Both operands go through LDS rings (one wave per SIMD), for layers whose K loop is bound by the round trip of
     * a two-stage ring (block 4, block3/unit_1's shortcut + conv1).  Differs from k_order 0 by fp32 rounding of the accumulation only. */
    int k_order;
    /* grouped launch: `batch` > 1 runs that many problems of this one shape as ONE launch (grid z); problem z reads and
     * writes at these BYTE offsets (multiples of 16, negative allowed) from problem 0: `in`, `w`, `out` / `out2`, `res`,
     * `scale` / `scale2` and `shift` / `shift2`.  A stride of 0 shares the operand.  With split_k, `ws` holds batch x the
     * planes.  Not with k_order, out_b, in2, pro_scale.  Problem z's results are those of its own launch, bit for bit
     * (the IEF's two delta regressors, src/models.py:343-361, run this way). */
    int batch;
    int64_t batch_in_bytes, batch_w_bytes, batch_out_bytes, batch_res_bytes, batch_scale_bytes, batch_shift_bytes;
} hmmr_conv_desc_t;

int hmmr_conv_gemm(const hmmr_conv_desc_t* d, void* stream);
size_t hmmr_conv_splitk_workspace_bytes(int m, int cout, int split_k);

/* ------------------------------------------------------------------------- *
 * ResNet-v2-50 image encoder: encoder_resnet (src/models.py:50-77) ->
 * tf.contrib.slim.nets.resnet_v2.resnet_v2_50(num_classes=None,
 * is_training=False) + squeeze.  images [n,224,224,3] fp32 in [-1,1] -> phi
 * [n,2048] fp32.
 * ------------------------------------------------------------------------- */
typedef struct {
    const void* w;         /* packed [cout_pad][K] in the struct's dtype */
    const float* scale;    /* folded BN scale or NULL */
    const float* shift;    /* folded BN shift or conv bias */
    int tile;              /* hmmr_conv_desc_t.tile for this layer's launch; 0 = library heuristic.
                              Results do not depend on it (same K order per output element);
                              the host may tune it per layer and batch size. */
    int k_order;           /* hmmr_conv_desc_t.k_order the filter rows of `w` were packed in (read for the 3x3 conv2 layers, and -- 2: the filter stream of
                              csrc/conv1x1_stream.hip -- for conv1 and the shortcut + conv1 bank sc_c1; such a unit has fuse_preact = 0) */
} hmmr_layer_t;

typedef struct {
    hmmr_layer_t conv1, conv2, conv3, shortcut;   /* shortcut.w == NULL: identity / subsample */
    hmmr_layer_t c3sc;         /* optional (stride-1 units with a conv shortcut): [W3 | Wsc] as one [depth][base + c_in]
                                  filter bank, shift = conv3 bias + shortcut bias: conv3 and the shortcut as ONE GEMM over
                                  {h2, preact} (hmmr_conv_desc_t.in2); needs fuse_preact == 0.  w == NULL: separate launches */
    hmmr_layer_t sc_c1;        /* optional: rows [shortcut (depth); conv1 (base)] of one [depth+base][c_in] filter
                                  bank with scale = [1..1; BN scale], shift = [bias; BN shift]: both convs as
                                  one column-split GEMM.  w == NULL: two launches. */
    const void* w3_frag;       /* f16x3 fused tail (fuse_tail == 1): this unit's conv3 filters ([W3 | Wsc] when c3sc is set) */
    const void* w1n_frag;      /* ... and the NEXT unit's conv1 filters, both FRAGMENT-MAJOR (hmmr_tail_desc_t); else NULL */
    const void* pair_stream;   /* f16x3 fused tail of blocks 2-3 (fuse_tail == 1): this unit's conv3 filters ([W3 | Wsc] when c3sc is
                                  set) and the NEXT unit's conv1 filters as ONE fragment stream (hmmr_tail_desc_t.pair_stream); else NULL */
    const void* conv1_frag;    /* f16x3, unit 0 only (optional): this unit's conv1 filters [base][c_in] FRAGMENT-MAJOR (as w3_frag): the fused stem
                                  computes block1/unit_1's conv1 on its pooled tile in the same launch (csrc/stem.hip); else NULL */
    const void* unit_stream;   /* f16x3 whole-unit kernel of block 1 (fuse_tail == 2, conv2.k_order == 2): conv2's, conv3's ([W3 | Wsc]) and
                                  the NEXT unit's conv1 filters as ONE fragment stream (hmmr_tail_desc_t.unit_stream); else NULL */
    const float* pre_scale;    /* this unit's folded `preact` BN, [c_in] */
    const float* pre_shift;
    int c_in, base, depth, stride;
    int fuse_preact;           /* 1: conv1/shortcut apply the preact while staging their operand;
                                  0: the previous unit's conv3 writes the preact tensor */
    int fuse_tail;             /* 4: stride-2 unit: conv2 + conv3 + add as one launch, no next conv1;
                                  3: as 2, and the unit's conv shortcut is computed in that launch too (c_in 64);
                                  2: as 1, with this unit's conv2 inside the same launch as well;
                                  1: this unit's conv3 + add and the NEXT unit's preact + conv1 run as one
                                  hmmr_bottleneck_tail launch (bf16, stride 1, block1 or block2 shapes, next unit
                                  of the same block with identity shortcut and fuse_preact) */
} hmmr_resnet_unit_t;

#define HMMR_RESNET_UNITS 16

typedef struct {
    int dtype;                         /* operand type of convs + activations */
    hmmr_layer_t stem;                 /* 7x7/2 packed as [128][8 taps x (8 px x 4 ch)] */
    hmmr_resnet_unit_t unit[HMMR_RESNET_UNITS];
    const float* post_scale;           /* postnorm BN folded */
    const float* post_shift;
} hmmr_resnet_weights_t;

/* ------------------------------------------------------------------------- *
 * Fused tail of a bottleneck unit (slim resnet_v2.bottleneck as invoked at src/models.py:65-75):
 *   out    = conv3(h2) * scale3 + shift3 + shortcut         (1x1, c_mid -> depth, no ReLU)
 *   out_h1 = relu(conv1'(relu(out * pre_scale + pre_shift)) * scale1 + shift1)
 * i.e. this unit's `conv3` + add and the NEXT unit's `preact` + `conv1` in one launch, so the next
 * conv1 does not re-read the trunk from HBM.  Bit-identical to the two hmmr_conv_gemm launches it
 * replaces (conv3 with a residual; conv1 with pro_scale/pro_shift).  bf16; (c_mid, depth, n2) = (64, 256, 64)
 * [block1] or (128, 512, 128) [block2].
 * The shortcut is read as rows of `ldr` elements, or (res_strided) as x[:, ::s, ::s] of an NHWC
 * tensor like hmmr_conv_desc_t's strided residual (ho, wo = output grid).
 * dtype HMMR_F16X3 (csrc/bottleneck_split.hip): the next conv1 always; conv2 in front (h1; w2 packed like every filter bank; stride 1, 8 x 8
 * pixel tiles: hin, win multiples of 8) for the 64 -> 256 -> 64 shape only; w3 and w1 are
 * FRAGMENT-MAJOR: [rows / 32][K / 16][64 lanes][hi 16 B | lo 16 B], lane = 32 * (k half) + row, the 16 bytes = the 8
 * bf16 of W[32 rb + row][16 kc + 8 half .. + 7] (one coalesced 2 KB read per MFMA A operand, straight from L2).  With xp
 * (and res == NULL) conv3's K is {h2, xp} against w3 = [W3 | Wsc], shift3 = the summed biases: the conv shortcut folded
 * into conv3 exactly as hmmr_conv_desc_t.in2 does.  Bit-identical to the launches it replaces in either form.
 * ------------------------------------------------------------------------- */
typedef struct {
    int dtype;                      /* HMMR_BF16 or HMMR_F16X3 */
    const void* h2; int m; int c_mid; int depth;
    const void* w3; const float* scale3; const float* shift3;      /* [depth][c_mid]; scale3 may be NULL */
    const void* res; int ldr; int res_strided; int64_t res_img_stride; int res_row_stride, res_px_stride; int ho, wo;
    void* out;                      /* [m][depth] */
    const float* pre_scale; const float* pre_shift;                /* [depth] */
    const void* w1; const float* scale1; const float* shift1; int relu1; int n2;   /* [n2][depth] */
    void* out_h1;                   /* [m][n2] */
    /* optional conv2 in front: h2 = NULL and h2 := relu(conv3x3(h1) * scale2 + shift2),
     * SAME padding, stride 1, computed per tile inside the same launch (slim bottleneck_v2 `conv2`) */
    const void* h1; int hin, win;   /* [m / (hin*win)][hin][win][c_mid] */
    const void* w2; const float* scale2; const float* shift2;      /* [c_mid][9 * c_mid], K = (ky, kx, ci) */
    /* optional conv shortcut computed in the launch (block1/unit_1: needs h1; res = NULL):
     * shortcut = xp x wsc + shift_sc, rounded to bf16 as the separate launch would store it */
    const void* xp;                 /* [m][64]: the (pre-activated) input of the unit */
    const void* wsc; const float* shift_sc;                        /* [depth][64], [depth] */
    /* conv2 stride (0/1 or 2; ho, wo = its output grid, m = images * ho * wo) and the single-phase form for a
     * block's stride-2 last unit: w1 == NULL -> no next conv1; `out` (raw trunk) and/or `out_pre`
     * (relu(out * pre_scale + pre_shift), what the next unit's conv1 AND conv shortcut read) are written */
    int conv2_stride;
    void* out_pre;                  /* [m][depth] or NULL */
    /* HMMR_F16X3, (c_mid, depth, n2) = (256, 1024, 256) [block3] or (128, 512, 128) [block2]: the register-resident unit pair
     * (csrc/unit_pair.hip).  w3 / w1 are not read; `pair_stream` holds both filter banks as the flat sequence of 2 KB MFMA
     * A-operand fragments ([hi plane: 64 lanes x 16 B][lo plane]; lane = 32 * (k half) + row, the 16 bytes = the 8 halves of
     * W[32 rb + row][16 kc + 8 half .. + 7], rows scaled like every split filter bank) the kernel consumes:
     * depth / 32 + 2 iterations; iteration `it` holds the kc3 = (c_mid + c_xp) / 16 fragments of conv3 row block `it` (K chunks in
     * order; zeros for it >= depth / 32) and the n2 / 16 fragments of conv1' K step it - 2 (K chunk 2 (it - 2) + {0, 1}, row blocks in
     * order; zeros for it < 2), fragment i of an iteration being a conv3 fragment when ((i + 1) kc3) / ft > (i kc3) / ft, ft = the
     * iteration's fragment count (hmmr_pair_stream_bytes; packing.pack_pair_stream).  With xp (c_xp = its channels, 256 for the
     * block2 shape; res == NULL) conv3's K is {h2, xp} as in the block1 form.  h2 only (no conv2 in front).  Bit-identical to
     * the launches it replaces. */
    const void* pair_stream;
    int c_xp;
    /* HMMR_F16X3, (c_mid, depth, n2) = (64, 256, 64) [block1] with conv2 in front (h1; hin x win whole images, win <= 56): the whole-unit
     * kernel of csrc/b1_unit.hip.  w2 / w3 / w1 are not read; `unit_stream` holds ALL filters of the unit as the flat sequence of 2 KB
     * MFMA A-operand fragments ([hi plane: 64 lanes x 16 B][lo plane], lane = 32 * (k half) + row, rows scaled like every split filter
     * bank) the kernel consumes (hmmr_b1_unit_stream_bytes; packing.pack_b1_unit_stream): conv2's stream exactly as k_order = 2 packs it
     * for cout = 64 (36 K steps kt = (ci / 16) * 9 + tap of two row blocks), then, with A(c) = the kc3 = (64 + c_xp) / 16 fragments of
     * conv3 row block c (32 output channels; K chunks in order) and B(c) = the four fragments of conv1' K chunks 2 c, 2 c + 1 (row blocks
     * 0, 1 of each): A(0) | A(1) B(0) | A(2) B(1) | ... | A(7) B(6) | B(7) (the order of the kernel's software pipeline).  scale2 / shift2: conv2's folded BN as the k_order 2 layer carries it.  With xp (c_xp = 64; res == NULL)
     * conv3's K is {h2, xp}.  Bit-identical to hmmr_conv_gemm(k_order 2) + conv3 + conv1 as three launches. */
    const void* unit_stream;
} hmmr_tail_desc_t;
size_t hmmr_pair_stream_bytes(int kc3, int depth, int n2);
size_t hmmr_b1_unit_stream_bytes(int c_xp);
/* measurement aid (bench.py's `roofline.mfma_sustained`): one launch of `workgroups` x 4 waves, one wave per SIMD, each issuing 8 * n8
 * v_mfma_f32_32x32x16_f16 (32768 FLOP each) on four independent accumulators and nothing else.  out: NULL or workgroups * 256 floats.
 * n8 < 0 (ABI 19): -n8 iterations on operands that CHANGE from MFMA to MFMA (four hashed fragment pairs in turn) -- the constant-operand loop is the
 * lowest-power case, so the clock the power cap allows it is higher than any real tensor gets. */
int hmmr_mfma_rate_probe(int workgroups, int n8, float* out, void* stream);
/* (ABI 19) measurement aid: one wave that stores n (s_memtime, s_memrealtime) pairs -- shader-clock ticks against the constant reference counter
 * (hipDeviceAttributeWallClockRate) -- `sleep` x ~4 us apart into `samples` (2 n words, zeroed by the caller), until n are taken or *stop (NULL or a device
 * int set from another stream) is non-zero.  Run on its own stream beside a measured region: the clock the part sustains under that load (bench.py). */
int hmmr_clock_probe(unsigned long long* samples, int n, int sleep, const int* stop, void* stream);
/* bytes of the filter stream of a k_order 2 layer of split tensors: (cout / 128) x 9 (cin / 16) K steps of 8 KB (bf16 tensors: half of it,
 * 9 (cin / 32) K steps) */
size_t hmmr_conv3x3_stream_bytes(int cin, int cout);
/* bytes of the filter stream of a 1x1 layer with k_order 2 (split tensors): (cout / 128) x (cin / 16) K steps of 8 KB */
size_t hmmr_conv1x1_stream_bytes(int cin, int cout);
int hmmr_bottleneck_tail(const hmmr_tail_desc_t* d, void* stream);

size_t hmmr_resnet50_workspace_bytes(int n, int dtype);
/* prof_ms: NULL, or a host array of HMMR_RESNET_PROF_SLOTS floats that receives
 * the HIP-event duration (ms) of every launch (forces a stream sync). */
#define HMMR_RESNET_PROF_SLOTS 64
/* n_zero: that many all-zero images are appended after the n real ones (the padding frames of
 * predict_all_images, tester.py:285-289, are zero IMAGES that still go through the encoder);
 * phi has n + n_zero rows, the workspace must be sized for n + n_zero. */
int hmmr_resnet50_fwd(const hmmr_resnet_weights_t* w, const float* images, int n, int n_zero,
                      float* phi, void* ws, size_t ws_bytes, void* stream, float* prof_ms);

/* ------------------------------------------------------------------------- *
 * f_movie temporal encoder: az_fc2_groupnorm / az_fc_block2
 * (src/models.py:121-228).  phi [b, t, 2048] fp32 -> strips [b, t, 2048] fp32.
 * ------------------------------------------------------------------------- */
typedef struct {
    const float* gn1_gamma; const float* gn1_beta;
    hmmr_layer_t conv1;                /* w [2048][3*2048], shift = bias */
    const float* gn2_gamma; const float* gn2_beta;
    hmmr_layer_t conv2;
} hmmr_temporal_block_t;

#define HMMR_MAX_TEMPORAL_BLOCKS 8

typedef struct {
    int dtype;
    int num_blocks;
    hmmr_temporal_block_t block[HMMR_MAX_TEMPORAL_BLOCKS];
} hmmr_temporal_weights_t;

size_t hmmr_temporal_workspace_bytes(int b, int t, int dtype);
int hmmr_temporal_fwd(const hmmr_temporal_weights_t* w, const float* phi, int b, int t,
                      float* strips, void* ws, size_t ws_bytes, void* stream);

/* Hallucinator fc2_res (src/models.py:270-296, pred_mode == 'hal'): phi [m,2048] fp32 ->
 * phi + fc3(relu(fc2(relu(fc1 phi)))) [m,2048] fp32; each w is [2048][2048], shift = bias. */
typedef struct {
    int dtype;
    hmmr_layer_t fc1, fc2, fc3;
} hmmr_hallucinator_weights_t;

size_t hmmr_hallucinator_workspace_bytes(int m, int dtype);
int hmmr_hallucinator_fwd(const hmmr_hallucinator_weights_t* w, const float* phi, int m,
                          float* out, void* ws, size_t ws_bytes, void* stream);

/* Standalone GroupNorm(+ReLU) over (time, channels-in-group), exposed for tests:
 * tf.contrib.layers.group_norm(reduction_axes=(-3,-2)), src/models.py:155-161. */
int hmmr_groupnorm_relu(const float* x, const float* gamma, const float* beta, int b, int t,
                        int c, int groups, void* out, int out_dtype, void* stream);

/* ------------------------------------------------------------------------- *
 * IEF regressors: batch_pred_omega / call_hmr_ief / hmr_ief /
 * encoder_fc3_dropout (src/models.py:80-116, 233-267, 299-415).
 * strips [m,2048] fp32 -> omega[r] [m,85] fp32 for r = 0 (present) and each
 * delta regressor, already in the final layout.  Default (all flags 0) = the
 * Tester configuration use_optcam=True, use_delta_from_pred=True
 * (tester.py:196-207): a delta regressor starts from omega0[:, 3:75], predicts
 * 72 values and returns [1,0,0 | pose | beta of omega0] (models.py:349-371).
 * no_optcam (use_optcam=False): it starts from [:, :75], predicts 75 values
 * and returns [cam, pose | beta] (models.py:357-373).  delta_from_start
 * (use_delta_from_pred=False): start and beta come from the IEF's own
 * starting point (omega_start / the mean theta) instead of omega0 (:349-351).
 * ------------------------------------------------------------------------- */
typedef struct {
    int nd;                            /* 85 (present); 72 (delta) or 75 (delta with no_optcam) */
    hmmr_layer_t fc1_phi;              /* w [1024][2048], shift = fc1 bias */
    hmmr_layer_t fc1_theta;            /* w [1024][128]: rows of fc1 for theta, zero padded */
    hmmr_layer_t fc2;                  /* w [1024][1024], shift = bias */
    hmmr_layer_t fc3;                  /* w [128][1024] (nd rows used), shift = bias padded */
} hmmr_ief_regressor_t;

#define HMMR_MAX_REGRESSORS 8

typedef struct {
    int dtype;
    int num_regressors;                /* [0] = present, then deltas in sorted delta_t order */
    int num_stages;                    /* 3 */
    hmmr_ief_regressor_t reg[HMMR_MAX_REGRESSORS];
    const float* mean_theta;           /* [85] */
    int no_optcam;                     /* 1: use_optcam=False (every delta regressor has nd == 75) */
    int delta_from_start;              /* 1: use_delta_from_pred=False */
} hmmr_ief_weights_t;

/* One set of layer buffers per delta regressor (num_regressors - 1 of them, at least one): the delta regressors run as
 * grouped launches, one per layer (hmmr_conv_desc_t.batch), when their filter banks / vectors lie at one common byte stride
 * -- always the case for two of them -- and one after the other otherwise; same results either way. */
size_t hmmr_ief_workspace_bytes(int m, int num_regressors, int dtype);
/* omegas: [num_regressors][m][85] fp32.  The IEF starts from w->mean_theta in every row (tester.py:79-83, 181). */
int hmmr_ief_fwd(const hmmr_ief_weights_t* w, const float* strips, int m, float* omegas,
                 void* ws, size_t ws_bytes, void* stream);
/* ... or from a caller-given starting point per row: omega_start [m][85] fp32 (`omega_mean` of batch_pred_omega,
 * src/models.py:231-249); NULL = w->mean_theta. */
int hmmr_ief_fwd_from(const hmmr_ief_weights_t* w, const float* strips, const float* omega_start, int m, float* omegas,
                      void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------- *
 * SMPL forward + keypoint projection: SMPL.__call__ (src/tf_smpl/batch_smpl.py:
 * 89-162), batch_rodrigues / batch_global_rigid_transformation
 * (src/tf_smpl/batch_lbs.py:42-60, 133-194), batch_orth_proj_idrot
 * (src/tf_smpl/projection.py:16-29), as driven by OmegasPred.compute_smpl
 * (src/omega.py:263-304).
 * ------------------------------------------------------------------------- */
typedef struct {
    int num_verts;                     /* 6890 */
    int num_kps;                       /* 25 (cocoplus) or 14 (lsp) */
    int lbs_nnz;                       /* ELL width of the skinning weights (<= 24) */
    int vpad;                          /* row stride of `dirs` in floats: a multiple of 128, >= num_verts */
    const float* dirs;                 /* [224][3][vpad] planar re-pack of the tf_smpl bases (rows >= 218 zero):
                                          row 0 v_template,
                                          1..10 shapedirs, 11..217 posedirs; dirs[k][c][v] =
                                          basis[k][3*v + c] (src/tf_smpl/batch_smpl.py:45-63) */
    const float* j_template;           /* [24*3]      J_regressor^T v_template            */
    const float* j_shapedirs;          /* [10][24*3]  J_regressor^T shapedirs (folded)    */
    const int32_t* parents;            /* [24], parents[0] ignored                         */
    const int32_t* lbs_idx;            /* [num_verts][lbs_nnz] joint ids                  */
    const float* lbs_w;                /* [num_verts][lbs_nnz] weights (0 for padding)    */
    const int32_t* kreg_ptr;           /* CSR by keypoint of cocoplus_regressor^T: [num_kps+1] */
    const int32_t* kreg_idx;           /* vertex ids   */
    const float* kreg_val;
    const void* dirs_split;            /* optional (NULL: the blend product runs on the vector units in fp32): `dirs` x 2^13 as fp16 hi/lo
                                          MFMA B-operand fragments, [224 / 16][3][2 (hi, lo)][2 (k half)][vpad][8 halves]:
                                          dirs_split[kc][c][plane][h][v][e] = plane(2^13 * dirs[16 kc + 8 h + e][c][v]), hi = fp16(x),
                                          lo = fp16(x - hi) (packing.pack_smpl).  With it the blend shapes are a split-fp16 MFMA
                                          product (three v_mfma_f32_32x32x16_f16 per operand pair, fp32 accumulate: 22 operand bits) */
} hmmr_smpl_consts_t;

size_t hmmr_smpl_workspace_bytes(int m);
/* theta: m rows of 72 floats with row stride ld_theta; beta: m rows of 10 floats
 * (stride ld_beta); cams: m rows of 3 floats (stride ld_cam) or NULL (no kps).
 * verts [m,V,3], joints [m,K,3], kps [m,K,2] (NULL ok), rs [m,24,3,3] (NULL ok). */
int hmmr_smpl_fwd(const hmmr_smpl_consts_t* c, const float* theta, int ld_theta,
                  const float* beta, int ld_beta, const float* cams, int ld_cam, int m,
                  float* verts, float* joints, float* kps, float* rs,
                  void* ws, size_t ws_bytes, void* stream);
/* Same, but instance i's outputs start at verts + i*ld_out, joints + i*ld_out, ... (floats):
 * the four outputs are fields of one packed per-frame record of ld_out floats, so the
 * record that the multi-GPU all-gather ships is written in place (no re-packing pass). */
int hmmr_smpl_fwd_strided(const hmmr_smpl_consts_t* c, const float* theta, int ld_theta,
                          const float* beta, int ld_beta, const float* cams, int ld_cam, int m,
                          float* verts, float* joints, float* kps, float* rs, int64_t ld_out,
                          void* ws, size_t ws_bytes, void* stream);
/* The tail of build_test_model for ALL containers at once (tester.py:196-227: OmegasPred.compute_all_smpl over the present
 * container and the delta containers + make_fetch_dict): omegas [num_containers][n][85] fp32 (container 0 = present, as
 * hmmr_ief_fwd writes them) -> every field of frame i's packed record rec[i * ld_rec ...]: container r's cams [3], joints
 * [K,3], kps [K,2], poses [24,3,3], shapes [10], verts [V,3] and raw omega [85] at float offset
 * field_offsets[r * 7 + {0..6}] (HOST array, that field order).  Every container is projected with container 0's camera
 * (tester.py:211-213), and `cams` holds that camera.  One launch set (three kernels) whatever num_containers is;
 * workspace = hmmr_smpl_workspace_bytes(num_containers * n). */
int hmmr_smpl_fwd_records(const hmmr_smpl_consts_t* c, const float* omegas, int num_containers, int n,
                          float* rec, int64_t ld_rec, const int32_t* field_offsets, void* ws, size_t ws_bytes,
                          void* stream);
/* batch_global_rigid_transformation on its own (src/tf_smpl/batch_lbs.py:133-194):
 * Rs [m,24,3,3], Js [m,24,3], parents [24] -> new_J [m,24,3], A [m,24,4,4] (relative transforms for LBS).
 * rotate_base != 0: the root rotation is R_0 . diag(1, -1, -1) (batch_lbs.py:151-158; the hot path passes 0). */
int hmmr_global_rigid_transformation(const float* Rs, const float* Js, const int32_t* parents, int m,
                                     float* new_j, float* A, int rotate_base, void* stream);

/* ------------------------------------------------------------------------- *
 * Crop before the path (process_image, src/evaluation/run_video.py:56-107; resize_img,
 * src/util/common.py:7-14): frames [n,h,w,3] uint8 -> out [n,224,224,3] fp32 in [-1,1].
 * geom [n][4] int32 = {scaled height, scaled width, u0, v0}: floor(h*scale), floor(w*scale) and the
 * scaled-image coordinates of crop pixel (0,0) (round(centre*factors) - 112, may be negative: the
 * edge padding).  The host mirror (evaluation/run_video.py) computes geom in float64.
 * ------------------------------------------------------------------------- */
int hmmr_crop_frames(const unsigned char* frames, const int32_t* geom, int n, int h, int w,
                     float* out, void* stream);

/* ------------------------------------------------------------------------- *
 * Hand-off to a rasteriser after the path (SURVEY f-3).  Replaces, for n frames in one launch, the
 * per-frame host code of src/util/render/nmr_renderer.py: the camera / keypoint change from the
 * 224x224 crop to the squared original image (visualize_img_orig :368-401) and the projection that
 * feeds nr.Renderer.render (VisRenderer.__call__ :139-144, torch_utils.py:11-29:
 * [s*(x+tx), -s*(y+ty), z]).  cams / verts / kps are read in place (row strides in floats, e.g.
 * straight out of the packed per-frame records of hmmr_smpl_fwd_strided).
 * geom [n][5] = {undo_scale, start_x, start_y, proc_size, img_size} per frame, or NULL to stay in
 * crop coordinates (visualize_img).  Outputs: new_cam [n][3] (may be NULL), proj_verts [n][nv][3],
 * kp_orig [n][nk][2] (may be NULL).
 * ------------------------------------------------------------------------- */
int hmmr_render_handoff(const float* cams, int64_t ld_cam, const float* verts, int64_t ld_verts,
                        const float* kps, int64_t ld_kps, const float* geom, int n, int nv, int nk,
                        float* new_cam, float* proj_verts, float* kp_orig, void* stream);

/* ------------------------------------------------------------------------- *
 * Evaluation metrics on device (src/evaluation/eval_util.py): per-frame MPJPE after pelvis alignment
 * and after Procrustes alignment (compute_error_3d :30-60 with align_by_pelvis :158 and
 * compute_similarity_transform :177), acceleration (compute_accel :14) and acceleration error
 * (compute_error_accel :63, before its visibility filter), vertex error (compute_error_verts :140).
 * gt / pred: [n, k, 3] fp32 (k <= 32); outputs [n] / [n-2] fp32; any output pointer may be NULL.
 * ------------------------------------------------------------------------- */
int hmmr_eval_joints(const float* gt, const float* pred, int n, int k, int left_id, int right_id,
                     float* mpjpe, float* pa_mpjpe, float* accel_pred, float* accel_err, void* stream);
/* err[i] = mean_v || gt[i, v] - pred[i, v] ||; row strides in floats (pred may be a field of the
 * packed per-frame record). */
int hmmr_eval_verts(const float* gt, int64_t ld_gt, const float* pred, int64_t ld_pred, int n, int nv,
                    float* err, void* stream);

/* ------------------------------------------------------------------------- *
 * Packers (ABI 18; csrc/pack.cpp): checkpoint variables by name -> the layouts and structs above, WITHOUT Python.
 *
 * The reference restores its graph from a TensorFlow checkpoint (src/evaluation/tester.py:92-116); the variable names and shapes are
 * SURVEY App. B's (resnet_v2_50/..., AZ_FC_block..., single_view_ief[_past5 | _future5]/3D_module/fc{1,2,3}, fc2_res/..., mean_param).  A caller
 * hands them over as fp32 host arrays in the checkpoint's own element order (conv filters HWIO, fully-connected [in][out]) and gets,
 * per stage, ONE blob in the device layout plus the stage's struct with every pointer already set to `device_base + offset`:
 *
 *     n = hmmr_pack_resnet_bytes(vars, n_vars, dtype);  hipMalloc(&dev, n);  host = malloc(n);
 *     hmmr_pack_resnet(vars, n_vars, dtype, host, n, dev, &weights);  hipMemcpy(dev, host, n, hipMemcpyHostToDevice);
 *     hmmr_resnet50_fwd(&weights, ...);
 *
 * No HIP call and no allocation in here; the functions are pure (same inputs -> same bytes) and return 0 or -1 with hmmr_last_error()
 * naming the variable that is missing or mis-sized.  They produce the SHIPPED configuration of every operand mode (f16x3: fused stem
 * with block1/unit_1's conv1, whole-unit kernels in block 1, unit pairs in blocks 2-3, the 3x3 / 1x1 filter streams, folded shortcuts;
 * bf16: fused units of blocks 1-2, 3x3 streams of blocks 3-4; f32: layer per launch) -- human_dynamics_amd/packing.py calls them for
 * exactly that and keeps Python forms only for its development switches, pinned to these bytes by tests/test_packers.py.
 * *_bytes(): the blob size for the same arguments (0 on error).
 * ------------------------------------------------------------------------- */
typedef struct {
    const char* name;      /* checkpoint variable name */
    const float* data;     /* fp32, host memory, the checkpoint's element order */
    int64_t numel;         /* number of values (the shape follows from the name; checked) */
} hmmr_var_t;

size_t hmmr_pack_resnet_bytes(const hmmr_var_t* vars, int n_vars, int dtype);
int hmmr_pack_resnet(const hmmr_var_t* vars, int n_vars, int dtype, void* host_blob, size_t blob_bytes, const void* device_base,
                     hmmr_resnet_weights_t* out);
size_t hmmr_pack_temporal_bytes(const hmmr_var_t* vars, int n_vars, int dtype, int num_blocks);
int hmmr_pack_temporal(const hmmr_var_t* vars, int n_vars, int dtype, int num_blocks, void* host_blob, size_t blob_bytes,
                       const void* device_base, hmmr_temporal_weights_t* out);
size_t hmmr_pack_hallucinator_bytes(const hmmr_var_t* vars, int n_vars, int dtype);
int hmmr_pack_hallucinator(const hmmr_var_t* vars, int n_vars, int dtype, void* host_blob, size_t blob_bytes, const void* device_base,
                           hmmr_hallucinator_weights_t* out);
/* delta_t: the delta regressors' offsets (config.delta_t_values, e.g. {-5, 5}): scopes single_view_ief_past5 / _future5; packed in sorted order */
size_t hmmr_pack_ief_bytes(const hmmr_var_t* vars, int n_vars, int dtype, const int* delta_t, int n_delta);
int hmmr_pack_ief(const hmmr_var_t* vars, int n_vars, int dtype, const int* delta_t, int n_delta, int num_stages, void* host_blob,
                  size_t blob_bytes, const void* device_base, hmmr_ief_weights_t* out);

/* The body model in the src/tf_smpl layout (batch_smpl.py:35-80: what SMPL.__init__ builds from the pkl, or the checkpoint's own
 * non-trainable variables of the same names), fp32 host arrays */
typedef struct {
    int num_verts;                 /* 6890 */
    int num_kps;                   /* columns of kp_regressor: 25 (19 cocoplus + 6) */
    const float* v_template;       /* [num_verts][3] */
    const float* shapedirs;        /* [10][3 num_verts]: column 3 v + c */
    const float* posedirs;         /* [207][3 num_verts] */
    const float* J_regressor;      /* [num_verts][24] (stored transposed, batch_smpl.py:51-55) */
    const float* lbs_weights;      /* [num_verts][24] */
    const float* kp_regressor;     /* cocoplus_regressor [num_verts][num_kps] */
    const int32_t* parents;        /* [24] */
} hmmr_smpl_source_t;
/* lsp != 0: joint_type 'lsp', the first 14 keypoints (batch_smpl.py:81-82); split != 0: also the split-fp16 MFMA form of the blend basis
 * (hmmr_smpl_consts_t.dirs_split; 0 for an all-fp32 engine) */
size_t hmmr_pack_smpl_bytes(const hmmr_smpl_source_t* src, int lsp, int split);
int hmmr_pack_smpl(const hmmr_smpl_source_t* src, int lsp, int split, void* host_blob, size_t blob_bytes, const void* device_base,
                   hmmr_smpl_consts_t* out);

#ifdef __cplusplus
}
#endif
#endif /* HMMR_HIP_H */
