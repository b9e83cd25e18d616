"""TF-slim's ResNet-v2 network definition, transcribed -- TEST INFRASTRUCTURE ONLY.

The reference reaches its image encoder through one call (src/models.py:65-75):

    from tensorflow.contrib.slim.python.slim.nets import resnet_v2
    with slim.arg_scope(resnet_v2.resnet_arg_scope(weight_decay=weight_decay)):
        net, end_points = resnet_v2.resnet_v2_50(x, num_classes=None, is_training=is_training,
                                                 reuse=reuse, scope='resnet_v2_50')

That code lives in TensorFlow (tensorflow/contrib/slim/python/slim/nets/resnet_utils.py and
resnet_v2.py, TF r1.8 -- the version the reference pins), not in the reference tree, and TensorFlow
cannot be installed in this image.  This file is a function-for-function transcription of those two
files, kept SEPARATE from the oracle's own restatement (oracle/hmmr_oracle.resnet_v2_50, written as
PyTorch convolutions from SURVEY.md App. A) so that the two can be checked against each other:
tests/golden/make_resnet_golden.py runs the reference's own `encoder_resnet` on this transcription
(through oracle/tf_shim.py's slim layers) and freezes the result as tests/golden/reference_resnet.npz.

Upstream locations are given per function as `file: function`; line numbers are those of the r1.8
sources as recalled and may be off by a few lines -- TensorFlow is not available here to check
them against.  Only the inference-relevant arguments are transcribed: `output_stride` (atrous
rates) must be None, as it is at the reference's call site.
"""
from __future__ import annotations

import collections
import types

from oracle import tf_shim as _tf

# the aliases the upstream files import (layers = layers_lib = tf.contrib.layers.python.layers.layers)
arg_scope, add_arg_scope = _tf.arg_scope, _tf.add_arg_scope
utils = _tf.slim_utils
variable_scope = _tf


class _Layers(object):
    conv2d = staticmethod(_tf._conv2d)
    batch_norm = staticmethod(_tf._batch_norm)
    max_pool2d = staticmethod(_tf._max_pool2d)


layers = layers_lib = _Layers


# =============================================================================================== #
# resnet_utils.py
# =============================================================================================== #
class Block(collections.namedtuple("Block", ["scope", "unit_fn", "args"])):
    """resnet_utils.py: class Block (~l.50-62): scope, unit function, list of per-unit kwargs."""


def subsample(inputs, factor, scope=None):
    """resnet_utils.py: subsample (~l.65-81): identity for factor 1, else a 1x1 max-pool with that
    stride, i.e. inputs[:, ::factor, ::factor, :]."""
    if factor == 1:
        return inputs
    else:
        return layers.max_pool2d(inputs, [1, 1], stride=factor, scope=scope)


def conv2d_same(inputs, num_outputs, kernel_size, stride, rate=1, scope=None):
    """resnet_utils.py: conv2d_same (~l.84-131): stride 1 -> 'SAME' conv; stride > 1 -> explicit
    zero padding of kernel_size_effective - 1 (the smaller half first) and a 'VALID' conv, so that the
    result does not depend on the input size's parity."""
    if stride == 1:
        return layers_lib.conv2d(inputs, num_outputs, kernel_size, stride=1, rate=rate, padding="SAME", scope=scope)
    else:
        kernel_size_effective = kernel_size + (kernel_size - 1) * (rate - 1)
        pad_total = kernel_size_effective - 1
        pad_beg = pad_total // 2
        pad_end = pad_total - pad_beg
        inputs = _tf.pad(inputs, [[0, 0], [pad_beg, pad_end], [pad_beg, pad_end], [0, 0]])
        return layers_lib.conv2d(inputs, num_outputs, kernel_size, stride=stride, rate=rate, padding="VALID",
                                 scope=scope)


@add_arg_scope
def stack_blocks_dense(net, blocks, output_stride=None, outputs_collections=None):
    """resnet_utils.py: stack_blocks_dense (~l.134-213) with output_stride=None: every block in its own
    variable scope, its units in 'unit_%d' scopes (1-based), each unit called with rate=1 and its kwargs."""
    assert output_stride is None, "atrous mode is not used by the reference"
    current_stride = 1
    for block in blocks:
        with _tf.variable_scope(block.scope, "block", [net]) as sc:
            for i, unit in enumerate(block.args):
                with _tf.variable_scope("unit_%d" % (i + 1), values=[net]):
                    net = block.unit_fn(net, rate=1, **unit)
                    current_stride *= unit.get("stride", 1)
            net = utils.collect_named_outputs(outputs_collections, sc.name, net)
    return net


def resnet_arg_scope(weight_decay=0.0001, batch_norm_decay=0.997, batch_norm_epsilon=1e-5, batch_norm_scale=True):
    """resnet_utils.py: resnet_arg_scope (~l.216-262): conv2d gets ReLU + batch_norm (decay 0.997,
    epsilon 1e-5, scale=True) + He init + L2; max_pool2d gets padding='SAME'."""
    batch_norm_params = {
        "decay": batch_norm_decay,
        "epsilon": batch_norm_epsilon,
        "scale": batch_norm_scale,
        "updates_collections": "update_ops",
    }
    with arg_scope([layers_lib.conv2d],
                   weights_regularizer=None,            # l2_regularizer(weight_decay): training only
                   weights_initializer=None,            # variance_scaling_initializer(): training only
                   activation_fn=_tf.nn.relu,
                   normalizer_fn=layers.batch_norm,
                   normalizer_params=batch_norm_params):
        with arg_scope([layers.batch_norm], **batch_norm_params):
            # 'SAME' here makes the feature maps of the root block and of conv2d_same line up
            with arg_scope([layers.max_pool2d], padding="SAME") as arg_sc:
                return arg_sc


resnet_utils = types.ModuleType("resnet_utils")
for _n in ("Block", "subsample", "conv2d_same", "stack_blocks_dense", "resnet_arg_scope"):
    setattr(resnet_utils, _n, globals()[_n])


# =============================================================================================== #
# resnet_v2.py
# =============================================================================================== #
@add_arg_scope
def bottleneck(inputs, depth, depth_bottleneck, stride, rate=1, outputs_collections=None, scope=None):
    """resnet_v2.py: bottleneck (~l.71-127), the full pre-activation unit:
    preact = relu(BN(inputs)); shortcut = subsample(INPUTS) when the depth is unchanged, else a 1x1 conv
    of PREACT with that stride and neither normaliser nor activation (so: a bias); residual = 1x1 conv
    (BN, ReLU) -> 3x3 conv2d_same(stride) (BN, ReLU) -> 1x1 conv without normaliser / activation (bias);
    output = shortcut + residual, no ReLU."""
    with _tf.variable_scope(scope, "bottleneck_v2", [inputs]) as sc:
        depth_in = utils.last_dimension(inputs.get_shape(), min_rank=4)
        preact = layers.batch_norm(inputs, activation_fn=_tf.nn.relu, scope="preact")
        if depth == depth_in:
            shortcut = resnet_utils.subsample(inputs, stride, "shortcut")
        else:
            shortcut = layers_lib.conv2d(preact, depth, [1, 1], stride=stride, normalizer_fn=None, activation_fn=None,
                                         scope="shortcut")

        residual = layers_lib.conv2d(preact, depth_bottleneck, [1, 1], stride=1, scope="conv1")
        residual = resnet_utils.conv2d_same(residual, depth_bottleneck, 3, stride, rate=rate, scope="conv2")
        residual = layers_lib.conv2d(residual, depth, [1, 1], stride=1, normalizer_fn=None, activation_fn=None,
                                     scope="conv3")

        output = shortcut + residual

        return utils.collect_named_outputs(outputs_collections, sc.name, output)


def resnet_v2(inputs, blocks, num_classes=None, is_training=True, global_pool=True, output_stride=None,
              include_root_block=True, reuse=None, scope=None):
    """resnet_v2.py: resnet_v2 (~l.130-231): root block = conv2d_same(64, 7, stride 2, 'conv1') WITHOUT
    normaliser / activation (the first unit's preact does that) + max_pool2d([3,3], stride 2, 'pool1');
    the blocks; 'postnorm' BN + ReLU; global average pool 'pool5' (keepdims); logits only when
    num_classes is given."""
    with _tf.variable_scope(scope, "resnet_v2", [inputs], reuse=reuse) as sc:
        end_points_collection = sc.original_name_scope + "_end_points"
        with arg_scope([layers_lib.conv2d, bottleneck, resnet_utils.stack_blocks_dense],
                       outputs_collections=end_points_collection):
            with arg_scope([layers.batch_norm], is_training=is_training):
                net = inputs
                if include_root_block:
                    assert output_stride is None
                    # We do not include batch normalization or activation functions in conv1 because the first
                    # ResNet unit will perform these.  (upstream comment)
                    with arg_scope([layers_lib.conv2d], activation_fn=None, normalizer_fn=None):
                        net = resnet_utils.conv2d_same(net, 64, 7, stride=2, scope="conv1")
                    net = layers.max_pool2d(net, [3, 3], stride=2, scope="pool1")
                net = resnet_utils.stack_blocks_dense(net, blocks, output_stride)
                # This is needed because the pre-activation variant does not have batch normalization or
                # activation functions in the residual unit output.  (upstream comment)
                net = layers.batch_norm(net, activation_fn=_tf.nn.relu, scope="postnorm")
                if global_pool:
                    # Global average pooling.
                    net = _tf.reduce_mean(net, [1, 2], name="pool5", keepdims=True)
                if num_classes is not None:
                    net = layers_lib.conv2d(net, num_classes, [1, 1], activation_fn=None, normalizer_fn=None,
                                            scope="logits")
                # Convert end_points_collection into a dictionary of end_points.
                end_points = utils.convert_collection_to_dict(end_points_collection)
                return net, end_points


def resnet_v2_block(scope, base_depth, num_units, stride):
    """resnet_v2.py: resnet_v2_block (~l.235-258): num_units - 1 units of stride 1, then ONE unit
    carrying the block's stride (the stride sits on the LAST unit)."""
    return resnet_utils.Block(scope, bottleneck, [{
        "depth": base_depth * 4,
        "depth_bottleneck": base_depth,
        "stride": 1
    }] * (num_units - 1) + [{
        "depth": base_depth * 4,
        "depth_bottleneck": base_depth,
        "stride": stride
    }])


def resnet_v2_50(inputs, num_classes=None, is_training=True, global_pool=True, output_stride=None, reuse=None,
                 scope="resnet_v2_50"):
    """resnet_v2.py: resnet_v2_50 (~l.264-283)."""
    blocks = [
        resnet_v2_block("block1", base_depth=64, num_units=3, stride=2),
        resnet_v2_block("block2", base_depth=128, num_units=4, stride=2),
        resnet_v2_block("block3", base_depth=256, num_units=6, stride=2),
        resnet_v2_block("block4", base_depth=512, num_units=3, stride=1),
    ]
    return resnet_v2(inputs, blocks, num_classes, is_training, global_pool, output_stride, include_root_block=True,
                     reuse=reuse, scope=scope)


# what `from tensorflow.contrib.slim.python.slim.nets import resnet_v2` binds: the MODULE
resnet_v2_module = types.ModuleType("resnet_v2")
for _n in ("bottleneck", "resnet_v2", "resnet_v2_block", "resnet_v2_50"):
    setattr(resnet_v2_module, _n, globals()[_n])
resnet_v2_module.resnet_arg_scope = resnet_utils.resnet_arg_scope   # resnet_v2.py: `resnet_arg_scope = resnet_utils.resnet_arg_scope`
resnet_v2_module.resnet_utils = resnet_utils
