"""Mirror of src/models.py's model-fn registry (the reference's only plugin
API): three string-keyed factories returning callables, plus
``batch_pred_omega``.  The callables take/return device tensors and run the
HIP stages of an ``HmmrEngine`` instead of building TF ops; TF-only arguments
(`is_training`, `reuse`, `weight_decay`) are accepted and must describe
inference.
"""
from __future__ import annotations


def get_image_encoder(model_type="resnet"):
    """src/models.py:12-23."""
    models = {"resnet": encoder_resnet}
    if model_type in models:
        return models[model_type]
    raise ValueError("Unknown image encoder: %s" % model_type)


def get_hallucinator_model(model_type="fc2_res"):
    """src/models.py:26-34."""
    models = {"fc2_res": fc2_res}
    if model_type in models:
        return models[model_type]
    raise ValueError("Unknown predict hal model: %s" % model_type)


def get_temporal_encoder(model_type="AZ_FC2GN"):
    """src/models.py:37-45."""
    models = {"AZ_FC2GN": az_fc2_groupnorm}
    if model_type in models:
        return models[model_type]
    raise ValueError("Unknown temporal encoder: %s" % model_type)


def _inference_only(is_training):
    if is_training:
        raise NotImplementedError("the MI355X path implements inference only (is_training must be False)")


def encoder_resnet(x, engine, is_training=False, weight_decay=0.001, reuse=False):
    """x [N,224,224,3] -> (phi [N,2048], 'resnet_v2_50')  (src/models.py:50-77)."""
    _inference_only(is_training)
    return engine.resnet(x), "resnet_v2_50"


def az_fc2_groupnorm(is_training, net, num_conv_layers, engine):
    """net [B,T,2048] -> movie strips [B,T,2048]  (src/models.py:121-141)."""
    _inference_only(is_training)
    if num_conv_layers != engine.num_conv_layers:
        raise ValueError("engine was packed with num_conv_layers=%d" % engine.num_conv_layers)
    return engine.temporal(net)


def fc2_res(phi, engine, name="fc2_res"):
    """phi [B,T,2048] -> hallucinated movie strip [B,T,2048]: phi + fc3(relu(fc2(relu(fc1 phi))))
    (src/models.py:270-296, pred_mode == 'hal')."""
    return engine.hallucinate(phi)


def batch_pred_omega(input_features, batch_size, is_training, num_output, omega_mean,
                     sequence_length, scope, engine, predict_delta_keys=(),
                     use_delta_from_pred=False, use_optcam=False):
    """[B,T,2048] -> (omega [B,T,85], {delta_t: [B,T,85]})  (src/models.py:233-267, call_hmr_ief :299-377).

    omega_mean: [B*T,85] starting point of the IEF (None = the checkpoint's mean theta in every row, what
    tester.py:79-83,181 passes).  use_delta_from_pred: the delta regressors start from the present prediction (True) or
    from omega_mean (False, models.py:349).  use_optcam: they regress 72 values and get the camera [1,0,0] (True) or
    regress camera + pose, 75 values (False, models.py:333-373) -- a property of the checkpoint's fc1 / fc3 shapes, so
    the argument is validated against the packed regressors.  `scope` is validated too."""
    _inference_only(is_training)
    if num_output != 85:
        raise ValueError("batch_pred_omega: num_output is 85 (3 camera + 72 pose + 10 shape), got %r" % (num_output,))
    if scope != "single_view_ief":
        raise ValueError("the engine's regressors were packed from scope 'single_view_ief', got %r" % (scope,))
    keys = [k for k in sorted(predict_delta_keys) if k != 0]
    if keys != [k for k in engine.reg_keys if k != 0]:
        raise ValueError("engine was packed for delta keys %s" % engine.reg_keys)
    if keys and bool(use_optcam) != engine.use_optcam:
        raise ValueError("use_optcam=%s, but the checkpoint's delta regressors are %d-wide" %
                         (use_optcam, 72 if engine.use_optcam else 75))
    om = engine.ief(input_features.reshape(batch_size * sequence_length, -1), omega_start=omega_mean,
                    use_delta_from_pred=use_delta_from_pred)
    om = om.reshape(om.shape[0], batch_size, sequence_length, 85)
    return om[0], {k: om[i] for i, k in enumerate(engine.reg_keys) if k != 0}
