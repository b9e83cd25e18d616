class Config(object):
    """the duck-typed config of Tester (tests/conftest.py's, for the tools)"""
    def __init__(self, **kw):
        self.load_path, self.batch_size, self.sequence_length, self.pred_mode = "synthetic:0", 8, 20, "pred"
        self.num_conv_layers, self.delta_t_values, self.smpl_model_path, self.num_kps = 3, ["-5", "5"], "synthetic:2", 25
        self.__dict__.update(kw)
