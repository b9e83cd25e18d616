"""Mirror of src/datasets/resnet_extractor.py: ResNet-only feature extraction
(BASELINE config 2).  ``FeatureExtractor(model_path, img_size=224, batch_size=64)``
with ``compute_phis(images[B,H,W,3]) -> [B,2048]`` and
``compute_all_phis(all_images[T,H,W,3]) -> [T,2048]`` (zero-padded tail batch,
resnet_extractor.py:74-98)."""
from __future__ import annotations

import numpy as np
import torch

from ..engine import DEFAULT_DTYPE, HmmrEngine
from ..evaluation.tester import load_weights


class FeatureExtractor(object):
    def __init__(self, model_path, img_size=224, batch_size=64, sess=None, weights=None,
                 dtype=DEFAULT_DTYPE, device="cuda:0"):
        if img_size != 224:
            raise ValueError("the ResNet stage is built for 224x224 crops")
        self.model_path = model_path
        self.img_size = img_size
        self.batch_size = batch_size
        if weights is None:
            weights = load_weights(model_path)
        self.engine = HmmrEngine(weights, None, dtype=dtype, device=device)

    def compute_phis(self, images):
        """images (BxHxWx3) -> phis (Bx2048) float32 ndarray."""
        phi = self.engine.resnet(np.asarray(images, np.float32))
        torch.cuda.synchronize(self.engine.device)
        return phi.cpu().numpy()

    def compute_all_phis(self, all_images):
        all_phis = []
        T = len(all_images)
        for i in range(0, T, self.batch_size):
            images = np.asarray(all_images[i:i + self.batch_size], np.float32)
            if len(images) < self.batch_size:          # pad the last batch with zeros
                leftover = self.batch_size - len(images)
                pad = np.zeros((leftover, self.img_size, self.img_size, 3), np.float32)
                images = np.vstack((images, pad))
            all_phis.append(self.compute_phis(images))
        return np.vstack(all_phis)[:T]
