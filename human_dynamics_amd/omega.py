"""Mirror of src/omega.py's prediction container for device tensors.

``OmegasPred`` slices the 85-D omega into cams [0:3], axis-angle pose [3:75]
and shape [75:85] (src/omega.py:231-235), runs SMPL once per container
(:263-304) and projects keypoints with the container's camera (:287-288,
:318-323).  Unlike the reference the instance registry is per ``Tester`` (a
list handed in by the owner) instead of a process-global class attribute, and
``compute_all_smpl`` evaluates all containers in ONE SMPL launch sequence.
"""
from __future__ import annotations

import torch


class OmegasPred(object):
    def __init__(self, config, smpl_engine, use_optcam=False, vis_max_batch=2, batch_size=None,
                 is_training=False, registry=None):
        if is_training:
            raise NotImplementedError("inference only")
        self.config = config
        self.engine = smpl_engine
        self.batch_size = batch_size if batch_size else config.batch_size
        self.use_optcam = use_optcam
        self.vis_max_batch = vis_max_batch
        self.raw = None
        self.cams = None
        self.length = 0
        self.smpl_computed = False
        self.joints = self.kps = self.poses_rot = self.verts = self.all_verts = None
        if registry is not None:
            registry.append(self)

    # -- src/omega.py:231-248 --------------------------------------------------
    def update_instance_vars(self):
        self.cams = self.raw[:, :, :3]
        self.poses_aa = self.raw[:, :, 3:3 + 24 * 3]
        self.shapes = self.raw[:, :, 3 + 24 * 3:85]
        self.length = self.raw.shape[1]

    def append_batched(self, omegas):
        B = self.batch_size
        omegas = omegas.reshape(B, -1, 85)
        self.raw = omegas if self.raw is None else torch.cat((self.raw, omegas), dim=1)
        self.update_instance_vars()
        self.smpl_computed = False

    def set_cams(self, cams):
        assert self.use_optcam                       # src/omega.py:322
        self.cams = cams

    # -- src/omega.py:263-304 --------------------------------------------------
    def _store(self, verts, joints, kps, rs):
        B, T = self.batch_size, self.length
        K = joints.shape[1]
        if K != self.config.num_kps:
            raise ValueError("SMPL regressor yields %d keypoints, config.num_kps=%d" % (K, self.config.num_kps))
        self.joints = joints.reshape(B, T, K, 3)
        self.kps = kps.reshape(B, T, K, 2)
        self.poses_rot = rs.reshape(B, T, 24, 3, 3)
        self.all_verts = verts.reshape(B, T, -1, 3)[:self.vis_max_batch]
        self.verts = self.all_verts
        self.smpl_computed = True

    def compute_smpl(self):
        B, T = self.batch_size, self.length
        raw = self.raw.reshape(B * T, 85)
        cams = self.cams.reshape(B * T, 3).contiguous()
        self._store(*self.engine.smpl(raw[:, 3:75], raw[:, 75:85], cams))

    @staticmethod
    def compute_all_smpl(instances):
        """One batched SMPL evaluation for every container (src/omega.py:338-342)."""
        if not instances:
            return
        eng = instances[0].engine
        raws = torch.cat([o.raw.reshape(-1, 85) for o in instances], dim=0)
        cams = torch.cat([o.cams.reshape(-1, 3) for o in instances], dim=0).contiguous()
        verts, joints, kps, rs = eng.smpl(raws[:, 3:75], raws[:, 75:85], cams)
        off = 0
        for o in instances:
            n = o.batch_size * o.length
            o._store(verts[off:off + n], joints[off:off + n], kps[off:off + n], rs[off:off + n])
            off += n

    # -- getters, src/omega.py:54-140, 306-336 ----------------------------------
    def get_cams(self, t=None):
        return self.cams if t is None else self.cams[:, t]

    def get_joints(self, t=None):
        return self.joints if t is None else self.joints[:, t]

    def get_kps(self, t=None):
        return self.kps if t is None else self.kps[:, t]

    def get_poses_aa(self, t=None):
        return self.poses_aa if t is None else self.poses_aa[:, t]

    def get_poses_rot(self, t=None):
        return self.poses_rot if t is None else self.poses_rot[:, t]

    def get_shapes(self, t=None):
        return self.shapes if t is None else self.shapes[:, t]

    def get_all_verts(self):
        return self.all_verts

    def get_verts(self):
        return self.verts

    def get_raw(self):
        return self.raw

    def __len__(self):
        return self.length
