"""SMPL forward on gfx950 behind the reference's src/tf_smpl interface."""
