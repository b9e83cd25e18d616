"""human_dynamics_amd -- MI355X-native (gfx950) implementation of HMMR's
per-frame inference hot path (ResNet-v2-50 -> f_movie temporal encoder ->
IEF regressors -> SMPL), behind the reference's ``Tester.predict`` surface.

Reference: akanazawa/human_dynamics, src/evaluation/tester.py:24-312.

Only the hot path lives here; the compute is hand-written HIP
behind the C-ABI library ``libhmmr_hip.so`` (include/hmmr_hip.h).  There is no
CPU fallback: importing the compute modules without the built library raises.
"""

__version__ = "0.1.0"
