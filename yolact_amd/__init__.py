"""yolact_amd: MI355X-native YOLACT post-processing (Detect) behind a C ABI.

Public surface: `Detect` (drop-in for the reference class), `PostProcessor`/`PostConfig`
(one yp_handle), `anchors.make_priors`, `synth` (synthetic head outputs of the BASELINE configs).
"""
from .detect import Detect, PostConfig, PostProcessor  # noqa: F401
from . import anchors, synth  # noqa: F401
