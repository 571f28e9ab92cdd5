"""openpano_amd -- MI355X-native (gfx950) hot path of the OpenPano panorama stitcher.

Only the data-parallel path is here: SIFT feature extraction, all-pairs
descriptor matching + RANSAC, warp + blend, as hand-written HIP kernels behind the C-ABI in
``include/openpano_hip.h``.  The Python layer mirrors the reference's class surface
(``FeatureDetector.detect_feature``, ``PairWiseMatcher``, ...) over that C-ABI; there is no CPU
fallback -- importing :mod:`openpano_amd.hip` fails loudly if the HIP library is missing.
"""
from .config import PanoConfig, DEFAULTS  # noqa: F401

__all__ = ["PanoConfig", "DEFAULTS"]
