"""Configuration mirror of the reference's ``namespace config`` (lib/config.hh:24-85).

The reference reads ``config.cfg`` (``KEY value`` lines, every value parsed as a *float*,
lib/config.cc:13-29) and copies the values into typed globals in ``init_config()``
(main.cc:237-292).  ``DEFAULTS`` holds the shipped ``src/config.cfg`` values; ``PanoConfig``
applies the same float -> int/bool/double narrowing, so the numbers that reach the kernels are
the ones the reference CPU path would use.
"""
from __future__ import annotations

import numpy as np

# src/config.cfg:2-69 (shipped defaults)
DEFAULTS = {
    "CYLINDER": 0, "ESTIMATE_CAMERA": 1, "TRANS": 0,
    "ORDERED_INPUT": 0, "CROP": 1, "MAX_OUTPUT_SIZE": 8000, "LAZY_READ": 1,
    "FOCAL_LENGTH": 37,
    "SIFT_WORKING_SIZE": 800, "NUM_OCTAVE": 4, "NUM_SCALE": 7,
    "SCALE_FACTOR": 1.4142135623, "GAUSS_SIGMA": 1.4142135623, "GAUSS_WINDOW_FACTOR": 6,
    "CONTRAST_THRES": 4e-2, "JUDGE_EXTREMA_DIFF_THRES": 2e-3, "EDGE_RATIO": 6,
    "PRE_COLOR_THRES": 5e-2, "CALC_OFFSET_DEPTH": 4, "OFFSET_THRES": 0.5,
    "ORI_RADIUS": 4.5, "ORI_HIST_SMOOTH_COUNT": 2,
    "DESC_HIST_SCALE_FACTOR": 3, "DESC_INT_FACTOR": 512,
    "MATCH_REJECT_NEXT_RATIO": 0.8,
    "RANSAC_ITERATIONS": 1500, "RANSAC_INLIER_THRES": 3.5,
    "INLIER_IN_MATCH_RATIO": 0.1, "INLIER_IN_POINTS_RATIO": 0.04,
    "STRAIGHTEN": 1, "SLOPE_PLAIN": 8e-3, "LM_LAMBDA": 5, "MULTIPASS_BA": 1,
    "MULTIBAND": 0,
}

_INT_KEYS = {
    "MAX_OUTPUT_SIZE", "SIFT_WORKING_SIZE", "NUM_OCTAVE", "NUM_SCALE", "GAUSS_WINDOW_FACTOR",
    "CALC_OFFSET_DEPTH", "ORI_HIST_SMOOTH_COUNT", "DESC_HIST_SCALE_FACTOR", "DESC_INT_FACTOR",
    "RANSAC_ITERATIONS", "MULTIPASS_BA", "MULTIBAND",
}
_BOOL_KEYS = {"CYLINDER", "ESTIMATE_CAMERA", "TRANS", "ORDERED_INPUT", "CROP", "LAZY_READ", "STRAIGHTEN"}


def parse_config_file(path: str) -> dict:
    """``ConfigParser`` (lib/config.cc:13-29): ``KEY value`` per line, ``#`` starts a comment."""
    out = {}
    with open(path) as f:
        for line in f:
            toks = line.split("#", 1)[0].split()
            if len(toks) >= 2:
                out[toks[0]] = float(np.float32(float(toks[1])))
    return out


class PanoConfig:
    """Typed view of the config, narrowed exactly like ``init_config()`` (main.cc:237-292)."""

    def __init__(self, **overrides):
        vals = dict(DEFAULTS)
        for k, v in overrides.items():
            if k not in vals:
                raise KeyError(f"Option {k} not found in config")  # lib/config.cc:31-35
            vals[k] = v
        self._raw = {k: np.float32(v) for k, v in vals.items()}
        for k, v in self._raw.items():
            if k in _BOOL_KEYS:
                setattr(self, k, bool(v))
            elif k in _INT_KEYS:
                setattr(self, k, int(v))
            else:
                setattr(self, k, float(v))  # float (or double from float for RANSAC_INLIER_THRES)
        modes = int(self.CYLINDER) + int(self.TRANS) + int(self.ESTIMATE_CAMERA)
        if modes >= 2:
            raise ValueError("You set two many modes...")  # main.cc:245-246
        if not self.ORDERED_INPUT and not self.ESTIMATE_CAMERA:
            raise ValueError("Require ORDERED_INPUT under this mode!")  # main.cc:257-258

    def raw_items(self):
        """(key, float32 value) pairs, as ``ConfigParser.get`` would return them."""
        return self._raw.items()
