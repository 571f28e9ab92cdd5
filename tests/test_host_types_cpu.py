"""CPU: the standalone C++ value types of the host mirror (openpano_amd/host/pano_types.hh).
MatchInfo text form (stitch/match_info.hh:26-50, stitch/debug.cc:111-140) against the reference's
own MatchInfo::serialize compiled in place (oracle/_ref)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "openpano_amd", "host")


def test_matchinfo_text_roundtrip(ref, tmp_path):
    exe = tmp_path / "types_selftest"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", HOST, "-o", str(exe), os.path.join(HOST, "types_selftest.cc")])
    rng = np.random.default_rng(4)
    conf = 0.3127
    homo = (rng.normal(0, 1, 9) * [1, 1, 300, 1, 1, 200, 1e-4, 1e-4, 1]).tolist()
    pts = (rng.uniform(-600, 600, (5, 4))).round(3)
    args = [repr(conf)] + [repr(float(x)) for x in homo] + ["5"] + [repr(float(x)) for x in pts.reshape(-1)] + [str(tmp_path / "mi.txt")]
    out = subprocess.run([str(exe)] + args, capture_output=True, text=True, check=True).stdout.splitlines()
    buf = C.create_string_buffer(4096)
    ref.lib.ref_matchinfo_serialize.argtypes = [C.c_float, np.ctypeslib.ndpointer(np.float64), np.ctypeslib.ndpointer(np.float64), C.c_int, C.c_char_p, C.c_int]
    n = ref.lib.ref_matchinfo_serialize(conf, np.array(homo, np.float64), np.ascontiguousarray(pts, np.float64).reshape(-1), 5, buf, 4096)
    want = buf.value.decode()
    assert n > 0 and out[0] == want                    # same text as the reference writes
    # the reloaded record serialises to the same text again (the format is its own fixed point)
    assert out[1] == want and out[2] == "1"
    text = open(tmp_path / "mi.txt").read().splitlines()
    assert text[0] == "0 2" and text[1] == want and text[2] == "2 0"
