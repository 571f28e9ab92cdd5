"""CPU: the acceptance epilogue's point-in-polygon test (openpano_amd/csrc/ransac_accept.hpp) answers the reference's
question -- k = (float)atan2(...), lower_bound over the sorted vertex angles, side of that wedge's edge (lib/polygon.cc:62-82)
-- without libm's atan2 for nearly every point and with the reference's own expression for the rest.  The harness
(tests/harness/ransac_accept_harness.cc, plain g++) measures fast_atan2 against libm and compares the two paths on overlap
polygons of random homographies: random points, points on every vertex direction and a few float ulps off them -- and the vector form of the count the
epilogue uses (count_in_polygon, ransac_accept_simd.cc) returns the sum of the reference's answers over those points."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fast_point_in_polygon_is_the_references(tmp_path):
    gxx = shutil.which("g++")
    if not gxx or not os.path.isdir("/opt/rocm/include"):
        pytest.skip("g++ or the HIP headers (ransac_math.hpp includes hip_runtime.h for its __host__ __device__ markers) missing")
    exe = str(tmp_path / "ransac_accept_harness")
    subprocess.check_call([gxx, "-std=c++17", "-O3", "-ffp-contract=off", "-Wno-psabi", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           "-I" + os.path.join(ROOT, "openpano_amd", "csrc"), os.path.join(ROOT, "tests", "harness", "ransac_accept_harness.cc"),
                           os.path.join(ROOT, "openpano_amd", "csrc", "ransac_accept_simd.cc"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"atan2_samples (\d+) max_abs_err (\S+) polygons (\d+) points (\d+) inside (\d+) on_vertex_direction (\d+) mismatches (\d+)", r.stdout)
    assert m, r.stdout
    assert int(m.group(1)) > 4_000_000 and float(m.group(2)) < 1e-10          # the margin of the wedge search (3e-7) assumes < 1e-10
    assert int(m.group(3)) > 2000 and int(m.group(4)) > 2_000_000 and int(m.group(6)) > 100_000
    assert 0.2 < int(m.group(5)) / int(m.group(4)) < 0.8                        # the points exercise both answers
    assert int(m.group(7)) == 0
    # the 8 / 4 / 2-lane keypoint count of the epilogue (count_in_polygon and every clone this CPU runs) == the sum of the reference's
    # expression over the same points, at three alignments of the blocks, special points (zero offsets, infinities, NaNs) included
    c = re.search(r"count_calls (\d+) count_mismatches (\d+)", r.stdout)
    assert c and int(c.group(1)) > 6000 and int(c.group(2)) == 0, r.stdout
    # the refit on all inliers with its Givens rotations walked by anti-diagonals (calc_transform_skewed) == calc_transform, bit for bit
    f = re.search(r"refits (\d+) refit_mismatches (\d+)", r.stdout)
    assert f and int(f.group(1)) >= 1200 and int(f.group(2)) == 0, r.stdout
