"""CPU: the host thread pool behind host_parallel_for (openpano_amd/csrc/host_pool.hpp: futex sleep, tree wake-up, items claimed by
compare-exchange on (loop, index), completion counted in items) under stress -- tests/harness/host_pool_harness.cc: thousands of
loops of random length and cost from two caller threads, every item exactly once and never outside its loop; with the pool at its
default size, oversubscribed (64 threads on whatever this machine has), with no workers at all, and under ThreadSanitizer."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "harness", "host_pool_harness.cc")


def _build(tmp_path, name, *flags):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("g++ missing")
    exe = str(tmp_path / name)
    subprocess.check_call([gxx, "-std=c++17", "-pthread", *flags, "-I" + os.path.join(ROOT, "openpano_amd", "csrc"), SRC, "-o", exe])
    return exe


def _run(exe, loops, threads=None):
    env = dict(os.environ)
    env.pop("OPENPANO_HOST_THREADS", None)
    if threads is not None:
        env["OPENPANO_HOST_THREADS"] = str(threads)
    r = subprocess.run([exe, str(loops)], capture_output=True, text=True, timeout=900, env=env)
    m = re.search(r"workers (\d+) loops (\d+) items (\d+) errors (\d+)", r.stdout)
    assert r.returncode == 0 and m, r.stdout + r.stderr
    assert int(m.group(4)) == 0 and int(m.group(2)) == 2 * loops
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-3000:]
    return int(m.group(1))


def test_host_pool_every_item_once(tmp_path):
    exe = _build(tmp_path, "hp", "-O2")
    _run(exe, 6000)
    assert _run(exe, 4000, threads=64) == 63
    assert _run(exe, 1000, threads=1) == 0


def test_host_pool_under_thread_sanitizer(tmp_path):
    exe = _build(tmp_path, "hp_tsan", "-O1", "-g", "-fsanitize=thread")
    probe = subprocess.run([exe, "1"], capture_output=True, text=True, timeout=300)
    if probe.returncode != 0 and "ThreadSanitizer" not in probe.stderr and "errors" not in probe.stdout:
        pytest.skip("ThreadSanitizer runtime does not start here: " + probe.stderr[-300:])
    _run(exe, 1000)
    _run(exe, 500, threads=33)
