"""The ONE JSON line bench.py prints, derived from the full record.

bench.py measures a lot (every BASELINE config, four CPU legs, stage rooflines, the whole pipeline); the driver parses
one line and gives up on a long one (round 5's 24.8 KB line was recorded with `parsed: null`).  So the full record goes
to a side file (`bench_detail.json`, next to bench.py and under gpurun_out/) and the line carries scalars only, under
a hard cap of MAX_LINE_BYTES -- asserted here, and by tests/test_bench_contract.py on a real N = 1 record and on a
fabricated N = 8 one.  No prose in the line: the definitions live in bench.py's docstring and DESIGN.md section 5.
"""
import json

MAX_LINE_BYTES = 6144          # the review asked for < 8 KB; r03's 10.5 KB line parsed, r05's 24.8 KB did not
DETAIL_FILE = "bench_detail.json"


def _r(x, sig=6):
    """floats to `sig` significant digits (17-digit reprs are a third of a naive line); containers recursively"""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float(f"{x:.{sig}g}")
    if isinstance(x, dict):
        return {k: _r(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, sig) for v in x]
    return x


def _pick(d, keys):
    d = d or {}
    return {k: d.get(k) for k in keys if k in d}


def _short(s, n):
    s = s or ""
    return s if len(s) <= n else s[: n - 1] + "~"


def _job(j):
    """a strong-scaled job section (config5 / strong_config4) reduced to its phase times and rates"""
    if not j:
        return None
    mr = j.get("match_roofline") or {}
    o = _pick(j, ("images", "n_gpus", "image_pairs", "descriptors", "phase_ms", "job_wall_ms", "keypoints_per_s", "image_pairs_per_s",
                  "matches_per_s", "ransac_image_pairs_per_s", "allgather_bytes_per_rank"))
    if mr:
        o["match_roofline"] = {"bound": "mfma", "frac": mr.get("frac"), "achieved": mr.get("achieved"), "peak": mr.get("peak"), "unit": mr.get("unit"),
                               "algorithmic_tflops": mr.get("algorithmic_tflops"),
                               "mfma_busy": mr.get("mfma_busy_config5_forward", mr.get("mfma_busy")),
                               "shader_clock_ghz": mr.get("shader_clock_ghz_under_the_sweep")}
    if j.get("per_rank_phase_ms"):                       # N > 1: the slowest rank per phase instead of N blocks
        pr = j["per_rank_phase_ms"]
        rows = pr.values() if isinstance(pr, dict) else pr
        agg = {}
        for row in rows:
            for k, v in (row or {}).items():
                if isinstance(v, (int, float)):
                    agg[k] = max(agg.get(k, 0.0), float(v))
        o["max_rank_phase_ms"] = agg
    if j.get("predicted"):
        o["predicted"] = _pick(j["predicted"], ("job_ms", "keypoints_per_s", "image_pairs_per_s"))
    if j.get("parity"):
        o["parity"] = _pick(j["parity"], ("ok", "pairs", "images"))
    return o


def compact(out):
    """full record -> the line (a dict).  Every contract key of the task's bench section, `roofline` and `cpu_baseline`
    in the shape the judge reads, then one scalar block per secondary measurement."""
    line = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                    "scaling", "vs_baseline", "dtype", "data")}
    cfg = out.get("config") or {}
    line["config"] = {"workload": _short(cfg.get("workload"), 120), **_pick(cfg, ("images_per_gpu", "images_in_job", "image", "keypoints_per_image"))}
    rc = out.get("rccl_ranks")
    line["rccl_ranks"] = _pick(rc, ("world_size", "allreduce_count", "ok", "backend")) if rc else None
    rf = out.get("roofline") or {}
    line["roofline"] = _pick(rf, ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic",
                                  "algorithmic_bytes_per_launch", "avg_launch_ms")) if rf else None
    line["stage_ms"] = out.get("stage_ms")
    pr = out.get("sift_path_roofline")
    if pr:
        line["sift_path_frac_of_hbm_peak"] = pr.get("frac")
    cb = out.get("cpu_baseline")
    if cb:
        c = _pick(cb, ("value", "unit", "cores", "host_cpus", "cpu_quota", "cpu_model", "kind", "single_thread_value"))
        c["flags"] = _short(cb.get("flags"), 60)
        c["sample"] = _short(cb.get("sample"), 110)
        line["cpu_baseline"] = c
    else:
        line["cpu_baseline"] = None
    for k in ("gpu_over_cpu", "value_natural", "gpu_over_cpu_natural", "value_protocol_f32", "value_protocol_u8",
              "gpu_over_cpu_protocol_f32", "gpu_over_cpu_protocol_u8"):
        if k in out:
            line[k] = out[k]
    if out.get("parity"):
        line["parity"] = _pick(out["parity"], ("ok", "images", "pairs", "descriptors"))
    line["parity_checked"] = out.get("parity_checked")
    m = out.get("match")
    if m:
        line["match"] = {**_pick(m, ("image_pairs_per_s", "matches_per_s", "image_pairs", "ms_per_step", "descriptor_allgather_ms")),
                         "cpu_exact_image_pairs_per_s": (m.get("cpu_baseline") or {}).get("exact_image_pairs_per_s"),
                         "cpu_flann_image_pairs_per_s": (m.get("cpu_baseline") or {}).get("flann_image_pairs_per_s"),
                         "mfma_frac": (m.get("roofline") or {}).get("frac")}
    r = out.get("ransac")
    if r:
        line["ransac"] = {**_pick(r, ("image_pairs_per_s", "ms_per_step", "pairs", "gpu_over_cpu")),
                          "cpu_image_pairs_per_s": (r.get("cpu_baseline") or {}).get("value")}
    b = out.get("blend")
    if b:
        line["blend"] = {k: {"mpix_per_s": v.get("output_mpix_per_s"), "ms": v.get("ms_per_blend"), "hbm_frac": (v.get("roofline") or {}).get("frac"),
                             "cpu_mpix_per_s": (v.get("cpu_baseline") or {}).get("value")}
                         for k, v in b.items() if isinstance(v, dict)}
    e = out.get("stitch_e2e")
    if e:
        line["stitch_e2e"] = _pick(e, ("images", "ms_total", "stage_ms"))
    cs = out.get("configs")
    if cs:
        line["configs"] = {k: {**_pick(v, ("keypoints_per_s", "sift_ms_per_step", "image_pairs_per_s")), "parity_ok": (v.get("parity") or {}).get("ok")}
                           for k, v in cs.items() if k in ("2", "3", "4_natural")}
    for key in ("config5", "strong_config4"):
        if out.get(key):
            line[key] = _job(out[key])
    if out.get("predicted"):
        line["predicted"] = _pick(out["predicted"], ("sift_ms_per_step", "value"))
    if out.get("failed_sections"):                       # a secondary section raised: named here, the line itself stands
        line["failed_sections"] = {k: _short(v, 80) for k, v in list(out["failed_sections"].items())[:6]}
    line["detail"] = DETAIL_FILE
    return _r(line)


def render(out):
    """-> the line as text, guaranteed under the cap (optional blocks are dropped, least important first, if a future
    field ever pushes it over; the contract keys, roofline and cpu_baseline are never dropped)"""
    line = compact(out)
    s = json.dumps(line, separators=(",", ":"))
    for k in ("stitch_e2e", "configs", "blend", "strong_config4", "stage_ms", "ransac", "match", "config5"):
        if len(s) <= MAX_LINE_BYTES:
            break
        line.pop(k, None)
        s = json.dumps(line, separators=(",", ":"))
    assert len(s) <= MAX_LINE_BYTES, len(s)
    return s
