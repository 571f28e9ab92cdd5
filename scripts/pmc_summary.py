#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (scripts/gpu_pmc.sh) per kernel.

HBM traffic per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes: FETCH_SIZE/WRITE_SIZE are in
KiB, and on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming
read (MI355X_MICROARCH.md "HBM"; cdna_hip_programming.md section 7) -- the correction is applied
here and the raw counters are kept next to it.  Writes gpurun_out/<tag>_pmc.json (copy it to
profiles/ and to profiles/pmc_latest.json, which bench.py reads for roofline.traffic)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")

# kernel-name fragment -> bench.py stage label
LABELS = {"k_grey_octaves": "resize + octave grey", "k_pyramid": "build pyramid", "k_pyramid_rows": "build pyramid",
          "k_extrema_scan": "extrema scan", "k_refine": "extrema refine", "k_sort_refined": "extrema refine",
          "k_orientation": "orientation", "k_descriptor": "sift descriptor",
          "k_match_sweep": "matcher mfma sweep", "k_match_slow": "matcher exact scan", "k_split_bf16": "matcher split", "k_orient_peaks": "orientation peaks",
          "k_ransac_hyp": "ransac", "k_blend_linear": "blend linear"}


def short(name):
    m = re.search(r"(k_[a-z0-9_]+)", name)
    return m.group(1) if m else name[:40]


acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))   # kernel -> counter -> [sum, n]
for d in sorted(glob.glob(os.path.join(root, f"{tag}_pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = short(row.get("Kernel_Name", ""))
                c = row.get("Counter_Name"); v = row.get("Counter_Value")
                if c is None or v is None:
                    continue
                a = acc[k][c]; a[0] += float(v); a[1] += 1
                if c == "GRBM_GUI_ACTIVE" and row.get("End_Timestamp"):      # the pass's own dispatch durations: what GRBM_GUI_ACTIVE is a count of
                    a = acc[k]["_gui_pass_ns"]; a[0] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"]); a[1] += 1
out = {}
for k, cs in acc.items():
    e = {c: s / max(n, 1) for c, (s, n) in cs.items()}
    e["launches_seen"] = max(n for _, n in cs.values())
    if "FETCH_SIZE" in e or "WRITE_SIZE" in e:
        e["hbm_bytes_per_launch"] = (2.0 * e.get("FETCH_SIZE", 0.0) + e.get("WRITE_SIZE", 0.0)) * 1024.0
        e["hbm_bytes_per_launch_uncorrected"] = (e.get("FETCH_SIZE", 0.0) + e.get("WRITE_SIZE", 0.0)) * 1024.0
    if e.get("SQ_VALU_MFMA_BUSY_CYCLES") and e.get("GRBM_GUI_ACTIVE"):
        # MfmaUtil (counter_defs.yaml): matrix-pipe busy cycles summed over the SIMDs / (kernel cycles x 1024 SIMDs).  Whether
        # rocprofv3 hands GRBM_GUI_ACTIVE per device or summed over its 8 XCDs is read off the data: cycles per nanosecond of the
        # dispatch must be a shader clock (1.5 - 2.5 GHz), not eight of them
        per_ns = e["GRBM_GUI_ACTIVE"] / max(e.get("_gui_pass_ns", 0.0), 1.0)
        xcd_sum = 8.0 if per_ns > 6.0 else 1.0
        e["gui_active_cycles_per_ns"] = per_ns / xcd_sum
        e["mfma_busy"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (e["GRBM_GUI_ACTIVE"] / xcd_sum * 1024.0)
        e["mfma_flop_executed"] = e.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0) * 512.0
    if e.get("SQ_BUSY_CYCLES") and e.get("SQ_ACTIVE_INST_VALU"):
        e["valu_active_over_wave_cycles"] = e["SQ_ACTIVE_INST_VALU"] / max(e.get("SQ_WAVE_CYCLES", 1.0), 1.0)
    out[k] = e
    lab = LABELS.get(k)
    if lab and lab not in out:
        out[lab] = e
out["_meta"] = {"tag": tag, "collected_by": "scripts/gpu_pmc.sh (rocprofv3 --kernel-trace --pmc, one counter set per pass)",
                "lib_sha256_16": None}
try:
    import hashlib
    lib = os.path.join(os.path.dirname(root), "openpano_amd", "libopenpano_hip.so")
    out["_meta"]["lib_sha256_16"] = hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16]
except OSError:
    pass
path = os.path.join(root, f"{tag}_pmc.json")
json.dump(out, open(path, "w"), indent=1, sort_keys=True)
for k in sorted(out):
    if k.startswith("k_") and isinstance(out[k], dict):
        e = out[k]
        print(f"{k:22s} hbm/launch {e.get('hbm_bytes_per_launch', float('nan')) / 1e6:10.2f} MB  "
              f"VALU insts {e.get('SQ_INSTS_VALU', float('nan')):.3g}  LDS insts {e.get('SQ_INSTS_LDS', float('nan')):.3g}  "
              f"waves {e.get('SQ_WAVES', float('nan')):.3g}")
print("wrote", path)
