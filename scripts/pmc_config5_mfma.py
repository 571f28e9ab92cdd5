#!/usr/bin/env python3
"""Matrix-pipe counters of the CONFIG-5 sweeps (the per-kernel averages of scripts/pmc_summary.py mix config 4's 703-pair launches
in): per dispatch of k_match_sweep<false> in a `mfma` counter pass, MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 1024
SIMDs) and the executed flop (SQ_INSTS_VALU_MFMA_MOPS_BF16 x 512); the config-5 launches are the ones with 8128 x 32 workgroups.
    python scripts/pmc_config5_mfma.py gpurun_out/<dir with *counter_collection.csv> [latest.json]  ->  JSON on stdout
With a second argument the per-sweep averages are also written in the form bench.py replays (profiles/config5_mfma_latest.json),
tied to the hash of the library the counters were collected on."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

d = sys.argv[1]
rows = defaultdict(dict)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_match_sweep" not in r["Kernel_Name"]:
            continue
        e = rows[r["Dispatch_Id"]]
        e["kernel"] = "forward" if "<false>" in r["Kernel_Name"] else "reverse"
        e["workgroups"] = int(r["Grid_Size"]) // int(r["Workgroup_Size"])
        e["ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        e[r["Counter_Name"]] = float(r["Counter_Value"])
out = []
for k, e in sorted(rows.items(), key=lambda kv: int(kv[0])):
    if not e.get("GRBM_GUI_ACTIVE") or not e.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        continue
    per_ns = e["GRBM_GUI_ACTIVE"] / max(e["ns"], 1.0)
    xcd = 8.0 if per_ns > 6.0 else 1.0                     # summed over the 8 XCDs or not: read off the data (a shader clock, not eight)
    busy = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (e["GRBM_GUI_ACTIVE"] / xcd * 1024.0)
    flop = e.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0) * 512.0
    out.append({"dispatch": int(k), "sweep": e["kernel"], "workgroups": e["workgroups"], "ms_under_counters": e["ns"] * 1e-6, "mfma_busy": round(busy, 4),
                "executed_tflops_under_counters": round(flop / max(e["ns"], 1.0) * 1e-3, 1), "shader_clock_ghz": round(per_ns / xcd, 3)})
big = [o for o in out if o["workgroups"] > 100000]
print(json.dumps({"config5_forward_sweeps": [o for o in big if o["sweep"] == "forward"], "config5_reverse_sweeps": [o for o in big if o["sweep"] == "reverse"],
                  "config4_forward_sweeps": [o for o in out if o["sweep"] == "forward" and o["workgroups"] < 100000][:4]}, indent=1))

if len(sys.argv) > 2:
    import hashlib

    def avg(rows, key):
        return sum(r[key] for r in rows) / max(len(rows), 1)
    f5 = [o for o in big if o["sweep"] == "forward"]; r5 = [o for o in big if o["sweep"] == "reverse"]
    f4 = [o for o in out if o["sweep"] == "forward" and o["workgroups"] < 100000]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.environ.get("OPENPANO_HIP_LIB") or os.path.join(root, "openpano_amd", "libopenpano_hip.so")
    latest = {"_meta": {"lib_sha256_16": hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16],
                        "collected_by": "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE -- python bench.py "
                                        "--steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-blend --no-ingest --no-configs; scripts/pmc_config5_mfma.py",
                        "definition": "per dispatch: mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); shader clock = GRBM_GUI_ACTIVE / 8 / dispatch ns"},
              "config5_forward": {"launches": len(f5), "workgroups": int(avg(f5, "workgroups")), "mfma_busy": round(avg(f5, "mfma_busy"), 4),
                                  "shader_clock_ghz": round(avg(f5, "shader_clock_ghz"), 3), "executed_tflops": round(avg(f5, "executed_tflops_under_counters"), 1),
                                  "ms": round(avg(f5, "ms_under_counters"), 2), "peak_tflops_at_measured_clock": round(1024 * 1024 * avg(f5, "shader_clock_ghz") * 1e-3, 1)},
              "config5_reverse": {"launches": len(r5), "mfma_busy": round(avg(r5, "mfma_busy"), 4), "shader_clock_ghz": round(avg(r5, "shader_clock_ghz"), 3), "ms": round(avg(r5, "ms_under_counters"), 2)},
              "config4_forward": {"launches": len(f4), "workgroups": int(avg(f4, "workgroups")), "mfma_busy": round(avg(f4, "mfma_busy"), 4), "shader_clock_ghz": round(avg(f4, "shader_clock_ghz"), 3)},
              "dispatches": out}
    assert f5, "no config-5 forward sweep (more than 100000 workgroups) among the dispatches"
    json.dump(latest, open(sys.argv[2], "w"), indent=1)
