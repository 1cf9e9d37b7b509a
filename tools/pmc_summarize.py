#!/usr/bin/env python3
"""Summarise the rocprofv3 output of tools/pmc_passes.sh into one small JSON + markdown table.

    python tools/pmc_summarize.py <dir written by pmc_passes.sh> <kernel-name-substring> [--copy-to profiles/...]

Per-launch means of every counter for the kernel whose name contains the substring (launches of
other kernels -- copies, the peak probe -- are ignored), the duration from the counter rows' own
timestamps and from the --stats pass, plus the code-object facts rocprofv3 records per dispatch
(VGPRs, scratch, LDS).  FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (128-byte
requests tallied as 64 B) -- both raw and corrected values are written.
"""
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys


def static_resources(tu, needle):
    """VGPRs / AGPRs / scratch / occupancy / LDS of the kernel from the compiler's own remarks on the translation unit
    (hipcc -Rpass-analysis=kernel-resource-usage): the `VGPR_Count` rocprofv3 records per dispatch is in allocation
    units on gfx950 (84 for a kernel that allocates 168 registers), which misled a reader of the round-2 summaries."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "forge_ec_amd", "csrc", tu)
    try:
        out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o",
                              "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, timeout=900).stderr
    except Exception as e:  # no hipcc here: say so instead of guessing
        return {"error": "hipcc remarks unavailable: %s" % e}
    blocks = re.split(r"remark: [^\n]*Function Name: ", out)[1:]
    res = []
    for b in blocks:
        name = b.split()[0]
        if needle.split("<")[0] not in name:   # (mangled names: the template arguments of the needle are not comparable)
            continue
        def field(label):
            m = re.search(re.escape(label) + r": (\d+)", b)
            return int(m.group(1)) if m else None
        res.append({"mangled": name, "vgprs": field("VGPRs"), "agprs": field("AGPRs"), "sgprs": field("SGPRs"),
                    "scratch_bytes_per_lane": field("ScratchSize [bytes/lane]"), "occupancy_waves_per_simd": field("Occupancy [waves/SIMD]"),
                    "lds_bytes_per_block": field("LDS Size [bytes/block]")})
    return res


def main():
    src, needle = sys.argv[1], sys.argv[2]
    workload = sys.argv[sys.argv.index("--workload") + 1] if "--workload" in sys.argv else None
    units = int(sys.argv[sys.argv.index("--units") + 1]) if "--units" in sys.argv else 1 << 20
    copy_to = sys.argv[sys.argv.index("--copy-to") + 1] if "--copy-to" in sys.argv else None
    counters, meta, durs = {}, {}, []
    for path in sorted(glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)):
        per_dispatch = {}
        for row in csv.DictReader(open(path)):
            if needle not in row["Kernel_Name"]:
                continue
            key = (row["Dispatch_Id"], row["Counter_Name"])
            per_dispatch[key] = per_dispatch.get(key, 0.0) + float(row["Counter_Value"])
            meta = {"kernel": row["Kernel_Name"][:120], "grid": int(row["Grid_Size"]),
                    "workgroup": int(row["Workgroup_Size"]), "lds_bytes": int(row["LDS_Block_Size"]),
                    "scratch_bytes_per_lane": int(row["Scratch_Size"]),
                    "vgpr_count_field_of_rocprofv3": int(row["VGPR_Count"]),   # allocation units, NOT registers: see code_object_static
                    "agpr": int(row["Accum_VGPR_Count"]), "sgpr": int(row["SGPR_Count"])}
            durs.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6)
        by_name = {}
        for (_, name), v in per_dispatch.items():
            by_name.setdefault(name, []).append(v)
        for name, vals in by_name.items():
            counters[name] = {"per_launch": sum(vals) / len(vals), "launches": len(vals)}
    stats = {}
    for path in glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if needle in row["Name"]:
                stats = {"calls": int(row["Calls"]), "avg_ms": float(row["AverageNs"]) * 1e-6,
                         "min_ms": float(row["MinNs"]) * 1e-6, "max_ms": float(row["MaxNs"]) * 1e-6}
        if copy_to:
            os.makedirs(copy_to, exist_ok=True)
            shutil.copy(path, os.path.join(copy_to, "kernel_stats.csv"))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from forge_ec_amd import build as fbuild
    # the library sources must be the ones that were profiled: run this right after the passes
    out = {"workload": workload, "units_per_launch": units, "source_hash": fbuild.source_hash(),
           "kernel_tu": fbuild.WORKLOAD_TU.get(workload), "kernel_source_hash": fbuild.tu_closure_hash(fbuild.WORKLOAD_TU[workload], kernel_code_only=True) if workload in fbuild.WORKLOAD_TU else None,
           "source_dir": src, "code_object": meta,
           "code_object_static": static_resources(fbuild.WORKLOAD_TU[workload], needle) if workload in fbuild.WORKLOAD_TU else None,
           "counters": counters, "kernel_stats": stats,
           "duration_under_pmc_ms": (sum(durs) / len(durs)) if durs else None}
    c = {k: v["per_launch"] for k, v in counters.items()}
    derived = {}
    if "FETCH_SIZE" in c:
        derived["fetch_bytes_corrected"] = c["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in c:
        derived["write_bytes"] = c["WRITE_SIZE"] * 1024
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        derived["hbm_bytes_per_launch"] = derived["fetch_bytes_corrected"] + derived["write_bytes"]
    if "SQ_WAVE_CYCLES" in c:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
            if k in c:
                derived[k + "_frac_of_wave_cycles"] = c[k] / c["SQ_WAVE_CYCLES"]
    if "SQ_INSTS_VALU" in c and "GRBM_GUI_ACTIVE" in c:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs, SQ_INSTS_VALU over the 1024 SIMDs (256 CUs x 4): cycles one SIMD spends
        # per VALU wave-instruction.  This is the "VALU-issue bound" statement; VALUBusy (a derived counter that reads
        # 102-104 % on these kernels) is kept in `counters` but is not a utilisation.
        derived["valu_issue_cycles_per_inst_per_simd"] = (c["GRBM_GUI_ACTIVE"] / 8.0) / (c["SQ_INSTS_VALU"] / 1024.0)
        derived["valu_insts_per_64_units"] = c["SQ_INSTS_VALU"] / (units / 64.0)
    if "SQ_INSTS_VALU" in c and meta:
        waves = meta["grid"] / 64.0
        derived["valu_insts_per_wave"] = c["SQ_INSTS_VALU"] / waves
        derived["valu_insts_per_wave_per_ladder_step"] = c["SQ_INSTS_VALU"] / waves / 256.0
        for k in ("SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM"):
            if k in c:
                derived[k.lower() + "_per_wave_per_ladder_step"] = c[k] / waves / 256.0
    out["derived"] = derived
    text = json.dumps(out, indent=1)
    print(text)
    if copy_to:
        os.makedirs(copy_to, exist_ok=True)
        with open(os.path.join(copy_to, "pmc.json"), "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
