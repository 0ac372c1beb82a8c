"""MFMA / LDS / issue counters per kernel from rocprofv3 --pmc passes (counter evidence beside the
timing-derived roofline fraction).

    python tools/pmc_mfma.py <dir of pass A> <dir of pass B> [out.json]

pass A: SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
        SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
pass B: SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_MFMA GRBM_GUI_ACTIVE

Units (MI355X_MICROARCH.md; rocprofv3's own MfmaUtil expression): counter values are sums over the chip; GRBM_GUI_ACTIVE
is the sum over the 8 XCDs, so the busy cycles of one XCD's clock are GUI/8; MfmaUtil = MFMA_BUSY / (GUI/8 * 1024 SIMDs);
SQ_INSTS_VALU_MFMA_MOPS_F64 * 512 = flops executed on the matrix pipe; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_*
count quad-cycles.
"""
import csv
import glob
import json
import re
import sys

SIMDS = 1024


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(?:void )?([A-Za-z_0-9]+)(<[^(]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def load(d):
    acc = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        seen = {}
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            a = acc.setdefault(k, {"launches": 0, "ns": 0.0})
            did = r["Dispatch_Id"]
            if did not in seen:
                seen[did] = 1
                a["launches"] += 1
                if r.get("Start_Timestamp") and r.get("End_Timestamp"):
                    a["ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            a[r["Counter_Name"]] = a.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return acc


A, B = load(sys.argv[1]), load(sys.argv[2])
out = {}
for k in sorted(A, key=lambda k: -A[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0)):
    a, b = A[k], B.get(k, {})
    gui = a.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    if gui <= 0:
        continue
    e = {"launches": a["launches"], "ms_per_launch": a["ns"] * 1e-6 / max(1, a["launches"]),
         "effective_clock_GHz": gui / a["ns"] if a["ns"] else None,
         "mfma_util": a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui * SIMDS),
         "mfma_f64_flops_executed": a.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0) * 512.0,
         "wave_cycles_quad": a.get("SQ_WAVE_CYCLES", 0.0),
         "wait_any_share": a.get("SQ_WAIT_ANY", 0.0) / max(1.0, a.get("SQ_WAVE_CYCLES", 0.0)),
         "wait_inst_any_share": a.get("SQ_WAIT_INST_ANY", 0.0) / max(1.0, a.get("SQ_WAVE_CYCLES", 0.0)),
         "active_inst_any_share": a.get("SQ_ACTIVE_INST_ANY", 0.0) / max(1.0, a.get("SQ_WAVE_CYCLES", 0.0))}
    if a["ns"]:
        e["mfma_tflops_executed"] = e["mfma_f64_flops_executed"] / (a["ns"] * 1e-9) / 1e12
        # the pipe's peak at the clock the kernel actually ran at: 64 cycles per v_mfma_f64_16x16x4_f64 (2048 flops) and SIMD
        e["mfma_peak_tflops_at_effective_clock"] = SIMDS * 2048.0 / 64.0 * e["effective_clock_GHz"] * 1e9 / 1e12
    if b:
        gb = b.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        e["lds_bank_conflict_per_idx_active"] = b.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(1.0, b.get("SQ_LDS_IDX_ACTIVE", 0.0))
        e["lds_idx_active_per_cu_cycle"] = b.get("SQ_LDS_IDX_ACTIVE", 0.0) / max(1.0, gb * 256)
        e["lds_addr_conflict"] = b.get("SQ_LDS_ADDR_CONFLICT", 0.0)
        e["insts_lds"] = b.get("SQ_INSTS_LDS", 0.0)
        e["insts_mfma"] = b.get("SQ_INSTS_MFMA", 0.0)
    out[k] = e
    if e["mfma_util"] > 0.01:
        print(f"{k}\n    launches={e['launches']} {e['ms_per_launch']:.3f} ms/launch  clock={e['effective_clock_GHz']:.3f} GHz  MfmaUtil={e['mfma_util']:.3f}"
              f"  executed={e.get('mfma_tflops_executed', 0):.1f} TF of {e.get('mfma_peak_tflops_at_effective_clock', 0):.1f} at that clock"
              f"  wait_any={e['wait_any_share']:.2f} wait_inst={e['wait_inst_any_share']:.2f} active={e['active_inst_any_share']:.2f}"
              + (f"  LDS conflict/idx_active={e['lds_bank_conflict_per_idx_active']:.3f}" if b else ""))
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
