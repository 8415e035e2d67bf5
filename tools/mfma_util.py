"""Matrix-core utilisation per kernel from one rocprofv3 SQ counter pass (MI355X_MICROARCH.md, "rocprofv3 PMC slots"):

  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS \
            SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d <dir> \
            -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
  python tools/mfma_util.py <dir> profiles/r03_mfma_util.json

Per kernel symbol (folded like bench.py's names) the counters are summed over its launches.  Derived:
  mfma_util      = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs)   -- share of the kernels' wall cycles in
                   which a SIMD's matrix pipe was busy: the gfx94x `MfmaUtil` formula (ROCm 7.2 ships no gfx950 derived
                   metrics) with GRBM_GUI_ACTIVE divided by the 8 XCDs it is summed over (checked: igemm2<64,64>'s
                   SQ_VALU_MFMA_BUSY_CYCLES equals its MFMA count x 64 cycles, and GUI_ACTIVE / 8 its profiled duration x ~2 GHz);
                   GUI_ACTIVE includes the idle cycles between the kernel's waves and the counter read-out, so this reads low
                   against the HIP-event TFLOP/s of bench.py;
  mfma_busy_share_of_wave_time = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_WAVE_CYCLES): matrix-pipe-busy cycles per resident
                   wave-cycle (one wave per SIMD: the share of a wave's life in which its SIMD's matrix pipe works);
  wait_any / wait_inst_any / active = the three disjoint shares of SQ_WAVE_CYCLES (quad-cycle units): waves parked on
                   s_waitcnt / barriers, issue stalls (MFMA dependency, pipe busy), issuing;
  lds_conflict_per_wave_cycle = SQ_LDS_BANK_CONFLICT / SQ_WAVE_CYCLES."""
import csv
import glob
import json
import os
import re
import sys


def fold(name):
    n = name.replace("void ", "").replace("(anonymous namespace)::", "")
    n = re.sub(r"\(.*\)$", "", n).replace(" ", "")
    m = re.match(r"(igemm2_kernel|wgrad2_kernel)<(\d+),(\d+)", n)
    if m:
        return "%s<%s,%s>" % m.groups()
    return n


def main():
    d, out_path = sys.argv[1], sys.argv[2]
    per = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = fold(r["Kernel_Name"])
            a = per.setdefault(k, {"dispatches": set()})
            a["dispatches"].add(r.get("Dispatch_Id"))
            a[r["Counter_Name"]] = a.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    out = {"note": __doc__.split("Per kernel")[0].strip().splitlines()[0] + "; one pass over `bench.py --steps 2 --warmup 1 --no-cpu-baseline`",
           "units": "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE cycles",
           "kernels": {}}
    for k, a in sorted(per.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)):
        n = len(a.pop("dispatches"))
        wave = a.get("SQ_WAVE_CYCLES", 0.0)
        gui = a.get("GRBM_GUI_ACTIVE", 0.0)
        row = {"launches": n}
        row.update({c: round(v) for c, v in sorted(a.items())})
        if gui > 0:
            row["mfma_util"] = round(a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8.0 * 256 * 4), 4)
        if wave > 0:
            row["mfma_busy_share_of_wave_time"] = round(a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * wave), 4)
            row["wait_any_share"] = round(a.get("SQ_WAIT_ANY", 0.0) / wave, 4)
            row["wait_inst_any_share"] = round(a.get("SQ_WAIT_INST_ANY", 0.0) / wave, 4)
            row["active_share"] = round(a.get("SQ_ACTIVE_INST_ANY", 0.0) / wave, 4)
            row["lds_conflict_per_wave_cycle"] = round(a.get("SQ_LDS_BANK_CONFLICT", 0.0) / wave, 5)
        out["kernels"][k] = row
    json.dump(out, open(out_path, "w"), indent=1)
    print(len(out["kernels"]), "kernels ->", out_path)


if __name__ == "__main__":
    main()
