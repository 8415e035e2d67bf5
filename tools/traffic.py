"""HBM-side traffic per kernel from two rocprofv3 counter passes (MI355X_MICROARCH.md, HBM section):

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
  python tools/traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r02_traffic.json

traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes: on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads
(the guide's correction), WRITE_SIZE is exact for 16-byte streaming stores.  Kernel names are folded the way bench.py
names them (igemm2_kernel<BM,BN>, wgrad2_kernel<BM,BN>)."""
import csv
import glob
import json
import os
import re
import sys


def fold(name):
    n = name.replace("void ", "").replace("(anonymous namespace)::", "")
    n = re.sub(r"\(.*\)$", "", n).replace(" ", "")
    m = re.match(r"(igemm2_kernel|wgrad2_kernel|igemm_kernel|wgrad_kernel)<(\d+),(\d+)", n)
    if m:
        return "%s<%s,%s>" % m.groups()
    return n


def load(d, counter):
    per = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            k = fold(r["Kernel_Name"])
            a = per.setdefault(k, [0, 0.0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return per


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    out = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 2 --warmup 1 "
                   "--no-cpu-baseline`; traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024: FETCH_SIZE reads 1/2 of wide coalesced "
                   "reads on gfx950 (MI355X_MICROARCH.md, HBM section). Counts L2 fabric requests, i.e. includes Infinity-Cache hits.",
           "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        nf, vf = fetch.get(k, [0, 0.0])
        nw, vw = write.get(k, [0, 0.0])
        n = max(nf, nw, 1)
        out["kernels"][k] = {"launches": n, "traffic_bytes_per_launch": round((2.0 * vf / max(nf, 1) + vw / max(nw, 1)) * 1024),
                             "fetch_size_kb_raw_per_launch": round(vf / max(nf, 1), 1), "write_size_kb_per_launch": round(vw / max(nw, 1), 1)}
    json.dump(out, open(sys.argv[3], "w"), indent=0)
    print(len(out["kernels"]), "kernels ->", sys.argv[3])


if __name__ == "__main__":
    main()
