"""Per-kernel HBM traffic from the two PMC passes of profiles/collect_traffic.sh.

Units and corrections as MI355X_MICROARCH.md prescribes for gfx950: rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB;
FETCH_SIZE tallies the 128-B requests of wide (16 B/lane) coalesced reads at 64 B, so it is DOUBLED; WRITE_SIZE is exact
for 16-B-per-lane stores and float atomics.  Reads served by the 256 MiB Infinity Cache are counted too (the counters sit
on the L2's fabric side), so "traffic" is an upper bound of DRAM bytes.

usage: python profiles/traffic_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/traffic.json [workload] [dtype]
"""
import collections
import csv
import glob
import json
import os
import re
import subprocess
import sys


_DEMANGLED = {}
_LLVM = "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"
CXXFILT = _LLVM if os.path.exists(_LLVM) else "c++filt"      # binutils' c++filt does not know the bf16 mangling (DF16b)


def demangle(name: str) -> str:
    if not name.startswith("_Z"):
        return name
    if name not in _DEMANGLED:
        try:
            _DEMANGLED[name] = subprocess.run([CXXFILT, name.replace(".kd", "").replace("DF16b", "u6__bf16")], capture_output=True, text=True).stdout.strip() or name
        except OSError:
            _DEMANGLED[name] = name
    return _DEMANGLED[name]


def short(name: str) -> str:
    """'void conv_igemm<__bf16, 256, 32>(ConvK)' -> 'conv_igemm<bf16,256,32>' (the names bench.py prints)."""
    n = re.sub(r"^void\s+", "", demangle(name))
    n = re.sub(r"\(.*$", "", n).replace("__bf16", "bf16").replace("__hip_bfloat16", "bf16").replace(" ", "")
    return n.replace(".kd", "")


def per_kernel(d: str, counter: str):
    fs = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
    if not fs:
        sys.exit(f"no counter_collection.csv under {d}")
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] != counter:
            continue
        a = agg[short(r["Kernel_Name"])]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return agg


def main():
    fd, wd, out = sys.argv[1:4]
    workload = sys.argv[4] if len(sys.argv) > 4 else "cfg3"
    dtype = sys.argv[5] if len(sys.argv) > 5 else "bf16"
    f, w = per_kernel(fd, "FETCH_SIZE"), per_kernel(wd, "WRITE_SIZE")
    res = {}
    for k in sorted(set(f) | set(w)):
        nf, sf = f.get(k, [0, 0.0])
        nw, sw = w.get(k, [0, 0.0])
        rd = 2.0 * 1024.0 * sf / max(nf, 1)          # KiB -> bytes, x2 gfx950 correction
        wr = 1024.0 * sw / max(nw, 1)
        res[k] = {"launches_seen": max(nf, nw), "read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr),
                  "bytes_per_launch": round(rd + wr)}
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE x2 (gfx950), KiB -> bytes",
               "workload": workload, "dtype": dtype, "kernels": res}, open(out, "w"), indent=1, sort_keys=True)
    top = sorted(res.items(), key=lambda kv: -kv[1]["bytes_per_launch"] * kv[1]["launches_seen"])[:12]
    for k, v in top:
        print(f"{k:60s} n={v['launches_seen']:5d} rd={v['read_bytes_per_launch']/1e6:9.2f} MB wr={v['write_bytes_per_launch']/1e6:9.2f} MB")


if __name__ == "__main__":
    main()
