"""profiles/r05/counters_raw.json (tools/pmc_summary.py: per-kernel means of every counter) -> profiles/r05/counters.json (the
ratios a reader needs, per hot kernel, with the formulas), profiles/traffic.json and profiles/rocprof_avg.json (what bench.py
prints beside its live numbers).  usage: python profiles/summarize_r05.py"""
import csv
import json
import os
import re
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CXXFILT = next((c for c in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "/usr/bin/llvm-cxxfilt", "/usr/bin/c++filt") if os.path.exists(c)), "c++filt")


def short(nm):
    """Kernel name as the sources spell it.  Our kernel templates take only types (bf16 / float), ints and bools, so the Itanium
    mangling is decoded by hand (llvm-cxxfilt of ROCm 7.2 garbles `DF16b` followed by integer arguments)."""
    nm = nm.replace(".kd", "")
    m = re.match(r"^_Z(\d+)", nm)
    if not m:
        nm = nm.replace("bool _Accum, int, E", "bf16,1")          # rocprofv3's own demangler on <__bf16, 1, ...> (DF16b Li1E)
        nm = re.sub(r"^void\s+", "", nm)
        return re.sub(r"\(.*$", "", nm).replace("__bf16", "bf16").replace(" ", "")
    n = int(m.group(1))
    base, rest = nm[m.end():m.end() + n], nm[m.end() + n:]
    if not rest.startswith("I"):
        return base
    args, i = [], 1
    while i < len(rest) and rest[i] != "E":
        if rest.startswith("DF16b", i):
            args.append("bf16"); i += 5
        elif rest[i] == "f":
            args.append("float"); i += 1
        elif rest.startswith("Lb", i):
            args.append("true" if rest[i + 2] == "1" else "false"); i += 4
        elif rest.startswith("Li", i):
            j = rest.index("E", i)
            args.append(rest[i + 2:j].replace("n", "-")); i = j + 1
        else:
            return subprocess.run([CXXFILT, nm], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void ", "") or nm
    return base + "<" + ",".join(args) + ">"


raw = json.load(open(os.path.join(HERE, "r05", "counters_raw.json")))
bench = json.loads(open(os.path.join(HERE, "r05", "bench.json")).read().strip().splitlines()[-1])
rows = list(csv.DictReader(open(os.path.join(HERE, "r05", "kernel_stats.csv"))))
steps = bench["steps"] * bench.get("timed_blocks", 1) + bench["warmup"] + 4 + 2          # + keep-busy steps + the two instrumented steps
stat = {short(r["Name"]): (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6) for r in rows}
tot_ms = sum(v[2] for v in stat.values())
out = {"source": "rocprofv3 --pmc (tools/pmc_kernel.sh: three SQ passes, FETCH_SIZE, WRITE_SIZE; kernel trace only) over `bench.py --steps 2 --warmup 1`, "
                 "means per launch; durations from the --kernel-trace --stats run (kernel_stats.csv)",
       "formulas": {"mfma_busy": "SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CYCLES)  [busy cycles are per SIMD-cycle, SQ_BUSY_CYCLES per-SE quad... reported as measured ratio, compare kernels with each other]",
                    "wait_any": "SQ_WAIT_ANY / SQ_WAVE_CYCLES (wave parked on s_waitcnt / barrier)",
                    "wait_inst_any": "SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES (issue stall)",
                    "active": "SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES",
                    "lds_bank_conflict": "SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE",
                    "mfma_cycles_per_wave_cycle": "SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_WAVE_CYCLES)  (SQ_WAVE_CYCLES counts quad-cycles)",
                    "hbm_read_bytes": "FETCH_SIZE [KiB] * 1024 * 2 (gfx950 halves wide streaming reads: MI355X_MICROARCH.md)",
                    "hbm_write_bytes": "WRITE_SIZE [KiB] * 1024"},
       "kernels": {}}
traffic = {}
for k, m in raw.items():
    name = short(k) if k.startswith("_Z") else k.replace(" ", "").replace("__bf16", "bf16")
    if name not in stat or "SQ_WAVE_CYCLES" not in m:
        continue
    calls, avg_us, total_ms = stat[name]
    e = {"launches_per_step": round(calls / steps, 1), "avg_us": round(avg_us, 2), "ms_per_step": round(total_ms / steps, 3),
         "share_of_gpu_time": round(total_ms / tot_ms, 4)}
    wc = m["SQ_WAVE_CYCLES"]
    for key, c in (("wait_any", "SQ_WAIT_ANY"), ("wait_inst_any", "SQ_WAIT_INST_ANY"), ("active", "SQ_ACTIVE_INST_ANY")):
        if c in m:
            e[key] = round(m[c] / wc, 3)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        e["mfma_cycles_per_wave_cycle"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * wc), 4)
        if m.get("SQ_BUSY_CYCLES"):
            e["mfma_busy_over_sq_busy"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / m["SQ_BUSY_CYCLES"], 3)
    if m.get("SQ_LDS_IDX_ACTIVE"):
        e["lds_bank_conflict"] = round(m.get("SQ_LDS_BANK_CONFLICT", 0.0) / m["SQ_LDS_IDX_ACTIVE"], 3)
    if "hbm_read_bytes" in m:
        e["hbm_read_MB"] = round(m["hbm_read_bytes"] / 1e6, 2)
    if "hbm_write_bytes" in m:
        e["hbm_write_MB"] = round(m["hbm_write_bytes"] / 1e6, 2)
    out["kernels"][name] = e
    if "hbm_read_bytes" in m or "hbm_write_bytes" in m:
        rd, wr = m.get("hbm_read_bytes", 0.0), m.get("hbm_write_bytes", 0.0)
        traffic[name] = {"launches_seen": int(m["dispatches"]), "read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr),
                         "bytes_per_launch": round(rd + wr)}
out["kernels"] = dict(sorted(out["kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"]))
out["hbm_bytes_per_step"] = round(sum(v["bytes_per_launch"] * out["kernels"][k]["launches_per_step"] for k, v in traffic.items() if k in out["kernels"]))
json.dump(out, open(os.path.join(HERE, "r05", "counters.json"), "w"), indent=1)
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE x2 (gfx950), KiB -> bytes; profiles/r05",
           "workload": "cfg3", "dtype": "bf16", "kernels": traffic}, open(os.path.join(HERE, "traffic.json"), "w"), indent=1, sort_keys=True)
names = bench["roofline"]["all_mfma_kernels"]


def rocprof_prefixes(name):            # bench.py::rocprof_prefixes (kept in step by hand: this script must not import bench.py's torch)
    if name.startswith("conv_strip<"):
        c = name[len("conv_strip<"):-1]
        return ("conv_strip" + c + "<", "conv_strip" + c + "s<")
    if name.startswith("conv_strip_g<"):
        c = name[len("conv_strip_g<"):-1]
        return ("conv_strip" + c + "_g<", "conv_strip" + c + "s_g<")
    if name == "wgrad_taps_kernel<32>":
        return (name, "wgrad_rows32<")
    if name == "wgrad_taps_kernel_g<32>":
        return (name, "wgrad_rows32_g<")
    if name in ("wgrad_rows64", "wgrad_rows64_g"):
        return (name + "<",)
    return (name[:-1] + ",",)


def fam_avg(name):
    fam = [(c, t) for nm, (c, _, t) in stat.items() if nm == name or nm.startswith(rocprof_prefixes(name)) or (name in ("conv_pw", "wgrad_pw") and nm.startswith(name + "<"))]
    return round(1e3 * sum(t for _, t in fam) / sum(c for c, _ in fam), 2) if fam else None


json.dump({"source": "profiles/r05/kernel_stats.csv (rocprofv3 --kernel-trace --stats)", "workload": "cfg3", "dtype": "bf16",
           "avg_us": {k: fam_avg(k) for k in names}}, open(os.path.join(HERE, "rocprof_avg.json"), "w"), indent=1)
print("GPU ms per step (sum of kernel durations):", round(tot_ms / steps, 3), " HBM GB per step:", round(out["hbm_bytes_per_step"] / 1e9, 2))
for k, e in list(out["kernels"].items())[:25]:
    print(f"{k[:46]:46s} {e['launches_per_step']:6.1f}/step {e['avg_us']:7.1f} us {e['ms_per_step']:6.3f} ms  wait {e.get('wait_any', 0):.2f} stall {e.get('wait_inst_any', 0):.2f} "
          f"mfma/wave {e.get('mfma_cycles_per_wave_cycle', 0):.3f} ldsconf {e.get('lds_bank_conflict', 0):.2f} rd {e.get('hbm_read_MB', 0):7.1f} wr {e.get('hbm_write_MB', 0):7.1f} MB")
