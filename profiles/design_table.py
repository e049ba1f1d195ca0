"""Markdown rows of DESIGN.md's per-kernel table from profiles/<round>/counters.json.  usage: python profiles/design_table.py [min_ms] [round, default r03]"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
c = json.load(open(os.path.join(HERE, sys.argv[2] if len(sys.argv) > 2 else "r03", "counters.json")))
min_ms = float(sys.argv[1]) if len(sys.argv) > 1 else 0.13
print("| kernel | launches/step | avg µs | ms/step | wait | stall | mfma | LDS conflicts | HBM MB rd / wr |")
print("|---|---:|---:|---:|---:|---:|---:|---:|---|")
for k, e in c["kernels"].items():
    if e["ms_per_step"] < min_ms:
        continue
    mf = e.get("mfma_cycles_per_wave_cycle", 0)
    print(f"| `{k}` | {e['launches_per_step']:g} | {e['avg_us']:.1f} | {e['ms_per_step']:.2f} | {e.get('wait_any', 0):.2f} | {e.get('wait_inst_any', 0):.2f} | "
          f"{(f'{mf:.2f}' if mf >= 0.005 else '–')} | {e.get('lds_bank_conflict', 0):.2f} | {e.get('hbm_read_MB', 0):.1f} / {e.get('hbm_write_MB', 0):.1f} |")
print(f"\nHBM-side bytes per step: {c['hbm_bytes_per_step'] / 1e9:.2f} GB")
