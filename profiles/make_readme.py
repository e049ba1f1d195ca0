"""Regenerates the tables of profiles/README.md from r01_final/{kernel_stats.csv,bench.json} and traffic.json.
usage: python profiles/make_readme.py   (prose sections are kept in this file)"""
import csv, json, re, subprocess

rows = list(csv.DictReader(open('profiles/r01_final/kernel_stats.csv')))
b = json.load(open('profiles/r01_final/bench.json'))
bu = json.load(open('profiles/r01_final/bench_unprofiled.json'))
tr = json.load(open('profiles/traffic.json'))['kernels']
n = b['steps'] + b['warmup'] + 4 + 1
tot = sum(float(r['TotalDurationNs']) for r in rows)


def short(nm):
    if nm.startswith('_Z'):
        nm = subprocess.run(['c++filt', nm.replace('DF16b', 'u6__bf16')], capture_output=True, text=True).stdout.strip() or nm
    nm = re.sub(r'^void\s+', '', nm)
    return re.sub(r'\(.*$', '', nm).replace('__bf16', 'bf16').replace(' ', '')


lines, rp = [], {}
for r in rows[:22]:
    nm = short(r['Name'])
    rp[nm] = float(r['AverageNs']) / 1e3
    t = tr.get(nm)
    tb = "%.1f / %.1f" % (t['read_bytes_per_launch'] / 1e6, t['write_bytes_per_launch'] / 1e6) if t else ""
    lines.append("| `%s` | %.1f | %.3f | %.1f | %.1f | %s |" % (nm[:48], int(r['Calls']) / n, float(r['TotalDurationNs']) / 1e6 / n,
                                                            float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot, tb))
table = "\n".join(lines)
rf = b['roofline']
# rocprofv3's average per bench kernel name (a bench name folds template instantiations: conv_dmap<128,128> = conv_dmap<128,128,64>)
allrows = [(short(r['Name']), int(r['Calls']), float(r['TotalDurationNs'])) for r in rows]


def rocprof_avg(name):
    fam = [(c, t) for nm, c, t in allrows if nm == name or nm.startswith(name[:-1] + ",")]
    return sum(t for _, t in fam) / sum(c for c, _ in fam) / 1e3 if fam else float('nan')


avgs = {k: round(rocprof_avg(k), 2) for k in rf['all_mfma_kernels']}
json.dump({"source": "profiles/r01_final/kernel_stats.csv (rocprofv3 --kernel-trace --stats)", "workload": b['config']['workload'].split(':')[0],
           "dtype": b['dtype'], "avg_us": avgs}, open('profiles/rocprof_avg.json', 'w'), indent=1)
dom_rp = avgs[rf['kernel']]
fam = "\n".join("| `%s` | %d | %.1f | %.1f | %.0f |" % (k, v['launches'], 1e3 * v['ms_per_step'] / v['launches'], avgs[k], v['tflops'])
                for k, v in rf['all_mfma_kernels'].items())
txt = f'''# profiles/

rocprofv3 summaries of `python bench.py` on one MI355X (cfg3: 256×256×6, multitask Tanimoto-dual, bs 8, bf16).  One
sub-directory per stage of round 1, each with the `rocprofv3 --kernel-trace --stats` kernel table (`kernel_stats.csv`) and
the bench JSON line printed by the same command (`bench.json`); `traffic.json` holds the PMC HBM bytes per kernel launch.

Commands (on the GPU box):
* `cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 10 --warmup 3`
* `bash profiles/collect_traffic.sh && python3 profiles/traffic_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/traffic.json`
* `python profiles/make_readme.py` (this file's tables)

| dir | state | ms/step | patches/s |
|---|---|---:|---:|
| `r01_first/` | first correct path (contended fp64 statistics atomics, no split-K, eager launches) | 37.1 | 216 |
| `r01_final/` | end of round 1 (`bench_unprofiled.json`; under rocprofv3, `bench.json`: {b['ms_per_step']:.1f} ms) | {bu['ms_per_step']:.1f} | {bu['value']:.0f} |

The ladder between the two, each step measured with the same command (see `git log`): replicated statistics buffers
36.4→24.0 ms; split-K for small output grids + one HIP graph per step →22.0; branch-free range-checked buffer loads,
division-free K iteration →17.5; all-taps weight gradient for the two top levels, no launches for exactly-zero bias
gradients →14.8; LDS-transposed weight prep, shuffle-reduced head gradient →14.1; fused BatchNorm finalize →13.2; branch-free
MFMA section, 256-row tiles, LDS-DMA triple-buffered conv →12.4; `conv_dmap` (per-tap address invariants, 64×64 wave tiles,
fragment prefetch across the stage barrier) + split-K through fp32 slabs instead of atomics →12.0; `conv_halo` (input + halo
resident in LDS, lattice tiles for dilation) + hoisted epilogue loads →11.6; statistics kernels sized for the ~180 ns
serialisation of same-address fp64 atomics →11.2; sliding-window all-taps weight gradient, plain read-modify-write where
K is not split →11.1; rolling B fragments in `conv_halo`, shuffle reductions in the stem / head gradients, vectorised weight
prep, BN grids →10.8; `wgrad_dmap` (weight gradient on the conv_dmap structure) at the 32×32×256 level →10.6;
`wgrad_pw` (per-wave streaming weight gradient of the narrow 1×1 convolutions) →10.4–10.5 (box to box); loss-side kernels
(head gradient in one resident round of blocks, 16-byte-load Tanimoto sums / metrics, wider partial reduce) →10.3–10.4;
`conv_dmap` with 64-row tiles where 128×128 tiles leave half the CUs idle →10.2; pooling kernels with compile-time windows (a
window row's loads issued together), learning rate advanced on the device, branch biases summed in the conv epilogue
→10.1; the ReLU masks of the head convs applied by the
gradient writers instead of five masking passes, `wgrad_dmap` at the 64×64×128 level →10.0; `conv_pw` (per-wave streaming
kernel for the narrow 1×1 convolutions of the top levels) →9.95 on the same box (A/B 10.04 → 9.96); boxes differ by up to
±1.5 % (the committed run is on a 10.1 box).

## r01_final: kernel table (per training step; {n} steps in the trace incl. warm-up and the instrumented step)

| kernel | launches/step | ms/step | avg µs | % of GPU time | HBM MB/launch read / write (PMC) |
|---|---:|---:|---:|---:|---|
{table}

Total GPU time per step {tot/1e6/n:.2f} ms ≈ wall {b['ms_per_step']:.2f} ms: the step is GPU-bound, launch gaps are gone (one HIP graph).

## Roofline line of `bench.json`

Dominant kernel by total time: `{rf['kernel']}` — {rf['launches_per_step']} launches/step, {rf['avg_launch_us']} µs average (live
HIP-event timing inside bench.py, GPU kept busy, calibrated bracket overhead of {rf['event_pair_overhead_us_subtracted']} µs
subtracted; rocprofv3 average for the same kernel in `kernel_stats.csv`: {dom_rp:.1f} µs — see "Live vs rocprofv3" below), {rf['algorithmic_gflop_per_launch']}
algorithmic GFLOP per launch ⇒ **{rf['achieved']} TFLOP/s = {100*rf['frac']:.1f} % of the 2.5 PFLOP/s dense bf16 MFMA peak**;
HBM traffic {rf['traffic']/1e6:.1f} MB per launch (PMC).  Whole step: {b['model_tflops_per_s']} TFLOP/s of model FLOPs
({100*rf['whole_step_frac_of_peak']:.1f} % of peak).

Per kernel, same run (`all_mfma_kernels` in bench.json; the second launch of a two-launch call — split-K finisher,
all-taps reduction — is its own row, timed from an event the library records between the two launches, so every row
matches one rocprofv3 kernel name):

| kernel | launches | live avg µs | rocprofv3 avg µs | TFLOP/s (live) |
|---|---:|---:|---:|---:|
{fam}

### Live vs rocprofv3 durations

Three clocks for the same kernel, `{rf['kernel']}`: rocprofv3's dispatch duration **{dom_rp:.1f} µs**; live events in the
profiled run (`r01_final/bench.json`, this table) **{rf['avg_launch_us']:.1f} µs** — the profiler serialises every dispatch behind its own
completion signal, which lands inside the bracket; live events in an unprofiled run (`r01_final/bench_unprofiled.json`, what the
driver's bench run sees) **{bu['roofline']['avg_launch_us']:.1f} µs** — what one more kernel costs in the stream, below rocprofv3's figure because a dispatch's
begin/end timestamps include ≈2 µs of dispatch set-up and end-of-kernel release that overlap the neighbouring kernels in a
stream or graph.  (Σ rocprofv3 durations = {tot/1e6/n:.2f} ms per step ≥ the unprofiled wall step of {bu['ms_per_step']:.2f} ms, gaps included.)
The live events are created with `hipEventDisableSystemFence | hipEventReleaseToDevice` (`rua_prof_event_create`): a default
event adds the L2 write-back of the bracketed kernel's output (≈3 µs for a 16 MB output) to the bracket.  The bracket's own
cost is calibrated differentially on an idempotent library kernel (one launch vs two back-to-back launches between the markers).

CPU baseline in the same JSON: {b['cpu_baseline']['value']} patches/s on {b['cpu_baseline']['cores']} host cores
({b['cpu_baseline']['sample']}).

## What the counters say (why these are the numbers)

* HBM traffic is at the algorithmic minimum for the convolutions (e.g. `conv_halo<32,2>`: ≈46 MB read + 34 MB written for a
  33.5 MB tensor in, one out and the mask tensor on two thirds of the launches), so HBM is not the limiter: the top-level
  convs move ≈80 MB in ≈25–40 µs = 2–3 TB/s.
* The implicit-GEMM kernels (`conv_dmap`, `conv_igemm`, `wgrad_kernel`) are bound by the L2→LDS staging rate: staged bytes
  = FLOPs·(BM+BN)/(BM·BN) ≈ 151 MB per 9.66 GFLOP launch at 128×128 tiles; measured ≈12 TB/s aggregate, i.e. the gather rate
  of the L2/Infinity-cache mix (`MI355X_MICROARCH.md`: 8.6–18 TB/s), with 42 % of a wave's life spent waiting for DMA data and the
  MFMA pipe 15–18 % busy.  `conv_halo` removes the 9× per-tap re-staging for the top level (C = 32) and is then bound by
  per-block latency at 4 resident blocks per CU (load → MFMA → epilogue are serial inside a block).
* Same-address fp64 atomics serialise at ≈180 ns each (measured: 1 024 adds per address = 180 µs); everything that
  accumulates statistics keeps ≤8–64 adds per address through replicas.
* Float atomics run at ≈1.3 TB/s chip-wide; split-K partial tiles and the 8×8-level weight gradient (37.7 MB) therefore use
  plain stores / read-modify-write.

## Other workloads (same build, `gpurun_out` logs of the same session)

Before the last two steps, on a 10.09 ms box (cfg3 793 patches/s): cfg2 (single-task) 863 patches/s (9.27 ms); cfg5 (128²×7, bs 32) 3 492 patches/s (9.17 ms);
cfg4 (d7, 512², bs 4) 230 patches/s (17.4 ms, 247 TFLOP/s of model FLOPs).  Data-parallel step on one GPU with a one-rank RCCL
group (`bench.py --force-dp`, end of round 1): 10.08 ms vs 9.95 ms for the single whole-step graph on the same box.
'''
open('profiles/README.md', 'w').write(txt)
print("written", len(lines), "rows; dominant rocprof avg", dom_rp)
