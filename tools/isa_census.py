"""Instruction census of the MFMA loops of a HIP source file (gfx950), from hipcc's own assembly.

    python3 tools/isa_census.py resunet_a_mltsk_keras_amd/csrc/conv_strip.hip [name-filter] [-D...] [--ops]

For every kernel whose (demangled) name contains the filter, every innermost-first loop (a label .. the last
backward branch to it) that holds MFMAs is listed with the STATIC count of its instructions by issue class:
MFMA, VALU (everything else v_*), SALU (s_* without waits / nops / barriers / branches), LDS (ds_*), VMEM
(buffer_* / global_*), wait (s_waitcnt, s_nop, s_barrier), branch.  Static = every path of the loop body once;
bodies here are straight-line apart from uniform skips.  --ops adds the histogram of VALU mnemonics.
"""
import collections
import re
import subprocess
import sys
import tempfile
import os

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def asm_of(src, defs):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-x", "hip", "--cuda-device-only", "-S", src, "-o", out] + defs
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    with open(out) as f:
        text = f.read()
    os.unlink(out)
    return text


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return dict(zip(names, p.stdout.splitlines()))


def classify(m):
    if m.startswith("v_mfma") or m.startswith("v_smfma"):
        return "MFMA"
    if m.startswith("v_"):
        return "VALU"
    if m.startswith("ds_"):
        return "LDS"
    if m.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "VMEM"
    if m.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep", "s_setprio")):
        return "wait"
    if m.startswith(("s_cbranch", "s_branch", "s_endpgm")):
        return "branch"
    if m.startswith("s_"):
        return "SALU"
    return "other"


def functions(text):
    cur, body = None, []
    for line in text.splitlines():
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
        if m:
            cur, body = m.group(1), []
            continue
        if cur is None:
            continue
        if line.startswith(".Lfunc_end"):
            yield cur, body
            cur = None
            continue
        body.append(line)


def census(body):
    # instruction list with label positions
    ins, labels = [], {}
    for line in body:
        s = line.strip()
        if not s or s.startswith((";", ".")) and not re.match(r"^\.LBB\w+:", s):
            if re.match(r"^\.LBB\w+:", s):
                pass
            else:
                continue
        m = re.match(r"^(\.LBB\w+):", s)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        tok = s.split()
        if not tok or tok[0].startswith((";", ".")):
            continue
        ins.append((tok[0], s))
    loops = {}
    for i, (m, s) in enumerate(ins):
        if m.startswith(("s_cbranch", "s_branch")):
            tgt = s.split()[-1]
            if tgt in labels and labels[tgt] <= i:
                loops[tgt] = max(loops.get(tgt, -1), i)
    out = []
    for tgt, end in sorted(loops.items(), key=lambda kv: kv[1] - labels[kv[0]]):
        seg = ins[labels[tgt]:end + 1]
        c = collections.Counter(classify(m) for m, _ in seg)
        if c["MFMA"] == 0:
            continue
        ops = collections.Counter(m for m, _ in seg if classify(m) == "VALU")
        out.append((tgt, len(seg), c, ops))
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("-")]
    defs = [a for a in sys.argv[1:] if a.startswith("-") and a != "--ops"]
    show_ops = "--ops" in sys.argv
    src = args[0]
    filt = args[1] if len(args) > 1 else ""
    text = asm_of(src, defs)
    funcs = list(functions(text))
    names = demangle([f for f, _ in funcs])
    for f, body in funcs:
        name = names.get(f, f)
        if filt and filt not in name:
            continue
        res = census(body)
        if not res:
            continue
        print(name.split("(")[0])
        for tgt, n, c, ops in res:
            print("  loop %-12s %5d instr  MFMA %3d  VALU %4d  SALU %4d  LDS %3d  VMEM %3d  wait %3d  branch %2d   VALU/MFMA %.1f" % (
                tgt, n, c["MFMA"], c["VALU"], c["SALU"], c["LDS"], c["VMEM"], c["wait"], c["branch"], c["VALU"] / c["MFMA"]))
            if show_ops:
                print("     " + "  ".join("%s %d" % kv for kv in ops.most_common(24)))


if __name__ == "__main__":
    main()
