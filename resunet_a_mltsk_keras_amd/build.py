"""Builds librua_hip.so (gfx950) in-tree with hipcc.  No torch involved: the library is a plain
C-ABI shared object (include/rua_hip.h)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.environ.get("RUA_BUILD_OUT") or os.path.join(HERE, "librua_hip.so")      # RUA_BUILD_OUT / RUA_BUILD_FLAGS: experiment builds (A/B through RUA_LIB_PATH)
SOURCES = ["conv_mfma.hip", "conv_strip.hip", "conv_band.hip", "conv_band64.hip", "conv_band128.hip", "conv_img2.hip", "elementwise.hip", "small_conv.hip", "loss_optim.hip", "capi.cpp"]
# per-file flags.  conv_strip: without the SLP vectorizer - it packed the per-lane statistics sums into v_pk_add_f32 / v_pk_fma_f32 on
# register pairs it first had to assemble with v_mov_b32 (32 per two row stages); the scalar forms issue in the MFMA shadows
FILE_FLAGS = {"conv_strip.hip": ["-fno-slp-vectorize"]}


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "rua_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        raise RuntimeError(f"hipcc not found at {hipcc}")
    objs, procs = [], []
    extra = os.environ.get("RUA_BUILD_FLAGS", "").split()
    bdir = os.path.join(HERE, "build" + ("_" + "".join(c if c.isalnum() else "_" for c in "".join(extra)) if extra else ""))
    os.makedirs(bdir, exist_ok=True)
    for s in SOURCES:
        o = os.path.join(bdir, s + ".o")
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17"] + FILE_FLAGS.get(s, []) + extra + ["-x", "hip", "-c", os.path.join(CSRC, s), "-o", o]
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(o)
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError(f"hipcc failed on {s}")
        if verbose and out.strip():
            sys.stderr.write(out.decode())
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
