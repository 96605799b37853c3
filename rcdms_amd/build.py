"""Build librcdm_hip.so (gfx950) in-tree with hipcc.  No cmake, no torch extension machinery: the
library is a plain C-ABI shared object (include/rcdm.h) loaded with ctypes."""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "librcdm_hip.so")
SOURCES = ["igemm.hip", "igemm8.hip", "igemm16.hip", "wino.hip", "rowff.hip", "norm.hip", "attn.hip", "misc.hip", "runtime.hip", "comm.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fno-slp-vectorize: hipcc's SLP pass packs adjacent scalar f32 adds / muls into v_pk_*_f32, which issue slower than
# the scalars they replace next to MFMAs on gfx950 (measured +0.5 % end to end without it); the packed forms that do pay
# are written explicitly (v_pk_fma_f32 in the softmax, v_cvt_pk*, fma_mix)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize",
         "-Wno-unused-result"] + os.environ.get("RCDM_CXXFLAGS", "").split()


def _stamp():
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)):
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode())
            h.update(f.read())
    with open(os.path.join(HERE, "..", "include", "rcdm.h"), "rb") as f:
        h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    """Compile every HIP source for gfx950 and link the shared library.  Idempotent (content stamp)."""
    os.makedirs(LIBDIR, exist_ok=True)
    stamp_file = os.path.join(LIBDIR, "build.stamp")
    stamp = _stamp()
    if not force and os.path.exists(LIB) and os.path.exists(stamp_file):
        with open(stamp_file) as f:
            if f.read().strip() == stamp:
                return LIB
    if not os.path.exists(HIPCC):
        # a node without the compiler can still RUN a library that was built elsewhere and shipped with the tree — but only the
        # one that belongs to these sources: never load a stale binary silently
        have = "missing" if not os.path.exists(LIB) else "present but built from OTHER sources (stamp mismatch)"
        raise RuntimeError(f"hipcc not found at {HIPCC} and the prebuilt {LIB} is {have}: build the library where hipcc exists "
                           f"(python -m rcdms_amd.build) and ship rcdms_amd/lib/ with the tree (expected stamp {stamp[:16]}...)")
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        cmd = [HIPCC, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print("[rcdms_amd.build]", " ".join(cmd), file=sys.stderr)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode(errors='replace')}")
        if verbose and out.strip():
            print(out.decode(errors="replace"), file=sys.stderr)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-ldl", "-o", LIB]
    if verbose:
        print("[rcdms_amd.build]", " ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    with open(stamp_file, "w") as f:
        f.write(stamp)
    return LIB


def verify():
    """The library every rank is about to load IS the one built from this tree's sources (content stamp over csrc/, rcdm.h and
    the flags): what the ranks that did not build call after the build barrier, and what a node without hipcc relies on.
    Raises RuntimeError otherwise; returns the library path."""
    stamp_file = os.path.join(LIBDIR, "build.stamp")
    if not os.path.exists(LIB) or not os.path.exists(stamp_file):
        raise RuntimeError(f"{LIB} (or its build.stamp) is missing on this node: the building rank failed or rcdms_amd/lib/ did not ship")
    with open(stamp_file) as f:
        have = f.read().strip()
    want = _stamp()
    if have != want:
        raise RuntimeError(f"{LIB} was built from other sources (stamp {have[:16]}... != {want[:16]}...): rebuild with python -m rcdms_amd.build")
    return LIB


def build_variant(name, extra_flags, verbose=False):
    """A second, differently-flagged build next to the product library (debug / A-B experiments): objects under
    lib/<name>/, library lib/librcdm_<name>.so; select it at run time with RCDM_LIB=<path>.  Never touches
    librcdm_hip.so or its stamp."""
    odir = os.path.join(LIBDIR, name)
    os.makedirs(odir, exist_ok=True)
    lib = os.path.join(LIBDIR, f"librcdm_{name}.so")
    procs, objs = [], []
    for src in SOURCES:
        obj = os.path.join(odir, src.replace(".hip", ".o"))
        cmd = [HIPCC, *FLAGS, *extra_flags, "-c", os.path.join(CSRC, src), "-o", obj]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode(errors='replace')}")
        if verbose and out.strip():
            print(out.decode(errors="replace"), file=sys.stderr)
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-ldl", "-o", lib])
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
