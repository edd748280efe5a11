"""Build libselavi_hip.so in-tree with hipcc for gfx950 (no cmake, no JIT cache).

    python -m selavi_amd.build [--force]

The .so is git-ignored but travels to the GPU box with the repo snapshot.
"""
import glob
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libselavi_hip.so")
STAMP = LIB + ".stamp"
# -pragma-unroll-threshold: the register-resident conv kernels (csrc/conv_cl16_s[rd].hip) are one straight line of
# 216-360 MFMAs per tile with compile-time fragment indices; past the default size limit "#pragma unroll" is silently
# dropped and the resident weight array lands in scratch memory
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-Wno-unused-result", "-Wno-comment", "-mllvm", "-pragma-unroll-threshold=262144"]


def _digest():
    h = hashlib.sha256(" ".join(FLAGS).encode())
    files = sorted(glob.glob(os.path.join(CSRC, "*")) +
                   glob.glob(os.path.join(HERE, "..", "include", "*.h")))
    for f in files:
        h.update(os.path.basename(f).encode())        # (not the absolute path: the GPU box mounts the repo elsewhere)
        h.update(open(f, "rb").read())
    return h.hexdigest()


def _obj_digest(src, headers):
    """An object is rebuilt when its source, any header of csrc/ or include/, or the flags change."""
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for f in [src] + headers:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def build(force=False, verbose=True):
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(STAMP) and open(STAMP).read() == dig:
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))
    headers = sorted(glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(CSRC, "*.h")) +
                     glob.glob(os.path.join(HERE, "..", "include", "*.h")))
    objs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    procs = []
    for s in srcs:
        o = os.path.join(HERE, "build", os.path.basename(s) + ".o")
        objs.append(o)
        od = _obj_digest(s, headers)
        if not force and os.path.exists(o) and os.path.exists(o + ".stamp") and open(o + ".stamp").read() == od:
            continue
        cmd = [hipcc, *FLAGS, "-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((s, o, od, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = None
    for s, o, od, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            failed = failed or s
        else:
            open(o + ".stamp", "w").write(od)
    if failed:
        raise RuntimeError(f"hipcc failed on {failed}")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-ldl", "-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    open(STAMP, "w").write(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
