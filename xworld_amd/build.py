"""Builds libxwb.so (HIP, gfx950 only) in-tree with hipcc.

    python -m xworld_amd.build [--force]

-ffp-contract=off is required: SimpleRace's float state must round exactly where the
reference's x86-64 build rounds (no FMA contraction); see DESIGN.md "SimpleRace arithmetic".
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libxwb.so")
SOURCES = ["kernels_simple.hip", "kernels_xworld.hip", "kernels_xworld_reset.hip", "kernels_xworld_ego.hip",
           "xwb_create.hip", "xwb_verbs.hip", "xwb_getters.hip", "xwb_checkpoint.hip", "xwb_comm.hip"]
HEADERS = [os.path.join(CSRC, "xwb_common.h"), os.path.join(CSRC, "xw_device.h"), os.path.join(CSRC, "xwb_language.h"), os.path.join(CSRC, "xwb_sim.h"),
           os.path.join(os.path.dirname(HERE), "include", "xwb.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function", "-fvisibility=hidden", "-fvisibility-inlines-hidden"]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def source_fingerprint():
    """sha256[:16] over the sources libxwb.so is built from (csrc/*.hip, csrc/*.h, include/*.h): stamps profiles and bench
    lines so that a stored measurement says which code it describes (the GPU box has no .git)."""
    import hashlib
    h = hashlib.sha256()
    inc = os.path.join(os.path.dirname(HERE), "include")
    files = sorted([os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))] +
                   [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith((".h", ".hpp"))])
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def build(force=False, verbose=False):
    cc = hipcc()
    objs = []
    # a build with other flags than the last one (a lab build with XWB_EXTRA_FLAGS, or the build after one) recompiles everything
    stamp = os.path.join(CSRC, ".build_flags")
    flags_now = " ".join(FLAGS + os.environ.get("XWB_EXTRA_FLAGS", "").split())
    try:
        with open(stamp) as fh:
            force = force or fh.read() != flags_now
    except OSError:
        force = True
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + HEADERS):
            cmd = [cc] + FLAGS + os.environ.get("XWB_EXTRA_FLAGS", "").split() + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            jobs.append(cmd)
    if jobs:                                                # the translation units are independent: compile them side by side
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as ex:
            for rc in ex.map(subprocess.call, jobs):
                if rc != 0:
                    raise subprocess.CalledProcessError(rc, "hipcc")
    if force or _stale(LIB, objs + [os.path.join(CSRC, "libxwb.map")]):
        cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl", "-Wl,--version-script=" + os.path.join(CSRC, "libxwb.map")]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    with open(stamp, "w") as fh:
        fh.write(flags_now)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
