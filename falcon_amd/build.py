"""Builds libfalcon_assemble.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libfalcon_assemble.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-munsafe-fp-atomics", "-Wall",
         "-Wno-unused-but-set-variable", "-Wno-unused-variable"]


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _newer(a, deps):
    if not os.path.exists(a):
        return False
    t = os.path.getmtime(a)
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force=False, verbose=False):
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(HERE, "..", "include", "falcon_assemble.h"))
    objs, jobs = [], []
    for s in sources():
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, s[:-4] + ".o")
        objs.append(obj)
        if force or not _newer(obj, [src] + hdrs):
            jobs.append([HIPCC] + FLAGS + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return r.stderr

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        for err in ex.map(run, jobs):
            if verbose and err:
                print(err)
    if jobs or not os.path.exists(OUT):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs)
    build_host(force or bool(jobs), verbose)
    return OUT


HOST_OUT = os.path.join(HERE, "libfalcon_host.so")


def build_host(force=False, verbose=False):
    """The C++ host mirror of the reference's plugin interface (falcon_amd/host), on top of the C ABI."""
    src = os.path.join(HERE, "host", "falcon_host.cpp")
    deps = [src, os.path.join(HERE, "host", "falcon_host.h"), os.path.join(HERE, "..", "include", "falcon_assemble.h")]
    if not force and _newer(HOST_OUT, deps):
        return HOST_OUT
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", HOST_OUT, src,
           "-L" + HERE, "-lfalcon_assemble", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("g++ failed:\n" + r.stdout + r.stderr)
    return HOST_OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
