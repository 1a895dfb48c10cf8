"""Build libessentials_amd.so (HIP, gfx950 only) in-tree with hipcc.

    python -m essentials_amd.build [--force] [--jobs N]

Every ``essentials_amd/csrc/*.hip`` is compiled to an object under
``essentials_amd/csrc/build/`` and linked into ``essentials_amd/libessentials_amd.so``.
hipcc cross-compiles for gfx950 without a GPU.  The library travels to the GPU
box as a built artefact (git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import argparse
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(PKG, "libessentials_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

FLAGS = ["-std=c++17", "-O3", f"--offload-arch={ARCH}", "-fPIC", "-I", os.path.join(ROOT, "include"),
         "-Wno-unused-result", "-Wno-inconsistent-missing-override"]


def _headers():
    hs = glob.glob(os.path.join(ROOT, "include", "**", "*.hxx"), recursive=True)
    hs += glob.glob(os.path.join(ROOT, "include", "*.h"))
    hs += glob.glob(os.path.join(CSRC, "*.hxx"))
    return hs


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths) if paths else 0.0


def _compile(src, obj, defines=()):
    cmd = [HIPCC, "-x", "hip"] + FLAGS + [f"-D{d}" for d in defines] + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {os.path.basename(src)}:\n{r.stderr[-6000:]}")
    return obj


def build_variant(name: str, defines, only=None, jobs: int | None = None) -> str:
    """Experiment build: essentials_amd/libessentials_amd.<name>.so with extra -D flags
    (loaded when ESSENTIALS_AMD_LIB points at it).  Not part of the product build."""
    # objects outside the tree: the tree is what gpurun snapshots (512 MiB cap)
    objdir = os.path.join(os.environ.get("TMPDIR", "/tmp"), "essentials_amd_variants", name)
    os.makedirs(objdir, exist_ok=True)
    sources = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    todo, objs = [], []
    for s in sources:
        base = os.path.basename(s)[:-4]
        if only and base not in only:
            objs.append(os.path.join(OBJ, base + ".o"))  # the product build's object, unchanged
            continue
        o = os.path.join(objdir, base + ".o")
        objs.append(o)
        todo.append((s, o))
    with ThreadPoolExecutor(jobs or 8) as ex:
        list(ex.map(lambda so: _compile(so[0], so[1], defines), todo))
    lib = os.path.join(PKG, f"libessentials_amd.{name}.so")
    cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", lib] + objs + \
          ["-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return lib


def build(force: bool = False, jobs: int | None = None, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    sources = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdr_time = _newest(_headers())
    todo, objs = [], []
    for s in sources:
        o = os.path.join(OBJ, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr_time):
            todo.append((s, o))
    if todo:
        if verbose:
            print(f"[build] compiling {len(todo)} HIP sources for {ARCH}", flush=True)
        jobs = jobs or min(len(todo), max(1, (os.cpu_count() or 2)))
        with ThreadPoolExecutor(jobs) as ex:
            list(ex.map(lambda so: _compile(*so), todo))
    if todo or not os.path.exists(LIB) or os.path.getmtime(LIB) < _newest(objs):
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs + \
              ["-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
        if verbose:
            print(f"[build] linked {LIB}", flush=True)
    return LIB


def build_cpp_tests(force: bool = False) -> list:
    """tests/cpp/engine_tests (+ a build with the schedule override macro) and boundary_tests:
    C++-level checks of the header surface, run by the GPU test suite."""
    outs = []
    jobs = []
    for name, source, defs in (("engine_tests", "engine_tests.hip", []),
                               ("engine_tests_override", "engine_tests.hip", ["-DGRX_ADVANCE_LB_OVERRIDE=bucketing"]),
                               ("boundary_tests", "boundary_tests.hip", [])):
        src = os.path.join(ROOT, "tests", "cpp", source)
        hdr_time = max(_newest(_headers()), os.path.getmtime(src))
        out = os.path.join(ROOT, "tests", "cpp", name)
        outs.append(out)
        if force or not os.path.exists(out) or os.path.getmtime(out) < hdr_time:
            jobs.append([HIPCC, "-x", "hip", "-std=c++17", "-O2", f"--offload-arch={ARCH}", "-I",
                         os.path.join(ROOT, "include"), "-Wno-unused-result"] + defs + [src, "-o", out])
    procs = [subprocess.Popen(j, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for j in jobs]
    for pr, j in zip(procs, jobs):
        _, err = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError(f"hipcc failed for {j[-1]}:\n{err[-4000:]}")
    return outs


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=None)
    ap.add_argument("--variant", default=None, help="experiment build name")
    ap.add_argument("-D", dest="defines", action="append", default=[])
    ap.add_argument("--only", action="append", default=None,
                    help="variant: recompile only these sources (basename without .hip), e.g. capi_bfs")
    a = ap.parse_args()
    try:
        if a.variant:
            print(build_variant(a.variant, a.defines, only=a.only, jobs=a.jobs))
        else:
            print(build(a.force, a.jobs, verbose=True))
    except RuntimeError as e:
        print(e, file=sys.stderr)
        sys.exit(1)
