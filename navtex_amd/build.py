"""Build helper: compiles libnavtex_amd.so (HIP kernels for gfx950 + host C/C++)
in-tree with hipcc, and -- for tests only -- the oracle library and the compiled
reference seams via oracle/Makefile.

    python navtex_amd/build.py            # product library
    python navtex_amd/build.py --oracle   # + oracle (and reference seams when /root/reference exists)

Run it as a script (or load it by path): importing the navtex_amd package itself
requires the library to exist already.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
CSRC = PKG / "csrc"
OBJ = PKG / "_obj"
LIB = PKG / "libnavtex_amd.so"
ARCH = "gfx950"

C_SOURCES = ["nvx_sitor.c", "nvx_wav.c", "nvx_synth_host.c", "nvx_store.c"]
HIP_SOURCES = ["nvx_cascade.hip", "nvx_fir3.hip", "nvx_demod.hip", "nvx_channelise.hip", "nvx_wideband_fused.hip", "nvx_synth.hip"]
CXX_SOURCES = ["nvx_api.cpp", "nvx_push.cpp", "nvx_shim.cpp", "nvx_capture.cpp", "nvx_wideband.cpp", "nvx_synth_dev.cpp", "nvx_fsm_host.cpp", "nvx_group.cpp"]

# -ffp-contract=off is part of the numerical contract: FIR products and sums are
# rounded separately, exactly as the reference's x86-64 build does.
COMMON = ["-O3", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden",
          f"-I{ROOT / 'include'}", f"-I{CSRC}"] + os.environ.get("NVX_EXTRA_CFLAGS", "").split()


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found: the HIP toolchain is required (there is no CPU build of this library)")


def _run(cmd):
    print(" ".join(str(c) for c in cmd), file=sys.stderr, flush=True)
    subprocess.run([str(c) for c in cmd], check=True, stdout=sys.stderr)


def _stale(out: Path, deps) -> bool:
    if not out.exists():
        return True
    t = out.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def build_lib(force: bool = False) -> Path:
    from concurrent.futures import ThreadPoolExecutor
    hipcc = _hipcc()
    OBJ.mkdir(exist_ok=True)
    headers = list(CSRC.glob("*.h")) + [ROOT / "include" / "navtex_amd.h", Path(__file__)]
    objs, jobs = [], []
    for src in C_SOURCES:
        o = OBJ / (src + ".o")
        if force or _stale(o, [CSRC / src] + headers):
            jobs.append([hipcc, "-x", "c", "-std=gnu11", "-Wall", "-Wextra", *COMMON, "-c", CSRC / src, "-o", o])
        objs.append(o)
    for src in HIP_SOURCES:
        o = OBJ / (src + ".o")
        if force or _stale(o, [CSRC / src] + headers):
            jobs.append([hipcc, f"--offload-arch={ARCH}", "-std=c++17", *COMMON, "-c", CSRC / src, "-o", o])
        objs.append(o)
    for src in CXX_SOURCES:
        o = OBJ / (src + ".o")
        if force or _stale(o, [CSRC / src] + headers):
            jobs.append([hipcc, "-x", "hip", "--offload-arch=" + ARCH, "-std=c++17", "-Wall", "-Wno-unused-value", "-Wno-unused-result", *COMMON, "-c", CSRC / src, "-o", o])
        objs.append(o)
    # the translation units are independent: compile them side by side (the cascade's twelve kernels dominate)
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        list(pool.map(_run, jobs))
    if force or _stale(LIB, objs):
        # link beside the target and rename: another process (a second rank, a test runner) never maps a half-written file
        tmp = LIB.with_name(LIB.name + f".tmp{os.getpid()}")
        _run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", tmp, "-lpthread", "-ldl"])
        os.replace(tmp, LIB)
    return LIB


def build_variant(name: str, flags, sources=("nvx_cascade.hip", "nvx_wideband_fused.hip")) -> Path:
    """TEST INFRASTRUCTURE: another build of the same library -- the named kernel sources recompiled with extra flags,
    everything else the product's own objects -- as tests/_variants/libnavtex_amd_<name>.so; load it with
    NAVTEX_AMD_LIB.  `inject` (-DNVX_INJECT_STALE=n) is the fault-injection build of the state hand-over's seal."""
    hipcc = _hipcc()
    build_lib()
    out_dir = ROOT / "tests" / "_variants"
    out_dir.mkdir(exist_ok=True)
    headers = list(CSRC.glob("*.h")) + [ROOT / "include" / "navtex_amd.h", Path(__file__)]
    objs, jobs = [], []
    for src in C_SOURCES + HIP_SOURCES + CXX_SOURCES:
        if src in sources:
            o = out_dir / f"{src}.{name}.o"
            if _stale(o, [CSRC / src] + headers):
                lang = ["-x", "c", "-std=gnu11"] if src in C_SOURCES else (["-x", "hip"] if src in CXX_SOURCES else []) + [f"--offload-arch={ARCH}", "-std=c++17"]
                jobs.append([hipcc, *lang, *COMMON, *flags, "-c", CSRC / src, "-o", o])
            objs.append(o)
        else:
            objs.append(OBJ / (src + ".o"))
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=2) as pool:
        list(pool.map(_run, jobs))
    lib = out_dir / f"libnavtex_amd_{name}.so"
    if _stale(lib, objs):
        tmp = lib.with_name(lib.name + f".tmp{os.getpid()}")
        _run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", tmp, "-lpthread", "-ldl"])
        os.replace(tmp, lib)
    return lib


INJECT_FLAGS = ["-DNVX_INJECT_STALE=5"]


def build_oracle() -> None:
    """Test infrastructure: our CPU restatement, and the reference itself when its
    sources are present (build container only)."""
    _run(["make", "-s", "-C", ROOT / "oracle", "oracle"])
    if Path("/root/reference/receiver").is_dir():
        _run(["make", "-s", "-C", ROOT / "oracle", "ref"])


if __name__ == "__main__":
    build_lib(force="--force" in sys.argv)
    if "--oracle" in sys.argv:
        build_oracle()
    # the fault-injection variant shares every object but two with the product: once it exists it is kept in step with
    # it (a variant left over from before an ABI change is refused by nvx_create, and the GPU suite's injection test with it)
    if "--inject" in sys.argv or (ROOT / "tests" / "_variants" / "libnavtex_amd_inject.so").exists():
        build_variant("inject", INJECT_FLAGS)
    if "--variant" in sys.argv:               # --variant NAME FLAG... [--sources a.hip,b.cpp]: A/B builds (tests/_variants/)
        rest = sys.argv[sys.argv.index("--variant") + 1:]
        srcs = ("nvx_cascade.hip", "nvx_wideband_fused.hip")
        if "--sources" in rest:
            k = rest.index("--sources"); srcs = tuple(rest[k + 1].split(",")); rest = rest[:k] + rest[k + 2:]
        print(build_variant(rest[0], rest[1:], srcs))
