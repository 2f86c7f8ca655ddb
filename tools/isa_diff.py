#!/usr/bin/env python3
"""Instruction-level comparison of the cascade kernels between two source trees.

    python tools/isa_diff.py <git rev> [kernel regex]

Compiles navtex_amd/csrc/nvx_cascade.hip of <git rev> and of the working tree for gfx950 (device only, -S) and compares
the instruction streams of the kernels whose demangled names match in both (labels and comments removed).  Used in
round 5 to show that pruning the A/B alternates out of the roofline kernel changed no instruction of the kernels that
ship (profiles/r05/a0_prune_isa_identical.txt); hipcc cross-compiles, no GPU needed."""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def compile_tree(tree: Path, out: Path) -> dict:
    csrc = tree / "navtex_amd" / "csrc"
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", f"-I{tree / 'include'}", f"-I{csrc}",
                    "--cuda-device-only", "-S", str(csrc / "nvx_cascade.hip"), "-o", str(out)], check=True, capture_output=True)
    text = out.read_text()
    kernels = {}
    for m in re.finditer(r"^(_Z\w+):.*?s_endpgm", text, flags=re.S | re.M):
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"^void |\(.*$", "", name)
        body = [re.sub(r"\.LBB\d+_\d+", "L", re.sub(r";.*", "", l)).strip() for l in m.group(0).splitlines()
                if l.startswith("\t") and not l.strip().startswith((".", ";"))]
        kernels[name] = body
    return kernels


def canonical(name: str) -> str:
    # round-4 trees carried two more template parameters (passes of prefetch, non-temporal loads): <RAW, NCH, 1, true>
    return re.sub(r"nvx_fir_cascade<(\w+), (\d), 1, true>", r"nvx_fir_cascade<\1, \2>", name)


def main():
    rev = sys.argv[1]
    pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else ".")
    with tempfile.TemporaryDirectory() as td:
        old = Path(td) / "old"
        old.mkdir()
        tar = subprocess.run(["git", "-C", str(ROOT), "archive", rev, "navtex_amd/csrc", "include"], check=True, capture_output=True).stdout
        subprocess.run(["tar", "x", "-C", str(old)], input=tar, check=True)
        a = {canonical(k): v for k, v in compile_tree(old, Path(td) / "a.s").items()}
        b = compile_tree(ROOT, Path(td) / "b.s")
    print(f"{rev}: {len(a)} kernels; working tree: {len(b)} kernels")
    same = True
    for name in sorted(b):
        if not pat.search(name):
            continue
        if name not in a:
            print(f"  {name}: not in {rev}"); continue
        eq = a[name] == b[name]
        same &= eq
        print(f"  {name}: {len(a[name])} / {len(b[name])} instructions, {'IDENTICAL' if eq else 'DIFFERENT'}")
    print("only in", rev + ":", sorted(set(a) - set(b)))
    return 0 if same else 1


if __name__ == "__main__":
    sys.exit(main())
