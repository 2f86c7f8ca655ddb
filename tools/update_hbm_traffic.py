#!/usr/bin/env python3
"""profiles/hbm_traffic.json from a PMC dump of tools/gpu_scripts/gpu_r05_final.sh (pmc.txt): one entry per (streams,
frames, stage-0 order) of the roofline workload; bytes = FETCH_SIZE [KB] x 1024 x 2 (the guide's gfx950 correction for wide
coalesced streaming reads, MI355X_MICROARCH.md HBM section) + WRITE_SIZE [KB] x 1024.
Each entry carries the hash of the kernel sources the dump was taken on (bench.py: KERNEL_SOURCES, kernel_source_hash):
the dump's own "# kernel sources <hash>" line if it has one, else the working tree's -- run this on the tree the PMC passes ran on.
    python tools/update_hbm_traffic.py profiles/r05/p_pmc.txt"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
src = Path(sys.argv[1])
vals = {}
src_hash = None
for line in src.read_text().splitlines():
    if line.startswith("# kernel sources "):
        src_hash = line.split()[3]
    f = line.split()
    if len(f) >= 4 and f[0] in ("h3", "h4", "c3", "c4") and "nvx_fir_cascade" in line:
        vals[(f[0], f[-2])] = float(f[-1])
if src_hash is None:
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_for_hash", ROOT / "bench.py")
    b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    src_hash = b.kernel_source_hash()
S, F = 4096, 12
alg = 4 * S * F * 645120
entries = []
for order, (pf, pw), kernel in ((1, ("h3", "h4"), "nvx_fir_cascade<true, 1>"), (3, ("c3", "c4"), "nvx_fir_cascade_cic3_1")):
    fetch, write = vals.get((pf, "FETCH_SIZE")), vals.get((pw, "WRITE_SIZE"))
    if fetch is None or write is None:
        print(f"no FETCH_SIZE / WRITE_SIZE pair for stage-0 order {order} in {src}", file=sys.stderr)
        continue
    b = int(fetch * 1024 * 2 + write * 1024)
    entries.append({"streams": S, "frames": F, "stage0_order": order, "kernel": kernel, "bytes_per_launch": b, "algorithmic_bytes": alg,
                    "ratio_to_algorithmic": round(b / alg, 4), "fetch_size_kb": fetch, "write_size_kb": write, "kernel_source_sha256_16": src_hash,
                    "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), {kernel}, last dispatch ({src.relative_to(ROOT) if src.is_absolute() else src})",
                    "correction": "FETCH_SIZE x2 on gfx950 for wide coalesced streaming reads (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as reported"})
out = ROOT / "profiles" / "hbm_traffic.json"
rec = json.loads(out.read_text())
keep = [e for e in rec.get("entries", []) if (e["streams"], e["frames"], e.get("stage0_order", 1)) not in {(x["streams"], x["frames"], x["stage0_order"]) for x in entries}]
rec["entries"] = keep + entries
out.write_text(json.dumps(rec, indent=1) + "\n")
for e in entries:
    print(f"stage-0 order {e['stage0_order']}: {e['bytes_per_launch']} B = {e['ratio_to_algorithmic']} x algorithmic")
