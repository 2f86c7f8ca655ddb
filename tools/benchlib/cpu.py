"""bench.py's cpu_baseline leg: the oracle (kind "port") timed on the host cores on a bounded sample of the bench batch,
and the strawman guard that runs the compiled reference itself beside it.  TEST INFRASTRUCTURE users only: this is one of
the three places allowed to call into oracle/ (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg)."""
from __future__ import annotations

import os
import time
from pathlib import Path

import numpy as np

from .placement import cpu_model, cpu_quota, physical_cores

ROOT = Path(__file__).resolve().parents[2]


def reference_check(ob, raw_stream, order=1):
    """Strawman guard: the reference ITSELF (oracle/_ref/ref_bits, built from /root/reference in the
    build container and shipped as a binary) against the port, one thread each, on the same 252 kS/s input
    (the stream's stage-0 output; the reference always runs both chains).  None when the binary is absent."""
    import subprocess, tempfile
    exe = ROOT / "oracle" / "_ref" / "ref_bits"
    if not exe.exists():
        return None
    try:
        iq252 = (ob.stage0_cic3 if order == 3 else ob.stage0)(raw_stream)[: 252000 * 4]
        with tempfile.TemporaryDirectory() as td:
            f = Path(td) / "in.bin"; iq252.tofile(f)
            t0 = time.perf_counter()
            subprocess.run([str(exe), str(f), str(Path(td) / "o")], check=True, stdout=subprocess.DEVNULL, timeout=120)
            t_ref = time.perf_counter() - t0
            ref518 = (Path(td) / "o.bits518.bin").read_bytes().decode()
        t0 = time.perf_counter()
        p = ob.Pipe(chain_mask=3, charlayer=False); p.push(iq252)
        t_port = time.perf_counter() - t0
        return {"input": f"{iq252.shape[0] / 1e6:.2f} M samples at 252 kS/s, both chains, 1 thread",
                "reference_msamples_per_s": round(iq252.shape[0] / t_ref / 1e6, 1),
                "port_msamples_per_s": round(iq252.shape[0] / t_port / 1e6, 1),
                "bits_identical": p.bits(0) == ref518}
    except Exception as e:                      # never let the guard break the benchmark line
        return {"error": str(e)[:200]}


def cpu_baseline_leg(ob, buf, pitch, n_per_stream, F, S, oraw, ncpu, args, nv):
    """The oracle (kind "port") on the host cores, on a bounded sample of the bench batch: one thread, this GPU's share of
    the box (ncpu threads) and every physical core the affinity mask allows -- stands for the reference's per-sample loop
    receiver/fir1cpp.C:80-136 and what hangs off it.  Each leg is sized to a few seconds of wall time."""
    have = os.sched_getaffinity(0)
    n_phys, smt = physical_cores(have)
    # "all cores" = the physical cores this process may really use at once: the affinity mask, cut down to the cgroup's
    # CPU quota when there is one (the pool's one-GPU boxes: 256 CPUs in the mask, a quota of 16 -- 128 threads there
    # only measure the throttle: 10.8 G samples/s against 19.4 G on 16, profiles/r03/a0_*)
    quota = cpu_quota()
    n_all = n_phys if quota is None else max(1, min(n_phys, int(quota)))
    n_all = int(os.environ.get("NVX_CPU_ALL_THREADS", n_all))
    n252 = F * nv.FRAME_IN
    # every thread owns at least one stream; the sample stays under ~4 GB of host memory
    n_cs = min(max(args.cpu_streams or 2 * ncpu, n_all), S)
    while n_cs > 2 * ncpu and n_cs * n_per_stream * 4 > (4 << 30):
        n_cs -= 1
    sample = np.empty((n_cs, n_per_stream, 2), dtype=np.int16)
    if pitch == n_per_stream:
        sample[:] = buf.download(n_cs * n_per_stream * 4, dtype=np.int16).reshape(n_cs, n_per_stream, 2)
    else:
        for s in range(n_cs):
            sample[s] = buf.download(n_per_stream * 4, offset=s * pitch * 4, dtype=np.int16).reshape(-1, 2)

    def timed(n_streams, threads, seconds):
        part = sample[:n_streams]
        t = ob.bench(part, n_streams, n252, oraw, 1, threads)[0]
        rep = max(1, int(seconds / max(t, 1e-3)))
        t = ob.bench(part, n_streams, n252, oraw, 1, threads, repeat=rep)[0]
        return n_streams * n_per_stream * rep / t / 1e6, rep, t

    n_share = min(n_cs, 2 * ncpu)
    v_share, rep, secs = timed(n_share, ncpu, 5.0)
    v_one, _, _ = timed(min(n_cs, 2), 1, 1.5)
    rate = float(n_per_stream) / (F * 0.32)                  # input samples per second of signal (2.016 M or 252 k)
    out = {
        "value": round(v_share, 2), "unit": "Msamples/s", "cores": ncpu, "cpu_model": cpu_model(), "kind": "port",
        "value_1thread": round(v_one, 2),
        "sample": f"all {F} frames of the first {n_share} streams of the bench batch ({n_share * n_per_stream / 1e6:.0f} M samples), "
                  f"processed {rep}x; oracle/nvx_oracle.c (gcc -O2 -ffp-contract=off), OpenMP over streams",
        "seconds": round(secs, 2),
        "x_real_time_per_core": round(v_one * 1e6 / rate, 1), "x_real_time": round(v_share * 1e6 / rate, 1),
    }
    where = f"{n_phys} physical cores in the affinity mask ({len(have)} CPUs, {smt} hardware threads per core), cgroup CPU quota {quota if quota is not None else 'none'}"
    if n_all > ncpu and n_cs >= n_all:
        v_all, rep_a, secs_a = timed(n_cs, n_all, 4.0)
        out.update({"value_all_cores": round(v_all, 2), "cores_all": n_all, "x_real_time_all_cores": round(v_all * 1e6 / rate, 1),
                    "all_cores_sample": f"all {F} frames of the first {n_cs} streams, processed {rep_a}x in {secs_a:.2f} s, one OpenMP thread per usable physical core: {where}"})
    else:
        # the share IS everything this process may use (or the sample cannot give every thread a stream): same measurement
        out.update({"value_all_cores": round(v_share, 2) if n_all <= ncpu else None, "cores_all": min(n_all, ncpu) if n_all <= ncpu else n_all,
                    "x_real_time_all_cores": round(v_share * 1e6 / rate, 1) if n_all <= ncpu else None,
                    "all_cores_sample": (f"= the {ncpu}-thread measurement above: {where}" if n_all <= ncpu else
                                         f"not run: {n_cs} sample streams for {n_all} threads; {where}")})
    out.update({"physical_cores_in_mask": n_phys, "smt": smt, "cpu_quota": quota})
    return out
