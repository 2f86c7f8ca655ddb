#!/usr/bin/env python3
"""bench.py -- headline benchmark of the NAVTEX hot path on MI355X.

Workload (BASELINE.json configs[3], the "HBM roofline run"; configs[4] is the
same per GPU): 4096 independent synthetic 170 Hz-shift FSK channels per GPU,
each an int16 IQ stream at 2.016 MS/s resident in HBM, run through
stage 0 (/8) -> FIR1 -> mixer -> FIR2 -> FIR3 -> FSK discriminator -> bit sync
-> mark/space decision -> host SITOR-B character layer.

One "step" = one pass of that path over the whole resident batch
(streams x frames x 645120 samples).  Multi-GPU: one process per GPU
(torch.distributed / RCCL only for the barrier, the max-over-ranks of the
elapsed time and the min-over-ranks of the parity verdict); streams shard one
subset per GPU, no data-path collective, weak scaling.

`python bench.py --gpus N` works by itself: without a launcher's RANK in the
environment it starts the N ranks as a child process (torch.distributed.run)
before touching torch or the GPU, and relays the child's line and exit status.
Under a launcher (the driver's `python -m torch.distributed.run ... bench.py
--gpus N`) it is simply one of the ranks.

Every rank checks its OWN shard against the oracle before the timed region (all
streams at N = 1, a spread sample of them per rank otherwise) AND after it: the
bits the handle has accumulated over the warm-up and the timed launches -- the
launches that were measured, with their state carried launch to launch --
against the oracle fed the same frames as many times (`parity_after_timed`).
The line's `parity` is the minimum over ranks of both, and a failed parity makes
the process exit with status 3 after printing the line.

At N = 1 the default line also carries, each outside the headline's timed
region and each with its own parity checks (first launch and last):
`stage0_third_order`, `push_path` (host-fed streaming through nvx_push_iq,
PCIe-inclusive), `live_latency` (two capture rings fed at the real rate),
`variant_a` (the 252 kS/s cascade kernel) and `wideband` (the fused channeliser
+ cascade kernel).  A leg that raises is named in `legs_failed` and the process
exits with status 4 after printing the line (--allow-leg-errors: status 0).

`python bench.py --gpus N --group` runs the same workload through ONE process
and the library's own multi-GPU object (nvx_group: one handle + host thread per
device, no collective) instead of one process per GPU.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)
BYTES_PER_SAMPLE = 4           # int16 I + int16 Q, each read from HBM exactly once
# fp64 roof of the 252 kS/s kernels: the reference's arithmetic is mul-then-add, never fused, so the roof is the
# fp64 ISSUE rate: 256 CUs x 4 SIMDs x 16 lanes per clock x 2.4 GHz (a wave64 fp64 instruction takes 4 cycles)
FP64_NOFMA_PEAK_TOPS = 256 * 4 * 16 * 2.4e9 / 1e12            # 39.3
# fp64 operations per 252 kS/s complex input sample: FIR1 37 taps x 2 components x (mul + add) / 4, then per chain
# mixer 6 / 4, FIR2 47 x 2 x 2 / 28, FIR3 71 x 2 x 2 / 280   
FLOP_FIR1, FLOP_FIR3, FLOP_PER_CHAIN = 37.0, 71 * 4 / 280, 6 / 4 + 47 * 4 / 28 + 71 * 4 / 280


def flops_per_sample(chains: int, fir3_inside: bool = True) -> float:
    """fp64 operations a cascade kernel executes per 252 kS/s input sample; fir3_inside False: the fused wideband kernel, whose
    waves end at FIR2 (FIR3 is nvx_fir3, a kernel of its own whose time is reported beside it) -- its roof fraction
    counts what IT executes, not the path's total."""
    return FLOP_FIR1 + chains * (FLOP_PER_CHAIN - (0.0 if fir3_inside else FLOP_FIR3))


# ---- what the fused wideband kernel's fp64 roof fraction is made of (r6) -----------------------------------------
# Timing-only elimination probes on the shipped form of nvx_wideband_fused (profiles/r05/b0_fused_elimination_probes.txt:
# three interleaved rounds, 512 streams x 12 frames): the phases of a pass ADD -- they run one after the other behind the
# barriers.  Shares of the kernel's time: removing the cascade pass leaves 6.47 of 17.889 ms, removing the channeliser's
# arithmetic leaves 14.141, removing both barriers 16.642.
WB_SHARE_CASCADE = round(1 - 6.470 / 17.889, 3)         # 0.638: FIR1, mixers, FIR2 of 8 sub-bands x 2 chains -- all of the credited fp64 work
WB_SHARE_CHANNELISER = round(1 - 14.141 / 17.889, 3)    # 0.210: integer arithmetic that earns no fp64 credit
WB_SHARE_BARRIERS = round(1 - 16.642 / 17.889, 3)       # 0.070
# Vector instructions of the channeliser phase (nvx_pfb.h, nvx_pfb_instant_split: a lane pair per output instant, one
# component each), counted in the compiled kernel between its two barriers (tests/test_isa.py holds the count): 128 per
# (instant, component) -- 48 v_dot2c_i32_i16 (one tap on one sample each), 16 shifts, 27 adds / subs, 8 v_med3 clamps, 8
# v_cvt_f64_i32, 8 moves, 5 DPP exchanges with the partner lane, two 64-bit products for the 45-degree twiddles -- beside 12
# ds_read_b128 and 8 ds_write_b64; two components, eight raw samples per instant.
WB_CHANNELISER_VALU_PER_LANE = 128
WB_INT_OPS_PER_RAW_SAMPLE = WB_CHANNELISER_VALU_PER_LANE * 2 / 8          # 32


def wideband_decomposition(frac, fps):
    """What a bare roof fraction of nvx_wideband_fused hides: the part of the kernel that does the credited fp64 work runs
    at frac / WB_SHARE_CASCADE of the roof (the efficiency of the stand-alone 252 kS/s kernel, variant_a), and the
    channeliser's integer instructions -- the same issue slots as fp64 ones on this chip, 4 cycles per wave64 -- are not in
    the numerator at all."""
    if not frac:
        return None
    return {"source": "profiles/r05/b0_fused_elimination_probes.txt: timing-only probe builds of the shipped kernel form; static shares applied to this run's time",
            "share_of_kernel_time": {"cascade_pass": WB_SHARE_CASCADE, "channeliser_arithmetic": WB_SHARE_CHANNELISER, "barriers": WB_SHARE_BARRIERS,
                                     "rest": round(1 - WB_SHARE_CASCADE - WB_SHARE_CHANNELISER - WB_SHARE_BARRIERS, 3)},
            "cascade_pass_frac_of_fp64_roof": round(frac / WB_SHARE_CASCADE, 4),
            "channeliser_int_ops_per_raw_sample": WB_INT_OPS_PER_RAW_SAMPLE,
            "valu_issue_frac_counting_integer_ops": round(frac * (fps + WB_INT_OPS_PER_RAW_SAMPLE) / fps, 4),
            "reading": "the phases of a pass add (barriers between them): the cascade pass, which does ALL the credited fp64 operations, takes 64 % of the kernel and "
                       "alone runs at cascade_pass_frac_of_fp64_roof (about variant_a's efficiency); the channeliser's ~32 integer vector instructions per raw "
                       "sample cost the same issue slots as fp64 ones and earn no credit -- counted like fp64 operations the kernel issues at "
                       "valu_issue_frac_counting_integer_ops of the roof"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--streams", type=int, default=4096, help="IQ streams per GPU")
    ap.add_argument("--frames", type=int, default=12, help="frames (0.32 s each) resident per stream")
    ap.add_argument("--cpu-streams", type=int, default=0, help="streams in the CPU-baseline sample (0 = auto)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--verify", type=int, default=-1, metavar="N",
                    help="streams per rank checked against the oracle before the timed region (-1 = all of them at N = 1, "
                         "32 spread over the rank's shard otherwise; 0 is not accepted: an unchecked number is no number)")
    ap.add_argument("--no-charlayer", action="store_true")
    ap.add_argument("--pitch-pad", type=int, default=0, help="extra complex samples between streams (multiple of 4)")
    ap.add_argument("--wideband", type=int, default=0, metavar="W",
                    help="wideband mode (SURVEY 8f rank 2, not the headline): W streams at 2.016 MS/s per GPU, each carrying "
                         "16 NAVTEX carriers; channeliser -> 8W sub-bands at 252 kS/s -> both chains")
    ap.add_argument("--stage0", choices=("boxcar", "cic3"), default="boxcar",
                    help="raw-rate front end: integrate-and-dump (default, the headline) or its third-order form "
                         "(nvx_config.stage0_order = 3: 76 dB of alias rejection instead of 25)")
    ap.add_argument("--no-stage0-extra", action="store_true", help="skip the third-order stage 0 measurement beside the headline (N = 1)")
    ap.add_argument("--no-legs", action="store_true",
                    help="skip the side legs of the default line (N = 1): variant_a, wideband, push_path (each a few seconds, outside the timed region)")
    ap.add_argument("--allow-leg-errors", action="store_true", help="a side leg that raises is recorded but does not fail the run (default: exit status 4)")
    ap.add_argument("--after-timed", type=int, default=64, metavar="N",
                    help="streams per rank whose accumulated bits are checked against the oracle AFTER the timed region (at least 32 per rank)")
    ap.add_argument("--cic3-verify", type=int, default=-1, metavar="N",
                    help="streams of the stage0_third_order leg checked against the oracle (-1 = every stream)")
    ap.add_argument("--group", action="store_true",
                    help="one process, nvx_group over devices 0..N-1 (NVX_BENCH_GROUP_DEVICES=0,0 names them explicitly) instead of one process per GPU")
    ap.add_argument("--variant-a", action="store_true",
                    help="reference-native input rate: streams at 252 kS/s, no stage 0 (SURVEY 8d Variant A; fp64-bound, "
                         "reported for completeness -- the headline workload is the default 2.016 MS/s Variant B)")
    return ap.parse_args()


# ----------------------------------------------------------------------------- host placement (no GPU call in here)
def _parse_cpulist(text: str) -> set:
    out = set()
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.update(range(int(a), int(b or a) + 1))
    return out


def _gpu_numa(dev: int):
    """(numa node, local CPU set) of HIP device `dev`, from the KFD topology in sysfs; (None, None) when unknown.
    No HIP call: the affinity must be in place before the runtime starts its own threads."""
    try:
        vis = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")
        if vis and all(v.strip().isdigit() for v in vis.split(",")):
            dev = int(vis.split(",")[dev])
        gpus = []
        for node in sorted(glob.glob("/sys/class/kfd/kfd/topology/nodes/*"), key=lambda p: int(os.path.basename(p))):
            props = dict(line.split(None, 1) for line in open(os.path.join(node, "properties")).read().splitlines() if " " in line)
            if int(props.get("simd_count", "0")) > 0:
                gpus.append(props)
        minor = int(gpus[dev]["drm_render_minor"])
        base = f"/sys/class/drm/renderD{minor}/device"
        return int(open(base + "/numa_node").read()), _parse_cpulist(open(base + "/local_cpulist").read())
    except Exception:
        return None, None


def place_rank(device: int, local_world: int, local_rank: int):
    """Bind this process to the CPUs of its GPU's NUMA node and size its host thread pools from its share of them."""
    numa, cpus = _gpu_numa(device)
    have = os.sched_getaffinity(0)
    bound = False
    if cpus:
        both = have & cpus
        if both:
            try:
                os.sched_setaffinity(0, both); have = both; bound = True
            except OSError:
                pass
    # ranks of this node that share the NUMA node (devices are dealt in rank order)
    sharing = 1
    if local_world > 1:
        mine = numa
        sharing = sum(1 for r in range(local_world) if _gpu_numa(r)[0] == mine) if mine is not None else local_world
        sharing = max(1, sharing)
    threads = max(1, min(16, len(have) // sharing))
    # one GPU of the pool's boxes comes with a 16-thread CPU share whatever the affinity mask says
    threads = int(os.environ.get("NVX_CPU_THREADS", threads))
    os.environ.setdefault("NVX_HOST_THREADS", str(threads))            # the library's character-layer pool
    return {"numa_node": numa, "bound": bound, "cpus": len(have), "ranks_on_numa_node": sharing, "threads": threads}


def physical_cores(have) -> tuple:
    """(physical cores among the CPUs of `have`, hardware threads per core) from the sysfs topology."""
    seen, smt = set(), 1
    for c in sorted(have):
        try:
            sib = _parse_cpulist(open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read())
        except (OSError, ValueError):
            sib = {c}
        smt = max(smt, len(sib))
        seen.add(min(sib))
    return max(1, len(seen)), smt


def cpu_quota():
    """CPUs the cgroup lets this process use at once (cpu.max), or None when unlimited / unknown."""
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            f = open(path).read().split()
            if path.endswith("cpu.max"):
                return None if f[0] == "max" else round(int(f[0]) / int(f[1]), 2)
            q = int(f[0])
            return None if q <= 0 else round(q / int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()), 2)
        except (OSError, ValueError, IndexError):
            continue
    return None


def self_launch(args, script=None, argv=None) -> None:
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD (torch.distributed.run, one process per
    GPU), relay its one JSON line and its exit status.  Runs before this process has imported torch or made any GPU
    call: a process that has touched the GPU must never exec, and this one neither touches it nor execs."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC only on this pool (RCCL needs it)
    env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script or Path(__file__).resolve())] + list(sys.argv[1:] if argv is None else argv)
    print("bench.py: launching " + " ".join(cmd[2:8]) + " ...", file=sys.stderr, flush=True)
    # the ranks in a session of their own: a SIGTERM / SIGINT / SIGHUP that reaches only this process (a driver's timeout
    # that is not a process-group kill) is passed on to all of them -- they must not be left holding the GPUs
    import signal
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)      # stderr goes straight through

    def pass_on(signum, _frame):
        try:
            os.killpg(child.pid, signal.SIGTERM)
            try:
                child.wait(timeout=10)
            except subprocess.TimeoutExpired:
                os.killpg(child.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        sys.exit(128 + signum)

    for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        signal.signal(sig, pass_on)
    out, _ = child.communicate()
    lines = out.splitlines()
    js = [l for l in lines if l.startswith("{")]
    for l in lines:
        if not js or l is not js[-1]:
            print(l, file=sys.stderr)
    if js:
        print(js[-1], flush=True)
    elif child.returncode == 0:
        raise SystemExit("bench.py: the ranks printed no JSON line")
    sys.exit(child.returncode)


# the sources that define the roofline kernels' device code: a PMC record of their traffic holds for exactly these bytes
KERNEL_SOURCES = ("nvx_cascade.hip", "nvx_cascade_wave.h", "nvx_kernels.h", "nvx_device.h", "nvx_tables.h")


def kernel_source_hash() -> str:
    import hashlib
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        h.update(name.encode()); h.update((ROOT / "navtex_amd" / "csrc" / name).read_bytes())
    return h.hexdigest()[:16]


def traffic_record(S: int, F: int, order: int):
    """(bytes per launch, where it comes from) from profiles/hbm_traffic.json: one PMC record per (streams, frames, stage-0
    order) -- rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, corrected as MI355X_MICROARCH.md prescribes.  A
    static record of this workload on an earlier box, not a measurement of this run -- and only of the kernel it was taken
    on: an entry carries the hash of the kernel's sources (KERNEL_SOURCES) at the time of the PMC passes, and a record of
    other sources is not quoted.  (None, why) for any other shape or source."""
    tf = ROOT / "profiles" / "hbm_traffic.json"
    have = []
    try:
        rec = json.loads(tf.read_text())
        for e in rec.get("entries", [rec] if "bytes_per_launch" in rec else []):
            have.append((e.get("streams"), e.get("frames"), e.get("stage0_order", 1)))
            if e.get("streams") == S and e.get("frames") == F and e.get("stage0_order", 1) == order:
                now = kernel_source_hash()
                if e.get("kernel_source_sha256_16") != now:
                    return None, (f"null: the PMC record in profiles/hbm_traffic.json was taken on kernel sources {e.get('kernel_source_sha256_16')}, "
                                  f"these are {now} (tools/gpu_scripts/gpu_r05_final.sh collects a new one, tools/update_hbm_traffic.py writes it)")
                return e.get("bytes_per_launch"), ("profiles/hbm_traffic.json (static: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload on these "
                                                   f"kernel sources ({now}), not measured by this run; {e.get('source', '')})")
    except Exception as e:
        return None, f"null: profiles/hbm_traffic.json unreadable ({type(e).__name__})"
    return None, (f"null: profiles/hbm_traffic.json holds PMC records of (streams, frames, stage-0 order) {have}, not of ({S}, {F}, {order}) "
                  "(tools/gpu_scripts/gpu_r05_final.sh collects them)")


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


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


# ----------------------------------------------------------------------------- side legs of the default line (N = 1)
# Every kernel family and the streaming path get a driver-timed number in the same record as the headline, each with its
# own parity sample against the oracle, each a few seconds, all OUTSIDE the headline's timed region.
def fir3_avg_ms(pipe, launches) -> float:
    """Average HIP-event time of nvx_fir3 per launch (wideband handles; 0 elsewhere, and with an older library in an A/B run)."""
    try:
        return pipe.kernel_time_stats(2)[0] / max(launches, 1)
    except Exception:
        return 0.0


def leg_stage0_cic3(nv, ob, fullsize, buf, pitch, n_per_stream, S, F, device, ncpu, char_layer, samples_per_step, bytes_per_step, n_verify, n_after, steps=10, warmup=2):
    """The headline's batch through nvx_config.stage0_order = 3 (the stage the vendor library's closed /8 stands for:
    receiver/capt_sched.c:412-413).  Checked like the headline: n_verify streams (-1: all) from reset, n_after after the
    timed launches; traffic from its own PMC record."""
    p3 = nv.Pipeline(n_streams=S, raw_rate=True, chain_mask=nv.CHAIN_518, max_frames=F, char_layer=char_layer, device=device, stage0_order=3)
    try:
        p3.process_resident(buf, pitch, 0, F); p3.fetch()
        ids3 = fullsize.spread(S, S if n_verify < 0 else min(S, max(1, n_verify)))
        checked3, bad3, secs3 = fullsize.verify_streams(ob, buf, pitch, n_per_stream, 3, lambda s: p3.bits(s, 0), ids3, ncpu)
        p3.reset()
        for _ in range(warmup): p3.process_resident(buf, pitch, 0, F)
        p3.fetch(); p3.enable_timing(True); p3.kernel_time_stats(0, reset=True); p3.wait_stats(reset=True)
        t3 = time.perf_counter()
        for _ in range(steps): p3.process_resident(buf, pitch, 0, F)
        p3.fetch()
        e3 = time.perf_counter() - t3
        c3, n3 = p3.kernel_time_stats(0)
        c3 /= max(n3, 1)
        w_polls, w_units, w_launches = p3.wait_stats()
        ids_after = fullsize.spread(S, min(S, n_after))
        checked_a, bad_a, secs_a = fullsize.verify_replay(ob, buf, pitch, n_per_stream, 3, lambda s: p3.bits(s, 0), ids_after, ncpu, warmup + steps)
        stale, failures, _ = p3.integrity_stats()
        traffic, traffic_source = traffic_record(S, F, 3)
        achieved = bytes_per_step / (c3 * 1e-3) / 1e9 if c3 > 0 else None
        if bad3 or bad_a:
            print(f"PARITY FAILURE (third-order stage 0): first launch {len(bad3)} of {checked3} streams differ (first {bad3[:8]}), "
                  f"after the timed launches {len(bad_a)} of {checked_a} (first {bad_a[:8]})", file=sys.stderr)
        return {"what": "the same batch with nvx_config.stage0_order = 3 (22-tap CIC^3, 76 dB of alias rejection at the NAVTEX offsets where the "
                        "headline's integrate-and-dump has 25: the front end a receiver would ship); not part of the timed region above",
                "steps": steps, "ms_per_step": round(e3 / steps * 1e3, 3), "value": round(samples_per_step * steps / e3 / 1e6, 1),
                "cascade_avg_launch_ms": round(c3, 3), "frac_of_hbm_peak": round(achieved / HBM_PEAK_GBS, 4) if achieved else None,
                "roofline": {"bound": "hbm", "kernel": "nvx_fir_cascade_cic3_1", "achieved": round(achieved, 1) if achieved else None, "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4) if achieved else None, "traffic": traffic, "traffic_source": traffic_source,
                             "algorithmic_bytes_per_launch": bytes_per_step, "avg_launch_ms": round(c3, 3), "launches": int(n3),
                             "handoff": {"units_waited_frac": round(w_units / max(1, w_launches * S * F), 4), "stale_detected": stale,
                                         "launches_failed_integrity": failures}},
                "parity": not bad3 and not bad_a, "parity_streams_checked": checked3, "parity_seconds": round(secs3, 1),
                "parity_after_timed": not bad_a, "parity_after_timed_streams": checked_a, "parity_after_timed_launches": warmup + steps,
                "parity_after_timed_seconds": round(secs_a, 1)}
    finally:
        p3.close()


def leg_variant_a(nv, ob, fullsize, signals, S, device, ncpu, char_layer, frames=96, steps=5, n_check=64):
    """Reference-native rate (SURVEY 8d Variant A): S streams x `frames` frames at 252 kS/s through nvx_fir_cascade<252k,1>
    (the same bytes per launch as the headline when frames = 96).  fp64-issue-bound: frac is of the 39.3 T no-FMA roof."""
    n_per = frames * nv.FRAME_IN
    buf = nv.DeviceBuffer(S * n_per * 4, device=device)
    try:
        nv.synth_device([signals.stream_params(nv, s, nv.RATE_IN)[0] for s in range(S)], nv.RATE_IN, n_per, buf, n_per)
        p = nv.Pipeline(n_streams=S, raw_rate=False, chain_mask=nv.CHAIN_518, max_frames=frames, char_layer=char_layer, device=device)
        p.process_resident(buf, n_per, 0, frames); p.fetch()
        checked, bad, _ = fullsize.verify_streams(ob, buf, n_per, n_per, False, lambda s: p.bits(s, 0), fullsize.spread(S, min(S, n_check)), ncpu)
        p.reset()
        p.process_resident(buf, n_per, 0, frames); p.fetch()
        p.enable_timing(True); p.kernel_time_stats(0, reset=True); p.wait_stats(reset=True)
        t0 = time.perf_counter()
        for _ in range(steps):
            p.process_resident(buf, n_per, 0, frames)
        p.fetch()
        el = time.perf_counter() - t0
        c_ms, n_l = p.kernel_time_stats(0); c_ms /= max(n_l, 1)
        w_polls, w_units, w_launches = p.wait_stats()
        # ... and what the timed launches left behind: 1 + steps launches over the same frames since the reset, state carried
        checked_a, bad_a, _ = fullsize.verify_replay(ob, buf, n_per, n_per, False, lambda s: p.bits(s, 0), fullsize.spread(S, min(S, n_check)), ncpu, 1 + steps)
        stale, failures, _ = p.integrity_stats()
        p.close()
        bad = list(bad) + list(bad_a)
        tops = flops_per_sample(1) * S * n_per / (c_ms * 1e-3) / 1e12 if c_ms > 0 else None
        return {"what": f"VARIANT A: {S} streams x {frames} frames at 252 kS/s ({S * n_per * 4 / 1e9:.1f} GB), no stage 0, one chain; not part of the timed region above",
                "kernel": "nvx_fir_cascade<252k,1>", "steps": steps, "ms_per_step": round(el / steps * 1e3, 3),
                "value": round(S * n_per * steps / el / 1e6, 1), "cascade_avg_launch_ms": round(c_ms, 3),
                "roofline": {"bound": "fp64_valu", "achieved": round(tops, 2) if tops else None, "peak": round(FP64_NOFMA_PEAK_TOPS, 1), "unit": "TFLOP/s",
                             "frac": round(tops / FP64_NOFMA_PEAK_TOPS, 4) if tops else None, "flop_per_sample": round(flops_per_sample(1), 2),
                             "hbm_gbs": round(S * n_per * 4 / (c_ms * 1e-3) / 1e9, 1) if c_ms > 0 else None},
                "handoff_units_waited_frac": round(w_units / max(1, w_launches * S * frames), 4),
                "handoff": {"stale_detected": stale, "launches_failed_integrity": failures},
                "parity": not bad, "parity_streams_checked": checked, "parity_after_timed": not bad_a, "parity_after_timed_streams": checked_a,
                "parity_after_timed_launches": 1 + steps}
    finally:
        buf.free()


def leg_wideband(nv, ob, signals, W, F, device, ncpu, char_layer, steps=8, n_check_wide=8):
    """Wideband path: W streams at 2.016 MS/s, 16 carriers each, through nvx_wideband_fused."""
    n_raw, n_sub = F * nv.FRAME_RAW, F * nv.FRAME_IN
    raw = nv.DeviceBuffer(W * n_raw * 4, device=device)
    try:
        nv.synth_device(wideband_streams(nv, signals, 0, W), nv.RATE_RAW, n_raw, raw, n_raw)
        p = nv.Pipeline(n_streams=W, wideband=True, chain_mask=3, max_frames=F, char_layer=char_layer, device=device)
        p.process_resident(raw, n_raw, 0, F); p.fetch()
        nw = min(W, n_check_wide)
        part = raw.download(nw * n_raw * 4, dtype=np.int16).reshape(nw, n_raw, 2)
        _secs, cpu_bits = ob.bench_wide(part, nw, n_sub, ncpu, want_bits=True)
        gpu_bits = [p.bits(s, c) for s in range(8 * nw) for c in (0, 1)]
        ok = gpu_bits == cpu_bits and all(len(b) > 0 for b in cpu_bits)
        p.reset()
        p.process_resident(raw, n_raw, 0, F); p.fetch()
        p.enable_timing(True); p.kernel_time_stats(0, reset=True)
        t0 = time.perf_counter()
        for _ in range(steps):
            p.process_resident(raw, n_raw, 0, F)
        p.fetch()
        el = time.perf_counter() - t0
        c_ms, n_l = p.kernel_time_stats(0); c_ms /= max(n_l, 1)
        f3_ms = fir3_avg_ms(p, n_l)
        # ... and what the timed launches left behind (1 + steps launches since the reset, channeliser halo and filter state carried)
        _secs, want_after = ob.replay_wide(part, nw, n_sub, ncpu, 1 + steps)
        got_after = [p.bits(s, c) for s in range(8 * nw) for c in (0, 1)]
        ok_after = got_after == want_after and all(len(b) > 0 for b in want_after)
        stale, failures, _ = p.integrity_stats()
        p.close()
        ok = ok and ok_after
        sub_samples = 8 * W * n_sub
        fps = flops_per_sample(2, fir3_inside=False)      # the fused kernel's waves end at FIR2: nvx_fir3 does the rest
        tops = fps * sub_samples / (c_ms * 1e-3) / 1e12 if c_ms > 0 else None
        return {"what": f"WIDEBAND: {W} streams x 2.016 MS/s x {F} frames, 16 NAVTEX carriers each (8 sub-bands x 2 chains) = {16 * W} carriers; "
                        "channeliser + two-chain cascades (FIR1, mixers, FIR2) in one kernel, FIR3 in nvx_fir3 beside the next launch; not part of the timed region above",
                "kernel": "nvx_wideband_fused",
                "steps": steps, "ms_per_step": round(el / steps * 1e3, 3), "value": round(W * n_raw * steps / el / 1e6, 1),
                "carrier_equivalent_msamples_per_s": round(16 * W * n_raw * steps / el / 1e6, 1),
                "kernel_avg_launch_ms": round(c_ms, 3), "fir3_avg_launch_ms": round(f3_ms, 3),
                "roofline": {"bound": "fp64_valu", "achieved": round(tops, 2) if tops else None, "peak": round(FP64_NOFMA_PEAK_TOPS, 1), "unit": "TFLOP/s",
                             "frac": round(tops / FP64_NOFMA_PEAK_TOPS, 4) if tops else None, "flop_per_sample": round(fps, 2),
                             "hbm_gbs": round(W * n_raw * 4 / (c_ms * 1e-3) / 1e9, 1) if c_ms > 0 else None,
                             "decomposition": wideband_decomposition(tops / FP64_NOFMA_PEAK_TOPS if tops else None, fps),
                             "note": "the fp64 operations the kernel itself executes (FIR1, mixers, FIR2 of both chains; since r4 FIR3 -- 2.03 of the path's "
                                     "55.46 operations per sample -- is nvx_fir3, fir3_avg_launch_ms, beside the next launch); the channeliser's integer work rides on top"},
                "handoff": {"stale_detected": stale, "launches_failed_integrity": failures},
                "parity": ok, "parity_carriers_checked": 16 * nw, "parity_after_timed": ok_after, "parity_after_timed_launches": 1 + steps}
    finally:
        raw.free()


def leg_push_path(nv, ob, buf, pitch, F, device, ncpu, n_streams=64, frames_per_push=4, passes=24, pushers=4):
    """Streaming runs are reported separately: `n_streams` streams fed from HOST memory through nvx_push_iq ->
    pinned staging -> hipMemcpyAsync -> kernels -> bits, the loop that replaces receiver/capt_sched.c:484-528.  PCIe-bound by
    nature (4 B per sample); never `value`."""
    fpp = min(frames_per_push, F)
    n_fr = (F // fpp) * fpp
    n_per = n_fr * nv.FRAME_RAW
    host = np.empty((n_streams, n_per, 2), dtype=np.int16)
    for s in range(n_streams):
        host[s] = buf.download(n_per * 4, offset=s * pitch * 4, dtype=np.int16).reshape(-1, 2)
    chunk = fpp * nv.FRAME_RAW
    # both chains, the reference's own wiring (receiver/nav_sched.C:10-17) -- which also keeps these small launches out
    # of the headline kernel's rocprofv3 statistics (they run nvx_fir_cascade<raw,2>)
    p = nv.Pipeline(n_streams=n_streams, raw_rate=True, chain_mask=nv.CHAIN_518 | nv.CHAIN_490, max_frames=fpp, push_mode=True, char_layer=True, device=device)

    # one "capture thread" per group of streams, as a receiver with several radios has them: big pushes copy into the
    # pinned staging without the handle's lock, so the threads fill their streams' staging side by side
    import threading
    n_thr = max(1, min(pushers, n_streams))

    def feed(t):
        for c0 in range(0, n_per, chunk):
            for s in range(t, n_streams, n_thr):
                p.push(s, host[s, c0:c0 + chunk])

    def one_pass():
        if n_thr == 1:
            return feed(0)
        ths = [threading.Thread(target=feed, args=(t,)) for t in range(n_thr)]
        for th in ths: th.start()
        for th in ths: th.join()

    one_pass(); p.flush()                                   # from reset state: the first checked pass (also the warm-up)
    _secs, want = ob.replay(host, n_streams, n_per // 8, True, 3, ncpu, 1)
    got = [[p.bits(s, 0), p.bits(s, 1)] for s in range(n_streams)]
    ok_first = got == want and all(len(b[0]) > 0 and len(b[1]) > 0 for b in want)
    t0 = time.perf_counter()
    for _ in range(passes):
        one_pass()
    p.flush()
    el = time.perf_counter() - t0
    # after the LAST pass: both chains of every stream, everything decoded since the reset (1 + passes passes over the same
    # frames, state carried from pass to pass) == the oracle fed the same
    _secs, want = ob.replay(host, n_streams, n_per // 8, True, 3, ncpu, 1 + passes)
    got = [[p.bits(s, 0), p.bits(s, 1)] for s in range(n_streams)]
    ok_last = got == want and all(len(b[0]) > 0 and len(b[1]) > 0 for b in want)
    stale, failures, _ = p.integrity_stats()
    partial = p.stream_stats(0)[2]
    # ... and the END of an input (nvx_finish): from reset, eight streams fed three frames and a ragged tail each (a different
    # length per stream, none a multiple of anything), ended in ONE launch at their true lengths -- the bits of both chains
    # are exactly the oracle's on the same samples: no padding decoded, nothing withheld (receiver/capt_sched.c:509-513 stops
    # with its last sample)
    p.reset()
    n_tail, ok_tail, tails = min(8, n_streams), True, []
    for s in range(n_tail):
        n_s = min(n_per, 3 * nv.FRAME_RAW + 2240 * (9 + 31 * s) + 17 * s + 3)      # (the bit timing is primed after 582 samples at 900 S/s: two frames)
        tails.append(n_s)
        p.push(s, host[s, :n_s])
    p.finish()
    for s in range(n_tail):
        ref = ob.Pipe(chain_mask=3, charlayer=False)
        ref.push_raw(host[s, : tails[s] // 8 * 8])
        ok_tail = ok_tail and p.bits(s, 0) == ref.bits(0) and p.bits(s, 1) == ref.bits(1) and len(ref.bits(0)) > 0
    p.close()
    ok = ok_first and ok_last and ok_tail
    n = passes * n_streams * n_per
    return {"what": f"HOST-FED: {n_streams} streams x 2.016 MS/s pushed from host memory by {n_thr} threads, {fpp} frames at a time (nvx_push_iq -> pinned staging -> "
                    f"hipMemcpyAsync -> kernels -> bits -> character layer), both chains of every stream decoded, {passes} passes over {n_fr} frames; PCIe-inclusive, never `value`",
            "value": round(n / el / 1e6, 1), "unit": "Msamples/s", "h2d_inclusive_gbs": round(4 * n / el / 1e9, 2),
            "x_real_time": round(n / el / nv.RATE_RAW, 1), "x_real_time_per_stream": round(n / el / nv.RATE_RAW / n_streams, 1),
            "seconds": round(el, 3), "pusher_threads": n_thr, "partial_launches": int(partial),
            "handoff": {"stale_detected": stale, "launches_failed_integrity": failures},
            "parity": ok, "parity_streams_checked": n_streams, "parity_chains_checked": 2 * n_streams,
            "parity_after_timed": ok_last, "parity_after_timed_passes": 1 + passes,
            "end_of_stream_parity": ok_tail, "end_of_stream_lengths": tails,
            "parity_note": "both chains of every stream == oracle after the first pass (from reset) AND after the last (everything decoded over "
                           f"{1 + passes} passes over the same frames, state carried)"}


def leg_live_latency(nv, ob, signals, device, seconds=8.0):
    """The live path's latency (the loop it replaces decodes synchronously per sample and calls add_message inline:
    receiver/capt_sched.c:484-528 with its 50 ms poll, receiver/nav_b_sm.C:87).  Two capture rings -- one handle fed at
    252 kS/s as the SDRplay callback delivers it, one at the ADC rate 2.016 MS/s -- each fed by a fake-SDR thread AT THE
    REAL RATE with jittered packet sizes (tests/fake_sdr.py), both at once.  Latency of frame k = (bits of frame k pollable
    and its messages delivered) - (entry of the callback that carried frame k's last sample), booked inside the library
    (nvx_capture_latency).  dropped must be 0 and the bits must equal the oracle's."""
    from fake_sdr import FakeSdr
    n_frames = max(4, int(seconds / 0.32))
    legs, threads = {}, []
    for name, raw in (("252k", False), ("2016k", True)):
        rate, frame = (nv.RATE_RAW, nv.FRAME_RAW) if raw else (nv.RATE_IN, nv.FRAME_IN)
        st, _ = signals.stream_params(nv, 31000 + int(raw), rate, n_phasing=20)
        iq = nv.synth_host(st, rate, n_frames * frame)
        # both chains, the reference's own wiring (receiver/nav_sched.C:10-17) -- which also keeps these small launches out of the
        # rocprofv3 statistics of the headline's and Variant A's kernels (they run nvx_fir_cascade<..., 2>)
        p = nv.Pipeline(n_streams=1, raw_rate=raw, chain_mask=nv.CHAIN_518 | nv.CHAIN_490, max_frames=2, push_mode=True, char_layer=True, device=device)
        cap = nv.Capture(p, 0, ring_seconds=2.0)
        sdr = FakeSdr(cap, iq, rate, frame, seed=5 + int(raw), packet=(1000, 1700) if raw else (150, 420))
        legs[name] = (p, cap, sdr, iq, raw)
    for _p, _c, sdr, _iq, _r in legs.values():
        sdr.start()
    for _p, _c, sdr, _iq, _r in legs.values():
        sdr.join()
    time.sleep(0.12)                                 # the last frame's collect: at most two polls of the consumer (50 ms each)
    out, ok_all = {}, True
    for name, (p, cap, sdr, iq, raw) in legs.items():
        lat = cap.latency()
        received, dropped, consumed = cap.stats()
        cap.stop()
        ref = ob.Pipe(chain_mask=3, charlayer=False)
        (ref.push_raw if raw else ref.push)(iq)
        same = p.bits(0, 0) == ref.bits(0) and p.bits(0, 1) == ref.bits(1)
        # parity of this leg is about BITS: everything the ring took reached the decoder and decoded like the oracle (a late
        # fake-SDR thread or a missing latency sample on a loaded host is visible in the figures below, not a parity failure)
        ok = same and len(ref.bits(0)) > 100 and dropped == 0
        ok_all = ok_all and ok
        out[name] = {"frames_booked": lat["frames"], "p50_ms": round(lat["p50_ms"], 2), "p99_ms": round(lat["p99_ms"], 2), "max_ms": round(lat["max_ms"], 2),
                     "dropped": dropped, "received": received, "bits_equal_oracle": bool(same), "bits": len(ref.bits(0)),
                     "messages": len(p.messages), "fake_sdr_behind_schedule_ms_max": round(sdr.late_ms, 2),
                     "callbacks_per_s": round(sdr.packets / (n_frames * 0.32), 0)}
        p.close()
    return {"what": f"LIVE PATH LATENCY: two capture rings (nvx_capture_callback -> ring -> consumer -> nvx_push_iq -> launch -> nvx_poll), one handle each, fed "
                    f"at the real rate for {n_frames * 0.32:.1f} s of signal by fake-SDR threads with jittered packet sizes, both at once; latency of a frame = bits "
                    "pollable and messages delivered - entry of the callback that carried its last sample (booked by the library: nvx_capture_latency)",
            "streams": out, "frame_seconds": 0.32,
            "bound_for_a_character_ms": "320 (its frame still filling) + the figures above (launch + collect; 50 ms at worst when no callback wakes the consumer)",
            "parity": ok_all}


def wideband_streams(nv, signals, rank, W, n_phasing=40):
    """W wideband streams: a carrier at k*252 kHz +-14 kHz for k = 0..7, each with its own text."""
    out = []
    for w in range(W):
        gid = rank * W + w
        carriers = []
        for k in range(8):
            centre = k * 252000 if k < 4 else (k - 8) * 252000
            for c, off in ((0, 14000), (1, -14000)):
                cid = gid * 16 + 2 * k + c
                h = signals.mix32(signals.GLOBAL_SEED ^ signals.mix32(cid + 0x10000))
                carriers.append(dict(freq_hz=centre + off, bits=nv.sitor_encode(signals.stream_text(cid), n_phasing),
                                     bit_offset=(signals.mix32(h ^ 0xA5A5A5A5) % 20160) | 1, phase0=signals.mix32(h ^ 0x3C3C3C3C),
                                     amplitude=1700))
        out.append(nv.make_stream(carriers, seed=signals.mix32(gid + 77), noise_amp=600))
    return out


class Ranks:
    """The few collectives the benchmark needs: barrier, max / min / sum of a number over the ranks."""

    def __init__(self, torch, dist, device, backend):
        self.torch, self.dist, self.device, self.backend = torch, dist, device, backend

    def sync(self):
        if self.device is not None:                  # None: CPU-only test of the orchestration
            self.torch.cuda.synchronize(self.device)
        if self.dist is not None:
            self.dist.barrier()
            if self.device is not None:
                self.torch.cuda.synchronize(self.device)

    def gather(self, value: float) -> list:
        """`value` of every rank, in rank order."""
        if self.dist is None:
            return [value]
        dev = f"cuda:{self.device}" if self.backend == "nccl" else "cpu"
        t = self.torch.tensor([value], dtype=self.torch.float64, device=dev)
        out = [self.torch.zeros_like(t) for _ in range(self.dist.get_world_size())]
        self.dist.all_gather(out, t)
        return [float(o.item()) for o in out]

    def describe(self, rank_ms: float, rank_checked: int, rank_casc_ms: float, device: int) -> dict:
        """What the job looked like from the ranks: the world size the backend actually formed, stragglers, who checked what."""
        ms = self.gather(rank_ms)
        return {"world_size_seen": self.dist.get_world_size() if self.dist is not None else 1,
                "backend": ("rccl" if self.backend == "nccl" else self.backend) if self.dist is not None else "none",
                "ms_per_step_per_rank": [round(v, 3) for v in ms], "ms_per_step_min": round(min(ms), 3), "ms_per_step_max": round(max(ms), 3),
                "parity_streams_checked_per_rank": [int(v) for v in self.gather(float(rank_checked))],
                "cascade_avg_launch_ms_per_rank": [round(v, 3) for v in self.gather(rank_casc_ms)],
                "device_per_rank": [int(v) for v in self.gather(float(device))]}

    def reduce(self, value: float, op: str) -> float:
        if self.dist is None:
            return value
        t = self.torch.tensor([value], dtype=self.torch.float64, device=f"cuda:{self.device}" if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op={"max": self.dist.ReduceOp.MAX, "min": self.dist.ReduceOp.MIN, "sum": self.dist.ReduceOp.SUM}[op])
        return float(t.item())


def finish(line, parity, ranks, rank, leg_errors=False):
    """Print the line on rank 0; a failed parity is a failed run (exit status 3) on every rank, a side leg that raised
    (named in legs_failed) one with status 4."""
    if rank == 0:
        print(json.dumps(line), flush=True)
    if ranks.dist is not None:
        ranks.dist.barrier()
        ranks.dist.destroy_process_group()
    if not parity:
        sys.exit(3)
    if leg_errors:
        print(f"bench.py: side legs failed: {line.get('legs_failed')} (the line above is complete otherwise; --allow-leg-errors to pass)", file=sys.stderr)
        sys.exit(4)


def run_wideband(args, nv, signals, ranks, rank, world, device, place):
    """Channeliser + 252 kS/s pipeline on W wideband streams per GPU."""
    import oracle_binding as ob
    W, F = args.wideband, args.frames
    n_raw, n_sub = F * nv.FRAME_RAW, F * nv.FRAME_IN
    t0 = time.time()
    raw = nv.DeviceBuffer(W * n_raw * 4, device=device)
    nv.synth_device(wideband_streams(nv, signals, rank, W), nv.RATE_RAW, n_raw, raw, n_raw)
    t_gen = time.time() - t0
    pipe = nv.Pipeline(n_streams=W, wideband=True, chain_mask=3, max_frames=F, char_layer=not args.no_charlayer, device=device,
                       bit_history=max(65536, (args.warmup + args.steps + 1) * F * 32 + 4096))

    def step():
        pipe.process_resident(raw, n_raw, 0, F)

    # ---- parity: every rank checks its own shard against the oracle chain (channeliser restatement -> 8 two-chain
    # pipelines): ALL wideband streams (16 carriers each) at N = 1, the first few otherwise
    ncpu = place["threads"]
    nw = W if (world == 1 and args.verify < 0) else min(W, max(2, (args.verify if args.verify > 0 else 32) // 16))
    step(); pipe.fetch()
    ok, sN, sample = True, 0.0, None
    for w0 in range(0, nw, 32):
        m = min(32, nw - w0)
        part = raw.download(m * n_raw * 4, offset=w0 * n_raw * 4, dtype=np.int16).reshape(m, n_raw, 2)
        secs, cpu_bits = ob.bench_wide(part, m, n_sub, ncpu, want_bits=True)
        gpu_bits = [pipe.bits(s, c) for s in range(8 * w0, 8 * (w0 + m)) for c in (0, 1)]
        ok = ok and gpu_bits == cpu_bits and all(len(b) > 0 for b in cpu_bits)
        if sample is None:
            sample, sN = part[: min(m, max(2, ncpu // 4))], secs * min(m, max(2, ncpu // 4)) / m
    if not ok:
        print(f"PARITY FAILURE (wideband, rank {rank}): GPU bits differ from the CPU oracle", file=sys.stderr)
    parity = ranks.reduce(1.0 if ok else 0.0, "min") > 0.5
    checked = int(ranks.reduce(16.0 * nw, "sum"))
    near_ties = int(ranks.reduce(float(pipe.tie_stats()[0]), "sum"))
    cpu = None
    if rank == 0 and not args.no_cpu and world == 1:
        ns = sample.shape[0]
        rep = max(1, int(5.0 / max(sN, 1e-3)))
        sN = ob.bench_wide(sample, ns, n_sub, ncpu, repeat=rep)[0]
        s1 = ob.bench_wide(sample[:1], 1, n_sub, 1, repeat=max(1, rep // 8))[0]
        cpu = {"value": round(ns * n_raw * rep / sN / 1e6, 2), "unit": "Msamples/s", "cores": ncpu, "cpu_model": cpu_model(), "kind": "port",
               "value_1thread": round(n_raw * max(1, rep // 8) / s1 / 1e6, 2),
               "sample": f"all {F} frames of the first {ns} wideband streams ({ns * n_raw / 1e6:.0f} M raw samples), processed {rep}x; "
                         f"oracle channeliser + 8 x 2-chain 252 kS/s pipes, OpenMP over streams", "seconds": round(sN, 2)}
    pipe.reset()
    for _ in range(args.warmup):
        step()
    pipe.fetch()
    pipe.enable_timing(True); pipe.kernel_time_stats(0, reset=True)
    pipe.wait_stats(reset=True)
    ranks.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    pipe.fetch()
    own_elapsed = time.perf_counter() - t0
    ranks.sync()
    elapsed = ranks.reduce(time.perf_counter() - t0, "max")
    # ---- parity AFTER the timed region, as in the headline mode: the first wideband streams' 16 carriers each, everything
    # decoded over warm-up + timed launches (channeliser halo, filter and demodulator state carried) against the oracle's replay
    nw_after = min(W, 8)
    part_after = raw.download(nw_after * n_raw * 4, dtype=np.int16).reshape(nw_after, n_raw, 2)
    _secs, want_after = ob.replay_wide(part_after, nw_after, n_sub, ncpu, args.warmup + args.steps)
    ok_after = [pipe.bits(s, c) for s in range(8 * nw_after) for c in (0, 1)] == want_after and all(len(b) > 0 for b in want_after)
    if not ok_after:
        print(f"PARITY FAILURE AFTER THE TIMED REGION (wideband, rank {rank}): GPU bits differ from the CPU oracle's replay", file=sys.stderr)
    parity_after = ranks.reduce(1.0 if ok_after else 0.0, "min") > 0.5
    parity = parity and parity_after
    stale, seal_failures, _ = pipe.integrity_stats()
    stale_all, seal_failures_all = int(ranks.reduce(float(stale), "sum")), int(ranks.reduce(float(seal_failures), "sum"))
    casc_ms, n_l = pipe.kernel_time_stats(0)
    dem_ms, _ = pipe.kernel_time_stats(1)
    w_polls, w_units, w_launches = pipe.wait_stats()
    casc_avg = casc_ms / max(n_l, 1)
    f3_avg = fir3_avg_ms(pipe, n_l)
    sub_samples = 8 * W * n_sub
    fps = flops_per_sample(2, fir3_inside=False)      # the fused kernel's waves end at FIR2: nvx_fir3 does the rest
    tops = fps * sub_samples / (casc_avg * 1e-3) / 1e12 if casc_avg else None
    line = {
        "metric": "IQ Msamples/s through FIR->FSK->bitsync", "value": round(world * W * n_raw * args.steps / elapsed / 1e6, 1),
        "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"WIDEBAND (SURVEY 8f-2, not the headline): {W} streams x 2.016 MS/s per GPU, 16 NAVTEX carriers each "
                               f"(8 sub-bands x 2 chains), {F} frames ({F * 0.32:.2f} s) resident in HBM",
                   "wide_streams_per_gpu": W, "carriers_per_gpu": 16 * W, "frames": F,
                   "parallelism": f"wideband streams sharded {world} ways, no collective"},
        "carriers_decoded": 16 * W * world,
        "carrier_equivalent_msamples_per_s": round(16 * world * W * n_raw * args.steps / elapsed / 1e6, 1),
        "roofline": {"bound": "fp64_valu", "kernel": "nvx_wideband_fused (channeliser + 8 x two-chain cascade up to FIR2; FIR3 = nvx_fir3)",
                     "fir3_avg_launch_ms": round(f3_avg, 3),
                     "achieved": round(tops, 2) if tops else None,
                     "peak": round(FP64_NOFMA_PEAK_TOPS, 1), "unit": "TFLOP/s", "frac": round(tops / FP64_NOFMA_PEAK_TOPS, 4) if tops else None,
                     "traffic": None, "flop_per_sample": round(fps, 2), "samples_per_launch": sub_samples,
                     "decomposition": wideband_decomposition(tops / FP64_NOFMA_PEAK_TOPS if tops else None, fps),
                     "avg_launch_ms": round(casc_avg, 3), "launches": int(n_l), "demod_span_ms": round(dem_ms / max(n_l, 1), 3),
                     "handoff": {"units_waited_frac": round(w_units / max(1, w_launches * W * F ), 4),
                                 "avg_polls_per_waiting_unit": round(w_polls / max(1, w_units), 1),
                                 "stale_detected": stale_all, "launches_failed_integrity": seal_failures_all},
                     "algorithmic_bytes_per_launch": W * n_raw * 4,
                     "note": "exact mul-then-add fp64 (no FMA): the roof is the fp64 issue rate at 2.4 GHz, 256 CUs x 4 SIMDs x 16 lanes; "
                             "only the cascade's fp64 operations are counted, the channeliser's integer work rides on top"},
        "form": "fused (one kernel, sub-bands stay in LDS)",
        "cpu_baseline": cpu, "parity": parity, "parity_streams_checked": checked, "parity_after_timed": parity_after,
        "parity_after_timed_streams": int(ranks.reduce(16.0 * nw_after, "sum")), "parity_after_timed_launches": args.warmup + args.steps,
        "demod": {"near_ties": near_ties},
        "host_threads": place["threads"], "placement": place, "gen_seconds": round(t_gen, 1),
        "ranks": ranks.describe(own_elapsed / args.steps * 1e3, 16 * nw, casc_avg, device),
    }
    pipe.close(); raw.free()
    finish(line, parity, ranks, rank)


def run_group(args):
    """`--gpus N --group`: ONE process, the library's own multi-GPU object (nvx_group, header section C': one handle, one host
    thread bound to the device's NUMA node and one result ring per device; streams shard one contiguous subset per
    device, no collective -- the independence it rests on: receiver/nav_b_sm.h:92-114, receiver/decoder.h:31-60) on the
    headline workload per device.  Same JSON line; `ranks.backend` = "group", per-member figures where the multi-process
    form has per-rank ones.  NVX_BENCH_GROUP_DEVICES=0,0 names the members' devices explicitly (one-GPU rehearsal)."""
    devs = [int(d) for d in os.environ["NVX_BENCH_GROUP_DEVICES"].split(",")] if os.environ.get("NVX_BENCH_GROUP_DEVICES") else list(range(args.gpus))
    n = len(devs)
    have = os.sched_getaffinity(0)
    threads = int(os.environ.get("NVX_CPU_THREADS", max(1, min(16 * n, len(have)))))
    import navtex_amd as nv
    import signals
    import oracle_binding as ob
    import fullsize
    import gc
    gc.collect(); gc.freeze()
    if nv.device_count() <= max(devs):
        raise SystemExit(f"bench.py --group: devices {devs} asked for, {nv.device_count()} present")
    S, F = args.streams, args.frames
    order = 3 if args.stage0 == "cic3" else 1
    oraw = 3 if order == 3 else True
    n_per_stream = F * nv.FRAME_RAW
    pitch = n_per_stream + args.pitch_pad
    samples_per_step = S * n_per_stream                     # per member
    bytes_per_step = samples_per_step * BYTES_PER_SAMPLE
    if order != 1:
        raise SystemExit("bench.py --group runs the headline front end (nvx_group passes cfg through; use the multi-process form for --stage0 cic3)")
    t0 = time.time()
    group = nv.Group(devs, n * S, raw_rate=True, chain_mask=nv.CHAIN_518, max_frames=F, char_layer=not args.no_charlayer,
                     host_threads=max(1, min(16, threads // n)))
    bufs = []
    for m, (dev, first, count) in enumerate(group.members):
        assert count == S
        b = nv.DeviceBuffer(S * pitch * BYTES_PER_SAMPLE, device=dev)
        nv.synth_device([signals.stream_params(nv, first + s, nv.RATE_RAW)[0] for s in range(S)], nv.RATE_RAW, n_per_stream, b, pitch)
        bufs.append(b)
    t_gen = time.time() - t0
    ptrs = [b.ptr for b in bufs]
    views = [group.member_view(m) for m in range(n)]

    def step():
        group.process_resident(ptrs, pitch, 0, F)

    # ---- parity gate, every member its own shard: first launch from reset ...
    step(); group.fetch()
    n_verify = args.verify if args.verify > 0 else (S if n == 1 else 32)
    checked_m, bad_all, verify_s = [], [], 0.0
    for m, (dev, first, count) in enumerate(group.members):
        ids = fullsize.spread(S, n_verify)
        c, bad, secs = fullsize.verify_streams(ob, bufs[m], pitch, n_per_stream, oraw, lambda s, f=first: group.bits(f + s, 0), ids, threads)
        checked_m.append(c); bad_all += [first + b for b in bad]; verify_s += secs
    group.reset()
    for _ in range(args.warmup):
        step()
    group.fetch()
    for v in views:
        v.enable_timing(True); v.kernel_time_stats(0, reset=True); v.wait_stats(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    group.fetch()                      # every member's launches done, bits on the host, characters decoded, messages delivered
    elapsed = time.perf_counter() - t0
    # ... and after the timed region: what the measured launches produced (state carried launch to launch)
    loops = args.warmup + args.steps
    after_m, bad_after, after_s = [], [], 0.0
    for m, (dev, first, count) in enumerate(group.members):
        ids = fullsize.spread(S, min(S, max(32, args.after_timed)))
        c, bad, secs = fullsize.verify_replay(ob, bufs[m], pitch, n_per_stream, oraw, lambda s, f=first: group.bits(f + s, 0), ids, threads, loops,
                                              gpu_count=lambda s, f=first: group.bit_count(f + s, 0))
        after_m.append(c); bad_after += [first + b for b in bad]; after_s += secs
    if bad_all or bad_after:
        print(f"PARITY FAILURE (group): first launch {len(bad_all)} streams differ (first {bad_all[:8]}), after the timed region {len(bad_after)} (first {bad_after[:8]})", file=sys.stderr)
    parity = not bad_all and not bad_after
    casc = []
    stale = failures = 0
    w_units = w_launches = 0
    for v in views:
        ms, nl = v.kernel_time_stats(0); casc.append(ms / max(nl, 1))
        a, b, _ = v.integrity_stats(); stale += a; failures += b
        _, wu, wl = v.wait_stats(); w_units += wu; w_launches += wl
    casc_avg = max(casc)
    achieved = bytes_per_step / (casc_avg * 1e-3) / 1e9 if casc_avg > 0 else None
    traffic, traffic_source = traffic_record(S, F, order)
    line = {
        "metric": "IQ Msamples/s through FIR->FSK->bitsync", "value": round(n * samples_per_step * args.steps / elapsed / 1e6, 1), "unit": "Msamples/s",
        "n_gpus": n, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{S} synthetic 170 Hz-shift FSK channels x 2.016 MS/s int16 IQ per GPU, {F} frames ({F * 0.32:.2f} s) resident in HBM "
                               f"(BASELINE configs[3] per device; x{n} devices behind ONE nvx_group in one process)",
                   "streams_per_gpu": S, "frames": F, "samples_per_step_per_gpu": samples_per_step, "stage0": "integrate-and-dump /8 (build-owned)",
                   "chains_per_stream": 1, "parallelism": f"nvx_group: streams sharded over {n} member handles (devices {devs}), one host thread per member, no collective"},
        "roofline": {"bound": "hbm", "kernel": "nvx_fir_cascade<raw,1>", "achieved": round(achieved, 1) if achieved else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4) if achieved else None, "traffic": traffic, "traffic_source": traffic_source,
                     "algorithmic_bytes_per_launch": bytes_per_step, "avg_launch_ms": round(casc_avg, 3), "per_device": "slowest member; ranks.cascade_avg_launch_ms_per_rank has all",
                     "handoff": {"units_waited_frac": round(w_units / max(1, w_launches * S * F), 4), "stale_detected": stale, "launches_failed_integrity": failures}},
        "cpu_baseline": None,
        "parity": parity, "parity_streams_checked": sum(checked_m), "parity_seconds": round(verify_s, 1),
        "parity_after_timed": not bad_after, "parity_after_timed_streams": sum(after_m), "parity_after_timed_launches": loops, "parity_after_timed_seconds": round(after_s, 1),
        "host_threads": threads,
        "ranks": {"world_size_seen": n, "backend": "group", "note": "one process; a 'rank' here is a member handle of the nvx_group (its own device, host thread and result ring)",
                  "ms_per_step_per_rank": [round(elapsed / args.steps * 1e3, 3)] * n, "ms_per_step_min": round(elapsed / args.steps * 1e3, 3),
                  "ms_per_step_max": round(elapsed / args.steps * 1e3, 3),
                  "parity_streams_checked_per_rank": checked_m, "parity_after_timed_streams_per_rank": after_m,
                  "cascade_avg_launch_ms_per_rank": [round(c, 3) for c in casc], "device_per_rank": [d for d, _, _ in group.members],
                  "first_stream_per_rank": [f for _, f, _ in group.members]},
        "hbm_gbs_whole_job": round(n * bytes_per_step * args.steps / elapsed / 1e9, 1), "gen_seconds": round(t_gen, 1),
        "messages_delivered": len(group.messages),
    }
    group.close()
    for b in bufs:
        b.free()
    print(json.dumps(line), flush=True)
    if not parity:
        sys.exit(3)


def main():
    args = parse()
    if args.group:
        if args.verify == 0:
            raise SystemExit("--verify 0: an unchecked number is no number")
        return run_group(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if "RANK" not in os.environ and args.gpus > 1:
        self_launch(args)                        # never returns: relays the child's line and exit status
    if world != args.gpus:
        args.gpus = world
    if args.verify == 0:
        raise SystemExit("--verify 0: an unchecked number is no number")

    # NVX_BENCH_BACKEND=gloo / NVX_BENCH_DEVICE=<n>: rehearsal of the multi-process path on a box
    # with fewer GPUs than ranks (RCCL refuses two ranks on one device); never used by the driver.
    backend = os.environ.get("NVX_BENCH_BACKEND", "nccl")
    device = int(os.environ.get("NVX_BENCH_DEVICE", local))
    # host placement first: the affinity is inherited by every thread the HIP runtime, RCCL and the library start
    place = place_rank(device, local_world, local)

    import torch
    # Importing torch leaves millions of long-lived Python objects behind; a full garbage collection then takes ~40 ms, and
    # the message callbacks of the timed loop (tuples, strings) are what triggers one (seen as a 42 ms step every ~30
    # steps at 64 streams x 62 frames).  Park what exists now in the permanent generation: collections stay cheap.
    import gc
    gc.collect(); gc.freeze()
    dist = None
    # NVX_BENCH_FORCE_DIST=1: form the process group even for one rank (a one-GPU rehearsal of the RCCL calls the
    # multi-GPU run makes: init, barrier, all_reduce, all_gather on device tensors)
    if world > 1 or (os.environ.get("NVX_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(device)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    # fresh checkout: rank 0 builds (hipcc, gfx950; linked beside the target and renamed), everybody else waits at the
    # barrier -- which every rank reaches whatever it sees on disk, so the collectives stay paired
    if rank == 0 and not (ROOT / "navtex_amd" / "libnavtex_amd.so").exists():
        import contextlib, importlib.util
        spec = importlib.util.spec_from_file_location("nvx_build", ROOT / "navtex_amd" / "build.py")
        mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
        with contextlib.redirect_stdout(sys.stderr):       # stdout carries exactly one JSON line
            mod.build_lib()
    if dist is not None:
        dist.barrier()
    import navtex_amd as nv
    import signals

    if nv.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X: navtex_amd has no CPU path")
    ranks = Ranks(torch, dist, device, backend)
    if args.wideband:
        run_wideband(args, nv, signals, ranks, rank, world, device, place)
        return
    import oracle_binding as ob
    import fullsize
    S, F = args.streams, args.frames
    raw = not args.variant_a
    order = 3 if (raw and args.stage0 == "cic3") else 1
    oraw = (3 if order == 3 else True) if raw else False          # what the oracle is told: its raw flag carries the order
    RATE, FRAME = (nv.RATE_RAW, nv.FRAME_RAW) if raw else (nv.RATE_IN, nv.FRAME_IN)
    n_per_stream = F * FRAME
    pitch = n_per_stream + args.pitch_pad
    samples_per_step = S * n_per_stream
    bytes_per_step = samples_per_step * BYTES_PER_SAMPLE

    # ---- synthetic input, generated on the device, resident in HBM ----------
    t0 = time.time()
    streams = []
    for s in range(S):
        gid = rank * S + s                                     # global stream id: subsets per GPU
        st, _ = signals.stream_params(nv, gid, RATE)
        streams.append(st)
    buf = nv.DeviceBuffer(S * pitch * BYTES_PER_SAMPLE, device=device)
    nv.synth_device(streams, RATE, n_per_stream, buf, pitch)
    t_gen = time.time() - t0

    # (the handle keeps a bounded poll history per chain -- a receiver runs for weeks --; the gate behind the timed region wants
    # everything decoded over warm-up + timed launches)
    pipe = nv.Pipeline(n_streams=S, raw_rate=raw, chain_mask=nv.CHAIN_518, max_frames=F,
                       char_layer=not args.no_charlayer, device=device, stage0_order=order,
                       bit_history=max(65536, (args.warmup + args.steps + 1) * F * 32 + 4096))
    ncpu = place["threads"]

    # ---- parity gate: EVERY rank checks its own shard against the oracle, from reset state -----------------------
    pipe.process_resident(buf, pitch, 0, F)
    pipe.fetch()
    n_verify = args.verify if args.verify > 0 else (S if world == 1 else 32)
    ids = fullsize.spread(S, n_verify)
    checked, bad, verify_s = fullsize.verify_streams(ob, buf, pitch, n_per_stream, oraw, lambda s: pipe.bits(s, 0), ids, ncpu)
    if bad:
        print(f"PARITY FAILURE (rank {rank}): {len(bad)} of {checked} streams differ from the CPU oracle, first {bad[:8]}", file=sys.stderr)
    parity = ranks.reduce(0.0 if bad else 1.0, "min") > 0.5
    checked_all = int(ranks.reduce(float(checked), "sum"))
    near, evals, margin = pipe.tie_stats()
    near_all = int(ranks.reduce(float(near), "sum"))
    margin_all = ranks.reduce(margin if evals else 1.0, "min")

    # ---- CPU baseline (rank 0, N = 1 leg of the contract; BASELINE.md 3: one thread AND all physical cores) -------
    cpu = None
    if rank == 0 and not args.no_cpu and world == 1:
        cpu = cpu_baseline_leg(ob, buf, pitch, n_per_stream, F, S, oraw, ncpu, args, nv)
        cpu["reference_check"] = reference_check(ob, buf.download(n_per_stream * 4, dtype=np.int16).reshape(-1, 2), order) if raw else None
    pipe.reset()

    # ---- warm-up, then EXACTLY K timed steps ------------------------------------
    def step():
        pipe.process_resident(buf, pitch, 0, F)

    for _ in range(args.warmup):
        step()
    pipe.fetch()
    pipe.enable_timing(True)
    pipe.kernel_time_stats(0, reset=True)
    pipe.wait_stats(reset=True)
    ranks.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    pipe.fetch()                       # all launches done, bits on the host, characters decoded
    own_elapsed = time.perf_counter() - t0          # this rank's own K steps (before the closing barrier): stragglers show
    ranks.sync()
    elapsed = ranks.reduce(time.perf_counter() - t0, "max")

    casc_ms, n_l = pipe.kernel_time_stats(0)
    dem_ms, _ = pipe.kernel_time_stats(1)
    w_polls, w_units, w_launches = pipe.wait_stats()
    total_bits = sum(pipe.bit_count(s, 0) for s in range(0, S, max(1, S // 64)))

    # ---- parity AFTER the timed region: what the measured launches produced.  The handle was reset before the warm-up;
    # since then it has run the same F frames (warmup + steps) times, its filter and demodulator state carried from launch
    # to launch (the reference's statics: receiver/fir1cpp.C:51-60, fir2cpp.C:74-83, fir3cpp.h:90-95, decoder.h:31-60) through
    # ~(warmup + steps) x S x F unit hand-overs.  The oracle is fed the same frames as many times; every bit must agree.
    loops = args.warmup + args.steps
    ids_after = fullsize.spread(S, min(S, max(32, args.after_timed)))
    checked_after, bad_after, after_s = fullsize.verify_replay(ob, buf, pitch, n_per_stream, oraw, lambda s: pipe.bits(s, 0), ids_after, ncpu, loops,
                                                                   gpu_count=lambda s: pipe.bit_count(s, 0))
    if bad_after:
        print(f"PARITY FAILURE AFTER THE TIMED REGION (rank {rank}): {len(bad_after)} of {checked_after} streams differ from the CPU oracle "
              f"after {loops} launches, first {bad_after[:8]}", file=sys.stderr)
    parity_after = ranks.reduce(0.0 if bad_after else 1.0, "min") > 0.5
    checked_after_all = int(ranks.reduce(float(checked_after), "sum"))
    parity = parity and parity_after
    stale, seal_failures, _ = pipe.integrity_stats()
    stale_all = int(ranks.reduce(float(stale), "sum"))
    seal_failures_all = int(ranks.reduce(float(seal_failures), "sum"))

    ms_per_step = elapsed / args.steps * 1e3
    value = world * samples_per_step * args.steps / elapsed / 1e6
    casc_avg = casc_ms / max(n_l, 1)
    handoff = {"units_waited_frac": round(w_units / max(1, w_launches * S * F), 4),
               "avg_polls_per_waiting_unit": round(w_polls / max(1, w_units), 1),
               # the state blocks' seals (nvx_cascade_integrity_stats), all launches of this handle since create, all ranks:
               # hand-overs that failed the check and were repaired by a pre-roll / launches whose inherited state failed it
               "stale_detected": stale_all, "launches_failed_integrity": seal_failures_all,
               "hand_overs_checked_about": int(w_launches) * S * (F + 1)}       # per stream and launch: F - 1 whole frames + 3 thirds take over from a predecessor
    if raw:
        achieved = bytes_per_step / (casc_avg * 1e-3) / 1e9 if casc_avg > 0 else None
        traffic, traffic_source = traffic_record(S, F, order)
        roofline = {"bound": "hbm", "kernel": "nvx_fir_cascade<raw,1>" if order == 1 else "nvx_fir_cascade_cic3_1", "achieved": round(achieved, 1) if achieved else None,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4) if achieved else None,
                    "traffic": traffic, "traffic_source": traffic_source, "algorithmic_bytes_per_launch": bytes_per_step,
                    "avg_launch_ms": round(casc_avg, 3), "launches": int(n_l), "demod_span_ms": round(dem_ms / max(n_l, 1), 3),
                    # hand-over of filter state between the frames of a stream: share of units that had to
                    # wait for their predecessor and the average wait of those (one poll ~ 1 us)
                    "handoff": handoff}
    else:
        tops = flops_per_sample(1) * samples_per_step / (casc_avg * 1e-3) / 1e12 if casc_avg > 0 else None
        roofline = {"bound": "fp64_valu", "kernel": "nvx_fir_cascade<252k,1>", "achieved": round(tops, 2) if tops else None,
                    "peak": round(FP64_NOFMA_PEAK_TOPS, 1), "unit": "TFLOP/s", "frac": round(tops / FP64_NOFMA_PEAK_TOPS, 4) if tops else None,
                    "traffic": None, "flop_per_sample": round(flops_per_sample(1), 2), "samples_per_launch": samples_per_step,
                    "avg_launch_ms": round(casc_avg, 3), "launches": int(n_l), "demod_span_ms": round(dem_ms / max(n_l, 1), 3),
                    "hbm_gbs": round(bytes_per_step / (casc_avg * 1e-3) / 1e9, 1) if casc_avg > 0 else None, "handoff": handoff,
                    "note": "exact mul-then-add fp64 (no FMA): the roof is the fp64 issue rate at 2.4 GHz, 256 CUs x 4 SIMDs x 16 lanes"}
    line = {
        "metric": "IQ Msamples/s through FIR->FSK->bitsync", "value": round(value, 1), "unit": "Msamples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": (f"{S} synthetic 170 Hz-shift FSK channels x 2.016 MS/s int16 IQ per GPU, "
                                f"{F} frames ({F * 0.32:.2f} s) resident in HBM (BASELINE configs[3]; x{world} GPUs = configs[4] shape)") if raw else
                               (f"VARIANT A (not the headline): {S} channels x 252 kS/s int16 IQ per GPU, {F} frames ({F * 0.32:.2f} s), "
                                f"no stage 0 -- fp64-issue-bound by design (SURVEY 7-2)"),
                   "streams_per_gpu": S, "frames": F, "samples_per_step_per_gpu": samples_per_step,
                   "stage0": ("none" if not raw else "integrate-and-dump /8 (build-owned)" if order == 1 else
                              "three cascaded 8-sample boxcars /8 = 22-tap CIC^3 (build-owned; NOT the headline front end)"),
                   "chains_per_stream": 1,
                   "parallelism": f"streams sharded {world} ways, no collective"},
        "roofline": roofline,
        "cpu_baseline": cpu,
        "parity": parity, "parity_streams_checked": checked_all, "parity_seconds": round(verify_s, 1),
        "parity_after_timed": parity_after, "parity_after_timed_streams": checked_after_all, "parity_after_timed_launches": loops,
        "parity_after_timed_seconds": round(after_s, 1),
        "parity_note": "parity = first launch from reset (every stream at N = 1) AND parity_after_timed: the bits accumulated over the warm-up and "
                       "the timed launches on a spread sample of streams == the oracle fed the same frames as many times (state carried throughout)",
        "demod": {"near_ties": near_all, "min_relative_margin": margin_all, "timing_evaluations_rank0": int(evals),
                  "span_note": "roofline.demod_span_ms is first-to-last event of the demodulator of launch k, which runs BESIDE the "
                               "cascade of launch k+1 (second stream) and becomes resident as CUs have room: alone it takes ~0.85 ms"},
        "host_threads": place["threads"], "placement": place,
        "ranks": ranks.describe(own_elapsed / args.steps * 1e3, checked, casc_avg, device),
        "hbm_gbs_whole_job": round(world * bytes_per_step * args.steps / elapsed / 1e9, 1),
        "gen_seconds": round(t_gen, 1), "bits_sampled": int(total_bits),
    }
    pipe.close()
    legs_failed = []

    def run_leg(name, fn):
        """A side leg never takes the headline's line down with it -- but it fails the RUN: wrong bits anywhere end the
        process with status 3, a leg that raises is named in legs_failed and ends it with status 4 (--allow-leg-errors: 0)."""
        nonlocal parity
        t_leg = time.perf_counter()
        try:
            rec = fn()
        except Exception as e:
            import traceback
            traceback.print_exc(file=sys.stderr)
            rec = {"error": f"{type(e).__name__}: {e}"[:300]}
            legs_failed.append(name)
        rec["leg_seconds"] = round(time.perf_counter() - t_leg, 1)
        line[name] = rec
        if rec.get("parity") is False:
            print(f"PARITY FAILURE ({name} leg): GPU bits differ from the CPU oracle", file=sys.stderr)
            parity = False; line["parity"] = False

    # ---- beside the headline (N = 1, default front end only): the same batch through the third-order stage 0, the front
    # end with real alias rejection -- on the same footing as the headline: every stream against the
    # oracle from reset, a spread sample after its timed launches, its own PMC traffic record
    if raw and order == 1 and world == 1 and not args.no_stage0_extra:
        run_leg("stage0_third_order", lambda: leg_stage0_cic3(nv, ob, fullsize, buf, pitch, n_per_stream, S, F, device, ncpu, not args.no_charlayer,
                                                              samples_per_step, bytes_per_step, args.cic3_verify, max(32, args.after_timed)))
    # ---- side legs (N = 1, default workload only): the streaming path and the other kernel families, driver-timed ----
    if raw and order == 1 and world == 1 and not args.no_legs:
        run_leg("push_path", lambda: leg_push_path(nv, ob, buf, pitch, F, device, ncpu, n_streams=min(64, S), pushers=min(4, ncpu)))
        buf.free()                                       # room for the 252 kS/s batch of the same size
        run_leg("live_latency", lambda: leg_live_latency(nv, ob, signals, device))
        run_leg("variant_a", lambda: leg_variant_a(nv, ob, fullsize, signals, S, device, ncpu, not args.no_charlayer,
                                                   frames=8 * F if S * 8 * F * nv.FRAME_IN * 4 <= (140 << 30) else F))
        run_leg("wideband", lambda: leg_wideband(nv, ob, signals, max(1, S // 8), F, device, ncpu, not args.no_charlayer))
    else:
        buf.free()
    line["legs_failed"] = legs_failed
    finish(line, parity, ranks, rank, leg_errors=bool(legs_failed) and not args.allow_leg_errors)


if __name__ == "__main__":
    main()
