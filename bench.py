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

Every rank checks its OWN shard against the oracle before the timed region (all
streams at N = 1, a spread sample of them per rank otherwise); the line's
`parity` is the minimum over ranks, and a failed parity makes the process exit
with status 3 after printing the line.

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
FLOP_FIR1, FLOP_PER_CHAIN = 37.0, 6 / 4 + 47 * 4 / 28 + 71 * 4 / 280


def flops_per_sample(chains: int) -> float:
    return FLOP_FIR1 + chains * FLOP_PER_CHAIN


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

    def reduce(self, value: float, op: str) -> float:
        if self.dist is None:
            return value
        t = self.torch.tensor([value], dtype=self.torch.float64, device=f"cuda:{self.device}" if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op={"max": self.dist.ReduceOp.MAX, "min": self.dist.ReduceOp.MIN, "sum": self.dist.ReduceOp.SUM}[op])
        return float(t.item())


def finish(line, parity, ranks, rank):
    """Print the line on rank 0; a failed parity is a failed run (exit status 3) on every rank."""
    if rank == 0:
        print(json.dumps(line), flush=True)
    if ranks.dist is not None:
        ranks.dist.barrier()
        ranks.dist.destroy_process_group()
    if not parity:
        sys.exit(3)


def run_wideband(args, nv, signals, ranks, rank, world, device, place):
    """Channeliser + 252 kS/s pipeline on W wideband streams per GPU."""
    import oracle_binding as ob
    W, F = args.wideband, args.frames
    n_raw, n_sub = F * nv.FRAME_RAW, F * nv.FRAME_IN
    t0 = time.time()
    raw = nv.DeviceBuffer(W * n_raw * 4, device=device)
    nv.synth_device(wideband_streams(nv, signals, rank, W), nv.RATE_RAW, n_raw, raw, n_raw)
    t_gen = time.time() - t0
    pipe = nv.Pipeline(n_streams=W, wideband=True, chain_mask=3, max_frames=F, char_layer=not args.no_charlayer, device=device)

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
    nv.lib.nvx_channelise_timing(1); nv.channelise_time_stats(reset=True)
    ranks.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    pipe.fetch()
    ranks.sync()
    elapsed = ranks.reduce(time.perf_counter() - t0, "max")
    casc_ms, n_l = pipe.kernel_time_stats(0)
    dem_ms, _ = pipe.kernel_time_stats(1)
    ch_ms, n_c = nv.channelise_time_stats()
    w_polls, w_units, w_launches = pipe.wait_stats()
    casc_avg, ch_avg = casc_ms / max(n_l, 1), ch_ms / max(n_c, 1)
    fused = n_c == 0                                # the fused kernel has no separate channeliser launch
    sub_samples = 8 * W * n_sub
    tops = flops_per_sample(2) * sub_samples / (casc_avg * 1e-3) / 1e12 if casc_avg else None
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
        "roofline": {"bound": "fp64_valu", "kernel": "nvx_wideband_fused (channeliser + 8 x two-chain cascade)" if fused else "nvx_fir_cascade<252k,2>",
                     "achieved": round(tops, 2) if tops else None,
                     "peak": round(FP64_NOFMA_PEAK_TOPS, 1), "unit": "TFLOP/s", "frac": round(tops / FP64_NOFMA_PEAK_TOPS, 4) if tops else None,
                     "traffic": None, "flop_per_sample": round(flops_per_sample(2), 2), "samples_per_launch": sub_samples,
                     "avg_launch_ms": round(casc_avg, 3), "launches": int(n_l), "demod_span_ms": round(dem_ms / max(n_l, 1), 3),
                     "handoff": {"units_waited_frac": round(w_units / max(1, w_launches * W * F * (1 if n_c == 0 else 8)), 4),
                                 "avg_polls_per_waiting_unit": round(w_polls / max(1, w_units), 1)},
                     "algorithmic_bytes_per_launch": W * n_raw * 4,
                     "note": "exact mul-then-add fp64 (no FMA): the roof is the fp64 issue rate at 2.4 GHz, 256 CUs x 4 SIMDs x 16 lanes; "
                             "only the cascade's fp64 operations are counted, the channeliser's integer work rides on top"},
        "channeliser": None if fused else {"kernel": "nvx_channelise", "avg_launch_ms": round(ch_avg, 3), "launches": int(n_c),
                        "read_plus_write_gbs": round((W * n_raw * 4 + sub_samples * 4) / (ch_avg * 1e-3) / 1e9, 1) if ch_avg else None},
        "form": "fused (one kernel, sub-bands stay in LDS)" if fused else "two kernels (NVX_WB_FUSED=0)",
        "cpu_baseline": cpu, "parity": parity, "parity_streams_checked": checked, "demod": {"near_ties": near_ties},
        "host_threads": place["threads"], "placement": place, "gen_seconds": round(t_gen, 1),
    }
    pipe.close(); raw.free()
    finish(line, parity, ranks, rank)


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one process per GPU)")
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
    dist = None
    if world > 1:
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

    pipe = nv.Pipeline(n_streams=S, raw_rate=raw, chain_mask=nv.CHAIN_518, max_frames=F,
                       char_layer=not args.no_charlayer, device=device, stage0_order=order)
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

    # ---- CPU baseline (rank 0, N = 1 leg of the contract) ---------------------------------------------------------
    cpu = None
    if rank == 0 and not args.no_cpu and world == 1:
        n_cs = min(args.cpu_streams or 2 * ncpu, S)
        sample = np.empty((n_cs, n_per_stream, 2), dtype=np.int16)
        for s in range(n_cs):
            sample[s] = buf.download(n_per_stream * 4, offset=s * pitch * 4, dtype=np.int16).reshape(-1, 2)
        n252 = F * nv.FRAME_IN
        per_pass = n_cs * n_per_stream
        # calibrate, then size the repeat count for ~6 s wall on all threads
        sN = ob.bench(sample, n_cs, n252, oraw, 1, ncpu)[0]
        rep = max(1, int(6.0 / max(sN, 1e-3)))
        sN = ob.bench(sample, n_cs, n252, oraw, 1, ncpu, repeat=rep)[0]
        one = min(n_cs, 2)
        s1 = ob.bench(sample[:one], one, n252, oraw, 1, 1, repeat=max(1, rep // 8))[0]
        cpu = {
            "value": round(per_pass * rep / sN / 1e6, 2), "unit": "Msamples/s",
            "cores": ncpu, "cpu_model": cpu_model(), "kind": "port",
            "value_1thread": round(one * n_per_stream * max(1, rep // 8) / s1 / 1e6, 2),
            "sample": f"all {F} frames of the first {n_cs} streams of the bench batch ({per_pass / 1e6:.0f} M raw samples), "
                      f"processed {rep}x; oracle/nvx_oracle.c (gcc -O2 -ffp-contract=off), OpenMP over streams",
            "seconds": round(sN, 2),
        }
        cpu["reference_check"] = reference_check(ob, sample[0], order) if raw else None
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
    ranks.sync()
    elapsed = ranks.reduce(time.perf_counter() - t0, "max")

    casc_ms, n_l = pipe.kernel_time_stats(0)
    dem_ms, _ = pipe.kernel_time_stats(1)
    w_polls, w_units, w_launches = pipe.wait_stats()
    total_bits = sum(pipe.bit_count(s, 0) for s in range(0, S, max(1, S // 64)))

    ms_per_step = elapsed / args.steps * 1e3
    value = world * samples_per_step * args.steps / elapsed / 1e6
    casc_avg = casc_ms / max(n_l, 1)
    handoff = {"units_waited_frac": round(w_units / max(1, w_launches * S * F), 4),
               "avg_polls_per_waiting_unit": round(w_polls / max(1, w_units), 1)}
    if raw:
        achieved = bytes_per_step / (casc_avg * 1e-3) / 1e9 if casc_avg > 0 else None
        traffic, traffic_source = None, None
        tf = ROOT / "profiles" / "hbm_traffic.json"
        if tf.exists():
            try:
                rec = json.loads(tf.read_text())
                if rec.get("streams") == S and rec.get("frames") == F and order == 1:
                    traffic = rec.get("bytes_per_launch")
                    traffic_source = "profiles/hbm_traffic.json (static: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload, not measured by this run)"
            except Exception:
                traffic = None
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
        "demod": {"near_ties": near_all, "min_relative_margin": margin_all, "timing_evaluations_rank0": int(evals),
                  "span_note": "roofline.demod_span_ms is first-to-last event of the demodulator of launch k, which runs BESIDE the "
                               "cascade of launch k+1 (second stream) and becomes resident as CUs have room: alone it takes ~0.85 ms"},
        "host_threads": place["threads"], "placement": place,
        "hbm_gbs_whole_job": round(world * bytes_per_step * args.steps / elapsed / 1e9, 1),
        "gen_seconds": round(t_gen, 1), "bits_sampled": int(total_bits),
    }
    pipe.close()
    # ---- beside the headline (N = 1, default front end only): the same batch through the third-order stage 0, the front
    # end with real alias rejection -- its own parity sample, ten timed steps, outside the headline's clock
    if raw and order == 1 and world == 1 and not args.no_stage0_extra:
        try:
            p3 = nv.Pipeline(n_streams=S, raw_rate=True, chain_mask=nv.CHAIN_518, max_frames=F,
                             char_layer=not args.no_charlayer, device=device, stage0_order=3)
            p3.process_resident(buf, pitch, 0, F); p3.fetch()
            ids3 = fullsize.spread(S, min(S, 64))
            checked3, bad3, _ = fullsize.verify_streams(ob, buf, pitch, n_per_stream, 3, lambda s: p3.bits(s, 0), ids3, ncpu)
            p3.reset()
            for _ in range(2): p3.process_resident(buf, pitch, 0, F)
            p3.fetch(); p3.enable_timing(True); p3.kernel_time_stats(0, reset=True)
            k3 = 10
            t3 = time.perf_counter()
            for _ in range(k3): p3.process_resident(buf, pitch, 0, F)
            p3.fetch()
            e3 = time.perf_counter() - t3
            c3, n3 = p3.kernel_time_stats(0)
            c3 /= max(n3, 1)
            line["stage0_third_order"] = {
                "what": "the same batch with nvx_config.stage0_order = 3 (22-tap CIC^3, 76 dB of alias rejection at the NAVTEX "
                        "offsets where the headline's integrate-and-dump has 25); not part of the timed region above",
                "steps": k3, "ms_per_step": round(e3 / k3 * 1e3, 3), "value": round(samples_per_step * k3 / e3 / 1e6, 1),
                "cascade_avg_launch_ms": round(c3, 3), "frac_of_hbm_peak": round(bytes_per_step / (c3 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if c3 > 0 else None,
                "parity": not bad3, "parity_streams_checked": checked3}
            if bad3:
                print(f"PARITY FAILURE (third-order stage 0): {len(bad3)} of {checked3} streams differ from the CPU oracle, first {bad3[:8]}", file=sys.stderr)
                parity = False; line["parity"] = False
            p3.close()
        except nv.NvxError as e:
            line["stage0_third_order"] = {"error": str(e)}
    buf.free()
    finish(line, parity, ranks, rank)


if __name__ == "__main__":
    main()
