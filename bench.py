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
sys.path.insert(0, str(ROOT / "tools"))

# The parts of this script live in tools/benchlib/, one module per concern; their names are re-exported here (tests and the
# tools/gpu_scripts reach them as bench.X).  What stays in this file is the run itself: arguments, the headline workload and
# its timed region, the wideband and nvx_group modes, the assembly of the ONE JSON line.
from benchlib.roofline import (BYTES_PER_SAMPLE, FLOP_FIR1, FLOP_FIR3, FLOP_PER_CHAIN, FP64_NOFMA_PEAK_TOPS, HBM_PEAK_GBS, KERNEL_SOURCES,  # noqa: E402,F401
                               WB_CHANNELISER_VALU_PER_LANE, WB_INT_OPS_PER_RAW_SAMPLE, WB_SHARE_BARRIERS, WB_SHARE_CASCADE, WB_SHARE_CHANNELISER,
                               flops_per_sample, kernel_source_hash, traffic_record, wideband_decomposition)
from benchlib.placement import _gpu_numa, _parse_cpulist, cpu_model, cpu_quota, physical_cores, place_rank  # noqa: E402,F401
from benchlib.launcher import Ranks, finish, self_launch  # noqa: E402,F401
from benchlib.cpu import cpu_baseline_leg, reference_check  # noqa: E402,F401
from benchlib.legs import (fir3_avg_ms, leg_live_latency, leg_push_path, leg_stage0_cic3, leg_variant_a, leg_wideband,  # noqa: E402,F401
                           wideband_streams)


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
