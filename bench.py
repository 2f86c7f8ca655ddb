#!/usr/bin/env python3
"""bench.py -- headline benchmark of the NAVTEX hot path on MI355X.

Workload (BASELINE.json configs[3], the "HBM roofline run"; configs[4] is the
same per GPU): 4096 independent synthetic 170 Hz-shift FSK channels per GPU,
each an int16 IQ stream at 2.016 MS/s resident in HBM, run through
stage 0 (/8) -> FIR1 -> mixer -> FIR2 -> FIR3 -> FSK discriminator -> bit sync
-> mark/space decision -> host SITOR-B character layer.

One "step" = one pass of that path over the whole resident batch
(streams x frames x 645120 samples).  Multi-GPU: one process per GPU
(torch.distributed / RCCL only for the barrier and the max-over-ranks of the
elapsed time); streams shard one subset per GPU, no data-path collective,
weak scaling.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--streams", type=int, default=4096, help="IQ streams per GPU")
    ap.add_argument("--frames", type=int, default=12, help="frames (0.32 s each) resident per stream")
    ap.add_argument("--cpu-streams", type=int, default=0, help="streams in the CPU-baseline sample (0 = auto)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-charlayer", action="store_true")
    ap.add_argument("--pitch-pad", type=int, default=0, help="extra complex samples between streams (multiple of 4)")
    ap.add_argument("--wideband", type=int, default=0, metavar="W",
                    help="wideband mode (SURVEY 8f rank 2, not the headline): W streams at 2.016 MS/s per GPU, each carrying "
                         "16 NAVTEX carriers; channeliser -> 8W sub-bands at 252 kS/s -> both chains")
    ap.add_argument("--variant-a", action="store_true",
                    help="reference-native input rate: streams at 252 kS/s, no stage 0 (SURVEY 8d Variant A; fp64-bound, "
                         "reported for completeness -- the headline workload is the default 2.016 MS/s Variant B)")
    return ap.parse_args()


def reference_check(ob, raw_stream):
    """Strawman guard: the reference ITSELF (oracle/_ref/ref_bits, built from /root/reference in the
    build container and shipped as a binary) against the port, one thread each, on the same 252 kS/s input
    (the stream's stage-0 output; the reference always runs both chains).  None when the binary is absent."""
    import subprocess, tempfile
    exe = ROOT / "oracle" / "_ref" / "ref_bits"
    if not exe.exists():
        return None
    try:
        iq252 = ob.stage0(raw_stream)[: 252000 * 4]
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


def run_wideband(args, nv, signals, torch, dist, rank, world, device, backend):
    """Channeliser + 252 kS/s pipeline on W wideband streams per GPU."""
    W, F = args.wideband, args.frames
    n_raw, n_sub = F * nv.FRAME_RAW, F * nv.FRAME_IN
    t0 = time.time()
    raw = nv.DeviceBuffer(W * n_raw * 4, device=device)
    sub = None
    nv.synth_device(wideband_streams(nv, signals, rank, W), nv.RATE_RAW, n_raw, raw, n_raw)
    t_gen = time.time() - t0
    # wideband handle: channeliser (own stream, double-buffered sub-bands) + 252 kS/s path; the channeliser
    # of step k+1 overlaps the cascade of step k
    pipe = nv.Pipeline(n_streams=W, wideband=True, chain_mask=3, max_frames=F, char_layer=not args.no_charlayer, device=device)
    del sub

    def step():
        pipe.process_resident(raw, n_raw, 0, F)

    def sync_all():
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier(); torch.cuda.synchronize(device)

    cpu, parity = None, None
    if rank == 0:
        import oracle_binding as ob
        ncpu = int(os.environ.get("NVX_CPU_THREADS", min(16, len(os.sched_getaffinity(0)))))
        nw = min(W, 2 if world > 1 else max(2, ncpu // 4))
        sample = raw.download(nw * n_raw * 4, dtype=np.int16).reshape(nw, n_raw, 2)
        step(); pipe.fetch()
        gpu_bits = [pipe.bits(s, c) for s in range(8 * nw) for c in (0, 1)]
        sN, cpu_bits = ob.bench_wide(sample, nw, n_sub, ncpu, want_bits=True)
        if not args.no_cpu and world == 1:
            rep = max(1, int(5.0 / max(sN, 1e-3)))
            sN = ob.bench_wide(sample, nw, n_sub, ncpu, repeat=rep)[0]
            s1 = ob.bench_wide(sample[:1], 1, n_sub, 1, repeat=max(1, rep // 8))[0]
            cpu = {"value": round(nw * n_raw * rep / sN / 1e6, 2), "unit": "Msamples/s", "cores": ncpu, "kind": "port",
                   "value_1thread": round(n_raw * max(1, rep // 8) / s1 / 1e6, 2),
                   "sample": f"all {F} frames of the first {nw} wideband streams ({nw * n_raw / 1e6:.0f} M raw samples), processed {rep}x; "
                             f"oracle channeliser + 8 x 2-chain 252 kS/s pipes, OpenMP over streams", "seconds": round(sN, 2)}
        parity = gpu_bits == cpu_bits and all(len(b) > 0 for b in cpu_bits)
        if not parity:
            print("PARITY FAILURE (wideband): GPU bits differ from the CPU oracle", file=sys.stderr)
    pipe.reset()
    for _ in range(args.warmup):
        step()
    pipe.fetch()
    pipe.enable_timing(True); pipe.kernel_time_stats(0, reset=True)
    nv.lib.nvx_channelise_timing(1); nv.channelise_time_stats(reset=True)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    pipe.fetch()
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{device}" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    casc_ms, n_l = pipe.kernel_time_stats(0)
    dem_ms, _ = pipe.kernel_time_stats(1)
    ch_ms, n_c = nv.channelise_time_stats()
    if rank == 0:
        casc_avg, ch_avg = casc_ms / max(n_l, 1), ch_ms / max(n_c, 1)
        sub_bytes = 8 * W * n_sub * 4
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
            "roofline": {"bound": "hbm", "kernel": "nvx_fir_cascade<252k,2>", "achieved": round(sub_bytes / (casc_avg * 1e-3) / 1e9, 1) if casc_avg else None,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(sub_bytes / (casc_avg * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if casc_avg else None,
                         "traffic": None, "algorithmic_bytes_per_launch": sub_bytes, "avg_launch_ms": round(casc_avg, 3), "launches": int(n_l),
                         "demod_avg_launch_ms": round(dem_ms / max(n_l, 1), 3),
                         "note": "fp64-issue-bound by design at 252 kS/s (DESIGN.md 3.1, Variant A)"},
            "channeliser": {"kernel": "nvx_channelise", "avg_launch_ms": round(ch_avg, 3), "launches": int(n_c),
                            "read_plus_write_gbs": round((W * n_raw * 4 + sub_bytes) / (ch_avg * 1e-3) / 1e9, 1) if ch_avg else None},
            "cpu_baseline": cpu, "parity": parity, "gen_seconds": round(t_gen, 1),
        }
        print(json.dumps(line), flush=True)
    pipe.close(); raw.free()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one process per GPU)")
        args.gpus = world

    import torch
    dist = None
    # NVX_BENCH_BACKEND=gloo / NVX_BENCH_DEVICE=<n>: rehearsal of the multi-process path on a box
    # with fewer GPUs than ranks (RCCL refuses two ranks on one device); never used by the driver.
    backend = os.environ.get("NVX_BENCH_BACKEND", "nccl")
    device = int(os.environ.get("NVX_BENCH_DEVICE", local))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(device)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    if not (ROOT / "navtex_amd" / "libnavtex_amd.so").exists():       # fresh checkout: build first (hipcc, gfx950)
        import importlib.util
        spec = importlib.util.spec_from_file_location("nvx_build", ROOT / "navtex_amd" / "build.py")
        mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
        if rank == 0:
            import contextlib
            with contextlib.redirect_stdout(sys.stderr):       # stdout carries exactly one JSON line
                mod.build_lib()
        if dist is not None:
            dist.barrier()
    import navtex_amd as nv
    import signals

    if nv.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X: navtex_amd has no CPU path")
    if args.wideband:
        run_wideband(args, nv, signals, torch, dist, rank, world, device, backend)
        if dist is not None:
            dist.barrier(); dist.destroy_process_group()
        return
    S, F = args.streams, args.frames
    raw = not args.variant_a
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
                       char_layer=not args.no_charlayer, device=device)

    def sync_all():
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(device)

    # ---- parity gate + CPU baseline (rank 0, N = 1 leg of the contract) ------
    cpu = None
    parity = None
    if rank == 0:
        import oracle_binding as ob
        # the GPU box gives one GPU a 16-thread CPU share whatever the affinity mask says
        ncpu = int(os.environ.get("NVX_CPU_THREADS", min(16, len(os.sched_getaffinity(0)))))
        n_cs = min(args.cpu_streams or 2 * ncpu, S)
        cf = F                         # whole batch length: every cascade dispatch of this run has the same shape
        # the sample = the first cf frames of the first n_cs streams, copied back from HBM
        sample = np.empty((n_cs, cf * FRAME, 2), dtype=np.int16)
        for s in range(n_cs):
            sample[s] = buf.download(cf * FRAME * 4, offset=s * pitch * 4, dtype=np.int16).reshape(-1, 2)
        # GPU bits for the same frames, from reset state
        pipe.process_resident(buf, pitch, 0, cf)
        pipe.fetch()
        gpu_bits = [pipe.bits(s, 0) for s in range(n_cs)]
        if not args.no_cpu and world == 1:
            n252 = cf * nv.FRAME_IN
            per_pass = n_cs * cf * FRAME
            # calibrate, then size the repeat count for ~6 s wall on all threads (~100 core-seconds
            # at 16 threads would exceed the "few minutes" budget; this is ~6 s x ncpu core-seconds)
            sN, cpu_bits = ob.bench(sample, n_cs, n252, raw, 1, ncpu, want_bits=True)
            rep = max(1, int(6.0 / max(sN, 1e-3)))
            sN = ob.bench(sample, n_cs, n252, raw, 1, ncpu, repeat=rep)[0]
            one = min(n_cs, 2)
            s1 = ob.bench(sample[:one], one, n252, raw, 1, 1, repeat=max(1, rep // 8))[0]
            cpu = {
                "value": round(per_pass * rep / sN / 1e6, 2), "unit": "Msamples/s",
                "cores": ncpu, "kind": "port",
                "value_1thread": round(one * cf * FRAME * max(1, rep // 8) / s1 / 1e6, 2),
                "sample": f"all {cf} frames of the first {n_cs} streams of the bench batch ({per_pass / 1e6:.0f} M raw samples), "
                          f"processed {rep}x; oracle/nvx_oracle.c (gcc -O2 -ffp-contract=off), OpenMP over streams",
                "seconds": round(sN, 2),
            }
            cpu["reference_check"] = reference_check(ob, sample[0])
        else:
            cpu_bits = []
            for s in range(n_cs if world == 1 else min(n_cs, 4)):
                o = ob.Pipe(chain_mask=1, charlayer=False)
                (o.push_raw if raw else o.push)(sample[s])
                cpu_bits.append(o.bits(0))
        parity = all(g == c for g, c in zip(gpu_bits, cpu_bits)) and len(cpu_bits) > 0 and all(len(c) > 0 for c in cpu_bits)
        if not parity:
            print("PARITY FAILURE: GPU bits differ from the CPU oracle", file=sys.stderr)
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
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    pipe.fetch()                       # all launches done, bits on the host, characters decoded
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{device}" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    casc_ms, n_l = pipe.kernel_time_stats(0)
    dem_ms, _ = pipe.kernel_time_stats(1)
    w_polls, w_units, w_launches = pipe.wait_stats()
    total_bits = sum(pipe.bit_count(s, 0) for s in range(0, S, max(1, S // 64)))

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * samples_per_step * args.steps / elapsed / 1e6
        casc_avg = casc_ms / max(n_l, 1)
        achieved = bytes_per_step / (casc_avg * 1e-3) / 1e9 if casc_avg > 0 else None
        traffic = None
        tf = ROOT / "profiles" / "hbm_traffic.json"
        if tf.exists():
            try:
                rec = json.loads(tf.read_text())
                if raw and rec.get("streams") == S and rec.get("frames") == F:
                    traffic = rec.get("bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "IQ Msamples/s through FIR->FSK->bitsync", "value": round(value, 1), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": (f"{S} synthetic 170 Hz-shift FSK channels x 2.016 MS/s int16 IQ per GPU, "
                                    f"{F} frames ({F * 0.32:.2f} s) resident in HBM (BASELINE configs[3]; x{world} GPUs = configs[4] shape)") if raw else
                                   (f"VARIANT A (not the headline): {S} channels x 252 kS/s int16 IQ per GPU, {F} frames ({F * 0.32:.2f} s), "
                                    f"no stage 0 -- fp64-issue-bound by design (SURVEY 7-2)"),
                       "streams_per_gpu": S, "frames": F, "samples_per_step_per_gpu": samples_per_step,
                       "stage0": "integrate-and-dump /8 (build-owned)" if raw else "none", "chains_per_stream": 1,
                       "parallelism": f"streams sharded {world} ways, no collective"},
            "roofline": {"bound": "hbm", "kernel": "nvx_fir_cascade<raw,1>" if raw else "nvx_fir_cascade<252k,1>", "achieved": round(achieved, 1) if achieved else None,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4) if achieved else None,
                         "traffic": traffic, "algorithmic_bytes_per_launch": bytes_per_step,
                         "avg_launch_ms": round(casc_avg, 3), "launches": int(n_l),
                         "demod_avg_launch_ms": round(dem_ms / max(n_l, 1), 3),
                         # hand-over of filter state between the frames of a stream: share of units that had to
                         # wait for their predecessor and the average wait of those (one poll ~ 1 us)
                         "handoff": {"units_waited_frac": round(w_units / max(1, w_launches * S * F), 4),
                                     "avg_polls_per_waiting_unit": round(w_polls / max(1, w_units), 1)}},
            "cpu_baseline": cpu,
            "parity": parity,
            "hbm_gbs_whole_job": round(world * bytes_per_step * args.steps / elapsed / 1e9, 1),
            "gen_seconds": round(t_gen, 1), "bits_sampled": int(total_bits),
        }
        print(json.dumps(line), flush=True)

    pipe.close()
    buf.free()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
