"""bench.py's roofline arithmetic: the peaks a kernel is priced against, the algorithmic work per sample, what the fused
wideband kernel's figure is made of, and the static PMC traffic record (profiles/hbm_traffic.json).  No GPU call in here."""
from __future__ import annotations

import json
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]

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
# component each), counted in the compiled kernel between its two barriers (tests/test_isa.py holds the count): 120 per
# (instant, component) -- 48 dot products (one tap on one sample each), 16 shifts, 27 adds / subs, 8 v_med3 clamps, 8
# v_cvt_f64_i32, 5 DPP exchanges with the partner lane, two 64-bit products for the 45-degree twiddles -- beside 12
# ds_read_b128 and 8 ds_write_b64; two components, eight raw samples per instant.  (128 until r6: eight moves cleared
# accumulators that the VOP3P form of a branch's first product does not need; kernel -0.8 %, profiles/r06/c1_*.)
WB_CHANNELISER_VALU_PER_LANE = 120
WB_INT_OPS_PER_RAW_SAMPLE = WB_CHANNELISER_VALU_PER_LANE * 2 / 8          # 30


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
                       "alone runs at cascade_pass_frac_of_fp64_roof (about variant_a's efficiency); the channeliser's ~30 integer vector instructions per raw "
                       "sample cost the same issue slots as fp64 ones and earn no credit -- counted like fp64 operations the kernel issues at "
                       "valu_issue_frac_counting_integer_ops of the roof"}


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
                                  f"these are {now} (tools/gpu_scripts/gpu_r06_final.sh collects a new one, tools/update_hbm_traffic.py writes it)")
                return e.get("bytes_per_launch"), ("profiles/hbm_traffic.json (static: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload on these "
                                                   f"kernel sources ({now}), not measured by this run; {e.get('source', '')})")
    except Exception as e:
        return None, f"null: profiles/hbm_traffic.json unreadable ({type(e).__name__})"
    return None, (f"null: profiles/hbm_traffic.json holds PMC records of (streams, frames, stage-0 order) {have}, not of ({S}, {F}, {order}) "
                  "(tools/gpu_scripts/gpu_r06_final.sh collects them)")
