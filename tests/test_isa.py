"""Guards on the generated gfx950 code (hipcc cross-compiles without a GPU):
the arithmetic contract (no fused multiply-add anywhere in the FIR cascade), no
scratch spills, and the wide coalesced loads the roofline kernel depends on."""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
pytestmark = pytest.mark.skipif(not Path(HIPCC).exists(), reason="hipcc not available")


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    tmp = tmp_path_factory.mktemp("isa")
    csrc = ROOT / "navtex_amd" / "csrc"
    kernels, meta = {}, ""
    for src in sorted(csrc.glob("*.hip")):                # every device translation unit of the product
        out = tmp / (src.stem + ".s")
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", f"-I{ROOT / 'include'}", f"-I{csrc}",
                        "--cuda-device-only", "-S", str(src), "-o", str(out)], check=True, capture_output=True)
        text = out.read_text()
        for m in re.finditer(r"^(_Z\w+|nvx_\w+):.*?s_endpgm", text, flags=re.S | re.M):
            kernels[m.group(1)] = m.group(0)
        meta += text[text.index("amdhsa.kernels"):] if "amdhsa.kernels" in text else ""
    return kernels, meta


def test_cascade_has_no_fused_multiply_add(isa):
    kernels, _ = isa
    casc = {k: v for k, v in kernels.items() if "nvx_fir_cascade" in k}
    assert len(casc) == 12
    for name, body in casc.items():
        assert not re.search(r"v_fma_f64|v_fmac_f64|v_fma_f32|v_fmac_f32|v_pk_fma", body), f"{name}: FMA breaks the reference's rounding"
        assert body.count("v_mul_f64") >= 37 * 2 - 2 + 47 + 71        # FIR1 (two outputs per lane; equal taps on one sample share a product) + FIR2 + FIR3, fully unrolled
    # r4: the fused wideband kernel's waves end at FIR2; its FIR3 is nvx_fir3 -- same contract
    fused = next(v for k, v in kernels.items() if "nvx_wideband_fused" in k)
    assert not re.search(r"v_fma_f64|v_fmac_f64", fused) and 37 * 2 - 2 + 47 <= fused.count("v_mul_f64") < 37 * 2 + 47 + 71
    fir3 = next(v for k, v in kernels.items() if "nvx_fir3" in k)
    assert not re.search(r"v_fma_f64|v_fmac_f64|v_fma_f32|v_fmac_f32|v_pk_fma", fir3), "nvx_fir3: FMA breaks the reference's rounding"
    assert fir3.count("v_mul_f64") % 71 == 0 and fir3.count("v_mul_f64") > 0 and fir3.count("v_add_f64") >= 70 and "scratch_" not in fir3
    assert "ds_read2_b64" not in fir3 and "ds_read2st64_b64" not in fir3


def test_channeliser_phase_instruction_count_behind_the_bench_lines_decomposition(isa):
    """bench.py's wideband.roofline.decomposition quotes the channeliser's vector instructions per raw sample
    (WB_CHANNELISER_VALU_PER_LANE: per output instant and component, between the fused kernel's two barriers): held here
    against the compiled kernel, so that the figure in the line cannot drift away from the code."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_for_isa", ROOT / "bench.py"); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    kernels, _ = isa
    fused = next(v for k, v in kernels.items() if "nvx_wideband_fused" in k)
    lines = [l.strip() for l in fused.splitlines() if l.strip() and not l.strip().startswith((";", "."))]
    dots = [i for i, l in enumerate(lines) if l.startswith("v_dot2")]
    bars = [i for i, l in enumerate(lines) if l.startswith("s_barrier")]
    assert len(dots) == 48                               # one (tap, sample) product per instruction: 48 taps
    assert sum(1 for i in dots if lines[i].startswith("v_dot2_i32_i16")) == 8      # each branch's first product carries the rounding constant (VOP3P, inline 16): no accumulator to clear
    lo, hi = max(b for b in bars if b < dots[0]), min(b for b in bars if b > dots[-1])
    phase = lines[lo:hi]
    valu = sum(1 for l in phase if l.startswith("v_"))
    assert abs(valu - bench.WB_CHANNELISER_VALU_PER_LANE) <= 4, valu
    assert sum(1 for l in phase if l.startswith("ds_read_b128")) == 12 and sum(1 for l in phase if l.startswith("ds_write_b64")) == 8
    assert not any(l.startswith(("v_mul_f64", "v_add_f64")) for l in phase)     # integer work only: no credited fp64 operation in this phase


def test_roofline_kernel_uses_wide_nt_loads_and_no_scratch(isa):
    kernels, meta = isa
    main = next(v for k, v in kernels.items() if "nvx_fir_cascadeILb1ELi1EE" in k)
    assert len(re.findall(r"global_load_dwordx4 .* nt", main)) >= 16   # 8 per pass, prologue + loop
    assert "scratch_" not in main and "buffer_store" not in main
    assert main.count("v_add_u32_sdwa") >= 32                          # stage 0: 4 half-word pair adds per load, 8 loads
    assert main.count("v_add_u32_dpp") >= 8 and "v_add3_u32" not in main.split("v_add_u32_sdwa", 1)[1].split("ds_read_b64", 1)[0]
    assert "ds_read2_b64" not in main and "ds_read2st64_b64" not in main   # paired 8-byte LDS reads run at half rate (nvx_device.h)
    assert "s_barrier" not in main                                     # single-wave workgroups: compiler fences only


@pytest.mark.parametrize("inst", ["ILb1ELi1EE", "ILb1ELi2EE", "ILb0ELi1EE", "ILb0ELi2EE", "_cic3_1", "_cic3_2",
                                  # r3: the kernels of launches that name their streams (participant list read with ONE scalar load)
                                  "_listILb1ELi1ELi1E", "_listILb1ELi2ELi1E", "_listILb0ELi1ELi1E", "_listILb0ELi2ELi1E", "_listILb1ELi1ELi3E", "_listILb1ELi2ELi3E"])
def test_unit_hand_over_is_fence_free_and_device_coherent(isa, inst):
    """The hand-over of filter state between the units of a stream (nvx_cascade.hip, state_load / state_store / done[])
    rests on per-instruction device coherence instead of cache-wide fences.  What the hardware needs for that
    (MI355X_MICROARCH.md, "Valid forms") must survive every compiler bump:
      * every access to the state block is an sc1 access (all 8-byte global loads of the kernel are state loads; all
        8-byte stores but the y3 output are state stores);
      * the producer drains its stores (s_waitcnt vmcnt(0)) and only then stores the flag, itself sc1, with no other
        store in between; the consumer polls the flag with sc1 loads;
      * no agent-scope fence (buffer_wbl2 / buffer_inv) in the shipped build."""
    kernels, _ = isa
    body = next(v for k, v in kernels.items() if "nvx_fir_cascade" + inst in k)
    lines = [l.strip() for l in body.splitlines() if l.startswith("\t")]
    assert not any(l.startswith(("buffer_wbl2", "buffer_inv")) for l in lines)
    ld8 = [l for l in lines if l.startswith("global_load_dwordx2")]
    st8 = [l for l in lines if l.startswith("global_store_dwordx2")]
    n_chains = 2 if ("Li2E" in inst or inst.endswith("_2")) else 1
    n_state = 2 * (1 + 3 * n_chains)                      # double2 = two 8-byte accesses: 252 kS/s window + per chain U, Y2, Y2 tail
    assert len(ld8) >= n_state and all(l.endswith(" sc1") for l in ld8), ld8
    assert sum(l.endswith(" sc1") for l in st8) >= n_state
    assert sum(not l.endswith(" sc1") for l in st8) <= 2 * n_chains          # the 900 S/s output, plain stores
    assert any(re.match(r"global_load_dword .* sc1$", l) for l in lines)    # done[] poll
    if "_list" in inst:                                                     # the list entry: a scalar load, not one per lane
        assert not any(l.startswith("global_load_dwordx2") and not l.endswith(" sc1") for l in lines)
    drains = [i for i, l in enumerate(lines) if l == "s_waitcnt vmcnt(0)"]
    assert len(drains) >= 2
    flag = max(i for i, l in enumerate(lines) if l.startswith("global_store_dword "))
    assert lines[flag].endswith(" sc1")
    last_state_store = max(i for i, l in enumerate(lines) if l.startswith("global_store_dwordx2") and l.endswith(" sc1"))
    drain = max(i for i in drains if i < flag)
    assert last_state_store < drain < flag, "the flag must follow a full drain of the state stores"
    between = lines[drain + 1:flag]
    assert not any(l.startswith(("global_store", "global_atomic", "flat_", "buffer_")) for l in between), between
    # ... and nothing of the unit follows the flag but the loop back-edge
    after = [l for l in lines[flag + 1:] if l.startswith(("global_", "ds_", "flat_"))]
    assert after == [], after


def test_third_order_stage0_code_shape_and_occupancy(isa):
    """The third-order stage 0 (nvx_cascade.hip, Stage0Cic3): per 1-KiB load four v_perm, twelve v_dot2, four wave
    rotations; no fused multiply-add behind it either; and the single-chain kernel stays within the 168 VGPRs that three
    waves per SIMD -- the 11 per CU its LDS allows -- can have."""
    kernels, meta = isa
    body = next(v for k, v in kernels.items() if "nvx_fir_cascade_cic3_1" in k)
    assert len(re.findall(r"v_dot2c?_i32_i16", body)) == 8 * 12
    assert body.count("wave_ror:1") == 8 * 4 and body.count("v_perm_b32") == 8 * 4
    assert not re.search(r"v_fma_f64|v_fmac_f64", body) and "scratch_" not in body
    assert len(re.findall(r"global_load_dwordx4 .* nt", body)) >= 16
    m = re.search(r"\.name:\s+_Z22nvx_fir_cascade_cic3_1.*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", meta, re.S)
    assert m and int(m.group(1)) <= 168 and int(m.group(2)) == 0, m and m.groups()


def test_no_kernel_spills(isa):
    _, meta = isa
    sizes = [int(x) for x in re.findall(r"\.private_segment_fixed_size:\s*(\d+)", meta)]
    assert len(sizes) == len(SHIPPED_KERNELS) and all(s == 0 for s in sizes), sizes


# Every kernel of the library, by (demangled) name.  r5: the library holds no kernel the GPU suite does not launch -- the
# A/B instantiations of earlier rounds (two passes of prefetch, plain loads, compile-time alternates) are gone;
# profiles/r05/suite_kernels.txt is the list of kernel names rocprofv3 saw while `pytest -m gpu` ran.
SHIPPED_KERNELS = sorted([
    "nvx_fir_cascade<false, 1>", "nvx_fir_cascade<false, 2>", "nvx_fir_cascade<true, 1>", "nvx_fir_cascade<true, 2>",
    "nvx_fir_cascade_cic3_1", "nvx_fir_cascade_cic3_2",
    "nvx_fir_cascade_list<false, 1, 1>", "nvx_fir_cascade_list<false, 2, 1>", "nvx_fir_cascade_list<true, 1, 1>",
    "nvx_fir_cascade_list<true, 2, 1>", "nvx_fir_cascade_list<true, 1, 3>", "nvx_fir_cascade_list<true, 2, 3>",
    "nvx_wideband_fused", "nvx_fir3", "nvx_channelise",
    "nvx_demod_front", "nvx_demod_front_head", "nvx_demod_front_tiles", "nvx_demod_fsm",
    "nvx_synth_kernel",
])


def test_the_library_holds_exactly_the_kernels_the_gpu_suite_launches(isa):
    _, meta = isa
    names = re.findall(r"^\s+\.name:\s+(\S+)\s*$", meta, flags=re.M)
    out = subprocess.run(["c++filt"] + names, capture_output=True, text=True, check=True).stdout.split("\n")
    got = sorted(re.sub(r"^void |\(.*$", "", n) for n in out if n)
    assert got == SHIPPED_KERNELS, got
    suite = ROOT / "profiles" / "r05" / "suite_kernels.txt"
    if suite.exists():                                      # the record of a `pytest -m gpu` run under rocprofv3 --kernel-trace
        seen = {re.sub(r"^void |\(.*$", "", l.strip()) for l in suite.read_text().splitlines() if l.strip() and not l.startswith("#")}
        assert set(SHIPPED_KERNELS) <= seen, sorted(set(SHIPPED_KERNELS) - seen)
