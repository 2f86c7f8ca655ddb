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
    assert len(casc) >= 8
    for name, body in casc.items():
        assert not re.search(r"v_fma_f64|v_fmac_f64|v_fma_f32|v_fmac_f32|v_pk_fma", body), f"{name}: FMA breaks the reference's rounding"
        assert body.count("v_mul_f64") >= 37 * 2 + 47 + 71            # FIR1 (I,Q) + FIR2 + FIR3 products, fully unrolled


def test_roofline_kernel_uses_wide_nt_loads_and_no_scratch(isa):
    kernels, meta = isa
    main = next(v for k, v in kernels.items() if "nvx_fir_cascadeILb1ELi1ELi1ELb1" in k)
    assert len(re.findall(r"global_load_dwordx4 .* nt", main)) >= 16   # 8 per pass, prologue + loop
    assert "scratch_" not in main and "buffer_store" not in main
    assert main.count("v_add_u32_sdwa") >= 32                          # stage 0: 4 half-word pair adds per load, 8 loads
    assert main.count("v_add_u32_dpp") >= 8 and "v_add3_u32" not in main.split("v_add_u32_sdwa", 1)[1].split("ds_read_b128", 1)[0]
    assert "s_barrier" not in main                                     # single-wave workgroups: compiler fences only


def test_no_kernel_spills(isa):
    _, meta = isa
    sizes = [int(x) for x in re.findall(r"\.private_segment_fixed_size:\s*(\d+)", meta)]
    assert len(sizes) >= 16 + 4 and all(s == 0 for s in sizes), sizes      # 16 cascade instantiations + demod x2, channeliser, generator
