"""The C-ABI boundary: the library loads without a GPU, exports every symbol
include/navtex_amd.h declares, and its device entry points fail loudly (no CPU
fallback) when there is no device.  CPU only -- no compute calls."""
import ctypes as C
import re
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
HEADER = (ROOT / "include" / "navtex_amd.h").read_text()


def declared_symbols():
    names = set(re.findall(r"NVX_API\s+[\w\s\*]+?\b(\w+)\s*\(", HEADER))
    names |= {"add_message"}                     # declared without NVX_API (weak default sink)
    return sorted(names)


def test_header_declares_the_reference_surface():
    names = declared_symbols()
    for sym in ("init_fir_filter1", "sample_in_1", "init_fir2_wrapper", "add_message", "nvx_StreamACallback",
                "nvx_create", "nvx_push_iq", "nvx_push_planar", "nvx_poll_bits", "nvx_flush", "nvx_destroy",
                "nvx_wav_open", "nvx_wav_read", "nvx_sitor_receive_bit"):
        assert sym in names
    assert len(names) > 50


@pytest.mark.parametrize("sym", declared_symbols())
def test_symbol_is_exported(nv, sym):
    assert hasattr(nv.lib, sym), f"{sym} is declared in navtex_amd.h but not exported by libnavtex_amd.so"


def test_no_torch_or_cxx_types_in_signatures():
    body = HEADER[HEADER.index("extern \"C\""):]
    assert "torch" not in body and "std::" not in body and "#include <hip" not in HEADER
    # the only includes are the two freestanding C headers
    assert re.findall(r"#include\s+<([^>]+)>", HEADER) == ["stddef.h", "stdint.h"]


def test_header_compiles_as_plain_c(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "navtex_amd.h"\nint main(void){ nvx_config c; nvx_config_default(&c); return c.n_streams != 1; }\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", f"-I{ROOT / 'include'}", "-c", str(src), "-o", str(tmp_path / "t.o")],
                   check=True)


def test_the_c_snippets_of_integration_md_compile(tmp_path):
    """Every ```c block of INTEGRATION.md that is a complete statement sequence compiles as C99 against the header (-Wall
    -Werror), wrapped in a function with the variables the text assumes: a maintainer pastes these."""
    text = (ROOT / "INTEGRATION.md").read_text()
    blocks = re.findall(r"```c\n(.*?)```", text, flags=re.S)
    assert len(blocks) >= 5
    prelude = """
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
#include "navtex_amd.h"
typedef void (*sdrplay_api_StreamCallback_t)(short *, short *, void *, unsigned int, unsigned int, void *);
struct { sdrplay_api_StreamCallback_t StreamACbFn; } cbFns;
struct { void *dev; } dev0, *chosenDevice = &dev0;
static void sdrplay_api_Init(void *d, void *fns, void *ctx) { (void)d; (void)fns; (void)ctx; }
static void die(const char *m) { fputs(m, stderr); exit(1); }
static void my_sink(void *u, int s, const char *b, const char *m, int f) { (void)u; (void)s; (void)b; (void)m; (void)f; }
static const void *upload(int dev, int first, int n) { (void)dev; (void)first; (void)n; return 0; }
"""
    compiled = 0
    for i, b in enumerate(blocks):
        if "nvx_" not in b or "#cgo" in b:
            continue
        body = b.replace("...", "").replace("#include \"navtex_amd.h\"", "")
        src = tmp_path / f"snip{i}.c"
        src.write_text(prelude + "void snippet(nvx_handle *handle, nvx_handle *h, nvx_capture *cap, void *d_iq, size_t pitch_samples, size_t pitch, int dev, int n_wide, int F,\n"
                       "             void *d_raw, void *d_sub, size_t pitch_raw, size_t pitch_sub, void *hist_in, void *hist_out, char *buf, size_t buf_cap) {\n"
                       "  (void)d_iq; (void)handle; (void)h; (void)cap; (void)pitch_samples; (void)pitch; (void)dev; (void)n_wide; (void)F;\n"
                       "  (void)d_raw; (void)d_sub; (void)pitch_raw; (void)pitch_sub; (void)hist_in; (void)hist_out; (void)buf; (void)buf_cap;\n"
                       "  {\n" + body + "\n  }\n}\n")
        r = subprocess.run(["gcc", "-std=gnu99", "-Wall", "-Werror", "-Wno-unused-variable", "-Wno-unused-but-set-variable", "-Wno-unused-function", "-Wno-shadow",
                            f"-I{ROOT / 'include'}", "-c", str(src), "-o", str(tmp_path / f"snip{i}.o")], capture_output=True, text=True)
        assert r.returncode == 0, f"INTEGRATION.md c block {i}:\n{b}\n{r.stderr[-1500:]}"
        compiled += 1
    assert compiled >= 4


def test_links_against_a_c_program_with_its_own_add_message(nv, tmp_path):
    """The drop-in claim at link level: a C program that defines add_message (as the
    receiver's message_store.o does) and calls the three reference symbols links against
    the library, and the library's weak default does not clash."""
    src = tmp_path / "host.c"
    src.write_text("""
        void init_fir_filter1(void); void sample_in_1(double, double); void init_fir2_wrapper(void);
        int add_message(char *bbbb, char *message, int freq) { (void)bbbb; (void)message; return freq; }
        int main(int argc, char **argv) { (void)argv; if (argc > 99) { init_fir_filter1(); init_fir2_wrapper(); sample_in_1(1.0, 2.0); } return 0; }
    """)
    exe = tmp_path / "host"
    lib = ROOT / "navtex_amd"
    subprocess.run(["gcc", str(src), "-o", str(exe), f"-L{lib}", "-lnavtex_amd", f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    subprocess.run([str(exe)], check=True)


def test_null_objects_are_errors_or_no_ops_never_crashes(nv, tmp_path):
    """tests/harness/null_args.c calls every entry point that takes an object or a pointer it must read with NULL, in a
    process of its own (a crash would be a signal, not a Python error): error codes and no-ops throughout.  The entry points
    the program does NOT call are listed here with the reason, and the two lists together are the header's."""
    src = ROOT / "tests" / "harness" / "null_args.c"
    exe = tmp_path / "null_args"
    lib = ROOT / "navtex_amd"
    subprocess.run(["gcc", "-O1", "-g", "-Wall", "-Werror", f"-I{ROOT / 'include'}", str(src), "-o", str(exe), f"-L{lib}", "-lnavtex_amd",
                    f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "null-safety ok" in out.stdout, (out.stdout[-1500:], out.stderr[-500:], out.returncode)
    not_called = {
        # the reference-shaped void surface creates the GPU singleton on first use (aborts without a device, by contract)
        "init_fir_filter1", "sample_in_1", "init_fir2_wrapper", "nvx_StreamACallback", "add_message",
        # no object argument: plain values, or device memory helpers that need a device
        "nvx_last_error", "nvx_version", "nvx_wav_err", "nvx_config_default", "nvx_sample_to_int16", "nvx_device_count", "nvx_device_alloc", "nvx_device_free",
        "nvx_memcpy_h2d", "nvx_memcpy_d2h", "nvx_device_sync", "nvx_stream_create", "nvx_stream_destroy", "nvx_synth_device",
        "nvx_channelise_resident", "nvx_bind_thread_to_device", "nvx_sitor_new",
    }
    text = src.read_text()
    called = {sym for sym in declared_symbols() if re.search(rf"\b{sym}\(", text)}
    assert called | not_called == set(declared_symbols()), sorted(set(declared_symbols()) - called - not_called)
    assert not (called & (not_called - {"nvx_sample_to_int16"}))


def test_device_entry_points_fail_loudly_without_a_gpu(nv):
    if nv.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(nv.NvxError) as e:
        nv.Pipeline()
    assert e.value.code == -2 and "no CPU path" in str(e.value)
    st = nv.make_stream([], seed=1, noise_amp=10)
    assert nv.lib.nvx_synth_device(0, C.byref(st), 1, nv.RATE_IN, 16, C.c_void_p(16), 16) == -2
    assert nv.lib.nvx_device_alloc(0, 16) is None


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under navtex_amd/ or include/ may
    include, link or name it."""
    for path in list((ROOT / "navtex_amd").rglob("*")) + list((ROOT / "include").rglob("*")):
        if path.is_file() and path.suffix in {".c", ".cpp", ".h", ".hip", ".py"}:
            text = path.read_text(errors="replace")
            assert "nvx_oracle" not in text and "nvxo_" not in text and "oracle_binding" not in text, path
    out = subprocess.run(["ldd", str(ROOT / "navtex_amd" / "libnavtex_amd.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out
