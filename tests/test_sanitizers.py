"""ASan + UBSan over the product's host-side C (CPU build only; GPU sanitizers are not
available on the pool).  Inputs: every golden character-layer case, fuzz, WAV edge
cases, generator sweeps, the SQLite sink, the demodulator FSM tables -- see tests/harness/sanitize_host.c."""
import json
import subprocess
from pathlib import Path

import cases

ROOT = Path(__file__).resolve().parent.parent
GOLD = json.loads((Path(__file__).parent / "golden" / "golden.json").read_text())


def test_host_c_is_clean_under_asan_ubsan(nv, tmp_path):
    exe = tmp_path / "sanitize_host"
    csrc = ROOT / "navtex_amd" / "csrc"
    subprocess.run(["gcc", "-std=gnu11", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
                    f"-I{ROOT / 'include'}", f"-I{csrc}", str(ROOT / "tests" / "harness" / "sanitize_host.c"),
                    str(csrc / "nvx_sitor.c"), str(csrc / "nvx_wav.c"), str(csrc / "nvx_synth_host.c"), str(csrc / "nvx_store.c"),
                    "-o", str(exe), "-ldl", "-lpthread"], check=True)
    lines = [cases.make_bits(nv, rec["spec"]) for rec in GOLD["charlayer"].values()]
    lines += [rec["bits518"] for rec in GOLD["iq"].values()]
    case_file = tmp_path / "cases.txt"
    case_file.write_text("\n".join(lines) + "\n")
    out = subprocess.run([str(exe), str(case_file), str(tmp_path / "s.wav"), str(tmp_path / "s.db")], capture_output=True, text=True, timeout=600,
                         env={"ASAN_OPTIONS": "detect_leaks=1", "PATH": "/usr/bin:/bin"})      # leaks inside libsqlite3 itself would show up too
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    assert "sanitize ok" in out.stdout


def test_capture_ring_is_clean_under_tsan(tmp_path):
    """ThreadSanitizer over the live-capture ring with the GPU pipeline replaced by an order checker:
    vendor-callback thread, consumer thread and a pause/resume thread; overruns are provoked and every
    accepted sample must come out exactly once, in order (tests/harness/tsan_capture.cpp)."""
    exe = tmp_path / "tsan_capture"
    csrc = ROOT / "navtex_amd" / "csrc"
    subprocess.run(["g++", "-std=c++17", "-g", "-O1", "-fsanitize=thread", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                    f"-I{ROOT / 'include'}", f"-I{csrc}", str(ROOT / "tests" / "harness" / "tsan_capture.cpp"),
                    str(csrc / "nvx_capture.cpp"), "-x", "c", str(csrc / "nvx_wav.c"), "-o", str(exe), "-lpthread"], check=True)
    out = subprocess.run([str(exe), str(tmp_path / "rec.wav")], capture_output=True, text=True, timeout=600,
                         env={"TSAN_OPTIONS": "halt_on_error=1", "PATH": "/usr/bin:/bin"})
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    assert "tsan capture ok" in out.stdout and "dropped 0 " not in out.stdout      # overruns really happened


def test_group_is_clean_under_tsan(tmp_path):
    """ThreadSanitizer over the multi-device group (nvx_group.cpp): four member workers with their job queues, a thread
    pushing into streams of every member, a thread polling bits, launches and fetches from the main thread; the handle
    behind every member is a stand-in with the real locking contract (tests/harness/tsan_group.cpp).  No race; every
    message exactly once, global stream ids, member order within a fetch; a member's failure is reported with its name."""
    exe = tmp_path / "tsan_group"
    csrc = ROOT / "navtex_amd" / "csrc"
    subprocess.run(["g++", "-std=c++17", "-g", "-O1", "-fsanitize=thread", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                    f"-I{ROOT / 'include'}", f"-I{csrc}", str(ROOT / "tests" / "harness" / "tsan_group.cpp"),
                    str(csrc / "nvx_group.cpp"), "-o", str(exe), "-lpthread"], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600,
                         env={"TSAN_OPTIONS": "halt_on_error=1", "PATH": "/usr/bin:/bin"})
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    assert "tsan group ok" in out.stdout


def test_host_pool_is_clean_under_tsan(tmp_path):
    """ThreadSanitizer over the persistent character-layer workers of a handle (nvx_pool.h): 3000 runs of changing width,
    every index exactly once, plain writes of the jobs visible to the caller afterwards (tests/harness/tsan_pool.cpp)."""
    exe = tmp_path / "tsan_pool"
    csrc = ROOT / "navtex_amd" / "csrc"
    subprocess.run(["g++", "-std=c++17", "-g", "-O1", "-fsanitize=thread", f"-I{csrc}", str(ROOT / "tests" / "harness" / "tsan_pool.cpp"),
                    "-o", str(exe), "-lpthread"], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600, env={"TSAN_OPTIONS": "halt_on_error=1", "PATH": "/usr/bin:/bin"})
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    assert "tsan pool ok" in out.stdout


def test_push_path_is_clean_under_tsan(tmp_path):
    """ThreadSanitizer over the host-input path (nvx_push.cpp): six pusher threads (callback-sized and replay-sized pushes,
    the big ones copying into the staging WITHOUT the handle's lock), a slow stream, a stream declared silent, a thread
    flushing; HIP and the launch are stand-ins that record what reached the "device".  No race; every stream's samples
    arrive exactly once and in order; partial launches really happened.  Then streams are ended (nvx_stream_finish, nvx_finish)
    under pushers that are in the middle of calls, in both launch modes: every call is accepted whole or refused whole, what
    reaches the "device" is exactly what was accepted, nothing is staged behind a stream's end and no launch ever names an
    ended stream -- the other streams keep launching (tests/harness/tsan_push.cpp)."""
    exe = tmp_path / "tsan_push"
    csrc = ROOT / "navtex_amd" / "csrc"
    subprocess.run(["g++", "-std=c++17", "-g", "-O1", "-fsanitize=thread", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                    f"-I{ROOT / 'include'}", f"-I{csrc}", str(ROOT / "tests" / "harness" / "tsan_push.cpp"),
                    str(csrc / "nvx_push.cpp"), "-x", "c", str(csrc / "nvx_wav.c"), "-o", str(exe), "-lpthread"], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600, env={"TSAN_OPTIONS": "halt_on_error=1", "PATH": "/usr/bin:/bin"})
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    assert "tsan push ok" in out.stdout
