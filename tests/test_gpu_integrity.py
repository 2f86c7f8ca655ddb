"""The carried filter state is sealed (nvx_kernels.h "The seal", nvx_cascade_integrity_stats).

The reference keeps its filter histories in statics (receiver/fir1cpp.C:51-60, receiver/fir2cpp.C:74-83,
receiver/fir3cpp.h:90-95); here they travel from work unit to work unit through a state block in HBM, inside a launch by
a fence-free, per-instruction-coherent protocol that the hardware guide calls measured rather than guaranteed.  So every
block carries a word over its contents and its position, and these tests show what the word is for: any single flipped
bit of a carried word, a block of the wrong position, stream or launch, and the swapped pairs tried below fail (it is a
stale / torn block detector, not a permutation check: two words whose slot and lane rotations add up to the same total
could be swapped unnoticed, nvx_cascade_wave.h); a stale hand-over is caught, repaired bit-exactly and counted; the
shipped build never sees one; and a launch that fails poisons its handle until nvx_reset."""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

import signals

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
INJECT = ROOT / "tests" / "_variants" / "libnavtex_amd_inject.so"

# layout of a state block in 8-byte words (nvx_kernels.h): 268 data entries of two words, the third-order stage 0's entry,
# the seal entry {word, tag}, padding
W_CIC3, W_SEAL, W_TAG, W_PAD = 2 * 268, 2 * 269, 2 * 269 + 1, 2 * 270


def _handle(nv, raw, masks, order=1, frames=2, total=4):
    rate, frame = (nv.RATE_RAW, nv.FRAME_RAW) if raw else (nv.RATE_IN, nv.FRAME_IN)
    S = len(masks)
    iqs = []
    for s in range(S):
        car = [dict(freq_hz=f, bits=nv.sitor_encode(signals.stream_text(40 + s), 10), bit_offset=(313 * (s + 1)) | 1, phase0=s * 999331, amplitude=6000)
               for c, f in ((0, 14000), (1, -14000)) if (masks[s] >> c) & 1]
        iqs.append(nv.synth_host(nv.make_stream(car, seed=40 + s, noise_amp=1500), rate, total * frame))
    buf = nv.DeviceBuffer(S * total * frame * 4)
    for s in range(S):
        buf.upload(iqs[s], offset=s * total * frame * 4)
    p = nv.Pipeline(n_streams=S, raw_rate=raw, chain_masks=masks, max_frames=frames, char_layer=False, stage0_order=order)
    return p, buf, total * frame


def _covered_words(mask, nch, order):
    """8-byte words of a block that a unit of this kernel stores (and the next one loads): the window, the chains the kernel
    handles (a one-chain kernel: the stream's only chain; a two-chain kernel: both, decoded or not), the third-order
    stage 0's entry, the seal word."""
    words = list(range(0, 2 * 36))
    for ch in range(2):
        if nch == 2 or (mask >> ch) & 1:
            words += list(range(2 * (36 + 116 * ch), 2 * (36 + 116 * (ch + 1))))
    if order == 3:
        words += [W_CIC3, W_CIC3 + 1]
    return words + [W_SEAL]


@pytest.mark.parametrize("raw,masks,order", [(False, [1], 1), (False, [2, 3], 1), (True, [2], 1), (True, [3, 1], 3), (True, [1], 3)],
                         ids=["252k-1ch", "252k-2ch", "raw-1ch", "raw-cic3-2ch", "raw-cic3-1ch"])
def test_any_changed_bit_of_the_inherited_state_fails_the_launch(nv, raw, masks, order):
    """Launch, then change ONE bit of one 8-byte word of the block the next launch will read -- a sample of words that
    covers every lane and every slot of the kernel's store list, the seal itself included -- and launch again: the
    launch is reported as failed (NVX_ERR_HIP at the fetch, launch_failures counts), its bits are discarded, and
    nvx_reset recovers.  Words the kernel does not carry (the other chain of a one-chain kernel, the tag in clear, the
    padding) may change freely."""
    p, buf, pitch = _handle(nv, raw, masks, order)
    nch = 2 if 3 in masks else 1
    s = len(masks) - 1
    covered = _covered_words(masks[s], nch, order)
    rng = np.random.default_rng(5)
    sample = sorted(set([covered[0], covered[-1], covered[-2], 70, 71, 72, 73] + [int(w) for w in rng.choice(covered, 40)]) & set(covered))
    free = [w for w in (W_TAG, W_PAD, W_PAD + 1, 543) + tuple(range(2 * 36, 2 * 268, 29)) if w not in covered]
    failures = 0
    try:
        for trial, w in enumerate(sample + free):
            p.reset()
            p.process_resident(buf, pitch, 0, 2); p.fetch()
            blk = p.debug_state(s)
            assert blk[W_SEAL] != 0 and int(blk[W_TAG]) == (s << 32) | 6, "the seal of position 6 thirds"
            blk[w] ^= np.uint64(1) << np.uint64((7 * trial + 3) % 64)
            p.debug_set_state(s, blk)
            p.process_resident(buf, pitch, 2, 2)
            if w in covered:
                with pytest.raises(nv.NvxError, match="integrity"):
                    p.fetch()
                failures += 1
                assert p.integrity_stats()[1] == failures, f"word {w}"
            else:
                p.fetch()
                assert p.integrity_stats()[1] == failures, f"word {w} is not carried by this kernel"
        # ... and an undisturbed run of the same launches is clean and decodes
        p.reset()
        p.process_resident(buf, pitch, 0, 2); p.process_resident(buf, pitch, 2, 2); p.fetch()
        assert p.integrity_stats(reset=True)[:2] == (0, failures) and p.integrity_stats()[:2] == (0, 0)
        assert len(p.bits(s, 0 if masks[s] & 1 else 1)) > 40
    finally:
        p.close(); buf.free()


def test_a_changed_bit_in_a_wideband_handles_state_fails_the_launch(nv):
    """The fused wideband kernel seals nine blocks per work unit: the filter state of its eight sub-bands (one wave each: the
    252 kS/s window and both chains' mixer outputs -- its waves end at FIR2, so no FIR2-output history) and, with
    sub-band 7's block, the 40-sample channeliser halo.  One changed bit in any carried word of any sub-band's block fails
    the next launch; the words a sub-band's wave does not carry do not."""
    W, F = 2, 2
    n = 2 * F * nv.FRAME_RAW
    rng = np.random.default_rng(9)
    raw = rng.integers(-9000, 9000, size=(W, n, 2), dtype=np.int16)
    buf = nv.DeviceBuffer(W * n * 4)
    buf.upload(raw)
    covered = list(range(0, 2 * 36)) + [w for ch in range(2) for w in range(2 * (36 + 116 * ch), 2 * (36 + 116 * ch + 46))] + [W_SEAL]
    free = [2 * (36 + 46) + 5, 2 * (36 + 116 + 46) + 8, W_TAG, W_PAD]
    failures = 0
    with nv.Pipeline(n_streams=W, wideband=True, chain_mask=3, max_frames=F, char_layer=False) as p:
        trial = 0
        for s in (0, 3, 7, 8 + 7, 8 + 2):                       # decoded streams 8 w + k: several sub-bands of both wideband streams
            for w in [covered[0], covered[-1], covered[-2], 70, 2 * 36 + 3] + [int(x) for x in rng.choice(covered, 6)] + free:
                p.reset()
                p.process_resident(buf, n, 0, F); p.fetch()
                blk = p.debug_state(s)
                assert int(blk[W_TAG]) == (s << 32) | (3 * F)
                blk[w] ^= np.uint64(1) << np.uint64((11 * trial + 5) % 64); trial += 1
                p.debug_set_state(s, blk)
                p.process_resident(buf, n, F, F)
                if w in covered:
                    with pytest.raises(nv.NvxError, match="integrity"):
                        p.fetch()
                    failures += 1
                else:
                    p.fetch()
                assert p.integrity_stats()[1] == failures, (s, w)
        p.reset()
        p.process_resident(buf, n, 0, F); p.process_resident(buf, n, F, F); p.fetch()
        assert p.integrity_stats()[0] == 0 and len(p.bits(5, 1)) > 40
    buf.free()


def test_the_seal_is_sensitive_to_position(nv):
    """Two entries swapped (same lane's slots: U and Y2; neighbouring lanes: Y2[k] and Y2[k+1]; across the halves of the
    wave: lanes 5 and 37), a block of the right stream but an EARLIER position (the block two launches old), and the block
    of ANOTHER stream at the same position: all fail."""
    p, buf, pitch = _handle(nv, False, [1, 1], frames=1, total=4)
    try:
        def run_to(n_launches):
            p.reset()
            for k in range(n_launches):
                p.process_resident(buf, pitch, k, 1)
            p.fetch()

        def expect_failure(what):
            p.process_resident(buf, pitch, 3, 1)
            with pytest.raises(nv.NvxError, match="integrity"):
                p.fetch()

        e = lambda entry: slice(2 * entry, 2 * entry + 2)
        u0, y0 = 36, 36 + 46                                  # chain 0: mixer outputs, FIR2 outputs
        for a, b in ((u0 + 9, y0 + 9), (y0 + 20, y0 + 21), (y0 + 5, y0 + 37)):
            run_to(3)
            blk = p.debug_state(0)
            assert not np.array_equal(blk[e(a)], blk[e(b)])
            t = blk[e(a)].copy(); blk[e(a)] = blk[e(b)]; blk[e(b)] = t
            p.debug_set_state(0, blk)
            expect_failure(f"entries {a} and {b} swapped")
        run_to(1); old = p.debug_state(0)                      # position 3 thirds
        run_to(3); p.debug_set_state(0, old)                   # ... where position 9 belongs
        expect_failure("a block two launches old")
        run_to(3); other = p.debug_state(1); mine = p.debug_state(0)
        assert int(other[W_TAG]) == (1 << 32) | 9 and int(mine[W_TAG]) == 9
        p.debug_set_state(0, other)
        expect_failure("another stream's block")
    finally:
        p.close(); buf.free()


def _run(env_extra, waiting_units=True):
    """tests/harness/integrity_run.py in a process of its own, hand-over form forced; waiting_units: a unit whose predecessor
    is still running WAITS for it (NVX_DYNAMIC_PREROLL=0), so that every unit but a stream's first takes a hand-over -- with
    a handful of streams the default (such a unit pre-rolls) leaves few real hand-overs."""
    env = dict(os.environ, NVX_INDEPENDENT="0", NVX_DYNAMIC_PREROLL="0" if waiting_units else "1", **env_extra)
    out = subprocess.run([sys.executable, str(ROOT / "tests" / "harness" / "integrity_run.py")], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    log = ROOT / "gpurun_out"
    if log.is_dir():
        with open(log / "integrity_runs.jsonl", "a") as f:
            f.write(json.dumps(dict(env=env_extra, waiting_units=waiting_units, **rec)) + "\n")
    return rec


def test_shipped_build_hands_over_cleanly():
    """Every kernel family with the hand-over form forced, against the oracle: bit-exact, and not one stale block."""
    for waiting in (True, False):
        rec = _run({}, waiting)
        assert all(c["ok"] for c in rec["cases"]), rec
        assert all(c["stale"] == 0 and c.get("failed", 0) == 0 for c in rec["cases"]), rec
        assert {c["kind"] for c in rec["cases"]} == {"resident", "list", "wideband_fused"}


def test_injected_stale_hand_overs_are_caught_repaired_and_counted():
    """The fault-injection build (-DNVX_INJECT_STALE=5: every fifth hand-over reads the block the stream's previous launch
    left): every kernel family still equals the oracle bit for bit -- complete 900 S/s output and bits -- and the seal
    counted the repairs, in the cascade kernels with and without a participant list and in the fused wideband kernel
    (state of one sub-band, the channeliser halo alone, everything)."""
    if not INJECT.exists():                                  # a fresh checkout on the GPU box: build it here, as conftest does for the product library
        import importlib.util
        spec = importlib.util.spec_from_file_location("nvx_build", ROOT / "navtex_amd" / "build.py")
        build = importlib.util.module_from_spec(spec); spec.loader.exec_module(build)
        build.build_variant("inject", build.INJECT_FLAGS)
    assert INJECT.exists(), "python navtex_amd/build.py --inject (__graft_entry__.build() does it)"
    rec = _run({"NAVTEX_AMD_LIB": str(INJECT)}, waiting_units=False)       # the default unit form: parity only
    assert all(c["ok"] and c.get("failed", 0) == 0 for c in rec["cases"]), rec
    rec = _run({"NAVTEX_AMD_LIB": str(INJECT)})
    assert all(c["ok"] for c in rec["cases"]), rec
    assert all(c.get("failed", 0) == 0 for c in rec["cases"]), rec
    for kind in ("resident", "wideband_fused"):
        assert all(c["stale"] > 0 for c in rec["cases"] if c["kind"] == kind), rec
    deep = [c for c in rec["cases"] if c["kind"] == "list" and c.get("deep")]
    assert len(deep) == 4 and all(c["stale"] > 0 and c["partial_launches"] > 0 for c in deep), rec


def test_a_failed_launch_poisons_the_handle_until_reset(nv, oracle):
    """A launch whose inherited state fails its seal does not only lose its own bits.  The unit that found the bad block ran
    on and sealed what it computed from it, the demodulator state moved on, and the launches already QUEUED behind it
    inherit both under valid seals -- their bits are garbage that nothing else would ever flag.  So the failure sticks:
    with three launches queued behind a corrupted block not one bit of any of them arrives; every later launch, push,
    poll, fetch and flush answers NVX_ERR_STATE; nvx_reset recovers, and the handle then decodes the same input bit-exactly."""
    n_frames = 8
    st, _ = signals.stream_params(nv, 3, nv.RATE_IN)
    iq = nv.synth_host(st, nv.RATE_IN, n_frames * nv.FRAME_IN)
    ref = oracle.Pipe(chain_mask=1, charlayer=False)
    ref.push(iq)
    buf = nv.DeviceBuffer(iq.nbytes)
    buf.upload(iq)
    got_msgs = []
    with nv.Pipeline(n_streams=1, raw_rate=False, chain_mask=nv.CHAIN_518, max_frames=2, push_mode=True, char_layer=True) as p:
        p.process_resident(buf, n_frames * nv.FRAME_IN, 0, 2); p.fetch()
        before = p.bit_count(0, 0)
        blk = p.debug_state(0)
        blk[40] ^= np.uint64(1) << np.uint64(17)
        p.debug_set_state(0, blk)
        for f in (2, 4, 6):                                    # three launches queued: the first inherits the bad block
            p.process_resident(buf, n_frames * nv.FRAME_IN, f, 2)
        with pytest.raises(nv.NvxError, match="integrity") as e:
            p.fetch()
        assert e.value.code == nv._native.ERR_HIP
        assert p.bit_count(0, 0) == before and p.integrity_stats()[1] == 1
        # everything answers "reset me" from here on, and still nothing arrives
        for call in (lambda: p.process_resident(buf, n_frames * nv.FRAME_IN, 0, 1), lambda: p.push(0, iq[:1000]), p.poll, p.fetch, p.flush, p.finish):
            with pytest.raises(nv.NvxError, match="nvx_reset") as e:
                call()
            assert e.value.code == nv._native.ERR_STATE
        assert p.bit_count(0, 0) == before
        p.reset()
        for f in (0, 2, 4, 6):
            p.process_resident(buf, n_frames * nv.FRAME_IN, f, 2)
        p.fetch()
        assert p.bits(0, 0) == ref.bits(0) and len(ref.bits(0)) > 150
        assert p.integrity_stats()[:2] == (0, 1)
    buf.free()
