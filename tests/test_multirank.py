"""Multi-process path on CPU (gloo, world size 2): the pieces of bench.py that are
about ranks -- stream sharding (one subset per rank, no data-path collective),
max-over-ranks timing and whole-job aggregation.  The per-rank "hot path" here is
the oracle (this is a CPU test of the orchestration, not of the kernels)."""
import os
import socket
import subprocess
import sys
import textwrap
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent

WORKER = textwrap.dedent("""
    import os, sys, json, time
    sys.path.insert(0, {root!r}); sys.path.insert(0, {root!r} + "/tests")
    import numpy as np
    import torch, torch.distributed as dist
    import navtex_amd as nv, oracle_binding as ob, signals
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    S, frames = 3, 3                                   # streams per rank (weak scaling)
    bits = {{}}
    t0 = time.perf_counter()
    for s in range(S):
        gid = rank * S + s                             # same sharding rule as bench.py
        st, _ = signals.stream_params(nv, gid, nv.RATE_IN)
        iq = nv.synth_host(st, nv.RATE_IN, frames * nv.FRAME_IN)
        p = ob.Pipe(chain_mask=1, charlayer=False); p.push(iq)
        bits[gid] = p.bits(0)
    elapsed = time.perf_counter() - t0 + 0.01 * rank   # make the ranks differ
    dist.barrier()
    t = torch.tensor([elapsed], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    gathered = [None] * world
    dist.all_gather_object(gathered, bits)
    if rank == 0:
        merged = {{}}
        for g in gathered: merged.update(g)
        print(json.dumps({{"max_elapsed": float(t.item()), "own_elapsed": elapsed, "ids": sorted(merged), "bits": merged,
                          "value": world * S * frames * nv.FRAME_IN / float(t.item())}}))
    dist.destroy_process_group()
""")


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_shard_streams_without_collective(tmp_path, nv, oracle):
    import json
    import signals
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=str(ROOT)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", str(free_port()), str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["ids"] == ["0", "1", "2", "3", "4", "5"] or rec["ids"] == [0, 1, 2, 3, 4, 5]
    assert rec["max_elapsed"] >= rec["own_elapsed"]          # MAX over ranks, not rank 0's own time
    # every global stream decodes exactly as a single-process run decodes it
    for gid in range(6):
        st, _ = signals.stream_params(nv, gid, nv.RATE_IN)
        p = oracle.Pipe(chain_mask=1, charlayer=False)
        p.push(nv.synth_host(st, nv.RATE_IN, 3 * nv.FRAME_IN))
        assert rec["bits"][str(gid)] == p.bits(0)
    # (all six are still inside the phasing preamble after 3 frames, so the bit strings
    # are shifted copies of one pattern; what matters is that each id was decoded by
    # exactly one rank and matches the single-process result above)
    assert len(rec["bits"]) == 6


def test_bench_refuses_multi_gpu_without_launcher():
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300,
                         env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert out.returncode != 0 and "torch.distributed.run" in (out.stderr + out.stdout)
