"""bench.py's process plumbing: starting the ranks as a child (`python bench.py --gpus N` without a launcher), the few
collectives the benchmark needs over the ranks, and the way a run ends (line on rank 0, exit status 3 / 4)."""
from __future__ import annotations

import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]


def self_launch(args, script=None, argv=None) -> None:
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD (torch.distributed.run, one process per
    GPU), relay its one JSON line and its exit status.  Runs before this process has imported torch or made any GPU
    call: a process that has touched the GPU must never exec, and this one neither touches it nor execs."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC only on this pool (RCCL needs it)
    env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script or ROOT / "bench.py")] + list(sys.argv[1:] if argv is None else argv)
    print("bench.py: launching " + " ".join(cmd[2:8]) + " ...", file=sys.stderr, flush=True)
    # the ranks in a session of their own: a SIGTERM / SIGINT / SIGHUP that reaches only this process (a driver's timeout
    # that is not a process-group kill) is passed on to all of them -- they must not be left holding the GPUs
    import signal
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)      # stderr goes straight through

    def pass_on(signum, _frame):
        try:
            os.killpg(child.pid, signal.SIGTERM)
            try:
                child.wait(timeout=10)
            except subprocess.TimeoutExpired:
                os.killpg(child.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        sys.exit(128 + signum)

    for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        signal.signal(sig, pass_on)
    out, _ = child.communicate()
    lines = out.splitlines()
    js = [l for l in lines if l.startswith("{")]
    for l in lines:
        if not js or l is not js[-1]:
            print(l, file=sys.stderr)
    if js:
        print(js[-1], flush=True)
    elif child.returncode == 0:
        raise SystemExit("bench.py: the ranks printed no JSON line")
    sys.exit(child.returncode)


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

    def gather(self, value: float) -> list:
        """`value` of every rank, in rank order."""
        if self.dist is None:
            return [value]
        dev = f"cuda:{self.device}" if self.backend == "nccl" else "cpu"
        t = self.torch.tensor([value], dtype=self.torch.float64, device=dev)
        out = [self.torch.zeros_like(t) for _ in range(self.dist.get_world_size())]
        self.dist.all_gather(out, t)
        return [float(o.item()) for o in out]

    def describe(self, rank_ms: float, rank_checked: int, rank_casc_ms: float, device: int) -> dict:
        """What the job looked like from the ranks: the world size the backend actually formed, stragglers, who checked what."""
        ms = self.gather(rank_ms)
        return {"world_size_seen": self.dist.get_world_size() if self.dist is not None else 1,
                "backend": ("rccl" if self.backend == "nccl" else self.backend) if self.dist is not None else "none",
                "ms_per_step_per_rank": [round(v, 3) for v in ms], "ms_per_step_min": round(min(ms), 3), "ms_per_step_max": round(max(ms), 3),
                "parity_streams_checked_per_rank": [int(v) for v in self.gather(float(rank_checked))],
                "cascade_avg_launch_ms_per_rank": [round(v, 3) for v in self.gather(rank_casc_ms)],
                "device_per_rank": [int(v) for v in self.gather(float(device))]}

    def reduce(self, value: float, op: str) -> float:
        if self.dist is None:
            return value
        t = self.torch.tensor([value], dtype=self.torch.float64, device=f"cuda:{self.device}" if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op={"max": self.dist.ReduceOp.MAX, "min": self.dist.ReduceOp.MIN, "sum": self.dist.ReduceOp.SUM}[op])
        return float(t.item())


def finish(line, parity, ranks, rank, leg_errors=False):
    """Print the line on rank 0; a failed parity is a failed run (exit status 3) on every rank, a side leg that raised
    (named in legs_failed) one with status 4."""
    if rank == 0:
        print(json.dumps(line), flush=True)
    if ranks.dist is not None:
        ranks.dist.barrier()
        ranks.dist.destroy_process_group()
    if not parity:
        sys.exit(3)
    if leg_errors:
        print(f"bench.py: side legs failed: {line.get('legs_failed')} (the line above is complete otherwise; --allow-leg-errors to pass)", file=sys.stderr)
        sys.exit(4)
