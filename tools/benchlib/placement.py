"""Host placement for bench.py: which CPUs a rank may use (NUMA node of its GPU, cgroup quota, physical cores).
No GPU call in here: the affinity must be in place before the HIP runtime starts its own threads."""
from __future__ import annotations

import glob
import os

# ----------------------------------------------------------------------------- host placement (no GPU call in here)
def _parse_cpulist(text: str) -> set:
    out = set()
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.update(range(int(a), int(b or a) + 1))
    return out


def _gpu_numa(dev: int):
    """(numa node, local CPU set) of HIP device `dev`, from the KFD topology in sysfs; (None, None) when unknown.
    No HIP call: the affinity must be in place before the runtime starts its own threads."""
    try:
        vis = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")
        if vis and all(v.strip().isdigit() for v in vis.split(",")):
            dev = int(vis.split(",")[dev])
        gpus = []
        for node in sorted(glob.glob("/sys/class/kfd/kfd/topology/nodes/*"), key=lambda p: int(os.path.basename(p))):
            props = dict(line.split(None, 1) for line in open(os.path.join(node, "properties")).read().splitlines() if " " in line)
            if int(props.get("simd_count", "0")) > 0:
                gpus.append(props)
        minor = int(gpus[dev]["drm_render_minor"])
        base = f"/sys/class/drm/renderD{minor}/device"
        return int(open(base + "/numa_node").read()), _parse_cpulist(open(base + "/local_cpulist").read())
    except Exception:
        return None, None


def place_rank(device: int, local_world: int, local_rank: int):
    """Bind this process to the CPUs of its GPU's NUMA node and size its host thread pools from its share of them."""
    numa, cpus = _gpu_numa(device)
    have = os.sched_getaffinity(0)
    bound = False
    if cpus:
        both = have & cpus
        if both:
            try:
                os.sched_setaffinity(0, both); have = both; bound = True
            except OSError:
                pass
    # ranks of this node that share the NUMA node (devices are dealt in rank order)
    sharing = 1
    if local_world > 1:
        mine = numa
        sharing = sum(1 for r in range(local_world) if _gpu_numa(r)[0] == mine) if mine is not None else local_world
        sharing = max(1, sharing)
    threads = max(1, min(16, len(have) // sharing))
    # one GPU of the pool's boxes comes with a 16-thread CPU share whatever the affinity mask says
    threads = int(os.environ.get("NVX_CPU_THREADS", threads))
    os.environ.setdefault("NVX_HOST_THREADS", str(threads))            # the library's character-layer pool
    return {"numa_node": numa, "bound": bound, "cpus": len(have), "ranks_on_numa_node": sharing, "threads": threads}


def physical_cores(have) -> tuple:
    """(physical cores among the CPUs of `have`, hardware threads per core) from the sysfs topology."""
    seen, smt = set(), 1
    for c in sorted(have):
        try:
            sib = _parse_cpulist(open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read())
        except (OSError, ValueError):
            sib = {c}
        smt = max(smt, len(sib))
        seen.add(min(sib))
    return max(1, len(seen)), smt


def cpu_quota():
    """CPUs the cgroup lets this process use at once (cpu.max), or None when unlimited / unknown."""
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            f = open(path).read().split()
            if path.endswith("cpu.max"):
                return None if f[0] == "max" else round(int(f[0]) / int(f[1]), 2)
            q = int(f[0])
            return None if q <= 0 else round(q / int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()), 2)
        except (OSError, ValueError, IndexError):
            continue
    return None


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"
