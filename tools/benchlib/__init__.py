"""Parts of bench.py (repo root) by concern: roofline arithmetic, host placement, process plumbing, the CPU baseline, the
side legs.  bench.py imports them and re-exports their names; the headline run itself lives there."""
