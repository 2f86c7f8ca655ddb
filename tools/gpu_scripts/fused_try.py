import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, navtex_amd as nv
W, F = int(sys.argv[1]), int(sys.argv[2])
n = F * nv.FRAME_RAW
raw = np.random.default_rng(1).integers(-3000, 3000, size=(W, n, 2), dtype=np.int16)
buf = nv.DeviceBuffer(W * n * 4); buf.upload(raw)
print("created", flush=True)
with nv.Pipeline(n_streams=W, wideband=True, chain_mask=3, max_frames=F, char_layer=False) as p:
    t0 = time.time()
    p.process_resident(buf, n, 0, F)
    print("launched", flush=True)
    try:
        p.fetch()
        print("fetched", time.time() - t0, [len(p.bits(s, 0)) for s in range(min(8 * W, 8))], flush=True)
    except Exception as e:
        print("ERR", e, flush=True)
