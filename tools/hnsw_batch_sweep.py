"""Developer tool: HNSW batch-size sweep on the bench workload (31,173 x 768 manifold data, ef 100), for every
waves-per-query setting the launcher can pick.  usage: python tools/hnsw_batch_sweep.py [ef]"""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import numpy as np
    import torch

    import bench
    from hnsw_clj_amd import engine

    ef = int(sys.argv[2])
    base = bench.make_31k("manifold", 42, 31173)
    queries = bench.make_31k("manifold", 43, 10000)
    dev = torch.device("cuda", 0)
    idx = engine.Index(base, "cosine", 0)
    idx.hnsw_build(16, 200, 42)
    res = []
    for nq in (1, 8, 32, 128, 256, 512, 768, 1024, 1536, 2048, 3072, 4096, 10000):
        Q = torch.from_numpy(queries[:nq]).to(dev)
        out = (torch.empty((nq, 10), dtype=torch.int32, device=dev), torch.empty((nq, 10), dtype=torch.float32, device=dev))
        for _ in range(3):
            idx.hnsw_search_dev(Q, 10, ef, out=out)
        torch.cuda.synchronize()
        steps = 20 if nq <= 1024 else 8
        t0 = time.perf_counter()
        for _ in range(steps):
            idx.hnsw_search_dev(Q, 10, ef, out=out)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        res.append("%d:%.3fms/%.0fk" % (nq, dt * 1e3, nq / dt / 1e3))
    print("NW=%-4s " % os.environ.get("HNSWGPU_HNSW_NW", "auto") + "  ".join(res), flush=True)
else:
    ef = sys.argv[1] if len(sys.argv) > 1 else "100"
    for nw in (None, "1", "2", "4"):
        env = dict(os.environ)
        if nw:
            env["HNSWGPU_TUNE"] = "HNSW_NW=%s" % nw
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", ef], env=env, check=False)
