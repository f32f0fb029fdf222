import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from hnsw_clj_amd import engine
dev = torch.device("cuda", 0)
for dim, n, nlist in [(128, 1_000_000, 1024), (1536, 300_000, 512), (3072, 100_000, 256), (96, 500_000, 512)]:
    g = torch.Generator(device=dev); g.manual_seed(1)
    cen = torch.randn(nlist, dim, generator=g, device=dev)
    x = cen[torch.randint(0, nlist, (n,), generator=g, device=dev)] + 0.3 * torch.randn(n, dim, generator=g, device=dev)
    Q = cen[torch.randint(0, nlist, (2048,), generator=g, device=dev)] + 0.3 * torch.randn(2048, dim, generator=g, device=dev)
    for metric in ("l2", "cosine"):
        idx = engine.Index(x, metric, 0)
        idx.set_rejection_test(2)
        idx.ivf_build(nlist, 3, 42)
        out = {}
        for mode in (2, 0):
            idx.set_rejection_test(mode)
            res = {}
            for nq in (1, 32, 1024, 2048):
                q = Q[:nq].contiguous()
                for _ in range(2): idx.ivf_search_dev(q, 10, 16)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(5): r = idx.ivf_search_dev(q, 10, 16)
                torch.cuda.synchronize(); res[nq] = ((time.perf_counter() - t0) / 5 * 1e3, r[0].cpu().numpy(), r[1].cpu().numpy())
            out[mode] = res
        line = []
        for nq in (1, 32, 1024, 2048):
            a, b = out[2][nq], out[0][nq]
            same = np.array_equal(a[1], b[1]) and np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32))
            line.append("nq %d: %.3f vs %.3f ms%s" % (nq, a[0], b[0], "" if (same or metric != "l2") else " MISMATCH"))
        print("dim %d n %d %s: stream vs plain f32: %s" % (dim, n, metric, "; ".join(line)), flush=True)
        idx.close()
    del x
