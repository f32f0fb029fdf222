#!/usr/bin/env python3
"""bench.py -- the reference's headline benchmark on MI355X.

Metric (BASELINE.json): QPS @ recall@10 >= 0.98 on 31,173 x 768 float32, k = 10 (HNSW, M=16,
ef_construction=200 -- the protocol of src/hnsw/wip/reproduce_02ms.clj), plus the IVF-FLAT list scan's
achieved HBM GB/s against the roofline (1M x 768, nlist=1024, nprobe=32).

A "step" is one pass of the hot path over one batch of `--nq` queries already resident in HBM:
one hnswgpu_hnsw_search_dev launch.  With N > 1 every rank holds a replica of the 31k index and its
own query batch (the 31k x 768 index fits one GPU, so the queries are what shard: "replicas", no
collective on the data path); value = all ranks' queries / max-over-ranks time.  The row-sharded
configurations (configs[3]: ONE IVF index whose lists are dealt to the GPUs; configs[4]: one HNSW
sub-graph per GPU) run beside it for N > 1 and are reported in `sharded_ivf` / `sharded_hnsw`.

`python bench.py --gpus N` starts its N ranks itself (child processes with RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* set; the parent never touches the GPU and relays rank 0's line); under
torch.distributed.run the ranks are already there and it just runs as one of them.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
# MI355X_MICROARCH.md "Indexed rows": uniformly random ~1 KB rows of a table that fits the 256 MiB Infinity Cache are
# gathered at 8.6 TB/s (38 MB table) to 7.4-7.9 TB/s (151 MB); random whole rows of a table far beyond it, gathered
# into registers, at 5.5-5.8 TB/s.  These, not the HBM spec, bound the traversal's row gather.
IC_GATHER_GBS = (7400.0, 8600.0)
HBM_GATHER_GBS = (5500.0, 5800.0)
N31K, DIM, K = 31173, 768, 10
M, EFC = 16, 200
# the reference's own ef (max(k, 50), ultra_fast.clj:355) first, then steps of ~4 % so that the operating point is the
# smallest ef that meets the recall bar, not the next power-of-two-ish value above it
EF_SWEEP = [50, 56, 64, 72, 80, 88, 96, 100, 104, 108, 112, 116, 120, 128, 136, 144, 152, 160, 176, 192, 224, 256, 320, 384,
            448, 512, 576, 640, 672, 704, 736, 768, 800, 832, 896, 1024, 1280, 1536, 2048, 3072, 4096]
# The index builder (hnswgpu_hnsw_build_ex): "heuristic" = neighbour selection by hnsw.hnsw-search's
# get-neighbors-heuristic (src/hnsw/graph.clj:162-198) for a node's links and for an over-full list, the dropped edge leaving
# the pruned list only (as prune-connections-ultra, ultra_fast.clj:279-299, leaves the reverse edge); "graph.clj" = the
# same with the dropped edge removed from BOTH lists (prune-connections, graph.clj:208-232); "ultra_fast.clj" = the m
# closest (ultra_fast.clj:216-299).  Both namespaces search with the same search-layer / search-knn algorithm -- the
# traversal kernel.  One-sided pruning keeps more ways INTO a cluster: on the 31k clustered set 0.98 at ef 704 against
# 736 for graph.clj, on configs[4]'s 1.25M x 1536 clustered rows 0.991 at ef 256 against a plateau of 0.979
# (profiles/r04_config5_hnsw_shard.txt); graph.clj is ahead on i.i.d. uniform rows (by_distribution).
BUILDERS = {"heuristic": dict(heuristic=True), "graph.clj": dict(heuristic=True, symmetric=True), "ultra_fast.clj": {},
            "heuristic+extend": dict(heuristic=True, extend=True)}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def make_31k(distribution, seed, n):
    """S1 of BASELINE.md: java.util.Random-compatible generator (test/data_generator.clj).  'clustered' =
    256 gaussian centres, noise 0.3, L2-normalised (the stand-in for the normalised sentence
    embeddings the reference's published run used); 'gaussian' = i.i.d. N(0,1)."""
    from hnsw_clj_amd import datagen

    if distribution == "manifold":
        # Latent-manifold stand-in for the normalised sentence embeddings of the published run (the real
        # data set is not in the repository): x = normalise(W z / sqrt(r) + 0.1 e), z in R^32, all draws from
        # the java.util.Random-compatible generator.  Intrinsic dimension ~32, no isolated clusters.
        r = 32
        z = datagen.generate_dataset(n, r, seed=seed, dtype=np.float64)
        w = datagen.generate_dataset(r, DIM, seed=7, dtype=np.float64)
        e = datagen.generate_dataset(n, DIM, seed=seed + 1000, dtype=np.float64)
        x = z @ w / np.sqrt(r) + 0.1 * e
        x /= np.linalg.norm(x, axis=1, keepdims=True)
        return x.astype(np.float32)
    if distribution == "clustered":
        x = datagen.generate_dataset(n, DIM, "clustered", num_clusters=256, noise_level=0.3, seed=seed,
                                     dtype=np.float64)
        x /= np.linalg.norm(x, axis=1, keepdims=True)
        return x.astype(np.float32)
    if distribution == "clustered_same_mixture":
        # base AND held-out queries from ONE mixture (the generator draws its centres from the seed: seed-43 queries of
        # 'clustered' point at other centres than the base's): rows [0, N31K) of the seed-42 stream are the base, the
        # rows behind them the queries
        x = datagen.generate_dataset(N31K + (0 if seed == 42 else n), DIM, "clustered", num_clusters=256, noise_level=0.3,
                                     seed=42, dtype=np.float64)
        x /= np.linalg.norm(x, axis=1, keepdims=True)
        return (x[:n] if seed == 42 else x[N31K:N31K + n]).astype(np.float32)
    if distribution == "uniform01":
        # what the published run and the hnswlib script used: (rand) / np.random.rand, i.i.d. uniform on [0, 1)
        # (wip/ultra_optimized.clj:293-296, scripts/benchmark_python_hnswlib.py:31) -- here java.util.Random.nextDouble
        return datagen.JavaRandom(seed).next_doubles(n * DIM).reshape(n, DIM).astype(np.float32)
    if distribution == "uniform_pm1":      # the generator's own :uniform, 2 (rand) - 1 on [-1, 1) (test/data_generator.clj:71)
        distribution = "uniform"
    return datagen.generate_dataset(n, DIM, distribution, seed=seed)


def recall_at_k(ids, truth):
    """bench.clj:86-92 calc-recall on id sets, averaged over queries (torch, on device)."""
    hit = (ids.unsqueeze(2) == truth.unsqueeze(1)).any(dim=2).sum(dim=1).float()
    return float((hit / truth.shape[1]).mean())


def find_ef(idx, Q, truth, ef_arg):
    """The operating point: the first ef of EF_SWEEP whose recall@10 against `truth` reaches 0.98 (or --ef)."""
    sweep = []
    ef, rec = ef_arg, None
    if ef <= 0:
        for e in EF_SWEEP:
            ids, _ = idx.hnsw_search_dev(Q, K, e)
            r = recall_at_k(ids, truth)
            sweep.append([e, round(r, 4)])
            if r >= 0.98:
                ef, rec = e, r
                break
        if ef <= 0:
            ef, rec = EF_SWEEP[-1], sweep[-1][1]
    else:
        ids, _ = idx.hnsw_search_dev(Q, K, ef)
        rec = recall_at_k(ids, truth)
    return ef, rec, sweep


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: N child processes, one rank per GPU.  This process has not
    touched the GPU (importing torch does not) and never will: it waits, relays rank 0's JSON line and returns the
    worst exit code."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    line, _ = procs[0].communicate()
    rc = procs[0].returncode
    deadline = time.time() + 120
    for p in procs[1:]:
        try:
            p.wait(timeout=max(1.0, deadline - time.time()))
        except subprocess.TimeoutExpired:      # rank 0 is gone and a peer still waits in a collective
            p.kill()
            p.wait()
        rc = rc or p.returncode
    sys.stdout.write(line.decode())
    sys.stdout.flush()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nq", type=int, default=10000, help="queries per step per GPU")
    ap.add_argument("--ef", type=int, default=0, help="0 = first ef of the sweep with recall@10 >= 0.98")
    ap.add_argument("--dist", default="clustered",
                    choices=["clustered", "gaussian", "uniform01", "uniform_pm1", "clustered_same_mixture", "manifold"],
                    help="31k x 768 data set; BASELINE.md section 3 names gaussian, clustered-normalised and uniform[0,1)")
    ap.add_argument("--builder", default="heuristic", choices=sorted(BUILDERS),
                    help="index builder: graph.clj's heuristic selection (one-sided pruning; 'graph.clj' = pruned on both "
                         "sides) or hnsw.ultra-fast's closest-m")
    ap.add_argument("--no-ivf", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--ivf-n", type=int, default=1_000_000)
    ap.add_argument("--sharded", action="store_true", help="also run the row-sharded IVF search (always on for N > 1)")
    ap.add_argument("--shard-rows", type=int, default=1_250_000, help="rows per GPU of the sharded IVF index")
    ap.add_argument("--hnsw-shard-rows", type=int, default=1_250_000,
                    help="rows per GPU of the sharded 1536-d HNSW index (0 = skip that leg)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 PMC passes that fill roofline.traffic")
    ap.add_argument("--no-dists", action="store_true", help="skip the gaussian / uniform / clustered operating points")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--launch-selftest", action="store_true", help=argparse.SUPPRESS)   # tests/: ranks report and exit
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.pmc_child:
        sys.exit(launch_ranks(args.gpus))
    if args.launch_selftest:     # no GPU needed: what a rank sees of the launcher's environment
        if int(os.environ.get("RANK", "0")) == 0:
            print(json.dumps({k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR")}))
        sys.exit(3 if os.environ.get("HNSWGPU_SELFTEST_FAIL_RANK") == os.environ.get("RANK", "0") else 0)
    # stdout carries exactly ONE JSON line: libraries that print banners to fd 1 (RCCL's version banner
    # does) are sent to stderr for the duration of the run
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.pmc_child:
        return pmc_child(args)
    traffic = None
    # PMC passes are child processes and start BEFORE this process touches the GPU.  Not under a profiler: its
    # preloaded library has already initialised the GPU here, and a profiler inside a profiler measures neither.
    profiled = any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
    if world == 1 and not args.no_ivf and not args.no_pmc and not profiled:
        traffic = pmc_traffic(args)
    ndev = max(torch.cuda.device_count(), 1)     # rehearsals may put several ranks on one GPU (local_rank >= ndev)
    torch.cuda.set_device(local_rank % ndev)
    dev = torch.device("cuda", local_rank % ndev)
    if world > 1 or args.sharded:
        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", "29517"
        backend = os.environ.get("HNSWGPU_BENCH_BACKEND", "nccl")   # "gloo": rehearse N ranks on fewer GPUs (RCCL refuses)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if world != args.gpus:
        log("[rank %d] --gpus %d but WORLD_SIZE %d: running with the %d ranks the launcher started" % (rank, args.gpus, world, world))

    from hnsw_clj_amd import engine

    # ------------------------------------------------------------------ HNSW 31k x 768 (configs[1])
    t0 = time.time()
    base = make_31k(args.dist, 42, N31K)
    queries = make_31k(args.dist, 43 + rank, args.nq)            # held-out, per-rank batch
    log("[rank %d] data %.1fs" % (rank, time.time() - t0))
    idx = engine.Index(base, "cosine", dev.index)
    t0 = time.time()
    idx.hnsw_build(M, EFC, 42, **BUILDERS[args.builder])
    build_s = time.time() - t0
    log("[rank %d] hnsw build on device %.2fs" % (rank, build_s))
    Q = torch.from_numpy(queries).to(dev)
    n_eval = args.nq                                              # recall is measured on the whole timed batch
    truth, _ = idx.exact_knn_dev(Q[:n_eval], K)                   # ground truth over the FULL base (bench.clj:72-84)
    ef, rec, sweep = find_ef(idx, Q[:n_eval], truth, args.ef)
    if world > 1:  # every rank times the same ef (rank 0's)
        t = torch.tensor([ef], device=dev)
        dist.broadcast(t, 0)
        ef = int(t.item())
    # reference protocol: the first 100 base rows as queries (reproduce_02ms.clj:38)
    Q100 = torch.from_numpy(base[:100]).to(dev)
    t100, _ = idx.exact_knn_dev(Q100, K)
    i100, _ = idx.hnsw_search_dev(Q100, K, ef)
    rec100 = recall_at_k(i100, t100)

    out_ids = torch.empty((args.nq, K), dtype=torch.int32, device=dev)
    out_d = torch.empty((args.nq, K), dtype=torch.float32, device=dev)
    stats = torch.zeros((args.nq, 2), dtype=torch.int64, device=dev)
    idx.hnsw_search_dev(Q, K, ef, out=(out_ids, out_d), stats=stats)
    torch.cuda.synchronize()
    evals = float(stats[:, 0].double().mean())
    hops = float(stats[:, 1].double().mean())

    for _ in range(args.warmup):
        idx.hnsw_search_dev(Q, K, ef, out=(out_ids, out_d))
    idx.set_profiling(True)
    idx.get_profile(engine.PROF_HNSW, reset=True)
    idx.rejection_stats(reset=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        idx.hnsw_search_dev(Q, K, ef, out=(out_ids, out_d))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    f32_rows, neighbours = idx.rejection_stats(reset=True)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_ms, kern_n = idx.get_profile(engine.PROF_HNSW, reset=True)
    idx.set_profiling(False)
    qps = world * args.nq * args.steps / elapsed
    # single-query latency (configs[1] "single-query latency"): one query per launch
    torch.cuda.synchronize()
    lat = []
    for i in range(50):
        t1 = time.perf_counter()
        idx.hnsw_search_dev(Q[i:i + 1], K, ef, out=(out_ids[:1], out_d[:1]))
        torch.cuda.synchronize()
        lat.append((time.perf_counter() - t1) * 1e3)
    lat = sorted(lat[5:])
    # ... and through the reference's own seam: search-knn takes and returns HOST arrays (ultra_fast.clj:346-374)
    lat_h = []
    for i in range(60):
        t1 = time.perf_counter()
        idx.hnsw_search(queries[i], K, ef)
        lat_h.append((time.perf_counter() - t1) * 1e3)
    lat_h = sorted(lat_h[10:])

    # algorithmic bytes of the traversal: E * 4*D + H * 4*M0 per query -- what the reference's algorithm
    # reads.  The kernel decides most neighbours from int8 rows (the rejection test, kernels.hpp) and fetches f32 rows
    # only for those that may be admitted: the bytes it really requests per query are counted, not assumed.
    hnsw_bytes_q = evals * 4 * DIM + hops * 4 * (2 * M)
    hnsw_avg_ms = kern_ms / max(kern_n, 1)
    hnsw_algo_gbs = hnsw_bytes_q * args.nq / (hnsw_avg_ms * 1e-3) / 1e9
    per_q = 1.0 / max(args.nq * args.steps, 1)
    f32_q, nb_q = f32_rows * per_q, neighbours * per_q
    code_row = 256 * ((DIM + 255) // 256)                        # int8 row: 64 lanes x NCH dwords
    tested = f32_q < 0.98 * nb_q                                 # the rejection test ran (it is off for small launches)
    moved_q = (nb_q * (code_row + 16) if tested else 0.0) + f32_q * (4 * DIM + 4) + hops * 4 * (2 * M)
    hnsw_gbs = moved_q * args.nq / (hnsw_avg_ms * 1e-3) / 1e9

    result = {
        "metric": "QPS @ recall@10>=0.98 (31k x 768, k=10); IVF scan achieved HBM GB/s vs roofline",
        "value": round(qps, 1),
        "unit": "queries/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "hnsw search-knn (ultra_fast.clj:346-374 = graph.clj:297-320), 31,173 x 768 f32 (%s, java.util.Random "
                        "seed 42; queries: the same generator, seed 43), k=10, M=16 ef_construction=200, index built on the "
                        "device by the %s builder, batched HIP traversal kernel" % (args.dist, args.builder),
            "builder": args.builder,
            "queries_per_step_per_gpu": args.nq,
            "ef_search": ef,
            "recall_at_10": round(rec, 4),
            "recall_at_10_first100_base_rows": round(rec100, 4),
            "ef_sweep": sweep,
            "parallelism": "replicas x%d (index replicated, queries sharded, no collective)" % world,
            "hnsw_build_s": round(build_s, 2),
            "dist_evals_per_query": round(evals, 1),
            "expansions_per_query": round(hops, 1),
            "single_query_latency_ms": {"p50": round(lat_h[len(lat_h) // 2], 4), "min": round(lat_h[0], 4),
                                        "p95": round(lat_h[int(len(lat_h) * 0.95)], 4),
                                        "path": "hnswgpu_hnsw_search, host buffers in and out (the reference's search-knn seam): "
                                                "query and results in mapped pinned memory, one launch, host-polled completion"},
            "launch_paths": {k: engine.debug_counter(k) for k in ("hnsw_wave", "hnsw_solo", "hnsw_helpers", "hnsw_rejection", "hnsw_plain")},
            "hnsw_rejection_state": dict(zip(("state", "int8_test_off", "f32_rows_per_neighbour"), idx.hnsw_rejection_state())),
            "single_query_latency_dev_ms": {"p50": round(lat[len(lat) // 2], 4), "min": round(lat[0], 4),
                                            "p95": round(lat[int(len(lat) * 0.95)], 4),
                                            "path": "hnswgpu_hnsw_search_dev on torch's stream + torch.cuda.synchronize()"},
        },
        "roofline_hnsw": {"bound": "infinity-cache gather", "achieved": round(hnsw_gbs, 1), "peak": IC_GATHER_GBS[1],
                          "unit": "GB/s", "frac": round(hnsw_gbs / IC_GATHER_GBS[1], 4),
                          "traffic": (traffic or {}).get("hnsw", None) and traffic["hnsw"]["traffic"],
                          "traffic_GBs": (traffic or {}).get("hnsw", None) and round(traffic["hnsw"]["traffic"] / 1e9 / (hnsw_avg_ms * 1e-3), 1),
                          "traffic_note": "fabric-side bytes of one timed launch (the traversal + its repeat pass): 2 x FETCH_SIZE "
                                          "+ WRITE_SIZE from rocprofv3 --pmc passes of a child process that builds the same index and "
                                          "issues the same launch (the factor 2 is the guide's gfx950 correction, calibrated for "
                                          "16-B-per-lane reads -- the f32 rows; the int8 rows are read 12 B per lane, uncalibrated: "
                                          "raw_fetch_size_bytes is beside it); below the requested bytes where the XCD L2s serve "
                                          "re-reads",
                          "raw_fetch_size_bytes": (traffic or {}).get("hnsw", None) and traffic["hnsw"]["raw_fetch_size_bytes"],
                          "peak_range": list(IC_GATHER_GBS), "frac_of_hbm_spec": round(hnsw_gbs / HBM_PEAK_GBS, 4),
                          "kernel": "hnsw_wave_kernel (one wave per query: main list in LDS + register-resident admission buffer, "
                                    "wave_kernels.hpp) -- %d of the launches of this process; hnsw_search_kernel serves the repeat "
                                    "pass and small launches" % engine.debug_counter("hnsw_wave"),
                          "avg_launch_ms": round(hnsw_avg_ms, 4),
                          "algorithmic_bytes_per_query": int(hnsw_bytes_q),
                          "algorithmic_GBs": round(hnsw_algo_gbs, 1),
                          "requested_bytes_per_query": int(moved_q),
                          "f32_rows_per_query": round(f32_q, 1), "neighbours_per_query": round(nb_q, 1),
                          "note": "achieved = bytes the kernel REQUESTS per launch / its duration: every evaluated neighbour's "
                                  "int8 row (+ 16 B of per-row scalars), an f32 row only for the neighbours whose lower bound "
                                  "does not already exclude them from the result list (counted on the device), and the "
                                  "adjacency rows.  algorithmic_GBs prices the same launch at the reference algorithm's bytes "
                                  "(every neighbour an f32 row) and may exceed any ceiling.  31,173 x 768: both copies (95.8 + "
                                  "23.9 MB) are Infinity-Cache resident, so the bound is the guide's measured ceiling for "
                                  "random ~1 KB rows of a 38-151 MB table (7.4-8.6 TB/s), not HBM; the HBM-resident traversal "
                                  "(1.25M x 1536, configs[4]) is the sharded_hnsw leg / tests, against 5.5-5.8 TB/s"},
    }
    result["config"]["qps_host_buffers"] = host_buffer_qps(idx, queries, ef, args.steps)
    if rank == 0 and world == 1 and not args.no_dists:
        result["config"]["by_distribution"] = by_distribution(engine, dev, args, not args.no_cpu)

    # ------------------------------------------------------------------ IVF-FLAT scan roofline (configs[2])
    if not args.no_ivf and rank == 0:
        result["roofline"] = ivf_roofline(engine, dev, args, traffic)
    elif rank == 0:
        result["roofline"] = result["roofline_hnsw"]

    # ------------------------------------------------------------------ row-sharded IVF over RCCL (configs[3])
    if world > 1 or args.sharded:
        try:
            sh = sharded_ivf(engine, dev, rank, world, args)
        except Exception as e:  # never let the secondary measurement take the headline number down
            sh = {"error": "%s: %s" % (type(e).__name__, e)}
        if rank == 0:
            result["sharded_ivf"] = sh
        if args.hnsw_shard_rows > 0:
            try:
                sh = sharded_hnsw(engine, dev, rank, world, args)
            except Exception as e:  # same guard: a secondary leg never takes the headline number down
                sh = {"error": "%s: %s" % (type(e).__name__, e)}
            if rank == 0:
                result["sharded_hnsw"] = sh

    # ------------------------------------------------------------------ the reference's own protocol (rank 0, N = 1)
    if rank == 0 and world == 1:
        result["config"]["reference_protocol"] = reference_protocol(ef)

    # ------------------------------------------------------------------ CPU baseline (rank 0, N = 1 only)
    if not args.no_cpu and rank == 0 and world == 1:
        result["cpu_baseline"] = cpu_baseline(idx, base, queries, ef)
        result["config"]["parity_checked"] = parity_check(idx, base, queries, ef, out_ids, out_d)
    idx.close()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        os.write(real_stdout, (json.dumps(result) + "\n").encode())


def host_buffer_qps(idx, queries, ef, steps):
    """The reference's own seam -- search-batch* takes and returns HOST arrays (api/protocol.clj:58-67): the same batch
    through hnswgpu_hnsw_search (PCIe upload of nq x 768 floats, launch, download of ids + distances).  Never `value`."""
    steps = max(3, min(steps, 10))
    idx.hnsw_search(queries, K, ef)
    t0 = time.perf_counter()
    for _ in range(steps):
        idx.hnsw_search(queries, K, ef)
    return round(len(queries) * steps / (time.perf_counter() - t0), 1)


def by_distribution(engine, dev, args, want_cpu):
    """SURVEY 8(d) S1, every set beside the headline's: gaussian (primary in the survey), clustered-normalised (queries by
    the same generator with seed 43: other centres than the base's), uniform01 = i.i.d. uniform on [0, 1) (what the
    published run and benchmark_python_hnswlib.py:31 used), uniform_pm1 = the generator's own :uniform on [-1, 1), plus
    clustered_same_mixture (held-out rows of the base's own mixture) and the latent-manifold set of rounds 1-3.  Each: build
    on the device by the run's builder, ground truth by GPU brute force over the full base, then the FIRST ef of the sweep
    whose recall@10 reaches 0.98 -- or the last point of the sweep if none does (ef <= 4096 is the kernel's limit).  Beside
    every operating point the two numbers that say what it is worth (bench.clj:72-92 defines both sides): exact_knn_qps =
    the same queries answered at recall 1.0 by the GPU's exact scan (hnswgpu_exact_knn_dev), cpu_qps = the CPU oracle (f64
    reference order) on the same graph at the same ef; closest_m = the same sweep on hnsw.ultra-fast's closest-m graph."""
    out = {}
    sweep = [50, 100, 200, 400, 600, 800, 1200, 1600, 2400, 3200, 4096]
    nq = min(args.nq, 4096)

    def run_sweep(idx, Q, truth):
        pts, hit = [], None
        for e in sweep:
            ids, _ = idx.hnsw_search_dev(Q, K, e)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(3):
                idx.hnsw_search_dev(Q, K, e)
            torch.cuda.synchronize()
            qps = 3 * nq / (time.perf_counter() - t1)
            r = recall_at_k(ids, truth)
            pts.append([e, round(r, 4), round(qps, 1)])
            if r >= 0.98:
                hit = pts[-1]
                break
        return pts, hit

    for name in ("gaussian", "clustered", "uniform01", "uniform_pm1", "clustered_same_mixture", "manifold"):
        t0 = time.time()
        base = make_31k(name, 42, N31K)
        qh = make_31k(name, 43, nq)
        Q = torch.from_numpy(qh).to(dev)
        with engine.Index(base, "cosine", dev.index) as idx:
            truth, _ = idx.exact_knn_dev(Q, K)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(3):
                idx.exact_knn_dev(Q, K)
            torch.cuda.synchronize()
            exact_qps = 3 * nq / (time.perf_counter() - t1)
            others = {}
            for b in ("ultra_fast.clj", "graph.clj", "heuristic"):
                if b == args.builder:
                    continue
                idx.hnsw_build(M, EFC, 42, **BUILDERS[b])
                opts, ohit = run_sweep(idx, Q, truth)
                of = ohit or opts[-1]
                others[b] = {"ef": of[0], "recall_at_10": of[1], "qps": of[2], "reached_0.98": ohit is not None}
            tb = time.time()
            idx.hnsw_build(M, EFC, 42, **BUILDERS[args.builder])
            tb = time.time() - tb
            pts, hit = run_sweep(idx, Q, truth)
            first = hit or pts[-1]
            out[name] = {"builder": args.builder, "ef": first[0], "recall_at_10": first[1], "qps": first[2],
                         "reached_0.98": hit is not None, "build_s": round(tb, 2),
                         "exact_knn_qps": round(exact_qps, 1), "sweep_ef_recall_qps": pts, "queries": nq,
                         "other_builders": others}
            if want_cpu:
                out[name].update(cpu_point(idx, base, qh, first[0]))
            o = out[name]
            o["better_at_0.98"] = "exact scan" if (not o["reached_0.98"] or o["exact_knn_qps"] > o["qps"]) else "hnsw"
            # The rule a caller can apply WITHOUT timing both (hnsw-clj_amd/ultra_fast.py: search_batch(route=True), hnsw.gpu/
            # search-knn-routed): the traversal evaluates E(ef) rows per query at random, the exact scan streams all n once per
            # batch through the matrix cores -- take the scan when E(ef) >= n / 3.  routed_qps = what that rule delivers here.
            st = torch.zeros((min(nq, 256), 2), dtype=torch.int64, device=dev)
            idx.hnsw_search_dev(Q[:min(nq, 256)], K, first[0], stats=st)
            torch.cuda.synchronize()
            evals = float(st[:, 0].double().mean())
            o["evals_per_query"] = round(evals, 1)
            o["routed_to"] = "exact scan" if (evals >= N31K / 3.0 or not o["reached_0.98"]) else "hnsw"
            o["routed_qps"] = o["exact_knn_qps"] if o["routed_to"] == "exact scan" else o["qps"]
        log("by_distribution %s: %s (%.1fs)" % (name, out[name], time.time() - t0))
    out["note"] = ("i.i.d. gaussian / uniform 768-d have no neighbourhood structure: recall 0.98 needs ef in the thousands, where "
                   "the traversal evaluates most of the base per query -- on those sets the GPU's exact scan (recall 1.0) is the "
                   "better answer, and better_at_0.98 says so.  On well-separated clusters closest-m pruning "
                   "(ultra_fast.clj:279-299) leaves the clusters disconnected (recall ~0.03 at every ef); the diversity heuristic "
                   "of graph.clj:162-198 keeps them linked.  DESIGN.md section 6.")
    return out


def cpu_point(idx, base, queries, ef, seconds=4.0):
    """The CPU oracle (f64, reference order) on THIS index's graph at `ef`, one task per query on the best thread count of
    a short pilot; a bounded sample of the same queries (about `seconds` of wall time)."""
    from oracle import oracle as O

    g = idx.get_graph()
    og = O.Graph(g.levels, g.l0_adj, g.up_off, g.up_adj, g.M, g.entry, g.max_level)
    ncpu = os.cpu_count() or 1
    try:
        ncpu = min(ncpu, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    t = max(1, min(32, ncpu))
    npilot = min(len(queries), 2 * t)
    _, _, _, ms = O.hnsw_search(base, og, queries[:npilot], K, ef=ef, nthreads=t)
    rate = npilot / max(ms, 1e-3) * 1e3
    nqc = int(max(npilot, min(len(queries), rate * seconds)))
    _, _, _, ms = O.hnsw_search(base, og, queries[:nqc], K, ef=ef, nthreads=t)
    return {"cpu_qps": round(nqc / (ms * 1e-3), 1), "cpu_threads": t, "cpu_sample_queries": nqc}


def ivf_roofline(engine, dev, args, traffic):
    """1M x 768 clustered-normalised, nlist=1024, nprobe=32.  The timed kernel is the
    list scan (scan_kernel over the probed lists).  Three byte counts per launch:
      algorithmic = sum over (query, probed list) pairs of len * (4*D + 4)  [rows + precomputed norms], SURVEY 8(d);
      unique      = the same with every list counted once per batch (pairs of one batch that probe the same list run
                    side by side on one XCD and the second reader finds the first one's lines in that L2);
      traffic     = what the fabric actually moved: PMC, 2 x FETCH_SIZE + WRITE_SIZE (pmc_traffic()).
    roofline.achieved / frac are TRAFFIC / kernel time against the 8 TB/s HBM spec -- a fraction of the roofline, never
    above 1; without a PMC pass (N > 1, --no-pmc) they fall back to the unique bytes, a lower bound of the traffic."""
    n, nlist, nprobe = args.ivf_n, 1024, 32
    x, Qa = ivf_dataset(dev, n, nlist, 16384)
    base_h = None if args.no_cpu else x.cpu().numpy()   # for the oracle beside every timed batch size (ivf_parity_check)
    idx = engine.Index(x, "cosine", dev.index)
    del x
    t0 = time.time()
    idx.ivf_build(nlist, 10, 42)
    build_s = time.time() - t0
    log("ivf build (k-means++ + 10 Lloyd on device) %.1fs" % build_s)
    cent_h, off, lids_h = idx.get_ivf()
    lens = np.diff(off)
    out = {}
    for nq in (1, 32, 256, 1024, 2048, 4096, 8192, 16384, "4096_nohome", "32_f32", "4096_f32"):
        f32_only = isinstance(nq, str) and nq.endswith("_f32")   # the same batch with the int8 / half-precision rows switched off
        no_home = isinstance(nq, str) and nq.endswith("_nohome")  # ... with the home-list pass of large batches switched off (A/B, same box)
        key, nq = nq, int(nq.split("_")[0]) if isinstance(nq, str) else nq
        idx.set_rejection_test(0 if f32_only else 1)
        engine.set_tuning("STREAM_HOME", 0 if no_home else None)
        Q = Qa[:nq].contiguous()
        _, _, probes = idx.ivf_search(Q.cpu().numpy(), K, nprobe, want_probes=True)
        rows = int(lens[probes.ravel()].sum())
        uniq = int(lens[np.unique(probes.ravel())].sum())
        alg_bytes = rows * (4 * DIM + 4)
        steps = 5 if nq >= 1024 else 50                # (short bursts run before the memory clocks have ramped up)
        for _ in range(3 if nq >= 1024 else 20):
            idx.ivf_search_dev(Q, K, nprobe)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            idx.ivf_search_dev(Q, K, nprobe)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / steps      # end-to-end search, profiling events off
        idx.set_profiling(True)                        # same launches again with hipEvents around the scan kernel
        idx.get_profile(engine.PROF_IVF_SCAN, reset=True)
        idx.rejection_stats(reset=True)
        for _ in range(steps):
            idx.ivf_search_dev(Q, K, nprobe)
        ms, cnt = idx.get_profile(engine.PROF_IVF_SCAN, reset=True)
        surv, cand = idx.rejection_stats(reset=True)
        idx.set_profiling(False)
        avg_ms = ms / max(cnt, 1)
        out[key] = {"nq": nq, "survivors_per_query": round(surv / max(steps * nq, 1), 1),
                   "candidates_per_query": round(rows / nq, 1), "unique_rows": uniq, "avg_scan_ms": round(avg_ms, 4), "search_wall_ms": round(wall * 1e3, 4),
                   "qps": round(nq / wall, 1), "algorithmic_GB": round(alg_bytes / 1e9, 4),
                   "unique_GB": round(uniq * (4 * DIM + 4) / 1e9, 4),
                   "algorithmic_GBs": round(alg_bytes / (avg_ms * 1e-3) / 1e9, 1),
                   "unique_GBs": round(uniq * (4 * DIM + 4) / (avg_ms * 1e-3) / 1e9, 1)}
        if base_h is not None:
            out[key]["parity_checked"] = ivf_parity_check(idx, base_h, cent_h, off, lids_h, Q, nprobe, surv / max(steps * nq, 1))
    idx.set_rejection_test(1)
    engine.set_tuning("STREAM_HOME", None)
    # single query, true latency: one call, one sync, host timer
    lat = []
    o1 = (torch.empty((1, K), dtype=torch.int32, device=dev), torch.empty((1, K), dtype=torch.float32, device=dev))
    for i in range(60):
        t1 = time.perf_counter()
        idx.ivf_search_dev(Qa[i:i + 1], K, nprobe, out=o1)
        torch.cuda.synchronize()
        lat.append((time.perf_counter() - t1) * 1e6)
    lat = sorted(lat[10:])
    # ... and through the reference's seam: host buffers in and out (hnswgpu_ivf_search: query and results in mapped pinned
    # memory, the caller spins on a flag the last launch sets)
    qh1 = Qa[:64].cpu().numpy()
    lat_host = []
    for i in range(60):
        t1 = time.perf_counter()
        idx.ivf_search(qh1[i:i + 1], K, nprobe)
        lat_host.append((time.perf_counter() - t1) * 1e6)
    lat_host = sorted(lat_host[10:])
    # recall of the IVF configuration against exact kNN (GPU brute force)
    ti, _ = idx.exact_knn_dev(Qa[:256].contiguous(), K)
    ii, _ = idx.ivf_search_dev(Qa[:256].contiguous(), K, nprobe)
    torch.cuda.synchronize()
    rec = recall_at_k(ii, ti)
    idx.close()
    # The regimes, each against its own bound:
    #  * batch 32 -> the survivor stream's bounds kernel over the int8 list rows: HBM-bound.  This is the `roofline` object
    #    (f32_scan: the same batch by one f32 GEMV per (query, list) pair, BASELINE.json configs[2] "fused GEMV").
    #  * batch 1 (configs[2] names no batch): the same kernel with nothing to share -- every list is read once.
    #  * batches 256 / 1024 (configs[3]'s batch) / 4096: the stream again, with the half-precision pass behind the bounds.
    #  * batch 4096 on a handle WITHOUT int8 rows -> pairs grouped by list, the f32-MFMA tile kernel: rows are fetched once
    #    per 32-query group, the bound is the f32 matrix rate (batched_mfma).
    r = out[32]
    f = out["32_f32"]
    b = out["4096_f32"]   # 128 (query, list) pairs per list without the int8 rows: the MFMA tile scan
    o = out[1]
    m = out[256]
    flops = 2.0 * b["algorithmic_GB"] * 1e9 / (4 * DIM + 4) * DIM      # 2 * rows scanned * D
    tf = flops / (b["avg_scan_ms"] * 1e-3) / 1e12
    code_row = 256 * ((DIM + 255) // 256) + 16                         # int8 row + (scale, bound terms, 1 / norm)
    # what the bounds kernel has to move per launch: every probed list once (all the pairs of a list are one group at
    # this batch size) in int8, and 16 bytes per survivor appended
    req_bytes = r["unique_rows"] * code_row + 16 * r["survivors_per_query"] * r["nq"]
    tr = traffic["traffic"] if traffic else None
    achieved = (tr if tr else req_bytes) / 1e9 / (r["avg_scan_ms"] * 1e-3)
    tr_f = traffic["traffic_f32_scan"] if traffic else None
    ach_f = (tr_f / 1e9 if tr_f else f["unique_GB"]) / (f["avg_scan_ms"] * 1e-3)
    # the whole search: every byte its launches moved (PMC, all kernels of one search) -- or, without a PMC pass, the bytes
    # it has to move: the probed lists once in int8, the survivors' f32 rows, the centroid table once per query group
    need_search = req_bytes + r["survivors_per_query"] * r["nq"] * (4 * DIM + 4) + nlist * (4 * DIM + 4) * max(1, r["nq"] // 2)
    tr_s = traffic["traffic_search"] if traffic else None
    ach_s = (tr_s if tr_s else need_search) / 1e9 / (r["search_wall_ms"] * 1e-3)
    res = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": tr,
           "achieved_from": "PMC traffic / kernel time" if tr else "requested bytes / kernel time (no PMC pass in this run: "
                            "--no-pmc, N > 1 or rocprofv3 unavailable; a lower bound of the traffic; profiles/ holds a PMC run)",
           "frac_of_copy_ceiling": round(achieved / 6290.0, 4),
           "frac_like_for_like": round(ach_f / HBM_PEAK_GBS, 4),
           "frac_like_for_like_note": "the kernel that does the REFERENCE's work per candidate (f32 list scan, one GEMV per (query, "
                                      "list) pair, bounds pass off: roofline.f32_scan) as traffic / time / 8 TB/s -- frac itself "
                                      "is the int8 bounds kernel's, which decides ~97 % of the candidates without their f32 row",
           "kernel": "stream_bounds_kernel<3, true, false, 1> (v_mfma_i32_32x32x32_i8; 1.6 queries per probed list at this batch: the "
                     "lane = row epilogue)",
           "workload": "hnsw.ivf-flat %d x 768, nlist=1024 nprobe=32, batch of 32 queries per launch: bounds of every candidate "
                       "from the int8 list rows against the query's running threshold, %.0f survivors per query appended "
                       "(their f32 distances, the top-k and the results: ivf_finish_kernel)" % (n, r["survivors_per_query"]),
           "avg_launch_ms": r["avg_scan_ms"], "requested_bytes_per_launch": int(req_bytes),
           "algorithmic_bytes_per_launch": int(r["algorithmic_GB"] * 1e9),
           "algorithmic_GBs": r["algorithmic_GBs"], "frac_algorithmic": round(r["algorithmic_GBs"] / HBM_PEAK_GBS, 4),
           "reuse_note": "algorithmic bytes are the reference algorithm's: every probed list in f32, once per (query, list) "
                         "pair (SURVEY 8d).  The search reads each probed list once per batch in int8 and f32 rows only for "
                         "the candidates whose lower bound does not exclude them from the k nearest, so algorithmic_GBs is "
                         "a throughput figure far above any memory rate -- frac is the kernel's own traffic over its time",
           "search": {"what": "ONE batch-32 search end to end (routing distances, routing tail with the work list built in the same launch, bounds pass, finish): "
                              "all the bytes its launches moved / its wall time",
                      "wall_ms": r["search_wall_ms"], "traffic": tr_s, "bytes_needed": int(need_search),
                      "achieved": round(ach_s, 1), "unit": "GB/s", "frac": round(ach_s / HBM_PEAK_GBS, 4),
                      "achieved_from": "PMC traffic of every kernel of the search / wall" if tr_s else "bytes needed / wall (no PMC pass)",
                      "per_kernel_bytes": traffic["per_kernel"] if traffic else None, "qps": r["qps"]},
           "f32_scan": {"kernel": "scan_kernel<3,8,false,ROLE_LIST_SCAN>", "avg_launch_ms": f["avg_scan_ms"],
                        "search_wall_ms": f["search_wall_ms"], "qps": f["qps"], "traffic": tr_f,
                        "achieved": round(ach_f, 1), "unit": "GB/s", "frac": round(ach_f / HBM_PEAK_GBS, 4),
                        "algorithmic_GBs": f["algorithmic_GBs"], "unique_bytes_GBs": f["unique_GBs"],
                        "frac_unique": round(f["unique_GBs"] / HBM_PEAK_GBS, 4),
                        "note": "the same batch with the bounds pass switched off (hnswgpu_set_rejection_test 0): one f32 GEMV "
                                "per (query, probed list) pair -- round 1 / 2's roofline kernel, same results bit for bit"},
           "batch_32": r,
           "batch_256": m,
           "batch_1024": out[1024],
           "batch_2048": out[2048],
           "batch_4096": out[4096],
           "batch_8192": out[8192],      # (from 256 (query, list) pairs per list the bounds pass takes TWO 32-query column blocks
           "batch_16384": out[16384],    #  per staged row: stream_bounds_kernel<3, false, true, 2>; each leg with its own parity_checked)
           "batch_4096_home_list_pass_off": out["4096_nohome"],
           "stream_note": "batches of 1.5 M candidates and more (48 queries here) pass the int8 survivors (survivors_per_query of batch_32: ~3 % of the "
                          "candidates) through half-precision list rows before any f32 row is fetched: survivors_per_query "
                          "of batch_256 / 1024 / 2048 / 4096 counts the f32 rows that remain.  From 512 queries (and half a query per list) "
                          "the half-precision rows of every query's NEAREST list go once through the matrix cores for all the queries "
                          "it is nearest to (ivf_home_kernel, v_mfma_f32_16x16x32_f16), the queries' thresholds come from there "
                          "(ivf_home_select_kernel) and the centroid distances from the f32 matrix cores in the GEMV order "
                          "(ivf_route_mfma_kernel); avg_scan_ms stays the int8 bounds kernel.  batch_4096_home_list_pass_off: the same "
                          "batch with that pass switched off (hnswgpu_set_tuning STREAM_HOME 0), same run",
           "batch_1": {"kernel_ms": o["avg_scan_ms"], "algorithmic_bytes": int(o["algorithmic_GB"] * 1e9),
                       "int8_bytes": int(o["unique_rows"] * code_row), "survivors": o["survivors_per_query"],
                       "achieved": round(o["unique_rows"] * code_row / 1e9 / (o["avg_scan_ms"] * 1e-3), 1), "unit": "GB/s",
                       "frac": round(o["unique_rows"] * code_row / 1e9 / (o["avg_scan_ms"] * 1e-3) / HBM_PEAK_GBS, 4),
                       "end_to_end_us": {"p50": round(lat[len(lat) // 2], 1), "min": round(lat[0], 1),
                                         "p95": round(lat[int(len(lat) * 0.95)], 1)},
                       "host_entry_us": {"p50": round(lat_host[len(lat_host) // 2], 1), "min": round(lat_host[0], 1),
                                         "p95": round(lat_host[int(len(lat_host) * 0.95)], 1),
                                         "path": "hnswgpu_ivf_search, host buffers in and out (search-knn's seam)"},
                       "end_to_end_frac_of_reference_bytes": round(o["algorithmic_GB"] * 1e9 / (lat[len(lat) // 2] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                       "note": "one query, 32 lists: the bounds pass reads them once in int8 (achieved / frac: those bytes over "
                               "the bounds kernel's time); end to end = routing, bounds, finish in one call + sync, priced at "
                               "the reference algorithm's f32 bytes"},
           "batched_mfma": {"bound": "mfma", "kernel": "tile_scan_kernel (v_mfma_f32_32x32x2_f32)",
                            "workload": "same index, batch of 4096 queries per launch WITHOUT the int8 / half-precision rows "
                                        "(hnswgpu_set_rejection_test 0; 128 pairs per list: past the boundary of 12), pairs grouped "
                                        "by list.  With them (the default) the survivor stream serves this batch too: batch_4096",
                            "achieved": round(tf, 2), "peak": 157.3, "unit": "TFLOP/s", "frac": round(tf / 157.3, 4),
                            "avg_launch_ms": b["avg_scan_ms"], "qps_end_to_end": b["qps"],
                            "algorithmic_GBs": b["algorithmic_GBs"], "unique_GB": b["unique_GB"],
                            "unique_GBs": b["unique_GBs"], "frac_unique_of_hbm": round(b["unique_GBs"] / HBM_PEAK_GBS, 4),
                            "note": "a group of <= 32 queries shares one fetch of a row: 2 * 32 * 768 flop per 3 KB = 16 flop/B, "
                                    "below the machine balance (157 TFLOP/s / 6.3 TB/s = 25 flop/B) -- at this batch the kernel is "
                                    "HBM-bound as well: every list is read at least once (unique_GBs is that lower bound of the "
                                    "traffic; the PMC pass in profiles/ has the measured bytes)"},
           "ivf_recall_at_10": round(rec, 4), "ivf_build_s": round(build_s, 1),
           "mean_list_len": float(lens.mean()), "max_list_len": int(lens.max())}
    if traffic:
        res["traffic_note"] = traffic["traffic_note"]
        res["raw_fetch_size_bytes"] = traffic["raw_fetch_size_bytes"]
    return res


_IVF_NORMS = {}


def ivf_parity_check(idx, base, cent, off, lids, Q, nprobe, f32_rows_per_query, n=16):
    """The timed IVF launch against the oracle, outside the timed region (ivf_flat.clj:217-294): the batch that was
    timed is issued once more on the same handle (same batch size, so the same kernels) and `n` of its queries, spread
    over the batch, are compared with the oracle on the same centroids and lists -- ids and distance bits against its
    device (GEMV) order (the f32 MFMA tile scan of handles without int8 rows: against its MFMA order), ids and distances
    against its f64 reference order within the north_star's tolerance (1e-4 relative)."""
    from oracle import oracle as O

    nq = Q.shape[0]
    gi, gd = idx.ivf_search_dev(Q, K, nprobe)
    torch.cuda.synchronize()
    sub = np.unique(np.linspace(0, nq - 1, min(n, nq)).astype(np.int64))
    gi, gd = gi.cpu().numpy()[sub], gd.cpu().numpy()[sub]
    qh = Q.cpu().numpy()[sub]
    res = {}
    for name, mode in (("gemv", O.MODE_DEV), ("mfma", O.MODE_MFMA)):
        if mode not in _IVF_NORMS:                      # the oracle's row norms in that summation order, once per run
            _IVF_NORMS[mode] = O.norms(base, mode)
        oi, od, _ = O.ivf_search(base, cent, off, lids, qh, K, nprobe, mode=mode, base_norms=_IVF_NORMS[mode])
        od32 = np.asarray(od, np.float64).astype(np.float32)
        res[name] = bool(np.array_equal(gi, oi)), bool(np.array_equal(gd.view(np.uint32), od32.view(np.uint32)))
        if all(res[name]):
            break
    order = "gemv" if all(res["gemv"]) else ("mfma" if all(res.get("mfma", (False,))) else "none")
    fi, fd, _ = O.ivf_search(base, cent, off, lids, qh, K, nprobe)
    close = np.abs(gd.astype(np.float64) - fd) <= 1e-4 * np.abs(fd) + 1e-6
    best = res[order] if order != "none" else res["gemv"]
    return {"queries": int(len(sub)), "launch": "%d queries, nprobe %d (the timed configuration)" % (nq, nprobe),
            "summation_order_matched": order,
            "ids_equal_oracle": best[0], "distance_bits_equal_oracle": best[1],
            "within_1e-4_of_f64_reference_order": bool(close.all()),
            "id_sets_equal_f64_reference_order": bool(all(set(a.tolist()) == set(b.tolist()) for a, b in zip(gi, fi))),
            "f32_rows_per_query_of_the_timed_launches": round(float(f32_rows_per_query), 1)}


def ivf_dataset(dev, n, nlist, nq_all):
    """S2 of BASELINE.md on the device: `nlist` gaussian centres, noise 0.3, L2-normalised rows; queries drawn the
    same way with their own seed."""
    g = torch.Generator(device=dev)
    g.manual_seed(42)
    centers = torch.randn(nlist, DIM, generator=g, device=dev)
    which = torch.randint(0, nlist, (n,), generator=g, device=dev)
    x = centers[which] + 0.3 * torch.randn(n, DIM, generator=g, device=dev)
    x /= x.norm(dim=1, keepdim=True)
    g.manual_seed(43)
    blocks = []                       # (in blocks of 4096: the first 4096 queries are the same draws whatever nq_all is)
    for b0 in range(0, nq_all, 4096):
        m = min(4096, nq_all - b0)
        qw = torch.randint(0, nlist, (m,), generator=g, device=dev)
        blocks.append(centers[qw] + 0.3 * torch.randn(m, DIM, generator=g, device=dev))
    Qa = torch.cat(blocks) if len(blocks) > 1 else blocks[0]
    Qa /= Qa.norm(dim=1, keepdim=True)
    return x, Qa


PMC_SCAN_KERNEL = "stream_bounds_kernel<3, true, false"    # the bounds kernel for dim 768 (batch 32: the lane = row epilogue)
PMC_F32_KERNEL = "scan_kernel<3, 8, false, 0>"       # ROLE_LIST_SCAN instantiation: the f32 scan (bounds pass off)
# every kernel of one batch-32 search through the survivor stream (roofline.search)
PMC_SEARCH_KERNELS = ("ivf_route_kernel", "ivf_route_dist_kernel", "ivf_route_tail_kernel", "ivf_worklist_kernel",
                      "stream_bounds_kernel", "ivf_mid_kernel", "ivf_finish_kernel")
PMC_CHILD_SEARCHES = 6


def pmc_child(args):
    """Run under `rocprofv3 --pmc <one counter>`: the same 1M x 768 index and the same batch-32 searches as
    ivf_roofline(), nothing else, so that every dispatch of the list-scan kernel in the counter file is one of
    the launches `roofline.achieved` is quoted on."""
    from hnsw_clj_amd import engine

    engine.set_tuning("IVF_CALIBRATE", 0)                 # (the handle's one-off calibration search is not one of the six)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    x, Qa = ivf_dataset(dev, args.ivf_n, 1024, 4096)       # the same query draws as ivf_roofline()
    idx = engine.Index(x, "cosine", 0)
    del x
    idx.ivf_build(1024, 10, 42)
    Q = Qa[:32].contiguous()
    for mode in (1, 0):                                # the bounds pipeline, then the f32 scan of the same batch
        idx.set_rejection_test(mode)
        for _ in range(PMC_CHILD_SEARCHES):
            idx.ivf_search_dev(Q, K, 32)
        torch.cuda.synchronize()
    idx.close()
    # ... and the timed HNSW launch (roofline_hnsw.traffic): the same index, builder, queries and operating point as main()
    base = make_31k(args.dist, 42, N31K)
    Q = torch.from_numpy(make_31k(args.dist, 43, args.nq)).to(dev)
    idx = engine.Index(base, "cosine", 0)
    idx.hnsw_build(M, EFC, 42, **BUILDERS[args.builder])
    truth, _ = idx.exact_knn_dev(Q, K)
    ef, _, _ = find_ef(idx, Q, truth, args.ef)
    out = (torch.empty((args.nq, K), dtype=torch.int32, device=dev), torch.empty((args.nq, K), dtype=torch.float32, device=dev))
    torch.cuda.synchronize()
    for _ in range(PMC_HNSW_LAUNCHES):                 # the LAST dispatches of hnsw_search_kernel in the counter file
        idx.hnsw_search_dev(Q, K, ef, out=out)
    torch.cuda.synchronize()
    idx.close()


PMC_HNSW_LAUNCHES = 3


def pmc_traffic(args):
    """roofline.traffic: HBM bytes per launch of the list-scan kernel from the PMC counters, collected as
    MI355X_MICROARCH.md (HBM section) prescribes -- FETCH_SIZE and WRITE_SIZE each in a rocprofv3 pass of its own
    (kernel trace only beside them), units KB; on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced read
    (16 B per lane, this kernel's only global read of any size), so it is doubled; WRITE_SIZE is exact.
    Returns None (traffic stays null) if rocprofv3 is missing or a pass fails."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        log("pmc: rocprofv3 not found, roofline.traffic stays null")
        return None
    got, search, hn = {}, {}, {}
    env = dict(os.environ, TMPDIR="/tmp")
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="hnswgpu_pmc_", dir="/tmp")
        cmd = [exe, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "--",
               sys.executable, os.path.abspath(__file__), "--pmc-child", "--ivf-n", str(args.ivf_n), "--dist", args.dist,
               "--builder", args.builder, "--nq", str(args.nq), "--ef", str(args.ef)]
        t0 = time.time()
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=env, timeout=300, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
            vals, vals_f, vals_h = [], [], []
            per_k = {}
            for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
                for row in csv.DictReader(open(f)):
                    if row["Counter_Name"] != ctr:
                        continue
                    if "hg::hnsw_search_kernel<" in row["Kernel_Name"] or "hg::hnsw_wave_kernel<" in row["Kernel_Name"]:  # (dispatch id, bytes): the timed launches are the last
                        vals_h.append((int(row.get("Dispatch_Id", len(vals_h))), float(row["Counter_Value"]) * 1024.0))
                    if PMC_SCAN_KERNEL in row["Kernel_Name"]:
                        vals.append(float(row["Counter_Value"]))
                    elif PMC_F32_KERNEL in row["Kernel_Name"]:
                        vals_f.append(float(row["Counter_Value"]))
                    for kn in PMC_SEARCH_KERNELS:      # every launch of the six searches through the survivor stream
                        if ("hg::" + kn + "<") in row["Kernel_Name"] or ("hg::" + kn + "(") in row["Kernel_Name"]:
                            per_k[kn] = per_k.get(kn, 0.0) + float(row["Counter_Value"]) * 1024.0 / PMC_CHILD_SEARCHES
            search[ctr] = per_k
            # a search is two dispatches (the traversal + its repeat pass for queries with hundreds of tied candidates)
            vals_h.sort()
            hn[ctr] = sum(v for _, v in vals_h[-2 * PMC_HNSW_LAUNCHES:]) / PMC_HNSW_LAUNCHES if len(vals_h) >= 2 * PMC_HNSW_LAUNCHES else None
            if r.returncode != 0 or not vals or not vals_f:
                log("pmc: %s pass failed (rc %d, %d + %d rows): %s" % (ctr, r.returncode, len(vals), len(vals_f), r.stderr.decode()[-300:]))
                return None
            got[ctr] = (sum(vals) / len(vals) * 1024.0, len(vals), sum(vals_f) / len(vals_f) * 1024.0, len(vals_f))
            log("pmc: %s pass %.0fs, %d launches of %s (%.0f B), %d of %s (%.0f B)" % (
                ctr, time.time() - t0, len(vals), PMC_SCAN_KERNEL, got[ctr][0], len(vals_f), PMC_F32_KERNEL, got[ctr][2]))
        except Exception as e:  # noqa: BLE001 -- a missing profiler must not take the bench down
            log("pmc: %s pass: %s: %s" % (ctr, type(e).__name__, e))
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    # gfx950: FETCH_SIZE tallies a 128-B request of a wide coalesced read at 64 B (MI355X_MICROARCH.md, HBM section:
    # calibrated there for 16 B per lane -- exactly what the bounds kernel's operand loads, the finish kernel's row
    # gathers and the routing kernels' centroid reads are); WRITE_SIZE is exact
    rd, wr = 2.0 * got["FETCH_SIZE"][0], got["WRITE_SIZE"][0]
    rd_f, wr_f = 2.0 * got["FETCH_SIZE"][2], got["WRITE_SIZE"][2]
    per_kernel = {kn: int(2.0 * search["FETCH_SIZE"].get(kn, 0.0) + search["WRITE_SIZE"].get(kn, 0.0))
                  for kn in PMC_SEARCH_KERNELS if kn in search["FETCH_SIZE"] or kn in search["WRITE_SIZE"]}
    hnsw_tr = None
    if hn.get("FETCH_SIZE") is not None and hn.get("WRITE_SIZE") is not None:
        hnsw_tr = {"traffic": int(2.0 * hn["FETCH_SIZE"] + hn["WRITE_SIZE"]), "raw_fetch_size_bytes": int(hn["FETCH_SIZE"]),
                   "write_size_bytes": int(hn["WRITE_SIZE"])}
    return {"hnsw": hnsw_tr, "traffic": int(rd + wr), "traffic_f32_scan": int(rd_f + wr_f),
            "traffic_search": int(sum(per_kernel.values())), "per_kernel": per_kernel,
            "raw_fetch_size_bytes": int(got["FETCH_SIZE"][0]), "raw_fetch_size_bytes_f32_scan": int(got["FETCH_SIZE"][2]),
            "traffic_note": "HBM bytes per launch at batch 32: 2 x FETCH_SIZE (gfx950 wide-read correction) + WRITE_SIZE, each "
                            "from its own rocprofv3 --pmc pass over %d launches of the same index and batch in a child "
                            "process (bounds kernel: read %d B, written %d B; f32 scan with the bounds pass off: read %d B, "
                            "written %d B)" % (got["FETCH_SIZE"][1], rd, wr, rd_f, wr_f)}


def sharded_ivf(engine, dev, rank, world, args):
    """BASELINE.json configs[3]: ONE 768-d IVF-FLAT index over all GPUs (1.25M rows per GPU = 10M x 768 on 8; nlist 1024,
    nprobe 32, k 10, batch 1024).  SURVEY 8(e)'s split: centroids replicated, whole inverted lists dealt to the ranks
    balanced by row count, every rank routes identically and scans the probed lists it holds, ONE all-gather of the
    per-rank top-k over RCCL/xGMI, merge by (distance, position in the whole index's candidate stream) on every rank.
    The build is distributed too (hnsw-clj_amd/sharded.py: seeds from rank 0, Lloyd over all rows with all-reduced list
    sums, all-to-all of the rows).  Rank 0 then rebuilds the UNSHARDED index (same centroids and lists; the rows are
    synthetic, so it regenerates every rank's) and the merged answer must equal its answer: ids and distance bits."""
    from hnsw_clj_amd.sharded import Comm, ShardedIVF, lists_from_assign

    n, nlist, nprobe, nq = args.shard_rows, 1024, 32, 1024
    g = torch.Generator(device=dev)
    g.manual_seed(7)                                   # the cluster centres are global
    centers = torch.randn(nlist, DIM, generator=g, device=dev)

    def rows_of(r):                                    # rank r's rows, reproducible on any rank
        g.manual_seed(1000 + r)
        which = torch.randint(0, nlist, (n,), generator=g, device=dev)
        x = centers[which] + 0.3 * torch.randn(n, DIM, generator=g, device=dev)
        return x / x.norm(dim=1, keepdim=True)

    g.manual_seed(43)                                  # the same query batch on every rank
    qw = torch.randint(0, nlist, (nq,), generator=g, device=dev)
    Q = centers[qw] + 0.3 * torch.randn(nq, DIM, generator=g, device=dev)
    Q /= Q.norm(dim=1, keepdim=True)
    x = rows_of(rank)
    comm = Comm(device=dev, always_collective=True)    # with one rank too: the RCCL calls of the N > 1 path are issued
    t0 = time.time()
    idx = ShardedIVF.build(x, "cosine", nlist, 10, 42, comm=comm)
    torch.cuda.synchronize()
    build_s = time.time() - t0
    del x
    for _ in range(3):
        ids, d = idx.search(Q, K, nprobe)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    steps = 10
    t0 = time.perf_counter()
    for _ in range(steps):
        ids, d = idx.search(Q, K, nprobe)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    held = comm.all_gather(torch.tensor([int(idx.shard.n)], dtype=torch.int64)).view(-1).tolist()
    assign_all = comm.all_gather(torch.from_numpy(idx.assign)).numpy().reshape(-1)     # every rank holds n rows
    res = {"workload": "hnsw.ivf-flat, ONE index of %d x 768 over %d GPU(s): centroids replicated, %d whole lists dealt by "
                       "row count, batch 1024, nprobe 32, all-gather of per-rank top-10 + keyed merge" % (world * n, world, nlist),
           "qps": round(nq * steps / el, 1), "ms_per_batch": round(el / steps * 1e3, 3),
           "rows_held_per_rank": held, "distributed_build_s": round(build_s, 1),
           "collective": "all_gather of %d B per rank" % (nq * K * 12)}
    if rank == 0:
        torch.cuda.empty_cache()
        xa = torch.cat([rows_of(r) for r in range(world)])
        with engine.Index(xa, "cosine", dev.index) as full:
            del xa
            off, lids = lists_from_assign(assign_all, nlist)
            full.set_ivf(idx.centroids, off, lids)
            ui, ud = full.ivf_search_dev(Q, K, nprobe)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(5):
                full.ivf_search_dev(Q, K, nprobe)
            torch.cuda.synchronize()
            res["unsharded_one_gpu_ms_per_batch"] = round((time.perf_counter() - t1) / 5 * 1e3, 3)
            res["ids_equal_unsharded"] = bool(torch.equal(ids, ui))
            res["distance_bits_equal_unsharded"] = bool(torch.equal(d.view(torch.int32), ud.view(torch.int32)))
            ti, _ = full.exact_knn_dev(Q[:256].contiguous(), K)
            res["recall_at_10"] = round(recall_at_k(ids[:256], ti), 4)
        torch.cuda.empty_cache()
    idx.close()
    return res


def sharded_hnsw(engine, dev, rank, world, args):
    """BASELINE.json configs[4]: 1536-d cosine HNSW, row-sharded (1.25M rows per GPU = 10M x 1536 on 8), one
    independent sub-graph per GPU (= PartitionedHNSWIndex, partitioned_hnsw.clj:23-27, searched with the full k),
    ef_search = 256, batch = 1024 queries replicated on every rank, ONE all-gather of the per-shard top-k and a merge
    kernel.  Weak scaling in rows.  Data: SURVEY S4 -- clustered-normalised rows (1024 centres per shard's worth of rows,
    noise 0.3; torch, on the device), queries held-out draws from the same mixture; the sub-graphs by the run's builder.
    Recall against exact kNN over ALL shards (same gather/merge)."""
    from hnsw_clj_amd.sharded import ShardedSearcher

    n, dim, ef, nq = args.hnsw_shard_rows, 1536, 256, 1024
    ncen = 1024 * world

    def mixture(gen, m, cen):
        out = torch.empty(m, dim, device=dev)
        for i in range(0, m, 250_000):
            c = min(250_000, m - i)
            x = cen[torch.randint(0, ncen, (c,), generator=gen, device=dev)] + 0.3 * torch.randn(c, dim, generator=gen, device=dev)
            out[i:i + c] = x / x.norm(dim=1, keepdim=True)
        return out

    g = torch.Generator(device=dev)
    g.manual_seed(11)                                  # the centres are global
    cen = torch.randn(ncen, dim, generator=g, device=dev)
    g.manual_seed(2000 + rank)                         # every shard draws its own rows
    x = mixture(g, n, cen)
    g.manual_seed(43)                                  # the same query batch on every rank
    Q = mixture(g, nq, cen)
    idx = engine.Index(x, "cosine", dev.index)
    del x
    t0 = time.time()
    idx.hnsw_build(M, EFC, 42, **BUILDERS[args.builder])
    build_s = time.time() - t0
    s = ShardedSearcher(lambda q, k: idx.hnsw_search_dev(q, k, ef), rank * n, always_collective=True)
    truth = ShardedSearcher(lambda q, k: idx.exact_knn_dev(q, k), rank * n, always_collective=True)
    ti, _ = truth.search(Q[:256].contiguous(), K)
    for _ in range(2):
        ids, d = s.search(Q, K)
    rec = recall_at_k(ids[:256], ti)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    steps = 10
    t0 = time.perf_counter()
    for _ in range(steps):
        ids, d = s.search(Q, K)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    ok = bool((ids >= 0).all() and (ids < world * n).all() and (d[:, 1:] >= d[:, :-1]).all())
    idx.close()
    return {"workload": "hnsw (M=16, ef_construction=200, %s builder) %d x 1536 cosine clustered-normalised, row-sharded over "
                        "%d GPU(s) (%d rows, one sub-graph each), ef_search 256, batch 1024, all-gather of per-shard top-10 + merge"
                        % (args.builder, world * n, world, n),
            "qps": round(nq * steps / el, 1), "ms_per_batch": round(el / steps * 1e3, 3), "recall_at_10": round(rec, 4),
            "valid": ok, "hnsw_build_s_per_shard": round(build_s, 1), "collective": "all_gather of %d B per rank" % (nq * K * 8)}


def reference_protocol(ef):
    """The protocol of the reference's published number (wip/reproduce_02ms.clj:37-92, helper/parallel_search.clj:15-49):
    T threads, each issuing single-query search-knn calls on one index -- here from plain C with pthreads
    (examples/parallel_callers.c; the synchronous C entry point combines concurrent callers into one launch).
    Returns {threads: QPS} or a note when the example cannot be built / run."""
    import re
    import shutil
    import subprocess
    import tempfile

    if shutil.which("gcc") is None:
        return {"note": "gcc not found"}
    try:
        exe = os.path.join(tempfile.mkdtemp(prefix="hnswgpu_pc_", dir="/tmp"), "parallel_callers")
        pkg = os.path.join(ROOT, "hnsw-clj_amd")
        subprocess.check_call(["gcc", "-O2", "-pthread", "-I" + os.path.join(ROOT, "include"),
                               os.path.join(ROOT, "examples", "parallel_callers.c"), "-L" + pkg, "-lhnswgpu",
                               "-Wl,-rpath," + pkg, "-lm", "-o", exe], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        out = subprocess.run([exe, str(N31K), str(DIM), str(ef)], capture_output=True, text=True, timeout=240)
        qps = {int(m.group(1)): float(m.group(2)) for m in re.finditer(r"(\d+) threads x .*? = +(\d+) QPS", out.stdout)}
        if out.returncode != 0 or not qps:
            return {"note": "parallel_callers failed: " + (out.stdout + out.stderr)[-200:]}
        return {"workload": "T threads of single-query hnswgpu_hnsw_search calls (plain C, pthreads), %d x %d, ef %d, k 10; "
                            "ids checked against one batch" % (N31K, DIM, ef),
                "qps_by_threads": qps, "published_reference": "4,719-5,376 QPS at 20 JVM threads, Apple M4 -- other hardware"}
    except Exception as e:  # noqa: BLE001 -- a side measurement never takes the bench down
        return {"note": "%s: %s" % (type(e).__name__, e)}


def parity_check(idx, base, queries, ef, out_ids, out_d, n=32):
    """The timed launches against the oracle, outside the timed region: the first `n` queries of the timed batch -- ids,
    distance bits and both traversal counters of the launch configuration that was timed (same batch size, same ef, so
    the same kernel instantiation) against the oracle's device-order mode, and ids / distances against its f64
    reference-order mode within the north_star's tolerance (1e-4 relative)."""
    from oracle import oracle as O

    g = idx.get_graph()
    og = O.Graph(g.levels, g.l0_adj, g.up_off, g.up_adj, g.M, g.entry, g.max_level)
    Q = torch.from_numpy(queries).to(out_ids.device)
    stats = torch.zeros((len(queries), 2), dtype=torch.int64, device=out_ids.device)
    idx.hnsw_search_dev(Q, K, ef, out=(out_ids, out_d), stats=stats)          # the timed launch once more, with counters
    torch.cuda.synchronize()
    gi, gd, gs = out_ids[:n].cpu().numpy(), out_d[:n].cpu().numpy(), stats[:n].cpu().numpy()
    oi, od, ost, _ = O.hnsw_search(base, og, queries[:n], K, ef=ef, mode=O.MODE_DEV)
    fi, fd, _, _ = O.hnsw_search(base, og, queries[:n], K, ef=ef)
    od32 = np.asarray(od, np.float64).astype(np.float32)
    close = np.abs(gd.astype(np.float64) - fd) <= 1e-4 * np.abs(fd) + 1e-6
    return {"queries": n, "launch": "%d queries, ef %d (the timed configuration)" % (len(queries), ef),
            "ids_equal_oracle": bool(np.array_equal(gi, oi)),
            "distance_bits_equal_oracle": bool(np.array_equal(gd.view(np.uint32), od32.view(np.uint32))),
            "counters_equal_oracle": bool(np.array_equal(gs, ost)),
            "within_1e-4_of_f64_reference_order": bool(close.all()),
            "id_sets_equal_f64_reference_order": bool(all(set(a.tolist()) == set(b.tolist()) for a, b in zip(gi, fi)))}


def cpu_baseline(idx, base, queries, ef):
    """The oracle (CPU restatement of the reference algorithm, f64, sequential sums) on the same graph,
    queries, ef and k, driven like parallel-search-futures (one task per query, T = nproc threads).
    Bounded sample sized for ~10-15 s.  The reference's JVM cannot run here (no JVM in the image)."""
    from oracle import oracle as O

    g = idx.get_graph()
    og = O.Graph(g.levels, g.l0_adj, g.up_off, g.up_adj, g.M, g.entry, g.max_level)
    # The box reports 256 logical CPUs but the job may own far fewer (cgroup quota): more threads than owned
    # cores only adds contention (measured: 508 / 3.7k / 7.2k / 7.0k / 3.2k QPS at 1 / 8 / 16 / 32 / 256 threads).
    # Pick the thread count with the best short pilot and report THAT as `cores`.
    ncpu = os.cpu_count() or 1
    try:
        ncpu = min(ncpu, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    cands = sorted({t for t in (4, 8, 16, 32, 64, 128, ncpu) if t <= ncpu})
    best = (0.0, 1)
    for t in cands:
        nqp = min(len(queries), 24 * t)
        _, _, _, ms = O.hnsw_search(base, og, queries[:nqp], K, ef=ef, nthreads=t)
        best = max(best, (nqp / max(ms, 1e-3), t))
    cores = best[1]
    nq = int(min(400000, max(256, 12000.0 * best[0])))                          # ~12 s of wall time
    qs = np.resize(queries, (nq, queries.shape[1]))                              # tile the timed queries
    _, _, _, ms, lat = O.hnsw_search(base, og, qs, K, ef=ef, nthreads=cores, want_lat=True)
    n1 = max(16, min(nq, int(3000.0 / max(ms * cores / nq, 1e-3))))              # ~3 s single thread
    _, _, _, ms1, lat1 = O.hnsw_search(base, og, qs[:n1], K, ef=ef, nthreads=1, want_lat=True)
    _, _, _, msf = O.hnsw_search(base, og, qs, K, ef=ef, mode=O.MODE_FAST, nthreads=cores)

    def pct(a):   # bench.clj:108-122: min / p50 / p95 / p99 / max / avg of the per-query times
        a = np.sort(np.asarray(a, np.float64))
        return {"min": round(float(a[0]), 4), "p50": round(float(a[len(a) // 2]), 4), "p95": round(float(a[int(len(a) * 0.95)]), 4),
                "p99": round(float(a[int(len(a) * 0.99)]), 4), "max": round(float(a[-1]), 4), "avg": round(float(a.mean()), 4)}

    return {"value": round(nq / (ms * 1e-3), 1), "unit": "queries/s", "cores": cores, "kind": "port",
            "latency_ms_per_query": pct(lat), "single_thread_latency_ms_per_query": pct(lat1),
            "reference_protocol_latency_ms": round(ms / nq, 5),     # wall / queries, reproduce_02ms.clj:80-83 (a throughput figure)
            "sample": "%d queries (the timed batch, tiled), same graph/ef/k, f64 reference-order oracle, one task per "
                      "query on %d threads (parallel_search.clj:15-49); thread count = best of a pilot over %s on a box "
                      "reporting %d logical CPUs" % (nq, cores, cands, os.cpu_count() or 1),
            "single_thread_qps": round(n1 / (ms1 * 1e-3), 1),
            "fair_fight_f32_qps": round(nq / (msf * 1e-3), 1),
            "published_reference": "5,376 QPS, 20 threads, Apple M4, JVM f64 (BENCHMARK_SUMMARY.md:16-17) -- other hardware"}


if __name__ == "__main__":
    main()
