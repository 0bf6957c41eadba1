#!/usr/bin/env python3
"""bench.py -- simulated bases/s of the HIP hot path on MI355X.

Workload (BASELINE.json configs[1]): --strategy wgs --method errhmm --errhmm
ERRHMM-ONT.model, default length/accuracy parameters, on a synthetic uniform
ACGT genome resident in HBM.  One "step" = one batch of reads through the whole
path: header draw -> (class,length) bucketing -> ERRHMM walk -> quota scan ->
FASTQ + MAF text emitted into HBM buffers.  Inputs are resident before the timed
region; outputs stay in HBM (PCIe-inclusive rate: DESIGN.md).

N > 1 (torchrun, one rank per GPU): rank 0's genome is broadcast over RCCL,
reads shard by contiguous read-index block per rank, the pass-0 base counts are
all-gathered to place each rank's quota prefix, and the statistics counters are
all-reduced.  Weak scaling: every rank walks a full batch per step.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

RECORD_LEN = 750_000_000     # one of the 4 records of the 3 Gbp genome (<= REF_SEQ_LEN_MAX, pbsim.cpp:24)
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def cpu_baseline(sample_bp=4_000_000):
    """Reference pbsim (oracle/_ref/pbsim_ref, compiled from /root/reference in the
    build container) or, if absent, the C oracle in glibc mode, timed single-threaded
    on a bounded sample of the same workload with compression bypassed."""
    import numpy as np
    import harness
    ref = harness.REF_GLIBC if os.path.exists(harness.REF_GLIBC) else None
    if ref is None:
        harness.build_oracle()
    with tempfile.TemporaryDirectory() as td:
        rng = np.random.default_rng(1)
        seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, sample_bp)]
        fa = os.path.join(td, "g.fa")
        with open(fa, "wb") as f:
            f.write(b">chr1\n")
            lines = seq.reshape(-1, 80)
            out = np.concatenate([lines, np.full((lines.shape[0], 1), 10, np.uint8)], axis=1)
            f.write(out.tobytes())
        model = harness.model_path("ERRHMM-ONT.model")
        args = ["--strategy", "wgs", "--method", "errhmm", "--errhmm", model, "--genome", fa,
                "--depth", "20", "--seed", "1", "--prefix", os.path.join(td, "out")]
        env = dict(os.environ)
        if ref:
            stubs = os.path.join(td, "stubs")
            harness.make_stubs(stubs)
            env["PATH"] = stubs + ":" + env["PATH"]
            cmd = [ref] + args
        else:
            cmd = [harness.ORACLE] + args + ["--rng", "glibc"]
        t0 = time.time()
        p = subprocess.run(cmd, env=env, capture_output=True, text=True)
        dt = time.time() - t0
        if p.returncode != 0:
            return None

        def count(prefix):
            n = 0
            for fn in os.listdir(td):
                if fn.startswith(prefix) and (fn.endswith(".fq") or fn.endswith(".fq.gz")):
                    with open(os.path.join(td, fn), "rb") as f:
                        for i, line in enumerate(f):
                            if i % 4 == 1:
                                n += len(line) - 1
            return n

        bases = count("out")
        # the same job once per host core, concurrently (distinct seeds and prefixes): what the box's CPU can do at best
        ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        ncopy = min(ncpu, 64)                # bounded: every copy writes ~85 MB of text into the temp dir
        procs = []
        t1 = time.time()
        for k in range(ncopy):
            a2 = list(cmd)
            a2[a2.index("--seed") + 1] = str(100 + k)
            a2[a2.index("--depth") + 1] = "5"
            a2[a2.index("--prefix") + 1] = os.path.join(td, "all%d" % k)
            procs.append(subprocess.Popen(a2, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL))
        rcs = [q.wait() for q in procs]
        dt_all = time.time() - t1
        if all(rc == 0 for rc in rcs):
            all_cores = {"value": count("all") / dt_all, "cores": ncopy,
                         "note": f"{ncopy} concurrent copies of the sample job at depth 5 on the box's {ncpu} host cores"}
        else:
            all_cores = {"value": None, "cores": ncopy, "note": f"{sum(rc != 0 for rc in rcs)} of {ncopy} copies failed"}
    return {"value": bases / dt, "unit": "bases/s", "cores": 1, "kind": "reference" if ref else "port",
            "sample": f"{sample_bp // 1_000_000} Mbp uniform genome x depth 20, ERRHMM-ONT, seed 1, "
                      f"{bases} bases in {dt:.1f}s, gzip bypassed (cat)",
            "all_cores": all_cores}


def bench_sample(a, torch, harness, P, local):
    """Sampling method at scale: 200 000 synthetic quality strings (lengths gamma mean 9 000 / sd 7 000
    clipped to 100..60 000, per-read quality level Q8..Q30 with jitter), a 100 Mbp record, depth 20 -> 2.0 Gbases.
    Whole job, text left in HBM; the chains make this path latency-bound (one lane per string)."""
    import numpy as np
    rng = np.random.default_rng(1)
    n = 200_000
    k = (9000.0 / 7000.0) ** 2
    lens = np.clip(rng.gamma(k, 9000.0 / k, n), 100, 60000).astype(np.int64)
    level = rng.integers(8, 31, n)
    quals = []
    for i in range(n):
        q = np.clip(level[i] + rng.integers(-5, 6, int(lens[i])), 0, 93).astype(np.uint8) + 33
        quals.append(q.tobytes())
    genome = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 100_000_000)].tobytes()
    p = P.default_params(strategy=P.STRATEGY_WGS, method=P.METHOD_SAMPLE, seed=1, depth=float(dict(kv.split("=") for kv in a.param).get("depth", 20)))
    ctx = P.Context(p, local)
    ctx.set_scratch_bytes(int(a.scratch_gib * (1 << 30)))
    ctx.set_sample_profile(quals)
    ctx.set_reference(genome, 1)
    ctx.simulate_sample(collect=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.simulate_sample(collect=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = ctx.stats()
    print(json.dumps({"metric": "whole job wall time", "value": dt, "unit": "s", "higher_is_better": False, "n_gpus": 1,
                      "bases": st.res_len_total, "reads": st.res_num, "bases_per_sec": st.res_len_total / dt,
                      "config": {"workload": f"wgs sample, {n} synthetic quality strings ({int(lens.sum())} bases), "
                                             f"100 Mbp record, depth {p.depth}, seed 1"}}))
    ctx.close()


def bench_trans(a, torch, harness, P, local):
    """BASELINE configs[3]: --strategy trans --method errhmm --errhmm ERRHMM-SEQUEL.model on 100 000 transcripts
    (lengths log-uniform 300..12000, plus ~ Geometric(mean 20), minus ~ Geometric(mean 0.1), seed 1).
    The job has a fixed read count, so the figure is the wall time of pbsim_simulate_trans with text left in HBM."""
    import numpy as np
    rng = np.random.default_rng(1)
    n = 100_000
    lens = np.exp(rng.uniform(np.log(300), np.log(12000), n)).astype(np.int64)
    plus = rng.geometric(1 / 21.0, n) - 1
    minus = rng.geometric(1 / 1.1, n) - 1
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    allseq = acgt[rng.integers(0, 4, int(lens.sum()))].tobytes()
    offs = np.concatenate([[0], np.cumsum(lens)])
    seqs = [allseq[offs[i]:offs[i + 1]] for i in range(n)]
    p = P.default_params(strategy=P.STRATEGY_TRANS, method=P.METHOD_ERR, seed=1)
    ctx = P.Context(p, local)
    ctx.set_scratch_bytes(int(a.scratch_gib * (1 << 30)))
    ctx.load_errhmm(harness.model_path("ERRHMM-SEQUEL.model"))
    t0 = time.perf_counter()
    ctx.set_transcripts(["T%d" % i for i in range(n)], [int(x) for x in plus], [int(x) for x in minus], seqs)
    t_set = time.perf_counter() - t0
    ctx.simulate_trans(collect=False)   # warm-up (pools)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.simulate_trans(collect=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = ctx.stats()
    print(json.dumps({"metric": "whole job wall time", "value": dt, "unit": "s", "higher_is_better": False, "n_gpus": 1,
                      "bases": st.res_len_total, "reads": st.res_num, "bases_per_sec": st.res_len_total / dt,
                      "set_transcripts_s": t_set,
                      "config": {"workload": "trans errhmm ERRHMM-SEQUEL, 100000 synthetic transcripts "
                                             f"({int(lens.sum())} bp), expression plus~Geom(20) minus~Geom(0.1), seed 1"}}))
    ctx.close()


def issue_bound(columns_per_launch, walk_s):
    """The walk's real ceiling is integer issue, not HBM: figures of the PMC pass of the same kernel (tools/pmc_kernel.sh)."""
    path = os.path.join(ROOT, "profiles", "r01x_walk_pmc.json")
    if not os.path.exists(path):
        return None
    pj = json.load(open(path))
    valu = pj["per_wave_step"]["valu"]          # VALU instructions per wave-step = per lane and MAF column
    return {"valu_busy_frac": pj["valu_busy_fraction"], "valu_per_wave_step": valu,
            "int_lane_ops_per_sec": valu * columns_per_launch / walk_s if walk_s > 0 else None,
            "lds_table_hit_rate": 1.0,          # every HMM table read is an LDS read (4 per column)
            "source": "profiles/r01x_walk_pmc.json (SQ_ACTIVE_INST_VALU x 4 cycles / SIMD-cycles)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--slots", type=int, default=2, help="batches kept in flight per GPU (engine slots)")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--record-len", type=int, default=RECORD_LEN)
    ap.add_argument("--scratch-gib", type=float, default=48.0)
    ap.add_argument("--batch-reads", type=int, default=0, help="reads per step per GPU (0 = what the scratch pool holds)")
    ap.add_argument("--model", default="ERRHMM-ONT.model")
    ap.add_argument("--workload", default="errhmm", choices=["errhmm", "qshmm10", "trans", "sample"],
                    help="errhmm = BASELINE configs[1] (headline); qshmm10 = configs[2]: QSHMM-RSII, --pass-num 10; "
                         "trans = configs[3]: ERRHMM-SEQUEL on a synthetic 100k-transcript profile (whole job, wall time)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--whole-job", action="store_true",
                    help="not the per-step metric: run the complete configs[1] job (4 records x --record-len, depth 20, "
                         "quota loop + tail + statistics, text left in HBM) through pbsim_simulate_wgs and report its wall time")
    ap.add_argument("--deflate", action="store_true",
                    help="with --whole-job: compress FASTQ + MAF on the GPU (pbsim_set_deflate) and copy the members to pinned "
                         "host memory -- everything the CLI does for .fq.gz/.maf.gz except the file writes")
    ap.add_argument("--param", action="append", default=[],
                    help="experiment only: override a pbsim_params field, e.g. --param len_sd=0 (not the headline workload)")
    a = ap.parse_args()

    import torch
    import harness
    import pbsim3_amd as P

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # PBSIM_BENCH_BACKEND=gloo + PBSIM_BENCH_ONE_GPU=1: plumbing check of the N>1 path on a 1-GPU box
        backend = os.environ.get("PBSIM_BENCH_BACKEND", "nccl")
        if os.environ.get("PBSIM_BENCH_ONE_GPU") == "1":
            local = 0
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cdev = dev if (world == 1 or dist.get_backend() == "nccl") else torch.device("cpu")  # where collectives run

    # ---- inputs resident in HBM: genome record (rank 0 generates, RCCL broadcasts) ----
    G = a.record_len
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    if rank == 0:
        gen = torch.Generator(device=dev)
        gen.manual_seed(1)
        genome = lut[torch.randint(0, 4, (G,), dtype=torch.uint8, device=dev, generator=gen).long()] \
            if G <= 64_000_000 else torch.cat([
                lut[torch.randint(0, 4, (min(64_000_000, G - o),), dtype=torch.uint8, device=dev, generator=gen).long()]
                for o in range(0, G, 64_000_000)])
    else:
        genome = torch.empty(G, dtype=torch.uint8, device=dev)
    if world > 1:
        if cdev == dev:
            dist.broadcast(genome, src=0)      # C1: reference broadcast over xGMI (RCCL)
        else:
            g = genome.cpu()
            dist.broadcast(g, src=0)
            genome = g.to(dev)
    torch.cuda.synchronize()

    if a.workload == "sample":
        return bench_sample(a, torch, harness, P, local)
    if a.workload == "trans":
        return bench_trans(a, torch, harness, P, local)
    qs = a.workload == "qshmm10"
    p = P.default_params(strategy=P.STRATEGY_WGS, method=P.METHOD_QS if qs else P.METHOD_ERR, seed=1, depth=20.0,
                         pass_num=10 if qs else 1)
    for kv in a.param:
        k, v = kv.split("=")
        setattr(p, k, type(getattr(p, k))(float(v)))
    ctx = P.Context(p, local)
    ctx.set_scratch_bytes(int(a.scratch_gib * (1 << 30)))
    if qs:
        ctx.load_qshmm(harness.model_path("QSHMM-RSII.model"))
    else:
        ctx.load_errhmm(harness.model_path(a.model))
    ctx.set_reference_device(genome.data_ptr(), G, 1)
    del genome
    torch.cuda.empty_cache()
    B = a.batch_reads or ctx.batch_capacity()
    quota = ctx.unit_quota()

    if a.whole_job:
        assert world == 1, "--whole-job is a single-GPU measurement"
        recs = []
        gen = torch.Generator(device=dev)
        for r in range(4):
            gen.manual_seed(100 + r)
            recs.append(torch.cat([lut[torch.randint(0, 4, (min(64_000_000, G - o),), dtype=torch.uint8, device=dev,
                                                    generator=gen).long()] for o in range(0, G, 64_000_000)]))
        gz_bytes = [0]
        if a.deflate:                            # sinks that only count: the members are already in pinned host memory
            import ctypes as C
            ctx.set_deflate(int(os.environ.get("PBSIM_BENCH_DEFLATE_MASK", "7")))   # bit 2: the two sinks from two host threads

            def count(user, text, n):
                gz_bytes[0] += n
                return 1
            cb = P.SINK_CB(count)
            sink = P.Sink(None, cb, cb)

            def run_record():
                P._check(ctx.lib.pbsim_simulate_wgs(ctx.h, C.byref(sink)))
        else:
            def run_record():
                ctx.simulate_wgs(collect=False)
        ctx.set_reference_device(recs[0].data_ptr(), G, 1)
        ctx.prefetch_reference_device(recs[1].data_ptr(), G)   # warm-up only: allocates the second reference buffers (dropped below)
        run_record()                             # warm-up: pools allocated
        torch.cuda.synchronize()
        gz_bytes[0] = 0
        t0 = time.perf_counter()
        tot_b = tot_r = 0
        for r in range(4):
            ctx.set_reference_device(recs[r].data_ptr(), G, r + 1)
            if r + 1 < 4:                        # the next record is uploaded and prepared beside this one's simulation
                ctx.prefetch_reference_device(recs[r + 1].data_ptr(), G)
            run_record()
            st = ctx.stats()
            tot_b += st.res_len_total
            tot_r += st.res_num
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps({"metric": "whole job wall time", "value": dt, "unit": "s", "higher_is_better": False,
                          "bases": tot_b, "reads": tot_r, "bases_per_sec": tot_b / dt, "n_gpus": 1,
                          "compressed_bytes": gz_bytes[0] if a.deflate else None,
                          "config": {"workload": f"wgs errhmm ERRHMM-ONT depth 20, 4 records x {G} bp, seed 1, "
                                                 "quota loop + serial tail + statistics; " +
                                                 ("FASTQ + MAF compressed on the GPU (BGZF members) and copied to pinned host memory"
                                                  if a.deflate else "text emitted into HBM, not copied out")}}))
        ctx.close()
        return

    S = max(1, min(a.slots, P.load().pbsim_slot_count()))

    def begin(i):
        ctx.select_slot(i % S)
        ctx.batch_walk_begin(1 + (i * world + rank) * B, B)

    def finish(i):
        ctx.select_slot(i % S)
        pass0 = ctx.batch_walk_end()
        before = 0
        if world > 1:
            mine = torch.tensor([pass0], dtype=torch.int64, device=cdev)
            allv = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allv, mine)        # C3: quota prefix across ranks
            before = int(sum(int(v.item()) for v in allv[:rank]))
        # weak-scaling bench: every rank walks a full batch per step, so the prefix is folded to
        # keep each batch below the record's quota (the quota cut itself is covered by the tests)
        return ctx.batch_finalize(before % max(1, quota // 4))

    def run(first_step, n_steps):
        """n_steps batches through an S-deep pipeline; returns their BatchInfos."""
        infos, pending, nxt = [], [], first_step
        while len(infos) < n_steps:
            while len(pending) < S and nxt < first_step + n_steps:
                begin(nxt)
                pending.append(nxt)
                nxt += 1
            infos.append(finish(pending.pop(0)))
        return infos

    run(0, S)                 # untimed setup: every slot allocates its pools once
    run(S, a.warmup)          # the W untimed warmup steps
    ctx.prof_reset()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bases = reads = ref_b = maf_c = text_b = 0
    for info in run(S + a.warmup, a.steps):
        bases += info.bases
        reads += info.n_final
        ref_b += info.ref_bases
        maf_c += info.maf_columns
        text_b += info.read_text_bytes + info.maf_text_bytes
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    walk_ms, launches, total_ms = ctx.prof_get()

    tot = torch.tensor([bases, reads, ref_b, maf_c, text_b], dtype=torch.int64, device=cdev)
    tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(tot)                   # C2: counters reduced across ranks
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    g_bases, g_reads, g_ref, g_maf, g_text = (int(x) for x in tot.tolist())
    dt_max = float(tmax.item())

    if rank == 0:
        # algorithmic bytes: ref_bases*1 + read_bases*2 + maf_columns*2
        alg_bytes_launch = (ref_b * 1 + bases * 2 + maf_c * 2) / max(1, launches)
        walk_s = walk_ms / 1e3 / max(1, launches)
        achieved = alg_bytes_launch / walk_s / 1e9 if walk_s > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01x_walk_traffic.json")
        if not qs and os.path.exists(tpath):     # PMC pass of the same kernel (tools/pmc_traffic.sh), scaled per base
            tj = json.load(open(tpath))
            traffic = tj["traffic_bytes_per_launch"] / tj["bases_per_launch"] * (bases / max(1, launches))
        out = {
            "metric": "simulated bases/sec", "value": g_bases / dt_max, "unit": "bases/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt_max * 1e3 / a.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "reads_per_sec": g_reads / dt_max,
            "config": {"workload": ("wgs qshmm QSHMM-RSII pass-num 10" if qs else "wgs errhmm ERRHMM-ONT") +
                                   " depth 20, default length/accuracy, uniform ACGT record "
                                   f"of {G} bp resident in HBM (one of the 4 records of the 3 Gbp genome)",
                       "param_overrides": a.param, "reads_per_step_per_gpu": B, "bases_per_step": g_bases // a.steps,
                       "text_bytes_per_step": g_text // a.steps, "parallelism": f"read-block x{world}", "slots_in_flight": S},
            "roofline": {"bound": "hbm", "kernel": "k_walk_qshmm" if qs else "k_walk_errhmm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": "profiles/r01x_walk_traffic.json (rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE, raw x1024; see note there)",
                         "alg_bytes_per_launch": alg_bytes_launch, "avg_launch_ms": walk_s * 1e3,
                         "walk_share_of_step": (walk_ms / 1e3) / dt if dt > 0 else None,
                         # the kernel's real ceiling is integer issue, not HBM: PMC pass of the same kernel
                         "issue_bound": None if qs else issue_bound(maf_c / max(1, launches), walk_s)},
        }
        if world == 1 and not a.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline()
            except Exception as e:  # the baseline is reported, never required
                out["cpu_baseline"] = {"error": str(e)}
        print(json.dumps(out))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
