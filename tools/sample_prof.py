"""Where the sampling walk's time goes: the same 2.0-Gbase job (100 Mbp record, depth 20) over profiles of different length
spread.  usage: python tools/sample_prof.py [clip ...]   (clip = longest string; 0 = every string 9 000 long)"""
import os, sys, time, json
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, "tests")]
import numpy as np
import torch
import pbsim3_amd as P


def run(clip, n=200_000):
    rng = np.random.default_rng(1)
    k = (9000.0 / 7000.0) ** 2
    lens = np.clip(rng.gamma(k, 9000.0 / k, n), 100, clip).astype(np.int64) if clip else np.full(n, 9000, dtype=np.int64)
    level = rng.integers(8, 31, n)
    big = rng.integers(-5, 6, int(lens.sum()), dtype=np.int8)
    quals, o = [], 0
    for i in range(n):
        q = np.clip(level[i] + big[o:o + lens[i]], 0, 93).astype(np.uint8) + 33
        quals.append(q.tobytes()); o += int(lens[i])
    genome = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 100_000_000)].tobytes()
    p = P.default_params(strategy=P.STRATEGY_WGS, method=P.METHOD_SAMPLE, seed=1, depth=20.0)
    ctx = P.Context(p, 0)
    ctx.set_scratch_bytes(24 << 30)
    ctx.set_sample_profile(quals)
    ctx.set_reference(genome, 1)
    ctx.simulate_sample(collect=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.simulate_sample(collect=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = ctx.stats()
    print(json.dumps({"clip": clip, "wall_ms": round(dt * 1e3, 1), "bases": st.res_len_total, "reads": st.res_num,
                      "Gbases_per_s": round(st.res_len_total / dt / 1e9, 1), "mean_len": float(lens.mean()), "max_len": int(lens.max())}), flush=True)
    ctx.close()


if __name__ == "__main__":
    torch.cuda.init()
    for c in [int(x) for x in sys.argv[1:]] or [60000, 20000, 0]:
        run(c)
