"""GPU box: ONE walk batch at a time, nothing overlapping (batch_walk is synchronous): the walk kernel's own speed.
usage: python tools/walk_solo.py [errhmm|qshmm10|onthq] [steps] [reads]   -> avg ms per launch, columns/s, bases/s"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import harness
import pbsim3_amd as P

kind = sys.argv[1] if len(sys.argv) > 1 else "errhmm"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
G = 750_000_000
dev = torch.device("cuda", 0)
lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
gen = torch.Generator(device=dev)
gen.manual_seed(1)
genome = torch.cat([lut[torch.randint(0, 4, (min(64_000_000, G - o),), dtype=torch.uint8, device=dev, generator=gen).long()]
                    for o in range(0, G, 64_000_000)])
qs = kind == "qshmm10"
p = P.default_params(strategy=P.STRATEGY_WGS, method=P.METHOD_QS if qs else P.METHOD_ERR, seed=1, depth=20.0, pass_num=10 if qs else 1)
ctx = P.Context(p, 0)
ctx.set_scratch_bytes(48 << 30)
model = {"errhmm": "ERRHMM-ONT.model", "onthq": "ERRHMM-ONT-HQ.model", "qshmm10": "QSHMM-RSII.model"}[kind]
(ctx.load_qshmm if qs else ctx.load_errhmm)(harness.model_path(model))
ctx.set_reference_device(genome.data_ptr(), G, 1)
B = int(sys.argv[3]) if len(sys.argv) > 3 else ctx.batch_capacity()
ctx.batch_walk(1, B)
ctx.batch_finalize(0)
ctx.prof_reset()
bases = cols = 0
for i in range(steps):
    ctx.batch_walk(1 + (i + 1) * B, B)
    info = ctx.batch_finalize(0)
    bases += info.bases
    cols += info.maf_columns
walk_ms, launches, _ = ctx.prof_get()
import json
print("WALK_SOLO_JSON " + json.dumps({"kind": kind, "model": model, "reads_per_launch": B, "launches": launches, "avg_ms": walk_ms / launches,
                                      "bases_per_launch": bases / steps, "maf_columns_per_launch": cols / steps,
                                      "wave_launches": ctx.prof_wave_launches(),
                                      "env": {k: os.environ[k] for k in ("PBSIM_WALK_LDS_KB", "PBSIM_COOP_LEN", "PBSIM_COOP_WG") if k in os.environ}}))
print("%s: %d reads/launch, walk avg %.2f ms, %.1f G columns/s, %.1f G bases/s (walk kernel alone)" %
      (kind, B, walk_ms / launches, cols / (walk_ms / 1e3) / 1e9, bases / (walk_ms / 1e3) / 1e9))
ctx.close()
