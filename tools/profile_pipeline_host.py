#!/usr/bin/env python3
"""Host-side cost of one GroupPipeline.submit (diagnostic, GPU box): cProfile over 300 groups at 1 component per rank with the
RCCL leg forced (world size 1), i.e. the per-rank load of an 8-GPU run, where a group's GPU time is shortest."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch, torch.distributed as dist
from gbnf_amd import native, sharded, synth
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1)
C, B, S = 1, 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 16
specs = synth.synth_boosted_specs("glow", 8, 43, 215, 5, seed=1)[:C]
mix = native.NativeMixture([native.NativeFlow(s) for s in specs])
rho = torch.full((C,), 1.0 / C, device=dev)
xs = [torch.from_numpy(synth.synth_batch(B, 43, seed=k)).to(dev) for k in range(S)]
pipe = sharded.GroupPipeline(mix, C, 0, C, rho, B, S, True)
for _ in range(50): pipe.submit(xs)
torch.cuda.synchronize()
n = 300
t0 = time.perf_counter()
for _ in range(n): pipe.submit(xs)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"S={S}: host issue time per group {1e6 * (t1 - t0) / n:.1f} us; wall per group incl. drain {1e6 * (t2 - t0) / n:.1f} us")
pr = cProfile.Profile(); pr.enable()
for _ in range(n): pipe.submit(xs)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
dist.destroy_process_group()
