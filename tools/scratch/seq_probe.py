import os, sys, types
sys.path.insert(0, os.getcwd())
import bench, torch, torch.distributed as dist
torch.cuda.set_device(0)
ctx = {"torch": torch, "dist": dist, "world": 1, "rank": 0, "local_rank": 0, "dist_backend": "nccl"}
args = types.SimpleNamespace(rays_per_wl=0, scaling="weak", blocks_per_cu=0, aggregate=-1, opt=[], warm_seconds=0.3)
for seq in sys.argv[1:]:
    for cfg in seq.split(","):
        r = bench.measure(cfg, args, ctx, 3, 3, 3, with_cpu=False)
        print(seq, "->", cfg, "%.3f ms/step cov %.3f" % (r["ms_per_step"], r["repeats"]["cov"]), flush=True)
