import os, sys, time
sys.path.insert(0, os.getcwd())
from ice_halo_sim_amd import scenes
def worker(rank, world, port):
    os.environ["MASTER_ADDR"]="127.0.0.1"; os.environ["MASTER_PORT"]=str(port)
    import torch, torch.distributed as dist
    from ice_halo_sim_amd import dist as D
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc, rd = scenes.config2_scene(), scenes.config2_render()
    for overlap in (True, False, True):
        tr = D.ShardedTracer(sc, rd, seed=42, device=0, rank=rank, world=world, overlap_reduce=overlap, **{"async": 1})
        T = {}
        def timed(name, fn):
            t0=time.perf_counter(); r=fn(); T[name]=T.get(name,0.0)+time.perf_counter()-t0; return r
        orig_complete = tr._complete
        tr._complete = lambda k: timed("complete", lambda: orig_complete(k))
        ob = tr.backend.bind_accumulator
        tr.backend.bind_accumulator = lambda *a: timed("bind", lambda: ob(*a))
        of = tr.backend.flush
        tr.backend.flush = lambda: timed("flush", lambda: of())
        for step in range(3):
            for w in (550.0, 610.0): tr.trace_session_layers(scenes.wl_discrete(w), 200000)
            tr.reduce_to_root()
        tr._join(); torch.cuda.synchronize(); dist.barrier(); T.clear()
        t0=time.perf_counter()
        for step in range(8):
            t1=time.perf_counter()
            for w in (550.0, 610.0): tr.trace_session_layers(scenes.wl_discrete(w), 200000)
            T["trace"]=T.get("trace",0.0)+time.perf_counter()-t1
            timed("reduce_to_root", tr.reduce_to_root)
        timed("join", tr._join); torch.cuda.synchronize(); dist.barrier()
        if rank==0: print("overlap", overlap, "%.2f ms/step" % ((time.perf_counter()-t0)*1e3/8), {k: round(v*1e3/8,2) for k,v in T.items()}, [round((r["t_done"]-r["t_call"])*1e3,1) if r["t_done"] else None for r in tr.reduce_log[-8:]], flush=True)
        tr.backend.close()
    dist.barrier(); dist.destroy_process_group()
if __name__=="__main__":
    import torch.multiprocessing as mp
    mp.spawn(worker, args=(2, 29783), nprocs=2, join=True)
