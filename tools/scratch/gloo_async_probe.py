import os, sys, time
def worker(rank, world, port):
    os.environ["MASTER_ADDR"]="127.0.0.1"; os.environ["MASTER_PORT"]=str(port)
    import torch, torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    buf = torch.ones(1920*1080*3+4, device="cuda"); buf2 = torch.ones_like(buf)
    side = torch.cuda.Stream()
    def run(name, fn, n=8):
        torch.cuda.synchronize(); dist.barrier(); t0=time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); dist.barrier()
        if rank==0: print("%-40s %.2f ms each" % (name, (time.perf_counter()-t0)*1e3/n), flush=True)
    run("sync all_reduce", lambda: dist.all_reduce(buf))
    run("async + wait", lambda: dist.all_reduce(buf, async_op=True).wait())
    def c():
        with torch.cuda.stream(side):
            w = dist.all_reduce(buf, async_op=True); w.wait()
    run("async under side stream + wait", c)
    def d():
        ev = torch.cuda.Event(); ev.record(); side.wait_event(ev)
        with torch.cuda.stream(side):
            w = dist.all_reduce(buf, async_op=True)
        w.get_future().add_done_callback(lambda f: None)
        with torch.cuda.stream(side):
            w.wait()
    run("side stream + future callback + wait", d)
    pend=[None]
    def e():
        with torch.cuda.stream(side):
            w = dist.all_reduce(buf if pend[0] is None or pend[0][1] is buf2 else buf2, async_op=True)
        if pend[0] is not None:
            with torch.cuda.stream(side): pend[0][0].wait()
        pend[0]=(w, buf if pend[0] is None or pend[0][1] is buf2 else buf2)
    run("pipelined: wait previous after queuing", e)
    pend[0][0].wait()
    dist.barrier(); dist.destroy_process_group()
if __name__=="__main__":
    import torch.multiprocessing as mp
    mp.spawn(worker, args=(2, 29781), nprocs=2, join=True)
