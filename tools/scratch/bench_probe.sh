export HALO_BENCH_BACKEND=gloo
python - <<'PY'
import re
s=open('bench.py').read()
s=s.replace('''        for _ in range(steps):
            step()                                           # dispatches are queued; nothing waits on the host per launch
        if world > 1:
            tracer._join()''','''        for _ in range(steps):
            _t=time.perf_counter(); step(); _a=time.perf_counter()-_t
            if rank==0: print("STEP host %.2f ms two=%s" % (_a*1e3, tracer.two), file=sys.stderr, flush=True)
        if world > 1:
            _t=time.perf_counter(); tracer._join()
            if rank==0: print("JOIN host %.2f ms" % ((time.perf_counter()-_t)*1e3), file=sys.stderr, flush=True)''')
open('/tmp/bench_probe.py','w').write(s.replace('ROOT = os.path.dirname(os.path.abspath(__file__))','ROOT = os.getcwd()'))
PY
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 /tmp/bench_probe.py --gpus 2 --steps 4 --warmup 1 --repeats 2 --rays-per-wl 200000 --no-cpu-baseline 2>&1 | grep "STEP\|JOIN"
