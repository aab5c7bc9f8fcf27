#!/bin/bash
# The N > 1 path of bench.py on ONE MI355X: ranks share the device over gloo (HALO_BENCH_BACKEND=gloo); the driver's 8-GPU run takes the same
# code over RCCL.  Prints the self-certification block (`multi_gpu`) of each run.   gpurun -- tools/gloo_rehearsal.sh > profiles/rNN_gloo_rehearsal.txt
export HALO_BENCH_BACKEND=gloo
for spec in "2 weak" "2 strong" "4 strong" "8 strong"; do
  set -- $spec
  echo "## --gpus $1 --scaling $2"
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port $((29500 + $1)) bench.py --gpus $1 --steps 2 --warmup 1 --repeats 2 --scaling $2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
lines=[l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')]
d=json.loads(lines[-1]); m=d['multi_gpu']
print(json.dumps({k:d[k] for k in ('metric','value','unit','n_gpus','steps','ms_per_step','scaling')}))
print('  ranks_seen', m['ranks_seen'], 'backend', m['backend'], 'distinct_devices', m['distinct_devices'], 'reduce_floats', m['reduce_floats'], 'reduce_ms_mean_over_ranks', m['reduce_ms_mean_over_ranks'])
for r in m['ranks']: print('  rank', r['rank'], 'local', r['local_rank'], 'dev', r['device_index'], r['name'], 'uuid', r['uuid'], 'pci', r['pci'], 'pid', r['pid'], 'reduce_ms mean %.3f max %.3f (%d timed)' % (r['reduce_ms_mean'], r['reduce_ms_max'], r['reduces_timed']), 'check sumY %.6g landed %.6g' % (r['check_sum_y'], r['check_landed']))
print('  check', json.dumps(m['check']))
print('  reduce_overlap', json.dumps(m.get('reduce_overlap')))
"
done
echo "## --gpus 1 (same box, for reference)"
python bench.py --steps 2 --warmup 1 --repeats 2 --no-cpu-baseline --no-others 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1])
print(json.dumps({k:d[k] for k in ('metric','value','unit','n_gpus','steps','ms_per_step','scaling')}), 'landed', d['config']['landed_weight_rank0_image'])"
