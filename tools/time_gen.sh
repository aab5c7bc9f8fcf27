cd /tmp && export TMPDIR=/tmp
for C in 4p 4; do
rm -rf /tmp/pp$C
rocprofv3 --kernel-trace --stats -d /tmp/pp$C -o p -- python /root/repo/bench.py --config $C --no-cpu-baseline --no-others --repeats 1 --steps 2 > /dev/null 2>&1
python /root/repo/tools/rocpd_summary.py $(find /tmp/pp$C -name '*.db' | head -1) 2>/dev/null | grep halo | grep -v vgpr | head -5
done
cd /root/repo
python bench.py --config 4p --no-cpu-baseline --no-others 2>/dev/null | tail -1 | cut -c1-300
