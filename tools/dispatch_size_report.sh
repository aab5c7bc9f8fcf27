#!/bin/bash
# profiles/rNN_dispatch_size.txt: wall time per session against session size, from C++ through the adapter (tools/dispatch_probe.cpp), for
# the default scheduling and with the round's scheduling switched off piece by piece.  Run on the GPU box:  tools/dispatch_size_report.sh > out.txt
cd "$(dirname "$0")/.."
echo "# tools/dispatch_probe.cpp: BeginSession / TraceLayer / EndSession of one regular prism (configs[1]'s scene, 1920x1080 fisheye), k sessions"
echo "# back to back, best of three; async=0: the caller reads every session's tallies (one host sync per session), async=1: what the glue does."
export PROBE_SIZES="15 16 17 18 19 20 21 22 23 24 25 26"
echo "== default options"
./tools/dispatch_probe.bin 400
export PROBE_ASYNC_ONLY=1
echo "== alt_log2=21 (launches above 2^21 rays keep one stream: the first half of round 5)"
./tools/dispatch_probe.bin 400 alt_log2=21
echo "== overlap=0 (one stream for everything)"
./tools/dispatch_probe.bin 400 overlap=0
echo "== overlap=0 table_cache=0 small_blocks_per_cu=0 (round 4's scheduling; the per-workgroup tallies of round 5 cannot be switched off)"
./tools/dispatch_probe.bin 400 overlap=0 table_cache=0 small_blocks_per_cu=0
