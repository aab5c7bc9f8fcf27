#!/usr/bin/env bash
# First contact of the glue with a compiler: run in a Lumice checkout on a machine with ROCm and an MI355X.
#   usage: integration/try_in_lumice.sh <lumice checkout> <engine checkout (this repository)>
# Steps: build the engine library, apply the backend patch, configure Lumice with LUMICE_HIP_ENABLED, build its CLI and its
# unit_correctness_test target, run the unit tests, then one end-to-end document on the HIP backend and on the legacy CPU path.
set -euo pipefail
LUMICE=${1:?lumice checkout}
ENGINE=${2:?engine checkout}
python3 -m ice_halo_sim_amd.build --force >/dev/null 2>&1 || (cd "$ENGINE" && python3 -m ice_halo_sim_amd.build --force)
test -f "$ENGINE/ice_halo_sim_amd/libhalo_hip.so"
cd "$LUMICE"
echo "== apply integration/lumice_hip_backend.patch by hand if 'patch' cannot place the hunks (they are anchored by comments, not by line numbers)"
patch -p1 --dry-run < "$ENGINE/integration/lumice_hip_backend.patch" || true
cmake -S . -B build/hip -DCMAKE_BUILD_TYPE=Release -DLUMICE_HIP_ENABLED=ON -DHALO_HIP_ROOT="$ENGINE" -DBUILD_TEST=ON
cmake --build build/hip -j --target Lumice unit_correctness_test
( cd build/hip && ctest -R LumiceUnitCorrectnessTest --output-on-failure )
export LD_LIBRARY_PATH="$ENGINE/ice_halo_sim_amd:${LD_LIBRARY_PATH:-}"
BIN=$(find build/hip -name Lumice -type f -perm -u+x | head -1)
for backend in hip legacy; do
  echo "== test/e2e/configs/halo_22.json on LUMICE_TRACE_BACKEND=$backend"
  LUMICE_TRACE_BACKEND=$backend "$BIN" -f test/e2e/configs/halo_22.json --benchmark
done
echo "== compare the two renders with test/e2e/_parity_metrics.py (block-mean Pearson >= 0.95, |sum-Y ratio - 1| <= 0.05: the reference's own gate)"
