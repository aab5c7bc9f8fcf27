#!/usr/bin/env bash
# Build container only (needs /root/reference): compiles tests/cpp/glue_mapping_main.cpp — integration/hip_backend_glue.hpp's ToHalo mapping
# driven on programmatic scenes — against the reference's real headers plus the scratch stubs of tools/glue_scratch_stubs.sh, links it with the
# reference's colour-table builders and math.cpp compiled where they lie (the glue calls BuildColorGateTable / BuildColorClassTable), runs it
# and prints its output (one JSON line per scene).  Everything is built in a temporary directory and removed; nothing here is an oracle or ships.
#   tools/glue_mapping_build.sh [reference checkout, default /root/reference]
set -uo pipefail
REF=${1:-/root/reference}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=$(mktemp -d)
trap 'rm -rf "$W"' EXIT
bash "$ROOT/tools/glue_scratch_stubs.sh" "$W" || exit 2
mkdir -p "$W/inc/core/backend"
cp "$ROOT/integration/hip_backend_glue.hpp" "$W/inc/core/backend/"
INC="-I$W/inc -I$W/stub -I$REF/src -I$ROOT/ice_halo_sim_amd/csrc -I$ROOT/include"
for f in config/color_gate_table.cpp config/color_class_table.cpp config/component_table.cpp core/math.cpp; do
  g++ -std=c++17 -O1 -c $INC "$REF/src/$f" -o "$W/$(basename $f).o" 2> "$W/err.txt" || { head -20 "$W/err.txt" >&2; exit 3; }
done
g++ -std=c++17 -O1 -c $INC "$ROOT/tests/cpp/glue_mapping_main.cpp" -o "$W/main.o" 2> "$W/err.txt" || { head -30 "$W/err.txt" >&2; exit 4; }
# (the glue's class methods call the C ABI; the mapping functions do not — unresolved halo_* symbols of never-called inline methods are left to
#  the dynamic linker, which never needs them)
g++ -o "$W/glue_mapping" "$W/main.o" "$W"/*.cpp.o -Wl,--unresolved-symbols=ignore-all 2> "$W/err.txt" || { head -30 "$W/err.txt" >&2; exit 5; }
"$W/glue_mapping"
