#!/usr/bin/env bash
# Build container only (needs /root/reference): gives integration/hip_backend_glue.hpp a COMPILER and integration/lumice_hip_backend.patch
# a `patch`.  What this is: a syntax / type check of OUR two files against the reference's real headers.  What this is not: a build of the
# reference, an oracle, or anything that ships — the third-party headers the image lacks (nlohmann/json.hpp >= 3.4, spdlog) are replaced by
# empty throw-away stubs in a scratch directory, which is enough for -fsyntax-only of a header that touches neither, and pins nothing.
#   tools/glue_syntax_check.sh [reference checkout, default /root/reference]      output: profiles/r06_glue_syntax_check.txt
set -uo pipefail
REF=${1:-/root/reference}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/profiles/r06_glue_syntax_check.txt
W=$(mktemp -d)
trap 'rm -rf "$W"' EXIT
bash "$ROOT/tools/glue_scratch_stubs.sh" "$W"
cat > "$W/tu.cpp" <<'S'
#include "core/backend/hip_backend_glue.hpp"
// instantiate through the base: an override that misses its virtual, or a pure virtual left open, fails here
int main() {
  std::unique_ptr<lumice::TraceBackend> p = std::make_unique<lumice::HipBackendGlue>();
  lumice::SessionSpec spec{};
  p->BeginSession(spec);
  auto h = p->TraceLayer(lumice::RootRaySource::FromHost(lumice::HostRayBatch{}));
  auto next = p->Recombine(std::move(h), lumice::RecombineSpec{});
  (void)next;
  std::vector<lumice::ExitRayRecord> ex;
  p->DrainExits(ex);
  p->EndSession();
  return p->SupportsDeviceXyzAccum() && p->SupportsThirdClockDrain() ? 0 : 1;
}
S
mkdir -p "$W/inc/core/backend"
cp "$ROOT/integration/hip_backend_glue.hpp" "$W/inc/core/backend/"
{
  echo "# tools/glue_syntax_check.sh — $(date -u +%Y-%m-%d) — build container, reference at $REF"
  echo "# (1) g++ -std=c++17 -fsyntax-only of integration/hip_backend_glue.hpp against the reference's headers (third-party JSON / spdlog headers: empty scratch stubs)"
  if g++ -std=c++17 -fsyntax-only -Wall -Wextra -Wno-unused-parameter -I"$W/inc" -I"$W/stub" -I"$REF/src" -I"$ROOT/ice_halo_sim_amd/csrc" -I"$ROOT/include" "$W/tu.cpp" 2>&1; then
    echo "glue: syntax check PASSED (every TraceBackend override matches a virtual, every reference field it reads exists with that type)"
  else
    echo "glue: syntax check FAILED"
  fi
  echo
  echo "# (2) patch --dry-run of integration/lumice_hip_backend.patch on a scratch copy of the reference tree"
  mkdir -p "$W/tree" && cp -r "$REF/src" "$REF/CMakeLists.txt" "$W/tree/" 2>/dev/null
  [ -d "$REF/include" ] && cp -r "$REF/include" "$W/tree/"
  (cd "$W/tree" && patch -p1 --dry-run < "$ROOT/integration/lumice_hip_backend.patch") 2>&1 && echo "patch: applies cleanly" || echo "patch: does NOT apply cleanly"
  echo
  echo "# (3) the patched src/server/server.cpp (dispatch default: kDefaultHipDispatchRayNum behind IsHipRoute) through g++ -fsyntax-only -DLUMICE_HIP_ENABLED=1"
  (cd "$W/tree" && patch -p1 -s < "$ROOT/integration/lumice_hip_backend.patch") >/dev/null 2>&1
  if g++ -std=c++17 -fsyntax-only -DLUMICE_HIP_ENABLED=1 -I"$W/inc" -I"$W/stub" -I"$W/tree/src" -I"$ROOT/ice_halo_sim_amd/csrc" -I"$ROOT/include" "$W/tree/src/server/server.cpp" 2>&1 | head -30; then :; fi
  g++ -std=c++17 -fsyntax-only -DLUMICE_HIP_ENABLED=1 -I"$W/inc" -I"$W/stub" -I"$W/tree/src" -I"$ROOT/ice_halo_sim_amd/csrc" -I"$ROOT/include" "$W/tree/src/server/server.cpp" >/dev/null 2>&1 \
    && echo "server.cpp (patched, HIP enabled): syntax check PASSED" || echo "server.cpp (patched, HIP enabled): syntax check FAILED"
  grep -n "kDefaultHipDispatchRayNum\|kIsHipRoute" "$W/tree/src/server/server.cpp"
  echo
  echo "# (4) the patched src/core/simulator.cpp (both CreateBackend sites construct HipBackendGlue) and src/server/c_api.cpp the same way"
  cp "$ROOT/integration/hip_backend_glue.hpp" "$W/tree/src/core/backend/"
  for f in core/simulator.cpp; do
    if g++ -std=c++17 -fsyntax-only -DLUMICE_HIP_ENABLED=1 -I"$W/stub" -I"$W/tree/src" -I"$ROOT/ice_halo_sim_amd/csrc" -I"$ROOT/include" "$W/tree/src/$f" >"$W/err.txt" 2>&1; then
      echo "$f (patched, HIP enabled): syntax check PASSED"
    else
      grep "error" "$W/err.txt" | head -20
      echo "$f (patched, HIP enabled): syntax check FAILED"
    fi
  done
} | sed "s#$W#<scratch>#g" | tee "$OUT"
