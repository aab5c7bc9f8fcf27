#!/usr/bin/env bash
# Build container only (needs /root/reference): gives integration/hip_backend_glue.hpp a COMPILER and integration/lumice_hip_backend.patch
# a `patch`.  What this is: a syntax / type check of OUR two files against the reference's real headers.  What this is not: a build of the
# reference, an oracle, or anything that ships — the third-party headers the image lacks (nlohmann/json.hpp >= 3.4, spdlog) are replaced by
# empty throw-away stubs in a scratch directory, which is enough for -fsyntax-only of a header that touches neither, and pins nothing.
#   tools/glue_syntax_check.sh [reference checkout, default /root/reference]      output: profiles/r05_glue_syntax_check.txt
set -uo pipefail
REF=${1:-/root/reference}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/profiles/r05_glue_syntax_check.txt
W=$(mktemp -d)
trap 'rm -rf "$W"' EXIT
mkdir -p "$W/stub/nlohmann" "$W/stub/spdlog/sinks"
cat > "$W/stub/nlohmann/json.hpp" <<'S'
#pragma once
// throw-away stand-in for -fsyntax-only: the glue never touches JSON; the reference's config headers only declare to_json / from_json
#include <cassert>
#include <map>
#include <string>
#include <vector>
namespace nlohmann {
struct json {
  template <class T> json& operator=(const T&) { return *this; }
  json& operator[](const char*) { return *this; }
  json& operator[](const std::string&) { return *this; }
  const json& at(const char*) const { return *this; }
  const json& at(const std::string&) const { return *this; }
  template <class T> void get_to(T&) const {}
  template <class T> T get() const { return T{}; }
  template <class T> T value(const char*, const T& d) const { return d; }
  bool contains(const char*) const { return false; }
  bool is_number() const { return false; }
  bool is_object() const { return false; }
  bool is_array() const { return false; }
  bool is_string() const { return false; }
  bool is_null() const { return true; }
  size_t size() const { return 0; }
  template <class T> void emplace_back(const T&) {}
  template <class T> void push_back(const T&) {}
  static json array() { return json{}; }
  static json object() { return json{}; }
  const json* begin() const { return nullptr; }
  const json* end() const { return nullptr; }
  struct exception : std::exception { int id = 0; };
  struct out_of_range : exception {};
  struct parse_error : exception {};
  struct type_error : exception {};
  static json parse(const std::string&) { return json{}; }
  std::string dump(int = -1) const { return {}; }
};
}  // namespace nlohmann
#define NLOHMANN_JSON_SERIALIZE_ENUM(ENUM_TYPE, ...)                  \
  inline void to_json(nlohmann::json&, const ENUM_TYPE&) {}           \
  inline void from_json(const nlohmann::json&, ENUM_TYPE&) {}
S
echo '#pragma once
#include "json.hpp"' > "$W/stub/nlohmann/json_fwd.hpp"
cat > "$W/stub/spdlog/spdlog.h" <<'S'
#pragma once
// throw-away stand-in for -fsyntax-only: the shapes util/logger.hpp and util/spdlog_levels.hpp name, nothing behind them
#include <ctime>
#include <memory>
#include <string>
namespace spdlog {
namespace level { enum level_enum { trace, debug, info, warn, err, critical, off }; }
struct formatter { virtual ~formatter() = default; };
namespace details { struct log_msg { level::level_enum level; }; }
struct memory_buf_t { void push_back(char) {} };
struct custom_flag_formatter {
  virtual ~custom_flag_formatter() = default;
  virtual void format(const details::log_msg&, const std::tm&, memory_buf_t&) = 0;
  virtual std::unique_ptr<custom_flag_formatter> clone() const = 0;
};
struct pattern_formatter : formatter {
  template <class T> pattern_formatter& add_flag(char) { return *this; }
  void set_pattern(const std::string&) {}
};
namespace sinks {
struct sink { virtual ~sink() = default; void set_formatter(std::unique_ptr<formatter>) {} };
struct dist_sink_mt : sink { void add_sink(std::shared_ptr<sink>) {} };
struct stdout_color_sink_mt : sink {};
}  // namespace sinks
struct logger {
  logger(const std::string&, std::shared_ptr<sinks::sink>) {}
  void set_formatter(std::unique_ptr<formatter>) {}
  void set_level(level::level_enum) {}
  template <class... A> void log(A&&...) {}
};
}  // namespace spdlog
#define SPDLOG_LOGGER_TRACE(l, ...) ((void)(l))
#define SPDLOG_LOGGER_DEBUG(l, ...) ((void)(l))
#define SPDLOG_LOGGER_INFO(l, ...) ((void)(l))
#define SPDLOG_LOGGER_WARN(l, ...) ((void)(l))
#define SPDLOG_LOGGER_ERROR(l, ...) ((void)(l))
#define SPDLOG_LOGGER_CRITICAL(l, ...) ((void)(l))
S
for h in pattern_formatter.h sinks/dist_sink.h sinks/stdout_color_sinks.h; do echo '#pragma once
#include "spdlog/spdlog.h"' > "$W/stub/spdlog/$h"; done
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
