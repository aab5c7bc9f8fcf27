#!/usr/bin/env bash
# Build container only: throw-away stand-ins for the third-party headers the image lacks (nlohmann/json.hpp >= 3.4, spdlog) in a SCRATCH directory,
# so that OUR reference-side glue (integration/hip_backend_glue.hpp) can be given a compiler next to the reference's real headers.  Used by
# tools/glue_syntax_check.sh (-fsyntax-only) and tools/glue_mapping_build.sh (the glue's ToHalo mapping executed on programmatic scenes).
# Nothing made here is an oracle, pins parity, ships, or is kept: the caller removes the directory.   usage: glue_scratch_stubs.sh <scratch dir>
set -uo pipefail
W=$1
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
