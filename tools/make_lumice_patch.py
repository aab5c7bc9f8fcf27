"""Build container only: writes integration/lumice_hip_backend.patch as a REAL unified diff against the reference tree — the Lumice-side
change that adds this repo's engine as a fourth trace backend (BackendKind::kHip).  The insertions below are anchored on short strings of
the reference's files; the script edits scratch copies and runs `diff -U1`, so the patch carries one line of context per side and applies
with `patch -p1` in a Lumice checkout (tools/glue_syntax_check.sh dry-runs it).

    python tools/make_lumice_patch.py [/root/reference]
"""
import os
import shutil
import subprocess
import sys
import tempfile

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def after(text, anchor, insert, nth=1):
    """insert `insert` after the line that holds the nth occurrence of `anchor`"""
    pos = -1
    for _ in range(nth):
        pos = text.index(anchor, pos + 1)
    eol = text.index("\n", pos + len(anchor) - 1) + 1   # the end of the line the anchor ENDS on
    return text[:eol] + insert + text[eol:]


def before(text, anchor, insert, nth=1):
    pos = -1
    for _ in range(nth):
        pos = text.index(anchor, pos + 1)
    bol = text.rfind("\n", 0, pos) + 1
    return text[:bol] + insert + text[bol:]


EDITS = {}

EDITS["src/core/backend/backend_kind.hpp"] = lambda t: after(t, "kCuda = 2,", "  kHip = 3,  // MI355X (gfx950) engine: libhalo_hip.so behind HipBackendGlue\n")


def simulator(t):
    t = after(t, '#include "util/queue.hpp"',
              '#if defined(LUMICE_HIP_ENABLED)\n#include "core/backend/hip_backend_glue.hpp"  // integration/hip_backend_glue.hpp of the engine repository, copied as it is\n#endif\n')
    # the LUMICE_TRACE_BACKEND override chain: a branch in front of the final `else`
    t = before(t, '    } else {\n      ILOG_WARN(logger, "Unknown LUMICE_TRACE_BACKEND={}', '''    } else if (name == "hip") {
#if defined(LUMICE_HIP_ENABLED)
      if (halo_device_count() > 0) {
        ILOG_INFO(logger, "LUMICE_TRACE_BACKEND=hip → routing via HipBackendGlue (libhalo_hip)");
        return std::make_unique<HipBackendGlue>();  // seeded by the first BeginSession's SessionSpec::seed, like the other backends
      }
      ILOG_WARN(logger, "LUMICE_TRACE_BACKEND=hip requested but no gfx950 device; falling back to legacy CPU");
      return nullptr;
#else
      ILOG_WARN(logger, "LUMICE_TRACE_BACKEND=hip requested but LUMICE_HIP_ENABLED not set; falling back to legacy CPU");
      return nullptr;
#endif
''')
    # the exhaustive switch: -Wswitch forces this case once kHip exists.  It goes behind kCuda's `#endif`, in front of the switch's brace.
    t = after(t, 'ILOG_WARN(logger, "preferred_backend=cuda but LUMICE_CUDA_ENABLED not set; falling back to legacy CPU");\n      return nullptr;\n#endif', '''    case BackendKind::kHip:
#if defined(LUMICE_HIP_ENABLED)
      if (halo_device_count() > 0) {
        ILOG_INFO(logger, "preferred_backend=hip → routing via HipBackendGlue (libhalo_hip)");
        return std::make_unique<HipBackendGlue>();
      }
      ILOG_WARN(logger, "preferred_backend=hip but no gfx950 device; falling back to legacy CPU");
#endif
      return nullptr;
''')
    return t


EDITS["src/core/simulator.cpp"] = simulator


def server(t):
    t = after(t, '#include "core/backend/backend_kind.hpp"' if '#include "core/backend/backend_kind.hpp"' in t else '#include "server/server.hpp"',
              '#if defined(LUMICE_HIP_ENABLED)\n#include "halo_trace.h"  // halo_device_count()\n#endif\n')
    t = before(t, "    // Unknown / unavailable override name", '''#if defined(LUMICE_HIP_ENABLED)
    if (name == "hip") {
      return halo_device_count() > 0;
    }
#endif
''')
    t = before(t, "  return false;  // kCpu, or the requested GPU backend is unavailable in this build", '''#if defined(LUMICE_HIP_ENABLED)
  if (preferred_backend == BackendKind::kHip) {
    return halo_device_count() > 0;  // single-engine sizing, like Metal / CUDA (doc/seam-design.md §5)
  }
#endif
''')
    # the dispatch default (server.cpp:1449-1457): without this a kHip route would inherit Metal's 32768 rays per dispatch, a
    # twentieth of this engine's rate (profiles/r05_dispatch_size.txt).  The route test below repeats ResolveGpuRoute's precedence
    # (env override first, then preferred_backend) so that a build with CUDA and HIP both enabled tells the two apart.
    t = after(t, "static constexpr size_t kDefaultCudaDispatchRayNum = 262144;", '''  // MI355X engine (libhalo_hip): a session of 2^16 rays costs ~20 us however few rays it holds (one latency-bound pass of the ray
  // loop) and the trace kernel needs ~2^22 rays to fill 256 CUs; 2^24 rays per dispatch run at the engine's plateau (~22.6 G rays/s)
  // in ~0.75 ms per dispatch, so UI commit cadence is unaffected (2^18: ~46 % of it, 2^20: ~71 %, 2^22: ~84 %).
  static constexpr size_t kDefaultHipDispatchRayNum = size_t{1} << 24;
''')
    t = before(t, "ServerImpl::ServerImpl(int num_workers, uint32_t sim_seed, BackendKind preferred_backend)", '''#if defined(LUMICE_HIP_ENABLED)
// True when the GPU route ResolveGpuRoute answered for is the HIP engine: same precedence (LUMICE_TRACE_BACKEND, then
// preferred_backend), so a build that enables CUDA and HIP side by side still sizes each route's dispatch by its own default.
bool IsHipRoute(BackendKind preferred_backend, Logger& logger) {
  if (std::optional<std::string> override = env::TraceBackendOverride(logger)) {
    if (*override == "hip") {
      return halo_device_count() > 0;
    }
    if (*override == "cpu_backend" || *override == "legacy" || *override == "metal" || *override == "cuda") {
      return false;
    }
  }
  return preferred_backend == BackendKind::kHip && halo_device_count() > 0;
}
#endif

''')
    t = t.replace('''#if defined(LUMICE_CUDA_ENABLED) && !defined(__APPLE__)
  const bool kIsCudaRoute = kGpuRoute;
#else''', '''#if defined(LUMICE_HIP_ENABLED)
  const bool kIsHipRoute = kGpuRoute && IsHipRoute(kPref, logger_);
#else
  const bool kIsHipRoute = false;
#endif
#if defined(LUMICE_CUDA_ENABLED) && !defined(__APPLE__)
  const bool kIsCudaRoute = kGpuRoute && !kIsHipRoute;
#else''')
    t = t.replace("  const size_t kDefaultDispatch = kIsCudaRoute ? kDefaultCudaDispatchRayNum :\n",
                  "  const size_t kDefaultDispatch = kIsHipRoute  ? kDefaultHipDispatchRayNum :\n                                  kIsCudaRoute ? kDefaultCudaDispatchRayNum :\n")
    assert "kIsHipRoute  ? kDefaultHipDispatchRayNum" in t and "kIsCudaRoute = kGpuRoute && !kIsHipRoute" in t
    return t


EDITS["src/server/server.cpp"] = server


def c_api(t):
    pos = t.index("int LUMICE_IsBackendAvailable(int backend)")
    head, tail = t[:pos], t[pos:]
    tail = before(tail, "    }\n    return 0;\n  } catch (...)", '''      case ns::BackendKind::kHip:
#if defined(LUMICE_HIP_ENABLED)
        return halo_device_count() > 0 ? 1 : 0;
#else
        return 0;
#endif
''')
    t = head + tail
    return after(t, '#include "core/backend/backend_kind.hpp"' if '#include "core/backend/backend_kind.hpp"' in t else '#include "include/lumice.h"',
                 '#if defined(LUMICE_HIP_ENABLED)\n#include "halo_trace.h"  // halo_device_count()\n#endif\n')


EDITS["src/server/c_api.cpp"] = c_api
EDITS["src/include/lumice.h"] = lambda t: after(t, "#define LUMICE_BACKEND_CUDA 2", "#define LUMICE_BACKEND_HIP 3\n")


def cmake(t):
    t = after(t, 'option(LUMICE_CUDA_ENABLED "Enable CUDA backend', '''option(LUMICE_HIP_ENABLED "Enable the MI355X (gfx950) trace backend: needs libhalo_hip.so and its headers" OFF)
set(HALO_HIP_ROOT "" CACHE PATH "checkout of the engine repository (holds include/, ice_halo_sim_amd/csrc/, ice_halo_sim_amd/libhalo_hip.so)")
''')
    t = before(t, "if(LUMICE_CUDA_ENABLED)\n  target_link_libraries(lumice_obj PUBLIC CUDA::cudart)", '''if(LUMICE_HIP_ENABLED)
  if(NOT EXISTS "${HALO_HIP_ROOT}/include/halo_trace.h")
    message(FATAL_ERROR "LUMICE_HIP_ENABLED needs -DHALO_HIP_ROOT=<engine checkout> (include/halo_trace.h not found)")
  endif()
  add_library(halo_hip SHARED IMPORTED)
  set_target_properties(halo_hip PROPERTIES IMPORTED_LOCATION "${HALO_HIP_ROOT}/ice_halo_sim_amd/libhalo_hip.so")
  target_include_directories(lumice_obj PUBLIC "${HALO_HIP_ROOT}/include" "${HALO_HIP_ROOT}/ice_halo_sim_amd/csrc")
  target_link_libraries(lumice_obj PUBLIC halo_hip)
  target_compile_definitions(lumice_obj PUBLIC LUMICE_HIP_ENABLED=1)
  configure_file("${HALO_HIP_ROOT}/integration/hip_backend_glue.hpp" "${CMAKE_SOURCE_DIR}/src/core/backend/hip_backend_glue.hpp" COPYONLY)
endif()

''')
    return t


EDITS["CMakeLists.txt"] = cmake


def main():
    out = ["# Lumice-side patch for a fourth trace backend (this repo's libhalo_hip.so): BackendKind::kHip, both CreateBackend sites, both\n"
           "# ResolveGpuRoute sites, LUMICE_IsBackendAvailable, LUMICE_BACKEND_HIP and the CMake option.  A real unified diff against the reference\n"
           "# tree (one line of context), generated by tools/make_lumice_patch.py; apply with `patch -p1` from the Lumice checkout root\n"
           "# (tools/glue_syntax_check.sh dry-runs it; integration/try_in_lumice.sh is the first-contact script).\n"]
    with tempfile.TemporaryDirectory() as td:
        for rel, fn in EDITS.items():
            a, b = os.path.join(td, "a", rel), os.path.join(td, "b", rel)
            os.makedirs(os.path.dirname(a), exist_ok=True)
            os.makedirs(os.path.dirname(b), exist_ok=True)
            shutil.copy(os.path.join(REF, rel), a)
            text = open(a, encoding="utf-8").read()
            new = fn(text)
            assert new != text, rel
            open(b, "w", encoding="utf-8").write(new)
            r = subprocess.run(["diff", "-U1", "--label", "a/" + rel, "--label", "b/" + rel, a, b], capture_output=True, text=True)
            assert r.returncode == 1, (rel, r.stderr)
            out.append(r.stdout)
    with open(os.path.join(ROOT, "integration", "lumice_hip_backend.patch"), "w", encoding="utf-8") as f:
        f.write("".join(out))
    print("wrote integration/lumice_hip_backend.patch (%d files)" % len(EDITS))


if __name__ == "__main__":
    main()
