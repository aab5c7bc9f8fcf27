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
