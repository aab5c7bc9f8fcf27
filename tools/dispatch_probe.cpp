// What a C++ caller (Lumice's Simulator::SimulateOneWavelengthWithBackend) gets per BeginSession / TraceLayer / EndSession from this
// engine at small session sizes: wall time per session from C++ (no Python in the loop), configs[1]'s scene, one wavelength, 1920x1080.
//   g++ -std=c++17 -O2 -o tools/dispatch_probe.bin tools/dispatch_probe.cpp -Lice_halo_sim_amd -lhalo_hip -Wl,-rpath,$PWD/ice_halo_sim_amd
//   tools/dispatch_probe.bin [reps at 2^18] [key=value backend options ...]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../ice_halo_sim_amd/csrc/hip_trace_backend.hpp"

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  const int reps = argc > 1 ? std::atoi(argv[1]) : 400;
  HaloScene sc;
  std::memset(&sc, 0, sizeof(sc));
  sc.sun_altitude = 20.0f;
  sc.sun_diameter = 0.5f;
  sc.max_hits = 7;
  sc.layer_count = 1;
  sc.layers[0].entry_count = 1;
  HaloEntry& e = sc.layers[0].entries[0];
  e.crystal.kind = HALO_CRYSTAL_PRISM;
  e.crystal.height[0] = {HALO_DIST_NONE, 1.3f, 0.0f};
  for (int i = 0; i < 6; i++) e.crystal.face_dist[i] = {HALO_DIST_NONE, 1.0f, 0.0f};
  e.axis.azimuth = {HALO_DIST_UNIFORM, 0.0f, 360.0f};
  e.axis.latitude = {HALO_DIST_GAUSS, 0.0f, 0.3f};
  e.axis.roll = {HALO_DIST_UNIFORM, 0.0f, 360.0f};
  e.proportion = 1.0f;
  e.crystal_config_id = 3;
  HaloRender rd;
  std::memset(&rd, 0, sizeof(rd));
  rd.lens_type = HALO_LENS_FISHEYE_EQUAL_AREA;
  rd.fov = 180.0f;
  rd.width = 1920;
  rd.height = 1080;
  rd.view_el = 30.0f;
  rd.visible = HALO_VISIBLE_UPPER;
  HaloWl wl = {550.0f, 1.0f, -1, 0};
  try {
    const char* only = std::getenv("PROBE_ASYNC_ONLY");
    for (int async = only ? 1 : 0; async <= 1; async++) {
      halo::HipTraceBackend be(0, 42);
      be.SetOption("async", async);
      for (int a = 2; a < argc; a++) {
        std::string kv(argv[a]);
        const size_t eq = kv.find('=');
        if (eq != std::string::npos) be.SetOption(kv.substr(0, eq).c_str(), std::atoll(kv.c_str() + eq + 1));
      }
      std::vector<float> img(static_cast<size_t>(rd.width) * rd.height * 3);
      const char* sizes_env = std::getenv("PROBE_SIZES");   // e.g. "20 21 22 23"
      std::vector<int> sizes = {15, 16, 17, 18, 20, 22, 24};
      if (sizes_env) {
        sizes.clear();
        for (const char* p = sizes_env; *p;) {
          char* end = nullptr;
          const long v = std::strtol(p, &end, 10);
          if (end == p) break;
          sizes.push_back(static_cast<int>(v));
          p = end;
        }
      }
      for (int lg : sizes) {
        const size_t n = size_t{1} << lg;
        const int k = std::max(3, lg <= 18 ? reps : reps >> (lg - 18));
        double best = 1e30, landed_sum = 0.0;
        for (int rep = 0; rep < 3; rep++) {
          (void)be.CollectStats();
          const double t0 = now_s();
          for (int i = 0; i < k; i++) {
            be.BeginSession(sc, rd, wl, n);
            be.TraceLayer(n);
            be.EndSession();
          }
          (void)be.CollectStats();   // waits for everything queued
          const double dt = now_s() - t0;
          best = dt < best ? dt : best;
          float landed = 0.0f;
          halo::XyzImageData xyz{img.data(), rd.width, rd.height};
          be.ReadbackXyzAccum(xyz, landed);
          landed_sum = landed;
        }
        std::printf("c++ async=%d  2^%d rays per session: %.1f us per session wall, %.2f G rays/s   (landed per ray %.4f)\n", async, lg, best * 1e6 / k,
                    static_cast<double>(k) * n / best / 1e9, landed_sum / (static_cast<double>(k) * n));
        std::fflush(stdout);
      }
    }
  } catch (const std::exception& ex) {
    std::printf("error: %s\n", ex.what());
    return 1;
  }
  return 0;
}
