// hip_backend_glue.hpp — the ONE file a Lumice maintainer adds (as src/core/backend/hip_backend_glue.hpp) to run the
// MI355X engine of this repo as a fourth trace backend behind `lumice::TraceBackend`.
//
// It is REFERENCE-SIDE code: it includes Lumice headers (config/*.hpp, core/backend/trace_backend.hpp) and therefore cannot
// be compiled in this repository (Lumice's config headers need nlohmann-json >= 3.4 and spdlog, which the build image lacks).
// It is kept as a tracked source, not as prose, and is checked two ways instead:
//   * everything on OUR side of it — include/halo_trace.h and ice_halo_sim_amd/csrc/hip_trace_backend.hpp — is compiled and
//     driven on the GPU by tests/cpp/adapter_main.cpp, and every struct it fills is size-pinned by the static_asserts at the
//     end of halo_trace.h, so a field added on either side breaks the build, not the image;
//   * INTEGRATION.md §2 lists every reference field this file reads, with the reference's file:line, and the Halo* field
//     it lands in — the table a reviewer checks it against.
//
// Every virtual of TraceBackend (src/core/backend/trace_backend.hpp:367-641) is overridden here; what each forwards to is
// stated at the override.  Link with -lhalo_hip; add `kHip` to BackendKind and one case to CreateBackend.
#ifndef CORE_BACKEND_HIP_BACKEND_GLUE_H_
#define CORE_BACKEND_HIP_BACKEND_GLUE_H_

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <random>
#include <string>
#include <type_traits>
#include <utility>
#include <variant>
#include <vector>

#include "config/color_class_table.hpp"
#include <cmath>
#include <cstring>

#include "config/color_gate_table.hpp"
#include "config/crystal_config.hpp"
#include "config/filter_config.hpp"
#include "config/light_config.hpp"
#include "config/proj_config.hpp"
#include "config/raypath_color_config.hpp"
#include "config/render_config.hpp"
#include "core/backend/trace_backend.hpp"
#include "core/crystal.hpp"
#include "core/backend/wl_pool.hpp"
#include "hip_trace_backend.hpp"  // this repo: ice_halo_sim_amd/csrc/hip_trace_backend.hpp (header-only, includes halo_trace.h)

namespace lumice {
namespace hip_glue {

// ---- Distribution / AxisDistribution (core/math.hpp:165-190, 271-310) -------------------------------------------------
inline HaloDist ToHalo(const Distribution& d) { return HaloDist{ static_cast<int32_t>(d.type), d.center, d.spread }; }
inline HaloAxis ToHalo(const AxisDistribution& a) { return HaloAxis{ ToHalo(a.azimuth_dist), ToHalo(a.latitude_dist), ToHalo(a.roll_dist) }; }

// ---- SimpleFilterParam (config/filter_config.hpp:15-41) -> HaloFilterTerm ----------------------------------------------
// Face ids in FilterConfig are crystal face NUMBERS, which is what the device matcher compares (halo_trace.h HaloFilterTerm).
inline HaloFilterTerm ToHalo(const SimpleFilterParam& p) {
  HaloFilterTerm t{};
  std::visit(
      [&](const auto& v) {
        using T = std::decay_t<decltype(v)>;
        if constexpr (std::is_same_v<T, NoneFilterParam>) {
          t.type = HALO_FILTER_NONE;
        } else if constexpr (std::is_same_v<T, RaypathFilterParam>) {
          t.type = HALO_FILTER_RAYPATH;
          t.raypath_len = static_cast<int32_t>(std::min<size_t>(v.raypath_.size(), HALO_MAX_HITS));
          for (int32_t i = 0; i < t.raypath_len; i++) t.raypath[i] = static_cast<uint8_t>(v.raypath_[static_cast<size_t>(i)]);
        } else if constexpr (std::is_same_v<T, EntryExitFilterParam>) {
          t.type = HALO_FILTER_ENTRY_EXIT;
          t.has_entry = v.entry_.has_value() ? 1 : 0;
          t.entry = v.entry_.has_value() ? static_cast<int32_t>(*v.entry_) : 0;
          t.has_exit = v.exit_.has_value() ? 1 : 0;
          t.exit_face = v.exit_.has_value() ? static_cast<int32_t>(*v.exit_) : 0;
          t.min_len = static_cast<uint32_t>(v.min_len_);
          t.max_len = v.max_len_.has_value() ? static_cast<uint32_t>(*v.max_len_) : 0u;  // 0 = unbounded
        } else if constexpr (std::is_same_v<T, DirectionFilterParam>) {
          t.type = HALO_FILTER_DIRECTION;
          t.az = v.lon_;
          t.el = v.lat_;
          t.radii = v.radii_;
        } else {  // CrystalFilterParam
          t.type = HALO_FILTER_CRYSTAL;
          t.crystal_id = static_cast<int32_t>(v.crystal_id_);
        }
      },
      p);
  return t;
}

// FilterConfig (config/filter_config.hpp:45-61) -> HaloFilter.  false = more OR-clauses / terms than one HaloFilter holds
// (64 / 64): the caller answers IsCompatible() == false for such a scene, nothing is truncated.
inline bool ToHalo(const FilterConfig& f, HaloFilter& out) {
  std::memset(&out, 0, sizeof(out));
  out.action = (f.action_ == FilterConfig::kFilterOut) ? 1 : 0;
  out.symmetry = f.symmetry_;
  if (const auto* simple = std::get_if<SimpleFilterParam>(&f.param_)) {
    out.is_complex = 0;
    out.terms[0] = ToHalo(*simple);
    return true;
  }
  const auto& cx = std::get<ComplexFilterParam>(f.param_);
  out.is_complex = 1;
  if (cx.filters_.size() > HALO_FILTER_MAX_OR) return false;
  out.or_count = static_cast<int32_t>(cx.filters_.size());
  int k = 0;
  for (size_t o = 0; o < cx.filters_.size(); o++) {
    out.and_counts[o] = static_cast<int32_t>(cx.filters_[o].size());
    for (const auto& id_and_term : cx.filters_[o]) {
      if (k >= HALO_FILTER_MAX_TERMS) return false;
      out.terms[k++] = ToHalo(id_and_term.second);
    }
  }
  return true;
}

inline bool IsPassAll(const FilterConfig& f) {  // the default ScatteringSetting::filter_: filter_in + NoneFilterParam
  const auto* simple = std::get_if<SimpleFilterParam>(&f.param_);
  return simple && std::holds_alternative<NoneFilterParam>(*simple) && f.action_ == FilterConfig::kFilterIn;
}

// ---- CrystalConfig (config/crystal_config.hpp) -> HaloCrystal ----------------------------------------------------------
// Slot order = RNG draw order: prism h | pyramid upper_h, prism_h, lower_h ; then d[0..5] (simulator.cpp:405-425).
// sync_group_ is indexed by ShapeScalar (crystal_config.hpp:31-45); HaloCrystal::sync_group by [h0, h1, h2, d0..d5].
inline HaloCrystal ToHalo(const CrystalParam& param) {
  HaloCrystal c{};
  std::visit(
      [&](const auto& p) {
        using P = std::decay_t<decltype(p)>;
        if constexpr (std::is_same_v<P, PrismCrystalParam>) {
          c.kind = HALO_CRYSTAL_PRISM;
          c.height[0] = ToHalo(p.h_);
          c.sync_group[0] = p.sync_group_[kShapeScalarHeight];
          c.wedge_upper_deg = c.wedge_lower_deg = 28.0f;
        } else {
          c.kind = HALO_CRYSTAL_PYRAMID;
          c.height[0] = ToHalo(p.h_pyr_u_);
          c.height[1] = ToHalo(p.h_prs_);
          c.height[2] = ToHalo(p.h_pyr_l_);
          c.sync_group[0] = p.sync_group_[kShapeScalarUpperH];
          c.sync_group[1] = p.sync_group_[kShapeScalarPrismH];
          c.sync_group[2] = p.sync_group_[kShapeScalarLowerH];
          c.wedge_upper_deg = p.wedge_angle_u_;  // already resolved from Miller indices by from_json (crystal_config.cpp:371-385)
          c.wedge_lower_deg = p.wedge_angle_l_;
        }
        for (int i = 0; i < 6; i++) {
          c.face_dist[i] = ToHalo(p.d_[i]);
          c.sync_group[3 + i] = p.sync_group_[kShapeScalarFace0 + i];
        }
      },
      param);
  return c;
}

// ---- RenderConfig (config/render_config.hpp:71-97) -> HaloRender -------------------------------------------------------
inline HaloRender ToHalo(const RenderConfig& r) {
  HaloRender o{};
  o.lens_type = static_cast<int32_t>(r.lens_.type_);  // LensParam::LensType values == HALO_LENS_* (projection_shared.h:136-146)
  o.fov = r.lens_.fov_;
  o.width = r.resolution_[0];
  o.height = r.resolution_[1];
  o.lens_shift[0] = r.lens_shift_[0];
  o.lens_shift[1] = r.lens_shift_[1];
  o.view_az = r.view_.az_;
  o.view_el = r.view_.el_;
  o.view_ro = r.view_.ro_;
  o.visible = static_cast<int32_t>(r.visible_);       // kUpper / kLower / kFull == HALO_VISIBLE_*
  o.overlap = r.overlap_;
  return o;
}

// ---- IlluminantType (util/illuminant_data.hpp:12-19) -> HALO_ILLUM_* (same order) ---------------------------------------
inline int32_t ToHalo(IlluminantType t) { return static_cast<int32_t>(t); }

// ---- Crystal (core/crystal.hpp: CfGeom(), the closed-form POD) -> HaloGeomTables: compact present faces with their unit normals, plane
// distances (d / |n|, PopulateFromCfGeom crystal.cpp:304-347) and face numbers, and the entry fan triangles (0, k, k+1) of each face with
// normal and area exactly as detail::BuildEntrySubTris forms them (simulator.cpp:90-129).  False when the crystal has no closed-form
// geometry or exceeds the engine's caps (then the entry's own crystal is traced).
inline bool ToHalo(const Crystal& crystal, HaloGeomTables& g) {
  const CrystalGeom& cf = crystal.CfGeom();
  g = HaloGeomTables{};
  if (cf.face_cnt <= 0) return false;
  int32_t faces = 0, tris = 0;
  for (int slot = 0; slot < cf.face_cnt; slot++) {
    if (!cf.face_present[slot]) continue;
    if (faces >= HALO_MAX_FACES) return false;
    const float* pc = cf.plane_coef + slot * 4;
    const float len = std::sqrt(pc[0] * pc[0] + pc[1] * pc[1] + pc[2] * pc[2]);
    for (int a = 0; a < 3; a++) g.face_n[faces * 3 + a] = cf.face_normal[slot * 3 + a];
    g.face_d[faces] = len > 0.0f ? pc[3] / len : 0.0f;
    g.face_number[faces] = cf.face_number[slot];
    const float* base = cf.face_vtx + static_cast<size_t>(slot) * kCrystalGeomMaxVtxPerFace * 3;
    for (int k = 1; k + 1 < cf.face_vtx_cnt[slot]; k++) {
      if (tris >= HALO_MAX_TRIS) return false;
      float* v = g.tri_v + tris * 9;
      std::memcpy(v + 0, base + 0, 3 * sizeof(float));
      std::memcpy(v + 3, base + k * 3, 3 * sizeof(float));
      std::memcpy(v + 6, base + (k + 1) * 3, 3 * sizeof(float));
      const float e1[3] = { v[3] - v[0], v[4] - v[1], v[5] - v[2] }, e2[3] = { v[6] - v[0], v[7] - v[1], v[8] - v[2] };
      float n[3] = { e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0] };
      const float raw = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
      g.tri_area[tris] = raw / 2.0f;
      for (int a = 0; a < 3; a++) g.tri_n[tris * 3 + a] = raw > 0.0f ? n[a] / raw : 0.0f;
      g.tri_face[tris] = faces;
      tris++;
    }
    faces++;
  }
  g.face_cnt = faces;
  g.tri_cnt = tris;
  return faces > 0;
}

// ---- SceneConfig (config/proj_config.hpp:15-38) -> HaloScene + filter table + colour tables ------------------------------
struct SceneTables {
  HaloScene scene{};
  std::vector<HaloFilter> filters;        // HaloEntry::filter_id indexes this table (1-based)
  std::vector<HaloColorSet> color_sets;   // HaloEntry::color_id indexes this table (1-based)
  std::vector<HaloColorClass> color_classes;
  bool representable = true;              // false -> IsCompatible() answers false (caps exceeded; never truncated)
  // raypath colour past the engine's caps (HALO_COLOR_MAX_CLASSES classes per session, HALO_COLOR_MAX_TERMS predicates per placement):
  // the tables above hold what fits, these count what did not.  BeginSession refuses such a scene (default) or renders it degraded and
  // reports the counts like the reference's GPU backends do (LUMICE_HIP_COLOR_OVERFLOW=degrade; cuda_trace_backend.cu:3296-3304, :3363-3380)
  size_t color_class_overflow = 0;
  size_t color_term_overflow = 0;
};

inline SceneTables ToHalo(const SceneConfig& s, const RaypathColorConfig* color) {
  SceneTables out;
  HaloScene& o = out.scene;
  o.sun_altitude = s.light_source_.param_.altitude_;
  o.sun_azimuth = s.light_source_.param_.azimuth_;
  o.sun_diameter = s.light_source_.param_.diameter_;
  o.max_hits = static_cast<int32_t>(s.max_hits_);
  if (s.ms_.size() > HALO_MAX_LAYERS) out.representable = false;
  o.layer_count = static_cast<int32_t>(std::min<size_t>(s.ms_.size(), HALO_MAX_LAYERS));

  // raypath colour: the reference's own builders assign the component bits and the class masks (color_gate_table.hpp:100,
  // color_class_table.hpp:68); this glue only re-shapes their output.
  ColorGateTable gate;
  if (color != nullptr && !color->classes_.empty()) {
    gate = BuildColorGateTable(*color, s);
    const ColorClassTable classes = BuildColorClassTable(*color, s, gate);
    if (classes.classes_.size() > HALO_COLOR_MAX_CLASSES) out.color_class_overflow = classes.classes_.size() - HALO_COLOR_MAX_CLASSES;
    for (size_t c = 0; c < classes.classes_.size() && c < HALO_COLOR_MAX_CLASSES; c++) {
      HaloColorClass hc{};
      hc.bits = classes.classes_[c].member_bits_;
      hc.combine_all = (classes.classes_[c].combine_ == ColorClassCombine::kAll) ? 1 : 0;
      out.color_classes.push_back(hc);
    }
  }

  for (int32_t l = 0; l < o.layer_count; l++) {
    const MsInfo& ms = s.ms_[static_cast<size_t>(l)];
    HaloLayer& hl = o.layers[l];
    hl.prob = ms.prob_;
    if (ms.setting_.size() > HALO_MAX_ENTRIES) out.representable = false;
    hl.entry_count = static_cast<int32_t>(std::min<size_t>(ms.setting_.size(), HALO_MAX_ENTRIES));
    for (int32_t e = 0; e < hl.entry_count; e++) {
      const ScatteringSetting& st = ms.setting_[static_cast<size_t>(e)];
      HaloEntry& he = hl.entries[e];
      he.crystal = ToHalo(st.crystal_.param_);
      he.axis = ToHalo(st.crystal_.axis_);
      he.proportion = st.crystal_proportion_;
      he.crystal_config_id = static_cast<int32_t>(st.crystal_.id_);
      he.filter_id = 0;
      if (!IsPassAll(st.filter_)) {                       // ScatteringSetting::filter_ (proj_config.hpp:17)
        HaloFilter hf;
        if (!ToHalo(st.filter_, hf)) out.representable = false;
        out.filters.push_back(hf);
        he.filter_id = static_cast<int32_t>(out.filters.size());
      }
      he.color_id = 0;
      if (!out.color_classes.empty()) {                   // this placement's predicates (color_gate_table.hpp:109-113)
        const ColorGatePlacement pl = ColorGatePlacementFor(gate, static_cast<IdType>(l), st.crystal_.id_);
        if (!pl.predicates_.empty()) {
          if (pl.predicates_.size() > HALO_COLOR_MAX_TERMS) out.color_term_overflow += pl.predicates_.size() - HALO_COLOR_MAX_TERMS;
          HaloColorSet cs{};
          cs.term_count = static_cast<int32_t>(std::min<size_t>(pl.predicates_.size(), HALO_COLOR_MAX_TERMS));
          for (int32_t k = 0; k < cs.term_count; k++) {
            cs.terms[k].predicate = ToHalo(pl.predicates_[static_cast<size_t>(k)]);
            cs.terms[k].symmetry = pl.symmetries_[static_cast<size_t>(k)];
            cs.terms[k].bit = pl.bits_[static_cast<size_t>(k)];
          }
          out.color_sets.push_back(cs);
          he.color_id = static_cast<int32_t>(out.color_sets.size());
        }
      }
    }
  }
  return out;
}

}  // namespace hip_glue

// =====================================================================================================================
class HipBackendGlue final : public TraceBackend {
 public:
  // CreateBackend (simulator.cpp:854-919) knows no seed: like the reference's backends (cuda_trace_backend.cu:1931-1943,
  // cpu_trace_backend.cpp:252-257) the engine is seeded ONCE, by the first BeginSession's SessionSpec::seed (0 = non-deterministic), and its
  // ray counters then run on across sessions.  `seed` != 0 here pins it instead (tests).  device = the process's GPU (one engine per
  // process; a multi-GPU launcher passes its local rank).  Throws BackendUnavailableError without a gfx950 device.
  explicit HipBackendGlue(uint32_t seed = 0u, int device = 0) : device_(device), pinned_seed_(seed) {
    if (halo_device_count() <= device) throw BackendUnavailableError("no gfx950 device for the HIP trace backend");
    if (seed != 0u) {
      Create(seed);
      seeded_ = true;
    }
  }

  // --- TraceBackend::BeginSession (trace_backend.hpp:374-378) ----------------------------------------------------------
  void BeginSession(const SessionSpec& spec) override {
    // Seeded ONCE per backend lifetime by the first NON-ZERO SessionSpec::seed (cpu_trace_backend.cpp:248-257: `spec.seed != 0 &&
    // !seeded_`): sessions with seed 0 run on a non-deterministic seed and leave the engine unseeded, so a later fixed seed still applies.
    if (!be_) {
      Create(spec.seed != 0u ? spec.seed : (static_cast<uint32_t>(std::random_device{}()) | 1u));
      seeded_ = spec.seed != 0u;
    } else if (!seeded_ && spec.seed != 0u) {
      Guard([&] { be_->SetOption("seed", static_cast<int64_t>(spec.seed)); });   // halo_set_option "seed": between sessions only
      seeded_ = true;
    }
    hip_glue::SceneTables t = hip_glue::ToHalo(*spec.scene, spec.raypath_color.get());       // SessionSpec::raypath_color
    if (!t.representable) throw BackendUnavailableError("scene exceeds the HIP backend's table caps (layers/entries/filter terms)");
    // Colour past the caps.  The reference's GPU backends drop the excess, log it and count it (GetLastColorDegradeCounts, trace_backend.hpp:626-632);
    // the CPU path has no caps.  Default here: refuse, so that the simulator's per-Run fallback (simulator.cpp:1049-1062) renders the scene
    // in full on the legacy path.  LUMICE_HIP_COLOR_OVERFLOW=degrade keeps the run on the GPU instead: classes past the 16th are dropped
    // exactly like the reference's (same cap, same count), a placement's predicates past the 16th are dropped and counted as OR-summands
    // (this engine has no per-symmetry-group cap, so symmetry_group_overflow stays 0).  Recomputed per BeginSession like the reference's.
    last_color_degrade_ = ColorDegradeCounts{};
    if (t.color_class_overflow != 0 || t.color_term_overflow != 0) {
      const char* mode = std::getenv("LUMICE_HIP_COLOR_OVERFLOW");
      if (mode == nullptr || std::string(mode) != "degrade")
        throw BackendUnavailableError("raypath colour exceeds the HIP backend's caps (16 classes, 16 predicates per placement); "
                                      "LUMICE_HIP_COLOR_OVERFLOW=degrade renders it with the excess dropped and counted");
      last_color_degrade_.color_class_overflow = t.color_class_overflow;
      last_color_degrade_.or_summand_overflow = t.color_term_overflow;
    }
    const HaloRender rd = hip_glue::ToHalo(*spec.render);
    HaloWl wl{};
    if (const auto* ill = std::get_if<IlluminantType>(&spec.scene->light_source_.spectrum_)) {
      // illuminant mode: the simulator passes a zero WlParam and expects a per-ray pool (simulator.cpp:1064-1083)
      wl.illuminant = hip_glue::ToHalo(*ill);
      wl.pool_size = static_cast<int32_t>(WlPoolSize());
    } else {
      wl.wavelength = spec.wl.wl_;       // WlParam (light_config.hpp:19-22)
      wl.weight = spec.wl.weight_;
      wl.illuminant = -1;
    }
    Guard([&] {
      if (spec.scene->geom_clock_ > 0) be_->SetOption("geom_clock", static_cast<int64_t>(spec.scene->geom_clock_));  // SceneConfig::geom_clock_
      be_->SetFilters(t.filters);                              // halo_set_filters: table referenced by HaloEntry::filter_id
      be_->SetColor(t.color_sets, t.color_classes);            // halo_set_color: empty tables switch colour off
      be_->BeginSession(t.scene, rd, wl, spec.ray_num);        // SessionSpec::ray_num
    });
  }

  // --- TraceLayer (trace_backend.hpp:380-389) ----------------------------------------------------------------------------
  // Host mode: `count` roots generated on the device; if the caller supplies pre-sampled rays (HostRayBatch.d/p/w/tf, the
  // reserved external-ingest path, :230-239) they are forwarded as crystal-local golden rays.  Device mode: consumes the
  // continuation of the preceding Recombine (it never left the backend).
  LayerHandlePtr TraceLayer(const RootRaySource& roots) override {
    Need("TraceLayer");
    halo::LayerHandle lh;
    Guard([&] {
      if (roots.is_device) {
        lh = be_->TraceLayer(0);
      } else if (roots.host.d && roots.host.p && roots.host.w && roots.host.tf) {
        std::vector<uint32_t> tf(roots.host.count);
        for (size_t i = 0; i < roots.host.count; i++) tf[i] = static_cast<uint32_t>(roots.host.tf[i]);   // IdType -> u32
        // HostRayBatch::crystal (trace_backend.hpp:230-239): the crystal the rays were sampled on rides along as the engine's geometry tables
        HaloGeomTables geom{};
        const bool has_crystal = roots.host.crystal != nullptr && hip_glue::ToHalo(*roots.host.crystal, geom);
        const HaloHostRays hr{ roots.host.d, roots.host.p, roots.host.w, tf.data(), has_crystal ? &geom : nullptr };
        lh = be_->TraceLayer(roots.host.count, &hr);
      } else {
        lh = be_->TraceLayer(roots.host.count);
      }
    });
    return std::make_unique<Handle>(lh);
  }

  // --- Recombine (trace_backend.hpp:391-395): pools swap roles inside the backend; the shuffle is applied by the next layer --
  RootRaySource Recombine(LayerHandlePtr h, const RecombineSpec& spec) override {
    Need("Recombine");
    size_t n = 0;
    Guard([&] { n = be_->Recombine(static_cast<Handle&>(*h).lh, spec.shuffle); });
    return RootRaySource::FromDevice(DeviceRayBatch{ nullptr, n });
  }

  // --- DrainExits (trace_backend.hpp:430-448): a device-accumulating backend materialises no exit records -------------------
  size_t DrainExits(std::vector<ExitRayRecord>& out) override {
    out.clear();
    return 0;
  }

  bool SupportsDeviceXyzAccum() const override { return true; }     // trace_backend.hpp:450-459
  bool SupportsThirdClockDrain() const override { return true; }    // accumulator persists across sessions (:495-509)

  // --- ReadbackXyzAccum (trace_backend.hpp:461-469): sync, ADD landed weight, copy W*H*3 floats, zero the device image -------
  void ReadbackXyzAccum(XyzImageData& xyz, float& landed_weight) override {
    if (!be_) return;   // no session yet: nothing accumulated, nothing to add (the engine is created by the first BeginSession)
    halo::XyzImageData x{ xyz.data, xyz.width, xyz.height };
    Guard([&] { be_->ReadbackXyzAccum(x, landed_weight); });
  }

  // --- ReadbackClassLanes (trace_backend.hpp:471-493): lane c at lane_data[c*W*H + py*W + px]; zeroes the device lanes -------
  void ReadbackClassLanes(std::vector<float>& lane_data, size_t& class_count) override {
    if (!be_) {
      lane_data.clear();
      class_count = 0;
      return;
    }
    Guard([&] { be_->ReadbackClassLanes(lane_data, class_count); });   // class_count == 0: lane_data stays empty (zero-cost default)
  }

  void EndSession() override {
    if (be_) Guard([&] { be_->EndSession(); });
  }

  // --- IsCompatible (trace_backend.hpp:511-519, called at simulator.cpp:946): every lens / visible range is supported; what
  // the backend refuses is an image beyond 2^25 pixels (the LDS pixel-cache key width) ---------------------------------------
  bool IsCompatible(const RenderConfig& render) const override {
    return render.resolution_[0] > 0 && render.resolution_[1] > 0 &&
           static_cast<uint64_t>(render.resolution_[0]) * static_cast<uint64_t>(render.resolution_[1]) <= (1ull << 25);
  }

  // --- WlPoolSize (trace_backend.hpp:521): > 0 = the backend samples the wavelength per ray from an M-entry pool in
  // illuminant mode (simulator.cpp:1076, :1652); kWlPoolSizeDefault / LUMICE_WL_POOL_SIZE semantics of wl_pool.hpp:36-48 ------
  uint32_t WlPoolSize() const override { return kWlPoolSizeDefault; }

  // --- sample-count getters (trace_backend.hpp:587, :625): real counts of what the kernels drew in the last session ----------
  size_t GetLastBatchStochasticCrystalSampleCount() const override { return be_ ? be_->GetLastBatchStochasticCrystalSampleCount() : 0; }
  size_t GetLastBatchStochasticOrientationSampleCount() const override { return be_ ? be_->GetLastBatchStochasticOrientationSampleCount() : 0; }
  // GetLastColorDegradeCounts (trace_backend.hpp:632): all zeros unless LUMICE_HIP_COLOR_OVERFLOW=degrade let a scene past the caps through
  // (BeginSession above); otherwise nothing degrades — a scene beyond the colour caps is
  // refused in BeginSession (BackendUnavailableError -> per-Run fallback to the legacy path, simulator.cpp:1049-1062)
  ColorDegradeCounts GetLastColorDegradeCounts() const override { return last_color_degrade_; }

 private:
  struct Handle : LayerHandle {
    explicit Handle(halo::LayerHandle h) : lh(h) {}
    halo::LayerHandle lh;
    size_t ContinuationCount() const override { return lh.ContinuationCount(); }
    // (a queued last layer — option "async", see Create — reports zeros here: its tallies arrive with the next synchronising call;
    // the production caller reads neither, simulator.cpp:1527-1545)
    LayerStats GetLayerStats() const override {   // trace_backend.hpp:296-299, :320
      return LayerStats{ static_cast<size_t>(lh.stats.exit_count), static_cast<float>(lh.stats.exit_w_sum) };
    }
  };
  template <class F>
  void Guard(F&& f) {
    try {
      f();
    } catch (const halo::BackendUnavailableError& e) {
      throw BackendUnavailableError(e.what());
    }
  }
  void Need(const char* what) const {   // a call that only makes sense inside a session, before any BeginSession created the engine
    if (!be_) throw BackendUnavailableError(std::string(what) + " before BeginSession");
  }
  void Create(uint32_t seed) {
    try {
      be_ = std::make_unique<halo::HipTraceBackend>(device_, seed);
      // The caller never looks at the LAST layer's handle (simulator.cpp:1527-1545: it goes out of scope; GetLayerStats has no production
      // reader) and reads the image on its own clock (third-clock drain, :1585-1611), so the last layer's dispatches are queued, not waited
      // for: back-to-back sessions then overlap the host's work with the device's.  Layers that feed a Recombine still return their
      // continuation count synchronously.
      be_->SetOption("async", 1);
      // LUMICE_HIP_NEXT_FACE=cpu: follow the legacy CPU path's next-face strategy (PropagateSlab's relaxed threshold on the child that leaves
      // through its own face, optics.cpp:116-155) instead of the CUDA backend's source-face skip — ray-for-ray comparisons against the
      // legacy backend on crystals whose tables are not clean polytopes; several times slower (generic kernels).  Default: the CUDA strategy.
      if (const char* nf = std::getenv("LUMICE_HIP_NEXT_FACE"))
        if (std::string(nf) == "cpu") be_->SetOption("rehit_strategy", 0);
    } catch (const halo::BackendUnavailableError& e) {
      throw BackendUnavailableError(e.what());
    }
  }
  int device_ = 0;
  uint32_t pinned_seed_ = 0u;   // constructor seed (tests): the engine exists and is seeded from the start
  bool seeded_ = false;          // a non-zero seed has been applied (constructor or a SessionSpec)
  ColorDegradeCounts last_color_degrade_{};   // of the last BeginSession (all zeros unless LUMICE_HIP_COLOR_OVERFLOW=degrade)
  std::unique_ptr<halo::HipTraceBackend> be_;
};

}  // namespace lumice

#endif  // CORE_BACKEND_HIP_BACKEND_GLUE_H_
