// halo_host.hpp — host-side producers of the tables the gfx950 kernels consume: crystal geometry, latitude
// LUT, projection POD, wavelength pool, ray partition.  Plain C++17, no device code, no torch.
#ifndef HALO_HOST_HPP_
#define HALO_HOST_HPP_

#include <array>
#include <cstdint>
#include <vector>

#include "halo_device.h"
#include "halo_geom.h"

namespace halo {
namespace host {

// math.hpp:21-31 constants (float, as the reference evaluates them)
constexpr float kPi = 3.14159265359f;
constexpr float kPiHalf = kPi / 2.0f;
constexpr float kFloatEps = 1e-5f;
constexpr float kDegToRad = kPi / 180.0f;
constexpr float kSqrt3 = 1.73205080757f;

// Closed-form hexagonal prism → kernel tables. Returns false for the empty crystal
// (Crystal::MakePrismClosedForm crystal.cpp:349-368).
bool BuildPrism(float h, const float dist[6], HaloGeomTables& out);
bool BuildPyramid(float wedge_u_deg, float wedge_l_deg, float h1, float h2, float h3, const float dist[6],
                  HaloGeomTables& out);
void ToShapeDev(const HaloGeomTables& g, ShapeDev& out);
void FromShapeDev(const ShapeDev& s, HaloGeomTables& out);
geom::CrystalRecipe MakeRecipe(const HaloCrystal& c);   // wedge trig evaluated once, on the host

struct LatLut {
  std::array<float, kLutNodes> theta{}, cdf{}, flip{};
};
LatLut BuildLatLut(const HaloDist& latitude);               // lat_lut.cpp:74-204
uint32_t SelectLatPath(const HaloAxis& axis);               // lat_path_selection.hpp:62-75

ProjDev BuildProj(const HaloRender& render);                // lens_proj_build.hpp:79-137
double IceRefractiveIndex(double wavelength_nm);            // optics.cpp:180-197
float IlluminantSpd(int illuminant, float wavelength_nm);   // util/illuminant.cpp:113-134
std::vector<WlEntryDev> BuildWlPool(const HaloWl& wl);      // wl_pool.hpp:67-91
std::vector<uint64_t> Partition(const float* proportions, int n, uint64_t ray_num, double* carry);  // simulator.cpp:519-582

// Emit-gate filter descriptor: BuildDeviceFilterDesc / BuildComplexSubDescs (device_filter_desc.cpp:121-170) with the
// canonical sequences reduced by Crystal::ReduceRaypath (crystal.cpp:536-600).
FilterDev BuildFilter(const HaloFilter& f, const HaloAxis& axis);
std::vector<uint8_t> ReduceRaypath(const std::vector<uint8_t>& rp, uint8_t symmetry, int sigma_a, bool d_applicable);
// Fast form of a filter (nullptr = none) and of a crystal entry's colour predicates for the kernels that hold the path in a
// 128-bit register (halo_device.h FastTables).  false = it does not fit (member / matrix pools full): the caller takes the
// generic kernels.  `crystal_id`: the dispatch's crystal (crystal terms are constants of the dispatch and fold into len_mode).  The class table
// of `out` is the caller's.
bool BuildFastTables(const HaloFilter* filter, const HaloColorSet* colors, const HaloAxis& axis, uint32_t crystal_id, FastTables& out);
// The members of a raypath term: every sequence whose reduction equals `canon` (packed {hi, lo}, newest face in the low byte).
std::vector<std::array<uint64_t, 2>> RaypathMembers(const std::vector<uint8_t>& canon, uint8_t symmetry, int sigma_a, bool d_applicable);
// The fast tables evaluated on the host, step for step like halo_trace.inl fast_filter (test hook).
bool FastFilterCheck(const FastTables& F, const uint8_t* path, uint32_t len, const float dir[3], uint32_t crystal_id);
// the colour pass of the production colour kernels (fast_color_bits) on the host: carried | bits of the matching predicates
uint64_t FastColorMask(const FastTables& F, uint64_t carried, const uint8_t* path, uint32_t len, const float dir[3], uint32_t crystal_id);
int ComputeSigmaA(float roll_center_deg);       // crystal.cpp:720-726
bool IsDApplicable(const HaloAxis& axis);       // crystal.cpp:728-730

bool IsDeterministic(const HaloCrystal& c);                 // simulator.cpp:453-471
// One sampled crystal instance (MakeCrystal simulator.cpp:448 with SyncGroupSampler :361-393), shape scalars
// drawn from the host PCG stream (seed, shape_index).
bool MakeShape(uint32_t seed, const HaloCrystal& c, uint64_t shape_index, HaloGeomTables& out);
bool MakeShapeDev(uint32_t seed, const HaloCrystal& c, uint64_t shape_index, ShapeDev& out);   // same, device table layout

// Tables of the fast entry pick (EntryFastDev); false = the shape is not a full prism in the expected layout.
bool BuildEntryFast(const ShapeDev& s, EntryFastDev& out);

uint32_t PcgHash(uint32_t x);

}  // namespace host
}  // namespace halo

#endif
