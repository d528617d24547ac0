// Device helpers shared by the voxel-map build and the pass kernels: voxel coordinates, packed keys, probing.
#pragma once
#include "rolo_internal.hpp"
#include "dev_math.hpp"

namespace rolo {

// vmp_voxel.hpp:199-201 (UNIFORM) and :208-211 (POLAR), fp64, true divisions as written in the reference.
ROLO_DEV void voxel_coord_dev(const VoxelTable& tab, double x, double y, double z, int& kx, int& ky, int& kz) {
  if (tab.voxel_type == ROLO_VOXEL_POLAR) {
    const double r = sqrt((x * x + y * y) + z * z);
    kx = (int)floor((atan2(y, x) + 3.14159265358979323846) / tab.polar_res[0]);
    ky = (int)floor(acos(z / r) / tab.polar_res[1]);
    kz = (int)floor(r / tab.polar_res[2]);
  } else {
    kx = (int)floor(x / tab.voxel_resolution - 0.5);
    ky = (int)floor(y / tab.voxel_resolution - 0.5);
    kz = (int)floor(z / tab.voxel_resolution - 0.5);
  }
}

// The same, also telling whether a POLAR coordinate lies within 1e-12 (in bins) of a bin edge: device atan2 / acos differ from glibc's by
// ulps, so only there could the integer key differ from the CPU path's (SURVEY §7 "hard parts"); counted by the map build, asserted 0 by
// the tests. UNIFORM keys are a division and a floor — correctly rounded on both sides, no hazard, nothing counted.
ROLO_DEV void voxel_coord_dev_edge(const VoxelTable& tab, double x, double y, double z, int& kx, int& ky, int& kz, bool& near_edge) {
  near_edge = false;
  if (tab.voxel_type == ROLO_VOXEL_POLAR) {
    const double r = sqrt((x * x + y * y) + z * z);
    const double a = (atan2(y, x) + 3.14159265358979323846) / tab.polar_res[0], b = acos(z / r) / tab.polar_res[1], c = r / tab.polar_res[2];
    const double fa = floor(a), fb = floor(b), fc = floor(c);
    kx = (int)fa; ky = (int)fb; kz = (int)fc;
    const double lo = 1e-12, hi = 1.0 - 1e-12;
    near_edge = (a - fa) < lo || (a - fa) > hi || (b - fb) < lo || (b - fb) > hi || (c - fc) < lo || (c - fc) > hi;
  } else {
    kx = (int)floor(x / tab.voxel_resolution - 0.5);
    ky = (int)floor(y / tab.voxel_resolution - 0.5);
    kz = (int)floor(z / tab.voxel_resolution - 0.5);
  }
}

ROLO_DEV bool pack_key(int kx, int ky, int kz, unsigned long long& key) {
  const unsigned ux = (unsigned)(kx + KEY_BIAS), uy = (unsigned)(ky + KEY_BIAS), uz = (unsigned)(kz + KEY_BIAS);
  if ((ux | uy | uz) >> 21) return false;
  key = (unsigned long long)ux | ((unsigned long long)uy << 21) | ((unsigned long long)uz << 42);
  return true;
}
ROLO_DEV void unpack_key(unsigned long long key, int& kx, int& ky, int& kz) {
  kx = (int)(key & 0x1fffffu) - KEY_BIAS;
  ky = (int)((key >> 21) & 0x1fffffu) - KEY_BIAS;
  kz = (int)((key >> 42) & 0x1fffffu) - KEY_BIAS;
}

ROLO_DEV unsigned hash_key(unsigned long long k) {  // splitmix64 finaliser
  k ^= k >> 30; k *= 0xbf58476d1ce4e5b9ull;
  k ^= k >> 27; k *= 0x94d049bb133111ebull;
  k ^= k >> 31;
  return (unsigned)k;
}

// lookup_voxel (vmp_voxel.hpp:226-233): compact voxel id or -1
ROLO_DEV int voxel_lookup(const VoxelTable& tab, int kx, int ky, int kz) {
  unsigned long long key;
  if (!pack_key(kx, ky, kz, key)) return -1;
  unsigned h = hash_key(key) & tab.mask;
  while (true) {
    const unsigned long long cur = tab.keys[h];
    if (cur == key) return tab.ids[h];
    if (cur == KEY_EMPTY) return -1;
    h = (h + 1) & tab.mask;
  }
}

}  // namespace rolo
