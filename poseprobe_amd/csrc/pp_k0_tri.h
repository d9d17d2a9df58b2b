// Trilinear stencil of the channels-last colour grid k0 [X,Y,Z,C] (DenseGrid.forward, lib/grid.py:47-58: align_corners, zeros
// padding): shared by the feature lookup / scatter kernels of pp_color.hip and the deterministic scatter of pp_scatter_sorted.hip.
#pragma once
#include "pp_common.h"

#define PP_FEAT_LD 64
#define PP_PACK_LD 16

struct K0Tri {
  float w0[3], w1[3];
  int i0[3];
  bool ok1[3];  // +1 corner inside the grid (zeros padding otherwise)
  bool ok0[3];
};

__device__ __forceinline__ void k0_setup(const SceneDev& sc, const float p[3], K0Tri& t) {
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float u = pp_grid_u(p[a], sc.mn[a], sc.mx[a], sc.sz[a]);
    float f = floorf(u);
    t.w1[a] = pp_sub(u, f);
    t.w0[a] = pp_sub(pp_add(f, 1.f), u);
    float fc = fminf(fmaxf(f, -2.f), (float)sc.sz[a]);
    int i = (int)fc;
    t.i0[a] = i;
    t.ok0[a] = (i >= 0) && (i < sc.sz[a]);
    t.ok1[a] = (i + 1 >= 0) && (i + 1 < sc.sz[a]);
  }
}

__device__ __forceinline__ bool k0_corner(const SceneDev& sc, const K0Tri& t, int c, size_t& off, float& w) {
  bool ok = ((c & 4) ? t.ok1[0] : t.ok0[0]) && ((c & 2) ? t.ok1[1] : t.ok0[1]) && ((c & 1) ? t.ok1[2] : t.ok0[2]);
  int ix = t.i0[0] + ((c >> 2) & 1), iy = t.i0[1] + ((c >> 1) & 1), iz = t.i0[2] + (c & 1);
  off = (((size_t)ix * sc.sz[1] + iy) * sc.sz[2] + iz) * (size_t)sc.C;
  w = ((c & 4) ? t.w1[0] : t.w0[0]) * ((c & 2) ? t.w1[1] : t.w0[1]) * ((c & 1) ? t.w1[2] : t.w0[2]);
  return ok;
}

