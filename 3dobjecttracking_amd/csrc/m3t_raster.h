// m3t_raster.h -- the rasterisation arithmetic of the focused renderers (m3t_render.hip), free of device-only
// constructs so that tests/cpp/raster_check.cpp can run it on the host: triangle set-up with OpenGL's rules (window
// coordinates snapped to 1/256 pixel, pixel centres at integer image coordinates, top-left fill rule; renderer.cpp,
// basic_depth_renderer.cpp:45-84), the per-pixel coverage + depth word of the oracle (raster_pixel: the definition),
// and the row scan the kernels use (raster_row: the same words for the same pixels, found with three additions per
// pixel instead of three edge functions, and leaving a row when the covered span has been passed).
//
// Exactness: the snapped coordinates are integers below 2^25 in magnitude, every edge-function value is an integer
// below 2^53 and therefore exact in f64 -- whether it is evaluated directly or carried from the neighbouring pixel
// by adding the (integer) step -256 * ey.  Coverage and depth words of the two forms are bit-identical.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define M3T_RASTER_FN __device__ __forceinline__
#define M3T_RASTER_MEMBER __device__ __forceinline__
#else
#define M3T_RASTER_FN static inline
#define M3T_RASTER_MEMBER inline
#endif

struct RasterM44 {
  float m[16];  // column-major
  M3T_RASTER_MEMBER float operator()(int r, int c) const { return m[c * 4 + r]; }
  M3T_RASTER_MEMBER float& operator()(int r, int c) { return m[c * 4 + r]; }
};

// One triangle after projection, snapping, culling: vertices re-ordered to positive area
struct RasterTriangle {
  double ax[3], ay[3], z[3], area;
  double rcp_area;  // 1 / area, correctly rounded (raster_quotient)
  int x0, x1, y0, y1;
};

// A vertex after projection and snapping: window coordinates in 1/256 pixel (integers, exact in f64 and -- inside the
// range the set-up accepts -- in int32) and the window depth.  false: the vertex lies behind the eye or so far off the
// image that the exact-integer range would be left; a triangle with such a vertex is dropped (no near-plane clipping).
M3T_RASTER_FN bool raster_vertex(const RasterM44& trans, const float* p, int S, double* sx, double* sy, float* wz) {
  const float half_s = 0.5f * (float)S;
  float cx = ((trans(0, 0) * p[0] + trans(0, 1) * p[1]) + trans(0, 2) * p[2]) + trans(0, 3);
  float cy = ((trans(1, 0) * p[0] + trans(1, 1) * p[1]) + trans(1, 2) * p[2]) + trans(1, 3);
  float cz = ((trans(2, 0) * p[0] + trans(2, 1) * p[1]) + trans(2, 2) * p[2]) + trans(2, 3);
  float cw = ((trans(3, 0) * p[0] + trans(3, 1) * p[1]) + trans(3, 2) * p[2]) + trans(3, 3);
  float wx = (cx / cw + 1.0f) * half_s;
  float wy = (cy / cw + 1.0f) * half_s;
  *wz = (cz / cw + 1.0f) * 0.5f;
  *sx = floor((double)wx * 256.0 + 0.5);
  *sy = floor((double)wy * 256.0 + 0.5);
  // behind the eye | anything this far off the image cannot touch it and would leave the exact-integer range
  return cw > 0.0f && fabs(*sx) < 3.0e7 && fabs(*sy) < 3.0e7;
}
// The triangle of three such vertices (all of them accepted by raster_vertex): culling, orientation, bounding box.
M3T_RASTER_FN bool raster_setup_snapped(const double (&sx)[3], const double (&sy)[3], const float (&wz)[3], bool culling,
                                        int S, RasterTriangle& o) {
  double area = (sx[1] - sx[0]) * (sy[2] - sy[0]) - (sy[1] - sy[0]) * (sx[2] - sx[0]);
  if (area == 0.0) return false;
  // counter-clockwise meshes seen from outside have negative area in the y-down image
  if (area > 0.0 && culling) return false;
  int i1 = 1, i2 = 2;
  if (area < 0.0) { i1 = 2; i2 = 1; area = -area; }
  o.ax[0] = sx[0]; o.ax[1] = sx[i1]; o.ax[2] = sx[i2];
  o.ay[0] = sy[0]; o.ay[1] = sy[i1]; o.ay[2] = sy[i2];
  o.z[0] = (double)wz[0]; o.z[1] = (double)wz[i1]; o.z[2] = (double)wz[i2];
  o.area = area;
  o.rcp_area = 1.0 / area;
  const double min_x = fmin(o.ax[0], fmin(o.ax[1], o.ax[2])), max_x = fmax(o.ax[0], fmax(o.ax[1], o.ax[2]));
  const double min_y = fmin(o.ay[0], fmin(o.ay[1], o.ay[2])), max_y = fmax(o.ay[0], fmax(o.ay[1], o.ay[2]));
  // pixels whose centre (256 p + 128) lies inside the bounding box: nothing else can pass the edge tests
  o.x0 = (int)fmax(ceil((min_x - 128.0) / 256.0), 0.0);
  o.x1 = (int)fmin(floor((max_x - 128.0) / 256.0), (double)(S - 1));
  o.y0 = (int)fmax(ceil((min_y - 128.0) / 256.0), 0.0);
  o.y1 = (int)fmin(floor((max_y - 128.0) / 256.0), (double)(S - 1));
  return o.x1 >= o.x0 && o.y1 >= o.y0;
}
// (the triangle by the coordinates of its three vertices, xyz xyz xyz: a caller that walks a triangle list can have the
// next triangles' indices and vertices on their way while this one is set up)
M3T_RASTER_FN bool raster_setup_vertices(const RasterM44& trans, const float (&xyz)[9], bool culling, int S,
                                         RasterTriangle& o) {
  double sx[3], sy[3];
  float wz[3];
  bool ok = true;
  for (int k = 0; k < 3; ++k) ok = raster_vertex(trans, xyz + 3 * k, S, &sx[k], &sy[k], &wz[k]) && ok;
  return ok && raster_setup_snapped(sx, sy, wz, culling, S, o);
}
M3T_RASTER_FN bool raster_setup(const RasterM44& trans, const float* vertices, const int* triangles, int t, bool culling,
                                int S, RasterTriangle& o) {
  float xyz[9];
  for (int k = 0; k < 3; ++k) {
    const float* p = vertices + (size_t)triangles[t * 3 + k] * 3;
    xyz[3 * k] = p[0]; xyz[3 * k + 1] = p[1]; xyz[3 * k + 2] = p[2];
  }
  return raster_setup_vertices(trans, xyz, culling, S, o);
}

// the packed z-buffer word of a covered pixel from its three edge-function values, or 0xffffffff when the depth
// leaves [0, 1] (nothing is drawn)
M3T_RASTER_FN uint32_t raster_depth_word(const RasterTriangle& t, double e0, double e1, double e2, uint32_t low_bits) {
  const double z = (e1 / t.area) * t.z[0] + (e2 / t.area) * t.z[1] + (e0 / t.area) * t.z[2];
  if (!(z >= 0.0 && z <= 1.0)) return 0xffffffffu;
  const uint32_t d16 = (uint32_t)floor(z * 65535.0 + 0.46);  // see the oracle / gl_model.py
  return (d16 << 16) | low_bits;
}

// e / area for the three barycentric quotients of a pixel WITHOUT a division: with y = RN(1 / b) the sequence
// q0 = RN(a y), r = a - b q0 (exact in an fma), q = RN(q0 + r y) returns the correctly rounded quotient RN(a / b)
// (Markstein 1990; the one exception, a divisor whose 53-bit significand is all ones, cannot occur: the areas are
// integers below 2^53).  A f64 division costs ~400 cycles on the device and a covered pixel needs three of them with
// the same divisor; this costs three fma-class operations each.  tests/cpp/raster_check.cpp compares 5 x 10^7 random
// quotients and every covered pixel of its triangles with the plain division.
M3T_RASTER_FN double raster_quotient(double a, double b, double rcp_b) {
  const double q0 = a * rcp_b;
  const double r = fma(-q0, b, a);
  return fma(r, rcp_b, q0);
}
M3T_RASTER_FN uint32_t raster_depth_word_fast(const RasterTriangle& t, double e0, double e1, double e2, uint32_t low_bits) {
  const double z = raster_quotient(e1, t.area, t.rcp_area) * t.z[0] + raster_quotient(e2, t.area, t.rcp_area) * t.z[1] +
                   raster_quotient(e0, t.area, t.rcp_area) * t.z[2];
  if (!(z >= 0.0 && z <= 1.0)) return 0xffffffffu;
  const uint32_t d16 = (uint32_t)floor(z * 65535.0 + 0.46);
  return (d16 << 16) | low_bits;
}

// The definition (the oracle's form): coverage of pixel (px, py) by the three edge functions in f64, top-left rule
// with y down.  sink(px, py, word) receives covered pixels.
template <typename Sink>
M3T_RASTER_FN void raster_pixel(const RasterTriangle& t, int px, int py, uint32_t low_bits, Sink&& sink) {
  const double cx = (double)px * 256.0 + 128.0, cy = (double)py * 256.0 + 128.0;
  double e[3];
  bool inside = true;
  for (int k = 0; k < 3; ++k) {
    const int k1 = (k + 1) % 3;
    const double ex = t.ax[k1] - t.ax[k], ey = t.ay[k1] - t.ay[k];
    e[k] = ex * (cy - t.ay[k]) - ey * (cx - t.ax[k]);
    const bool owns = ey < 0.0 || (ey == 0.0 && ex > 0.0);
    inside = inside && (e[k] > 0.0 || (e[k] == 0.0 && owns));
  }
  if (!inside) return;
  const uint32_t word = raster_depth_word(t, e[0], e[1], e[2], low_bits);
  if (word != 0xffffffffu) sink(px, py, word);
}

// The row scan: pixels xa .. xb of row py.  The edge functions are evaluated once, at xa, with the expression of
// raster_pixel, and carried along the row by their exact step; the covered pixels of a row form one span (each edge
// admits a half-line of x), so the scan stops at the first uncovered pixel behind a covered one.
template <typename Sink>
M3T_RASTER_FN void raster_row(const RasterTriangle& t, int py, int xa, int xb, uint32_t low_bits, Sink&& sink) {
  const double cx = (double)xa * 256.0 + 128.0, cy = (double)py * 256.0 + 128.0;
  double e[3], step[3];
  bool owns[3];
  for (int k = 0; k < 3; ++k) {
    const int k1 = (k + 1) % 3;
    const double ex = t.ax[k1] - t.ax[k], ey = t.ay[k1] - t.ay[k];
    e[k] = ex * (cy - t.ay[k]) - ey * (cx - t.ax[k]);
    step[k] = -256.0 * ey;
    owns[k] = ey < 0.0 || (ey == 0.0 && ex > 0.0);
  }
  bool entered = false;
  for (int px = xa; px <= xb; ++px) {
    const bool inside = (e[0] > 0.0 || (e[0] == 0.0 && owns[0])) && (e[1] > 0.0 || (e[1] == 0.0 && owns[1])) &&
                        (e[2] > 0.0 || (e[2] == 0.0 && owns[2]));
    if (inside) {
      entered = true;
      const uint32_t word = raster_depth_word_fast(t, e[0], e[1], e[2], low_bits);
      if (word != 0xffffffffu) sink(px, py, word);
    } else if (entered) {
      return;
    }
    e[0] += step[0];
    e[1] += step[1];
    e[2] += step[2];
  }
}
