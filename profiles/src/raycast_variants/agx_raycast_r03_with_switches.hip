// Ray-cast depth / range / segmentation / point-cloud sensors for gfx950.
//
// Replaces the Warp kernels of aerial_gym/sensors/warp/warp_kernels/{warp_camera_kernels,
// warp_lidar_kernels}.py and wp.mesh_query_ray.  Design (CDNA4-first, not a Warp port):
//
//   * one workgroup per (env, sensor); the env's BVH nodes (64 B each, both child boxes inline)
//     and triangles (36 B each) are read through wave-uniform addresses, i.e. one L2 request
//     (scalar load) per visited node for the whole packet; the env's tree (127 KB) stays in the
//     XCD's L2 for the frame, so HBM sees each scene once per frame.  (An LDS-staged variant of
//     the same traversal is kept behind AGX_RAY_USE_LDS; it is slower here, see below.)
//   * a wavefront (64 lanes) owns an 8x8 pixel tile and walks the tree as ONE packet:
//     every node fetch is a wave-uniform (scalar) load, lanes vote with __ballot on which
//     children to visit and in which order (majority near-first), and the packet's
//     traversal stack lives in ONE VGPR spread over the 64 lanes (entry k in lane k,
//     pushed with a lane-select v_cndmask, popped with v_readlane) -- no per-lane stacks, no scratch,
//     no divergent control flow;
//   * pixels are written [env, cam, y, x] with x fastest inside each 8-wide tile row
//     (the reference's (x, y) thread order gives stride-W stores);
//   * exactness: ray/triangle uses Warp's watertight Woop test incl. its fmaf-compensated
//     edge functions and fp64 fallback; closest hit keeps the smallest (t, face) pair, so
//     the result is independent of traversal order and bit-identical to a brute-force
//     loop over all triangles (DESIGN.md "closest-hit semantics").
//
// No MFMA: traversal is branchy gather work; the kernel is VALU-issue bound (4300 VALU per 64-ray packet,
// profiles/r01_sq_counters_navigation.json), with HBM traffic ~ scene + image bytes.
#include "agx_common.h"
#include "agx_device_math.h"
#include "agx_nav_parts.h"

namespace agx {

// Measured on MI355X (profiles/r01_raycast_variants.txt, config 3, 8192 envs): traversing straight from
// L2 with wave-uniform (scalar) node loads at full occupancy beats staging the tree in LDS, because
// the 127 KB LDS footprint caps a CU at one workgroup:   LDS/512 thr 4.74 ms, LDS/1024 thr 3.39 ms,
// L2/512 thr 2.62 ms, L2/256 thr 2.54 ms per env step.  The LDS path is kept for A/B runs.
#ifndef AGX_RAY_THREADS
#define AGX_RAY_THREADS 256
#endif
#ifndef AGX_RAY_TRI_VARIANT
#define AGX_RAY_TRI_VARIANT 1  // branch structure of the triangle test: 0 = an early return per condition (the reference's shape), 1 = one early-out (shipped: -3 %), 2 = none
#endif
#ifndef AGX_RAY_HOIST_UPID
#define AGX_RAY_HOIST_UPID 1  // one copy of the traversal loop per packet (axis, orientation) instead of a switch per triangle (-2 %)
#endif
#ifndef AGX_RAY_ADDR32
#define AGX_RAY_ADDR32 1  // 32-bit offsets for node / triangle fetches: scalar loads with a register offset (-2.5 %)
#endif
#ifndef AGX_RAY_NOWANT
#define AGX_RAY_NOWANT 1  // lanes that missed the leaf's box run the triangle test too (no exec branch around it); only the update is masked (-2 %)
#endif
#ifndef AGX_RAY_PAIRLOAD
#define AGX_RAY_PAIRLOAD 1  // both triangles of a two-triangle leaf fetched before the first test (-1 %)
#endif
#ifndef AGX_RAY_PREFETCH_LEAF
#define AGX_RAY_PREFETCH_LEAF 0  // experiment: the triangles of a left leaf child requested before the node's box tests
#endif
#ifndef AGX_RAY_NOACTIVE
#define AGX_RAY_NOACTIVE 1  // retired lanes carry best = -inf instead of being masked in every slab test (-1 % camera)
#endif
#ifndef AGX_RAY_FLAT
#define AGX_RAY_FLAT 1  // one comparison per slab test, |.|-min3 for the zero-edge test, hit update as selects: no exec regions (-7 %)
#endif
#ifndef AGX_RAY_VOTEMASK
#define AGX_RAY_VOTEMASK 1  // votes of conjunctions as mask arithmetic on the votes of their terms (-2 %)
#endif
#ifndef AGX_RAY_BOX_OCTANT
#define AGX_RAY_BOX_OCTANT 0  // experiment (needs AGX_RAY_HOIST_UPID): octant-uniform packets pick near / far planes on the scalar unit
#endif
#ifndef AGX_RAY_BOX_AXIS
#define AGX_RAY_BOX_AXIS 1  // (needs AGX_RAY_HOIST_UPID) no min / max along the packet's dominant axis in the slab test (-2 %)
#endif
#ifndef AGX_RAY_WIDE
#define AGX_RAY_WIDE 0  // experiment: 4-wide nodes (profiles/wide_probe.py)
#endif
#ifndef AGX_RAY_USE_LDS
#define AGX_RAY_USE_LDS 0
#endif
constexpr int kRayThreads = AGX_RAY_THREADS;  // waves per workgroup = kRayThreads / 64
// pixel tile of a 64-ray packet.  Measured (profiles/r02_raycast_variants.txt, bit-exact either way): the pinhole camera
// wants the compact 8 x 8 tile (64 x 48 depth frame 2.65 ms; 16 x 4: 2.98 ms -- the wider frustum visits more nodes),
// the 32 x 512 LiDAR the 16 x 4 one (5.57 -> 5.29 ms: its rows are azimuth sweeps of 0.7 degrees per ray, 4 rows of
// elevation 2.9 degrees apart, so 16 x 4 is the more compact bundle there; also 64-byte row segments for the stores).
#ifndef AGX_RAY_TILE_W_CAMERA
#define AGX_RAY_TILE_W_CAMERA 8
#endif
#ifndef AGX_RAY_TILE_W_LIDAR
#define AGX_RAY_TILE_W_LIDAR 16
#endif
template <bool LIDAR>
struct Tile {
  static constexpr int W = LIDAR ? AGX_RAY_TILE_W_LIDAR : AGX_RAY_TILE_W_CAMERA, H = 64 / W;
  static_assert(W * H == 64 && (W & (W - 1)) == 0, "tile must hold one wave");
};
constexpr int kStackDepth = 64;
constexpr float kNoHitRay = 1000.0f;  // warp_camera_kernels.py:3
constexpr int kNoHitSeg = -2;         // warp_camera_kernels.py:4

// warp quat.h quat_rotate
AGX_DEV V3 wp_quat_rotate(Q4 q, V3 x) {
  float c = 2.0f * q.w * q.w - 1.0f;
  float d = 2.0f * (q.x * x.x + q.y * x.y + q.z * x.z);
  return V3{x.x * c + q.x * d + (q.y * x.z - q.z * x.y) * q.w * 2.0f, x.y * c + q.y * d + (q.z * x.x - q.x * x.z) * q.w * 2.0f,
            x.z * c + q.z * d + (q.x * x.y - q.y * x.x) * q.w * 2.0f};
}
// warp vec.h normalize
AGX_DEV V3 wp_normalize(V3 a) {
  float l = sqrtf(dot(a, a));
  if (l > 0.0f) return V3{a.x / l, a.y / l, a.z / l};
  return V3{0.0f, 0.0f, 0.0f};
}
// warp intersect.h diff_product
AGX_DEV float diff_product(float a, float b, float c, float d) {
  float cd = c * d;
  float diff = fmaf(a, b, -cd);
  float error = fmaf(-c, d, cd);
  return diff + error;
}

#ifdef AGX_RAY_STATS  // experimental builds only (profiles/raystats.py): traversal counters
__device__ unsigned long long g_ray_stats[8];
#define AGX_STAT(i, v) if ((threadIdx.x & 63) == 0) atomicAdd(&g_ray_stats[i], (unsigned long long)(v))
#else
#define AGX_STAT(i, v)
#endif

struct Ray {
  V3 op, d;   // op: the origin's components in the order (kx, ky, kz) the triangle test uses them
  V3 rcp, orcp;  // 1 / d clamped to +-1e30, and o * rcp (slab test only; the triangle test uses d)
  int kz;     // dominant axis
  bool swap;  // d[kz] < 0 : kx/ky swapped
  float Sx, Sy, Sz;
  float best;
  int face;
  bool active;
};

AGX_DEV float pick(V3 v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : v.z); }

AGX_DEV void ray_setup(Ray &r, V3 o, V3 d, float max_t, bool active) {
  r.d = d;
  const float kRcpMax = 1.0e30f;
  r.rcp = V3{fminf(fmaxf(1.0f / d.x, -kRcpMax), kRcpMax), fminf(fmaxf(1.0f / d.y, -kRcpMax), kRcpMax),
             fminf(fmaxf(1.0f / d.z, -kRcpMax), kRcpMax)};
  r.orcp = V3{o.x * r.rcp.x, o.y * r.rcp.y, o.z * r.rcp.z};
  float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
  int kz = (ax > ay) ? ((ax > az) ? 0 : 2) : ((ay > az) ? 1 : 2);
  int kx = kz == 2 ? 0 : kz + 1;
  int ky = kx == 2 ? 0 : kx + 1;
  float dz = pick(d, kz);
  r.swap = dz < 0.0f;
  if (r.swap) { int t = kx; kx = ky; ky = t; }
  r.kz = kz;
  r.op = V3{pick(o, kx), pick(o, ky), pick(o, kz)};
  r.Sx = pick(d, kx) / dz;
  r.Sy = pick(d, ky) / dz;
  r.Sz = 1.0f / dz;
  r.best = max_t;
  r.face = -1;
  r.active = active;
}

// warp intersect.h intersect_ray_tri_woop (t only).
// The vertices are wave-uniform (scalar loads); which of their components plays x / y / z depends on the ray's dominant
// axis.  `ukz` >= 0: every active ray of the packet has the same dominant axis and orientation (`ukz`, `uswap`: the usual
// case for an 8 x 8 pixel tile or a 16 x 4 LiDAR bundle) -- the component choice is then made ONCE on the scalar unit
// instead of with two v_cndmask per component and lane (18 of the ~95 vector instructions of a triangle test): the leaf
// test switches on the packet's (axis, orientation) into one of six instances with the components fixed at compile time.
// Mixed packet: per-lane choice.  Same operands, same operations either way.
// UPID: 2 * kz + swap when known at compile time (the caller switches on the packet's wave-uniform value), -1 = per lane
template <int UPID>
AGX_DEV bool ray_tri(const Ray &r, V3 a, V3 b, V3 c, float &t_out) {
  float Akx, Aky, Akz, Bkx, Bky, Bkz, Ckx, Cky, Ckz;
  if (UPID >= 0) {
    constexpr int kz = UPID >> 1;
    constexpr int kx0 = kz == 2 ? 0 : kz + 1;
    constexpr int ky0 = kx0 == 2 ? 0 : kx0 + 1;
    constexpr int kx = (UPID & 1) ? ky0 : kx0, ky = (UPID & 1) ? kx0 : ky0;
    Akx = pick(a, kx) - r.op.x; Aky = pick(a, ky) - r.op.y; Akz = pick(a, kz) - r.op.z;
    Bkx = pick(b, kx) - r.op.x; Bky = pick(b, ky) - r.op.y; Bkz = pick(b, kz) - r.op.z;
    Ckx = pick(c, kx) - r.op.x; Cky = pick(c, ky) - r.op.y; Ckz = pick(c, kz) - r.op.z;
  } else {
    int kz = r.kz;
    int kx = kz == 2 ? 0 : kz + 1;
    int ky = kx == 2 ? 0 : kx + 1;
    if (r.swap) { int t = kx; kx = ky; ky = t; }
    Akx = pick(a, kx) - r.op.x; Aky = pick(a, ky) - r.op.y; Akz = pick(a, kz) - r.op.z;
    Bkx = pick(b, kx) - r.op.x; Bky = pick(b, ky) - r.op.y; Bkz = pick(b, kz) - r.op.z;
    Ckx = pick(c, kx) - r.op.x; Cky = pick(c, ky) - r.op.y; Ckz = pick(c, kz) - r.op.z;
  }
  float Ax = Akx - r.Sx * Akz, Ay = Aky - r.Sy * Akz;
  float Bx = Bkx - r.Sx * Bkz, By = Bky - r.Sy * Bkz;
  float Cx = Ckx - r.Sx * Ckz, Cy = Cky - r.Sy * Ckz;
  float U = diff_product(Cx, By, Cy, Bx);
  float V = diff_product(Ax, Cy, Ay, Cx);
  float W = diff_product(Bx, Ay, By, Ax);
#if AGX_RAY_FLAT
  if (fminf(fminf(fabsf(U), fabsf(V)), fabsf(W)) == 0.0f) {  // any of the three exactly 0 (one v_min3 with |.| modifiers, one compare)
#else
  if (U == 0.0f || V == 0.0f || W == 0.0f) {
#endif
    double CxBy = (double)Cx * (double)By, CyBx = (double)Cy * (double)Bx;
    U = (float)(CxBy - CyBx);
    double AxCy = (double)Ax * (double)Cy, AyCx = (double)Ay * (double)Cx;
    V = (float)(AxCy - AyCx);
    double BxAy = (double)Bx * (double)Ay, ByAx = (double)By * (double)Ax;
    W = (float)(BxAy - ByAx);
  }
#if AGX_RAY_TRI_VARIANT == 0
  if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return false;
  float det = U + V + W;
  if (det == 0.0f) return false;
  float Az = r.Sz * Akz, Bz = r.Sz * Bkz, Cz = r.Sz * Ckz;
  float T = U * Az + V * Bz + W * Cz;
  uint32_t ds = __float_as_uint(det) & 0x80000000u;
  if (__uint_as_float(__float_as_uint(T) ^ ds) < 0.0f) return false;
  float rcp = 1.0f / det;
  t_out = T * rcp;
  return true;
#else
  // Fewer exec-mask branches around the same arithmetic (profiles/r03_raycast_variants.txt).  1: one early-out (edge signs +
  // determinant), the rest unconditional; 2: no early-out at all.  A lane that is rejected computes values
  // nobody reads (1 / 0 included); an accepted lane goes through exactly the operations of variant 0.
  const bool mixed = (U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f);
  const float det = U + V + W;
  const bool ok = !mixed && det != 0.0f;
#if AGX_RAY_TRI_VARIANT == 1
  if (!ok) return false;
#endif
  const float Az = r.Sz * Akz, Bz = r.Sz * Bkz, Cz = r.Sz * Ckz;
  const float T = U * Az + V * Bz + W * Cz;
  const uint32_t ds = __float_as_uint(det) & 0x80000000u;
  const bool front = !(__uint_as_float(__float_as_uint(T) ^ ds) < 0.0f);
  const float rcp = 1.0f / det;
  t_out = T * rcp;
  return ok && front;
#endif
}

// ANY: occlusion query -- the first accepted hit retires the lane (it stops voting in ray_box)
// CUPID (AGX_RAY_HOIST_UPID builds): the packet's (axis, orientation) as a template parameter of the whole traversal instead of
// a switch per triangle; -2 = decide here
template <bool ANY, int CUPID = -2>
AGX_DEV void test_leaf(Ray &r, const float *__restrict__ tris, int f, bool want, int upid) {
#if !AGX_RAY_NOWANT
  if (!want) return;
#endif
#if AGX_RAY_ADDR32
  const float *t = reinterpret_cast<const float *>(reinterpret_cast<const char *>(tris) + (uint32_t)f * 36u);
#else
  const float *t = tris + (size_t)f * 9;
#endif
  float th = 0.0f;
  const V3 a = V3{t[0], t[1], t[2]}, b = V3{t[3], t[4], t[5]}, c = V3{t[6], t[7], t[8]};
  bool hit;
  if (CUPID != -2) {
    hit = ray_tri<CUPID>(r, a, b, c, th);
  } else
  switch (upid) {  // wave-uniform
    case 0: hit = ray_tri<0>(r, a, b, c, th); break;
    case 1: hit = ray_tri<1>(r, a, b, c, th); break;
    case 2: hit = ray_tri<2>(r, a, b, c, th); break;
    case 3: hit = ray_tri<3>(r, a, b, c, th); break;
    case 4: hit = ray_tri<4>(r, a, b, c, th); break;
    case 5: hit = ray_tri<5>(r, a, b, c, th); break;
    default: hit = ray_tri<-1>(r, a, b, c, th); break;
  }
  if (AGX_RAY_NOWANT ? (hit && want) : hit) {
    if (ANY) {
      if (th >= 0.0f && th < r.best) {
        r.face = f;
        r.active = false;
        if (AGX_RAY_NOACTIVE) r.best = -INFINITY;
      }
    } else if (th >= 0.0f && (th < r.best || (th == r.best && r.face >= 0 && f < r.face))) {
      r.best = th;
      r.face = f;
    }
  }
}

// closest hit: smaller t wins, on an exact tie the smaller face index (DESIGN.md "closest-hit semantics"); any-hit: the first
// accepted hit retires the lane.  AGX_RAY_FLAT: the condition as mask arithmetic and the update as selects -- no exec regions.
template <bool ANY>
AGX_DEV void accept_hit(Ray &r, bool hit, float th, int f) {
#if AGX_RAY_FLAT
  if (ANY) {
    const bool acc = hit & (th >= 0.0f) & (th < r.best);
    r.face = acc ? f : r.face;
    r.active = acc ? false : r.active;
    if (AGX_RAY_NOACTIVE) r.best = acc ? -INFINITY : r.best;
  } else {
    const bool acc = hit & (th >= 0.0f) & ((th < r.best) | ((th == r.best) & (r.face >= 0) & (f < r.face)));
    r.best = acc ? th : r.best;
    r.face = acc ? f : r.face;
  }
#else
  if (ANY) {
    if (hit && th >= 0.0f && th < r.best) { r.face = f; r.active = false; if (AGX_RAY_NOACTIVE) r.best = -INFINITY; }
  } else if (hit && th >= 0.0f && (th < r.best || (th == r.best && r.face >= 0 && f < r.face))) {
    r.best = th;
    r.face = f;
  }
#endif
}

struct TriPair {
  V3 a1, b1, c1, a2, b2, c2;
};
AGX_DEV TriPair load_tri_pair(const float *__restrict__ tris, int f1, int f2) {
  const float *t1 = reinterpret_cast<const float *>(reinterpret_cast<const char *>(tris) + (uint32_t)f1 * 36u);
  const float *t2 = reinterpret_cast<const float *>(reinterpret_cast<const char *>(tris) + (uint32_t)(f2 >= 0 ? f2 : f1) * 36u);
  return TriPair{V3{t1[0], t1[1], t1[2]}, V3{t1[3], t1[4], t1[5]}, V3{t1[6], t1[7], t1[8]},
                 V3{t2[0], t2[1], t2[2]}, V3{t2[3], t2[4], t2[5]}, V3{t2[6], t2[7], t2[8]}};
}
template <bool ANY, int CUPID>
AGX_DEV void test_tri_pair(Ray &r, const TriPair &P, int f1, int f2, bool want) {
  float th = 0.0f;
  bool hit = ray_tri<CUPID>(r, P.a1, P.b1, P.c1, th) && want;
  accept_hit<ANY>(r, hit, th, f1);
  if (f2 >= 0) {
    th = 0.0f;
    hit = ray_tri<CUPID>(r, P.a2, P.b2, P.c2, th) && want;
    accept_hit<ANY>(r, hit, th, f2);
  }
}
// A leaf and the second triangle of a two-triangle leaf (f2 < 0: none).  AGX_RAY_PAIRLOAD: both triangles are fetched before the
// first is tested (one exposed scalar-load latency per leaf instead of two); the tests and updates stay in order.
template <bool ANY, int CUPID>
AGX_DEV void test_leaf_pair(Ray &r, const float *__restrict__ tris, int f1, int f2, bool want, int upid) {
#if AGX_RAY_PAIRLOAD
  if (CUPID != -2) {
    const TriPair P = load_tri_pair(tris, f1, f2);
    test_tri_pair<ANY, CUPID>(r, P, f1, f2, want);
    return;
  }
#endif
  test_leaf<ANY, CUPID>(r, tris, f1, want, upid);
  if (f2 >= 0) test_leaf<ANY, CUPID>(r, tris, f2, want, upid);
}

// Conservative slab test, one fma per plane: t = b * rcp - o * rcp, with rcp CLAMPED to +-1e30 in ray_setup.
// Why this never culls a box that holds a hit (boxes are grown by kBoxEps = 1e-3 at build time; |coords| < 10 km):
//   * |d_c| > 1e-30: the products are finite (|b| |rcp| < 1e34); the only new error vs (b - o) * rcp is the
//     rounding of o * rcp, <= 6e-8 |o| in space units, far inside the 1e-3 growth.
//   * |d_c| <= 1e-30 (incl. exactly 0, where 1/d = +-inf would give inf - inf = NaN or a wrong-signed inf --
//     measured: occlusion rays with d_z == 0 lost their occluder, test_stereo_occlusion_ray_with_zero_direction_component):
//     a hit at t <= max_t moves < 1e-26 along c, so the origin lies inside the un-grown slab up to that, i.e.
//     >= 1e-3 inside the grown one; b * 1e30 - o * 1e30 then has the right sign and magnitude >= 1e27 > any max_t.
// CUPID >= 0 (a packet whose rays share the dominant axis kz = CUPID >> 1 and its sign, CUPID & 1: d[kz] < 0): along THAT axis
// the order of the two plane distances is known -- fma is monotone in its first argument, lo <= hi, and the sign of rcp is the
// packet's -- so the min / max pair of that axis is dropped; the values that remain are the ones min / max would have picked.
template <int CUPID = -1>
AGX_DEV bool ray_box(const Ray &r, float lx, float ly, float lz, float hx, float hy, float hz, float &tnear) {
  constexpr int kzc = CUPID >= 0 ? (CUPID >> 1) : -1;
  constexpr bool neg = CUPID >= 0 && (CUPID & 1);
  float t0 = fmaf(lx, r.rcp.x, -r.orcp.x), t1 = fmaf(hx, r.rcp.x, -r.orcp.x);
  float tmin = kzc == 0 ? (neg ? t1 : t0) : fminf(t0, t1), tmax = kzc == 0 ? (neg ? t0 : t1) : fmaxf(t0, t1);
  t0 = fmaf(ly, r.rcp.y, -r.orcp.y); t1 = fmaf(hy, r.rcp.y, -r.orcp.y);
  tmin = fmaxf(tmin, kzc == 1 ? (neg ? t1 : t0) : fminf(t0, t1)); tmax = fminf(tmax, kzc == 1 ? (neg ? t0 : t1) : fmaxf(t0, t1));
  t0 = fmaf(lz, r.rcp.z, -r.orcp.z); t1 = fmaf(hz, r.rcp.z, -r.orcp.z);
  tmin = fmaxf(tmin, kzc == 2 ? (neg ? t1 : t0) : fminf(t0, t1)); tmax = fminf(tmax, kzc == 2 ? (neg ? t0 : t1) : fmaxf(t0, t1));
  tmax *= 1.0000004f;
  tnear = tmin;
#if AGX_RAY_NOACTIVE && AGX_RAY_FLAT
  // tmax >= 0 and tmax >= tmin and tmin <= best, with best >= 0 for a live lane and -inf for a retired one (traverse), is the
  // ONE comparison max(tmin, 0) <= min(tmax, best): a v_cmp that writes the packet's mask directly (the ballot of a conjunction
  // goes through a VGPR)
  return fmaxf(tmin, 0.0f) <= fminf(tmax, r.best);
#elif AGX_RAY_NOACTIVE  // a retired lane carries best = -inf (traverse): the last comparison fails for it
  return (tmax >= 0.0f) && (tmax >= tmin) && (tmin <= r.best);
#else
  return r.active && (tmax >= 0.0f) && (tmax >= tmin) && (tmin <= r.best);
#endif
}

// The same test for a packet whose rays all point into the same OCTANT (`oct`, wave-uniform: bit a set = every active ray has
// 1 / d_a < 0): which plane of a slab is the near one is then the packet's choice, made on the scalar unit (the planes are
// wave-uniform), and no min / max is left per lane.  Same plane distances, same comparisons as ray_box.
AGX_DEV bool ray_box_oct(const Ray &r, float lx, float ly, float lz, float hx, float hy, float hz, int oct, float &tnear) {
  const bool nx = (oct & 1) != 0, ny = (oct & 2) != 0, nz = (oct & 4) != 0;
  const float ax = nx ? hx : lx, bx = nx ? lx : hx, ay = ny ? hy : ly, by = ny ? ly : hy, az = nz ? hz : lz, bz = nz ? lz : hz;
  float tmin = fmaf(ax, r.rcp.x, -r.orcp.x), tmax = fmaf(bx, r.rcp.x, -r.orcp.x);
  tmin = fmaxf(tmin, fmaf(ay, r.rcp.y, -r.orcp.y)); tmax = fminf(tmax, fmaf(by, r.rcp.y, -r.orcp.y));
  tmin = fmaxf(tmin, fmaf(az, r.rcp.z, -r.orcp.z)); tmax = fminf(tmax, fmaf(bz, r.rcp.z, -r.orcp.z));
  tmax *= 1.0000004f;
  tnear = tmin;
  return r.active && (tmax >= 0.0f) && (tmax >= tmin) && (tmin <= r.best);
}

AGX_DEV unsigned long long vote(bool p) { return __builtin_amdgcn_ballot_w64(p); }  // the mask itself, no round trip through a VGPR

#if AGX_RAY_WIDE
// EXPERIMENT (profiles/wide_probe.py; built only with -DAGX_RAY_WIDE=1): 4-wide nodes.  A record is 32 floats: box k = 0..3 at
// [6 k .. 6 k + 5] (lo xyz, hi xyz), its reference at [24 + k] (>= 0: record index, < 0: leaf ~triangle, INT_MIN: no entry),
// the second triangle of a two-triangle leaf at [28 + k].  Entries 0, 1 come from the binary node's left child, 2, 3 from its
// right child (a leaf child occupies the first entry of its pair), so the visiting order can follow the binary tree's votes.
constexpr int kNoEntry = (int)0x80000000;
template <bool ANY = false>
AGX_DEV void traverse(Ray &r, const float *__restrict__ nodes, const float *__restrict__ tris, int nt) {
  int upid = -1;
  {
    const int pid = r.kz * 2 + (r.swap ? 1 : 0);
    const unsigned long long act = vote(r.active);
    if (act) {
      const int p0 = __builtin_amdgcn_readlane(pid, __ffsll((long long)act) - 1);
      if (vote(r.active && pid != p0) == 0ull) upid = p0;
    }
  }
  if (nt == 1) {
    test_leaf<ANY>(r, tris, 0, r.active, upid);
    return;
  }
  int sp = 0, node = 0, stack = 0;
  const int lane = threadIdx.x & 63;
  AGX_STAT(0, 1);
  while (true) {
    AGX_STAT(1, 1);
    const float4 *nd = reinterpret_cast<const float4 *>(nodes + (size_t)node * 32);
    const float4 w0 = nd[0], w1 = nd[1], w2 = nd[2], w3 = nd[3], w4 = nd[4], w5 = nd[5], w6 = nd[6], w7 = nd[7];
    const int r0 = __float_as_int(w6.x), r1 = __float_as_int(w6.y), r2 = __float_as_int(w6.z), r3 = __float_as_int(w6.w);
    const int s0 = __float_as_int(w7.x), s1 = __float_as_int(w7.y), s2 = __float_as_int(w7.z), s3 = __float_as_int(w7.w);
    float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f, t3 = 0.0f;
    bool h0 = false, h1 = false, h2 = false, h3 = false;
    if (r0 != kNoEntry) h0 = ray_box(r, w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, t0);
    if (r1 != kNoEntry) h1 = ray_box(r, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w, t1);
    if (r2 != kNoEntry) h2 = ray_box(r, w3.x, w3.y, w3.z, w3.w, w4.x, w4.y, t2);
    if (r3 != kNoEntry) h3 = ray_box(r, w4.z, w4.w, w5.x, w5.y, w5.z, w5.w, t3);
    unsigned long long m0 = vote(h0), m1 = vote(h1), m2 = vote(h2), m3 = vote(h3);
    // leaves first, in entry order; whatever comes after a leaf test votes again with the shortened rays
    bool shortened = false;
#define AGX_WIDE_LEAF(R, S, H, T, M)                                   \
  if (R < 0 && R != kNoEntry) {                                         \
    if (shortened && M) {                                               \
      H = H && (T <= r.best) && r.active;                               \
      M = vote(H);                                                      \
    }                                                                   \
    if (M) {                                                            \
      test_leaf<ANY>(r, tris, ~R, H, upid);                             \
      if (S >= 0) test_leaf<ANY>(r, tris, S, H, upid);                  \
      AGX_STAT(2, S >= 0 ? 2 : 1);                                      \
      shortened = true;                                                 \
    }                                                                   \
    M = 0;                                                              \
    H = false;                                                          \
  }
    AGX_WIDE_LEAF(r0, s0, h0, t0, m0)
    AGX_WIDE_LEAF(r1, s1, h1, t1, m1)
    AGX_WIDE_LEAF(r2, s2, h2, t2, m2)
    AGX_WIDE_LEAF(r3, s3, h3, t3, m3)
#undef AGX_WIDE_LEAF
    if (shortened) {
      if (m0) { h0 = h0 && (t0 <= r.best) && r.active; m0 = vote(h0); }
      if (m1) { h1 = h1 && (t1 <= r.best) && r.active; m1 = vote(h1); }
      if (m2) { h2 = h2 && (t2 <= r.best) && r.active; m2 = vote(h2); }
      if (m3) { h3 = h3 && (t3 <= r.best) && r.active; m3 = vote(h3); }
    }
    // order: within each pair and between the pairs, the nearer first by majority among the lanes that hit both
    bool a_swap = false, b_swap = false;  // entry 1 before entry 0 / entry 3 before entry 2
    if (m0 && m1) {
      const unsigned long long both = vote(h0 && h1), first = vote(h0 && h1 && t0 <= t1);
      a_swap = both ? (2 * __popcll(first) < __popcll(both)) : (__popcll(m0) < __popcll(m1));
    } else {
      a_swap = m0 == 0ull;
    }
    if (m2 && m3) {
      const unsigned long long both = vote(h2 && h3), first = vote(h2 && h3 && t2 <= t3);
      b_swap = both ? (2 * __popcll(first) < __popcll(both)) : (__popcll(m2) < __popcll(m3));
    } else {
      b_swap = m2 == 0ull;
    }
    const unsigned long long ma = m0 | m1, mb = m2 | m3;
    bool b_first = false;
    if (ma && mb) {
      const bool ha = h0 || h1, hb = h2 || h3;
      const float ta = fminf(h0 ? t0 : 3.0e38f, h1 ? t1 : 3.0e38f), tb = fminf(h2 ? t2 : 3.0e38f, h3 ? t3 : 3.0e38f);
      const unsigned long long both = vote(ha && hb), first = vote(ha && hb && ta <= tb);
      b_first = both ? (2 * __popcll(first) < __popcll(both)) : (__popcll(ma) < __popcll(mb));
    } else {
      b_first = ma == 0ull;
    }
    // the four entries near -> far: (reference, hit by anybody)
    const int a0r = a_swap ? r1 : r0, a1r = a_swap ? r0 : r1, b0r = b_swap ? r3 : r2, b1r = b_swap ? r2 : r3;
    const bool a0h = (a_swap ? m1 : m0) != 0ull, a1h = (a_swap ? m0 : m1) != 0ull, b0h = (b_swap ? m3 : m2) != 0ull, b1h = (b_swap ? m2 : m3) != 0ull;
    const int e0 = b_first ? b0r : a0r, e1 = b_first ? b1r : a1r, e2 = b_first ? a0r : b0r, e3 = b_first ? a1r : b1r;
    const bool g0 = b_first ? b0h : a0h, g1 = b_first ? b1h : a1h, g2 = b_first ? a0h : b0h, g3 = b_first ? a1h : b1h;
    int next = -1;
    // push far -> near everything behind the first hit entry
    const bool before3 = g0 || g1 || g2, before2 = g0 || g1, before1 = g0;
    if (g3) {
      if (before3) { stack = (lane == (sp & (kStackDepth - 1))) ? e3 : stack; ++sp; } else next = e3;
    }
    if (g2) {
      if (before2) { stack = (lane == (sp & (kStackDepth - 1))) ? e2 : stack; ++sp; } else next = e2;
    }
    if (g1) {
      if (before1) { stack = (lane == (sp & (kStackDepth - 1))) ? e1 : stack; ++sp; } else next = e1;
    }
    if (g0) next = e0;
    if (next < 0) {
      if (sp == 0) break;
      --sp;
      next = __builtin_amdgcn_readlane(stack, sp & (kStackDepth - 1));
    }
    node = __builtin_amdgcn_readfirstlane(next);
  }
}
#else
// Packet traversal: the whole wave follows one path.  The stack is a single VGPR whose lane k
// holds entry k (depth <= 64 > 30 Morton bits + log2(T) tie bits of the LBVH).
template <bool ANY, int CUPID, bool OCT = false>
AGX_DEV void traverse_impl(Ray &r, const float *__restrict__ nodes, const float *__restrict__ tris, int nt, int upid, int oct = 0) {
  if (nt == 1) {
    test_leaf<ANY, CUPID>(r, tris, 0, r.active, upid);
    return;
  }
  int sp = 0;
  int node = 0;
  int stack = 0;
  const int lane = threadIdx.x & 63;
  AGX_STAT(0, 1);  // packets
  while (true) {
    AGX_STAT(1, 1);  // node visits
#if AGX_RAY_ADDR32
    // 32-bit byte offsets from the env's node block (< 2^31 bytes): a scalar load with a register offset, no 64-bit address arithmetic
    const float4 *nd = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(nodes) + ((uint32_t)node << 6));
#else
    const float4 *nd = reinterpret_cast<const float4 *>(nodes + (size_t)node * 16);
#endif
    float4 n0 = nd[0], n1 = nd[1], n2 = nd[2], n3 = nd[3];
    int cl = __float_as_int(n0.w), cr = __float_as_int(n1.w);
    const int cl2 = __float_as_int(n2.w), cr2 = __float_as_int(n3.w);  // second triangle of a two-triangle leaf, or -1
#if AGX_RAY_PREFETCH_LEAF
    TriPair PL;  // a leaf on the left: its triangles are requested before the box tests run
    constexpr bool kPrefetch = CUPID != -2;
    if (kPrefetch && cl < 0) PL = load_tri_pair(tris, ~cl, cl2);
#endif
    float tl, tr;
    bool hl, hr;
    if (OCT) {
      hl = ray_box_oct(r, n0.x, n0.y, n0.z, n1.x, n1.y, n1.z, oct, tl);
      hr = ray_box_oct(r, n2.x, n2.y, n2.z, n3.x, n3.y, n3.z, oct, tr);
    } else {
      hl = ray_box<AGX_RAY_BOX_AXIS ? CUPID : -1>(r, n0.x, n0.y, n0.z, n1.x, n1.y, n1.z, tl);
      hr = ray_box<AGX_RAY_BOX_AXIS ? CUPID : -1>(r, n2.x, n2.y, n2.z, n3.x, n3.y, n3.z, tr);
    }
    unsigned long long ml = vote(hl), mr = vote(hr);
    if (cl < 0) {
      if (ml) {
#if AGX_RAY_PREFETCH_LEAF
        if (kPrefetch) test_tri_pair<ANY, CUPID>(r, PL, ~cl, cl2, hl);
        else
#endif
        test_leaf_pair<ANY, CUPID>(r, tris, ~cl, cl2, hl, upid);
        AGX_STAT(2, cl2 >= 0 ? 2 : 1); AGX_STAT(3, __popcll(ml));
      }
      ml = 0;
    }
    if (cr < 0) {
      if (cl < 0 && mr) {  // the left leaf may just have shortened the rays: vote again with the new `best`
#if AGX_RAY_VOTEMASK && AGX_RAY_NOACTIVE
        mr &= vote(tr <= r.best);  // (the ballot of ONE comparison is the comparison's own mask; a conjunction goes through a VGPR)
        hr = hr && (tr <= r.best);
#else
        hr = hr && (tr <= r.best) && (AGX_RAY_NOACTIVE || r.active);
        mr = vote(hr);
#endif
      }
      if (mr) {
        test_leaf_pair<ANY, CUPID>(r, tris, ~cr, cr2, hr, upid);
        AGX_STAT(2, cr2 >= 0 ? 2 : 1); AGX_STAT(3, __popcll(mr));
      }
      mr = 0;
    }
    int next = -1;
    if (ml && mr) {
      // majority vote on which child is nearer among lanes that hit both
#if AGX_RAY_VOTEMASK
      const unsigned long long both = ml & mr;
      const unsigned long long lfirst = both & vote(tl <= tr);
      const bool left_first = both ? (2 * (int)__popcll(lfirst) >= (int)__popcll(both)) : ((int)__popcll(ml) >= (int)__popcll(mr));
#else
      unsigned long long both = vote(hl && hr);
      unsigned long long lfirst = vote(hl && hr && tl <= tr);
      bool left_first = both ? (2 * __popcll(lfirst) >= __popcll(both)) : (__popcll(ml) >= __popcll(mr));
#endif
      next = left_first ? cl : cr;
      int far = left_first ? cr : cl;
      stack = (lane == (sp & (kStackDepth - 1))) ? far : stack;  // push: entry sp lives in lane sp
      ++sp;
    } else if (ml) {
      next = cl;
    } else if (mr) {
      next = cr;
    }
    if (next < 0) {
      if (sp == 0) break;
      --sp;
      next = __builtin_amdgcn_readlane(stack, sp & (kStackDepth - 1));
    }
    node = __builtin_amdgcn_readfirstlane(next);
  }
}


template <bool ANY = false>
AGX_DEV void traverse(Ray &r, const float *__restrict__ nodes, const float *__restrict__ tris, int nt) {
  // do all active rays of the packet share the dominant axis and its orientation?
  int upid = -1;  // 2 * kz + swap, or -1 (mixed)
  {
    const int pid = r.kz * 2 + (r.swap ? 1 : 0);
    const unsigned long long act = vote(r.active);
    if (act) {
      const int p0 = __builtin_amdgcn_readlane(pid, __ffsll((long long)act) - 1);
      if (vote(r.active && pid != p0) == 0ull) upid = p0;
    }
  }
#if AGX_RAY_NOACTIVE
  if (!r.active) r.best = -INFINITY;  // (a lane outside the image: nobody reads its `best`)
#endif
#if AGX_RAY_HOIST_UPID && AGX_RAY_BOX_OCTANT
  // ... and the same octant?  (sign bits of the clamped reciprocals: 1 / +0 counts as positive, 1 / -0 as negative, like ray_box)
  int oct = -1;
  if (upid >= 0) {
    const int sb = (int)(__float_as_uint(r.rcp.x) >> 31) | ((int)(__float_as_uint(r.rcp.y) >> 31) << 1) | ((int)(__float_as_uint(r.rcp.z) >> 31) << 2);
    const unsigned long long act = vote(r.active);
    const int s0 = __builtin_amdgcn_readlane(sb, __ffsll((long long)act) - 1);
    if (vote(r.active && sb != s0) == 0ull) oct = s0;
  }
  if (oct >= 0) {
    switch (upid) {
      case 0: traverse_impl<ANY, 0, true>(r, nodes, tris, nt, upid, oct); return;
      case 1: traverse_impl<ANY, 1, true>(r, nodes, tris, nt, upid, oct); return;
      case 2: traverse_impl<ANY, 2, true>(r, nodes, tris, nt, upid, oct); return;
      case 3: traverse_impl<ANY, 3, true>(r, nodes, tris, nt, upid, oct); return;
      case 4: traverse_impl<ANY, 4, true>(r, nodes, tris, nt, upid, oct); return;
      default: traverse_impl<ANY, 5, true>(r, nodes, tris, nt, upid, oct); return;
    }
  }
#endif
#if AGX_RAY_HOIST_UPID
  switch (upid) {  // wave-uniform, once per packet: seven copies of the loop
    case 0: traverse_impl<ANY, 0>(r, nodes, tris, nt, upid); break;
    case 1: traverse_impl<ANY, 1>(r, nodes, tris, nt, upid); break;
    case 2: traverse_impl<ANY, 2>(r, nodes, tris, nt, upid); break;
    case 3: traverse_impl<ANY, 3>(r, nodes, tris, nt, upid); break;
    case 4: traverse_impl<ANY, 4>(r, nodes, tris, nt, upid); break;
    case 5: traverse_impl<ANY, 5>(r, nodes, tris, nt, upid); break;
    default: traverse_impl<ANY, -1>(r, nodes, tris, nt, upid); break;
  }
#else
  traverse_impl<ANY, -2>(r, nodes, tris, nt, upid);
#endif
}
#endif  // AGX_RAY_WIDE

struct CamArgs {
  int n, ns, width, height;
  float k00, k02, k11, k12;
  float far_plane;
  int c_x, c_y, mode;
  float baseline;  // stereo partner at cam_pos + R(q) (-baseline, 0, 0)
};

struct LidarArgs {
  int n, ns, width, height;
  float far_plane;
  int mode;
};

// WarpSensor.apply_range_limits + normalize_observation (warp_sensor.py:216-247) on the scalar pixel before it is stored:
// the operations of k_sensor_postprocess without noise, in its order, so the image needs no second pass
struct RangeEpilogue {
  int enabled;
  float min_range, max_range, far_oor, near_oor;
  int normalize;
};
AGX_DEV float apply_range_limits(const RangeEpilogue &RL, float p) {
  if (p > RL.max_range) p = RL.far_oor;
  if (p < RL.min_range) p = RL.near_oor;
  if (RL.normalize) p = p / RL.max_range;
  return p;
}

// VARIANT: what happens after the closest hit is known
//   RAY_BASIC   depth / range / point cloud (+ segmentation)             warp_camera_kernels.py:176-282, warp_lidar_kernels.py
//   RAY_NORMAL  geometric normal + face index                           warp_camera_kernels.py:70-121, warp_lidar_kernels.py:90-126
//   RAY_STEREO  BASIC, valid only where the stereo partner sees the point too (second, any-hit ray)
//                                                                       warp_stereo_camera_kernels.py:13-299
enum { RAY_BASIC = 0, RAY_NORMAL = 1, RAY_STEREO = 2 };
constexpr float kInvalidPixel = -1.0f;  // warp_stereo_camera_kernels.py:3

// Register budget: every instance is compiled for 8 waves per SIMD (64 VGPRs; BASIC / NORMAL fit without a VGPR spill)
// except STEREO, which carries two rays' worth of state over the second traversal next to the six compile-time instances
// of the triangle test: 6 waves (80 VGPRs), no spill.  Measured round 2 (profiles/r02_raycast_variants.txt).
#ifndef AGX_RAY_WAVES
#define AGX_RAY_WAVES 8
#endif
#ifndef AGX_RAY_STEREO_WAVES
#define AGX_RAY_STEREO_WAVES 6
#endif
template <bool LIDAR, bool USE_LDS, int VARIANT>
__global__ void __launch_bounds__(kRayThreads, VARIANT == 2 ? AGX_RAY_STEREO_WAVES : AGX_RAY_WAVES) k_raycast(CamArgs CA, LidarArgs LA, RangeEpilogue RL, const float *__restrict__ ray_vectors,
                                                          const float *__restrict__ sensor_pos,
                                                          const float *__restrict__ sensor_quat,
                                                          const float *__restrict__ tri_world,
                                                          const int32_t *__restrict__ tri_seg,
                                                          const float *__restrict__ nodes_g, int nt,
                                                          float *__restrict__ pixels, int32_t *__restrict__ seg) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int env = blockIdx.x, s = blockIdx.y;
  const int ns = LIDAR ? LA.ns : CA.ns;
  const int width = LIDAR ? LA.width : CA.width, height = LIDAR ? LA.height : CA.height;
  const int mode = LIDAR ? LA.mode : CA.mode;
  const float far_plane = LIDAR ? LA.far_plane : CA.far_plane;
  const int n_nodes = nt - 1;
  const float *g_nodes = nodes_g + (size_t)env * n_nodes * (AGX_RAY_WIDE ? 32 : 16);
  const float *g_tris = tri_world + (size_t)env * nt * 9;
  const float *nodes = g_nodes, *tris = g_tris;
  if (USE_LDS) {
    float *l_nodes = reinterpret_cast<float *>(smem);
    float *l_tris = l_nodes + (size_t)n_nodes * 16;
    // 16-byte coalesced staging (node block is 64 B aligned, triangle block 4-float padded)
    const float4 *src = reinterpret_cast<const float4 *>(g_nodes);
    float4 *dst = reinterpret_cast<float4 *>(l_nodes);
    for (int i = threadIdx.x; i < n_nodes * 4; i += kRayThreads) dst[i] = src[i];
    for (int i = threadIdx.x; i < nt * 9; i += kRayThreads) l_tris[i] = g_tris[i];
    __syncthreads();
    nodes = l_nodes;
    tris = l_tris;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t sidx = (size_t)env * ns + s;
  const V3 ro = V3{sensor_pos[sidx * 3], sensor_pos[sidx * 3 + 1], sensor_pos[sidx * 3 + 2]};
  const Q4 sq = Q4{sensor_quat[sidx * 4], sensor_quat[sidx * 4 + 1], sensor_quat[sidx * 4 + 2], sensor_quat[sidx * 4 + 3]};
  V3 rdp = V3{0, 0, 0};
  if (!LIDAR) {
    V3 uvp = V3{CA.k00 * (float)CA.c_x + CA.k02, CA.k11 * (float)CA.c_y + CA.k12, 1.0f};
    if (mode >= AGX_RAY_POINTCLOUD) uvp = wp_normalize(uvp);
    rdp = wp_normalize(wp_quat_rotate(sq, uvp));
  }
  V3 partner = ro;
  if (VARIANT == RAY_STEREO) partner = ro + wp_quat_rotate(sq, V3{-CA.baseline, 0.0f, 0.0f});
  constexpr int kTileW = Tile<LIDAR>::W, kTileH = Tile<LIDAR>::H;
  const int tiles_x = (width + kTileW - 1) / kTileW, tiles_y = (height + kTileH - 1) / kTileH;
  // gridDim.z workgroups share the tiles of one (env, sensor): small batches still fill the GPU
  for (int tile = blockIdx.z * (kRayThreads / 64) + wave; tile < tiles_x * tiles_y; tile += gridDim.z * (kRayThreads / 64)) {
    const int x = (tile % tiles_x) * kTileW + (lane % kTileW), y = (tile / tiles_x) * kTileH + (lane / kTileW);
    const bool active = x < width && y < height;
    V3 local = V3{0.0f, 0.0f, 1.0f};
    if (active) {
      if (LIDAR) {
        const float *rv = ray_vectors + ((size_t)y * width + x) * 3;
        local = wp_normalize(V3{rv[0], rv[1], rv[2]});
      } else {
        // wp.transform_vector(K_inv, (x, y, 1)) (warp_camera_kernels.py:199-200)
        local = V3{CA.k00 * (float)x + CA.k02, CA.k11 * (float)y + CA.k12, 1.0f};
        if (mode >= AGX_RAY_POINTCLOUD) local = wp_normalize(local);
      }
    }
    V3 rd = wp_normalize(wp_quat_rotate(sq, local));
    float mult = 1.0f;
    if (!LIDAR && mode == AGX_RAY_DEPTH) mult = dot(rd, rdp);
    float max_t = (!LIDAR && mode <= AGX_RAY_DEPTH) ? far_plane / mult : far_plane;
    Ray r;
    ray_setup(r, ro, rd, max_t, active);
    traverse(r, nodes, tris, nt);
    const size_t px = ((sidx * height) + y) * width + x;
    if (VARIANT == RAY_NORMAL) {
      // miss: zero normal, face -1 (the reference's `n`, `f` stay at their initial values)
      V3 nrm = V3{0.0f, 0.0f, 0.0f};
      if (active && r.face >= 0) {
        const float *t = tris + (size_t)r.face * 9;
        const V3 a = V3{t[0], t[1], t[2]}, b = V3{t[3], t[4], t[5]}, c = V3{t[6], t[7], t[8]};
        nrm = wp_normalize(cross_plain(b - a, c - a));  // warp intersect.h out_normal, mesh.h normalize(min_normal)
      }
      if (mode == AGX_RAY_NORMAL) {
        if (LIDAR) {
          nrm = wp_normalize(wp_quat_rotate(Q4{-sq.x, -sq.y, -sq.z, sq.w}, nrm));  // quat_inverse(lidar_quaternion)
        } else {
          nrm = V3{dot(nrm, rdp), dot(nrm, cross_plain(rdp, V3{0.0f, 0.0f, 1.0f})), dot(nrm, cross_plain(rdp, V3{0.0f, 1.0f, 0.0f}))};
        }
      }
      if (active) {
        pixels[3 * px] = nrm.x;
        pixels[3 * px + 1] = nrm.y;
        pixels[3 * px + 2] = nrm.z;
        if (seg) seg[px] = r.face;
      }
      continue;
    }
    const bool hit = r.face >= 0;
    bool visible = true;
    if (VARIANT == RAY_STEREO) {
      // ro + rd * t*0.999  |  ro + rd * far_plane / multiplier  (point clouds: no multiplier)
      V3 endpoint = hit ? ro + (rd * r.best) * 0.999f : (mode <= AGX_RAY_DEPTH ? ro + (rd * far_plane) / mult : ro + rd * far_plane);
      V3 back = partner - endpoint;
      Ray r2;
      ray_setup(r2, endpoint, wp_normalize(back), sqrtf(dot(back, back)), active);
      traverse<true>(r2, nodes, tris, nt);
      visible = r2.face < 0;
    }
    if (active) {
      float dist = VARIANT == RAY_STEREO ? kInvalidPixel : kNoHitRay;
      int sv = kNoHitSeg;
      if (visible) {
        if (VARIANT == RAY_STEREO) dist = kNoHitRay;
        if (hit) {
          dist = (!LIDAR && mode <= AGX_RAY_DEPTH) ? mult * r.best : r.best;
          if (seg) sv = tri_seg[(size_t)env * nt + r.face];
        }
      }
      if (mode <= AGX_RAY_DEPTH) {
        pixels[px] = RL.enabled ? apply_range_limits(RL, dist) : dist;
      } else if (mode == AGX_RAY_POINTCLOUD_WORLD) {
        pixels[3 * px] = ro.x + dist * rd.x;
        pixels[3 * px + 1] = ro.y + dist * rd.y;
        pixels[3 * px + 2] = ro.z + dist * rd.z;
      } else {
        pixels[3 * px] = dist * local.x;
        pixels[3 * px + 1] = dist * local.y;
        pixels[3 * px + 2] = dist * local.z;
      }
      if (seg) seg[px] = sv;
    }
  }
}

// WarpSensor.update pose composition (warp_sensor.py:177-187)
__global__ void __launch_bounds__(256) k_sensor_pose(AgxEnvBuffers B, int n, int ns, const float *__restrict__ local_pos,
                                                      const float *__restrict__ local_quat, Q4 frame_quat,
                                                      float *__restrict__ pos, float *__restrict__ quat) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * ns) return;
  sensor_pose_env(B, n, idx / ns, idx, local_pos, local_quat, frame_quat, pos, quat);
}

// WarpSensor.apply_noise / apply_range_limits / normalize_observation (warp_sensor.py:202-247)
__global__ void __launch_bounds__(256) k_sensor_postprocess(size_t count, float *__restrict__ pixels,
                                                             const float *__restrict__ z_normal,
                                                             const float *__restrict__ u_dropout, float std_a, float std_b,
                                                             float std_c, float mean_offset, float dropout_prob,
                                                             float min_range, float max_range, float far_oor, float near_oor,
                                                             int normalize) {
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < count; k += (size_t)gridDim.x * blockDim.x) {
    float p = pixels[k];
    if (z_normal) {
      float sd = std_a * (p * p) + std_b * p + std_c;
      p = (p - mean_offset) + sd * z_normal[k];
      if (u_dropout && u_dropout[k] < dropout_prob) p = near_oor;
    }
    if (p > max_range) p = far_oor;
    if (p < min_range) p = near_oor;
    if (normalize) p = p / max_range;
    pixels[k] = p;
  }
}

// point-cloud branch of the same three functions: noise / dropout per component, range limits on
// the point's norm (all three components replaced); `limits` = 0 for world-frame clouds
__global__ void __launch_bounds__(256) k_sensor_postprocess_points(size_t count, float *__restrict__ pixels,
                                                                    const float *__restrict__ z_normal,
                                                                    const float *__restrict__ u_dropout, float std_a,
                                                                    float std_b, float std_c, float mean_offset,
                                                                    float dropout_prob, float min_range, float max_range,
                                                                    float far_oor, float near_oor, int limits, int normalize) {
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < count; k += (size_t)gridDim.x * blockDim.x) {
    float v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float p = pixels[3 * k + c];
      if (z_normal) {
        float sd = std_a * (p * p) + std_b * p + std_c;
        p = (p - mean_offset) + sd * z_normal[3 * k + c];
        if (u_dropout && u_dropout[3 * k + c] < dropout_prob) p = near_oor;
      }
      v[c] = p;
    }
    if (limits) {
      float nrm = norm(V3{v[0], v[1], v[2]});  // Tensor.norm(dim = 4)
      if (nrm > max_range) v[0] = v[1] = v[2] = far_oor;
      nrm = norm(V3{v[0], v[1], v[2]});
      if (nrm < min_range) v[0] = v[1] = v[2] = near_oor;
      if (normalize) {
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = v[c] / max_range;
      }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) pixels[3 * k + c] = v[c];
  }
}

// NavigationTask.post_image_reward_addition (navigation_task.py:351-357): per-env min of 10*img
// with negative pixels replaced by 10.  One wave per env.
__global__ void __launch_bounds__(256) k_image_min(int n, int ppe, const float *__restrict__ pixels, float *__restrict__ out) {
  const int env = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (env >= n) return;
  const int lane = threadIdx.x & 63;
  float m = INFINITY;
  for (int k = lane; k < ppe; k += 64) {
    float v = 10.0f * pixels[(size_t)env * ppe + k];
    if (v < 0.0f) v = 10.0f;
    m = fminf(m, v);
  }
  for (int off = 32; off > 0; off >>= 1) m = fminf(m, __shfl_xor(m, off));
  if (lane == 0) out[env] = m;
}

static size_t ray_lds_bytes(int nt) {
  return (size_t)(nt - 1) * 64 + (size_t)nt * 36 + 16;
}

template <bool LIDAR, int VARIANT>
static int launch_raycast(const CamArgs &CA, const LidarArgs &LA, const AgxRangeLimits *limits, const float *ray_vectors, const float *pos, const float *quat,
                          const float *tri_world, const int32_t *tri_seg, const float *nodes, int nt, float *pixels,
                          int32_t *seg, void *stream) {
  const int n = LIDAR ? LA.n : CA.n, ns = LIDAR ? LA.ns : CA.ns;
  RangeEpilogue RL{};
  if (limits) {
    AGX_REQUIRE((LIDAR ? LA.mode : CA.mode) <= AGX_RAY_DEPTH, "range limits are fused for scalar images only (modes RANGE / DEPTH)");
    RL = RangeEpilogue{1, limits->min_range, limits->max_range, limits->far_oor, limits->near_oor, limits->normalize};
  }
  size_t lds = ray_lds_bytes(nt);
#if AGX_RAY_USE_LDS
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_raycast<LIDAR, true, VARIANT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
#endif
  // 256 CUs x 32 waves = 8192 resident waves, and tails want ~2x that in the grid: split an image's 8x8 tiles over several workgroups when the
  // batch alone cannot provide them (256 envs: 199 -> see profiles/r01_small_batch.txt us per frame)
  const int width = LIDAR ? LA.width : CA.width, height = LIDAR ? LA.height : CA.height;
  constexpr int kTileW = Tile<LIDAR>::W, kTileH = Tile<LIDAR>::H;
  const int tiles = ((width + kTileW - 1) / kTileW) * ((height + kTileH - 1) / kTileH), waves_per_wg = kRayThreads / 64;
  int split = (16384 + n * ns * waves_per_wg - 1) / (n * ns * waves_per_wg);
  const int max_split = (tiles + waves_per_wg - 1) / waves_per_wg;
  split = split < 1 ? 1 : (split > max_split ? max_split : split);
  dim3 grid(n, ns, AGX_RAY_USE_LDS ? 1 : split);
#if AGX_RAY_USE_LDS  // experimental builds only: the LDS-staged kernels are not instantiated otherwise
  if (lds <= 160 * 1024) {
    hipLaunchKernelGGL((k_raycast<LIDAR, true, VARIANT>), grid, dim3(kRayThreads), lds, (hipStream_t)stream, CA, LA, RL, ray_vectors,
                       pos, quat, tri_world, tri_seg, nodes, nt, pixels, seg);
    return check_launch(LIDAR ? "agx_raycast_lidar" : "agx_raycast_camera");
  }
#endif
  (void)lds;  // default: traverse from L2 with wave-uniform loads
  hipLaunchKernelGGL((k_raycast<LIDAR, false, VARIANT>), grid, dim3(kRayThreads), 0, (hipStream_t)stream, CA, LA, RL, ray_vectors,
                     pos, quat, tri_world, tri_seg, nodes, nt, pixels, seg);
  return check_launch(LIDAR ? "agx_raycast_lidar" : "agx_raycast_camera");
}

}  // namespace agx

using namespace agx;

#ifdef AGX_RAY_STATS
extern "C" int agx_debug_ray_stats(unsigned long long *out, int reset) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ray_stats), sizeof(g_ray_stats));
  if (reset) {
    unsigned long long z[8] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ray_stats), z, sizeof(z));
  }
  return 0;
}
#endif

extern "C" int agx_sensor_pose(const AgxEnvBuffers *B, int n, int ns, const float *local_pos, const float *local_quat,
                               const float *frame_quat, float *pos, float *quat, void *stream) {
  AGX_REQUIRE(B && B->state && n > 0 && ns > 0, "bad arguments");
  AGX_REQUIRE(local_pos && local_quat && frame_quat && pos && quat, "null buffer");
  Q4 fq = Q4{frame_quat[0], frame_quat[1], frame_quat[2], frame_quat[3]};  // HOST pointer: 4 config scalars
  hipLaunchKernelGGL(k_sensor_pose, dim3(blocks_for(n * ns, 256)), dim3(256), 0, (hipStream_t)stream, *B, n, ns, local_pos,
                     local_quat, fq, pos, quat);
  return check_launch("agx_sensor_pose");
}

extern "C" int agx_raycast_camera(int n, int ns, int width, int height, const float *kinv, float far_plane, int c_x, int c_y,
                                  int mode, const float *cam_pos, const float *cam_quat, const float *tri_world,
                                  const int32_t *tri_seg, const float *nodes, int nt, float *pixels, int32_t *seg,
                                  const AgxRangeLimits *limits, void *stream) {
  AGX_REQUIRE(n > 0 && ns > 0 && width > 0 && height > 0 && nt >= 1, "bad sizes");
  AGX_REQUIRE(mode >= AGX_RAY_RANGE && mode <= AGX_RAY_NORMAL_WORLD, "bad mode %d", mode);
  AGX_REQUIRE(kinv && cam_pos && cam_quat && tri_world && pixels && (nt == 1 || nodes), "null buffer");
  AGX_REQUIRE(!seg || tri_seg || mode >= AGX_RAY_NORMAL, "segmentation output needs tri_seg");
  CamArgs CA{n, ns, width, height, kinv[0], kinv[1], kinv[2], kinv[3], far_plane, c_x, c_y, mode, 0.0f};  // kinv: HOST pointer
  LidarArgs LA{};
  if (mode >= AGX_RAY_NORMAL)
    return launch_raycast<false, RAY_NORMAL>(CA, LA, limits, nullptr, cam_pos, cam_quat, tri_world, tri_seg, nodes, nt, pixels, seg, stream);
  return launch_raycast<false, RAY_BASIC>(CA, LA, limits, nullptr, cam_pos, cam_quat, tri_world, tri_seg, nodes, nt, pixels, seg, stream);
}

extern "C" int agx_raycast_stereo_camera(int n, int ns, int width, int height, const float *kinv, float far_plane, float baseline,
                                         int c_x, int c_y, int mode, const float *cam_pos, const float *cam_quat,
                                         const float *tri_world, const int32_t *tri_seg, const float *nodes, int nt,
                                         float *pixels, int32_t *seg, const AgxRangeLimits *limits, void *stream) {
  AGX_REQUIRE(n > 0 && ns > 0 && width > 0 && height > 0 && nt >= 1, "bad sizes");
  AGX_REQUIRE(mode >= AGX_RAY_RANGE && mode <= AGX_RAY_POINTCLOUD_WORLD, "bad mode %d", mode);
  AGX_REQUIRE(kinv && cam_pos && cam_quat && tri_world && pixels && (nt == 1 || nodes), "null buffer");
  AGX_REQUIRE(!seg || tri_seg, "segmentation output needs tri_seg");
  CamArgs CA{n, ns, width, height, kinv[0], kinv[1], kinv[2], kinv[3], far_plane, c_x, c_y, mode, baseline};
  LidarArgs LA{};
  return launch_raycast<false, RAY_STEREO>(CA, LA, limits, nullptr, cam_pos, cam_quat, tri_world, tri_seg, nodes, nt, pixels, seg, stream);
}

extern "C" int agx_raycast_lidar(int n, int ns, int width, int height, const float *ray_vectors, float far_plane, int mode,
                                 const float *pos, const float *quat, const float *tri_world, const int32_t *tri_seg,
                                 const float *nodes, int nt, float *pixels, int32_t *seg, const AgxRangeLimits *limits,
                                 void *stream) {
  AGX_REQUIRE(n > 0 && ns > 0 && width > 0 && height > 0 && nt >= 1, "bad sizes");
  AGX_REQUIRE(mode == AGX_RAY_RANGE || (mode >= AGX_RAY_POINTCLOUD && mode <= AGX_RAY_NORMAL_WORLD), "bad mode %d", mode);
  AGX_REQUIRE(ray_vectors && pos && quat && tri_world && pixels && (nt == 1 || nodes), "null buffer");
  AGX_REQUIRE(!seg || tri_seg || mode >= AGX_RAY_NORMAL, "segmentation output needs tri_seg");
  CamArgs CA{};
  LidarArgs LA{n, ns, width, height, far_plane, mode};
  if (mode >= AGX_RAY_NORMAL)
    return launch_raycast<true, RAY_NORMAL>(CA, LA, limits, ray_vectors, pos, quat, tri_world, tri_seg, nodes, nt, pixels, seg, stream);
  return launch_raycast<true, RAY_BASIC>(CA, LA, limits, ray_vectors, pos, quat, tri_world, tri_seg, nodes, nt, pixels, seg, stream);
}

extern "C" int agx_sensor_postprocess(size_t count, float *pixels, const float *z_normal, const float *u_dropout, float std_a,
                                      float std_b, float std_c, float mean_offset, float dropout_prob, float min_range,
                                      float max_range, float far_oor, float near_oor, int normalize, void *stream) {
  AGX_REQUIRE(pixels && count > 0, "null buffer");
  int blocks = (int)((count + 255) / 256);
  if (blocks > 256 * 8) blocks = 256 * 8;
  hipLaunchKernelGGL(k_sensor_postprocess, dim3(blocks), dim3(256), 0, (hipStream_t)stream, count, pixels, z_normal, u_dropout,
                     std_a, std_b, std_c, mean_offset, dropout_prob, min_range, max_range, far_oor, near_oor, normalize);
  return check_launch("agx_sensor_postprocess");
}

extern "C" int agx_sensor_postprocess_points(size_t count, float *pixels, const float *z_normal, const float *u_dropout,
                                             float std_a, float std_b, float std_c, float mean_offset, float dropout_prob,
                                             float min_range, float max_range, float far_oor, float near_oor, int limits,
                                             int normalize, void *stream) {
  AGX_REQUIRE(pixels && count > 0, "null buffer");
  int blocks = (int)((count + 255) / 256);
  if (blocks > 256 * 8) blocks = 256 * 8;
  hipLaunchKernelGGL(k_sensor_postprocess_points, dim3(blocks), dim3(256), 0, (hipStream_t)stream, count, pixels, z_normal,
                     u_dropout, std_a, std_b, std_c, mean_offset, dropout_prob, min_range, max_range, far_oor, near_oor, limits,
                     normalize);
  return check_launch("agx_sensor_postprocess_points");
}

extern "C" int agx_image_min(int n, int ppe, const float *pixels, float *min_pixel, void *stream) {
  AGX_REQUIRE(n > 0 && ppe > 0 && pixels && min_pixel, "bad arguments");
  hipLaunchKernelGGL(k_image_min, dim3(blocks_for(n, 4)), dim3(256), 0, (hipStream_t)stream, n, ppe, pixels, min_pixel);
  return check_launch("agx_image_min");
}
