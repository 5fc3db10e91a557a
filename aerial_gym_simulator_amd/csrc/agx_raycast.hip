// Ray-cast depth / range / segmentation / point-cloud sensors for gfx950.
//
// Replaces the Warp kernels of aerial_gym/sensors/warp/warp_kernels/{warp_camera_kernels,
// warp_lidar_kernels}.py and wp.mesh_query_ray.  Design (CDNA4-first, not a Warp port):
//
//   * one workgroup per (env, sensor); the env's BVH nodes (64 B each, both child boxes inline)
//     and triangles (36 B each) are read through wave-uniform addresses, i.e. one L2 request
//     (scalar load) per visited node for the whole packet; the env's tree (127 KB) stays in the
//     XCD's L2 for the frame, so HBM sees each scene once per frame.  (Staging the whole tree in LDS caps a CU at one
//     workgroup and measured slower: profiles/r01_raycast_variants.txt; that variant and every other rejected one live in
//     profiles/src/raycast_variants/, not here.)
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

// Build-time shape parameters (measured: profiles/r01..r03_raycast_variants.txt).  The algorithmic variants that were tried and
// rejected -- LDS-staged trees, 4-wide nodes, octant-uniform slab tests, leaf prefetch, per-condition early returns, the
// unspecialised traversal loop -- are archived with their switches in profiles/src/raycast_variants/agx_raycast_r03_with_switches.hip;
// this file holds the shipped path only.
#ifndef AGX_RAY_THREADS
#define AGX_RAY_THREADS 256
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
// axis.  UPID >= 0: every active ray of the packet has the same dominant axis and orientation (2 * kz + swap: the usual case
// for an 8 x 8 pixel tile or a 16 x 4 LiDAR bundle) -- the component choice is then made at COMPILE time (the whole traversal
// loop is instantiated per UPID, `traverse`) instead of with two v_cndmask per component and lane (18 of the ~95 vector
// instructions of a triangle test).  UPID = -1: mixed packet, per-lane choice.  Same operands, same operations either way.
// Branch structure: ONE early-out (mixed edge signs | zero determinant); T, the sign test and 1 / det are unconditional -- a lane
// that is rejected computes values nobody reads (1 / 0 included), an accepted lane goes through exactly the operations of the
// reference's test (which returns early per condition: three nested exec regions more per triangle, +3 %).
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
  if (fminf(fminf(fabsf(U), fabsf(V)), fabsf(W)) == 0.0f) {  // any of the three exactly 0 (one v_min3 with |.| modifiers, one compare)
    double CxBy = (double)Cx * (double)By, CyBx = (double)Cy * (double)Bx;
    U = (float)(CxBy - CyBx);
    double AxCy = (double)Ax * (double)Cy, AyCx = (double)Ay * (double)Cx;
    V = (float)(AxCy - AyCx);
    double BxAy = (double)Bx * (double)Ay, ByAx = (double)By * (double)Ax;
    W = (float)(BxAy - ByAx);
  }
  const bool mixed = (U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f);
  const float det = U + V + W;
  const bool ok = !mixed && det != 0.0f;
  if (!ok) return false;
  const float Az = r.Sz * Akz, Bz = r.Sz * Bkz, Cz = r.Sz * Ckz;
  const float T = U * Az + V * Bz + W * Cz;
  const uint32_t ds = __float_as_uint(det) & 0x80000000u;
  const bool front = !(__uint_as_float(__float_as_uint(T) ^ ds) < 0.0f);
  const float rcp = 1.0f / det;
  t_out = T * rcp;
  return ok && front;
}

// closest hit: smaller t wins, on an exact tie the smaller face index (DESIGN.md "closest-hit semantics"); ANY (occlusion query):
// the first accepted hit retires the lane -- it carries best = -inf from then on, so every later slab test fails for it.
// The condition is mask arithmetic and the update two selects: no exec regions.
template <bool ANY>
AGX_DEV void accept_hit(Ray &r, bool hit, float th, int f) {
  if (ANY) {
    const bool acc = hit & (th >= 0.0f) & (th < r.best);
    r.face = acc ? f : r.face;
    r.active = acc ? false : r.active;
    r.best = acc ? -INFINITY : r.best;
  } else {
    const bool acc = hit & (th >= 0.0f) & ((th < r.best) | ((th == r.best) & (r.face >= 0) & (f < r.face)));
    r.best = acc ? th : r.best;
    r.face = acc ? f : r.face;
  }
}

// A leaf: triangle f1 and, in a two-triangle leaf, f2 (< 0: none).  Both triangles are fetched (32-bit byte offsets from the
// env's block: scalar loads with a register offset) before the first is tested -- one exposed scalar-load latency per leaf
// instead of two; the tests and updates stay in order.  Lanes that missed the leaf's box (`want` false) run the tests too, only
// the update is masked: no exec branch around a triangle test.
template <bool ANY, int UPID>
AGX_DEV void test_leaf_pair(Ray &r, const float *__restrict__ tris, int f1, int f2, bool want) {
  const float *t1 = reinterpret_cast<const float *>(reinterpret_cast<const char *>(tris) + (uint32_t)f1 * 36u);
  const float *t2 = reinterpret_cast<const float *>(reinterpret_cast<const char *>(tris) + (uint32_t)(f2 >= 0 ? f2 : f1) * 36u);
  const V3 a1 = V3{t1[0], t1[1], t1[2]}, b1 = V3{t1[3], t1[4], t1[5]}, c1 = V3{t1[6], t1[7], t1[8]};
  const V3 a2 = V3{t2[0], t2[1], t2[2]}, b2 = V3{t2[3], t2[4], t2[5]}, c2 = V3{t2[6], t2[7], t2[8]};
  float th = 0.0f;
  bool hit = ray_tri<UPID>(r, a1, b1, c1, th) && want;
  accept_hit<ANY>(r, hit, th, f1);
  if (f2 >= 0) {
    th = 0.0f;
    hit = ray_tri<UPID>(r, a2, b2, c2, th) && want;
    accept_hit<ANY>(r, hit, th, f2);
  }
}

AGX_DEV unsigned long long vote(bool p) { return __builtin_amdgcn_ballot_w64(p); }  // the mask itself, no round trip through a VGPR

// The ray's origin in world order, from the copy ray_setup keeps in the triangle test's axis order
template <int UPID>
AGX_DEV V3 ray_origin(const Ray &r) {
  // (values first, then selects on VALUES: a select between loads of r.op's members becomes a load through a selected address,
  //  and the Ray then lives in scratch instead of registers)
  const float p0 = r.op.x, p1 = r.op.y, p2 = r.op.z;
  int kx, ky;
  if (UPID >= 0) {
    constexpr int kz = UPID >> 1;
    constexpr int kx0 = kz == 2 ? 0 : kz + 1;
    constexpr int ky0 = kx0 == 2 ? 0 : kx0 + 1;
    kx = (UPID & 1) ? ky0 : kx0;
    ky = (UPID & 1) ? kx0 : ky0;
  } else {
    const int kz = r.kz;
    kx = kz == 2 ? 0 : kz + 1;
    ky = kx == 2 ? 0 : kx + 1;
    if (r.swap) { const int t = kx; kx = ky; ky = t; }
  }
  V3 o;
  o.x = kx == 0 ? p0 : (ky == 0 ? p1 : p2);
  o.y = kx == 1 ? p0 : (ky == 1 ? p1 : p2);
  o.z = kx == 2 ? p0 : (ky == 2 ? p1 : p2);
  return o;
}

// OBJECT NODE (AGX_BVH_BOX_OBJECTS): the tree ends at a box whose frame the 64-byte record holds (axes nx, ny, nz, half extents,
// centre, first triangle).  The result stays what it is defined to be -- the smallest (t, face) over the triangles the EXACT test
// accepts -- because only triangles that test cannot accept are skipped:
//   * a = R^T (o - c), b = R^T d: the ray in the box's frame; plane +-k is met at t = (+-h_k - a_k) / b_k, evaluated as
//     fma(+-h_k, 1 / b_k, -a_k / b_k).  Every quantity carries an error bound far below the tolerance it is compared with:
//     |delta t| <= 1e-6 s / |b_k| with s = |a|_1 + |h|_1 + 1 (the roundings of the dot products, of 1 / b and of the two products
//     that cancel in the fma are each < 2e-7 s / |b_k|); eps_k = 1e-3 + 1e-6 s |1 / b_k| is used for plane k, in metres and in units of t alike
//     (|d| = 1).  1 / b_k is clamped to +-1e30: a ray parallel to a plane has eps ~ 1e24 and keeps every face (the exact test decides).
//   * a box is convex: the FIRST crossing of a ray with it is at t_first = the slab entry max_k(near_k) if that is >= 0 (origin
//     outside), else the slab exit min_k(far_k) (origin inside); the exact test can accept a triangle of face F only where the ray
//     really crosses F (up to ~1e-6 m), i.e. at plane-crossing time t_F inside the face's rectangle.  A crossing later than t_first + eps
//     loses the (t, face) minimum to the first one, unless the exact test rejects the first one -- which it can only where the first
//     crossing sits within rounding of the face's rim, and then another face is crossed within eps of t_first, too (the surface is
//     closed; the exact test is watertight).  So the candidates are the faces with |t_F - t_first| <= 2 eps (the slab overlap says the
//     crossing point lies inside the face's rectangle, up to eps); t_first itself must lie in [-eps, best + eps].  An origin within the tolerance of the surface keeps
//     the faces around the EXIT as well (t_alt).  Occlusion rays (ANY) ask whether SOME triangle is hit in [0, best): the first
//     crossing answers that as well.
//   * then the face's two triangles go through the same exact test as a two-triangle leaf (test_leaf_pair), masked by the lane's
//     candidate flag.
//     Faces of trimesh's box by triangle: -x (0, 2)  +x (10, 11)  -y (1, 5)  +y (7, 9)  -z (3, 8)  +z (4, 6).
template <int UPID>
AGX_DEV uint32_t box_face_candidates(const Ray &r, float4 n0, float4 n1, float4 n2, float4 n3) {
  const V3 o = ray_origin<UPID>(r);
  const V3 oc = V3{o.x - n3.x, o.y - n3.y, o.z - n3.z};
  // (this arithmetic is the accelerator's own -- conservative, never part of a result -- so it may contract: explicit fma, a third
  //  fewer instructions than the mul / add chains the library's -ffp-contract=off would make of the plain expressions)
  const float a0 = fmaf(n0.z, oc.z, fmaf(n0.y, oc.y, n0.x * oc.x)), a1 = fmaf(n1.z, oc.z, fmaf(n1.y, oc.y, n1.x * oc.x)),
              a2 = fmaf(n2.z, oc.z, fmaf(n2.y, oc.y, n2.x * oc.x));
  const float b0 = fmaf(n0.z, r.d.z, fmaf(n0.y, r.d.y, n0.x * r.d.x)), b1 = fmaf(n1.z, r.d.z, fmaf(n1.y, r.d.y, n1.x * r.d.x)),
              b2 = fmaf(n2.z, r.d.z, fmaf(n2.y, r.d.y, n2.x * r.d.x));
  const float h0 = n0.w, h1 = n1.w, h2 = n2.w;
  const float kMax = 1.0e30f;
  // (v_rcp_f32, 1 ulp: the tolerance below absorbs it; a correctly rounded 1 / b is ten instructions each)
#ifndef AGX_RAY_EXACT_RCP
  const float rb0 = fminf(fmaxf(__builtin_amdgcn_rcpf(b0), -kMax), kMax), rb1 = fminf(fmaxf(__builtin_amdgcn_rcpf(b1), -kMax), kMax),
              rb2 = fminf(fmaxf(__builtin_amdgcn_rcpf(b2), -kMax), kMax);
#else
  const float rb0 = fminf(fmaxf(1.0f / b0, -kMax), kMax), rb1 = fminf(fmaxf(1.0f / b1, -kMax), kMax), rb2 = fminf(fmaxf(1.0f / b2, -kMax), kMax);
#endif
  const float s = 1.0e-6f * (((fabsf(a0) + fabsf(a1)) + (fabsf(a2) + h0)) + ((h1 + h2) + 1.0f));
  const float e0 = fmaf(s, fabsf(rb0), 1.0e-3f), e1 = fmaf(s, fabsf(rb1), 1.0e-3f), e2 = fmaf(s, fabsf(rb2), 1.0e-3f);
  const float eps = fmaxf(fmaxf(e0, e1), e2);
  // plane-crossing times: minus / plus face of each axis, (+-h - a) / b as fma(+-h, 1 / b, -a / b)
  const float ar0 = -a0 * rb0, ar1 = -a1 * rb1, ar2 = -a2 * rb2;
  const float tm0 = fmaf(-h0, rb0, ar0), tp0 = fmaf(h0, rb0, ar0);
  const float tm1 = fmaf(-h1, rb1, ar1), tp1 = fmaf(h1, rb1, ar1);
  const float tm2 = fmaf(-h2, rb2, ar2), tp2 = fmaf(h2, rb2, ar2);
  const float t_enter = fmaxf(fmaxf(fminf(tm0, tp0), fminf(tm1, tp1)), fminf(tm2, tp2));
  const float t_exit = fminf(fminf(fmaxf(tm0, tp0), fmaxf(tm1, tp1)), fmaxf(tm2, tp2));
  const float t_first = t_enter >= -eps ? t_enter : t_exit;
  const float two_eps = 2.0f * eps;
  // an origin within the tolerance of the surface: the exact test may place the entry at t < 0 and reject it -- then the exit is the
  // first hit it accepts (stereo occlusion rays start 0.1 % of the range in front of a surface: sub-millimetre at close range)
  const float t_alt = fabsf(t_enter) <= two_eps ? t_exit : t_first;
  // the box is crossed at all (entry before exit, up to the tolerance), and its first crossing is a possible result
  const bool live = (t_enter <= t_exit + two_eps) & (t_first >= -eps) & (t_first <= r.best + eps);
  // Which faces: the one(s) whose plane is crossed AT the first crossing (|t_F - t_first| <= 2 eps: the entry face is the face whose
  // near plane gives the slab entry; at an edge or a corner two or three tie), plus those at the exit for an origin on the surface.
  // The crossing point needs no test of its own: `live` says the slabs overlap, i.e. the point at t_first is inside the box's other
  // two slabs.  The six verdicts are packed into ONE register before the first triangle test: nothing of the box-frame arithmetic
  // stays live across the exact tests (they take the kernel's whole register budget).
  uint32_t cbits = 0u;
#define AGX_BOX_FACE(BIT, TF, TREF) cbits |= (live & (fabsf((TF) - (TREF)) <= two_eps)) ? (1u << (BIT)) : 0u;
  AGX_BOX_FACE(0, tm0, t_first)
  AGX_BOX_FACE(1, tp0, t_first)
  AGX_BOX_FACE(2, tm1, t_first)
  AGX_BOX_FACE(3, tp1, t_first)
  AGX_BOX_FACE(4, tm2, t_first)
  AGX_BOX_FACE(5, tp2, t_first)
#ifndef AGX_RAY_BOX_ALT_ALWAYS
  // t_alt differs from t_first only for an origin within the tolerance of the surface (|t_enter| <= 2 eps and t_enter >= -eps): some
  // lane of the packet in that position is rare (a sensor touching an obstacle, stereo occlusion rays at close range) -- the six
  // comparisons against the exit are made only then (the same candidate sets; round 5 made them for every visit)
  if (vote(live & (t_alt != t_first)))
#endif
  {
    AGX_BOX_FACE(0, tm0, t_alt)
    AGX_BOX_FACE(1, tp0, t_alt)
    AGX_BOX_FACE(2, tm1, t_alt)
    AGX_BOX_FACE(3, tp1, t_alt)
    AGX_BOX_FACE(4, tm2, t_alt)
    AGX_BOX_FACE(5, tp2, t_alt)
  }
#undef AGX_BOX_FACE
  return cbits;  // bit k: face k (-x +x -y +y -z +z) may hold this ray's result; its triangles: nibble k of 0x4371A0 / 0x6895B2
}

// Conservative slab test, one fma per plane: t = b * rcp - o * rcp, with rcp CLAMPED to +-1e30 in ray_setup.
// Why this never culls a box that holds a hit (boxes are grown by kBoxEps = 1e-3 at build time; |coords| < 10 km):
//   * |d_c| > 1e-30: the products are finite (|b| |rcp| < 1e34); the only new error vs (b - o) * rcp is the
//     rounding of o * rcp, <= 6e-8 |o| in space units, far inside the 1e-3 growth.
//   * |d_c| <= 1e-30 (incl. exactly 0, where 1/d = +-inf would give inf - inf = NaN or a wrong-signed inf --
//     measured: occlusion rays with d_z == 0 lost their occluder, test_stereo_occlusion_ray_with_zero_direction_component):
//     a hit at t <= max_t moves < 1e-26 along c, so the origin lies inside the un-grown slab up to that, i.e.
//     >= 1e-3 inside the grown one; b * 1e30 - o * 1e30 then has the right sign and magnitude >= 1e27 > any max_t.
// UPID >= 0 (a packet whose rays share the dominant axis kz = UPID >> 1 and its sign, UPID & 1: d[kz] < 0): along THAT axis the
// order of the two plane distances is known -- fma is monotone in its first argument, lo <= hi, and the sign of rcp is the
// packet's -- so the min / max pair of that axis is dropped; the values that remain are the ones min / max would have picked.
// The verdict tmax >= 0 and tmax >= tmin and tmin <= best -- with best >= 0 for a live lane and -inf for a retired one or a lane
// outside the image -- is the ONE comparison max(tmin, 0) <= min(tmax, best): a v_cmp that writes the packet's mask directly (the
// ballot of a conjunction goes through a VGPR).
template <int UPID>
AGX_DEV bool ray_box(const Ray &r, float lx, float ly, float lz, float hx, float hy, float hz, float &tnear) {
  constexpr int kzc = UPID >= 0 ? (UPID >> 1) : -1;
  constexpr bool neg = UPID >= 0 && (UPID & 1);
  float t0 = fmaf(lx, r.rcp.x, -r.orcp.x), t1 = fmaf(hx, r.rcp.x, -r.orcp.x);
  float tmin = kzc == 0 ? (neg ? t1 : t0) : fminf(t0, t1), tmax = kzc == 0 ? (neg ? t0 : t1) : fmaxf(t0, t1);
  t0 = fmaf(ly, r.rcp.y, -r.orcp.y); t1 = fmaf(hy, r.rcp.y, -r.orcp.y);
  tmin = fmaxf(tmin, kzc == 1 ? (neg ? t1 : t0) : fminf(t0, t1)); tmax = fminf(tmax, kzc == 1 ? (neg ? t0 : t1) : fmaxf(t0, t1));
  t0 = fmaf(lz, r.rcp.z, -r.orcp.z); t1 = fmaf(hz, r.rcp.z, -r.orcp.z);
  tmin = fmaxf(tmin, kzc == 2 ? (neg ? t1 : t0) : fminf(t0, t1)); tmax = fminf(tmax, kzc == 2 ? (neg ? t0 : t1) : fmaxf(t0, t1));
  tmax *= 1.0000004f;
  tnear = tmin;
  return fmaxf(tmin, 0.0f) <= fminf(tmax, r.best);
}


// Packet traversal: the whole wave follows one path.  The stack is a single VGPR whose lane k holds entry k (depth <= 64 > 30
// Morton bits + log2(T) tie bits of the LBVH).  Votes of conjunctions are mask arithmetic on the votes of their terms (the
// ballot of ONE comparison is the comparison's own mask).
template <bool ANY, int UPID>
AGX_DEV void traverse_impl(Ray &r, const float *__restrict__ nodes, const float *__restrict__ tris) {
  int sp = 0;
  int node = 0;
  int stack = 0;
  const int lane = threadIdx.x & 63;
  while (true) {
    // 32-bit byte offsets from the env's node block (< 2^31 bytes): a scalar load with a register offset, no 64-bit address arithmetic
    const float4 *nd = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(nodes) + (((uint32_t)node & 0x3FFFFFFFu) << 6));
    float4 n0 = nd[0], n1 = nd[1], n2 = nd[2], n3 = nd[3];
    int next = -1;
    // (the ordinary node is the fall-through of the wave-uniform branch, the object node the arm behind it: one taken branch per
    //  object visit instead of one per node visit)
    if (__builtin_expect((node & AGX_BVH_OBJECT_REF) == 0, 1)) {
      int cl = __float_as_int(n0.w), cr = __float_as_int(n1.w);
      const int cl2 = __float_as_int(n2.w), cr2 = __float_as_int(n3.w);  // second triangle of a two-triangle leaf, or -1
      float tl, tr;
      bool hl = ray_box<UPID>(r, n0.x, n0.y, n0.z, n1.x, n1.y, n1.z, tl);
      bool hr = ray_box<UPID>(r, n2.x, n2.y, n2.z, n3.x, n3.y, n3.z, tr);
      unsigned long long ml = vote(hl), mr = vote(hr);
      if (cl < 0) {
        if (ml) test_leaf_pair<ANY, UPID>(r, tris, ~cl, cl2, hl);
        ml = 0;
      }
      if (cr < 0) {
        if (cl < 0 && mr) {  // the left leaf may just have shortened the rays: vote again with the new `best`
          mr &= vote(tr <= r.best);
          hr = hr && (tr <= r.best);
        }
        if (mr) test_leaf_pair<ANY, UPID>(r, tris, ~cr, cr2, hr);
        mr = 0;
      }
      if (ml && mr) {
        // majority vote on which child is nearer among lanes that hit both
        const unsigned long long both = ml & mr;
        const unsigned long long lfirst = both & vote(tl <= tr);
        const bool left_first = both ? (2 * (int)__popcll(lfirst) >= (int)__popcll(both)) : ((int)__popcll(ml) >= (int)__popcll(mr));
        next = left_first ? cl : cr;
        int far = left_first ? cr : cl;
        stack = (lane == (sp & (kStackDepth - 1))) ? far : stack;  // push: entry sp lives in lane sp
        ++sp;
      } else if (ml) {
        next = cl;
      } else if (mr) {
        next = cr;
      }
    } else {  // an OBJECT NODE: the box itself, no children
      const uint32_t cbits = box_face_candidates<UPID>(r, n0, n1, n2, n3);
      const int f0 = __float_as_int(n3.w);
      // ONE copy of the exact test, looped over the faces somebody wants (unrolled, the scheduler hoists all twelve triangles' scalar
      // loads in front of the first test: 108 more live scalars in a kernel that spills scalars to lanes already)
#pragma nounroll
      for (int k = 0; k < 6; ++k) {
        const bool want = (cbits & (1u << k)) != 0u;
        if (vote(want)) test_leaf_pair<ANY, UPID>(r, tris, f0 + (int)((0x4371A0u >> (4 * k)) & 15u), f0 + (int)((0x6895B2u >> (4 * k)) & 15u), want);
      }
    }
    if (next < 0) {
      if (sp == 0) break;
      --sp;
      next = __builtin_amdgcn_readlane(stack, sp & (kStackDepth - 1));
    }
    node = __builtin_amdgcn_readfirstlane(next);
  }
}

// ANY: occlusion query -- the first accepted hit retires the lane
template <bool ANY = false>
AGX_DEV void traverse(Ray &r, const float *__restrict__ nodes, const float *__restrict__ tris, int nt) {
  if (!r.active) r.best = -INFINITY;  // (a lane outside the image: nobody reads its `best`)
  if (nt == 1) {  // a one-triangle scene has no node: the generic instance of the test, once
    test_leaf_pair<ANY, -1>(r, tris, 0, -1, r.active);
    return;
  }
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
  switch (upid) {  // wave-uniform, once per packet: seven copies of the loop (-2 % vs a switch in front of every triangle test)
    case 0: traverse_impl<ANY, 0>(r, nodes, tris); break;
    case 1: traverse_impl<ANY, 1>(r, nodes, tris); break;
    case 2: traverse_impl<ANY, 2>(r, nodes, tris); break;
    case 3: traverse_impl<ANY, 3>(r, nodes, tris); break;
    case 4: traverse_impl<ANY, 4>(r, nodes, tris); break;
    case 5: traverse_impl<ANY, 5>(r, nodes, tris); break;
    default: traverse_impl<ANY, -1>(r, nodes, tris); break;
  }
}

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
template <bool LIDAR, int VARIANT>
__global__ void __launch_bounds__(kRayThreads, VARIANT == 2 ? AGX_RAY_STEREO_WAVES : AGX_RAY_WAVES) k_raycast(CamArgs CA, LidarArgs LA, RangeEpilogue RL, const float *__restrict__ ray_vectors,
                                                          const float *__restrict__ sensor_pos,
                                                          const float *__restrict__ sensor_quat,
                                                          const float *__restrict__ tri_world,
                                                          const int32_t *__restrict__ tri_seg,
                                                          const float *__restrict__ nodes_g, int nt,
                                                          float *__restrict__ pixels, int32_t *__restrict__ seg, int split) {
  // XCD-aware workgroup -> (env, sensor, part) mapping.  The dispatcher deals consecutive workgroup ids round-robin to the 8 XCDs,
  // and each XCD has its own 4 MB L2.  An (env, sensor) image is cut into `split` parts (workgroups); the parts of ONE image get
  // ids that are 8 apart and consecutive in their XCD's queue, so they run side by side on the SAME XCD: the XCD's L2 then holds
  // the trees of (resident workgroups / split) envs instead of one env per workgroup (127 KB each on configs[2]: 256 resident
  // workgroups = 32 MB of trees against 4 MB of L2 when split = 1 -- every node visit an L2 miss).
  const int ns = LIDAR ? LA.ns : CA.ns;
  const unsigned wg = blockIdx.x, xcd = wg & 7u, q = wg >> 3;
  const unsigned part = q % (unsigned)split, image = (q / (unsigned)split) * 8u + xcd;
  if (image >= (unsigned)((LIDAR ? LA.n : CA.n) * ns)) return;  // (the last group of 8 images may be short)
  const int env = (int)(image / (unsigned)ns), s = (int)(image % (unsigned)ns);
  const int width = LIDAR ? LA.width : CA.width, height = LIDAR ? LA.height : CA.height;
  const int mode = LIDAR ? LA.mode : CA.mode;
  const float far_plane = LIDAR ? LA.far_plane : CA.far_plane;
  const float *nodes = nodes_g + (size_t)env * (nt - 1) * 16;  // traversed straight from L2 through wave-uniform loads
  const float *tris = tri_world + (size_t)env * nt * 9;
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
  for (int tile = (int)part * (kRayThreads / 64) + wave; tile < tiles_x * tiles_y; tile += split * (kRayThreads / 64)) {
    // the pixel's ray in the sensor frame: wp.transform_vector(K_inv, (x, y, 1)) (warp_camera_kernels.py:199-200) / the ray table
    auto local_ray = [&](int px_x, int px_y, bool on) {
      V3 l = V3{0.0f, 0.0f, 1.0f};
      if (on) {
        if (LIDAR) {
          const float *rv = ray_vectors + ((size_t)px_y * width + px_x) * 3;
          l = wp_normalize(V3{rv[0], rv[1], rv[2]});
        } else {
          l = V3{CA.k00 * (float)px_x + CA.k02, CA.k11 * (float)px_y + CA.k12, 1.0f};
          if (mode >= AGX_RAY_POINTCLOUD) l = wp_normalize(l);
        }
      }
      return l;
    };
    Ray r;
    {
      const int x0 = (tile % tiles_x) * kTileW + (lane % kTileW), y0 = (tile / tiles_x) * kTileH + (lane / kTileW);
      const bool on = x0 < width && y0 < height;
      const V3 rd0 = wp_normalize(wp_quat_rotate(sq, local_ray(x0, y0, on)));
      const float mult0 = (!LIDAR && mode == AGX_RAY_DEPTH) ? dot(rd0, rdp) : 1.0f;
      ray_setup(r, ro, rd0, (!LIDAR && mode <= AGX_RAY_DEPTH) ? far_plane / mult0 : far_plane, on);
    }
    traverse(r, nodes, tris, nt);
    // Nothing but the ray's own state is carried across the traversal (the loop takes the whole register budget; what a store needs
    // was spilled to scratch around it): pixel coordinates, the sensor-frame ray and the depth multiplier are evaluated AGAIN here --
    // the same operations on the same operands, the same bits.  (The lane id goes through an empty asm so that the compiler does not
    // recognise the expressions and keep their first values alive.)
    int lane_again = lane;
    asm volatile("" : "+v"(lane_again));
    const int x = (tile % tiles_x) * kTileW + (lane_again % kTileW), y = (tile / tiles_x) * kTileH + (lane_again / kTileW);
    const bool active = x < width && y < height;
    const V3 rd = r.d;
    const float mult = (!LIDAR && mode == AGX_RAY_DEPTH) ? dot(rd, rdp) : 1.0f;
    V3 local = V3{0.0f, 0.0f, 1.0f};
    if (mode >= AGX_RAY_POINTCLOUD && mode != AGX_RAY_POINTCLOUD_WORLD && VARIANT != RAY_NORMAL) local = local_ray(x, y, active);
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

// How many workgroups share one (env, sensor) image.  Two rules, the larger wins:
//   * small batches: 256 CUs x 32 waves = 8192 resident waves, and tails want ~2x that in the grid (256 envs:
//     profiles/r01_small_batch.txt);
//   * locality (round 4, profiles/r04_raycast_variants.txt): with the XCD-aware mapping of k_raycast the parts of an image run
//     side by side on one XCD, so the more parts, the fewer distinct trees compete for that XCD's L2 and its CUs' scalar caches
//     (configs[2], 8192 envs: L2 misses 9.1 M -> 4.8 M per frame, HBM read requests 8.8 M -> 4.6 M, scalar-cache hit rate 29 % ->
//     40 %).  Measured optimum: ~4 tiles per wave for the pinhole camera (64 x 48: 3 parts, frame -8 %), ~2 for the 360-degree
//     LiDAR whose every bundle walks the whole tree (32 x 512: 32 parts, frame -16 %).
static int ray_split_policy(int images, int tiles, bool lidar) {
  const int waves_per_wg = kRayThreads / 64;
  const int fill = (16384 + images * waves_per_wg - 1) / (images * waves_per_wg);
  const int local = tiles / (waves_per_wg * (lidar ? 2 : 4));
  return fill > local ? fill : local;
}

// -> workgroups of the launch; *split = workgroups per image (see the kernel's mapping comment)
static unsigned ray_launch_shape(int images, int width, int height, bool lidar, int *split_out) {
  const int tw = lidar ? Tile<true>::W : Tile<false>::W, th = 64 / tw;
  const int tiles = ((width + tw - 1) / tw) * ((height + th - 1) / th), waves_per_wg = kRayThreads / 64;
  const int max_split = (tiles + waves_per_wg - 1) / waves_per_wg;
  int split = ray_split_policy(images, tiles, lidar);
  if (const int forced = option_ray_split()) split = forced;  // tuning knob: agx_set_option("ray_split", n) (profiles/raycast_split_probe.py)
  split = split < 1 ? 1 : (split > max_split ? max_split : split);
  *split_out = split;
  return (((unsigned)images + 7u) / 8u) * 8u * (unsigned)split;
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
  const int width = LIDAR ? LA.width : CA.width, height = LIDAR ? LA.height : CA.height;
  int split;
  dim3 grid(ray_launch_shape(n * ns, width, height, LIDAR, &split));
  hipLaunchKernelGGL((k_raycast<LIDAR, VARIANT>), grid, dim3(kRayThreads), 0, (hipStream_t)stream, CA, LA, RL, ray_vectors, pos, quat, tri_world,
                     tri_seg, nodes, nt, pixels, seg, split);
  return check_launch(LIDAR ? "agx_raycast_lidar" : "agx_raycast_camera");
}

}  // namespace agx

using namespace agx;

extern "C" int agx_raycast_kernel(int n, int ns, int width, int height, int lidar, int variant, char *out, int cap) {
  AGX_REQUIRE(out && cap > 0 && n > 0 && ns > 0 && width > 0 && height > 0 && variant >= 0 && variant <= 2, "bad arguments");
  int split;
  const unsigned wgs = ray_launch_shape(n * ns, width, height, lidar != 0, &split);
  snprintf(out, (size_t)cap, "k_raycast<%s,%d>_%llu", lidar ? "true" : "false", variant, (unsigned long long)wgs * kRayThreads);
  return AGX_OK;
}

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
