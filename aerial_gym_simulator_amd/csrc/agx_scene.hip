// Scene kernels for gfx950: vertex transform at reset, obstacle OBBs for the collision
// test, and an LBVH built per env entirely in LDS (one workgroup per env).
//
// These replace WarpEnv.reset_idx's tf_apply over all vertices + per-env wp.Mesh.refit()
// (warp_env_manager.py:40-54) and the wp.Mesh BVH build (warp_env_manager.py:162-166).
// Warp only REFITS the boxes of its initial topology after obstacles were moved; here the
// tree is rebuilt from the new positions (Morton order), which keeps traversal tight.
//
// The BVH is a pure accelerator: the ray-cast result is defined over all triangles with
// a (t, face-index) tie-break (DESIGN.md "closest-hit semantics"), so any topology is valid as
// long as node boxes are conservative -- they are grown by kBoxEps like Warp's 1e-3.
#include "agx_common.h"
#include "agx_device_math.h"
#include "agx_nav_parts.h"

namespace agx {

constexpr float kBoxEps = 1.0e-3f;
constexpr int kBvhThreads = 512;
constexpr int kBvhMaxTris = 2944;  // 160 KiB of LDS: 8 B x 4096 padded keys + 44 B per triangle (bvh_lds_bytes); forest_env = 2148

// triangle f of env: local frame -> world frame through its asset's pose
AGX_DEV void transform_triangle(int env, int f, int nt, int na, const float *__restrict__ tri_local, const int32_t *__restrict__ tri_asset,
                                const float *__restrict__ asset_state, float *__restrict__ tri_world) {
  const float *as = asset_state + ((size_t)env * na + tri_asset[f]) * 13;
  const V3 t = V3{as[0], as[1], as[2]};
  const Q4 q = Q4{as[3], as[4], as[5], as[6]};
  const float *src = tri_local + ((size_t)env * nt + f) * 9;
  float *dst = tri_world + ((size_t)env * nt + f) * 9;
#pragma unroll
  for (int v = 0; v < 3; ++v) {
    V3 w = tf_apply(q, t, V3{src[3 * v], src[3 * v + 1], src[3 * v + 2]});
    dst[3 * v] = w.x; dst[3 * v + 1] = w.y; dst[3 * v + 2] = w.z;
  }
}

__global__ void __launch_bounds__(256) k_scene_transform(int n, int nt, int na, const float *__restrict__ tri_local,
                                                          const int32_t *__restrict__ tri_asset,
                                                          const float *__restrict__ asset_state,
                                                          const uint8_t *__restrict__ mask, float *__restrict__ tri_world) {
  const int env = blockIdx.y;
  if (mask && !mask[env]) return;
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= nt) return;
  transform_triangle(env, f, nt, na, tri_local, tri_asset, asset_state, tri_world);
}

AGX_DEV void box_from_asset(int env, int k, int n, int na, const float *__restrict__ asset_state, const float *__restrict__ half_extents,
                            float *__restrict__ boxes);

__global__ void __launch_bounds__(256) k_boxes_from_assets(int n, int na, const float *__restrict__ asset_state,
                                                            const float *__restrict__ half_extents,
                                                            const uint8_t *__restrict__ mask, float *__restrict__ boxes) {
  const int env = blockIdx.x * blockDim.x + threadIdx.x;
  const int k = blockIdx.y;
  if (env >= n) return;
  if (mask && !mask[env]) return;
  box_from_asset(env, k, n, na, asset_state, half_extents, boxes);
}

// collision box k of env (SoA rows [k][11][n]): pose, half extents, bounding-sphere radius
AGX_DEV void box_from_asset(int env, int k, int n, int na, const float *__restrict__ asset_state, const float *__restrict__ half_extents,
                            float *__restrict__ boxes) {
  const float *as = asset_state + ((size_t)env * na + k) * 13;
  const float *he = half_extents + ((size_t)env * na + k) * 3;
  float *bx = boxes + (size_t)k * 11 * n + env;
#pragma unroll
  for (int c = 0; c < 7; ++c) bx[(size_t)c * n] = as[c];
#pragma unroll
  for (int c = 0; c < 3; ++c) bx[(size_t)(7 + c) * n] = he[c];
  // bounding-sphere radius (for the conservative cull), rounded up
  bx[(size_t)10 * n] = sqrtf(he[0] * he[0] + he[1] * he[1] + he[2] * he[2]) * 1.000001f;
}

// ------------------------------------------------------------------------------------ LBVH
constexpr float kBvhLargeFraction = 0.75f;
#ifdef AGX_SCENE_PHASE_CLOCK  // profiles/scene_phase_probe_r06.py: where a dirty env's refresh spends its time (100 MHz wall clock)
__device__ unsigned long long g_phase_clock[16];
#define AGX_PHASE(k) do { if (threadIdx.x == 0) g_phase_clock[k] = wall_clock64(); } while (0)
#define AGX_PHASE_LAST(k) do { if (threadIdx.x == kBvhThreads - 1) g_phase_clock[k] = wall_clock64(); } while (0)
#else
#define AGX_PHASE(k) do { } while (0)
#define AGX_PHASE_LAST(k) do { } while (0)
#endif
#ifdef AGX_BVH_EXPERIMENT  // profiles/bvh_sah_experiment.py: object sort codes supplied by the host
__device__ const uint32_t *g_obj_codes = nullptr;
#endif

// Multi-primitive assets (URDFs with several links: the reference's `trees`): every primitive is its own rigid
// piece whose pose follows the asset's,  prim = asset (x) local.  prim_state [N][P][13] then plays the role of
// asset_state for agx_scene_transform / agx_boxes_from_assets.
__global__ void __launch_bounds__(256) k_prims_from_assets(int n, int np_, int na, const int32_t *__restrict__ prim_asset,
                                                            const float *__restrict__ asset_state,
                                                            const float *__restrict__ local_pos,
                                                            const float *__restrict__ local_quat,
                                                            const uint8_t *__restrict__ mask, float *__restrict__ prim_state) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * np_) return;
  const int env = idx / np_;
  if (mask && !mask[env]) return;
  const float *as = asset_state + ((size_t)env * na + prim_asset[idx]) * 13;  // per env: the free assets are shuffled
  const float *lp = local_pos + (size_t)idx * 3, *lq = local_quat + (size_t)idx * 4;
  const Q4 qa = Q4{as[3], as[4], as[5], as[6]};
  const V3 c = tf_apply(qa, V3{as[0], as[1], as[2]}, V3{lp[0], lp[1], lp[2]});
  const Q4 q = quat_mul(qa, Q4{lq[0], lq[1], lq[2], lq[3]});
  float *o = prim_state + (size_t)idx * 13;
  o[0] = c.x; o[1] = c.y; o[2] = c.z;
  o[3] = q.x; o[4] = q.y; o[5] = q.z; o[6] = q.w;
#pragma unroll
  for (int k = 7; k < 13; ++k) o[k] = as[k];
}

// Kinematic obstacles (EnvManager.step(actions, env_actions), obstacle_manager.py:40-44): the env action of an
// obstacle is its twist (world-frame linear and angular velocity), written into the root state every sub-step;
// the pose follows with the integrator's rule (p += v dt, exponential map for q) for k sub-steps.
__global__ void __launch_bounds__(256) k_assets_integrate(int count, float *__restrict__ asset_state, const float *__restrict__ twist,
                                                           float dt, int k) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= count) return;
  float *st = asset_state + (size_t)a * 13;
  const float *tw = twist + (size_t)a * 6;
  V3 p = V3{st[0], st[1], st[2]};
  Q4 q = Q4{st[3], st[4], st[5], st[6]};
  const V3 v = V3{tw[0], tw[1], tw[2]}, w = V3{tw[3], tw[4], tw[5]};
  const float wm2 = dot(w, w);
  float x1 = 0.0f, y1 = 0.0f, z1 = 0.0f, cs = 1.0f;
  if (wm2 != 0.0f) {
    float wm = sqrtf(wm2);
    float sn;
    sincos_bounded(fminf(dt * wm * 0.5f, 60.0f), sn, cs);
    float sc = sn / wm;
    x1 = w.x * sc; y1 = w.y * sc; z1 = w.z * sc;
  }
  for (int s = 0; s < k; ++s) {
    p = V3{p.x + v.x * dt, p.y + v.y * dt, p.z + v.z * dt};
    if (wm2 != 0.0f) {
      float rx = x1 * q.w + y1 * q.z - z1 * q.y;
      float ry = y1 * q.w + z1 * q.x - x1 * q.z;
      float rz = z1 * q.w + x1 * q.y - y1 * q.x;
      float rw = -(x1 * q.x) - y1 * q.y - z1 * q.z;
      rx += q.x * cs; ry += q.y * cs; rz += q.z * cs; rw += q.w * cs;
      float nn = sqrtf(rx * rx + ry * ry + rz * rz + rw * rw);
      q = Q4{rx / nn, ry / nn, rz / nn, rw / nn};
    }
  }
  st[0] = p.x; st[1] = p.y; st[2] = p.z;
  st[3] = q.x; st[4] = q.y; st[5] = q.z; st[6] = q.w;
#pragma unroll
  for (int c = 0; c < 6; ++c) st[7 + c] = tw[c];  // obstacle_linvel / obstacle_angvel
}

// obstacle parked outside the env by the curriculum (asset_manager.py:71 puts it at -1000 m)
AGX_DEV bool tri_parked(const float *t) { return t[0] < -900.0f && t[1] < -900.0f && t[2] < -900.0f; }

AGX_DEV uint32_t expand_bits10(uint32_t v) {
  v = (v * 0x00010001u) & 0xFF0000FFu;
  v = (v * 0x00000101u) & 0x0F00F00Fu;
  v = (v * 0x00000011u) & 0xC30C30C3u;
  v = (v * 0x00000005u) & 0x49249249u;
  return v;
}

AGX_DEV int delta_keys(const unsigned long long *keys, int nt, int i, int j) {
  if (j < 0 || j >= nt) return -1;
  return __clzll(keys[i] ^ keys[j]);  // keys are unique (the triangle index is in the low word)
}

// floats of the LDS region that holds the internal nodes' boxes and, before them, the bounds reduction's scratch
AGX_DEV constexpr int bvh_box_floats_c(int nt, int threads) { return (nt - 1) * 6 > 6 * threads ? (nt - 1) * 6 : 6 * threads; }
// box of tree node c: internal nodes from LDS, leaves (c >= n_int) from their triangle, grown by kBoxEps
AGX_DEV void node_box(const float *box, const unsigned long long *keys, const float *__restrict__ tris, int n_int, int c, float (&o)[6]) {
  if (c >= n_int) {
    const int f = (int)(uint32_t)(keys[c - n_int] & 0xFFFFFFFFull);
    const float *t = tris + (size_t)f * 9;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      o[k] = fminf(fminf(t[k], t[3 + k]), t[6 + k]) - kBoxEps;
      o[3 + k] = fmaxf(fmaxf(t[k], t[3 + k]), t[6 + k]) + kBoxEps;
    }
  } else {
#pragma unroll
    for (int k = 0; k < 6; ++k) o[k] = box[(size_t)c * 6 + k];
  }
}

// Bitonic sort of m (a power of two) 64-bit keys in LDS by the whole workgroup.
// A thread owns elements tid + q * kBvhThreads, so a WAVE owns whole 64-element blocks: the stages with j < 64 exchange
// inside a block, i.e. inside the wave (LDS operations of a wave execute in order), and need no workgroup barrier.
// Barriers remain where the next stage reads what other waves wrote: 21 of the 66 stages for 2048 keys.
AGX_DEV void bitonic_sort_lds(unsigned long long *keys, int m, int tid) {
  for (int k = 2; k <= m; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < m; i += kBvhThreads) {
        int ixj = i ^ j;
        if (ixj > i) {
          unsigned long long a = keys[i], b = keys[ixj];
          bool up = (i & k) == 0;
          if ((a > b) == up) { keys[i] = b; keys[ixj] = a; }
        }
      }
      const int next_j = j > 1 ? (j >> 1) : k;  // (the stage after j == 1 is (2k, k))
      if (j >= 64 || next_j >= 64) {
        __syncthreads();
      } else {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // compiler: keep the stages' LDS accesses in order
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
}

// AGX_BVH_BOX_OBJECTS.  An internal node whose leaves are exactly ONE object's 12 triangles becomes an OBJECT NODE if those triangles
// make an orthogonal box of trimesh.creation.box's topology -- decided from the world-frame triangles alone, every vertex checked,
// in three parallel steps (a thread that validated an object alone was a chain of ~120 dependent loads: +20 us on a 60 us build):
//   1. per internal node: is its key range one object's twelve keys?  (LDS only)              -> objroot[object] = node
//   2. per triangle: are its three vertices corners of its object's box, in the plane of the face the ray-cast's table expects it
//      in; per face pair: do the two triangles have four distinct corners between them (they tile the rectangle)?  -> objroot[o] = -1
//   3. per object: the verdict into bit 0 of counter[node] (read by the node itself and by its parent in the emit stage).
// Faces of trimesh's box (corners ({0,1}^3 - 0.5) * extents, index 4 x + 2 y + z) by triangle:
//   0 = (1, 3, 0), 1 = (4, 1, 0), 2 = (0, 3, 2), 11 = (7, 5, 6);  -x (0, 2)  +x (10, 11)  -y (1, 5)  +y (7, 9)  -z (3, 8)  +z (4, 6).
struct BoxFrame {
  V3 nx, ny, nz, cen;
  float hx, hy, hz, tol;
};

// the box four corners of an object's first triangles span (corners 0, 1, 2, 4; the opposite corner 7 must close it)
AGX_DEV bool box_frame(const float *__restrict__ t, BoxFrame &F) {
  const V3 v1 = V3{t[0], t[1], t[2]}, v0 = V3{t[6], t[7], t[8]};          // triangle 0 = (1, 3, 0)
  const V3 v4 = V3{t[9], t[10], t[11]};                                     // triangle 1 = (4, 1, 0)
  const V3 v2 = V3{t[18 + 6], t[18 + 7], t[18 + 8]};                        // triangle 2 = (0, 3, 2)
  const V3 v7 = V3{t[99], t[100], t[101]};                                  // triangle 11 = (7, 5, 6)
  const V3 ex = v4 - v0, ey = v2 - v0, ez = v1 - v0;
  const float lx = sqrtf(dot(ex, ex)), ly = sqrtf(dot(ey, ey)), lz = sqrtf(dot(ez, ez));
  if (!(lx > 1.0e-5f && ly > 1.0e-5f && lz > 1.0e-5f) || !(lx < 1.0e4f && ly < 1.0e4f && lz < 1.0e4f)) return false;
  F.nx = ex * (1.0f / lx); F.ny = ey * (1.0f / ly); F.nz = ez * (1.0f / lz);
  if (fabsf(dot(F.nx, F.ny)) > 1.0e-4f || fabsf(dot(F.ny, F.nz)) > 1.0e-4f || fabsf(dot(F.nz, F.nx)) > 1.0e-4f) return false;
  const V3 far = v0 + ex + ey + ez - v7;  // the opposite corner is where a box has it
  if (!(sqrtf(dot(far, far)) <= 1.0e-4f * (lx + ly + lz))) return false;
  F.cen = v0 + (ex + ey + ez) * 0.5f;
  F.hx = 0.5f * lx; F.hy = 0.5f * ly; F.hz = 0.5f * lz;
  F.tol = 1.0e-4f * (lx + ly + lz) + 1.0e-5f;
  return true;
}

// triangle q of the object (9 floats at t): three distinct corners of the box, all in the plane of face `kFaceOf[q]`; -> the corners' ids
AGX_DEV bool box_triangle_ok(const BoxFrame &F, const float *__restrict__ t, int q, uint32_t &ids) {
  const int face = (int)((0x113435254020ull >> (4 * q)) & 15ull);  // {0, 2, 0, 4, 5, 2, 5, 3, 4, 3, 1, 1}[q] = 2 * axis + (plus side)
  const int axis = face >> 1;
  const float want = (face & 1) ? 1.0f : -1.0f;
  ids = 0u;
  bool ok = true;
#pragma unroll
  for (int v = 0; v < 3; ++v) {
    const V3 p = V3{t[3 * v] - F.cen.x, t[3 * v + 1] - F.cen.y, t[3 * v + 2] - F.cen.z};
    const float l0 = dot(F.nx, p), l1 = dot(F.ny, p), l2 = dot(F.nz, p);
    ok = ok && !(fabsf(fabsf(l0) - F.hx) > F.tol || fabsf(fabsf(l1) - F.hy) > F.tol || fabsf(fabsf(l2) - F.hz) > F.tol);
    const float la = axis == 0 ? l0 : (axis == 1 ? l1 : l2);
    ok = ok && (la * want > 0.0f);
    ids |= 1u << ((l0 > 0.0f ? 4 : 0) | (l1 > 0.0f ? 2 : 0) | (l2 > 0.0f ? 1 : 0));
  }
  return ok && __popc(ids) == 3;
}

// the object node's record (include/aerial_gym_hip.h): the frame of object `obj`
AGX_DEV void box_object_record(const float *__restrict__ tris, int obj, float *rec) {
  BoxFrame F;
  box_frame(tris + (size_t)obj * 12 * 9, F);  // (validated before)
  rec[0] = F.nx.x; rec[1] = F.nx.y; rec[2] = F.nx.z; rec[3] = F.hx;
  rec[4] = F.ny.x; rec[5] = F.ny.y; rec[6] = F.ny.z; rec[7] = F.hy;
  rec[8] = F.nz.x; rec[9] = F.nz.y; rec[10] = F.nz.z; rec[11] = F.hz;
  rec[12] = F.cen.x; rec[13] = F.cen.y; rec[14] = F.cen.z; rec[15] = __int_as_float(obj * 12);
}

// ---------------------------------------------------------------------------------------------------------------------
// OBJECT-LEVEL BUILD (round 6).  With AGX_BVH_BOX_OBJECTS only 2K - 1 records of the triangle-level tree are ever reached on a box
// scene (K - 1 internal nodes over the objects, K object nodes) -- yet it Morton-sorted all 12 K triangles and built 12 K - 1 nodes
// through ~25 workgroup barriers (55-60 us for ONE env).  Here the tree is built over the K OBJECTS:
//   phase 1 (all waves)   per (object, face): the bounds of the face's two triangles -> LDS, and whether they are where
//                         trimesh.creation.box has them in the object's box (box_frame / box_triangle_ok, the same verdicts as the
//                         triangle-level build's three steps)
//   -- one workgroup barrier --
//   phase 2 (waves 0-3)   a lane per object: AABB, sort centre, Morton code (the triangle-level build's object key, bit for bit); rank
//                         sort of the K keys; Karras radix tree over them; boxes bottom-up; the K - 1 internal records with both child
//                         boxes inline.  Five workgroup barriers (one wave doing all of it needs none but runs every dependent LDS
//                         round trip of two objects in sequence: 15 us against 8, profiles/r06_scene_refresh_phases.txt).
//   phase 3 (waves 4-7)   per object, concurrently: its OBJECT NODE record (a recognised box), or -- parked beyond the curriculum level
//                         at -1000 m where float32 no longer resolves a box, or simply not a box -- a fixed five-node subtree over its
//                         six triangle pairs (two-triangle leaves), so that every triangle stays reachable.
// Record layout: internal node i of the object tree at index i (root 0); object o owns indices K - 1 + 5 o .. + 4: its object record
// at the first, or its subtree (root at the fifth).  2 <= K <= 256.  The tree is a pure accelerator (closest hit over ALL triangles,
// ties by index): frames are bit-identical whatever tree is built, which is what the test suites hold.
constexpr int kObjMax = 256;
__host__ __device__ constexpr int obj_lds_bytes_c(int nt) { return kObjMax * (8 + 8 + 24 + 24 + 8 + 8 + 4 + 8 + 4 + 1 + 6) + (nt / 2) * 24 + 64; }

AGX_DEV void obj_store_node(float *__restrict__ o, const float (&a)[6], const float (&b)[6], int rl, int rr, int sl, int sr) {
  o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = __int_as_float(rl);
  o[4] = a[3]; o[5] = a[4]; o[6] = a[5]; o[7] = __int_as_float(rr);
  o[8] = b[0]; o[9] = b[1]; o[10] = b[2]; o[11] = __int_as_float(sl);
  o[12] = b[3]; o[13] = b[4]; o[14] = b[5]; o[15] = __int_as_float(sr);
}
AGX_DEV void box_union(const float (&a)[6], const float (&b)[6], float (&o)[6]) {
#pragma unroll
  for (int c = 0; c < 3; ++c) { o[c] = fminf(a[c], b[c]); o[3 + c] = fmaxf(a[3 + c], b[3 + c]); }
}

// the six faces of trimesh's box by triangle pair: -x (0, 2)  +x (10, 11)  -y (1, 5)  +y (7, 9)  -z (3, 8)  +z (4, 6)
AGX_DEV constexpr int pair_a(int fc) { return (int)((0x4371A0u >> (4 * fc)) & 15u); }  // nibble fc of 0x4371A0 / 0x6895B2 (the ray-cast kernel's face table)
AGX_DEV constexpr int pair_b(int fc) { return (int)((0x6895B2u >> (4 * fc)) & 15u); }

AGX_DEV void bvh_build_objects_env(int env, int nt, const float *__restrict__ tri_world, float *__restrict__ nodes) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int K = nt / 12, tid = threadIdx.x, lane = tid & 63;
  unsigned long long *okey = reinterpret_cast<unsigned long long *>(smem);              // [kObjMax] code . object
  unsigned long long *skey = okey + kObjMax;                                            // [kObjMax] the same, sorted
  float *oaabb = reinterpret_cast<float *>(skey + kObjMax);                             // [kObjMax][6] object AABBs
  float *nbox = oaabb + kObjMax * 6;                                                    // [kObjMax][6] internal nodes' boxes
  int *parent = reinterpret_cast<int *>(nbox + kObjMax * 6);                            // [2 kObjMax] internal i at i, sorted leaf p at K - 1 + p
  int *child = parent + 2 * kObjMax;                                                    // [kObjMax][2] (a leaf child c is K - 1 + p)
  int *counter = child + 2 * kObjMax;                                                   // [kObjMax]
  int *boxoff = counter + kObjMax;                                                      // [kObjMax][2]
  float *red = reinterpret_cast<float *>(boxoff + 2 * kObjMax);                         // [4][6] the waves' bounds of the sort centres (+ pad)
  uint8_t *objok = reinterpret_cast<uint8_t *>(red + kObjMax);                          // [kObjMax] a recognised box
  uint8_t *fok = objok + kObjMax;                                                       // [6 kObjMax] this face's two triangles are where the box has them
  float *pb = reinterpret_cast<float *>(fok + 6 * kObjMax);                             // [6 K][6] bounds of the face pairs
  const float *tris = tri_world + (size_t)env * nt * 9;
  float *out = nodes + (size_t)env * (nt - 1) * 16;

  // ---- phase 1: a thread per (object, face): the two triangles trimesh.creation.box puts on that face
  for (int t = tid; t < 6 * K; t += kBvhThreads) {
    const int obj = t / 6, fc = t - obj * 6;
    const int qa = (int)((0x4371A0u >> (4 * fc)) & 15u), qb = (int)((0x6895B2u >> (4 * fc)) & 15u);  // kPairA / kPairB as nibbles
    const float *to = tris + (size_t)obj * 12 * 9;
    float a[9], b[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) { a[k] = to[9 * qa + k]; b[k] = to[9 * qb + k]; }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      pb[6 * t + c] = fminf(fminf(fminf(a[c], a[3 + c]), a[6 + c]), fminf(fminf(b[c], b[3 + c]), b[6 + c]));
      pb[6 * t + 3 + c] = fmaxf(fmaxf(fmaxf(a[c], a[3 + c]), a[6 + c]), fmaxf(fmaxf(b[c], b[3 + c]), b[6 + c]));
    }
    bool ok = !tri_parked(to);
    if (ok) {
      BoxFrame F;
      uint32_t ids = 0u, ids2 = 0u;
      ok = box_frame(to, F) && box_triangle_ok(F, a, qa, ids) && box_triangle_ok(F, b, qb, ids2) && __popc(ids | ids2) == 4;
    }
    fok[t] = (ok ? 1 : 0) | (tri_parked(to) ? 2 : 0);  // (face 0 holds triangle 0, whose first vertex decides "parked")
  }
  __syncthreads();
  AGX_PHASE(4);

  constexpr int kTreeThreads = kObjMax;  // waves 0-3: one object / sorted position / tree node per lane; waves 4-7: phase 3
  if (tid >= kTreeThreads) {
    // ---- phase 3 (waves 4-7): the records the objects own
    for (int o = tid - kTreeThreads; o < K; o += kBvhThreads - kTreeThreads) {
      bool ok = true;
#pragma unroll
      for (int j = 0; j < 6; ++j) ok = ok && (fok[6 * o + j] & 1) != 0;
      const int base = K - 1 + 5 * o, g0 = 12 * o;
      float *rec = out + (size_t)base * 16;
      if (ok) {
        box_object_record(tris, o, rec);
        continue;
      }
      // six two-triangle leaves (the face pairs of trimesh's box; for anything else just six pairs), five nodes over them
      // (one pair of faces at a time: two boxes in, one node out, their union kept -- 36 live floats at once spilled)
      float un[3][6];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        float p0[6], p1[6];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          p0[c] = pb[6 * (6 * o + 2 * j) + c] - kBoxEps; p0[3 + c] = pb[6 * (6 * o + 2 * j) + 3 + c] + kBoxEps;
          p1[c] = pb[6 * (6 * o + 2 * j + 1) + c] - kBoxEps; p1[3 + c] = pb[6 * (6 * o + 2 * j + 1) + 3 + c] + kBoxEps;
        }
        obj_store_node(rec + 16 * j, p0, p1, ~(g0 + pair_a(2 * j)), ~(g0 + pair_a(2 * j + 1)), g0 + pair_b(2 * j), g0 + pair_b(2 * j + 1));
        box_union(p0, p1, un[j]);
      }
      obj_store_node(rec + 48, un[0], un[1], base + 0, base + 1, -1, -1);
      float d[6];
      box_union(un[0], un[1], d);
      obj_store_node(rec + 64, d, un[2], base + 3, base + 2, -1, -1);
    }
    AGX_PHASE_LAST(8);
  }

  // ---- phase 2 (waves 0-3, lane = object, then sorted position, then node): the tree over the objects.  Every wave of the
  // workgroup takes part in the five barriers between its steps.
  const int o = tid;
  const bool has_obj = tid < K;
  float cen[3] = {0.0f, 0.0f, 0.0f}, big = -1.0f;
  {
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    if (has_obj) {
      float alo[3] = {INFINITY, INFINITY, INFINITY}, ahi[3] = {-INFINITY, -INFINITY, -INFINITY};
      bool ok = true;
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        ok = ok && (fok[6 * o + j] & 1) != 0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          alo[c] = fminf(alo[c], pb[6 * (6 * o + j) + c]);
          ahi[c] = fmaxf(ahi[c], pb[6 * (6 * o + j) + 3 + c]);
        }
      }
      objok[o] = ok ? 1 : 0;
      const bool parked = (fok[6 * o] & 2) != 0;
      float bg = 0.0f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        oaabb[6 * o + c] = alo[c] - kBoxEps;      // (grown once, here: every box above is a union of these)
        oaabb[6 * o + 3 + c] = ahi[c] + kBoxEps;
        cen[c] = 0.5f * (alo[c] + ahi[c]);
        bg = fmaxf(bg, ahi[c] - alo[c]);
        if (!parked) { lo[c] = fminf(lo[c], cen[c]); hi[c] = fmaxf(hi[c], cen[c]); }
      }
      big = parked ? -1.0f : bg;
    }
    if (tid < kTreeThreads) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float a = lo[c], b = hi[c];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
          a = fminf(a, __shfl_xor(a, off));
          b = fmaxf(b, __shfl_xor(b, off));
        }
        if (lane == 0) { red[6 * (tid >> 6) + c] = a; red[6 * (tid >> 6) + 3 + c] = b; }
      }
    }
  }
  __syncthreads();
  AGX_PHASE(5);
  if (has_obj) {
    float blo[3], inv[3], max_ext = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float a = INFINITY, b = -INFINITY;
#pragma unroll
      for (int wv = 0; wv < kTreeThreads / 64; ++wv) { a = fminf(a, red[6 * wv + c]); b = fmaxf(b, red[6 * wv + 3 + c]); }
      blo[c] = a;
      const float ext = b - a;
      inv[c] = (ext > 0.0f && ext < INFINITY) ? 1023.0f / ext : 0.0f;
      max_ext = c == 0 ? ext : fmaxf(max_ext, ext);
    }
    uint32_t code = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float qv = fminf(fmaxf((cen[c] - blo[c]) * inv[c], 0.0f), 1023.0f);
      code |= expand_bits10((uint32_t)qv) << (2 - c);
    }
    const bool parked = big < 0.0f;
    if (big > kBvhLargeFraction * max_ext) code |= 1u << 30;  // wall slabs: their own subtree under the root
    if (parked) code = 0xFFFFFFFEu;                            // one subtree at the end of the order, culled at its root
    okey[o] = ((unsigned long long)(parked ? code : (code & ~7u)) << 32) | (unsigned long long)(uint32_t)o;
  }
  __syncthreads();
  if (has_obj) {  // rank sort: the keys are unique (the object index is the low word)
    const unsigned long long mine = okey[o];
    int rank = 0;
#pragma unroll 8
    for (int g = 0; g < K; ++g) rank += okey[g] < mine ? 1 : 0;  // (the same address in every lane: a broadcast read)
    skey[rank] = mine;
  }
  __syncthreads();
  AGX_PHASE(6);
  // Karras 2012 over the K sorted keys: children, parents, where the children's boxes lie
  const int n_int = K - 1;
  if (tid < n_int) {
    const int i = tid;
    const int d = (delta_keys(skey, K, i, i + 1) - delta_keys(skey, K, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = delta_keys(skey, K, i, i - d);
    int lmax = 2;
    while (delta_keys(skey, K, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
      if (delta_keys(skey, K, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta_keys(skey, K, i, j);
    int s = 0, t = l;
    do {
      t = (t + 1) >> 1;
      if (delta_keys(skey, K, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + min(d, 0);
    const int left = (min(i, j) == gamma) ? (n_int + gamma) : gamma;
    const int right = (max(i, j) == gamma + 1) ? (n_int + gamma + 1) : (gamma + 1);
    child[2 * i] = left;
    child[2 * i + 1] = right;
    parent[left] = i;
    parent[right] = i;
    counter[i] = 0;
    // float offsets from oaabb (nbox follows it): the bottom-up pass reads the boxes without going through the key table
    boxoff[2 * i] = left >= n_int ? 6 * (int)(uint32_t)(skey[left - n_int] & 0xFFFFFFFFull) : 6 * (kObjMax + left);
    boxoff[2 * i + 1] = right >= n_int ? 6 * (int)(uint32_t)(skey[right - n_int] & 0xFFFFFFFFull) : 6 * (kObjMax + right);
  }
  if (tid == 0) parent[0] = -1;
  __syncthreads();
  AGX_PHASE(9);
  // bottom-up: the second arriver at a node merges its children's boxes
  if (has_obj) {
    int node = parent[n_int + tid];
    while (node >= 0) {
      __threadfence_block();  // release: this thread's box is written before the arrival
      const int arrived = atomicAdd(&counter[node], 1);
      const int o0 = boxoff[2 * node], o1 = boxoff[2 * node + 1], up = parent[node];
      if (arrived == 0) break;
      __threadfence_block();  // acquire: the sibling's box is read after the arrival
      float a[6], b[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) { a[k] = oaabb[o0 + k]; b[k] = oaabb[o1 + k]; }
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        nbox[6 * node + c] = fminf(a[c], b[c]);
        nbox[6 * node + 3 + c] = fmaxf(a[3 + c], b[3 + c]);
      }
      node = up;
    }
  }
  __syncthreads();
  AGX_PHASE(10);
  // emit the K - 1 internal records, both child boxes inline
  if (tid < n_int) {
    const int i = tid;
    float bx[2][6];
    int ref[2];
#pragma unroll
    for (int side = 0; side < 2; ++side) {
      const int c = child[2 * i + side];
      if (c >= n_int) {
        const int ob = (int)(uint32_t)(skey[c - n_int] & 0xFFFFFFFFull);
#pragma unroll
        for (int k = 0; k < 6; ++k) bx[side][k] = oaabb[6 * ob + k];
        ref[side] = objok[ob] ? ((K - 1 + 5 * ob) | AGX_BVH_OBJECT_REF) : (K - 1 + 5 * ob + 4);
      } else {
#pragma unroll
        for (int k = 0; k < 6; ++k) bx[side][k] = nbox[6 * c + k];
        ref[side] = c;
      }
    }
    obj_store_node(out + (size_t)i * 16, bx[0], bx[1], ref[0], ref[1], -1, -1);
  }
  AGX_PHASE(7);
}

// Node record written to HBM (16 floats):
//   [0..2] lo_left  [3] child_left (int bits)   [4..6] hi_left  [7] child_right (int bits)
//   [8..10] lo_right [11] second_left (int)     [12..14] hi_right [15] second_right (int)
// child >= 0: internal node index, child < 0: leaf holding triangle ~child and, if second >= 0, that triangle too.
// OBJ: the launch is known to take the object-level build (object_level_build() on the host: launch arguments only) -- each kernel
// instance carries ONE builder (both inlined in one kernel spilled loop-carried values to scratch)
template <bool OBJ>
AGX_DEV void bvh_build_env(int env, int nt, int npad, int ppo, const float *__restrict__ tri_world, float *__restrict__ nodes) {
  if (OBJ) {
    bvh_build_objects_env(env, nt, tri_world, nodes);
    return;
  }
  const bool force_full_sort = (ppo & AGX_BVH_FULL_SORT) != 0;
  const bool box_objects = (ppo & AGX_BVH_BOX_OBJECTS) != 0;
  ppo &= ~(AGX_BVH_FULL_SORT | AGX_BVH_BOX_OBJECTS | AGX_BVH_OBJECT_TREE);
  extern __shared__ __align__(16) unsigned char smem[];
  unsigned long long *keys = reinterpret_cast<unsigned long long *>(smem);         // [npad]
  // Only the INTERNAL nodes' boxes live in LDS: a leaf's box is three min / max over its triangle, recomputed where it is
  // needed (twice).  That, and the reduction scratch sharing the box region (it is dead before the first box is written),
  // takes the footprint from 115 KB to 72 KB for T = 1272: TWO workgroups per CU, i.e. 512 envs in flight instead of 256
  // (with 300-400 dirty envs per step at 8192 envs the rebuild was two rounds: 150 -> see profiles/r02_small_batch.txt).
  float *box = reinterpret_cast<float *>(keys + npad);                              // [nt-1][6] (>= the reduction scratch)
  int *parent = reinterpret_cast<int *>(box + bvh_box_floats_c(nt, kBvhThreads));                  // [2nt-1]
  int *child = parent + (2 * nt - 1);                                               // [nt-1][2]
  int *counter = child + 2 * (nt - 1);                                              // [nt-1]
  float *red = box;                                                                 // [6][kBvhThreads], first phase only
  const int tid = threadIdx.x;
  const float *tris = tri_world + (size_t)env * nt * 9;

  // --- Morton grid = bounds of the sort centres of everything that is in the env.  Obstacles beyond the
  // curriculum level are parked at -1000 m (asset_manager.py:71): letting them into the bounds would stretch
  // the 10-bit grid over a kilometre.  With ppo > 0 the scene is a soup of K = nt / ppo objects of ppo
  // consecutive triangles each (boxes: 12): the sort centre is the OBJECT's, so the triangles of an object
  // stay together (their keys differ only in the index word) and the radix tree becomes a two-level
  // hierarchy, objects on top.  Measured on the config-3 scene: 105 -> see profiles/r01_raycast_variants.txt.
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  float *ocen = reinterpret_cast<float *>(counter);  // [K][8] centre, largest extent (< 0: parked), AABB lo, -; counter is not live yet
  const int K = ppo > 0 ? nt / ppo : 0;
  // (every thread takes triangles -- nine independent loads each -- and leaves their bounds in LDS; the object's thread then
  //  reduces ppo LDS entries.  One thread per object walking its 3 ppo vertices was 36 memory latencies in sequence: the longest
  //  phase of the build.  The scratch is the box region, one entry longer than it: it runs into `parent`, not live yet.)
  float *tb = box;  // [nt][6]
  if (ppo > 0) {
    for (int f = tid; f < nt; f += kBvhThreads) {
      const float *t = tris + (size_t)f * 9;
      float v[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) v[k] = t[k];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        tb[6 * f + c] = fminf(fminf(v[c], v[3 + c]), v[6 + c]);
        tb[6 * f + 3 + c] = fmaxf(fmaxf(v[c], v[3 + c]), v[6 + c]);
      }
    }
    __syncthreads();
  }
  for (int o = tid; o < K; o += kBvhThreads) {
    const float *t = tris + (size_t)o * ppo * 9;
    float alo[3] = {INFINITY, INFINITY, INFINITY}, ahi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int j = 0; j < ppo; ++j)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        alo[c] = fminf(alo[c], tb[6 * (o * ppo + j) + c]);
        ahi[c] = fmaxf(ahi[c], tb[6 * (o * ppo + j) + 3 + c]);
      }
    const bool parked = tri_parked(t);
    float big = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float cen = 0.5f * (alo[c] + ahi[c]);
      ocen[8 * o + c] = cen;
      ocen[8 * o + 4 + c] = alo[c];
      big = fmaxf(big, ahi[c] - alo[c]);
      if (!parked) { lo[c] = fminf(lo[c], cen); hi[c] = fmaxf(hi[c], cen); }
    }
    ocen[8 * o + 3] = parked ? -1.0f : big;
  }
  for (int f = tid; f < nt && ppo <= 0; f += kBvhThreads) {
    const float *t = tris + (size_t)f * 9;
    if (tri_parked(t)) continue;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float cen = (t[c] + t[3 + c] + t[6 + c]) * (1.0f / 3.0f);
      lo[c] = fminf(lo[c], cen);
      hi[c] = fmaxf(hi[c], cen);
    }
  }
  if (ppo > 0) __syncthreads();  // the reduction scratch below shares the region the triangle bounds were read from
  // bounds of the sort centres over the workgroup: inside a wave by shuffles (min / max are exact: any order gives the same bits),
  // then the eight waves' results through LDS -- two barriers (the pairwise tree over 512 LDS entries took ten)
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float a = lo[c], b = hi[c];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      a = fminf(a, __shfl_xor(a, off));
      b = fmaxf(b, __shfl_xor(b, off));
    }
    if ((tid & 63) == 0) {
      red[c * (kBvhThreads / 64) + (tid >> 6)] = a;
      red[(3 + c) * (kBvhThreads / 64) + (tid >> 6)] = b;
    }
  }
  __syncthreads();
  float blo[3], bhi[3], inv[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float a = INFINITY, b = -INFINITY;
#pragma unroll
    for (int wv = 0; wv < kBvhThreads / 64; ++wv) {
      a = fminf(a, red[c * (kBvhThreads / 64) + wv]);
      b = fmaxf(b, red[(3 + c) * (kBvhThreads / 64) + wv]);
    }
    blo[c] = a;
    bhi[c] = b;
    float ext = b - a;
    inv[c] = (ext > 0.0f && ext < INFINITY) ? 1023.0f / ext : 0.0f;  // ext = -inf - inf when every primitive is parked
  }
  float max_ext = fmaxf(fmaxf(bhi[0] - blo[0], bhi[1] - blo[1]), bhi[2] - blo[2]);
  // --- keys: [large flag | 30-bit Morton code of the sort centre] . [triangle index]
  // --- objects of ppo triangles each (boxes: 12): the order of the keys is the order of the OBJECTS' codes, and inside an
  // object (face class, index).  So sort the K object keys (128 instead of 2048: the full sort was 35 of the build's 67 us) and
  // RANK every triangle inside its object against the ppo - 1 others.  This is the full sort's order exactly unless two
  // objects that are in the env share all 27 upper Morton bits -- their triangles would interleave by face class --, which is
  // looked for, and then the full sort runs.  (Parked objects all carry the code 0xFFFFFFFE: index order, which the object
  // keys' low word gives.)
  bool presorted = false;
#ifndef AGX_BVH_EXPERIMENT
  if (ppo >= 2 && K >= 2 && K * ppo == nt && !force_full_sort) {
    __syncthreads();  // (the reduction's result is in registers everywhere: its scratch is reused)
    int Kpad = 1;
    while (Kpad < K) Kpad <<= 1;
    unsigned long long *okey = reinterpret_cast<unsigned long long *>(box);  // [Kpad] code . object
    uint32_t *ocode = reinterpret_cast<uint32_t *>(okey + Kpad);             // [K]    the object's code (Morton | large flag, or parked)
    int *orank = reinterpret_cast<int *>(ocode + K);                         // [K]    position of the object in the order
    int *tie = orank + K;                                                     // [1]
    uint8_t *cls = reinterpret_cast<uint8_t *>(tie + 1);                      // [nt]   face class of the triangle (0 when parked)
    if (tid == 0) *tie = 0;
    for (int o = tid; o < Kpad; o += kBvhThreads) {
      unsigned long long k64 = ~0ull;
      if (o < K) {
        const float *oc = ocen + 8 * o;
        uint32_t code = 0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float qv = fminf(fmaxf((oc[c] - blo[c]) * inv[c], 0.0f), 1023.0f);
          code |= expand_bits10((uint32_t)qv) << (2 - c);
        }
        const float big = oc[3];
        const bool parked = big < 0.0f;
        if (big > kBvhLargeFraction * max_ext) code |= 1u << 30;
        if (parked) code = 0xFFFFFFFEu;
        ocode[o] = code;
        k64 = ((unsigned long long)(parked ? code : (code & ~7u)) << 32) | (unsigned long long)(uint32_t)o;
      }
      okey[o] = k64;
    }
    for (int f = tid; f < nt; f += kBvhThreads) {
      const float *t = tris + (size_t)f * 9;
      uint32_t fc = 0;
      if (!(ocen[8 * (f / ppo) + 3] < 0.0f)) {
        V3 e1 = V3{t[3] - t[0], t[4] - t[1], t[5] - t[2]}, e2 = V3{t[6] - t[0], t[7] - t[1], t[8] - t[2]};
        V3 nrm = cross_plain(e1, e2);
        float ax = fabsf(nrm.x), ay = fabsf(nrm.y), az = fabsf(nrm.z);
        int d = (ax >= ay && ax >= az) ? 0 : (ay >= az ? 1 : 2);
        float comp = d == 0 ? nrm.x : (d == 1 ? nrm.y : nrm.z);
        fc = (uint32_t)(2 * d + (comp < 0.0f ? 1 : 0));
      }
      cls[f] = (uint8_t)fc;
    }
    __syncthreads();
    bitonic_sort_lds(okey, Kpad, tid);
    __syncthreads();
    for (int p = tid; p < K; p += kBvhThreads) {
      const unsigned long long a = okey[p];
      orank[(int)(uint32_t)(a & 0xFFFFFFFFull)] = p;
      if (p + 1 < K) {
        const uint32_t c0 = (uint32_t)(a >> 32), c1 = (uint32_t)(okey[p + 1] >> 32);
        if (c0 == c1 && c0 != 0xFFFFFFFEu) *tie = 1;  // (every writer stores the same value)
      }
    }
    __syncthreads();
    presorted = *tie == 0;  // the same for the whole workgroup
    if (presorted) {
      for (int f = tid; f < npad; f += kBvhThreads) {
        if (f < nt) {
          const int o = f / ppo, g0 = o * ppo;
          const uint32_t oc = ocode[o];
          const uint32_t cf = cls[f];
          const uint32_t code = oc == 0xFFFFFFFEu ? oc : ((oc & ~7u) | cf);
          int r = 0;
          for (int g = g0; g < g0 + ppo; ++g) {
            const uint32_t cg = cls[g];
            r += (cg < cf || (cg == cf && g < f)) ? 1 : 0;
          }
          keys[orank[o] * ppo + r] = ((unsigned long long)code << 32) | (unsigned long long)(uint32_t)f;
        } else {
          keys[f] = ~0ull;
        }
      }
    }
    __syncthreads();
  }
#endif
  for (int f = tid; f < npad && !presorted; f += kBvhThreads) {
    unsigned long long key = ~0ull;
    if (f < nt) {
      const float *t = tris + (size_t)f * 9;
      uint32_t code = 0;
      float big = 0.0f;
      bool parked;
      if (ppo > 0) {
        const float *oc = ocen + 8 * (f / ppo);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float qv = fminf(fmaxf((oc[c] - blo[c]) * inv[c], 0.0f), 1023.0f);
          code |= expand_bits10((uint32_t)qv) << (2 - c);
        }
        big = oc[3];
        parked = big < 0.0f;
#ifdef AGX_BVH_EXPERIMENT
        if (g_obj_codes) { code = g_obj_codes[(size_t)env * (nt / ppo) + f / ppo]; big = 0.0f; }
#endif
      } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float cen = (t[c] + t[3 + c] + t[6 + c]) * (1.0f / 3.0f);
          float qv = fminf(fmaxf((cen - blo[c]) * inv[c], 0.0f), 1023.0f);
          code |= expand_bits10((uint32_t)qv) << (2 - c);
          big = fmaxf(big, fmaxf(fmaxf(t[c], t[3 + c]), t[6 + c]) - fminf(fminf(t[c], t[3 + c]), t[6 + c]));
        }
        parked = tri_parked(t);
      }
      // primitives larger than most of the obstacle field (the room's wall slabs) get their own subtree under
      // the root: their huge boxes no longer inflate every level of the obstacle tree
      if (big > kBvhLargeFraction * max_ext) code |= 1u << 30;
      if (parked) code = 0xFFFFFFFEu;  // one subtree at the end of the order, culled at its root
      // The three finest Morton bits (sub-centimetre cells) are replaced by the triangle's FACE CLASS (dominant axis
      // and sign of its normal): inside an object (same code) the two triangles of a box face then share the whole
      // upper word and become sibling leaves, which the emit stage folds into one two-triangle leaf -- half the
      // in-object nodes.  A collision of classes only shapes the tree; hits never depend on it.  (The index stays
      // alone in the low word: hipcc dropped an `& 0xFFFFFF` on the LDS read when the class lived there.)
      if (ppo > 0 && !parked) {
        V3 e1 = V3{t[3] - t[0], t[4] - t[1], t[5] - t[2]}, e2 = V3{t[6] - t[0], t[7] - t[1], t[8] - t[2]};
        V3 nrm = cross_plain(e1, e2);
        float ax = fabsf(nrm.x), ay = fabsf(nrm.y), az = fabsf(nrm.z);
        int d = (ax >= ay && ax >= az) ? 0 : (ay >= az ? 1 : 2);
        float comp = d == 0 ? nrm.x : (d == 1 ? nrm.y : nrm.z);
        code = (code & ~7u) | (uint32_t)(2 * d + (comp < 0.0f ? 1 : 0));
      }
      key = ((unsigned long long)code << 32) | (unsigned long long)(uint32_t)f;
    }
    keys[f] = key;
  }
  __syncthreads();
  // --- sort
  if (!presorted) bitonic_sort_lds(keys, npad, tid);
  __syncthreads();
  // --- Karras 2012 radix tree: internal nodes 0..nt-2, leaves nt-1+i (i = sorted position)
  const int n_int = nt - 1;
  for (int i = tid; i < n_int; i += kBvhThreads) {
    int d = (delta_keys(keys, nt, i, i + 1) - delta_keys(keys, nt, i, i - 1)) >= 0 ? 1 : -1;
    int dmin = delta_keys(keys, nt, i, i - d);
    int lmax = 2;
    while (delta_keys(keys, nt, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
      if (delta_keys(keys, nt, i, i + (l + t) * d) > dmin) l += t;
    int j = i + l * d;
    int dnode = delta_keys(keys, nt, i, j);
    int s = 0;
    int t = l;
    do {
      t = (t + 1) >> 1;
      if (delta_keys(keys, nt, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    int gamma = i + s * d + min(d, 0);
    int left = (min(i, j) == gamma) ? (n_int + gamma) : gamma;
    int right = (max(i, j) == gamma + 1) ? (n_int + gamma + 1) : (gamma + 1);
    child[2 * i] = left;
    child[2 * i + 1] = right;
    parent[left] = i;
    parent[right] = i;
    counter[i] = j << 2;  // arrival count in the two low bits; the other end of the node's key range above them (emit stage)
  }
  if (tid == 0) parent[0] = -1;
  __syncthreads();
  // --- bottom-up box propagation: the second arriver at a node merges its children
  for (int i = tid; i < nt; i += kBvhThreads) {
    int node = parent[n_int + i];
    while (node >= 0) {
      __threadfence_block();  // release: this thread's box is written before the arrival
      if ((atomicAdd(&counter[node], 1) & 3) == 0) break;
      __threadfence_block();  // acquire: the sibling's box is read after the arrival
      float a[6], b[6];
      node_box(box, keys, tris, n_int, child[2 * node], a);
      node_box(box, keys, tris, n_int, child[2 * node + 1], b);
      float *o = box + (size_t)node * 6;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        o[c] = fminf(a[c], b[c]);
        o[3 + c] = fmaxf(a[3 + c], b[3 + c]);
      }
      node = parent[node];
    }
  }
  __syncthreads();
  // --- AGX_BVH_BOX_OBJECTS: which internal nodes are the root of ONE box object?  (three parallel steps, see box_frame)
  // bit 0 of counter[] takes the verdict (the arrival count is no longer needed; the other end of the node's key range stays above
  // bit 1); objroot[] lives in parent[], dead since the propagation.
  int *objroot = parent;  // [nt / 12]
  const int n_obj = nt / 12;
  if (box_objects) {
    for (int o = tid; o < n_obj; o += kBvhThreads) objroot[o] = -1;
    __syncthreads();
    for (int i = tid + 1; i < n_int; i += kBvhThreads) {  // (never the root: it has no parent to carry the reference)
      const int j = counter[i] >> 2;
      const int first = min(i, j), last = max(i, j);
      if (last - first != 11) continue;
      const int obj = (int)(uint32_t)(keys[first] & 0xFFFFFFFFull) / 12;
      bool same = true;
      for (int r = 1; r < 12; ++r) same = same && ((int)(uint32_t)(keys[first + r] & 0xFFFFFFFFull) / 12 == obj);
      if (same) objroot[obj] = i;  // (one node at most has exactly this range)
    }
    __syncthreads();
    for (int f = tid; f < n_obj * 12; f += kBvhThreads) {
      const int obj = f / 12, q = f - obj * 12;
      if (objroot[obj] < 0) continue;
      const float *t = tris + (size_t)obj * 12 * 9;
      BoxFrame F;
      uint32_t ids = 0u, ids2 = 0u;
      bool ok = box_frame(t, F) && box_triangle_ok(F, t + 9 * q, q, ids);
      // the first triangle of each face also looks at its partner: four distinct corners between them
      const int mate = q == 0 ? 2 : (q == 10 ? 11 : (q == 1 ? 5 : (q == 7 ? 9 : (q == 3 ? 8 : (q == 4 ? 6 : -1)))));
      if (ok && mate >= 0) ok = box_triangle_ok(F, t + 9 * mate, mate, ids2) && __popc(ids | ids2) == 4;
      if (!ok) objroot[obj] = -1;  // (every writer stores the same value)
    }
    __syncthreads();
  }
  for (int i = tid; i < n_int; i += kBvhThreads) counter[i] &= ~3;
  __syncthreads();
  if (box_objects) {
    for (int o = tid; o < n_obj; o += kBvhThreads)
      if (objroot[o] >= 0) counter[objroot[o]] |= 1;
    __syncthreads();
  }
  // --- emit nodes with both child boxes inline.  A child whose own two children are both leaves is folded into a
  // two-triangle leaf (first triangle in the child slot, second in the pad slot); the folded node is never visited.
  float *out = nodes + (size_t)env * (nt - 1) * 16;
  for (int i = tid; i < n_int; i += kBvhThreads) {
    int ref[2], second[2];
    float bx[2][6];
#pragma unroll
    for (int side = 0; side < 2; ++side) {
      int c = child[2 * i + side];
      node_box(box, keys, tris, n_int, c, bx[side]);
      second[side] = -1;
      if (c >= n_int) {
        ref[side] = ~(int)(uint32_t)(keys[c - n_int] & 0xFFFFFFFFull);
      } else if (counter[c] & 1) {
        ref[side] = c | AGX_BVH_OBJECT_REF;  // the child is an OBJECT NODE: the tree ends at the box
      } else if (child[2 * c] >= n_int && child[2 * c + 1] >= n_int) {
        ref[side] = ~(int)(uint32_t)(keys[child[2 * c] - n_int] & 0xFFFFFFFFull);
        second[side] = (int)(uint32_t)(keys[child[2 * c + 1] - n_int] & 0xFFFFFFFFull);
      } else {
        ref[side] = c;
      }
    }
    float *o = out + (size_t)i * 16;
    if (counter[i] & 1) {  // this node IS an object node (the verdict its parent reads, too): the box's frame is its record
      const int j = counter[i] >> 2;
      box_object_record(tris, (int)(uint32_t)(keys[min(i, j)] & 0xFFFFFFFFull) / 12, o);
      continue;
    }
    const float *a = bx[0], *b = bx[1];
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = __int_as_float(ref[0]);
    o[4] = a[3]; o[5] = a[4]; o[6] = a[5]; o[7] = __int_as_float(ref[1]);
    o[8] = b[0]; o[9] = b[1]; o[10] = b[2]; o[11] = __int_as_float(second[0]);
    o[12] = b[3]; o[13] = b[4]; o[14] = b[5]; o[15] = __int_as_float(second[1]);
  }
}

// The 100+ KB LDS footprint lets one workgroup live on a CU, so a grid of one workgroup per env costs
// ~5 us of scheduling per env even when the env has nothing to rebuild (167 us per step at 8192 envs
// with 2 % dirty envs).  Instead: one small workgroup compacts the dirty env ids into a work list and a
// CU-sized persistent grid pulls envs from it through an atomic cursor (perfect balance, no idle slots).
//   work[0] = number of dirty envs, work[1] = cursor, work[2 ...] = env ids
__global__ void __launch_bounds__(1024) k_compact_mask(int n, const uint8_t *__restrict__ mask, int32_t *__restrict__ work) {
  __shared__ int count;
  if (threadIdx.x == 0) count = 0;
  __syncthreads();
  for (int env = threadIdx.x; env < n; env += blockDim.x)
    if (mask[env]) work[2 + atomicAdd(&count, 1)] = env;
  __syncthreads();
  if (threadIdx.x == 0) {
    work[0] = count;
    work[1] = 0;
  }
}

template <bool OBJ>
__global__ void __launch_bounds__(kBvhThreads, 4) k_bvh_build(int n, int nt, int npad, int ppo, const float *__restrict__ tri_world,
                                                            int32_t *__restrict__ work, float *__restrict__ nodes) {
  if (!work) {  // every env
    for (int env = blockIdx.x; env < n; env += gridDim.x) {
      bvh_build_env<OBJ>(env, nt, npad, ppo, tri_world, nodes);
      __syncthreads();  // LDS is reused by the next env
    }
    return;
  }
  __shared__ int next;
  const int count = work[0];
  while (true) {
    if (threadIdx.x == 0) next = atomicAdd(&work[1], 1);
    __syncthreads();
    const int idx = next;
    __syncthreads();  // everybody has read `next` before thread 0 overwrites it; also fences LDS reuse
    if (idx >= count) break;
    bvh_build_env<OBJ>(work[2 + idx], nt, npad, ppo, tri_world, nodes);
  }
}

// The masked refresh of a navigation step in ONE persistent launch: the workgroup that pulls a dirty env from the work list
// writes that env's world-frame triangles and collision boxes itself, then builds its tree.  As three launches the two small
// ones cost a dispatch over ALL envs each (40 960 and 3 392 workgroups at 8192 envs x 106 obstacles that look at the mask and
// leave: 22 + 9 us per step for a few dozen dirty envs).  Same device functions, same arithmetic.
// the asset reset of a dirty env in the refresh launch itself (agx_scene_reset_refresh; device generator only)
struct SceneResetArgs {
  int enabled, num_obstacles, num_keep;
  AgxEnvBuffers B;
  AgxResetArgs R;
  const float *min_ratio, *max_ratio;
};

template <bool OBJ>
AGX_DEV void scene_refresh_env(int env, int n, int nt, int npad, int ppo, int na, const float *__restrict__ tri_local,
                               const int32_t *__restrict__ tri_asset, float *asset_state, const float *__restrict__ half_extents,
                               float *tri_world, float *__restrict__ boxes, float *__restrict__ nodes, const SceneResetArgs &S) {
  AGX_PHASE(0);
  if (S.enabled) {  // AssetManager.reset_idx for this env first: the poses the triangles are about to be moved to
    for (int a = threadIdx.x; a < na; a += kBvhThreads)
      reset_asset_one(S.B, S.R, env, a, na, nullptr, nullptr, nullptr, S.min_ratio, S.max_ratio, S.num_obstacles, S.num_keep, asset_state);
    __syncthreads();  // (workgroup scope: the poses are read back by this workgroup only)
  }
  AGX_PHASE(1);
  for (int f = threadIdx.x; f < nt; f += kBvhThreads) transform_triangle(env, f, nt, na, tri_local, tri_asset, asset_state, tri_world);
  AGX_PHASE(2);
  if (boxes)
    for (int k = threadIdx.x; k < na; k += kBvhThreads) box_from_asset(env, k, n, na, asset_state, half_extents, boxes);
  __syncthreads();  // the env's triangles are in memory (workgroup scope) before the build reads them
  AGX_PHASE(3);
  bvh_build_env<OBJ>(env, nt, npad, ppo, tri_world, nodes);
}

template <bool OBJ>
__global__ void __launch_bounds__(kBvhThreads, 4) k_scene_refresh(int n, int nt, int npad, int ppo, int na, const float *__restrict__ tri_local,
                                                                const int32_t *__restrict__ tri_asset,
                                                                float *asset_state,
                                                                const float *__restrict__ half_extents, float *tri_world,
                                                                float *__restrict__ boxes, int32_t *__restrict__ work,
                                                                float *__restrict__ nodes, const uint8_t *__restrict__ mask, SceneResetArgs S) {
  if (mask) {
    // DIRECT mode (small batches: one workgroup per env, no work list): a clean env's workgroup leaves at once -- the launch that
    // compacted the mask (and the one that reset the assets, S.enabled) are gone from the step, and a step without a dirty env
    // costs one dispatch
    const int env = blockIdx.x;
    if (S.enabled && S.B.reset_flag[S.B.flag_parity] == 0) return;
    if (!mask[env]) return;
    scene_refresh_env<OBJ>(env, n, nt, npad, ppo, na, tri_local, tri_asset, asset_state, half_extents, tri_world, boxes, nodes, S);
    return;
  }
  __shared__ int next;
  const int count = work[0];
  while (true) {
    if (threadIdx.x == 0) next = atomicAdd(&work[1], 1);
    __syncthreads();
    const int idx = next;
    __syncthreads();  // everybody has read `next` before thread 0 overwrites it; also fences LDS reuse
    if (idx >= count) break;
    scene_refresh_env<OBJ>(work[2 + idx], n, nt, npad, ppo, na, tri_local, tri_asset, asset_state, half_extents, tri_world, boxes, nodes, S);
  }
}

static size_t bvh_lds_bytes(int nt, int npad) {
  const size_t box_floats = (size_t)(nt - 1) * 6 > (size_t)6 * kBvhThreads ? (size_t)(nt - 1) * 6 : (size_t)6 * kBvhThreads;
  const size_t tri_level = (size_t)npad * 8 + box_floats * 4 + (size_t)(2 * nt - 1) * 4 + (size_t)(nt - 1) * 8 + (size_t)(nt - 1) * 4 + 64;
  const size_t obj_level = (size_t)obj_lds_bytes_c(nt);
  return tri_level > obj_level ? tri_level : obj_level;
}
// what a launch that is KNOWN to take the object-level build needs (the triangle-level build's 72 KB for 1272 triangles allow two
// workgroups per CU; 41 KB allow three)
static bool object_level_build(int nt, int prims_per_object) {
  const int ppo = prims_per_object & ~(AGX_BVH_FULL_SORT | AGX_BVH_BOX_OBJECTS | AGX_BVH_OBJECT_TREE);
  return (prims_per_object & AGX_BVH_OBJECT_TREE) && (prims_per_object & AGX_BVH_BOX_OBJECTS) && !(prims_per_object & AGX_BVH_FULL_SORT) && ppo == 12 &&
         nt >= 24 && nt / 12 <= kObjMax;
}
static size_t bvh_lds_bytes_for(int nt, int npad, int prims_per_object) {
  return object_level_build(nt, prims_per_object) ? (size_t)obj_lds_bytes_c(nt) : bvh_lds_bytes(nt, npad);
}

}  // namespace agx

using namespace agx;

extern "C" int agx_scene_transform(int n, int nt, int na, const float *tri_local, const int32_t *tri_asset,
                                   const float *asset_state, const uint8_t *mask, float *tri_world, void *stream) {
  AGX_REQUIRE(n > 0 && nt > 0 && na > 0, "bad sizes n=%d nt=%d na=%d", n, nt, na);
  AGX_REQUIRE(tri_local && tri_asset && asset_state && tri_world, "null buffer");
  dim3 grid(blocks_for(nt, 256), n);
  hipLaunchKernelGGL(k_scene_transform, grid, dim3(256), 0, (hipStream_t)stream, n, nt, na, tri_local, tri_asset, asset_state,
                     mask, tri_world);
  return check_launch("agx_scene_transform");
}

extern "C" int agx_boxes_from_assets(int n, int na, const float *asset_state, const float *half_extents, const uint8_t *mask,
                                     float *boxes, void *stream) {
  AGX_REQUIRE(n > 0 && na > 0, "bad sizes");
  AGX_REQUIRE(asset_state && half_extents && boxes, "null buffer");
  dim3 grid(blocks_for(n, 256), na);
  hipLaunchKernelGGL(k_boxes_from_assets, grid, dim3(256), 0, (hipStream_t)stream, n, na, asset_state, half_extents, mask, boxes);
  return check_launch("agx_boxes_from_assets");
}

extern "C" size_t agx_bvh_nodes_bytes(int n, int nt) { return nt > 1 ? (size_t)n * (nt - 1) * 16 * sizeof(float) : 0; }

#ifdef AGX_SCENE_PHASE_CLOCK
extern "C" int agx_debug_phase_clock(unsigned long long *out16) {
  return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_phase_clock), sizeof(unsigned long long) * 16);
}
#endif
#ifdef AGX_BVH_EXPERIMENT
extern "C" int agx_debug_set_obj_codes(const uint32_t *codes) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_obj_codes), &codes, sizeof(codes));
}
#endif

extern "C" int agx_prims_from_assets(int n, int num_prims, int num_assets, const int32_t *prim_asset, const float *asset_state,
                                     const float *local_pos, const float *local_quat, const uint8_t *mask, float *prim_state,
                                     void *stream) {
  AGX_REQUIRE(n > 0 && num_prims > 0 && num_assets > 0, "bad sizes");
  AGX_REQUIRE(prim_asset && asset_state && local_pos && local_quat && prim_state, "null buffer");
  hipLaunchKernelGGL(k_prims_from_assets, dim3(blocks_for(n * num_prims, 256)), dim3(256), 0, (hipStream_t)stream, n, num_prims,
                     num_assets, prim_asset, asset_state, local_pos, local_quat, mask, prim_state);
  return check_launch("agx_prims_from_assets");
}

extern "C" int agx_assets_integrate(int n, int num_assets, float *asset_state, const float *twist, float dt, int k, void *stream) {
  AGX_REQUIRE(n > 0 && num_assets > 0 && k >= 0 && k <= AGX_MAX_SUBSTEPS && dt > 0.0f, "bad arguments");
  AGX_REQUIRE(asset_state && twist, "null buffer");
  const int count = n * num_assets;
  hipLaunchKernelGGL(k_assets_integrate, dim3(blocks_for(count, 256)), dim3(256), 0, (hipStream_t)stream, count, asset_state, twist,
                     dt, k);
  return check_launch("agx_assets_integrate");
}

// argument checks and launch shape shared by the two LBVH builders; `kernel`'s dynamic LDS limit is raised once (*attr_set)
static int bvh_launch_shape(int nt, int prims_per_object, int *npad_out, size_t *lds_out, const void *kernel, bool *attr_set) {
  AGX_REQUIRE(nt >= 2 && nt <= kBvhMaxTris, "num_tris %d outside [2, %d] (LDS-resident LBVH build)", nt, kBvhMaxTris);
  const int ppo = prims_per_object & ~(AGX_BVH_FULL_SORT | AGX_BVH_BOX_OBJECTS | AGX_BVH_OBJECT_TREE);  // (flags: include/aerial_gym_hip.h)
  AGX_REQUIRE(!(prims_per_object & AGX_BVH_BOX_OBJECTS) || ppo == 12, "AGX_BVH_BOX_OBJECTS goes with objects of 12 triangles");
  AGX_REQUIRE(!(prims_per_object & AGX_BVH_OBJECT_TREE) || (prims_per_object & AGX_BVH_BOX_OBJECTS), "AGX_BVH_OBJECT_TREE goes with AGX_BVH_BOX_OBJECTS");
  AGX_REQUIRE(ppo == 0 || (ppo >= 9 && nt % ppo == 0),
              "prims_per_object must be 0 or >= 9 (8 floats of LDS scratch per object) and divide num_tris");
  int npad = 1;
  while (npad < nt) npad <<= 1;
  const size_t lds = bvh_lds_bytes_for(nt, npad, prims_per_object);
  if (!*attr_set) {
    // the kernel also owns 4 bytes of static LDS: the dynamic maximum must leave room for them
    hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    AGX_REQUIRE(e == hipSuccess, "hipFuncSetAttribute(LBVH build): %s", hipGetErrorString(e));
    *attr_set = true;
  }
  AGX_REQUIRE(lds <= 160 * 1024 - 256, "LBVH build needs %zu bytes of LDS (> 160 KiB)", lds);
  *npad_out = npad;
  *lds_out = lds;
  return AGX_OK;
}

// AssetManager's geometry refresh behind a reset (asset_manager.py:51-71 -> warp mesh refit, warp_env ...): world-frame
// triangles, collision boxes and the tree of the envs flagged in `mask`; mask == nullptr: every env (the three stand-alone
// launches).  One C call; with a mask, two launches (compaction + the persistent k_scene_refresh).
extern "C" int agx_reset_assets(const AgxEnvBuffers *B, int n, int K, const AgxResetArgs *R, const float *u1, const float *u2,
                                const float *u_sel, const float *min_ratio, const float *max_ratio, int num_obstacles,
                                int num_keep, float *asset_state, void *stream);

constexpr int kDirectRefreshEnvs = 2048;  // up to here a workgroup per env (clean ones leave at once) beats compaction + a persistent grid

// the masked refresh: direct (one workgroup per env, optionally with the asset reset in it) or compaction + persistent grid
static int scene_refresh_launch(int n, int nt, int na, const float *tri_local, const int32_t *tri_asset, float *asset_state,
                                const float *half_extents, int prims_per_object, const uint8_t *mask, float *tri_world, float *boxes,
                                float *nodes, int32_t *work, const SceneResetArgs *reset, void *stream) {
  static bool attr_set[2] = {false, false};
  const bool obj = object_level_build(nt, prims_per_object);
  auto kernel = obj ? k_scene_refresh<true> : k_scene_refresh<false>;
  int npad = 0;
  size_t lds = 0;
  if (int e = bvh_launch_shape(nt, prims_per_object, &npad, &lds, reinterpret_cast<const void *>(kernel), &attr_set[obj])) return e;
  SceneResetArgs S{};
  if (reset) S = *reset;
  if (n <= kDirectRefreshEnvs) {
    hipLaunchKernelGGL(kernel, dim3(n), dim3(kBvhThreads), lds, (hipStream_t)stream, n, nt, npad, prims_per_object, na, tri_local,
                       tri_asset, asset_state, half_extents, tri_world, boxes, work, nodes, mask, S);
    return check_launch("agx_scene_refresh");
  }
  hipLaunchKernelGGL(k_compact_mask, dim3(1), dim3(1024), 0, (hipStream_t)stream, n, mask, work);
  const int grid = n < 512 ? n : 512;  // two resident workgroups per CU (LDS bound: 72 KB each for T = 1272 triangle-level, 41 KB object-level)
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBvhThreads), lds, (hipStream_t)stream, n, nt, npad, prims_per_object, na, tri_local,
                     tri_asset, asset_state, half_extents, tri_world, boxes, work, nodes, (const uint8_t *)nullptr, S);
  return check_launch("agx_scene_refresh");
}

extern "C" int agx_scene_refresh(int n, int nt, int na, const float *tri_local, const int32_t *tri_asset, const float *asset_state,
                                 const float *half_extents, int prims_per_object, const uint8_t *mask, float *tri_world,
                                 float *boxes, float *nodes, int32_t *work, void *stream) {
  AGX_REQUIRE(n > 0 && nt > 0 && na > 0, "bad sizes n=%d nt=%d na=%d", n, nt, na);
  AGX_REQUIRE(tri_local && tri_asset && asset_state && tri_world && nodes, "null buffer");
  AGX_REQUIRE(!boxes || half_extents, "collision boxes need the half extents");
  if (!mask) {
    if (int e = agx_scene_transform(n, nt, na, tri_local, tri_asset, asset_state, nullptr, tri_world, stream)) return e;
    if (int e = agx_bvh_build(n, nt, prims_per_object, tri_world, nullptr, nodes, work, stream)) return e;
    return boxes ? agx_boxes_from_assets(n, na, asset_state, half_extents, nullptr, boxes, stream) : AGX_OK;
  }
  AGX_REQUIRE(work, "a masked refresh needs the work buffer (int32[num_envs + 2])");
  return scene_refresh_launch(n, nt, na, tri_local, tri_asset, const_cast<float *>(asset_state), half_extents, prims_per_object, mask, tri_world, boxes,
                              nodes, work, nullptr, stream);
}

// AssetManager.reset_idx + the geometry refresh behind it (agx_reset_assets + agx_scene_refresh) for the envs of buf->reset_mask, device
// generator only.  Up to kDirectRefreshEnvs envs: ONE launch (a workgroup per env; a clean env's leaves at once); above: the three
// launches (asset reset, mask compaction, persistent refresh).  Same device functions either way: the same poses, triangles and trees.
extern "C" int agx_scene_reset_refresh(const AgxEnvBuffers *B, int n, int nt, int na, const AgxResetArgs *R, const float *min_ratio,
                                       const float *max_ratio, int num_obstacles, int num_keep, float *asset_state, const float *tri_local,
                                       const int32_t *tri_asset, const float *half_extents, int prims_per_object, float *tri_world,
                                       float *boxes, float *nodes, int32_t *work, void *stream) {
  AGX_REQUIRE(B && R && n > 0 && nt > 0 && na > 0, "bad arguments");
  AGX_REQUIRE(min_ratio && max_ratio && asset_state && tri_local && tri_asset && tri_world && nodes && work, "null buffer");
  AGX_REQUIRE(B->reset_flag && B->reset_mask && B->episode_count, "the device generator needs reset_flag, reset_mask and episode_count");
  AGX_REQUIRE(!R->u_state, "agx_scene_reset_refresh draws from the device generator (strict draws: agx_reset_assets + agx_scene_refresh)");
  AGX_REQUIRE(!boxes || half_extents, "collision boxes need the half extents");
  SceneResetArgs S{1, num_obstacles, num_keep, *B, *R, min_ratio, max_ratio};
  if (n > kDirectRefreshEnvs) {
    if (int e = agx_reset_assets(B, n, na, R, nullptr, nullptr, nullptr, min_ratio, max_ratio, num_obstacles, num_keep, asset_state, stream)) return e;
    return scene_refresh_launch(n, nt, na, tri_local, tri_asset, asset_state, half_extents, prims_per_object, B->reset_mask, tri_world, boxes, nodes,
                                work, nullptr, stream);
  }
  return scene_refresh_launch(n, nt, na, tri_local, tri_asset, asset_state, half_extents, prims_per_object, B->reset_mask, tri_world, boxes, nodes, work,
                              &S, stream);
}

extern "C" int agx_bvh_build(int n, int nt, int prims_per_object, const float *tri_world, const uint8_t *mask, float *nodes,
                             int32_t *work, void *stream) {
  AGX_REQUIRE(n > 0, "bad num_envs");
  AGX_REQUIRE(tri_world && nodes, "null buffer");
  AGX_REQUIRE(!mask || work, "a masked rebuild needs the work buffer (int32[num_envs + 2])");
  static bool attr_set[2] = {false, false};
  const bool obj = object_level_build(nt, prims_per_object);
  auto kernel = obj ? k_bvh_build<true> : k_bvh_build<false>;
  int npad = 0;
  size_t lds = 0;
  if (int e = bvh_launch_shape(nt, prims_per_object, &npad, &lds, reinterpret_cast<const void *>(kernel), &attr_set[obj])) return e;
  if (mask) hipLaunchKernelGGL(k_compact_mask, dim3(1), dim3(1024), 0, (hipStream_t)stream, n, mask, work);
  const int grid = n < 512 ? n : 512;  // two resident workgroups per CU (LDS bound: 72 KB each for T = 1272)
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBvhThreads), lds, (hipStream_t)stream, n, nt, npad,
                     prims_per_object, tri_world, mask ? work : nullptr, nodes);  // (with the AGX_BVH_FULL_SORT bit, if set)
  return check_launch("agx_bvh_build");
}
