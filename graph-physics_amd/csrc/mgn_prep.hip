// MI355X (gfx950) kernels either side of the message-passing path: on-device input construction
// (faces -> symmetric coalesced edges, Cartesian / Distance edge features) and the Simulator's
// pre / post processing (one-hot + slice + concat + online normalisers, delta target,
// inverse-normalise + ground-truth re-imposition).  All of it is HBM-bound integer / elementwise
// work: one pass per tensor, coalesced rows, no atomics (two-stage column statistics).
// Second translation unit of libmgn_hip.so; the sort / unique primitives are rocPRIM (header-only).
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>
#include <stdint.h>
#include <stdio.h>
#include "mgn_hip.h"

static thread_local char g_perr[256] = "";
extern "C" const char* mgn_prep_last_error(void) { return g_perr; }
static int pfail(int code, const char* msg) {
  snprintf(g_perr, sizeof(g_perr), "%s", msg);
  return code;
}
static int pcheck(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_perr, sizeof(g_perr), "%s: %s", what, hipGetErrorString(e));
    return 2;
  }
  return 0;
}

// ===================================================================== faces -> edges
// T.FaceToEdge(remove_faces=False) + to_undirected (reference dataset/preprocessing.py:421-424;
// torch-geometric 2.6.1): every pair of corners of a face in both directions, coalesced =
// sorted by (src, dst), duplicates removed; self loops of degenerate faces dropped.
// key = src * N + dst; invalid corner (outside [0,N)) -> err flag, key = all ones (sorts last).
__global__ void k_face_keys(const int64_t* __restrict__ face, int K, long F, long N, uint64_t* __restrict__ keys, int* __restrict__ err) {
  const long f = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F) return;
  int64_t v[4];
  bool ok = true;
  for (int k = 0; k < K; ++k) {
    v[k] = face[(size_t)k * F + f];
    ok = ok && v[k] >= 0 && v[k] < N;
  }
  if (!ok) *err = 1;
  const int npair = K * (K - 1) / 2;
  int p = 0;
  for (int a = 0; a < K; ++a)
    for (int b = a + 1; b < K; ++b, ++p) {
      const bool keep = ok && v[a] != v[b];
      keys[(size_t)p * F + f] = keep ? (uint64_t)v[a] * (uint64_t)N + (uint64_t)v[b] : ~0ull;
      keys[(size_t)(npair + p) * F + f] = keep ? (uint64_t)v[b] * (uint64_t)N + (uint64_t)v[a] : ~0ull;
    }
}

__global__ void k_decode_keys(const uint64_t* __restrict__ uniq, const size_t* __restrict__ n_uniq, long N, long cap,
                              int64_t* __restrict__ src, int64_t* __restrict__ dst, int64_t* __restrict__ n_edges) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  size_t n = *n_uniq;
  if (n > 0 && uniq[n - 1] == ~0ull) --n;  // the dropped pairs collapse into one trailing key
  if (i == 0) *n_edges = (int64_t)n;
  if (i >= cap || (size_t)i >= n) return;
  const uint64_t k = uniq[i];
  src[i] = (int64_t)(k / (uint64_t)N);
  dst[i] = (int64_t)(k % (uint64_t)N);
}

struct F2EPlan {
  size_t keys, sorted, uniq, count, err, tmp, tmp_bytes, total;
};
static F2EPlan f2e_plan(int64_t F, int K) {
  F2EPlan p;
  const size_t n = (size_t)F * K * (K - 1);
  size_t t1 = 0, t2 = 0;
  rocprim::radix_sort_keys(nullptr, t1, (uint64_t*)nullptr, (uint64_t*)nullptr, n, 0, 64, (hipStream_t)0);
  rocprim::unique(nullptr, t2, (uint64_t*)nullptr, (uint64_t*)nullptr, (size_t*)nullptr, n, rocprim::equal_to<uint64_t>(), (hipStream_t)0);
  p.tmp_bytes = (t1 > t2 ? t1 : t2) + 256;
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  p.keys = 0;
  p.sorted = al(p.keys + n * 8);
  p.uniq = al(p.sorted + n * 8);
  p.count = al(p.uniq + n * 8);
  p.err = p.count + 64;
  p.tmp = al(p.err + 64);
  p.total = p.tmp + p.tmp_bytes;
  return p;
}

extern "C" size_t mgn_faces_to_edges_workspace_bytes(int64_t F, int K) {
  if (F < 0 || (K != 3 && K != 4)) return 0;
  return f2e_plan(F, K).total + 256;
}

extern "C" int mgn_faces_to_edges(const int64_t* face, int K, int64_t F, int64_t N, int64_t* src, int64_t* dst,
                                  int64_t* n_edges, void* ws, size_t ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (K != 3 && K != 4) return pfail(1, "mgn_faces_to_edges: faces must have 3 or 4 corners");
  if (F < 0 || N < 1 || N > 3037000499LL) return pfail(1, "mgn_faces_to_edges: size out of range (N*N must fit 63 bits)");
  const F2EPlan p = f2e_plan(F, K);
  const size_t base = ((size_t)ws + 255) & ~(size_t)255;
  if (ws_bytes < (base - (size_t)ws) + p.total) return pfail(1, "mgn_faces_to_edges: workspace too small");
  char* w = (char*)base;
  uint64_t *keys = (uint64_t*)(w + p.keys), *sorted = (uint64_t*)(w + p.sorted), *uniq = (uint64_t*)(w + p.uniq);
  size_t* count = (size_t*)(w + p.count);
  int* err = (int*)(w + p.err);
  const size_t n = (size_t)F * K * (K - 1);
  if (hipMemsetAsync(count, 0, 128, s) != hipSuccess) return pfail(2, "mgn_faces_to_edges: memset");
  if (n == 0) {
    if (hipMemsetAsync(n_edges, 0, sizeof(int64_t), s) != hipSuccess) return pfail(2, "mgn_faces_to_edges: memset");
    return 0;
  }
  hipLaunchKernelGGL(k_face_keys, dim3((unsigned)((F + 255) / 256)), dim3(256), 0, s, face, K, (long)F, (long)N, keys, err);
  size_t tb = p.tmp_bytes;
  if (rocprim::radix_sort_keys(w + p.tmp, tb, keys, sorted, n, 0, 64, s) != hipSuccess) return pfail(2, "mgn_faces_to_edges: sort");
  tb = p.tmp_bytes;
  if (rocprim::unique(w + p.tmp, tb, sorted, uniq, count, n, rocprim::equal_to<uint64_t>(), s) != hipSuccess)
    return pfail(2, "mgn_faces_to_edges: unique");
  hipLaunchKernelGGL(k_decode_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, uniq, count, (long)N, (long)n, src, dst, n_edges);
  if (int rc = pcheck("mgn_faces_to_edges")) return rc;
  int herr = 0;
  if (hipMemcpyAsync(&herr, err, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess) return pfail(2, "mgn_faces_to_edges: memcpy");
  if (hipStreamSynchronize(s) != hipSuccess) return pfail(2, "mgn_faces_to_edges: sync failed");
  if (herr) return pfail(3, "mgn_faces_to_edges: face corner outside [0, N)");
  return 0;
}

// ======================================================================= node renumbering
// Locality order of the nodes of ONE mesh for the message-passing rounds: the gathers Pd[dst], Ps[src] and the
// source-grouped backward scatter read 512-byte node rows through the edge list; with a numbering that follows
// space (neighbours in the mesh = neighbours in memory) those rows come from a few L2-resident lines, with the
// raw numbering of a mesh generator (uniform random points) every row is a fresh HBM request.  The reference has
// no counterpart (PyTorch indexes whatever numbering the dataset has, layers.py:1017-1018); the engine renumbers
// on entry and un-does it on exit (ops.Topology(renumber=...)).
// order[i] = old id of the node at new position i (sorted by Morton key of its position, ties by old id:
// radix sort is stable); rank[old] = new.  Keys: D = 2 -> 2 x 31 bits, D = 3 -> 3 x 21 bits, quantised on the
// bounding box (computed on the device: no host round trip).
__device__ __forceinline__ unsigned f2ord(float f) {  // order-preserving map float -> unsigned (NaN sorts last)
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o); }
__global__ void __launch_bounds__(256) k_bbox(const float* __restrict__ pos, int ld, int D, long N, unsigned* __restrict__ box) {
  unsigned lo[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, hi[3] = {0u, 0u, 0u};
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < N; i += (long)gridDim.x * 256)
    for (int d = 0; d < D; ++d) {
      const float v = pos[i * ld + d];
      if (v == v && fabsf(v) <= 3.0e38f) {
        const unsigned o = f2ord(v);
        lo[d] = o < lo[d] ? o : lo[d];
        hi[d] = o > hi[d] ? o : hi[d];
      }
    }
  for (int d = 0; d < D; ++d) {
    for (int off = 32; off > 0; off >>= 1) {
      const unsigned a = __shfl_xor(lo[d], off), b = __shfl_xor(hi[d], off);
      lo[d] = a < lo[d] ? a : lo[d];
      hi[d] = b > hi[d] ? b : hi[d];
    }
    if ((threadIdx.x & 63) == 0) {
      atomicMin(box + d, lo[d]);
      atomicMax(box + 3 + d, hi[d]);
    }
  }
}
__device__ __forceinline__ uint64_t spread2(uint64_t v) {  // 31 bits -> every second bit
  v &= 0x7fffffffull;
  v = (v | (v << 16)) & 0x0000ffff0000ffffull;
  v = (v | (v << 8)) & 0x00ff00ff00ff00ffull;
  v = (v | (v << 4)) & 0x0f0f0f0f0f0f0f0full;
  v = (v | (v << 2)) & 0x3333333333333333ull;
  v = (v | (v << 1)) & 0x5555555555555555ull;
  return v;
}
__device__ __forceinline__ uint64_t spread3(uint64_t v) {  // 21 bits -> every third bit
  v &= 0x1fffffull;
  v = (v | (v << 32)) & 0x001f00000000ffffull;
  v = (v | (v << 16)) & 0x001f0000ff0000ffull;
  v = (v | (v << 8)) & 0x100f00f00f00f00full;
  v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
  v = (v | (v << 2)) & 0x1249249249249249ull;
  return v;
}
__global__ void __launch_bounds__(256) k_morton_keys(const float* __restrict__ pos, int ld, int D, long N, const unsigned* __restrict__ box,
                                                    uint64_t* __restrict__ keys, int32_t* __restrict__ ids) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const int bits = (D == 2) ? 31 : 21;
  const double top = (double)((1ull << bits) - 1);
  uint64_t q[3] = {0, 0, 0};
  for (int d = 0; d < D; ++d) {
    const float lo = ord2f(box[d]), hi = ord2f(box[3 + d]);
    const float v = pos[i * ld + d];
    double t = 0.0;
    if (v == v && hi > lo) t = ((double)v - (double)lo) / ((double)hi - (double)lo);
    t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
    q[d] = (uint64_t)(t * top);
  }
  keys[i] = (D == 2) ? (spread2(q[0]) | (spread2(q[1]) << 1)) : (spread3(q[0]) | (spread3(q[1]) << 1) | (spread3(q[2]) << 2));
  ids[i] = (int32_t)i;
}
__global__ void __launch_bounds__(256) k_inverse_perm(const int32_t* __restrict__ order, long N, int32_t* __restrict__ rank) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < N) rank[order[i]] = (int32_t)i;
}
struct MortonPlan {
  size_t keys, keys2, ids, box, tmp, tmp_bytes, total;
};
static MortonPlan morton_plan(int64_t N) {
  MortonPlan p;
  size_t t = 0;
  rocprim::radix_sort_pairs(nullptr, t, (uint64_t*)nullptr, (uint64_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr, (size_t)N, 0, 64, (hipStream_t)0);
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  p.tmp_bytes = t + 256;
  p.keys = 0;
  p.keys2 = al((size_t)N * 8);
  p.ids = al(p.keys2 + (size_t)N * 8);
  p.box = al(p.ids + (size_t)N * 4);
  p.tmp = al(p.box + 64);
  p.total = p.tmp + p.tmp_bytes;
  return p;
}
extern "C" size_t mgn_morton_order_workspace_bytes(int64_t N) { return N < 0 ? 0 : morton_plan(N).total + 256; }
extern "C" int mgn_morton_order(const float* pos, int ld, int D, int64_t N, int32_t* order, int32_t* rank, void* ws, size_t ws_bytes,
                                void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (pos == nullptr || order == nullptr || rank == nullptr || (D != 2 && D != 3) || ld < D || N < 0 || N > 2147483647LL)
    return pfail(1, "mgn_morton_order: bad arguments (positions with 2 or 3 coordinates)");
  if (N == 0) return 0;
  const MortonPlan p = morton_plan(N);
  const size_t base = ((size_t)ws + 255) & ~(size_t)255;
  if (ws == nullptr || ws_bytes < (base - (size_t)ws) + p.total) return pfail(1, "mgn_morton_order: workspace too small");
  char* w = (char*)base;
  uint64_t *keys = (uint64_t*)(w + p.keys), *keys2 = (uint64_t*)(w + p.keys2);
  int32_t* ids = (int32_t*)(w + p.ids);
  unsigned* box = (unsigned*)(w + p.box);
  if (hipMemsetAsync(box, 0xff, 12, s) != hipSuccess || hipMemsetAsync(box + 3, 0, 12, s) != hipSuccess)
    return pfail(2, "mgn_morton_order: memset");
  const unsigned nb = (unsigned)((N + 255) / 256);
  hipLaunchKernelGGL(k_bbox, dim3(nb < 1024 ? nb : 1024), dim3(256), 0, s, pos, ld, D, (long)N, box);
  hipLaunchKernelGGL(k_morton_keys, dim3(nb), dim3(256), 0, s, pos, ld, D, (long)N, box, keys, ids);
  size_t tb = p.tmp_bytes;
  if (rocprim::radix_sort_pairs(w + p.tmp, tb, keys, keys2, ids, order, (size_t)N, 0, 64, s) != hipSuccess)
    return pfail(2, "mgn_morton_order: sort");
  hipLaunchKernelGGL(k_inverse_perm, dim3(nb), dim3(256), 0, s, order, (long)N, rank);
  return pcheck("mgn_morton_order");
}

// ======================================================================= world edges
// add_world_edges of the reference (dataset/preprocessing.py:92-140): all node pairs whose WORLD
// positions are within `radius` (scipy cKDTree.query_pairs: Euclidean distance <= r, evaluated in
// double on the float32 coordinates) with one end OBSTACLE and the other NORMAL, added to the mesh
// edges in both directions and coalesced.  The candidate set is (#OBSTACLE x #NORMAL), a few 1e5
// pairs on the plate meshes, so this is a tiled all-pairs kernel: NORMAL nodes stream through LDS,
// every thread owns one node i and tests it against the tile when it is an OBSTACLE.  Matches are
// appended through one atomic counter (order irrelevant: the keys are sorted afterwards).
#define WE_TILE 256
template <int D>
__global__ void __launch_bounds__(WE_TILE) k_world_pairs(const float* __restrict__ x, int x_w, int pos0, int type_idx, long N, double r2,
                                                        uint64_t* __restrict__ keys, unsigned long long* __restrict__ count, unsigned long long cap) {
  __shared__ double tp[WE_TILE][D];
  __shared__ int tt[WE_TILE];
  const long i = (long)blockIdx.x * WE_TILE + threadIdx.x;
  double pi[D];
  int ti = -1;
  if (i < N) {
    ti = (int)(long)x[i * x_w + type_idx];
#pragma unroll
    for (int d = 0; d < D; ++d) pi[d] = (double)x[i * x_w + pos0 + d];
  }
  for (long j0 = 0; j0 < N; j0 += WE_TILE) {
    const long j = j0 + threadIdx.x;
    __syncthreads();
    tt[threadIdx.x] = (j < N) ? (int)(long)x[j * x_w + type_idx] : -1;
#pragma unroll
    for (int d = 0; d < D; ++d) tp[threadIdx.x][d] = (j < N) ? (double)x[j * x_w + pos0 + d] : 0.0;
    __syncthreads();
    if (ti == MGN_NODE_OBSTACLE) {
      for (int k = 0; k < WE_TILE; ++k) {
        if (tt[k] != MGN_NODE_NORMAL) continue;
        double s = 0.0;
#pragma unroll
        for (int d = 0; d < D; ++d) {
          const double df = pi[d] - tp[k][d];
          s += df * df;
        }
        if (s <= r2) {
          const unsigned long long at = atomicAdd(count, 2ull);
          if (at + 1 < cap) {
            const uint64_t jj = (uint64_t)(j0 + k);
            keys[at] = (uint64_t)i * (uint64_t)N + jj;
            keys[at + 1] = jj * (uint64_t)N + (uint64_t)i;
          }
        }
      }
    }
  }
}

__global__ void k_edge_keys(const int64_t* __restrict__ src, const int64_t* __restrict__ dst, long E, long N, uint64_t* __restrict__ keys, int* __restrict__ err) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int64_t a = src[e], b = dst[e];
  if (a < 0 || a >= N || b < 0 || b >= N) {
    *err = 1;
    keys[e] = keys[E + e] = ~0ull;
    return;
  }
  keys[e] = (uint64_t)a * (uint64_t)N + (uint64_t)b;      // to_undirected: both directions
  keys[E + e] = (uint64_t)b * (uint64_t)N + (uint64_t)a;
}

// ws layout: [keys cap][sorted cap][uniq cap][count, err][rocPRIM temp]
struct WEPlan {
  size_t keys, sorted, uniq, count, tmp, tmp_bytes, total;
};
static WEPlan we_plan(size_t cap) {
  WEPlan p;
  size_t t1 = 0, t2 = 0;
  rocprim::radix_sort_keys(nullptr, t1, (uint64_t*)nullptr, (uint64_t*)nullptr, cap, 0, 64, (hipStream_t)0);
  rocprim::unique(nullptr, t2, (uint64_t*)nullptr, (uint64_t*)nullptr, (size_t*)nullptr, cap, rocprim::equal_to<uint64_t>(), (hipStream_t)0);
  p.tmp_bytes = (t1 > t2 ? t1 : t2) + 256;
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  p.keys = 0;
  p.sorted = al(cap * 8);
  p.uniq = al(p.sorted + cap * 8);
  p.count = al(p.uniq + cap * 8);
  p.tmp = al(p.count + 256);
  p.total = p.tmp + p.tmp_bytes;
  return p;
}

extern "C" size_t mgn_world_edges_workspace_bytes(int64_t E, int64_t max_world_pairs) {
  if (E < 0 || max_world_pairs < 0) return 0;
  return we_plan(2 * (size_t)E + 2 * (size_t)max_world_pairs + 1).total + 256;
}

extern "C" int mgn_add_world_edges(const float* x, int x_w, int pos_start, int D, int type_idx, int64_t N, double radius,
                                   const int64_t* src_in, const int64_t* dst_in, int64_t E, int64_t max_world_pairs,
                                   int64_t* src, int64_t* dst, int64_t* n_edges, void* ws, size_t ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (D != 2 && D != 3) return pfail(1, "mgn_add_world_edges: world positions must be 2-D or 3-D");
  if (N < 1 || N > 3037000499LL || E < 0 || max_world_pairs < 0) return pfail(1, "mgn_add_world_edges: size out of range");
  const size_t cap = 2 * (size_t)E + 2 * (size_t)max_world_pairs + 1;
  const WEPlan p = we_plan(cap);
  const size_t base = ((size_t)ws + 255) & ~(size_t)255;
  if (ws_bytes < (base - (size_t)ws) + p.total) return pfail(1, "mgn_add_world_edges: workspace too small");
  char* w = (char*)base;
  uint64_t *keys = (uint64_t*)(w + p.keys), *sorted = (uint64_t*)(w + p.sorted), *uniq = (uint64_t*)(w + p.uniq);
  unsigned long long* count = (unsigned long long*)(w + p.count);  // [0] world keys appended, [1] unique count (size_t), [2] err
  size_t* n_uniq = (size_t*)(count + 1);
  int* err = (int*)(count + 2);
  if (hipMemsetAsync(count, 0, 64, s) != hipSuccess) return pfail(2, "mgn_add_world_edges: memset");
  if (hipMemsetAsync(keys, 0xff, cap * 8, s) != hipSuccess) return pfail(2, "mgn_add_world_edges: memset");  // unused slots sort last
  if (E > 0) hipLaunchKernelGGL(k_edge_keys, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, s, src_in, dst_in, (long)E, (long)N, keys, err);
  const unsigned grid = (unsigned)((N + WE_TILE - 1) / WE_TILE);
  const unsigned long long wcap = 2ull * (unsigned long long)max_world_pairs;
  if (D == 2)
    hipLaunchKernelGGL(k_world_pairs<2>, dim3(grid), dim3(WE_TILE), 0, s, x, x_w, pos_start, type_idx, (long)N, radius * radius, keys + 2 * E, count, wcap);
  else
    hipLaunchKernelGGL(k_world_pairs<3>, dim3(grid), dim3(WE_TILE), 0, s, x, x_w, pos_start, type_idx, (long)N, radius * radius, keys + 2 * E, count, wcap);
  size_t tb = p.tmp_bytes;
  if (rocprim::radix_sort_keys(w + p.tmp, tb, keys, sorted, cap, 0, 64, s) != hipSuccess) return pfail(2, "mgn_add_world_edges: sort");
  tb = p.tmp_bytes;
  if (rocprim::unique(w + p.tmp, tb, sorted, uniq, n_uniq, cap, rocprim::equal_to<uint64_t>(), s) != hipSuccess)
    return pfail(2, "mgn_add_world_edges: unique");
  hipLaunchKernelGGL(k_decode_keys, dim3((unsigned)((cap + 255) / 256)), dim3(256), 0, s, uniq, n_uniq, (long)N, (long)cap, src, dst, n_edges);
  if (int rc = pcheck("mgn_add_world_edges")) return rc;
  unsigned long long h[3] = {0, 0, 0};
  if (hipMemcpyAsync(h, count, sizeof(h), hipMemcpyDeviceToHost, s) != hipSuccess) return pfail(2, "mgn_add_world_edges: memcpy");
  if (hipStreamSynchronize(s) != hipSuccess) return pfail(2, "mgn_add_world_edges: sync failed");
  if (*(int*)&h[2]) return pfail(3, "mgn_add_world_edges: edge index outside [0, N)");
  if (h[0] > wcap) return pfail(4, "mgn_add_world_edges: more world pairs than max_world_pairs");
  return 0;
}

// ==================================================================== edge features
// T.Cartesian(norm=False) then T.Distance(norm=False) (preprocessing.py:16-23):
//   edge_attr[e] = [ pos[src] - pos[dst] (D values), || pos[dst] - pos[src] ||_2 ]
template <int D>
__global__ void k_edge_features(const float* __restrict__ pos, const int64_t* __restrict__ src, const int64_t* __restrict__ dst,
                                long E, float* __restrict__ out) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const float* ps = pos + src[e] * D;
  const float* pd = pos + dst[e] * D;
  float ss = 0.f;
  float* o = out + e * (D + 1);
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const float c = ps[d] - pd[d];
    const float r = pd[d] - ps[d];
    o[d] = c;
    ss = __fadd_rn(ss, __fmul_rn(r, r));  // plain mul / add like torch.norm's sum of squares (no fma contraction)
  }
  o[D] = sqrtf(ss);
}

extern "C" int mgn_edge_features(const float* pos, int D, const int64_t* src, const int64_t* dst, int64_t E, float* edge_attr, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (D != 2 && D != 3) return pfail(1, "mgn_edge_features: positions must be 2-D or 3-D");
  if (E <= 0) return 0;
  const unsigned grid = (unsigned)((E + 255) / 256);
  if (D == 2)
    hipLaunchKernelGGL(k_edge_features<2>, dim3(grid), dim3(256), 0, s, pos, src, dst, (long)E, edge_attr);
  else
    hipLaunchKernelGGL(k_edge_features<3>, dim3(grid), dim3(256), 0, s, pos, src, dst, (long)E, edge_attr);
  return pcheck("mgn_edge_features");
}

// ========================================================= Simulator pre / post (N1)
// Reference: Simulator._build_input_graph / build_outputs (models/simulator.py:112-191) and
// Normalizer (models/layers.py:331-391).  One description struct for the three normalised
// streams of a step:
//   node features  v = cat[x[:, fs:fe], one_hot(x[:, type_idx], 9)]        width Wn = (fe-fs) + 9
//   target delta   v = y[:, 0:O] - x[:, os:oe]                              width O
//   edge features  v = edge_attr                                            width We
// Statistics (training, while the normaliser still accumulates): column sums and sums of squares
// of v, two-stage and atomics-free: k_stats_partial writes one partial per workgroup and column,
// k_stats_final adds them in a fixed order into the running buffers (acc_sum, acc_sum_squared,
// acc_count, num_accumulations) -- before the normalisation, as Normalizer.forward does.
// Normalisation:  (v - mean) / max(std, eps),  mean = sum / max(count, 1),
//                 std = sqrt(max(sumsq / max(count,1) - mean^2, 0)).
#define SIM_MAXW 32
#define SIM_PART 256  // partial workgroups per stream

__device__ __forceinline__ float sim_value(const mgn_sim_desc& d, int stream, long row, int col) {
  if (stream == 0) {
    const int nf = d.feat_end - d.feat_start;
    if (col < nf) return d.x[row * d.x_w + d.feat_start + col];
    const float t = d.x[row * d.x_w + d.type_idx];
    const int ti = (int)(long)t;
    if (col == nf && d.type_err != nullptr && (ti < 0 || ti >= MGN_NODE_TYPES)) *d.type_err = 1;  // F.one_hot raises
    return (ti == col - nf) ? 1.f : 0.f;  // one_hot(node_type.long(), 9)
  }
  if (stream == 1) return d.y[row * d.y_w + col] - d.x[row * d.x_w + d.out_start + col];
  return d.edge_attr[row * d.edge_w + col];
}
__device__ __forceinline__ int sim_width(const mgn_sim_desc& d, int stream) {
  return stream == 0 ? (d.feat_end - d.feat_start) + MGN_NODE_TYPES : stream == 1 ? d.out_w : d.edge_w;
}
__device__ __forceinline__ long sim_rows(const mgn_sim_desc& d, int stream) { return stream == 2 ? d.E : d.N; }

// grid (SIM_PART, 3): workgroup b of stream s sums rows b, b+SIM_PART*R, ... ; thread = (row lane, column)
__global__ void __launch_bounds__(256) k_stats_partial(const mgn_sim_desc d, float* __restrict__ part) {
  const int stream = blockIdx.y;
  if (!d.accumulate[stream] || !(*d.num_acc[stream] < d.max_accumulations)) return;
  const int W = sim_width(d, stream);
  const long M = sim_rows(d, stream);
  __shared__ float s1[256], s2[256];
  const int col = threadIdx.x % SIM_MAXW, rl = threadIdx.x / SIM_MAXW;  // 32 columns x 8 row lanes
  float a = 0.f, b = 0.f;
  if (col < W)
    for (long r = (long)blockIdx.x * 8 + rl; r < M; r += (long)SIM_PART * 8) {
      const float v = sim_value(d, stream, r, col);
      a += v;
      b = fmaf(v, v, b);
    }
  s1[threadIdx.x] = a;
  s2[threadIdx.x] = b;
  __syncthreads();
  if (rl == 0) {
    for (int k = 1; k < 8; ++k) {
      a += s1[k * SIM_MAXW + col];
      b += s2[k * SIM_MAXW + col];
    }
    float* p = part + ((size_t)(stream * SIM_PART + blockIdx.x) * 2) * SIM_MAXW;
    p[col] = a;
    p[SIM_MAXW + col] = b;
  }
}

__global__ void __launch_bounds__(64) k_stats_final(const mgn_sim_desc d, const float* __restrict__ part) {
  const int stream = blockIdx.x;
  // one wave: every lane reads the counter before lane 0 bumps it at the end
  if (!d.accumulate[stream] || !(*d.num_acc[stream] < d.max_accumulations)) return;
  const int W = sim_width(d, stream);
  const int col = threadIdx.x % SIM_MAXW, which = threadIdx.x / SIM_MAXW;  // 0: sum, 1: sum of squares
  if (col < W) {
    float t = 0.f;
    for (int b = 0; b < SIM_PART; ++b) t += part[((size_t)(stream * SIM_PART + b) * 2 + which) * SIM_MAXW + col];
    float* dst = which == 0 ? d.acc_sum[stream] : d.acc_sumsq[stream];
    dst[col] += t;
  }
  if (threadIdx.x == 0) {
    *d.acc_count[stream] += (float)sim_rows(d, stream);
    *d.num_acc[stream] += 1.f;
  }
}

// grid.y = stream; one thread per output element
__global__ void __launch_bounds__(256) k_sim_normalize(const mgn_sim_desc d) {
  const int stream = blockIdx.y;
  const int W = sim_width(d, stream);
  const long M = sim_rows(d, stream);
  float* out = stream == 0 ? d.node_out : stream == 1 ? d.target_out : d.edge_out;
  if (out == nullptr) return;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= M * W) return;
  const long row = i / W;
  const int col = (int)(i % W);
  const float cnt = fmaxf(*d.acc_count[stream], 1.0f);
  const float mean = d.acc_sum[stream][col] / cnt;
  const float var = d.acc_sumsq[stream][col] / cnt - mean * mean;
  const float sd = fmaxf(sqrtf(fmaxf(var, 0.f)), d.std_eps);
  out[i] = (sim_value(d, stream, row, col) - mean) / sd;
}

extern "C" size_t mgn_sim_workspace_bytes(void) { return (size_t)3 * SIM_PART * 2 * SIM_MAXW * sizeof(float); }

extern "C" int mgn_sim_pre(const mgn_sim_desc* desc, void* ws, size_t ws_bytes, void* stream) {
  const mgn_sim_desc& d = *desc;
  hipStream_t s = (hipStream_t)stream;
  const int Wn = (d.feat_end - d.feat_start) + MGN_NODE_TYPES;
  if (d.feat_end < d.feat_start || Wn > SIM_MAXW || d.out_w < 1 || d.out_w > SIM_MAXW || d.edge_w > SIM_MAXW)
    return pfail(1, "mgn_sim_pre: feature widths out of range (<= 32 columns per stream)");
  if (d.N < 0 || d.E < 0 || d.x == nullptr) return pfail(1, "mgn_sim_pre: bad arguments");
  if (d.feat_start < 0 || d.feat_end > d.x_w || d.type_idx < 0 || d.type_idx >= d.x_w)
    return pfail(1, "mgn_sim_pre: feature / node-type columns outside x");
  if (d.y != nullptr && (d.out_start < 0 || d.out_start + d.out_w > d.x_w || d.out_w > d.y_w))
    return pfail(1, "mgn_sim_pre: output columns outside x / y");
  if (Wn != d.norm_w[0]) return pfail(1, "mgn_sim_pre: node feature width differs from the node normaliser's");
  if (d.y != nullptr && d.out_w != d.norm_w[1]) return pfail(1, "mgn_sim_pre: output width differs from the output normaliser's");
  if (d.edge_attr != nullptr && d.edge_w != d.norm_w[2]) return pfail(1, "mgn_sim_pre: edge feature width differs from the edge normaliser's");
  if (d.target_out != nullptr && d.y == nullptr) return pfail(1, "mgn_sim_pre: target requested without y");
  if (d.accumulate[0] || d.accumulate[1] || d.accumulate[2]) {
    if (ws_bytes < mgn_sim_workspace_bytes()) return pfail(1, "mgn_sim_pre: workspace too small");
    hipLaunchKernelGGL(k_stats_partial, dim3(SIM_PART, 3), dim3(256), 0, s, d, (float*)ws);
    hipLaunchKernelGGL(k_stats_final, dim3(3), dim3(64), 0, s, d, (const float*)ws);
  }
  long mx = d.N * Wn;
  if (d.N * d.out_w > mx) mx = d.N * d.out_w;
  if (d.E * d.edge_w > mx) mx = d.E * d.edge_w;
  if (mx > 0) hipLaunchKernelGGL(k_sim_normalize, dim3((unsigned)((mx + 255) / 256), 3), dim3(256), 0, s, d);
  return pcheck("mgn_sim_pre");
}

// build_outputs + the rollout's ground-truth re-imposition (simulator.py:186-191,
// lightning_module.py:27-35,375-409):  pred = x[:, os:oe] + net_out * std + mean, and where the
// node type is not NORMAL / OUTFLOW the ground truth y is written instead (mask_truth != 0).
__global__ void __launch_bounds__(256) k_sim_post(const float* __restrict__ x, int x_w, int out_start, int type_idx,
                                                  const float* __restrict__ y, int y_w, const float* __restrict__ net_out, int O,
                                                  const float* __restrict__ acc_sum, const float* __restrict__ acc_sumsq,
                                                  const float* __restrict__ acc_count, float std_eps, int mask_truth,
                                                  long N, float* __restrict__ pred) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= N * O) return;
  const long row = i / O;
  const int col = (int)(i % O);
  const float cnt = fmaxf(*acc_count, 1.0f);
  const float mean = acc_sum[col] / cnt;
  const float var = acc_sumsq[col] / cnt - mean * mean;
  const float sd = fmaxf(sqrtf(fmaxf(var, 0.f)), std_eps);
  float p = x[row * x_w + out_start + col] + (net_out[i] * sd + mean);
  if (mask_truth) {
    const int t = (int)(long)x[row * x_w + type_idx];
    if (!(t == MGN_NODE_NORMAL || t == MGN_NODE_OUTFLOW)) p = y[row * y_w + col];
  }
  pred[i] = p;
}

extern "C" int mgn_sim_post(const float* x, int x_w, int out_start, int type_idx, const float* y, int y_w, const float* net_out,
                            int O, const float* acc_sum, const float* acc_sumsq, const float* acc_count, float std_eps,
                            int mask_truth, int64_t N, float* pred, void* stream) {
  if (O < 1 || O > SIM_MAXW || N < 0) return pfail(1, "mgn_sim_post: bad arguments");
  if (mask_truth && y == nullptr) return pfail(1, "mgn_sim_post: ground-truth re-imposition needs y");
  if (N == 0) return 0;
  hipLaunchKernelGGL(k_sim_post, dim3((unsigned)((N * O + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, x_w, out_start, type_idx, y,
                     y_w, net_out, O, acc_sum, acc_sumsq, acc_count, std_eps, mask_truth, (long)N, pred);
  return pcheck("mgn_sim_post");
}

// =============================================== fused gradient clipping + AdamW (R8)
// The optimiser tail of the reference's training step (Trainer(gradient_clip_val=1.0), train.py:288;
// AdamW(lr, betas=(0.9,0.95), weight_decay=1e-4), training/lightning_module.py:494-511) for ~300
// small tensors: torch needs ~25 multi-tensor launches (norms, clip scaling, 9 x fused AdamW);
// here the tensors travel as kernel arguments (<= 96 per launch) and the whole tail is
//   k_sumsq_partial  per-block sums of squares of every gradient chunk (+ step counter += 1)
//   k_clip_adamw     every block re-adds the partials in a fixed order -> total norm -> clip
//                    coefficient min(1, max_norm / (norm + 1e-6)); grads scaled in place (as
//                    clip_grad_norm_ does), then the AdamW update of its chunk.
// lr and the step counter live on the device so that a hipGraph replay sees their current values.
#define OPT_MAX_T 96
#define OPT_CHUNK 4096  // elements per workgroup
struct OptLaunch {
  int n;                       // tensors in this launch
  int blk0[OPT_MAX_T + 1];     // first workgroup of each tensor (within this launch)
  int part0;                   // index of this launch's first partial
  int n_part_total;            // partials of all launches
  float* p[OPT_MAX_T];
  float* g[OPT_MAX_T];
  float* m[OPT_MAX_T];
  float* v[OPT_MAX_T];
  int len[OPT_MAX_T];
};

__device__ __forceinline__ int opt_find(const OptLaunch& L, int b) {
  int lo = 0, hi = L.n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (L.blk0[mid] <= b) lo = mid; else hi = mid - 1;
  }
  return lo;
}

__global__ void __launch_bounds__(256) k_sumsq_partial(const OptLaunch L, float* __restrict__ part, float* __restrict__ step, int bump) {
  __shared__ float red[256];
  const int t = opt_find(L, blockIdx.x);
  const long i0 = (long)(blockIdx.x - L.blk0[t]) * OPT_CHUNK;
  const float* g = L.g[t];
  const int n = L.len[t];
  float s = 0.f;
  for (long i = i0 + threadIdx.x; i < i0 + OPT_CHUNK && i < n; i += 256) s = fmaf(g[i], g[i], s);
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    part[L.part0 + blockIdx.x] = red[0];
    if (bump && blockIdx.x == 0) *step += 1.f;
  }
}

__global__ void __launch_bounds__(256) k_clip_adamw(const OptLaunch L, const float* __restrict__ part, const float* __restrict__ lr_p,
                                                    const float* __restrict__ step_p, float beta1, float beta2, float eps, float wd,
                                                    float max_norm, float* __restrict__ norm_out) {
  __shared__ float red[256];
  float s = 0.f;
  for (int i = threadIdx.x; i < L.n_part_total; i += 256) s += part[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  const float norm = sqrtf(red[0]);
  if (norm_out != nullptr && blockIdx.x == 0 && threadIdx.x == 0 && L.part0 == 0) *norm_out = norm;
  float coef = 1.f;
  if (max_norm > 0.f) coef = fminf(max_norm / (norm + 1e-6f), 1.f);  // clip_grad_norm_
  const float lr = *lr_p, step = *step_p;
  const float bc1 = 1.f - powf(beta1, step), bc2 = 1.f - powf(beta2, step);
  const float step_size = lr / bc1, inv_sqrt_bc2 = 1.f / sqrtf(bc2);
  const int t = opt_find(L, blockIdx.x);
  const long i0 = (long)(blockIdx.x - L.blk0[t]) * OPT_CHUNK;
  float *p = L.p[t], *g = L.g[t], *m = L.m[t], *v = L.v[t];
  const int n = L.len[t];
  for (long i = i0 + threadIdx.x; i < i0 + OPT_CHUNK && i < n; i += 256) {
    const float gi = g[i] * coef;
    g[i] = gi;
    float pi = p[i] * (1.f - lr * wd);           // decoupled weight decay
    const float mi = beta1 * m[i] + (1.f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    pi -= step_size * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
    p[i] = pi;
  }
}

static int opt_blocks(int64_t n) { return (int)((n + OPT_CHUNK - 1) / OPT_CHUNK) > 0 ? (int)((n + OPT_CHUNK - 1) / OPT_CHUNK) : 1; }

extern "C" size_t mgn_clip_adamw_workspace_bytes(int n, const mgn_opt_tensor* t) {
  size_t blocks = 0;
  for (int i = 0; i < n; ++i) blocks += (size_t)opt_blocks(t[i].n);
  return (blocks + 64) * sizeof(float);
}

extern "C" int mgn_clip_adamw(int n, const mgn_opt_tensor* t, float max_norm, const float* lr, float* step, float beta1, float beta2,
                              float eps, float weight_decay, float* grad_norm_out, void* ws, size_t ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (n < 1 || lr == nullptr || step == nullptr) return pfail(1, "mgn_clip_adamw: bad arguments");
  if (ws_bytes < mgn_clip_adamw_workspace_bytes(n, t)) return pfail(1, "mgn_clip_adamw: workspace too small");
  int total_parts = 0;
  for (int i = 0; i < n; ++i) {
    if (t[i].p == nullptr || t[i].g == nullptr || t[i].m == nullptr || t[i].v == nullptr || t[i].n < 0 || t[i].n > 2147483647LL)
      return pfail(1, "mgn_clip_adamw: null tensor / size out of range");
    total_parts += opt_blocks(t[i].n);
  }
  float* part = (float*)ws;
  for (int pass = 0; pass < 2; ++pass) {
    int part0 = 0;
    for (int i0 = 0; i0 < n; i0 += OPT_MAX_T) {
      OptLaunch L;
      L.n = (n - i0 < OPT_MAX_T) ? n - i0 : OPT_MAX_T;
      int b = 0;
      for (int k = 0; k < L.n; ++k) {
        const mgn_opt_tensor& q = t[i0 + k];
        L.blk0[k] = b;
        L.p[k] = q.p, L.g[k] = q.g, L.m[k] = q.m, L.v[k] = q.v, L.len[k] = (int)q.n;
        b += opt_blocks(q.n);
      }
      L.blk0[L.n] = b;
      L.part0 = part0;
      L.n_part_total = total_parts;
      if (pass == 0)
        hipLaunchKernelGGL(k_sumsq_partial, dim3(b), dim3(256), 0, s, L, part, step, (int)(i0 == 0));
      else
        hipLaunchKernelGGL(k_clip_adamw, dim3(b), dim3(256), 0, s, L, (const float*)part, lr, (const float*)step, beta1, beta2, eps,
                           weight_decay, max_norm, grad_norm_out);
      part0 += b;
    }
  }
  return pcheck("mgn_clip_adamw");
}

// ============================================================== masked mean-squared error (R8)
// loss = mean over the rows whose node type is one of `types`, and over their O columns, of (out - target)^2
// (graphphysics/training/loss.py:70-75 with the masks of lightning_module.py:27-35) -- in torch a dozen elementwise / reduction launches
// forward and as many backward; here two launches forward (per-block partials, then a fixed-order finish that also stores
// 1 / (rows x O) for the backward) and one backward: d_out = g * 2 (out - target) * w / (rows x O).  Deterministic, no atomics.
#define MSE_PART 256
__device__ __forceinline__ float mse_w(float t, float t0, float t1, float t2, float t3) {
  return (t == t0 || t == t1 || t == t2 || t == t3) ? 1.f : 0.f;
}
__global__ void __launch_bounds__(256) k_mse_partial(const float* __restrict__ out, int ldo, const float* __restrict__ tgt, int ldt,
                                                     const float* __restrict__ type, int ldty, long N, int O, float t0, float t1, float t2,
                                                     float t3, float* __restrict__ part) {
  __shared__ float s1[256], s2[256];
  float a = 0.f, c = 0.f;
  for (long n = (long)blockIdx.x * 256 + threadIdx.x; n < N; n += (long)MSE_PART * 256) {
    const float w = mse_w(type[n * ldty], t0, t1, t2, t3);
    if (w != 0.f) {
      for (int o = 0; o < O; ++o) {
        const float d = out[n * ldo + o] - tgt[n * ldt + o];
        a = fmaf(d, d, a);
      }
      c += 1.f;
    }
  }
  s1[threadIdx.x] = a;
  s2[threadIdx.x] = c;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) {
      s1[threadIdx.x] += s1[threadIdx.x + k];
      s2[threadIdx.x] += s2[threadIdx.x + k];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) part[2 * blockIdx.x] = s1[0], part[2 * blockIdx.x + 1] = s2[0];
}
__global__ void __launch_bounds__(256) k_mse_final(const float* __restrict__ part, int O, float* __restrict__ loss, float* __restrict__ inv) {
  __shared__ float s1[256], s2[256];
  s1[threadIdx.x] = part[2 * threadIdx.x];
  s2[threadIdx.x] = part[2 * threadIdx.x + 1];
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) {
      s1[threadIdx.x] += s1[threadIdx.x + k];
      s2[threadIdx.x] += s2[threadIdx.x + k];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float den = s2[0] * (float)O;   // 0 selected rows: 0 / 0 = nan, as torch's mean of an empty selection
    *loss = s1[0] / den;
    *inv = 1.f / den;
  }
}
__global__ void __launch_bounds__(256) k_mse_bwd(const float* __restrict__ out, int ldo, const float* __restrict__ tgt, int ldt,
                                                 const float* __restrict__ type, int ldty, long N, int O, float t0, float t1, float t2, float t3,
                                                 const float* __restrict__ inv, const float* __restrict__ g, float* __restrict__ d_out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= N * O) return;
  const long n = i / O;
  const int o = (int)(i % O);
  const float w = mse_w(type[n * ldty], t0, t1, t2, t3);
  d_out[i] = (w != 0.f) ? (*g) * 2.f * (out[n * ldo + o] - tgt[n * ldt + o]) * (*inv) : 0.f;
}
// types[ntypes <= 4]: the node-type codes that take part; part: 2 * 256 floats of scratch; loss, inv: device scalars
extern "C" int mgn_masked_mse_fwd(const float* out, int ldo, const float* tgt, int ldt, const float* type, int ldty, int64_t N, int O,
                                  const float* types, int ntypes, float* part, float* loss, float* inv, void* stream) {
  if (out == nullptr || tgt == nullptr || type == nullptr || part == nullptr || loss == nullptr || inv == nullptr || N < 0 || O < 1 ||
      ntypes < 1 || ntypes > 4 || types == nullptr)
    return pfail(1, "mgn_masked_mse_fwd: bad arguments (1..4 node types)");
  float t[4];
  for (int k = 0; k < 4; ++k) t[k] = types[k < ntypes ? k : 0];
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_mse_partial, dim3(MSE_PART), dim3(256), 0, s, out, ldo, tgt, ldt, type, ldty, (long)N, O, t[0], t[1], t[2], t[3], part);
  hipLaunchKernelGGL(k_mse_final, dim3(1), dim3(256), 0, s, (const float*)part, O, loss, inv);
  return pcheck("mgn_masked_mse_fwd");
}
extern "C" int mgn_masked_mse_bwd(const float* out, int ldo, const float* tgt, int ldt, const float* type, int ldty, int64_t N, int O,
                                  const float* types, int ntypes, const float* inv, const float* g, float* d_out, void* stream) {
  if (out == nullptr || tgt == nullptr || type == nullptr || inv == nullptr || g == nullptr || d_out == nullptr || N < 0 || O < 1 ||
      ntypes < 1 || ntypes > 4 || types == nullptr)
    return pfail(1, "mgn_masked_mse_bwd: bad arguments (1..4 node types)");
  if (N == 0) return 0;
  float t[4];
  for (int k = 0; k < 4; ++k) t[k] = types[k < ntypes ? k : 0];
  hipLaunchKernelGGL(k_mse_bwd, dim3((unsigned)((N * O + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, ldo, tgt, ldt, type, ldty,
                     (long)N, O, t[0], t[1], t[2], t[3], inv, g, d_out);
  return pcheck("mgn_masked_mse_bwd");
}

// ---- the same tail in TWO launches whatever the number of tensors (376 for the 15-round model: four launches of each kernel above,
// each under-filling the GPU -- 130 us of a 3.3 ms one-mesh step).  What is fixed across steps (parameter / moment pointers, lengths,
// the block -> tensor map) lives in a device table built once (mgn_clip_adamw_table); only the gradient pointers change from step
// to step, and 480 of them fit the kernel-argument block.
#define OPT_MAX_G 480
struct OptStatic {
  float* p;
  float* m;
  float* v;
  int len;
  int blk0;
};
struct OptG {
  int n, total_blocks;
  const OptStatic* tab;
  const int* blk2t;
  float* g[OPT_MAX_G];
};
__global__ void __launch_bounds__(256) k_sumsq_partial_t(const OptG L, float* __restrict__ part, float* __restrict__ step) {
  __shared__ float red[256];
  const int t = L.blk2t[blockIdx.x];
  const OptStatic T = L.tab[t];
  const long i0 = (long)((int)blockIdx.x - T.blk0) * OPT_CHUNK;
  const float* g = L.g[t];
  float s = 0.f;
  for (long i = i0 + threadIdx.x; i < i0 + OPT_CHUNK && i < T.len; i += 256) s = fmaf(g[i], g[i], s);
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    part[blockIdx.x] = red[0];
    if (blockIdx.x == 0) *step += 1.f;
  }
}
__global__ void __launch_bounds__(256) k_clip_adamw_t(const OptG L, const float* __restrict__ part, const float* __restrict__ lr_p,
                                                      const float* __restrict__ step_p, float beta1, float beta2, float eps, float wd,
                                                      float max_norm, float* __restrict__ norm_out) {
  __shared__ float red[256];
  float s = 0.f;
  for (int i = threadIdx.x; i < L.total_blocks; i += 256) s += part[i];   // same partials in the same order as k_clip_adamw: same norm
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  const float norm = sqrtf(red[0]);
  if (norm_out != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *norm_out = norm;
  float coef = 1.f;
  if (max_norm > 0.f) coef = fminf(max_norm / (norm + 1e-6f), 1.f);  // clip_grad_norm_
  const float lr = *lr_p, step = *step_p;
  const float bc1 = 1.f - powf(beta1, step), bc2 = 1.f - powf(beta2, step);
  const float step_size = lr / bc1, inv_sqrt_bc2 = 1.f / sqrtf(bc2);
  const int t = L.blk2t[blockIdx.x];
  const OptStatic T = L.tab[t];
  const long i0 = (long)((int)blockIdx.x - T.blk0) * OPT_CHUNK;
  float *p = T.p, *g = L.g[t], *m = T.m, *v = T.v;
  for (long i = i0 + threadIdx.x; i < i0 + OPT_CHUNK && i < T.len; i += 256) {
    const float gi = g[i] * coef;
    g[i] = gi;
    float pi = p[i] * (1.f - lr * wd);           // decoupled weight decay
    const float mi = beta1 * m[i] + (1.f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    pi -= step_size * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
    p[i] = pi;
  }
}

static size_t opt_table_bytes(int n, const mgn_opt_tensor* t) {
  size_t blocks = 0;
  for (int i = 0; i < n; ++i) blocks += (size_t)opt_blocks(t[i].n);
  return (size_t)n * sizeof(OptStatic) + blocks * sizeof(int);
}
extern "C" size_t mgn_clip_adamw_table_bytes(int n, const mgn_opt_tensor* t) {
  if (n < 1 || n > OPT_MAX_G) return 0;   // 0: too many tensors for the table form -- use mgn_clip_adamw
  return opt_table_bytes(n, t);
}
// builds the device table of the tensors' FIXED fields (p, m, v, n; the g fields are ignored) -- a blocking host-to-device copy,
// once per parameter set, outside any stream capture
extern "C" int mgn_clip_adamw_table(int n, const mgn_opt_tensor* t, void* table, size_t table_bytes) {
  if (n < 1 || n > OPT_MAX_G || table == nullptr) return pfail(1, "mgn_clip_adamw_table: bad arguments (at most 480 tensors)");
  if (table_bytes < opt_table_bytes(n, t)) return pfail(1, "mgn_clip_adamw_table: table too small");
  size_t blocks = 0;
  for (int i = 0; i < n; ++i) {
    if (t[i].p == nullptr || t[i].m == nullptr || t[i].v == nullptr || t[i].n < 0 || t[i].n > 2147483647LL)
      return pfail(1, "mgn_clip_adamw_table: null tensor / size out of range");
    blocks += (size_t)opt_blocks(t[i].n);
  }
  const size_t bytes = (size_t)n * sizeof(OptStatic) + blocks * sizeof(int);
  char* host = (char*)malloc(bytes);
  if (host == nullptr) return pfail(2, "mgn_clip_adamw_table: out of host memory");
  OptStatic* tab = (OptStatic*)host;
  int* b2t = (int*)(host + (size_t)n * sizeof(OptStatic));
  int b = 0;
  for (int i = 0; i < n; ++i) {
    tab[i].p = t[i].p, tab[i].m = t[i].m, tab[i].v = t[i].v, tab[i].len = (int)t[i].n, tab[i].blk0 = b;
    const int nb = opt_blocks(t[i].n);
    for (int k = 0; k < nb; ++k) b2t[b + k] = i;
    b += nb;
  }
  const hipError_t e = hipMemcpy(table, host, bytes, hipMemcpyHostToDevice);
  free(host);
  if (e != hipSuccess) return pfail(2, "mgn_clip_adamw_table: copy failed");
  return 0;
}
// the step itself: g[i] = this step's gradient of tensor i of the table; two launches; ws as for mgn_clip_adamw
extern "C" int mgn_clip_adamw_t(int n, const mgn_opt_tensor* t, const void* table, float max_norm, const float* lr, float* step, float beta1,
                                float beta2, float eps, float weight_decay, float* grad_norm_out, void* ws, size_t ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (n < 1 || n > OPT_MAX_G || table == nullptr || lr == nullptr || step == nullptr) return pfail(1, "mgn_clip_adamw_t: bad arguments");
  if (ws_bytes < mgn_clip_adamw_workspace_bytes(n, t)) return pfail(1, "mgn_clip_adamw_t: workspace too small");
  OptG L;
  L.n = n;
  int b = 0;
  for (int i = 0; i < n; ++i) {
    if (t[i].g == nullptr) return pfail(1, "mgn_clip_adamw_t: null gradient");
    L.g[i] = t[i].g;
    b += opt_blocks(t[i].n);
  }
  L.total_blocks = b;
  L.tab = (const OptStatic*)table;
  L.blk2t = (const int*)((const char*)table + (size_t)n * sizeof(OptStatic));
  float* part = (float*)ws;
  hipLaunchKernelGGL(k_sumsq_partial_t, dim3(b), dim3(256), 0, s, L, part, step);
  hipLaunchKernelGGL(k_clip_adamw_t, dim3(b), dim3(256), 0, s, L, (const float*)part, lr, (const float*)step, beta1, beta2, eps, weight_decay,
                     max_norm, grad_norm_out);
  return pcheck("mgn_clip_adamw_t");
}

// ============================================================== halo exchange (8e)
// The partitioned large-mesh path exchanges one [rows, H] block per round and neighbour: the send
// rows are packed straight from the node kernel's output (mgn_gather_rows) and, in the backward
// pass, the ghost-row gradients that come back are summed into their owners in a fixed order
// (mgn_halo_unpack_add: the send list grouped by node, no atomics => bit-deterministic gradients).
// Both are HBM-bound row copies: 16-byte lanes, H/4 lanes per row.
__global__ void __launch_bounds__(256) k_gather_rows(const float* __restrict__ src, const int32_t* __restrict__ idx, long n, int H,
                                                    float* __restrict__ out) {
  const int lpr = H / 4;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const long r = t / lpr;
  const int l = (int)(t % lpr);
  if (r >= n) return;
  const float4 v = *(const float4*)(src + (size_t)idx[r] * H + 4 * l);
  *(float4*)(out + (size_t)r * H + 4 * l) = v;
}

extern "C" int mgn_gather_rows(const float* src, const int32_t* idx, int64_t n, int H, float* out, void* stream) {
  if (n < 0 || H < 4 || (H & 3)) return pfail(1, "mgn_gather_rows: bad arguments");
  if (n == 0) return 0;
  const long lanes = (long)n * (H / 4);
  hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, idx, (long)n, H, out);
  return pcheck("mgn_gather_rows");
}

// dst[nodes[j], :] += sum_{k = rowptr[j]}^{rowptr[j+1]-1} rows[perm[k], :]     (k ascending: fixed order)
__global__ void __launch_bounds__(256) k_halo_unpack_add(const float* __restrict__ rows, const int32_t* __restrict__ nodes,
                                                        const int32_t* __restrict__ rowptr, const int32_t* __restrict__ perm, long n_nodes,
                                                        int H, float* __restrict__ dst) {
  const int lpr = H / 4;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const long j = t / lpr;
  const int l = (int)(t % lpr);
  if (j >= n_nodes) return;
  float* d = dst + (size_t)nodes[j] * H + 4 * l;
  float4 s = *(float4*)d;
  for (int k = rowptr[j]; k < rowptr[j + 1]; ++k) {
    const float4 v = *(const float4*)(rows + (size_t)perm[k] * H + 4 * l);
    s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
  }
  *(float4*)d = s;
}

extern "C" int mgn_halo_unpack_add(const float* rows, const int32_t* nodes, const int32_t* rowptr, const int32_t* perm, int64_t n_nodes,
                                   int H, float* dst, void* stream) {
  if (n_nodes < 0 || H < 4 || (H & 3)) return pfail(1, "mgn_halo_unpack_add: bad arguments");
  if (n_nodes == 0) return 0;
  const long lanes = (long)n_nodes * (H / 4);
  hipLaunchKernelGGL(k_halo_unpack_add, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rows, nodes, rowptr, perm,
                     (long)n_nodes, H, dst);
  return pcheck("mgn_halo_unpack_add");
}

// ====================================================== GraphNetBlock variants (N3)
// Elementwise / gather stages of the reference's optional block features; the GEMMs around them
// run on the MLP kernels.  HBM-bound: one 16-byte lane per 4 features, rows of H floats.

// ---- sigmoid gate on the aggregate (layers.py:1091-1098):
//   gate = sigmoid(G + phi[n] * gate_pos[j]),  G = gate_proj(x);   agg_out = agg * gate
__device__ __forceinline__ float sigmoid_f(float v) { return 1.0f / (1.0f + expf(-v)); }

__global__ void __launch_bounds__(256) k_gate_fwd(const float* __restrict__ G, const float* __restrict__ phi, const float* __restrict__ gate_pos,
                                                 const float* __restrict__ agg, long n, int H, float* __restrict__ gate_out,
                                                 float* __restrict__ agg_out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n * H) return;
  float logit = G[i];
  if (phi != nullptr && gate_pos != nullptr) logit += phi[i / H] * gate_pos[i % H];
  const float gt = sigmoid_f(logit);
  if (gate_out != nullptr) gate_out[i] = gt;
  agg_out[i] = agg[i] * gt;
}

// dAgg = dAggG * gate;  dG = dAggG * agg * gate * (1 - gate)      (dAgg may alias dAggG)
__global__ void __launch_bounds__(256) k_gate_bwd(const float* dAggG, const float* __restrict__ agg, const float* __restrict__ gate, long n, int H,
                                                 float* dAgg, float* __restrict__ dG) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n * H) return;
  const float d = dAggG[i], gt = gate[i];
  dG[i] = d * agg[i] * gt * (1.0f - gt);
  dAgg[i] = d * gt;
}

extern "C" int mgn_gate_fwd(const float* G, const float* phi, const float* gate_pos, const float* agg, int64_t N, int H, float* gate_out,
                            float* agg_out, void* stream) {
  if (N < 0 || H < 1 || G == nullptr || agg == nullptr || agg_out == nullptr) return pfail(1, "mgn_gate_fwd: bad arguments");
  if (N == 0) return 0;
  hipLaunchKernelGGL(k_gate_fwd, dim3((unsigned)((N * H + 255) / 256)), dim3(256), 0, (hipStream_t)stream, G, phi, gate_pos, agg, (long)N, H,
                     gate_out, agg_out);
  return pcheck("mgn_gate_fwd");
}

extern "C" int mgn_gate_bwd(const float* dAggG, const float* agg, const float* gate, int64_t N, int H, float* dAgg, float* dG, void* stream) {
  if (N < 0 || H < 1 || dAggG == nullptr || agg == nullptr || gate == nullptr || dAgg == nullptr || dG == nullptr)
    return pfail(1, "mgn_gate_bwd: bad arguments");
  if (N == 0) return 0;
  hipLaunchKernelGGL(k_gate_bwd, dim3((unsigned)((N * H + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dAggG, agg, gate, (long)N, H, dAgg, dG);
  return pcheck("mgn_gate_bwd");
}

// ---- relative RoPE on the source features (layers.py:1020-1026,1104-1149).  The first
// rope_dim = 2 * pair_count * axes channels are rotated pairwise: pair i of axis a (channels
// 2*(a*pair_count + i), +1) by theta = (pos[src] - pos[dst])[a] * inv_freq[i]:
//   even' = even cos - odd sin,  odd' = even sin + odd cos ;   the remaining channels pass through.
// One thread per (edge, channel pair).  SIGN = +1 forward, -1 the transpose (rotation by -theta).
__device__ __forceinline__ void rope_angle(const float* __restrict__ pos, int pos_w, const float* __restrict__ inv_freq, int pair_count, int axes,
                                           int s, int d, int pair, float& cs, float& sn, bool& rot) {
  rot = pair < pair_count * axes;
  cs = 1.f, sn = 0.f;
  if (rot) {
    const int axis = pair / pair_count, i = pair % pair_count;
    const float delta = pos[(size_t)s * pos_w + axis] - pos[(size_t)d * pos_w + axis];
    const float theta = delta * inv_freq[i];
    cs = cosf(theta);
    sn = sinf(theta);
  }
}

__global__ void __launch_bounds__(256) k_rope_gather(const float* __restrict__ x, const float* __restrict__ pos, int pos_w,
                                                    const float* __restrict__ inv_freq, int pair_count, int axes, const int32_t* __restrict__ src,
                                                    const int32_t* __restrict__ dst, long E, int H, float* __restrict__ out) {
  const int hp = H / 2;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const long k = t / hp;
  const int pair = (int)(t % hp);
  if (k >= E) return;
  const int s = src[k], d = dst[k];
  float cs, sn;
  bool rot;
  rope_angle(pos, pos_w, inv_freq, pair_count, axes, s, d, pair, cs, sn, rot);
  const float2 v = *(const float2*)(x + (size_t)s * H + 2 * pair);
  float2 o = v;
  if (rot) {
    o.x = v.x * cs - v.y * sn;
    o.y = v.x * sn + v.y * cs;
  }
  *(float2*)(out + (size_t)k * H + 2 * pair) = o;
}

// out[j] = resid[j] + sum over the edges k whose source is j (src-grouped CSR: rowptr_src / perm_src
// index the dst-sorted edge rows) of R_k^T T[k], summed in CSR order: deterministic, atomics-free.
__global__ void __launch_bounds__(256) k_rope_scatter(const float* __restrict__ T, const float* __restrict__ pos, int pos_w,
                                                     const float* __restrict__ inv_freq, int pair_count, int axes,
                                                     const int32_t* __restrict__ src, const int32_t* __restrict__ dst,
                                                     const int32_t* __restrict__ rowptr_src, const int32_t* __restrict__ perm_src, long N, int H,
                                                     const float* __restrict__ resid, float* __restrict__ out) {
  const int hp = H / 2;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const long j = t / hp;
  const int pair = (int)(t % hp);
  if (j >= N) return;
  float2 acc = (resid != nullptr) ? *(const float2*)(resid + (size_t)j * H + 2 * pair) : make_float2(0.f, 0.f);
  for (int q = rowptr_src[j]; q < rowptr_src[j + 1]; ++q) {
    const int k = perm_src[q];
    float cs, sn;
    bool rot;
    rope_angle(pos, pos_w, inv_freq, pair_count, axes, src[k], dst[k], pair, cs, sn, rot);
    const float2 v = *(const float2*)(T + (size_t)k * H + 2 * pair);
    if (rot) {  // transpose of the rotation
      acc.x += v.x * cs + v.y * sn;
      acc.y += -v.x * sn + v.y * cs;
    } else {
      acc.x += v.x;
      acc.y += v.y;
    }
  }
  *(float2*)(out + (size_t)j * H + 2 * pair) = acc;
}

extern "C" int mgn_rope_gather(const float* x, const float* pos, int pos_w, const float* inv_freq, int pair_count, int axes,
                               const int32_t* src, const int32_t* dst, int64_t E, int H, float* out, void* stream) {
  if (E < 0 || H < 2 || (H & 1) || pair_count < 0 || axes < 1 || axes > 3 || pos_w < axes || 2 * pair_count * axes > H)
    return pfail(1, "mgn_rope_gather: bad arguments");
  if (E == 0) return 0;
  hipLaunchKernelGGL(k_rope_gather, dim3((unsigned)((E * (H / 2) + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, pos, pos_w, inv_freq,
                     pair_count, axes, src, dst, (long)E, H, out);
  return pcheck("mgn_rope_gather");
}

extern "C" int mgn_rope_scatter(const float* T, const float* pos, int pos_w, const float* inv_freq, int pair_count, int axes,
                                const int32_t* src, const int32_t* dst, const int32_t* rowptr_src, const int32_t* perm_src, int64_t N, int H,
                                const float* resid, float* out, void* stream) {
  if (N < 0 || H < 2 || (H & 1) || pair_count < 0 || axes < 1 || axes > 3 || pos_w < axes || 2 * pair_count * axes > H)
    return pfail(1, "mgn_rope_scatter: bad arguments");
  if (N == 0) return 0;
  hipLaunchKernelGGL(k_rope_scatter, dim3((unsigned)((N * (H / 2) + 255) / 256)), dim3(256), 0, (hipStream_t)stream, T, pos, pos_w, inv_freq,
                     pair_count, axes, src, dst, rowptr_src, perm_src, (long)N, H, resid, out);
  return pcheck("mgn_rope_scatter");
}

// ================================================================ noise injection (N2)
// add_noise of the reference (dataset/preprocessing.py:177-238): Gaussian noise of standard deviation
// scale[r] on the feature columns [start[r], end[r]) of the NORMAL nodes, in place.  The reference
// draws torch.randn_like; here the stream is COUNTER-BASED so that any implementation can reproduce
// it (the oracle does, in numpy): element (row n, column c, range r) of call `offset` takes
//   (r0, r1, ..) = Philox4x32-10(key = seed, counter = (n_lo, n_hi, c | r << 16, offset))
//   u1 = ((r0 >> 8) + 1) * 2^-24  in (0,1],   u2 = (r1 >> 8) * 2^-24  in [0,1)
//   z  = sqrt(-2 ln u1) * cos(2 pi u2)                                   (Box-Muller)
// One thread per (row, noised column).
#define NOISE_MAX_RANGES 8
struct NoiseRanges {
  int n;
  int start[NOISE_MAX_RANGES], end[NOISE_MAX_RANGES];
  float scale[NOISE_MAX_RANGES];
  int col0[NOISE_MAX_RANGES + 1];  // prefix sums of the range widths
};

__device__ __forceinline__ void philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t& r0,
                                              uint32_t& r1) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1, c3 = (uint32_t)p0, c0 = n0, c2 = n2;
    k0 += 0x9E3779B9u, k1 += 0xBB67AE85u;
  }
  r0 = c0, r1 = c1;
}

__global__ void __launch_bounds__(256) k_add_noise(float* __restrict__ x, int x_w, long N, const NoiseRanges R, int type_idx, uint64_t seed,
                                                  uint32_t offset) {
  const int W = R.col0[R.n];
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const long n = t / W;
  const int j = (int)(t % W);
  if (n >= N) return;
  if (x[n * x_w + type_idx] != (float)MGN_NODE_NORMAL) return;  // noise only on NORMAL nodes; the FLOAT is compared (:219-222,230-231)
  int r = 0;
  while (r + 1 < R.n && j >= R.col0[r + 1]) ++r;
  const int c = R.start[r] + (j - R.col0[r]);
  uint32_t r0, r1;
  philox4x32_10((uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)n, (uint32_t)((uint64_t)n >> 32), (uint32_t)c | ((uint32_t)r << 16), offset, r0, r1);
  const float u1 = (float)((r0 >> 8) + 1u) * 5.9604644775390625e-8f;  // 2^-24
  const float u2 = (float)(r1 >> 8) * 5.9604644775390625e-8f;
  const float z = sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
  x[n * x_w + c] += z * R.scale[r];
}

extern "C" int mgn_add_noise(float* x, int x_w, int64_t N, int n_ranges, const int* starts, const int* ends, const float* scales, int type_idx,
                             uint64_t seed, uint32_t offset, void* stream) {
  if (x == nullptr || N < 0 || n_ranges < 1 || n_ranges > NOISE_MAX_RANGES || type_idx < 0 || type_idx >= x_w)
    return pfail(1, "mgn_add_noise: bad arguments (1..8 ranges)");
  NoiseRanges R;
  R.n = n_ranges;
  R.col0[0] = 0;
  for (int r = 0; r < n_ranges; ++r) {
    if (starts[r] < 0 || ends[r] < starts[r] || ends[r] > x_w || ends[r] > 65535) return pfail(1, "mgn_add_noise: column range outside x");
    R.start[r] = starts[r], R.end[r] = ends[r], R.scale[r] = scales[r];
    R.col0[r + 1] = R.col0[r] + (ends[r] - starts[r]);
  }
  // the node-type column is read by every thread while the noised columns are written: it must not be noised itself
  // (the reference takes its mask BEFORE the loop, :219-222 -- a noised type column has no meaning there either)
  bool overlap = false;
  for (int r = 0; r < n_ranges; ++r) {
    if (type_idx >= starts[r] && type_idx < ends[r]) return pfail(1, "mgn_add_noise: node_type_index lies inside a noised column range");
    for (int q = 0; q < r; ++q) overlap = overlap || (starts[r] < ends[q] && starts[q] < ends[r]);
  }
  if (R.col0[n_ranges] == 0 || N == 0) return 0;
  if (!overlap) {
    const long tot = (long)N * R.col0[n_ranges];
    hipLaunchKernelGGL(k_add_noise, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, x_w, (long)N, R, type_idx, seed, offset);
    return pcheck("mgn_add_noise");
  }
  // overlapping ranges: the reference applies them one after the other (:224-236); one launch per range keeps every
  // `+=` of a launch on a distinct element (no race), the stream orders the launches.  The random stream of range r is
  // keyed by r (counter word 2), so the values equal the single-launch ones.
  for (int r = 0; r < n_ranges; ++r) {
    NoiseRanges R1 = R;
    for (int q = 0; q < n_ranges; ++q)
      if (q != r) R1.end[q] = R1.start[q];                      // empty: every other range contributes no column
    R1.col0[0] = 0;
    for (int q = 0; q < n_ranges; ++q) R1.col0[q + 1] = R1.col0[q] + (R1.end[q] - R1.start[q]);
    const long tot = (long)N * R1.col0[n_ranges];
    if (tot == 0) continue;
    hipLaunchKernelGGL(k_add_noise, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, x_w, (long)N, R1, type_idx, seed, offset);
  }
  return pcheck("mgn_add_noise");
}
