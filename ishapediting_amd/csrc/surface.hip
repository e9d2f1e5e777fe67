// Level-set surface of the decoded occupancy volume, on the device.
// Reference: triplane_decoder/visualize.py:100-104 (PyMCubes marching cubes at level 0, third-party, absent here),
// drag_utils.py:300 (Open3D filter_smooth_simple, 10 iterations), meshProcess.py:18-35 (Chamfer distance).
// PyMCubes' triangulation tables are not reproduced; the surface is extracted with MARCHING TETRAHEDRA: every grid cell
// is cut into the 6 tetrahedra around its main diagonal, each tetrahedron emits 0/1/2 triangles (16-case table).  The
// vertex SET is the one marching cubes produces on the 12 cube edges plus the crossings on face / body diagonals; every
// vertex sits on a grid edge at the linear zero crossing, is owned by the edge's lower end point, and is shared by all
// the triangles around it (indexed mesh, watertight inside the volume).
// Pipeline (HBM-bound scans over res^3 voxels, deterministic output order = voxel order):
//   count : per voxel, the 7 owned edges (+x, +y, +xy, +z, +xz, +yz, +xyz) that cross the level -> bit mask; per cell, the
//           number of triangles; per-block sums          -> one-block exclusive scan of the block sums
//   emit  : vertices (block-local scan + block offset), per-voxel vertex offsets, then triangles through the
//           (owner voxel, edge type) -> vertex index lookup.
#include "common.h"
#include "mc_table.h"

namespace {

constexpr int SB_THREADS = 256;
constexpr int SB_ITEMS = 8;                       // voxels per thread
constexpr int SB_TILE = SB_THREADS * SB_ITEMS;    // voxels per workgroup

// cube corner i = (i & 1, (i >> 1) & 1, (i >> 2) & 1) -> (dx, dy, dz); the 6 tetrahedra share the diagonal 0-7
__constant__ unsigned char c_cube_tets[6][4] = {{0, 1, 3, 7}, {0, 3, 2, 7}, {0, 2, 6, 7}, {0, 6, 4, 7}, {0, 4, 5, 7}, {0, 5, 1, 7}};
__constant__ unsigned char c_tet_edges[6][2] = {{0, 1}, {0, 2}, {0, 3}, {1, 2}, {1, 3}, {2, 3}};
__constant__ signed char c_tet_tri[16][6] = {{-1, -1, -1, -1, -1, -1}, {1, 0, 2, -1, -1, -1}, {4, 0, 3, -1, -1, -1}, {1, 4, 2, 1, 3, 4},
                                             {3, 1, 5, -1, -1, -1},   {2, 3, 0, 2, 5, 3},   {1, 4, 0, 1, 5, 4},   {4, 2, 5, -1, -1, -1},
                                             {4, 5, 2, -1, -1, -1},   {4, 1, 0, 4, 5, 1},   {3, 2, 0, 3, 5, 2},   {1, 3, 5, -1, -1, -1},
                                             {4, 1, 2, 4, 3, 1},      {3, 0, 4, -1, -1, -1}, {2, 0, 1, -1, -1, -1}, {-1, -1, -1, -1, -1, -1}};
__constant__ unsigned char c_tet_ntri[16] = {0, 1, 1, 2, 1, 2, 2, 1, 1, 2, 2, 1, 2, 1, 1, 0};

struct SurfArgs {
  const float* vol;
  int res;
  float level;
  long long n;              // res^3
  unsigned char* emask;     // [n] crossing owned edges, bit (d-1) for direction bits d = dx | dy<<1 | dz<<2
  unsigned char* tcnt;      // [n] triangles of the cell whose lower corner is this voxel
  unsigned* voff;           // [n] index of the voxel's first vertex
  unsigned* bsum;           // [2][nblocks] per-block (vertices, triangles) -> exclusive prefix after the scan
  unsigned* counts;         // [2] totals
  int nblocks;
  int method;               // 0: marching tetrahedra (7 edge types per voxel), 1: marching cubes (the 3 axis edges, 256-case table)
};

// inside bits of the 8 corners of the cell at (x, y, z); corners outside the grid read as "same as the voxel itself",
// which makes their edges non-crossing
__device__ __forceinline__ unsigned corner_bits(const SurfArgs& a, int x, int y, int z, bool& cell) {
  const int r = a.res;
  const long long p = ((long long)x * r + y) * r + z;
  const bool in0 = a.vol[p] - a.level > 0.f;
  const bool hx = x + 1 < r, hy = y + 1 < r, hz = z + 1 < r;
  cell = hx && hy && hz;
  unsigned bits = 0;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const bool ok = (!(c & 1) || hx) && (!(c & 2) || hy) && (!(c & 4) || hz);
    bool in = in0;
    if (ok && c) in = a.vol[p + ((long long)(c & 1) * r + ((c >> 1) & 1)) * r + ((c >> 2) & 1)] - a.level > 0.f;
    bits |= (in ? 1u : 0u) << c;
  }
  return bits;
}

__device__ __forceinline__ int cell_triangles(unsigned bits) {
  if (bits == 0u || bits == 255u) return 0;
  int n = 0;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    unsigned code = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) code |= ((bits >> c_cube_tets[k][i]) & 1u) << i;
    n += c_tet_ntri[code];
  }
  return n;
}

// workgroup exclusive scan of one value per thread; returns the prefix and leaves the total in *total
__device__ __forceinline__ unsigned block_exclusive_scan(unsigned v, unsigned* lds, unsigned* total) {
  const int t = threadIdx.x;
  lds[t] = v;
  __syncthreads();
  for (int o = 1; o < SB_THREADS; o <<= 1) {
    unsigned add = t >= o ? lds[t - o] : 0u;
    __syncthreads();
    lds[t] += add;
    __syncthreads();
  }
  const unsigned incl = lds[t];
  *total = lds[SB_THREADS - 1];
  __syncthreads();
  return incl - v;
}

__global__ __launch_bounds__(SB_THREADS) void surf_count_kernel(SurfArgs a) {
  __shared__ unsigned red[2][SB_THREADS];
  const int r = a.res;
  unsigned nv = 0, nt = 0;
  const long long base = (long long)blockIdx.x * SB_TILE + (long long)threadIdx.x * SB_ITEMS;
#pragma unroll
  for (int i = 0; i < SB_ITEMS; ++i) {
    const long long p = base + i;
    if (p >= a.n) break;
    const int z = (int)(p % r), y = (int)((p / r) % r), x = (int)(p / ((long long)r * r));
    bool cell;
    const unsigned bits = corner_bits(a, x, y, z, cell);
    const unsigned in0 = bits & 1u;
    unsigned m = 0;
#pragma unroll
    for (int d = 1; d < 8; ++d) m |= ((((bits >> d) & 1u) ^ in0) & 1u) << (d - 1);   // out-of-grid corners equal in0: never cross
    if (a.method == 1) m &= 0x0Bu;                                                    // marching cubes: only the axis edges d = 1, 2, 4
    const int tc = !cell ? 0 : (a.method == 1 ? (int)c_mc_ntri[bits] : cell_triangles(bits));
    a.emask[p] = (unsigned char)m;
    a.tcnt[p] = (unsigned char)tc;
    nv += __popc(m);
    nt += tc;
  }
  red[0][threadIdx.x] = nv;
  red[1][threadIdx.x] = nt;
  __syncthreads();
  for (int o = SB_THREADS / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { a.bsum[blockIdx.x] = red[0][0]; a.bsum[a.nblocks + blockIdx.x] = red[1][0]; }
}

// one workgroup: exclusive scan of the per-block sums (both arrays), totals -> counts
__global__ __launch_bounds__(1024) void surf_scan_blocks_kernel(SurfArgs a) {
  __shared__ unsigned part[1024];
  for (int arr = 0; arr < 2; ++arr) {
    unsigned* s = a.bsum + (long long)arr * a.nblocks;
    const int per = (a.nblocks + 1023) / 1024;
    const int b0 = threadIdx.x * per, b1 = min(a.nblocks, b0 + per);
    unsigned sum = 0;
    for (int i = b0; i < b1; ++i) sum += s[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      unsigned add = (int)threadIdx.x >= o ? part[threadIdx.x - o] : 0u;
      __syncthreads();
      part[threadIdx.x] += add;
      __syncthreads();
    }
    unsigned run = part[threadIdx.x] - sum;
    if (threadIdx.x == 1023) a.counts[arr] = part[1023];
    for (int i = b0; i < b1; ++i) { const unsigned v = s[i]; s[i] = run; run += v; }
    __syncthreads();
  }
}

__global__ __launch_bounds__(SB_THREADS) void surf_emit_vertices_kernel(SurfArgs a, float* __restrict__ verts) {
  __shared__ unsigned lds[SB_THREADS];
  const int r = a.res;
  const long long base = (long long)blockIdx.x * SB_TILE + (long long)threadIdx.x * SB_ITEMS;
  unsigned cnt = 0;
  unsigned char m[SB_ITEMS];
#pragma unroll
  for (int i = 0; i < SB_ITEMS; ++i) {
    m[i] = base + i < a.n ? a.emask[base + i] : 0;
    cnt += __popc((unsigned)m[i]);
  }
  unsigned total;
  unsigned idx = a.bsum[blockIdx.x] + block_exclusive_scan(cnt, lds, &total);
#pragma unroll
  for (int i = 0; i < SB_ITEMS; ++i) {
    const long long p = base + i;
    if (p >= a.n) break;
    a.voff[p] = idx;
    if (!m[i]) continue;
    const int z = (int)(p % r), y = (int)((p / r) % r), x = (int)(p / ((long long)r * r));
    const float va = a.vol[p] - a.level;
#pragma unroll
    for (int d = 1; d < 8; ++d) {
      if (!((m[i] >> (d - 1)) & 1)) continue;
      const int dx = d & 1, dy = (d >> 1) & 1, dz = (d >> 2) & 1;
      const float vb = a.vol[p + ((long long)dx * r + dy) * r + dz] - a.level;
      const float t = va / (va - vb);
      verts[3LL * idx + 0] = (float)x + t * (float)dx;
      verts[3LL * idx + 1] = (float)y + t * (float)dy;
      verts[3LL * idx + 2] = (float)z + t * (float)dz;
      ++idx;
    }
  }
}

__global__ __launch_bounds__(SB_THREADS) void surf_emit_triangles_kernel(SurfArgs a, int* __restrict__ tris) {
  __shared__ unsigned lds[SB_THREADS];
  const int r = a.res;
  const long long base = (long long)blockIdx.x * SB_TILE + (long long)threadIdx.x * SB_ITEMS;
  unsigned cnt = 0;
#pragma unroll
  for (int i = 0; i < SB_ITEMS; ++i) cnt += base + i < a.n ? a.tcnt[base + i] : 0;
  unsigned total;
  unsigned idx = a.bsum[a.nblocks + blockIdx.x] + block_exclusive_scan(cnt, lds, &total);
  for (int i = 0; i < SB_ITEMS; ++i) {
    const long long p = base + i;
    if (p >= a.n) break;
    if (!a.tcnt[p]) continue;
    const int z = (int)(p % r), y = (int)((p / r) % r), x = (int)(p / ((long long)r * r));
    bool cell;
    const unsigned bits = corner_bits(a, x, y, z, cell);
    if (a.method == 1) {
      // marching cubes: each table entry names a cube edge = (lower corner, axis); its vertex is owned by that corner's
      // voxel under direction bit d = 1 << axis
      const int ntri = c_mc_ntri[bits];
      for (int tr = 0; tr < ntri; ++tr) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const int e = c_mc_tri[bits][3 * tr + c];
          const unsigned lo = c_mc_edge_lo[e], d = 1u << (e >> 2);
          const long long owner = p + ((long long)(lo & 1) * r + ((lo >> 1) & 1)) * r + ((lo >> 2) & 1);
          const unsigned below = (unsigned)a.emask[owner] & ((1u << (d - 1)) - 1u);
          tris[3LL * idx + c] = (int)(a.voff[owner] + __popc(below));
        }
        ++idx;
      }
      continue;
    }
    for (int k = 0; k < 6; ++k) {
      unsigned code = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) code |= ((bits >> c_cube_tets[k][j]) & 1u) << j;
      const int ntri = c_tet_ntri[code];
      for (int tr = 0; tr < ntri; ++tr) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const int e = c_tet_tri[code][3 * tr + c];
          const unsigned ca = c_cube_tets[k][c_tet_edges[e][0]], cb = c_cube_tets[k][c_tet_edges[e][1]];
          const unsigned lo = ca & cb, hi = ca | cb;          // the corners of a tetrahedron form a chain: one contains the other
          const unsigned d = hi ^ lo;                          // direction bits of the edge, 1..7
          const long long owner = p + ((long long)(lo & 1) * r + ((lo >> 1) & 1)) * r + ((lo >> 2) & 1);
          const unsigned below = (unsigned)a.emask[owner] & ((1u << (d - 1)) - 1u);
          tris[3LL * idx + c] = (int)(a.voff[owner] + __popc(below));
        }
        ++idx;
      }
    }
  }
}

// ---- filter_smooth_simple: v <- (v + sum of neighbours) / (1 + #neighbours) -------------------------------------
// Neighbours are listed face by face (each directed edge of a face lists its head under its tail).  Inside a closed
// surface every undirected edge lies in exactly two faces, so each neighbour is listed twice: v <- (v + acc/2) / (1 + cnt/2).
// The lists are built ONCE per call as a CSR adjacency (count -> exclusive scan -> fill; 9 32-bit atomics per triangle in
// total) and every sweep is a gather, one thread per vertex, from the previous sweep's positions -- no atomics in the sweeps
// (round 4; the earlier form added every head to its tail with 64-bit atomics in every sweep: 18 ms per sweep on the 26 M
// triangles of a noise volume, 0.2 ms on a 300 k-triangle sphere).  The order of a vertex's list depends on the fill's
// atomics; the sum does not: it is formed in 64-bit fixed point (integer adds commute), so the result is bitwise
// reproducible and identical to the atomic form's.
constexpr float SMOOTH_SCALE = 1048576.f;     // 2^20: coordinates < 2^11, valence sums < 2^20 -> < 2^51

// which of the six faces of the grid box [0, bmax]^3 a vertex lies on (bit 2*axis: coordinate 0, bit 2*axis+1: bmax)
__global__ void smooth_boundary_mask_kernel(const float* __restrict__ v, long long nverts, float bmax, unsigned* __restrict__ mask) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nverts) return;
  unsigned m = 0;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float x = v[3 * i + c];
    m |= (x <= 0.f ? 1u : 0u) << (2 * c);
    m |= (x >= bmax ? 1u : 0u) << (2 * c + 1);
  }
  mask[i] = m;
}
// list lengths: every face lists two heads under each of its corners
__global__ void smooth_degree_kernel(const int* __restrict__ tris, long long ntris, unsigned* __restrict__ deg) {
  const long long f = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= ntris) return;
#pragma unroll
  for (int e = 0; e < 3; ++e) atomicAdd(deg + tris[3 * f + e], 2u);
}
// exclusive scan of `x` in place, three launches: per-block scan (SCAN_ITEMS per thread) + block totals, scan of the totals
// by one workgroup, add-back; x[n] receives the grand total
constexpr int SCAN_THREADS = 256, SCAN_ITEMS = 8, SCAN_BLOCK = SCAN_THREADS * SCAN_ITEMS;
__device__ __forceinline__ unsigned block_exclusive_scan(unsigned v, unsigned* lds, unsigned& total) {
  // wave-level inclusive scan, then the wave totals through LDS
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  unsigned inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned t = __shfl_up(inc, d);
    if (lane >= d) inc += t;
  }
  if (lane == 63) lds[wave] = inc;
  __syncthreads();
  unsigned base = 0, tot = 0;
  for (int w = 0; w < nw; ++w) {
    const unsigned t = lds[w];
    if (w < wave) base += t;
    tot += t;
  }
  __syncthreads();
  total = tot;
  return base + inc - v;
}
__global__ __launch_bounds__(SCAN_THREADS) void scan_blocks_kernel(unsigned* __restrict__ x, long long n, unsigned* __restrict__ totals) {
  __shared__ unsigned lds[16];
  const long long i0 = (long long)blockIdx.x * SCAN_BLOCK + (long long)threadIdx.x * SCAN_ITEMS;
  unsigned v[SCAN_ITEMS], sum = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) { v[k] = i0 + k < n ? x[i0 + k] : 0u; sum += v[k]; }
  unsigned total;
  unsigned run = block_exclusive_scan(sum, lds, total);
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    if (i0 + k < n) x[i0 + k] = run;
    run += v[k];
  }
  if (threadIdx.x == 0) totals[blockIdx.x] = total;
}
__global__ __launch_bounds__(1024) void scan_totals_kernel(unsigned* __restrict__ totals, long long nb, unsigned* __restrict__ grand) {
  __shared__ unsigned lds[16];
  unsigned carry = 0;
  for (long long b0 = 0; b0 < nb; b0 += 1024) {
    const long long i = b0 + threadIdx.x;
    const unsigned v = i < nb ? totals[i] : 0u;
    unsigned total;
    const unsigned ex = block_exclusive_scan(v, lds, total);
    if (i < nb) totals[i] = carry + ex;
    carry += total;
  }
  if (threadIdx.x == 0) *grand = carry;
}
__global__ __launch_bounds__(SCAN_THREADS) void scan_add_kernel(unsigned* __restrict__ x, long long n, const unsigned* __restrict__ totals) {
  const unsigned add = totals[blockIdx.x];
  const long long i0 = (long long)blockIdx.x * SCAN_BLOCK + (long long)threadIdx.x * SCAN_ITEMS;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k)
    if (i0 + k < n) x[i0 + k] += add;
}
// An edge of a level-set mesh that lies IN a face of the grid box belongs to one triangle only (there is no cell on the
// other side), every other edge to two; the lone occurrence of a boundary edge therefore counts double (bit 31 of its list
// entry), so that after the halving in the sweep each neighbour has weight one -- Open3D's unique-adjacency rule -- on open
// meshes too.
__global__ void smooth_fill_kernel(const int* __restrict__ tris, long long ntris, const unsigned* __restrict__ off,
                                   unsigned* __restrict__ cursor, unsigned* __restrict__ adj, const unsigned* __restrict__ bmask) {
  const long long f = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= ntris) return;
  const int i[3] = {tris[3 * f], tris[3 * f + 1], tris[3 * f + 2]};
  unsigned bm[3] = {0u, 0u, 0u};
  if (bmask) { bm[0] = bmask[i[0]]; bm[1] = bmask[i[1]]; bm[2] = bmask[i[2]]; }
#pragma unroll
  for (int e = 0; e < 3; ++e) {
    const int h1 = (e + 1) % 3, h2 = (e + 2) % 3;
    const unsigned slot = off[i[e]] + atomicAdd(cursor + i[e], 2u);
    adj[slot] = (unsigned)i[h1] | ((bm[e] & bm[h1]) ? 0x80000000u : 0u);
    adj[slot + 1] = (unsigned)i[h2] | ((bm[e] & bm[h2]) ? 0x80000000u : 0u);
  }
}
__global__ void smooth_sweep_kernel(const float* __restrict__ vin, float* __restrict__ vout, long long nverts,
                                    const unsigned* __restrict__ off, const unsigned* __restrict__ adj) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nverts) return;
  const unsigned lo = off[i], hi = off[i + 1];
  long long acc[3] = {0, 0, 0};
  unsigned cnt = 0;
  for (unsigned k = lo; k < hi; ++k) {
    const unsigned e = adj[k];
    const long long h = (long long)(e & 0x7fffffffu);
    const bool dbl = (e >> 31) != 0u;
    const float w = (dbl ? 2.f : 1.f) * SMOOTH_SCALE;
#pragma unroll
    for (int c = 0; c < 3; ++c) acc[c] += __float2ll_rn(vin[3 * h + c] * w);
    cnt += dbl ? 2u : 1u;
  }
  const float half_n = 0.5f * (float)cnt;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float sm = (float)acc[c] * (1.f / SMOOTH_SCALE);
    vout[3 * i + c] = (vin[3 * i + c] + 0.5f * sm) / (1.f + half_n);
  }
}

// ---- Chamfer: for every point of A the squared distance to its nearest point of B (exact differences) ------------
__global__ __launch_bounds__(256) void nearest_sq_kernel(const float* __restrict__ A, long long na, const float* __restrict__ B,
                                                          long long nb, float* __restrict__ out) {
  __shared__ float tile[1024 * 3];
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  float ax = 0.f, ay = 0.f, az = 0.f;
  if (i < na) { ax = A[3 * i]; ay = A[3 * i + 1]; az = A[3 * i + 2]; }
  float best = 3.0e38f;
  for (long long j0 = 0; j0 < nb; j0 += 1024) {
    const int m = (int)min(1024LL, nb - j0);
    __syncthreads();
    for (int k = threadIdx.x; k < m * 3; k += 256) tile[k] = B[3 * j0 + k];
    __syncthreads();
#pragma unroll 8
    for (int k = 0; k < m; ++k) {
      const float dx = ax - tile[3 * k], dy = ay - tile[3 * k + 1], dz = az - tile[3 * k + 2];
      best = fminf(best, dx * dx + dy * dy + dz * dz);
    }
  }
  if (i < na) out[i] = best;
}
// mean of n floats, fixed summation order (one workgroup, double accumulators)
__global__ __launch_bounds__(1024) void mean_kernel(const float* __restrict__ x, long long n, float* __restrict__ out) {
  __shared__ double red[1024];
  double s = 0.0;
  const long long per = (n + 1023) / 1024;
  const long long b0 = threadIdx.x * per, b1 = min(n, b0 + per);
  for (long long i = b0; i < b1; ++i) s += (double)x[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = n > 0 ? (float)(red[0] / (double)n) : 0.f;
}

// ---- occupancy sampling of an input mesh (drag_utils.py:411-440: Open3D sample_points_uniformly + RaycastingScene
//      .compute_occupancy in the reference) ----------------------------------------------------------------------------
__global__ void tri_area_kernel(const float* __restrict__ v, const int* __restrict__ t, long long nt, float* __restrict__ area) {
  const long long f = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= nt) return;
  const float* A = v + 3LL * t[3 * f]; const float* B = v + 3LL * t[3 * f + 1]; const float* C = v + 3LL * t[3 * f + 2];
  const float ux = B[0] - A[0], uy = B[1] - A[1], uz = B[2] - A[2];
  const float wx = C[0] - A[0], wy = C[1] - A[1], wz = C[2] - A[2];
  const float cx = uy * wz - uz * wy, cy = uz * wx - ux * wz, cz = ux * wy - uy * wx;
  area[f] = 0.5f * sqrtf(cx * cx + cy * cy + cz * cz);
}
// uniform point on triangle f = idx[i] from two uniforms (u, w): (1-sqrt(u)) A + sqrt(u)(1-w) B + sqrt(u) w C
__global__ void tri_point_kernel(const float* __restrict__ v, const int* __restrict__ t, const int* __restrict__ idx,
                                 const float* __restrict__ uw, long long n, float* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long long f = idx[i];
  const float* A = v + 3LL * t[3 * f]; const float* B = v + 3LL * t[3 * f + 1]; const float* C = v + 3LL * t[3 * f + 2];
  const float su = sqrtf(uw[2 * i]), w = uw[2 * i + 1];
  const float a = 1.f - su, b = su * (1.f - w), c = su * w;
#pragma unroll
  for (int k = 0; k < 3; ++k) out[3 * i + k] = a * A[k] + b * B[k] + c * C[k];
}
// inside / outside by the parity of the crossings of the ray p + s*(1,0,0), s > 0.  Triangles are staged through LDS
// (256 at a time); a thread owns one point.  A crossing needs the point's (y,z) strictly inside the triangle's (y,z)
// projection, so rays through an edge or a vertex (measure zero for random samples) are not counted.
__global__ __launch_bounds__(256) void occupancy_kernel(const float* __restrict__ v, const int* __restrict__ t, long long nt,
                                                        const float* __restrict__ pts, long long np, float* __restrict__ occ) {
  __shared__ float tri[256][9];
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  float px = 0.f, py = 0.f, pz = 0.f;
  if (i < np) { px = pts[3 * i]; py = pts[3 * i + 1]; pz = pts[3 * i + 2]; }
  unsigned crossings = 0;
  for (long long f0 = 0; f0 < nt; f0 += 256) {
    const int m = (int)min(256LL, nt - f0);
    __syncthreads();
    if ((int)threadIdx.x < m) {
      const long long f = f0 + threadIdx.x;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float* P = v + 3LL * t[3 * f + c];
        tri[threadIdx.x][3 * c] = P[0]; tri[threadIdx.x][3 * c + 1] = P[1]; tri[threadIdx.x][3 * c + 2] = P[2];
      }
    }
    __syncthreads();
    for (int k = 0; k < m; ++k) {
      // barycentric coordinates in the (y,z) plane relative to vertex A: differences of nearby numbers stay accurate
      // however far the point is; a triangle seen edge-on (negligible projected area) cannot be crossed
      const float uy = tri[k][4] - tri[k][1], uz = tri[k][5] - tri[k][2];
      const float wy = tri[k][7] - tri[k][1], wz = tri[k][8] - tri[k][2];
      const float qy = py - tri[k][1], qz = pz - tri[k][2];
      const float D = uy * wz - uz * wy;
      if (fabsf(D) <= 1e-6f * (fabsf(uy) + fabsf(uz)) * (fabsf(wy) + fabsf(wz))) continue;
      const float sB = (qy * wz - qz * wy) / D, sC = (uy * qz - uz * qy) / D;
      if (sB > 0.f && sC > 0.f && sB + sC < 1.f) {
        const float xh = tri[k][0] + sB * (tri[k][3] - tri[k][0]) + sC * (tri[k][6] - tri[k][0]);
        crossings += xh > px ? 1u : 0u;
      }
    }
  }
  if (i < np) occ[i] = (crossings & 1u) ? 1.f : 0.f;
}

int fill(SurfArgs& a, const float* volume, int res, float level, void* scratch, unsigned* counts) {
  ISHAP_REQUIRE(volume && scratch && res >= 2 && res <= 1024, "surface: volume, scratch and 2 <= res <= 1024");
  a.vol = volume; a.res = res; a.level = level; a.n = (long long)res * res * res;
  a.nblocks = (int)((a.n + SB_TILE - 1) / SB_TILE);
  char* s = (char*)scratch;
  a.voff = (unsigned*)s;               s += a.n * sizeof(unsigned);
  a.bsum = (unsigned*)s;               s += 2LL * a.nblocks * sizeof(unsigned);
  a.emask = (unsigned char*)s;         s += a.n;
  a.tcnt = (unsigned char*)s;
  a.counts = counts;
  a.method = 0;
  return 0;
}

}  // namespace

extern "C" long long ishap_surface_scratch_bytes(int res) {
  const long long n = (long long)res * res * res;
  const long long nb = (n + SB_TILE - 1) / SB_TILE;
  return n * 6 + 2 * nb * 4 + 256;
}

extern "C" int ishap_surface_count(const float* volume, int res, float level, int method, void* scratch, unsigned* counts,
                                   void* stream) {
  SurfArgs a;
  ISHAP_TRY(fill(a, volume, res, level, scratch, counts));
  ISHAP_REQUIRE(counts && (method == 0 || method == 1), "method: 0 marching tetrahedra, 1 marching cubes");
  a.method = method;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(surf_count_kernel, dim3(a.nblocks), dim3(SB_THREADS), 0, s, a);
  hipLaunchKernelGGL(surf_scan_blocks_kernel, dim3(1), dim3(1024), 0, s, a);
  ISHAP_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int ishap_surface_emit(const float* volume, int res, float level, int method, void* scratch, float* verts, int* tris,
                                  void* stream) {
  SurfArgs a;
  ISHAP_TRY(fill(a, volume, res, level, scratch, nullptr));
  ISHAP_REQUIRE(verts && tris && (method == 0 || method == 1), "null argument / method");
  a.method = method;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(surf_emit_vertices_kernel, dim3(a.nblocks), dim3(SB_THREADS), 0, s, a, verts);
  hipLaunchKernelGGL(surf_emit_triangles_kernel, dim3(a.nblocks), dim3(SB_THREADS), 0, s, a, tris);
  ISHAP_CHECK_HIP(hipGetLastError());
  return 0;
}

// scratch layout: off[nverts + 1] | cursor[nverts] | block totals | boundary mask[nverts] | second vertex buffer | adj[6 ntris]
namespace {
struct SmoothLayout { long long off, cursor, totals, bmask, vtmp, adj, bytes, nblocks; };
SmoothLayout smooth_layout(long long nverts, long long ntris) {
  auto up = [](long long b) { return (b + 255) / 256 * 256; };
  SmoothLayout L;
  L.nblocks = (nverts + 1 + SCAN_BLOCK - 1) / SCAN_BLOCK;
  L.off = 0;
  L.cursor = up(L.off + 4 * (nverts + 1));
  L.totals = up(L.cursor + 4 * nverts);
  L.bmask = up(L.totals + 4 * (L.nblocks + 1));
  L.vtmp = up(L.bmask + 4 * nverts);
  L.adj = up(L.vtmp + 12 * nverts);
  L.bytes = up(L.adj + 24 * ntris);
  return L;
}
}  // namespace

extern "C" long long ishap_mesh_smooth_scratch_bytes(long long nverts, long long ntris) {
  if (nverts < 0 || ntris < 0) return -1;
  return smooth_layout(nverts, ntris).bytes;
}

extern "C" int ishap_mesh_smooth(float* verts, long long nverts, const int* tris, long long ntris, int iterations, float box_max,
                                 void* scratch, long long scratch_bytes, void* stream) {
  ISHAP_REQUIRE(verts && tris && scratch && nverts >= 0 && ntris >= 0 && iterations >= 0, "mesh_smooth arguments");
  ISHAP_REQUIRE(scratch_bytes >= ishap_mesh_smooth_scratch_bytes(nverts, ntris), "mesh_smooth: scratch smaller than ishap_mesh_smooth_scratch_bytes(nverts, ntris)");
  ISHAP_REQUIRE(nverts < (1ll << 31) && 6 * ntris < (1ll << 32), "mesh_smooth: 32-bit vertex indices and list offsets");
  if (nverts == 0 || ntris == 0 || iterations == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const SmoothLayout L = smooth_layout(nverts, ntris);
  char* base = (char*)scratch;
  unsigned* off = (unsigned*)(base + L.off);
  unsigned* cursor = (unsigned*)(base + L.cursor);
  unsigned* totals = (unsigned*)(base + L.totals);
  unsigned* bmask = box_max > 0.f ? (unsigned*)(base + L.bmask) : nullptr;
  float* vtmp = (float*)(base + L.vtmp);
  unsigned* adj = (unsigned*)(base + L.adj);
  const unsigned tb = (unsigned)((ntris + 255) / 256), vb = (unsigned)((nverts + 255) / 256);
  ISHAP_CHECK_HIP(hipMemsetAsync(base, 0, (size_t)L.totals, s));              // list lengths and fill cursors
  if (bmask) hipLaunchKernelGGL(smooth_boundary_mask_kernel, dim3(vb), dim3(256), 0, s, verts, nverts, box_max, bmask);
  hipLaunchKernelGGL(smooth_degree_kernel, dim3(tb), dim3(256), 0, s, tris, ntris, off);
  hipLaunchKernelGGL(scan_blocks_kernel, dim3((unsigned)L.nblocks), dim3(SCAN_THREADS), 0, s, off, nverts + 1, totals);
  hipLaunchKernelGGL(scan_totals_kernel, dim3(1), dim3(1024), 0, s, totals, L.nblocks, totals + L.nblocks);
  hipLaunchKernelGGL(scan_add_kernel, dim3((unsigned)L.nblocks), dim3(SCAN_THREADS), 0, s, off, nverts + 1, totals);
  hipLaunchKernelGGL(smooth_fill_kernel, dim3(tb), dim3(256), 0, s, tris, ntris, off, cursor, adj, bmask);
  for (int it = 0; it < iterations; ++it) {
    const float* src = (it & 1) ? vtmp : verts;
    float* dst = (it & 1) ? verts : vtmp;
    hipLaunchKernelGGL(smooth_sweep_kernel, dim3(vb), dim3(256), 0, s, src, dst, nverts, off, adj);
  }
  if (iterations & 1) ISHAP_CHECK_HIP(hipMemcpyAsync(verts, vtmp, (size_t)nverts * 12, hipMemcpyDeviceToDevice, s));
  ISHAP_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int ishap_chamfer(const float* a, long long na, const float* b, long long nb, float* nearest, float* out2, void* stream) {
  ISHAP_REQUIRE(a && b && nearest && out2 && na > 0 && nb > 0, "chamfer arguments");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(nearest_sq_kernel, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, s, a, na, b, nb, nearest);
  hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(1024), 0, s, nearest, na, out2);
  hipLaunchKernelGGL(nearest_sq_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, s, b, nb, a, na, nearest);
  hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(1024), 0, s, nearest, nb, out2 + 1);
  ISHAP_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int ishap_mesh_tri_areas(const float* verts, const int* tris, long long ntris, float* areas, void* stream) {
  ISHAP_REQUIRE(verts && tris && areas && ntris > 0, "mesh_tri_areas arguments");
  hipLaunchKernelGGL(tri_area_kernel, dim3((unsigned)((ntris + 255) / 256)), dim3(256), 0, (hipStream_t)stream, verts, tris, ntris, areas);
  ISHAP_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int ishap_mesh_points_on_tris(const float* verts, const int* tris, const int* tri_idx, const float* uw, long long n,
                                         float* pts, void* stream) {
  ISHAP_REQUIRE(verts && tris && tri_idx && uw && pts && n > 0, "mesh_points_on_tris arguments");
  hipLaunchKernelGGL(tri_point_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, verts, tris, tri_idx, uw, n, pts);
  ISHAP_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int ishap_mesh_occupancy(const float* verts, const int* tris, long long ntris, const float* pts, long long npts,
                                    float* occ, void* stream) {
  ISHAP_REQUIRE(verts && tris && pts && occ && ntris > 0 && npts > 0, "mesh_occupancy arguments");
  hipLaunchKernelGGL(occupancy_kernel, dim3((unsigned)((npts + 255) / 256)), dim3(256), 0, (hipStream_t)stream, verts, tris, ntris, pts,
                     npts, occ);
  ISHAP_CHECK_HIP(hipGetLastError());
  return 0;
}
