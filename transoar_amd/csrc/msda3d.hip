// MSDeformAttn-3D for MI355X (gfx950): the C ABI of include/transoar_msda3d.h
// and the host-side dispatch onto the kernels in
//   msda3d_gather.hpp   forward + grad_loc/grad_attn (query-stationary gathers)
//   msda3d_scatter.hpp  grad_value (cell sort + voxel-stationary pull, no atomics)
//   msda3d_generic.hpp  any-C / fp64 correctness kernels
//
// What is computed is fixed by the reference (semantics: SURVEY.md appendix A;
// ops/src/cuda/ms_deform_im2col_cuda.cuh:31-114, 116-241, 370-439).  How is
// not: see the kernel headers.  Bound: HBM/L2 bandwidth (18 flop per gathered
// byte); no MFMA in here.
#include <algorithm>
#include <cstring>

#include "../../include/transoar_msda3d.h"
#include "msda3d_common.hpp"
#include "msda3d_brick.hpp"
#include "msda3d_mma.hpp"
#include "msda3d_pcm.hpp"
#include "msda3d_q16.hpp"
#include "msda3d_tile.hpp"
#include "msda3d_cells_mma.hpp"
#include "msda3d_gather.hpp"
#include "msda3d_generic.hpp"
#include "msda3d_scatter.hpp"

#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace transoar {

// ---- optional per-kernel timing with HIP events on the launch stream -------
// bench.py turns this on for the timed region: every kernel launched by the
// two entry points is bracketed by an event pair recorded on `stream`; the
// pairs are resolved later by transoar_msda3d_profile_read (which synchronises
// on them).  Off by default: zero cost.
struct ProfPair { hipEvent_t beg, end; int kind; };
static std::mutex g_prof_mu;
static bool g_prof_on = false;
static std::vector<ProfPair> g_prof_live;   // recorded, not yet read
static std::vector<ProfPair> g_prof_free;   // reusable event pairs

struct ProfScope {
  hipStream_t st;
  ProfPair pair;
  bool on;
  ProfScope(int kind, hipStream_t s) : st(s), on(false) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_prof_on) return;
    if (!g_prof_free.empty()) {
      pair = g_prof_free.back();
      g_prof_free.pop_back();
    } else if (hipEventCreate(&pair.beg) != hipSuccess || hipEventCreate(&pair.end) != hipSuccess) {
      return;
    }
    pair.kind = kind;
    on = hipEventRecord(pair.beg, st) == hipSuccess;
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(pair.end, st);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_live.push_back(pair);
  }
};

// Side stream for the coarse-level walk of grad_value: it and the fine-level tile walk are both
// latency-bound and independent until the final row store, so they are forked onto two streams
// (event fork / join on the caller's stream; captured as two branches under HIP graph capture).
// One pair per device, created on first use (an eager call, before any capture).
struct SideStream {
  hipStream_t stream = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
  bool ok = false;
};
static SideStream* side_stream() {
  static std::mutex mu;
  static SideStream per_device[16];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  std::lock_guard<std::mutex> lk(mu);
  SideStream& s = per_device[dev];
  if (!s.ok && s.stream == nullptr) {
    if (hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) == hipSuccess &&
        hipEventCreateWithFlags(&s.fork, hipEventDisableTiming) == hipSuccess &&
        hipEventCreateWithFlags(&s.join, hipEventDisableTiming) == hipSuccess)
      s.ok = true;
  }
  return s.ok ? &s : nullptr;
}

// zero-fill as a kernel launch (16 bytes per thread; n16 = number of 16-byte pieces)
__global__ __launch_bounds__(256) void msda3d_zero16(u32x4* __restrict__ p, long n16) {
  const long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
  if (i < n16) p[i] = u32x4{0u, 0u, 0u, 0u};
}
static inline hipError_t zero_async(void* p, size_t bytes, hipStream_t st) {
  if (getenv("TRANSOAR_MSDA_MEMSET_NODE")) return hipMemsetAsync(p, 0, bytes, st);
  const long n16 = static_cast<long>((bytes + 15) / 16);           // every region is 16-byte padded
  hipLaunchKernelGGL(msda3d_zero16, dim3(static_cast<unsigned>((n16 + 255) / 256)), dim3(256), 0, st,
                     static_cast<u32x4*>(p), n16);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// host dispatch
// ---------------------------------------------------------------------------
struct Dims { int N, S, M, C, L, Lq, P; };

static inline int lpv_log2(int C, int elt) {
  const long bytes = static_cast<long>(C) * elt;
  if (bytes == 128) return 3;
  if (bytes == 256) return 4;
  if (bytes == 512) return 5;
  return -1;
}

// Brick schedule from host-side level shapes (NULL -> linear order).  `rows` is what the
// kernel's row index ranges over (S for voxel rows; Lq for queries, usable only when Lq == S,
// i.e. the queries ARE the pyramid's voxels as in the refine block's self-attention).
static BrickOrder make_order(const int64_t* host_shapes, const Dims& d, int rows) {
  BrickOrder o{};
  if (!host_shapes || rows != d.S) return o;
  long start = 0, pad = 0;
  for (int l = 0; l < d.L; ++l) {
    const long D = host_shapes[3 * l], H = host_shapes[3 * l + 1], W = host_shapes[3 * l + 2];
    if (D <= 0 || H <= 0 || W <= 0) return BrickOrder{};
    o.D[l] = static_cast<int>(D); o.H[l] = static_cast<int>(H); o.W[l] = static_cast<int>(W);
    o.start[l] = static_cast<int>(start);
    o.nbh[l] = static_cast<int>((H + kBrickH - 1) / kBrickH);
    o.nbw[l] = static_cast<int>((W + kBrickW - 1) / kBrickW);
    o.pad_start[l] = static_cast<int>(pad);
    pad += ((D + kBrickD - 1) / kBrickD) * o.nbh[l] * o.nbw[l] * kBrickSlots;
    start += D * H * W;
  }
  if (start != d.S || pad * d.N * d.M >= (1L << 31)) return BrickOrder{};
  o.pad_start[d.L] = static_cast<int>(pad);
  o.L = d.L;
  o.enabled = 1;
  return o;
}
static inline long order_units(const BrickOrder& o, const Dims& d, int rows) {
  return static_cast<long>(d.N) * d.M * (o.enabled ? o.pad_start[o.L] : rows);
}

static inline size_t align16(size_t x) { return (x + 15) & ~static_cast<size_t>(15); }

// Launch constants of the brick-scheduled kernels (BrickOrder: ~300 bytes; CoarseLevels) live in device memory and
// the kernels read them through a pointer: kernel arguments stay a handful of scalars and pointers, which is what
// replays correctly from a captured HIP graph with ROCm's graph packet capture (large by-value kernel arguments
// are the suspect of the corrupted replays of DESIGN.md section 8).  One immutable copy per distinct content and
// device, made on first use -- a synchronous hipMalloc + hipMemcpy, so first use must be an eager call (the
// training step always runs eagerly before it is captured); returns nullptr if that is not possible.
static const void* device_const(const void* host, size_t bytes, hipStream_t st) {
  static std::mutex mu;
  static std::map<std::pair<int, std::string>, void*> cache;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::pair<int, std::string> key(dev, std::string(static_cast<const char*>(host), bytes));
  std::lock_guard<std::mutex> lk(mu);
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  // not cached yet: making the copy needs hipMalloc + a synchronous hipMemcpy, neither of which may run while `st`
  // is being captured (in global capture mode the hipMalloc alone invalidates the capture): refuse instead
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return nullptr;
  if (cache.size() >= 256) return nullptr;            // bounded: a few entries per (device, pyramid shape) are expected
  void* p = nullptr;
  if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
  if (hipMemcpy(p, host, bytes, hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(p);
    return nullptr;
  }
  cache[key] = p;
  return p;
}
template <typename T> static const T* device_const(const T& v, hipStream_t st) {
  return static_cast<const T*>(device_const(&v, sizeof(T), st));
}

// Can the vector kernels run this problem?  (32-bit byte offsets into value,
// 32-bit bin / point indices.)
static inline int vec_lpv(const Dims& d, int elt, unsigned flags) {
  if (flags & TRANSOAR_MSDA3D_FORCE_GENERIC) return -1;
  const long value_bytes = static_cast<long>(d.N) * d.S * d.M * d.C * elt;
  const long n_points = static_cast<long>(d.N) * d.Lq * d.M * d.L * d.P;
  const long n_bins = static_cast<long>(d.N) * d.M * 8 * d.S;
  if (value_bytes >= 0xfffffff0L || n_points >= (1L << 31) - 1 || n_bins >= (1L << 31) - 1) return -1;
  // row index and row pitch go through the 24-bit multiplier
  if (static_cast<long>(d.N) * d.S >= (1L << 24) || static_cast<long>(d.M) * d.C * elt >= (1L << 24)) return -1;
  if (n_points * 32 >= 0xfffffff0L) return -1;   // record array addressed with 32-bit byte offsets
  if (static_cast<long>(d.N) * d.Lq * d.M >= (1L << 24) ||
      static_cast<long>(d.N) * d.Lq * d.M * d.C * elt >= 0x7ffffff0L) return -1;
  return lpv_log2(d.C, elt);
}

struct BwdWorkspace {
  size_t count, tile_sums, rank, rec_item, recs, coarse, det_keys, det_alt, det_temp, det_temp_bytes, total;
  long n_bins, n_points, n_scan, n_tiles;
};

int sort_keys64(unsigned long long* keys, unsigned long long* alt, long n, int end_bit, void* temp, size_t* temp_bytes,
                unsigned long long** sorted, hipStream_t st);          // msda3d_sort.hip

static BwdWorkspace bwd_workspace(const Dims& d, size_t acc_size, unsigned flags = 0u) {
  BwdWorkspace w;
  w.n_points = static_cast<long>(d.N) * d.Lq * d.M * d.L * d.P;
  w.n_bins = static_cast<long>(d.N) * d.M * 8 * d.S;   // (D+1)(H+1)(W+1) <= 8*D*H*W
  w.n_scan = w.n_bins + 1;
  w.n_tiles = (w.n_scan + kScanTile - 1) / kScanTile;
  size_t off = 0;
  w.count = off;     off += align16(sizeof(int) * w.n_scan);
  w.tile_sums = off; off += align16(sizeof(int) * w.n_tiles);
  w.rank = off;      off += align16(sizeof(int) * w.n_points);
  w.rec_item = off;  off += align16(sizeof(int) * w.n_points);
  w.recs = off;      off += align16(8 * acc_size * w.n_points);   // PointW8 (tile path) or PointRec
  w.coarse = off;    off += align16(sizeof(float) * static_cast<size_t>(d.N) * d.S * d.M * d.C);   // fp32 rows of the coarse levels
  w.det_keys = w.det_alt = w.det_temp = off;
  w.det_temp_bytes = 0;
  if (flags & TRANSOAR_MSDA3D_DETERMINISTIC) {      // two key buffers + the radix sort's scratch (histograms: a bound, checked at run time)
    w.det_keys = off;  off += align16(sizeof(unsigned long long) * w.n_points);
    w.det_alt = off;   off += align16(sizeof(unsigned long long) * w.n_points);
    w.det_temp_bytes = align16((size_t(32) << 20) + static_cast<size_t>(w.n_points));
    w.det_temp = off;  off += w.det_temp_bytes;
  }
  w.total = off;
  return w;
}

// Does the point-column gather (msda3d_pcm.hpp) cover this problem?  Queries = voxels of a <= 4-level pyramid
// known on the host, 64 channels, 4 points, every buffer addressable with 32-bit byte offsets.
static bool pcm_ok(const BrickOrder& order, const Dims& d) {
  if (!order.enabled || d.L > kPcmLevels || d.C != 64 || d.P != 4 || d.Lq != d.S) return false;
  for (int l = 0; l < d.L; ++l)
    if (order.D[l] > 1000 || order.H[l] > 1000 || order.W[l] > 1000) return false;
  const long n_pts = static_cast<long>(d.N) * d.Lq * d.M * d.L * d.P;
  const long n_wave = static_cast<long>(d.N) * (order.pad_start[order.L] >> 7) * d.M * 16;
  const long vbytes = static_cast<long>(d.N) * d.S * d.M * d.C * 2;
  return n_pts * 12 < 0xfffffff0L && n_wave < (1L << 31) - 8 && vbytes < 0xffffff00L;
}
static float host_bf16_round(float x) {
  unsigned u;
  memcpy(&u, &x, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  u &= 0xffff0000u;
  memcpy(&x, &u, 4);
  return x;
}
static PcmConst make_pcm_const(const BrickOrder& order) {
  PcmConst c;
  memset(&c, 0, sizeof(c));
  c.order = order;
  for (int l = 0; l < 4; ++l) {
    const int ls = l < order.L ? l : 0;
    c.fD[l] = static_cast<float>(order.D[ls]); c.fH[l] = static_cast<float>(order.H[ls]); c.fW[l] = static_cast<float>(order.W[ls]);
    c.dD[l] = host_bf16_round(c.fD[l]); c.dH[l] = host_bf16_round(c.fH[l]); c.dW[l] = host_bf16_round(c.fW[l]);
  }
  return c;
}

// Launch constants of msda3d_fwd_q16: the point-column constants + the reciprocals of its unit decode
static Q16Const make_q16_const(const BrickOrder& order, int M) {
  Q16Const c;
  memset(&c, 0, sizeof(c));
  c.pc = make_pcm_const(order);
  auto mg = [](long d) -> unsigned long long { return (1ull << 40) / static_cast<unsigned long long>(d > 0 ? d : 1) + 1ull; };
  c.mg_M = mg(M);
  c.mg_bricks = mg(order.pad_start[order.L] >> 7);
  for (int l = 0; l < 4; ++l) {
    c.mg_nbw[l] = mg(l < order.L ? order.nbw[l] : 1);
    c.mg_nbh[l] = mg(l < order.L ? order.nbh[l] : 1);
  }
  return c;
}
static int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}

template <typename VT, typename LT>
static int launch_fwd(const void* value, const int64_t* shapes, const int64_t* lsi,
                      const void* loc, const void* attn, void* out, const Dims& d,
                      const int64_t* host_shapes, unsigned flags, hipStream_t st) {
  const long n_items = static_cast<long>(d.N) * d.Lq * d.M;
  const BrickOrder order = make_order(host_shapes, d, d.Lq);
  const long n_units = order_units(order, d, d.Lq);
  const long n_blocks = ((order.enabled ? n_units : n_items) + kWavesPerBlock - 1) / kWavesPerBlock;
  const int lg = vec_lpv(d, sizeof(VT), flags);
  const dim3 block(64 * kWavesPerBlock);
  auto v = static_cast<const VT*>(value);
  auto lo = static_cast<const LT*>(loc);
  auto at = static_cast<const LT*>(attn);
  auto o = static_cast<VT*>(out);
  // LDS-tiled brick kernel: queries are the pyramid's voxels (Lq == S), host knows the shapes,
  // 64 channels per head, 4 points per level (the refine block's configuration)
  // (16-bit storage only: an fp32 box of the finest level does not fit the 48 KiB tile)
  if constexpr (sizeof(VT) == 2) {
    // matrix-core gather: one wave per 32 queries (2x4x4 sub-brick) and head
    bool small = d.L <= kMmaLevels;
    for (int l = 0; small && l < d.L; ++l) small = order.D[l] <= 1000 && order.H[l] <= 1000 && order.W[l] <= 1000;
    // point-column form (msda3d_pcm.hpp): one wave per 8 queries (2x2x2 sub-brick) and head; fp32 locations
    if constexpr (sizeof(LT) == 4) {
      // 16 queries per wave (msda3d_q16.hpp, round 6): measured SLOWER than the 8-query kernel below;
      // kept behind its flag as the carving's speed-of-light probe, parity-tested
      if (lg >= 0 && small && pcm_ok(order, d) && (flags & TRANSOAR_MSDA3D_Q16) &&
          !(flags & (TRANSOAR_MSDA3D_NO_BRICK | TRANSOAR_MSDA3D_NO_MMA | TRANSOAR_MSDA3D_MMA_Q32))) {
        ProfScope prof(TRANSOAR_PROF_FWD, st);
        static const int upw_env = env_int("TRANSOAR_MSDA3D_Q16_UPW", 4), probe = env_int("TRANSOAR_MSDA3D_Q16_PROBE", 0);
        const unsigned upw = static_cast<unsigned>(upw_env < 1 ? 1 : (upw_env > 64 ? 64 : upw_env));
        const long n_units = static_cast<long>(d.N) * (order.pad_start[order.L] >> 7) * d.M * 8;
        const long n_waves = (n_units + upw - 1) / upw;
        const unsigned vbytes = static_cast<unsigned>(static_cast<long>(d.N) * d.S * d.M * d.C * sizeof(VT));
        const long n_pts = static_cast<long>(d.N) * d.Lq * d.M * d.L * d.P;
        const Q16Const* cst = device_const(make_q16_const(order, d.M), st);
        if (cst == nullptr) return TRANSOAR_ERR_CONST;
        const dim3 grid(static_cast<unsigned>(((n_waves + 7) / 8) * 8));
#define TRANSOAR_Q16(PROBE)                                                                                             \
  hipLaunchKernelGGL((msda3d_fwd_q16<VT, PROBE>), grid, dim3(64), 0, st, v, lo, at, o, d.S, d.M, d.L, vbytes,               \
                     static_cast<unsigned>(n_pts * 12), static_cast<unsigned>(n_pts * 4), static_cast<unsigned>(n_units), upw, cst)
        if (probe == 1) TRANSOAR_Q16(1);
        else if (probe == 2) TRANSOAR_Q16(2);
        else if (probe == 3) TRANSOAR_Q16(3);
        else TRANSOAR_Q16(0);
#undef TRANSOAR_Q16
        return static_cast<int>(hipGetLastError());
      }
      if (lg >= 0 && small && pcm_ok(order, d) &&
          !(flags & (TRANSOAR_MSDA3D_NO_BRICK | TRANSOAR_MSDA3D_NO_MMA | TRANSOAR_MSDA3D_MMA_Q32))) {
        ProfScope prof(TRANSOAR_PROF_FWD, st);
        const long n_wave = static_cast<long>(d.N) * (order.pad_start[order.L] >> 7) * d.M * 16;
        const unsigned vbytes = static_cast<unsigned>(static_cast<long>(d.N) * d.S * d.M * d.C * sizeof(VT));
        const long n_pts = static_cast<long>(d.N) * d.Lq * d.M * d.L * d.P;
        const PcmConst* cst = device_const(make_pcm_const(order), st);
        if (cst == nullptr) return TRANSOAR_ERR_CONST;
        hipLaunchKernelGGL((msda3d_fwd_pcm<VT, false>), dim3(static_cast<unsigned>(((n_wave + 7) / 8) * 8)), dim3(64), 0, st,
                           v, lo, at, nullptr, nullptr, 0u, o, d.S, d.M, d.L, vbytes, static_cast<unsigned>(n_pts * 12),
                           static_cast<unsigned>(n_pts * 4), static_cast<unsigned>(n_wave), cst);
        return static_cast<int>(hipGetLastError());
      }
    }
    if (lg >= 0 && order.enabled && small && d.C == 64 && d.P == 4 &&
        !(flags & (TRANSOAR_MSDA3D_NO_BRICK | TRANSOAR_MSDA3D_NO_MMA))) {
      ProfScope prof(TRANSOAR_PROF_FWD, st);
      const long n_wave = static_cast<long>(d.N) * (order.pad_start[order.L] >> 7) * d.M * 4;
      const unsigned vbytes = static_cast<unsigned>(static_cast<long>(d.N) * d.S * d.M * d.C * sizeof(VT));
      const BrickOrder* order_d = device_const(order, st);
      if (order_d == nullptr) return TRANSOAR_ERR_CONST;
      hipLaunchKernelGGL((msda3d_fwd_mma<VT, LT>), dim3(static_cast<unsigned>(((n_wave + 7) / 8) * 8)), dim3(64), 0, st,
                         v, lo, at, o, d.S, d.M, d.L, vbytes, n_wave, order_d);
      return static_cast<int>(hipGetLastError());
    }
  }
  if (lg >= 0 && order.enabled && d.C == 64 && d.P == 4 && sizeof(VT) == 2 && !(flags & TRANSOAR_MSDA3D_NO_BRICK)) {
    ProfScope prof(TRANSOAR_PROF_FWD, st);
    const long n_wg = static_cast<long>(d.N) * (order.pad_start[order.L] >> 7) * d.M;
    const dim3 bgrid(((n_wg + 7) / 8) * 8);
    hipLaunchKernelGGL((msda3d_fwd_brick<VT, LT, 4, 64>), bgrid, dim3(kBrickThreads), kTileBytes, st, v, lo, at, o,
                       d.S, d.M, d.L, n_wg, order);
    return static_cast<int>(hipGetLastError());
  }
  if (lg < 0) {
    ProfScope prof(TRANSOAR_PROF_FWD_GENERIC, st);
    hipLaunchKernelGGL((msda3d_fwd_generic<VT, LT>), dim3((n_items + kWavesPerBlock - 1) / kWavesPerBlock),
                       block, 0, st, v, shapes, lsi, lo, at, o, d.S, d.M, d.C, d.L, d.Lq, d.P, n_items);
  } else {
    ProfScope prof(TRANSOAR_PROF_FWD, st);
    const dim3 grid(((n_blocks + 7) / 8) * 8);
    const unsigned vbytes = static_cast<unsigned>(static_cast<long>(d.N) * d.S * d.M * d.C * sizeof(VT));
#define TRANSOAR_FWD(LG)                                                                      \
  hipLaunchKernelGGL((msda3d_fwd_vec<VT, LT, LG>), grid, block, 0, st, v, shapes, lsi, lo, at, \
                     o, d.S, d.M, d.C, d.L, d.Lq, d.P, vbytes, n_units, n_blocks, order)
    if (lg == 3) TRANSOAR_FWD(3);
    else if (lg == 4) TRANSOAR_FWD(4);
    else TRANSOAR_FWD(5);
#undef TRANSOAR_FWD
  }
  return static_cast<int>(hipGetLastError());
}

// forward with the module's sampling head fused into the gather's prologue (msda3d_pcm.hpp, FUSED)
template <typename VT>
static int launch_fwd_fused(const void* value, const void* proj, const float* ref, long ref_rows, void* out, const Dims& d,
                            const int64_t* host_shapes, hipStream_t st) {
  const BrickOrder order = make_order(host_shapes, d, d.Lq);
  if (vec_lpv(d, sizeof(VT), 0) < 0 || !pcm_ok(order, d)) return TRANSOAR_ERR_DIM;
  if (ref_rows != d.Lq && ref_rows != static_cast<long>(d.N) * d.Lq) return TRANSOAR_ERR_DIM;
  ProfScope prof(TRANSOAR_PROF_FWD, st);
  const long n_wave = static_cast<long>(d.N) * (order.pad_start[order.L] >> 7) * d.M * 16;
  const unsigned vbytes = static_cast<unsigned>(static_cast<long>(d.N) * d.S * d.M * d.C * sizeof(VT));
  const PcmConst* cst = device_const(make_pcm_const(order), st);
  if (cst == nullptr) return TRANSOAR_ERR_CONST;
  const unsigned ref_bstride = ref_rows == d.Lq ? 0u : static_cast<unsigned>(d.Lq) * d.L * 12u;
  const long proj_bytes = static_cast<long>(d.N) * d.S * 4 * d.M * d.L * d.P * 2;
  hipLaunchKernelGGL((msda3d_fwd_pcm<VT, true>), dim3(static_cast<unsigned>(((n_wave + 7) / 8) * 8)), dim3(64), 0, st,
                     static_cast<const VT*>(value), nullptr, nullptr, static_cast<const unsigned short*>(proj), ref, ref_bstride,
                     static_cast<VT*>(out), d.S, d.M, d.L, vbytes, static_cast<unsigned>(proj_bytes),
                     static_cast<unsigned>(ref_rows * d.L * 12), static_cast<unsigned>(n_wave), cst);
  return static_cast<int>(hipGetLastError());
}

// deterministic mode: sorted position j takes the record of the canonical slot in the low half of its key
__global__ __launch_bounds__(256) void msda3d_det_gather(const unsigned long long* __restrict__ sorted, const PointR16* __restrict__ slots,
                                                          PointR16* __restrict__ recs, long n) {
  const long j = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
  if (j >= n) return;
  const unsigned long long k = sorted[j];
  if (k == ~0ull) return;
  recs[j] = slots[static_cast<unsigned>(k)];
}

template <typename VT, typename LT>
static size_t bwd_workspace_bytes(const Dims& d, unsigned flags) {
  using A = typename Elem<VT>::acc;
  if (vec_lpv(d, sizeof(VT), flags) >= 0) return bwd_workspace(d, sizeof(A), flags).total;
  // generic path: fp32 accumulator for 16-bit storage
  return sizeof(VT) == 2 ? align16(sizeof(float) * static_cast<size_t>(d.N) * d.S * d.M * d.C) : 0;
}

#define TRANSOAR_CHECK_HIP(expr)                         \
  do {                                                   \
    const hipError_t e_ = (expr);                        \
    if (e_ != hipSuccess) return static_cast<int>(e_);   \
  } while (0)

template <typename VT, typename LT>
static int launch_bwd(const void* value, const int64_t* shapes, const int64_t* lsi,
                      const void* loc, const void* attn, const void* grad_out, void* grad_value,
                      void* grad_loc, void* grad_attn, void* workspace, size_t workspace_bytes,
                      const Dims& d, const int64_t* host_shapes, unsigned flags, hipStream_t st, void* grad_proj = nullptr) {
  using A = typename Elem<VT>::acc;
  if (workspace_bytes < bwd_workspace_bytes<VT, LT>(d, flags)) return TRANSOAR_ERR_WORKSPACE;
  // grad_proj: the sampling head's backward folded into the query kernel (transoar_msda3d_backward_proj): only the
  // matrix-core chain writes it
  if (grad_proj != nullptr && (sizeof(VT) != 2 || sizeof(LT) != 4 || vec_lpv(d, sizeof(VT), flags) < 0)) return TRANSOAR_ERR_MODE;
  const long n_items = static_cast<long>(d.N) * d.Lq * d.M;
  const BrickOrder q_order = make_order(host_shapes, d, d.Lq);
  const long q_units = order_units(q_order, d, d.Lq);
  const long n_blocks = (q_units + kWavesPerBlock - 1) / kWavesPerBlock;
  const int lg = vec_lpv(d, sizeof(VT), flags);
  const dim3 block(64 * kWavesPerBlock);
  auto v = static_cast<const VT*>(value);
  auto lo = static_cast<const LT*>(loc);
  auto at = static_cast<const LT*>(attn);
  auto go = static_cast<const VT*>(grad_out);
  auto gl = static_cast<LT*>(grad_loc);
  auto ga = static_cast<LT*>(grad_attn);
  const size_t value_elems = static_cast<size_t>(d.N) * d.S * d.M * d.C;

  if (lg < 0) {
    // scatter with fp atomics into a zeroed accumulator of type A
    ProfScope prof(TRANSOAR_PROF_BWD_GENERIC, st);
    A* gv = sizeof(VT) == 2 ? static_cast<A*>(workspace) : static_cast<A*>(grad_value);
    TRANSOAR_CHECK_HIP(hipMemsetAsync(gv, 0, sizeof(A) * value_elems, st));
    hipLaunchKernelGGL((msda3d_bwd_generic<VT, LT>), dim3((n_items + kWavesPerBlock - 1) / kWavesPerBlock),
                       block, 0, st, v, shapes, lsi, lo, at, go, gv, gl, ga, d.S, d.M, d.C, d.L, d.Lq, d.P,
                       n_items);
    if (sizeof(VT) == 2) {
      const int cast_blocks = static_cast<int>(std::min<size_t>((value_elems + 255) / 256, 1 << 16));
      hipLaunchKernelGGL((msda3d_cast_rows<VT>), dim3(cast_blocks), dim3(256), 0, st,
                         reinterpret_cast<const float*>(gv), static_cast<VT*>(grad_value),
                         static_cast<long>(value_elems));
    }
    return static_cast<int>(hipGetLastError());
  }

  const dim3 grid(((n_blocks + 7) / 8) * 8);
  const unsigned vbytes = static_cast<unsigned>(value_elems * sizeof(VT));
  const BwdWorkspace w = bwd_workspace(d, sizeof(A), flags);
  char* ws = static_cast<char*>(workspace);
  int* count = reinterpret_cast<int*>(ws + w.count);
  int* tile_sums = reinterpret_cast<int*>(ws + w.tile_sums);
  int* rank = reinterpret_cast<int*>(ws + w.rank);
  int* rec_item = reinterpret_cast<int*>(ws + w.rec_item);
  auto recs = reinterpret_cast<PointRec<A>*>(ws + w.recs);
  // cells per (batch, head) slab: needs the level shapes on the host; without them the binning
  // runs as its own kernel (msda3d_cell_count) after the gather
  long cells_per_slab = 0;
  if (host_shapes)
    for (int l = 0; l < d.L; ++l)
      cells_per_slab += (host_shapes[3 * l] + 1) * (host_shapes[3 * l + 1] + 1) * (host_shapes[3 * l + 2] + 1);
  const bool fold_count = host_shapes != nullptr && cells_per_slab * d.N * d.M <= w.n_bins;
  // entries of the count / offset array that are in use: one per cell + the end of the last run (the upper bound
  // n_bins + 1 = 8 S N M + 1 without host shapes: zeroing and scanning it was 45 MB four times over at the flagship size)
  // (deterministic mode reads the walks' offsets one entry further on -- see below -- hence one more scanned entry)
  const bool det_req = (flags & TRANSOAR_MSDA3D_DETERMINISTIC) != 0;
  const long n_scan = fold_count ? cells_per_slab * d.N * d.M + 1 + (det_req && cells_per_slab * d.N * d.M + 2 <= w.n_scan ? 1 : 0) : w.n_scan;
  const long n_tiles = (n_scan + kScanTile - 1) / kScanTile;
  TRANSOAR_CHECK_HIP(zero_async(count, sizeof(int) * n_scan, st));
  auto scan = [&]() {
    ProfScope prof(TRANSOAR_PROF_SCAN, st);
    hipLaunchKernelGGL(msda3d_scan_tiles, dim3(static_cast<unsigned>(n_tiles)), dim3(kScanThreads), 0,
                       st, count, tile_sums, static_cast<int>(n_scan));
    hipLaunchKernelGGL(msda3d_scan_tile_sums, dim3(1), dim3(kScanThreads), 0, st, tile_sums,
                       static_cast<int>(n_tiles));
    hipLaunchKernelGGL(msda3d_scan_add, dim3(static_cast<unsigned>(n_tiles)), dim3(kScanThreads), 0, st,
                       count, tile_sums, static_cast<int>(n_scan));
  };

  // 1. grad_loc / grad_attn (+ the binning pass of the point sort when folded)
#define TRANSOAR_BWDQ(LG)                                                                   \
  hipLaunchKernelGGL((msda3d_bwd_query_vec<VT, LT, LG>), grid, block, 0, st, v, shapes, lsi, lo, \
                     at, go, gl, ga, fold_count ? count : nullptr, rank, static_cast<int>(cells_per_slab), \
                     d.S, d.M, d.C, d.L, d.Lq, d.P, vbytes, q_units, n_blocks, q_order)
  bool brick_done = false, recs_done = false, det = false;
  if constexpr (sizeof(VT) == 2) {
    bool small = d.L <= kMmaLevels;
    for (int l = 0; small && l < d.L; ++l) small = q_order.D[l] <= 1000 && q_order.H[l] <= 1000 && q_order.W[l] <= 1000;
    if (q_order.enabled && fold_count && small && d.C == 64 && d.P == 4 &&
        !(flags & (TRANSOAR_MSDA3D_NO_BRICK | TRANSOAR_MSDA3D_NO_MMA))) {
      // Matrix-core chain: count the points per cell (locations only), scan, then ONE kernel computes grad_loc /
      // grad_attn and writes every point's 16-byte record at its sorted position.  The counts go to count[cell + 1]:
      // after the exclusive scan that slot holds the first position of cell `cell` and serves as its cursor; when the
      // records are written it has advanced to the first position of cell + 1, i.e. count[] is the offset array the
      // grad_value walks read (offset[c], offset[c + 1]) without another pass.
      const long n_wave = static_cast<long>(d.N) * (q_order.pad_start[q_order.L] >> 7) * d.M * 4;
      const BrickOrder* order_d = device_const(q_order, st);
      if (order_d == nullptr) return TRANSOAR_ERR_CONST;
      const dim3 qgrid(static_cast<unsigned>(((n_wave + 7) / 8) * 8));
      {
        ProfScope prof(TRANSOAR_PROF_CELL_COUNT, st);
        hipLaunchKernelGGL((msda3d_cell_count_mma<LT>), qgrid, dim3(64), 0, st, lo, count + 1,
                           static_cast<int>(cells_per_slab), d.S, d.M, d.L, n_wave, order_d);
      }
      scan();
      det = det_req && cells_per_slab * d.N * d.M + 2 <= w.n_scan && w.n_points < (1L << 31);      // the sort takes the count as an int
      if (det_req && !det) return TRANSOAR_ERR_MODE;
      if (!det) {
        ProfScope prof(TRANSOAR_PROF_BWD_QUERY, st);
        hipLaunchKernelGGL((msda3d_bwd_query_mma<VT, LT>), qgrid, dim3(64), 0, st,
                           v, lo, at, go, gl, ga, count + 1, reinterpret_cast<PointR16*>(ws + w.recs), nullptr,
                           static_cast<int>(cells_per_slab), d.S, d.M, d.L, vbytes, n_wave, order_d,
                           static_cast<unsigned short*>(grad_proj));
      } else {
        // Deterministic order: every point's record goes to its canonical slot (upper half of the record region) with
        // the key (cell << 32 | slot); a stable radix sort of the keys (skipped points keep the all-ones key and end up
        // last) and one gather put the records of a cell next to each other in slot order.  The exclusive scan of the
        // counts is left as it is (no cursor advanced): count[c + 1] is the FIRST position of cell c, i.e. the walks
        // read their (offset[c], offset[c + 1]) pairs from count + 1.
        auto keys = reinterpret_cast<unsigned long long*>(ws + w.det_keys);
        auto alt = reinterpret_cast<unsigned long long*>(ws + w.det_alt);
        PointR16* slots = reinterpret_cast<PointR16*>(ws + w.recs) + w.n_points;
        TRANSOAR_CHECK_HIP(hipMemsetAsync(keys, 0xff, sizeof(unsigned long long) * w.n_points, st));
        {
          ProfScope prof(TRANSOAR_PROF_BWD_QUERY, st);
          hipLaunchKernelGGL((msda3d_bwd_query_mma<VT, LT>), qgrid, dim3(64), 0, st,
                             v, lo, at, go, gl, ga, count + 1, slots, keys,
                             static_cast<int>(cells_per_slab), d.S, d.M, d.L, vbytes, n_wave, order_d,
                             static_cast<unsigned short*>(grad_proj));
        }
        ProfScope prof(TRANSOAR_PROF_CELL_FILL, st);
        int cell_bits = 1;
        while ((cells_per_slab * d.N * d.M) >> cell_bits) ++cell_bits;
        size_t need = 0;
        int rc = sort_keys64(keys, alt, w.n_points, 32 + cell_bits, nullptr, &need, nullptr, st);
        if (rc != 0) return rc;
        if (need > w.det_temp_bytes) return TRANSOAR_ERR_WORKSPACE;
        unsigned long long* sorted = nullptr;
        size_t temp_bytes = w.det_temp_bytes;
        rc = sort_keys64(keys, alt, w.n_points, 32 + cell_bits, ws + w.det_temp, &temp_bytes, &sorted, st);
        if (rc != 0) return rc;
        hipLaunchKernelGGL(msda3d_det_gather, dim3(static_cast<unsigned>((w.n_points + 255) / 256)), dim3(256), 0, st,
                           sorted, slots, reinterpret_cast<PointR16*>(ws + w.recs), w.n_points);
      }
      brick_done = recs_done = true;
    }
  }
  if (det_req && !det) return TRANSOAR_ERR_MODE;           // no silent fall-back to an order that depends on atomics
  if (grad_proj != nullptr && !recs_done) return TRANSOAR_ERR_MODE;      // the other query kernels write grad_loc / grad_attn
  if constexpr (sizeof(VT) == 2) {
    if (!brick_done && q_order.enabled && fold_count && d.C == 64 && d.P == 4 && !(flags & TRANSOAR_MSDA3D_NO_BRICK)) {
      ProfScope prof(TRANSOAR_PROF_BWD_QUERY, st);
      const long n_wg = static_cast<long>(d.N) * (q_order.pad_start[q_order.L] >> 7) * d.M;
      hipLaunchKernelGGL((msda3d_bwd_query_brick<VT, LT, 4, 64>), dim3(((n_wg + 7) / 8) * 8), dim3(kBrickThreads),
                         kTileBytes, st, v, lo, at, go, gl, ga, count, rank, static_cast<int>(cells_per_slab), d.S, d.M,
                         d.L, n_wg, q_order);
      brick_done = true;
    }
  }
  if (!brick_done) {
    ProfScope prof(TRANSOAR_PROF_BWD_QUERY, st);
    if (lg == 3) TRANSOAR_BWDQ(3);
    else if (lg == 4) TRANSOAR_BWDQ(4);
    else TRANSOAR_BWDQ(5);
  }
#undef TRANSOAR_BWDQ

  // 2. sort the sampling points by cell
  const dim3 pgrid(static_cast<unsigned>((w.n_points + 255) / 256));
  if (!fold_count) {
    ProfScope prof(TRANSOAR_PROF_CELL_COUNT, st);
    hipLaunchKernelGGL((msda3d_cell_count<LT, A>), pgrid, dim3(256), 0, st, lo, at, shapes, lsi, count,
                       rank, d.M, d.L, d.Lq, d.P, w.n_points);
  }
  if (!recs_done) scan();
  BrickOrder r_order = make_order(host_shapes, d, d.S);
  if constexpr (sizeof(A) == 4) {
    // brick-owner schedule: fp32 LDS tile per 4x4x8 brick, 8-weight point records
    if (r_order.enabled && fold_count && d.C == kTileC && !(flags & TRANSOAR_MSDA3D_NO_BRICK)) {
      auto recs8 = reinterpret_cast<PointW8<float>*>(ws + w.recs);
      auto recs4 = reinterpret_cast<PointR16*>(ws + w.recs);       // matrix-core walks: 16 bytes with the row index inside
      bool mma = false;
      if constexpr (sizeof(VT) == 2) mma = !(flags & TRANSOAR_MSDA3D_NO_MMA);
      if (!recs_done) {
        ProfScope prof(TRANSOAR_PROF_CELL_FILL, st);
        if (mma)
          hipLaunchKernelGGL((msda3d_cell_fill_r16<LT>), pgrid, dim3(256), 0, st, lo, at, shapes, lsi, count, rank, recs4,
                             d.M, d.L, d.Lq, d.P, w.n_points);
        else
          hipLaunchKernelGGL((msda3d_cell_fill_w8<LT, float>), pgrid, dim3(256), 0, st, lo, at, shapes, lsi, count,
                             rank, recs8, rec_item, d.M, d.L, d.Lq, d.P, w.n_points);
      }
      // levels whose voxels receive >= kCoarsePointsPerVoxel points each go to the chunked walk
      CoarseLevels cl{d.L, static_cast<int>(cells_per_slab), d.S, 0, 0};
      const int* offsets = det ? count + 1 : count;
      for (int l = d.L - 1; l >= 0 && !det; --l) {          // deterministic mode: no coarse level (their walk flushes with fp32 atomics)
        const long vox = host_shapes[3 * l] * host_shapes[3 * l + 1] * host_shapes[3 * l + 2];
        if (static_cast<long>(d.Lq) * d.P < kCoarsePointsPerVoxel * vox) break;
        cl.first = l;
        cl.cell_start -= static_cast<int>((host_shapes[3 * l] + 1) * (host_shapes[3 * l + 1] + 1) * (host_shapes[3 * l + 2] + 1));
        cl.row_start = r_order.start[l];
      }
      cl.rows = d.S - cl.row_start;
      const int coarse_levels = d.L - cl.first;
      const int cell_chunk = mma ? kCmCellChunk : kCellChunk;
      cl.chunks_per_slab = static_cast<int>((static_cast<long>(d.Lq) * d.P * coarse_levels + cell_chunk - 1) / cell_chunk) + 1;
      const int fine_bricks = r_order.pad_start[cl.first] >> 7;
      const BrickOrder* r_order_d = device_const(r_order, st);
      const CoarseLevels* cl_d = device_const(cl, st);
      if (r_order_d == nullptr || cl_d == nullptr) return TRANSOAR_ERR_CONST;
      float* scratch = reinterpret_cast<float*>(ws + w.coarse);
      const long scratch_elems = static_cast<long>(d.N) * cl.rows * d.M * kTileC;
      // opt-in (TRANSOAR_MSDA3D_FORK): 3.90 -> 3.59 ms per eager backward at the flagship, but 1.1 ms per
      // step SLOWER when the step is replayed as a HIP graph (fork/join nodes), which is how bench.py runs
      SideStream* side = (g_prof_on || !(flags & TRANSOAR_MSDA3D_FORK) || coarse_levels == 0 || fine_bricks == 0)
                             ? nullptr : side_stream();
      hipStream_t cst = st;
      if (side != nullptr) {
        TRANSOAR_CHECK_HIP(hipEventRecord(side->fork, st));
        TRANSOAR_CHECK_HIP(hipStreamWaitEvent(side->stream, side->fork, 0));
        cst = side->stream;
      }
      if (coarse_levels > 0) {
        ProfScope prof(TRANSOAR_PROF_VALUE_CELLS, cst);
        TRANSOAR_CHECK_HIP(zero_async(scratch, sizeof(float) * scratch_elems, cst));
        const long waves = static_cast<long>(d.N) * d.M * cl.chunks_per_slab;
        if (mma) {
          if constexpr (sizeof(VT) == 2)       // 16 sorted points per MFMA K-step (msda3d_cells_mma.hpp)
            hipLaunchKernelGGL((msda3d_bwd_value_cells_mma<VT>), dim3(static_cast<unsigned>((waves + 3) / 4)), dim3(256), 0,
                               cst, go, offsets, recs4, scratch, static_cast<int>(cells_per_slab), d.N * d.M, d.M,
                               cl_d, r_order_d);
        } else {
          hipLaunchKernelGGL((msda3d_bwd_value_cells<VT>), dim3(static_cast<unsigned>((waves + 3) / 4)), dim3(256), 0, cst,
                             go, count, recs8, rec_item, scratch, static_cast<int>(cells_per_slab), d.N * d.M, d.M, cl_d,
                             r_order_d);
        }
      }
      if (side != nullptr) TRANSOAR_CHECK_HIP(hipEventRecord(side->join, side->stream));
      ProfScope prof(TRANSOAR_PROF_VALUE_TILE, st);
      if (fine_bricks > 0) {
        const long n_wg = static_cast<long>(d.N) * fine_bricks * d.M;
        if (mma) {
          if constexpr (sizeof(VT) == 2)
            hipLaunchKernelGGL((msda3d_bwd_value_tile_mma<VT>), dim3(static_cast<unsigned>(n_wg)), dim3(kBrickThreads), 0, st,
                               go, offsets, recs4, static_cast<VT*>(grad_value), static_cast<int>(cells_per_slab),
                               d.S, d.M, fine_bricks, n_wg, r_order_d);
        } else {
          hipLaunchKernelGGL((msda3d_bwd_value_tile<VT>), dim3(static_cast<unsigned>(n_wg)), dim3(kBrickThreads), 0, st, go,
                             count, recs8, rec_item, static_cast<VT*>(grad_value), static_cast<int>(cells_per_slab),
                             d.S, d.M, fine_bricks, n_wg, r_order_d);
        }
      }
      if (side != nullptr) TRANSOAR_CHECK_HIP(hipStreamWaitEvent(st, side->join, 0));
      if (coarse_levels > 0) {
        const long n4 = scratch_elems / 4;
        hipLaunchKernelGGL((msda3d_coarse_rows_store<VT>), dim3(static_cast<unsigned>((n4 + 255) / 256)), dim3(256), 0, st,
                           scratch, static_cast<VT*>(grad_value), d.S, d.M, cl.rows, cl.row_start, n4);
      }
      return static_cast<int>(hipGetLastError());
    }
  }
  {
    ProfScope prof(TRANSOAR_PROF_CELL_FILL, st);
    hipLaunchKernelGGL((msda3d_cell_fill<LT, A>), pgrid, dim3(256), 0, st, lo, at, shapes, lsi, count,
                       rank, recs, rec_item, d.M, d.L, d.Lq, d.P, w.n_points);
  }

  // 3. grad_value rows
  if (r_order.enabled && (flags & TRANSOAR_MSDA3D_PULL_HEAD_MAJOR)) r_order.enabled = 2;
  const long n_rows = order_units(r_order, d, d.S);
  const long r_blocks = (n_rows + kWavesPerBlock - 1) / kWavesPerBlock;
  const dim3 rgrid(((r_blocks + 7) / 8) * 8);
#define TRANSOAR_PULL(LG)                                                                        \
  hipLaunchKernelGGL((msda3d_bwd_value_pull<VT, A, LG>), rgrid, block, 0, st, go, shapes, lsi, count, \
                     recs, rec_item, static_cast<VT*>(grad_value), d.S, d.M, d.C, d.L, n_rows, r_blocks, r_order, \
                     static_cast<unsigned>(w.n_points * 4 * sizeof(A)))
  {
    ProfScope prof(TRANSOAR_PROF_PULL, st);
    if (lg == 3) TRANSOAR_PULL(3);
    else if (lg == 4) TRANSOAR_PULL(4);
    else TRANSOAR_PULL(5);
  }
#undef TRANSOAR_PULL
  return static_cast<int>(hipGetLastError());
}

static int check_common(const Dims& d, int value_dtype, int loc_dtype) {
  if (d.N <= 0 || d.S <= 0 || d.M <= 0 || d.C <= 0 || d.L <= 0 || d.Lq <= 0 || d.P <= 0)
    return TRANSOAR_ERR_DIM;
  if (d.L > TRANSOAR_MSDA3D_MAX_LEVELS) return TRANSOAR_ERR_LEVELS;
  // item index and row index are 32-bit inside the kernels
  if (static_cast<long>(d.N) * d.Lq * d.M >= (1L << 31) || static_cast<long>(d.N) * d.S >= (1L << 31))
    return TRANSOAR_ERR_DIM;
  const bool half = value_dtype == TRANSOAR_BF16 || value_dtype == TRANSOAR_F16;
  if (value_dtype < 0 || value_dtype > TRANSOAR_F16) return TRANSOAR_ERR_DTYPE;
  if (!(loc_dtype == value_dtype || (half && loc_dtype == TRANSOAR_F32))) return TRANSOAR_ERR_DTYPE;
  return TRANSOAR_OK;
}

static inline bool misaligned(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) != 0; }

}  // namespace transoar

using namespace transoar;

#define TRANSOAR_DISPATCH(VD, LD, CALL)                                            \
  do {                                                                             \
    if (VD == TRANSOAR_F32) { using VT = float; using LT = float; return CALL; }   \
    if (VD == TRANSOAR_F64) { using VT = double; using LT = double; return CALL; } \
    if (VD == TRANSOAR_BF16 && LD == TRANSOAR_BF16) { using VT = bf16_t; using LT = bf16_t; return CALL; } \
    if (VD == TRANSOAR_BF16) { using VT = bf16_t; using LT = float; return CALL; } \
    if (VD == TRANSOAR_F16 && LD == TRANSOAR_F16) { using VT = f16_t; using LT = f16_t; return CALL; }     \
    { using VT = f16_t; using LT = float; return CALL; }                           \
  } while (0)

extern "C" int transoar_msda3d_forward(const void* value, const int64_t* spatial_shapes,
                                       const int64_t* level_start_index, const void* sampling_loc,
                                       const void* attn_weight, void* out, int N, int S, int M,
                                       int C, int L, int Lq, int P, int value_dtype, int loc_dtype,
                                       const int64_t* host_spatial_shapes, unsigned flags,
                                       void* hip_stream) {
  if (!value || !spatial_shapes || !level_start_index || !sampling_loc || !attn_weight || !out)
    return TRANSOAR_ERR_NULL;
  const Dims d{N, S, M, C, L, Lq, P};
  const int rc = check_common(d, value_dtype, loc_dtype);
  if (rc != TRANSOAR_OK) return rc;
  if (misaligned(value) || misaligned(sampling_loc) || misaligned(attn_weight) || misaligned(out))
    return TRANSOAR_ERR_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  TRANSOAR_DISPATCH(value_dtype, loc_dtype,
                    (launch_fwd<VT, LT>(value, spatial_shapes, level_start_index, sampling_loc,
                                        attn_weight, out, d, host_spatial_shapes, flags, st)));
}

extern "C" int transoar_msda3d_forward_fused(const void* value, const void* proj, const float* ref, long ref_rows,
                                             void* out, int N, int S, int M, int C, int L, int P, int value_dtype,
                                             const int64_t* host_spatial_shapes, void* hip_stream) {
  if (!value || !proj || !ref || !out || !host_spatial_shapes) return TRANSOAR_ERR_NULL;
  const Dims d{N, S, M, C, L, S, P};
  const int rc = check_common(d, value_dtype, TRANSOAR_F32);
  if (rc != TRANSOAR_OK) return rc;
  if (value_dtype != TRANSOAR_BF16 && value_dtype != TRANSOAR_F16) return TRANSOAR_ERR_DTYPE;
  if (misaligned(value) || misaligned(proj) || misaligned(ref) || misaligned(out)) return TRANSOAR_ERR_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  if (value_dtype == TRANSOAR_BF16) return launch_fwd_fused<bf16_t>(value, proj, ref, ref_rows, out, d, host_spatial_shapes, st);
  return launch_fwd_fused<f16_t>(value, proj, ref, ref_rows, out, d, host_spatial_shapes, st);
}

extern "C" int transoar_msda3d_backward(const void* value, const int64_t* spatial_shapes,
                                        const int64_t* level_start_index, const void* sampling_loc,
                                        const void* attn_weight, const void* grad_out,
                                        void* grad_value, void* grad_sampling_loc,
                                        void* grad_attn_weight, void* workspace,
                                        size_t workspace_bytes, int N, int S, int M, int C, int L,
                                        int Lq, int P, int value_dtype, int loc_dtype,
                                        const int64_t* host_spatial_shapes, unsigned flags,
                                        void* hip_stream) {
  if (!value || !spatial_shapes || !level_start_index || !sampling_loc || !attn_weight ||
      !grad_out || !grad_value || !grad_sampling_loc || !grad_attn_weight)
    return TRANSOAR_ERR_NULL;
  const Dims d{N, S, M, C, L, Lq, P};
  const int rc = check_common(d, value_dtype, loc_dtype);
  if (rc != TRANSOAR_OK) return rc;
  if (misaligned(value) || misaligned(sampling_loc) || misaligned(attn_weight) ||
      misaligned(grad_out) || misaligned(grad_value) || misaligned(grad_sampling_loc) ||
      misaligned(grad_attn_weight) || misaligned(workspace))
    return TRANSOAR_ERR_ALIGN;
  if (!workspace && workspace_bytes != 0) return TRANSOAR_ERR_NULL;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  TRANSOAR_DISPATCH(value_dtype, loc_dtype,
                    (launch_bwd<VT, LT>(value, spatial_shapes, level_start_index, sampling_loc,
                                        attn_weight, grad_out, grad_value, grad_sampling_loc,
                                        grad_attn_weight, workspace, workspace_bytes, d,
                                        host_spatial_shapes, flags, st)));
}

extern "C" int transoar_msda3d_backward_proj(const void* value, const int64_t* spatial_shapes,
                                             const int64_t* level_start_index, const void* sampling_loc,
                                             const void* attn_weight, const void* grad_out,
                                             void* grad_value, void* grad_proj, void* workspace,
                                             size_t workspace_bytes, int N, int S, int M, int C, int L,
                                             int Lq, int P, int value_dtype, int loc_dtype,
                                             const int64_t* host_spatial_shapes, unsigned flags,
                                             void* hip_stream) {
  if (!value || !spatial_shapes || !level_start_index || !sampling_loc || !attn_weight ||
      !grad_out || !grad_value || !grad_proj)
    return TRANSOAR_ERR_NULL;
  const Dims d{N, S, M, C, L, Lq, P};
  const int rc = check_common(d, value_dtype, loc_dtype);
  if (rc != TRANSOAR_OK) return rc;
  if (loc_dtype != TRANSOAR_F32 || (value_dtype != TRANSOAR_BF16 && value_dtype != TRANSOAR_F16) || L * P != 16) return TRANSOAR_ERR_MODE;
  if (misaligned(value) || misaligned(sampling_loc) || misaligned(attn_weight) ||
      misaligned(grad_out) || misaligned(grad_value) || misaligned(grad_proj) || misaligned(workspace))
    return TRANSOAR_ERR_ALIGN;
  if (!workspace && workspace_bytes != 0) return TRANSOAR_ERR_NULL;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  TRANSOAR_DISPATCH(value_dtype, loc_dtype,
                    (launch_bwd<VT, LT>(value, spatial_shapes, level_start_index, sampling_loc,
                                        attn_weight, grad_out, grad_value, nullptr, nullptr, workspace, workspace_bytes, d,
                                        host_spatial_shapes, flags, st, grad_proj)));
}

extern "C" size_t transoar_msda3d_backward_workspace_bytes(int N, int S, int M, int C, int L, int Lq,
                                                           int P, int value_dtype, int loc_dtype,
                                                           unsigned flags) {
  const Dims d{N, S, M, C, L, Lq, P};
  if (check_common(d, value_dtype, loc_dtype) != TRANSOAR_OK) return 0;
  TRANSOAR_DISPATCH(value_dtype, loc_dtype, (bwd_workspace_bytes<VT, LT>(d, flags)));
}

extern "C" const char* transoar_msda3d_strerror(int code) {
  switch (code) {
    case TRANSOAR_OK: return "success";
    case TRANSOAR_ERR_NULL: return "a required pointer is NULL";
    case TRANSOAR_ERR_DIM: return "a dimension is non-positive or exceeds the 32-bit index range";
    case TRANSOAR_ERR_DTYPE: return "unsupported value/loc dtype combination";
    case TRANSOAR_ERR_ALIGN: return "a device buffer is not 16-byte aligned";
    case TRANSOAR_ERR_LEVELS: return "too many feature levels";
    case TRANSOAR_ERR_WORKSPACE: return "workspace is smaller than transoar_msda3d_backward_workspace_bytes()";
    case TRANSOAR_ERR_MODE: return "the deterministic backward does not cover this form (16-bit storage, C = 64, P = 4, <= 4 host-known levels, queries = pyramid voxels)";
    case TRANSOAR_ERR_CONST: return "could not place the launch constants in device memory (first call for a shape must not be inside a stream capture)";
    default: return code > 0 ? hipGetErrorString(static_cast<hipError_t>(code)) : "unknown error";
  }
}

extern "C" void transoar_msda3d_profile_enable(int on) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_on = on != 0;
}

extern "C" int transoar_msda3d_profile_read(double* total_ms, long* launches) {
  std::vector<ProfPair> live;
  {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    live.swap(g_prof_live);
  }
  for (int k = 0; k < TRANSOAR_PROF_KINDS; ++k) {
    total_ms[k] = 0.0;
    launches[k] = 0;
  }
  int rc = 0;
  for (const ProfPair& p : live) {
    float ms = 0.f;
    hipError_t e = hipEventSynchronize(p.end);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, p.beg, p.end);
    if (e != hipSuccess) { rc = static_cast<int>(e); continue; }
    total_ms[p.kind] += ms;
    launches[p.kind] += 1;
  }
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_free.insert(g_prof_free.end(), live.begin(), live.end());
  return rc;
}

extern "C" int transoar_msda3d_abi_version(void) { return 6; }
