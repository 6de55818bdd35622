// Internal declarations shared by the HIP translation units of libpcabo.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#define PCABO_BS 64          // tile edge of the Gram / Cholesky / root-inverse kernels
#define PCABO_SLAB 16        // rows of R handled by one acquisition work-group
#define PCABO_MAXD 128       // largest ambient / reduced dimension supported
#define PCABO_TLD 66         // LDS leading dimension (doubles) of a 64x64 tile: conflict-free ds_read_b64

#define PCABO_QA_MAX 416      // query coordinates that travel as kernel arguments (3.25 KB; 10 restarts x 40 dims = 400 fit).  With 448 the
                              // kernarg segment of k_acq_fast was exactly HIP's 4096-byte maximum; 416 leaves 256 bytes of head room
#define PCABO_CNT_DONE 0x3fff // capacity of the per-query ticket array
#define PCABO_INLAUNCH_MAXQ 32 // largest batch finished inside the launch (results + flags straight to the host)

typedef double double4_t __attribute__((ext_vector_type(4)));
struct QueryArgs { double x[PCABO_QA_MAX]; };
// Mailbox of the resident ("server") acquisition kernel: (value, tag) pairs of 16 bytes.  A pair is written value
// first, tag second (one 16-byte store on the device) and read as one 16-byte snapshot, so a matching tag implies the
// value.  Pair 0 is the header (value = number of queries of the round, 0 = leave), pairs 1.. the query coordinates.
struct alignas(16) MailPair { double v; unsigned long long tag; };
// Pairs WRITTEN BY THE HOST carry tag = sequence number XOR mail_mix(bits of the value): a reader that sees a pair torn
// between two rounds (new tag, old value - possible in principle when a write-combining buffer of the host is flushed in
// pieces) computes a different sequence number and simply polls again.
static inline __host__ __device__ unsigned long long mail_mix(unsigned long long vbits) {
  const unsigned int lo = (unsigned int)vbits, hi = (unsigned int)(vbits >> 32);
  return (unsigned long long)((lo ^ hi ^ (hi >> 11)) & 0xFFFFFFu);
}
#define PCABO_MAIL_PAIRS (1 + PCABO_QA_MAX + PCABO_INLAUNCH_MAXQ + 1)   // [0] unused, coordinates, one control pair per query, probe
#define PCABO_SERVER_TIMEOUT_TICKS 200000000ull // 2 s of wall_clock64 (100 MHz): every wait in the kernel is bounded

#ifdef __HIPCC__
// fp64 reciprocal / reciprocal square root from the hardware estimate (v_rcp_f64 / v_rsq_f64) plus two
// Newton steps: <= 2 ulp, ~6 instructions instead of the ~25-40 of an IEEE divide / sqrt sequence.  Used where
// a sequential dependency chain makes instruction count the bottleneck (panel factorisation, Jacobi rotations).
__device__ inline double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}
__device__ inline double fast_rsq(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * fma(-0.5 * x * y, y, 1.5);
  y = y * fma(-0.5 * x * y, y, 1.5);
  return y;
}
#endif

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// Model / problem constants handed to the acquisition kernels by value.
struct AcqParams {
  double best_f;        // already rounded like torch.as_tensor(float) does
  double y_mean, y_std; // filled on device from the standardisation kernel (host copy unused)
  double inv_ls;        // 1 / lengthscale
  int maximize;
  int acq;              // PCABO_ACQ_*
  int kernel;           // PCABO_KERNEL_*
  int want_grad;
};

// Batched acquisition launches (all zero / null for a single context).  Runs are addressed as in zrun(); per run the
// reduced dimension and best_f come from device memory (they differ between the runs of a batch).
struct AcqBatch {
  size_t zs = 0, hzs = 0;          // strides of the device region / the pinned host region
  const int* k_dev = nullptr;      // run 0's dK      (null: the k argument)
  const double* bestf = nullptr;   // run 0's dBestF  (null: prm.best_f)
  int table = 0;                   // 1: blockIdx.y indexes the active-query table that travels in the QueryArgs slot
                                   //    (32-bit entries: run << 16 | query of that run); 0: run = blockIdx.z
  int xq_host = 0;                 // 1: Xq is a pinned HOST pointer (stride hzs), read by the kernel over PCIe
};

// Small results the host needs after a device phase; lives in pinned host memory.
struct HostMirror {
  int k;                 // reduced dimension chosen by the wPCA
  int chol_info;         // 0 or 1 + index of the failing pivot
  int pad0, pad1;
  double y_mean, y_std;
  double norm_lo[PCABO_MAXD], norm_hi[PCABO_MAXD];
  double acq_lo[PCABO_MAXD], acq_hi[PCABO_MAXD];
  volatile unsigned long long qflag[PCABO_INLAUNCH_MAXQ];   // per query: sequence number written after its results
};

// ---- launchers (defined in kernels_*.hip); all asynchronous on `s` -------------------
#if defined(__HIPCC__)
// 16-byte loads / stores that go to the fabric (system scope: no L1/L2 copy is trusted or left behind)
typedef unsigned int pcabo_u4 __attribute__((ext_vector_type(4)));
__device__ inline pcabo_u4 ld_pair_sys(const void* p) {
  pcabo_u4 r;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
  return r;
}
__device__ inline void st_pair_sys(void* p, pcabo_u4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
}
// up to 8 pair loads in flight, one wait (cnt <= 8; unused slots repeat slot 0)
__device__ inline void ld_pairs_sys8(const void* const* p, pcabo_u4* o) {
  asm volatile(
      "global_load_dwordx4 %0, %8, off sc0 sc1\n\tglobal_load_dwordx4 %1, %9, off sc0 sc1\n\t"
      "global_load_dwordx4 %2, %10, off sc0 sc1\n\tglobal_load_dwordx4 %3, %11, off sc0 sc1\n\t"
      "global_load_dwordx4 %4, %12, off sc0 sc1\n\tglobal_load_dwordx4 %5, %13, off sc0 sc1\n\t"
      "global_load_dwordx4 %6, %14, off sc0 sc1\n\tglobal_load_dwordx4 %7, %15, off sc0 sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7])
      : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7])
      : "memory");
}
__device__ inline pcabo_u4 make_pair(double v, unsigned long long tag) {
  pcabo_u4 r;
  r.x = (unsigned int)__double2loint(v); r.y = (unsigned int)__double2hiint(v);
  r.z = (unsigned int)(tag & 0xffffffffull); r.w = (unsigned int)(tag >> 32);
  return r;
}
__device__ inline unsigned long long pair_tag(pcabo_u4 v) { return ((unsigned long long)v.w << 32) | v.z; }
__device__ inline double pair_value(pcabo_u4 v) { return __hiloint2double((int)v.y, (int)v.x); }
// sequence number of a host-written pair (see mail_mix)
__device__ inline unsigned long long mail_seq(pcabo_u4 v) {
  return pair_tag(v) ^ mail_mix(((unsigned long long)v.y << 32) | v.x);
}

// Batched launches (pcabo_batch_*): blockIdx.z = run.  The contexts of a batch share one layout, so run b's copy of any
// device buffer sits b * zs bytes behind run 0's (b * hzs for the pinned host mirror).  zs = 0 / gridDim.z = 1 otherwise.
// (plain pointer arithmetic on the kernel argument, no detour through an integer: the compiler then still knows the
// result points to GLOBAL memory and emits global_load/global_store; through uintptr_t it falls back to flat_* accesses,
// which also count on the LDS counter - every LDS wait then waits for the global loads in flight as well)
template <typename T>
__device__ inline T* zrun(T* p, size_t stride, unsigned run) {
  typedef typename std::remove_const<T>::type U;
  return p ? reinterpret_cast<T*>(reinterpret_cast<char*>(const_cast<U*>(p)) + (size_t)run * stride) : p;   // (NULL stays NULL)
}
#define ZRUN(p) p = zrun(p, zs, blockIdx.z)

// XCD-aware placement of a batched launch (grid (gx, gy, B), B runs of gx * gy tiles each).  Work-groups are dealt round-robin
// over the 8 XCDs in dispatch order (x fastest), so the tiles of ONE run land on eight different L2s and every XCD fetches
// the operand tiles those work-groups share (the block row L[J][.] of a look-back, the diagonal block of a panel, the L
// tiles a column chunk of the root inverse walks) for itself.  The linear work-group id is read as (run, tile) such that all
// tiles of a run have the same id % 8: one XCD, one L2 copy of the shared operands.  Placement only - every (run, tile)
// pair is still visited exactly once, whatever the hardware does with the ids; runs beyond the last multiple of 8 (and
// single contexts) keep the identity mapping.
struct XcdTile { unsigned run, x, y; };
#if defined(__HIPCC__)
__device__ inline XcdTile xcd_tile() {
  const unsigned gx = gridDim.x, gy = gridDim.y, T = gx * gy;
  const unsigned lin = (blockIdx.z * gy + blockIdx.y) * gx + blockIdx.x, full = (gridDim.z & ~7u) * T;
  XcdTile t;
  if (lin < full) {
    const unsigned slot = lin >> 3, tile = slot % T;
    t.run = (slot / T) * 8u + (lin & 7u); t.x = tile % gx; t.y = tile / gx;
  } else { t.run = blockIdx.z; t.x = blockIdx.x; t.y = blockIdx.y; }
  return t;
}
#endif
#define ZRUNX(p) p = zrun(p, zs, xt_.run)

// Wave-wide sum without LDS traffic.  `__shfl_xor` compiles to ds_bpermute_b32 (two per double, each followed by an
// lgkmcnt wait: ~100 cycles of dependent latency per step); in a kernel whose whole budget is ~15 us the reductions
// were the largest single item.  DPP steps stay inside the VALU: quad_perm x2, row_half_mirror, row_mirror leave the
// row sum in all 16 lanes; row_bcast15 / row_bcast31 carry it across the four rows into lane 63.
template <int CTRL, int ROW_MASK>
__device__ inline double dpp_get(double v) {
  int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
  int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ inline double row_sum16(double v) {        // every lane: sum over its row of 16 lanes
  v += dpp_get<0xB1, 0xf>(v);      // quad_perm [1,0,3,2]
  v += dpp_get<0x4E, 0xf>(v);      // quad_perm [2,3,0,1]
  v += dpp_get<0x141, 0xf>(v);     // row_half_mirror
  v += dpp_get<0x140, 0xf>(v);     // row_mirror
  return v;
}
__device__ inline double wave_sum(double v) {         // uniform result: sum over all 64 lanes
  v = row_sum16(v);
  v += dpp_get<0x142, 0xa>(v);     // row_bcast15 -> rows 1, 3
  v += dpp_get<0x143, 0xc>(v);     // row_bcast31 -> rows 2, 3
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63),
                          __builtin_amdgcn_readlane(__double2loint(v), 63));
}
#endif

void launch_rank(hipStream_t s, const double* f, int n, int maximize, long long* ranks);
// B > 1: batched launch over the B contexts of a batch (blockIdx.z = run, buffer strides zs / hzs bytes, see zrun)
struct ZB { int B = 1; size_t zs = 0, hzs = 0; };
void launch_wpca_prep(hipStream_t s, const double* X, const long long* ranks, const double* noise, int n, int d,
                      int DP, double* weights, double* data_mean, double* pca_mean, double* Wc, ZB zb = ZB());
void launch_cov(hipStream_t s, const double* Wc, int n, int DP, double* C, ZB zb = ZB());
// eigen-decomposition + selection (components sorted by variance, evr, k, sign rule) in one launch
void launch_jacobi(hipStream_t s, const double* C, int d, int DP, const double* V0, double* G, double* lam, int* sweeps,
                   int n, double var_threshold, int n_components, double* comps, double* evr, int* k_dev, HostMirror* hm,
                   ZB zb = ZB());
void launch_project(hipStream_t s, const double* X, const double* data_mean, const double* pca_mean,
                    const double* comps, const int* k_dev, int n, int d, double* Z, ZB zb = ZB());
void launch_zstats(hipStream_t s, const double* Z, const double* y, int n, int k, const double* user_norm_bounds,
                   double* bounds4 /*norm_lo,norm_hi,acq_lo,acq_hi each MAXD*/, double* zn_mean, double* ystats,
                   double* ys, HostMirror* hm, const int* k_dev = nullptr, ZB zb = ZB());
void launch_znorm(hipStream_t s, const double* Z, int n, int k, int NP, int KP, int ld, const double* bounds4,
                  const double* zn_mean, double inv_ls, double* ZnT, double* AT, double* nrm,
                  const int* k_dev = nullptr, ZB zb = ZB());   // k_dev != NULL: k (and KP) are read on the device, the arguments ignored
void launch_gram(hipStream_t s, const double* AT, const double* nrm, int n, int NP, int KP, int ld, double noise,
                 int kernel, double* K, const int* k_dev = nullptr, double* K2 = nullptr, int* info_reset = nullptr,
                 ZB zb = ZB());
void launch_add_jitter(hipStream_t s, double* K, int n, int ld, double jitter);
void launch_cholesky(hipStream_t s, double* L, int NP, int ld, int* info, double* diag_scratch, ZB zb = ZB());
void launch_trinv(hipStream_t s, const double* L, int NP, int ld, double* R, ZB zb = ZB());
void launch_chol_panel(hipStream_t s, double* L, int p, int nblocks, int ld, int* info, double* diag_scratch, ZB zb = ZB());
void launch_chol_steps(hipStream_t s, double* L, int NP, int ld, int* info, double* diag_scratch /* two tiles */, ZB zb = ZB());
void launch_trinv_diag_w(hipStream_t s, const double* L, int nblk, int ld, double* R, ZB zb = ZB());
void launch_alpha(hipStream_t s, const double* R, const double* ys, int n, int NP, int ld, double* tmp, double* alpha,
                  ZB zb = ZB());
void launch_acq(hipStream_t s, const QueryArgs* qa, const double* Xq, int q, int n, int k, int NP, int ld,
                const double* ZnT, const double* R, const double* alpha, const double* bounds4, const double* ystats,
                AcqParams p, double* partial, unsigned int* counters, double* val,
                double* grad, double* host_val, double* host_grad, HostMirror* hm, unsigned long long seq,
                MailPair* dev_mail = nullptr, MailPair* part_pairs = nullptr,
                AcqBatch ab = AcqBatch(), int B = 1, int table_entries = 0);
// throughput variant: one work-group per (restart group of <= 5 queries, 64-row slab); `tab` holds `entries` 32-bit words
// run << 16 | first query << 8 | count (see k_acq_group)
#define PCABO_GROUP_Q 5
#define PCABO_GROUP_CNT_OFFSET 8192      // its tickets live in the upper half of the counter array (other slab count)
bool acq_group_possible(int NP, int k);
int launch_acq_group(hipStream_t st, const QueryArgs* tab, int entries, const double* Xq, int n, int k, int NP, int ld,
                     const double* ZnT, const double* R, const double* alpha, const double* bounds4, const double* ystats,
                     AcqParams p, double* partial, unsigned int* counters, double* val, double* grad, double* host_val,
                     double* host_grad, HostMirror* hm, unsigned long long seq, AcqBatch ab);   // 0, or -1: nothing launched
// value-only scoring of a large batch as a GEMM (KS, then V = R KS^T on MFMA, then the scalar chain); KS: q x ld scratch
bool score_gemm_possible(int q);
void launch_score(hipStream_t st, const double* Xq, int q, int n, int k, int NP, int ld, const double* ZnT, const double* R,
                  const double* alpha, const double* bounds4, const double* ystats, AcqParams p, double* KS, double* partial,
                  double* val, AcqBatch ab = AcqBatch(), int B = 1);
// resident mode available for this shape? (fast path + every group of the grid co-resident)
bool acq_server_possible(int q, int n, int k, int NP);
int acq_slabs(int NP);
void launch_inverse_map(hipStream_t s, const double* z, const double* comps, const double* data_mean,
                        const double* pca_mean, int k, int d, double* x, const int* k_dev = nullptr, ZB zb = ZB(),
                        double* host_x = nullptr, HostMirror* hm = nullptr, unsigned long long seq = 0);   // host_x / hm: results + flag to pinned host
// Device-resident L-BFGS-B of a batch's restart groups (kernels_lbfgsb.hip): RT = transposed root inverse (zeros above the
// diagonal and beyond n), then one work-group per table entry (run << 16 | first query << 8 | count).  mode 1: optimise from the
// initial conditions in Xq ([num_restarts x k | lower k | upper k] per run), candidates to out_x, values / counters to out_v;
// mode 0: one value+gradient evaluation at the points in Xq.  Returns 0, or -1 when the launch could not be set up.
void launch_rt_build(hipStream_t s, const double* R, int n, int NP, int ld, double* RT, ZB zb = ZB());
bool lbfgsb_device_possible(int NP, int kmax, int batch_limit);
int launch_lbfgsb_group(hipStream_t st, const unsigned* table, int entries, int mode, int num_restarts, int maxiter, int n, int NP,
                        int ld, const double* Xq, const double* ZnT, const double* R, const double* RT, const double* alpha,
                        const double* bounds4, const double* ystats, const double* bestf, const int* k_dev, double inv_ls,
                        int maximize, int acq, int kernel, double* out_x, double* out_v, size_t zs);
