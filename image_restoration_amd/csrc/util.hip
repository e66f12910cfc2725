// Small shared kernels: deterministic row reduction, AdamW on flat buffers, casts, L1 loss.
#include <stdarg.h>
#include <stdlib.h>

#include <atomic>
#include <algorithm>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "common.h"

namespace mi {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---------------------------------------------------------------- A/B switches, read once (common.h: MI_ENV)
struct EnvTable { char val[E_ENV_COUNT][48]; bool set[E_ENV_COUNT]; };
static EnvTable g_env;
static std::once_flag g_env_once;
static void env_load() {
  static const char* const names[E_ENV_COUNT] = {
#define MI_ENV_NAME(n) #n,
      MI_ENV_LIST(MI_ENV_NAME)
#undef MI_ENV_NAME
  };
  for (int i = 0; i < E_ENV_COUNT; ++i) {
    const char* e = getenv(names[i]);
    g_env.set[i] = e != nullptr && e[0] != 0;          // set-but-empty counts as unset (the Python side reads it the same way)
    g_env.val[i][0] = 0;
    if (e) { strncpy(g_env.val[i], e, sizeof(g_env.val[i]) - 1); g_env.val[i][sizeof(g_env.val[i]) - 1] = 0; }
  }
}
const char* env_get(int id) {
  std::call_once(g_env_once, env_load);
  return (id >= 0 && id < E_ENV_COUNT && g_env.set[id]) ? g_env.val[id] : nullptr;
}

// ---------------------------------------------------------------- profiler
struct ProfRec { int kid; double bytes, flops; hipEvent_t e0, e1; };
static std::mutex g_prof_mu;
static std::vector<ProfRec> g_prof_recs;
static std::atomic<int> g_prof_on{0};
static const char* const g_kernel_names[K_COUNT] = {
    "ln_fwd", "ln_bwd", "dwconv_fwd", "dwconv_gate_fwd", "dwconv_bwd_data", "dwconv_gate_bwd_data", "dwconv_wgrad",
    "pw_gemm", "gram", "gram_reduce", "attn_fold", "attn_bwd_small", "reduce_rows", "chan_sum", "adamw", "cast", "l1_loss",
    "pw_pack", "gap", "im2col3x3", "col2im3x3", "gdfn_fused_fwd", "gdfn_fused_bwd", "mdta_fused_a", "fused_pack", "moe_route", "patch_circconv", "ewise", "conv3x3", "mdta_qk", "mdta_av", "bwd_tail", "bwd_tail_finish", "adair_fre"};

ProfScope::ProfScope(hipStream_t stream, int kernel_id, double bytes, double flops)
    : st(stream), kid(kernel_id), on(g_prof_on.load(std::memory_order_relaxed) != 0) {
  if (!on) return;
  ProfRec r{kernel_id, bytes, flops, nullptr, nullptr};
  if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) { on = false; return; }
  (void)hipEventRecord(r.e0, st);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_recs.push_back(r);
  kid = (int)g_prof_recs.size() - 1;  // reuse the field as the record index for the destructor
}
ProfScope::~ProfScope() {
  if (!on) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (kid >= 0 && kid < (int)g_prof_recs.size()) (void)hipEventRecord(g_prof_recs[kid].e1, st);
}

// out[g][c] = (accumulate ? out[g][c] : 0) + scale * sum_{r in group g} part[r*ld + c].  Rows are walked in a fixed
// order -> bitwise reproducible.  256 threads = 32 columns x 8 row-phases, 4 loads in flight per thread.
// Row group g = blockIdx.y covers rows [g*rpg, min(rows, (g+1)*rpg)) and writes out + g*out_gs.
__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                          int64_t rows, int64_t cols, int64_t ld, int accumulate,
                                                          float scale, int64_t rpg, int64_t out_gs,
                                                          float* __restrict__ out2, int64_t split) {
  __shared__ float sm[8][32];
  const int cx = threadIdx.x & 31, ph = threadIdx.x >> 5;
  const int64_t c = (int64_t)blockIdx.x * 32 + cx;
  const int64_t r_begin = (int64_t)blockIdx.y * rpg;
  int64_t r_end = r_begin + rpg;
  if (r_end > rows) r_end = rows;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (c < cols) {
    int64_t r = r_begin + ph;
    for (; r + 24 < r_end; r += 32) {
      a0 += part[r * ld + c];
      a1 += part[(r + 8) * ld + c];
      a2 += part[(r + 16) * ld + c];
      a3 += part[(r + 24) * ld + c];
    }
    for (; r < r_end; r += 8) a0 += part[r * ld + c];
  }
  sm[ph][cx] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (ph == 0 && c < cols) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += sm[k][cx];
    // columns [split, cols) go to a second tensor (two parameter gradients reduced by one launch)
    float* o = (out2 && c >= split) ? out2 + (c - split) : out + (int64_t)blockIdx.y * out_gs + c;
    *o = (accumulate ? *o : 0.f) + scale * t;
  }
}

// tmp (optional, REDUCE_GROUPS*cols floats): when given and rows is large the sum runs in two stages so that the
// long row walk is spread over REDUCE_GROUPS x (cols/32) workgroups instead of cols/32.
int launch_reduce_rows(const float* part, float* out, int64_t rows, int64_t cols, int64_t part_ld, int accumulate,
                       float scale, hipStream_t st, float* tmp, float* out2, int64_t split) {
  if (cols <= 0) return MI_OK;
  // partials a producer placed in the deferred arena: record the sum, mi_deferred_flush runs it (common.h)
  if (deferred_reduce_rows(part, out, rows, cols, part_ld, accumulate, scale, out2, split)) return MI_OK;
  if (tmp && rows > 4 * REDUCE_GROUPS) {
    const int64_t rpg = (rows + REDUCE_GROUPS - 1) / REDUCE_GROUPS;
    {
      ProfScope ps(st, K_REDUCE_ROWS, (double)(rows + REDUCE_GROUPS) * cols * 4, (double)rows * cols);
      hipLaunchKernelGGL(reduce_rows_kernel, dim3(cdiv(cols, 32), REDUCE_GROUPS), dim3(256), 0, st, part, tmp, rows, cols,
                         part_ld, 0, 1.0f, rpg, cols, (float*)nullptr, (int64_t)0);
    }
    MI_LAUNCH_CHECK();
    part = tmp; rows = REDUCE_GROUPS; part_ld = cols;
  }
  ProfScope ps(st, K_REDUCE_ROWS, (double)(rows + 1) * cols * 4, (double)rows * cols);
  hipLaunchKernelGGL(reduce_rows_kernel, dim3(cdiv(cols, 32), 1), dim3(256), 0, st, part, out, rows, cols, part_ld,
                     accumulate, scale, rows, (int64_t)0, out2, split);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

// ---------------------------------------------------------------- deferred parameter-gradient reductions (common.h)
struct DrJob {
  const float* part; float* tmp; float* out; float* out2;
  int64_t rows, cols, ld, split, rpg;
  int groups, accumulate; float scale;
  int blk1, blk2;                 // first workgroup of this job within its generation's launch / its generation
};
constexpr int DR_MAX_JOBS = 2048;
constexpr size_t DR_TABLE_BYTES = DR_MAX_JOBS * sizeof(DrJob);
static struct {
  std::mutex mu;
  bool active = false;
  bool recording = false;         // producers defer only while the owner says so (mi_deferred_record): its backward window
  char* arena = nullptr;
  size_t bytes = 0, off = 0;
  std::vector<DrJob> jobs;
  DrJob* host_tab[2] = {nullptr, nullptr};   // pinned staging for the job table, used alternately
  hipEvent_t copied[2] = {nullptr, nullptr}; // recorded behind each staging buffer's copy: waited for before its reuse
  int flip = 0;
  int nblk1 = 0, nblk2 = 0;
  size_t high = 0;                // high-water mark of the arena (diagnostics: mi_deferred_high_water)
  std::unordered_map<const float*, int> out_gen;   // pending outputs -> generation of their latest job
  std::vector<int> gen_blocks;    // workgroups per generation
} g_dr;

static bool stream_capturing(hipStream_t st) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return false; }
  return cs != hipStreamCaptureStatusNone;
}

float* deferred_take(size_t nfloats, hipStream_t st) {
  {
    std::lock_guard<std::mutex> lk(g_dr.mu);
    if (!g_dr.active || !g_dr.recording) return nullptr;
  }
  if (stream_capturing(st)) return nullptr;              // a captured step sums at once: the flush is not capturable
  std::lock_guard<std::mutex> lk(g_dr.mu);
  if (!g_dr.active || !g_dr.recording || g_dr.jobs.size() + 1 >= (size_t)DR_MAX_JOBS) return nullptr;
  const size_t need = align_up(nfloats * sizeof(float), 256);
  if (g_dr.off + need > g_dr.bytes) return nullptr;
  float* p = reinterpret_cast<float*>(g_dr.arena + g_dr.off);
  g_dr.off += need;
  if (g_dr.off > g_dr.high) g_dr.high = g_dr.off;
  return p;
}
bool deferred_owns(const void* p) {
  std::lock_guard<std::mutex> lk(g_dr.mu);
  return g_dr.active && (const char*)p >= g_dr.arena + DR_TABLE_BYTES && (const char*)p < g_dr.arena + g_dr.bytes;
}
bool deferred_reduce_rows(const float* part, float* out, int64_t rows, int64_t cols, int64_t part_ld, int accumulate, float scale,
                          float* out2, int64_t split) {
  if (cols <= 0 || rows <= 0) return false;
  if (!deferred_owns(part)) return false;
  DrJob j;
  j.part = part; j.out = out; j.out2 = out2; j.rows = rows; j.cols = cols; j.ld = part_ld; j.split = split;
  j.accumulate = accumulate; j.scale = scale;
  j.groups = 1; j.rpg = rows; j.tmp = nullptr;
  std::lock_guard<std::mutex> lk(g_dr.mu);
  if (!g_dr.active || g_dr.jobs.size() >= (size_t)DR_MAX_JOBS) return false;
  // generation: one past the latest pending job that writes the same gradient (a plain read-modify-write per job, and
  // workgroups of one launch run concurrently: same-output jobs must be separate, ordered launches)
  int gen = 0;
  for (const float* o : {(const float*)out, (const float*)out2}) {
    if (!o) continue;
    auto it = g_dr.out_gen.find(o);
    if (it != g_dr.out_gen.end() && it->second + 1 > gen) gen = it->second + 1;
  }
  g_dr.out_gen[out] = gen;
  if (out2) g_dr.out_gen[out2] = gen;
  if ((int)g_dr.gen_blocks.size() <= gen) g_dr.gen_blocks.resize(gen + 1, 0);
  j.blk2 = gen;
  j.blk1 = g_dr.gen_blocks[gen];
  g_dr.gen_blocks[gen] += (int)cdiv(cols, 256);
  g_dr.nblk1 += (int)cdiv(cols, 256);
  g_dr.jobs.push_back(j);
  return true;
}

// One workgroup = 256 consecutive columns of one job, one column per thread, the rows walked in order with eight loads in
// flight (eight partial sums combined in a fixed pattern: reproducible).  Coalesced 1-KiB row segments; the job is found by a
// binary search over the jobs' first-workgroup indices.
__global__ __launch_bounds__(256) void deferred_reduce_kernel(const DrJob* __restrict__ jobs, int njobs) {
  const int bid = blockIdx.x;
  int lo = 0, hi = njobs - 1;                            // the last job whose first workgroup is <= bid
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].blk1 <= bid) lo = mid; else hi = mid - 1;
  }
  const DrJob j = jobs[lo];
  const int64_t c = (int64_t)(bid - j.blk1) * 256 + threadIdx.x;
  if (c >= j.cols) return;
  const float* p = j.part + c;
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int64_t r = 0;
  for (; r + 7 < j.rows; r += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] += p[(r + u) * j.ld];
  }
  for (; r < j.rows; ++r) a[0] += p[r * j.ld];
  const float t = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  float* o = (j.out2 && c >= j.split) ? j.out2 + (c - j.split) : j.out + c;
  *o = (j.accumulate ? *o : 0.f) + j.scale * t;
}

// AdamW, torch semantics (decoupled weight decay, bias-corrected moments).
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, int64_t n, float lr,
                                                    float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt,
                                                    float gscale, const float* __restrict__ dev_scalars) {
  if (dev_scalars) { lr = dev_scalars[0]; bc1 = dev_scalars[1]; bc2_sqrt = dev_scalars[2]; }
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 3 < n) {
      f32x4 pv = *reinterpret_cast<f32x4*>(p + i), gv = *reinterpret_cast<const f32x4*>(g + i);
      f32x4 mv = *reinterpret_cast<f32x4*>(m + i), vv = *reinterpret_cast<f32x4*>(v + i);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float gr = gv[j] * gscale;
        pv[j] *= 1.0f - lr * wd;
        mv[j] = b1 * mv[j] + (1.0f - b1) * gr;
        vv[j] = b2 * vv[j] + (1.0f - b2) * gr * gr;
        const float denom = sqrtf(vv[j]) / bc2_sqrt + eps;
        pv[j] -= (lr / bc1) * (mv[j] / denom);
      }
      *reinterpret_cast<f32x4*>(p + i) = pv;
      *reinterpret_cast<f32x4*>(m + i) = mv;
      *reinterpret_cast<f32x4*>(v + i) = vv;
    } else {
      for (int64_t k = i; k < n; ++k) {
        const float gr = g[k] * gscale;
        float pv = p[k] * (1.0f - lr * wd);
        const float mv = b1 * m[k] + (1.0f - b1) * gr;
        const float vv = b2 * v[k] + (1.0f - b2) * gr * gr;
        pv -= (lr / bc1) * (mv / (sqrtf(vv) / bc2_sqrt + eps));
        p[k] = pv; m[k] = mv; v[k] = vv;
      }
    }
  }
}

// Global average pool of the MoCE router (moce_ir.py:703-707 AdaptiveAvgPool2d(1)): one workgroup per (image, channel).
template <typename T>
__global__ __launch_bounds__(256) void gap_fwd_kernel(const T* __restrict__ x, float* __restrict__ out, int64_t N) {
  __shared__ float sm[4];
  const T* row = x + (int64_t)blockIdx.x * N;
  float acc = 0.f;
  for (int64_t n = threadIdx.x; n < N; n += 256) acc += to_f32(row[n]);
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = ((sm[0] + sm[1]) + (sm[2] + sm[3])) / (float)N;
}
// dx[b,c,:] = dout[b,c] / N
template <typename T>
__global__ __launch_bounds__(256) void gap_bwd_kernel(const float* __restrict__ dout, T* __restrict__ dx, int64_t N) {
  const float v = dout[blockIdx.x] / (float)N;
  T* row = dx + (int64_t)blockIdx.x * N;
  for (int64_t n = threadIdx.x; n < N; n += 256) row[n] = Cvt<T>::from(v);
}

template <typename S, typename D>
__global__ __launch_bounds__(256) void cast_kernel(const S* __restrict__ s, D* __restrict__ d, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) d[i] = Cvt<D>::from(to_f32(s[i]));
}

// per-block partial of sum|a-b| into part[blockIdx.x]; da = sign(a-b)*scale
template <typename T>
__global__ __launch_bounds__(256) void l1_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ da,
                                                 float* __restrict__ part, int64_t n, float scale) {
  __shared__ float sm[4];
  float acc = 0.f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float d = to_f32(a[i]) - to_f32(b[i]);
    acc += fabsf(d);
    if (da) da[i] = Cvt<T>::from(d > 0.f ? scale : (d < 0.f ? -scale : 0.f));
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

}  // namespace mi

using namespace mi;

extern "C" int mi_version(void) { return MI_RESTORE_VERSION; }

extern "C" int mi_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& r : g_prof_recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  g_prof_recs.clear();
  g_prof_on.store(on ? 1 : 0);
  return MI_OK;
}
extern "C" int mi_prof_kernel_count(void) { return K_COUNT; }
extern "C" const char* mi_prof_kernel_name(int kid) { return (kid >= 0 && kid < K_COUNT) ? g_kernel_names[kid] : ""; }
extern "C" int mi_prof_collect(double* ms, double* bytes, double* flops, int64_t* launches, int n) {
  MI_CHECK_ARG(ms && bytes && flops && launches && n >= K_COUNT, "prof_collect: need arrays of %d entries", K_COUNT);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (int i = 0; i < n; ++i) { ms[i] = 0; bytes[i] = 0; flops[i] = 0; launches[i] = 0; }
  // MI_PROF_TRACE=<file>: also append one line per launch (kernel, algorithmic bytes, flops, ms) - the bytes/flops pair
  // identifies the shape, which is how the per-shape tables in profiles/ are made
  const char* trace_path = getenv("MI_PROF_TRACE");
  FILE* trace = trace_path ? fopen(trace_path, "a") : nullptr;
  for (auto& r : g_prof_recs) {
    MI_CHECK_HIP(hipEventSynchronize(r.e1));
    float t = 0.f;
    MI_CHECK_HIP(hipEventElapsedTime(&t, r.e0, r.e1));
    ms[r.kid] += t; bytes[r.kid] += r.bytes; flops[r.kid] += r.flops; launches[r.kid] += 1;
    if (trace) fprintf(trace, "%s %.0f %.0f %.6f\n", g_kernel_names[r.kid], r.bytes, r.flops, (double)t);
  }
  if (trace) fclose(trace);
  return MI_OK;
}
extern "C" const char* mi_last_error(void) { return g_err; }
extern "C" int mi_deferred_begin(void* arena, size_t bytes) {
  MI_CHECK_ARG(arena && aligned16(arena) && bytes >= mi::DR_TABLE_BYTES + (1u << 20), "deferred_begin: arena too small (job table + 1 MiB)");
  std::lock_guard<std::mutex> lk(mi::g_dr.mu);
  MI_CHECK_ARG(!mi::g_dr.active, "deferred_begin: already active (one deferral context per process)");
  for (int i = 0; i < 2; ++i)
    if (!mi::g_dr.host_tab[i]) {
      MI_CHECK_HIP(hipHostMalloc((void**)&mi::g_dr.host_tab[i], mi::DR_TABLE_BYTES, hipHostMallocDefault));
      MI_CHECK_HIP(hipEventCreateWithFlags(&mi::g_dr.copied[i], hipEventDisableTiming));
      MI_CHECK_HIP(hipEventRecord(mi::g_dr.copied[i], nullptr));
    }
  mi::g_dr.arena = (char*)arena; mi::g_dr.bytes = bytes; mi::g_dr.off = mi::DR_TABLE_BYTES;
  mi::g_dr.jobs.clear(); mi::g_dr.nblk1 = mi::g_dr.nblk2 = 0;
  mi::g_dr.out_gen.clear(); mi::g_dr.gen_blocks.clear();
  mi::g_dr.active = true;
  mi::g_dr.recording = false;
  return MI_OK;
}
extern "C" int mi_deferred_record(int on) {
  std::lock_guard<std::mutex> lk(mi::g_dr.mu);
  MI_CHECK_ARG(mi::g_dr.active || !on, "deferred_record: no context (mi_deferred_begin)");
  mi::g_dr.recording = on != 0;
  return MI_OK;
}
extern "C" size_t mi_deferred_high_water(void) {
  std::lock_guard<std::mutex> lk(mi::g_dr.mu);
  return mi::g_dr.high;
}
extern "C" int mi_deferred_pending(void) {
  std::lock_guard<std::mutex> lk(mi::g_dr.mu);
  return mi::g_dr.active ? (int)mi::g_dr.jobs.size() : 0;
}
extern "C" int mi_deferred_flush(void* stream) {
  hipStream_t st = (hipStream_t)stream;
  std::lock_guard<std::mutex> lk(mi::g_dr.mu);
  if (!mi::g_dr.active || mi::g_dr.jobs.empty()) { if (mi::g_dr.active) mi::g_dr.off = mi::DR_TABLE_BYTES; return MI_OK; }
  if (mi::stream_capturing(st)) {
    mi::set_error("deferred_flush: %d recorded gradient sums are pending and the stream is being captured into a HIP graph; flush "
                  "before the capture begins (FlatTrainer.zero_grad / reduce_gradients do)", (int)mi::g_dr.jobs.size());
    return MI_ERR_ARG;
  }
  const int n = (int)mi::g_dr.jobs.size();
  const int ngen = (int)mi::g_dr.gen_blocks.size();
  if (ngen > 1)                                           // the table goes out grouped by generation (stable: blk1 stays sorted)
    std::stable_sort(mi::g_dr.jobs.begin(), mi::g_dr.jobs.end(), [](const mi::DrJob& x, const mi::DrJob& y) { return x.blk2 < y.blk2; });
  // two pinned staging tables used alternately: the one taken now was last copied from two flushes ago
  const int f = mi::g_dr.flip;
  mi::g_dr.flip ^= 1;
  MI_CHECK_HIP(hipEventSynchronize(mi::g_dr.copied[f]));
  memcpy(mi::g_dr.host_tab[f], mi::g_dr.jobs.data(), (size_t)n * sizeof(mi::DrJob));
  MI_CHECK_HIP(hipMemcpyAsync(mi::g_dr.arena, mi::g_dr.host_tab[f], (size_t)n * sizeof(mi::DrJob), hipMemcpyHostToDevice, st));
  MI_CHECK_HIP(hipEventRecord(mi::g_dr.copied[f], st));
  double bytes = 0;
  for (const auto& j : mi::g_dr.jobs) bytes += 4.0 * (double)(j.rows + 1) * j.cols;
  for (int g = 0, j0 = 0; g < ngen; ++g) {                // one launch per generation, in order
    int j1 = j0;
    while (j1 < n && mi::g_dr.jobs[j1].blk2 == g) ++j1;
    if (j1 > j0) {
      ProfScope ps(st, K_REDUCE_ROWS, bytes / ngen, bytes / 4.0 / ngen);
      hipLaunchKernelGGL(mi::deferred_reduce_kernel, dim3(mi::g_dr.gen_blocks[g]), dim3(256), 0, st,
                         (const mi::DrJob*)mi::g_dr.arena + j0, j1 - j0);
      MI_LAUNCH_CHECK();
    }
    j0 = j1;
  }
  mi::g_dr.jobs.clear(); mi::g_dr.nblk1 = mi::g_dr.nblk2 = 0; mi::g_dr.off = mi::DR_TABLE_BYTES;
  mi::g_dr.out_gen.clear(); mi::g_dr.gen_blocks.clear();
  return MI_OK;
}
extern "C" int mi_deferred_end(void) {
  std::lock_guard<std::mutex> lk(mi::g_dr.mu);
  MI_CHECK_ARG(mi::g_dr.jobs.empty(), "deferred_end: %d reductions still pending (call mi_deferred_flush first)", (int)mi::g_dr.jobs.size());
  mi::g_dr.active = false; mi::g_dr.recording = false; mi::g_dr.arena = nullptr; mi::g_dr.bytes = mi::g_dr.off = 0;
  return MI_OK;
}
extern "C" int mi_env_reload(void) {
  std::call_once(mi::g_env_once, mi::env_load);
  mi::env_load();
  return MI_OK;
}

extern "C" int mi_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                             float eps, float weight_decay, int step, float grad_scale, const float* dev_scalars,
                             void* stream) {
  MI_CHECK_ARG(p && g && m && v && n > 0 && (step >= 1 || dev_scalars), "adamw: bad arguments");
  if (step < 1) step = 1;
  MI_CHECK_ARG(aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v), "adamw: buffers must be 16-byte aligned");
  const float bc1 = 1.0f - powf(beta1, (float)step);
  const float bc2 = sqrtf(1.0f - powf(beta2, (float)step));
  int blocks = cdiv(n, 256 * 4);
  if (blocks > 2048) blocks = 2048;
  ProfScope ps((hipStream_t)stream, K_ADAMW, 28.0 * n, 12.0 * n);
  hipLaunchKernelGGL(adamw_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2, eps,
                     weight_decay, bc1, bc2, grad_scale, dev_scalars);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_gap_fwd(const void* x, float* out, int B, int C, int64_t N, int dtype, void* stream) {
  MI_CHECK_ARG(x && out && B > 0 && C > 0 && N > 0, "gap_fwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(st, K_GAP, (double)B * C * N * dtype_size(dtype), (double)B * C * N);
  if (dtype == MI_F32) hipLaunchKernelGGL((gap_fwd_kernel<float>), dim3(B * C), dim3(256), 0, st, (const float*)x, out, N);
  else if (dtype == MI_BF16) hipLaunchKernelGGL((gap_fwd_kernel<bf16>), dim3(B * C), dim3(256), 0, st, (const bf16*)x, out, N);
  else { set_error("gap_fwd: bad dtype"); return MI_ERR_ARG; }
  MI_LAUNCH_CHECK();
  return MI_OK;
}
extern "C" int mi_gap_bwd(const float* dout, void* dx, int B, int C, int64_t N, int dtype, void* stream) {
  MI_CHECK_ARG(dout && dx && B > 0 && C > 0 && N > 0, "gap_bwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(st, K_GAP, (double)B * C * N * dtype_size(dtype), 0.0);
  if (dtype == MI_F32) hipLaunchKernelGGL((gap_bwd_kernel<float>), dim3(B * C), dim3(256), 0, st, dout, (float*)dx, N);
  else if (dtype == MI_BF16) hipLaunchKernelGGL((gap_bwd_kernel<bf16>), dim3(B * C), dim3(256), 0, st, dout, (bf16*)dx, N);
  else { set_error("gap_bwd: bad dtype"); return MI_ERR_ARG; }
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_cast(const void* src, int sdt, void* dst, int ddt, int64_t n, void* stream) {
  MI_CHECK_ARG(src && dst && n > 0, "cast: bad arguments");
  int blocks = cdiv(n, 256);
  if (blocks > 4096) blocks = 4096;
  hipStream_t st = (hipStream_t)stream;
  if (sdt == MI_F32 && ddt == MI_BF16)
    hipLaunchKernelGGL((cast_kernel<float, bf16>), dim3(blocks), dim3(256), 0, st, (const float*)src, (bf16*)dst, n);
  else if (sdt == MI_BF16 && ddt == MI_F32)
    hipLaunchKernelGGL((cast_kernel<bf16, float>), dim3(blocks), dim3(256), 0, st, (const bf16*)src, (float*)dst, n);
  else if (sdt == MI_F32 && ddt == MI_F32)
    hipLaunchKernelGGL((cast_kernel<float, float>), dim3(blocks), dim3(256), 0, st, (const float*)src, (float*)dst, n);
  else if (sdt == MI_BF16 && ddt == MI_BF16)
    hipLaunchKernelGGL((cast_kernel<bf16, bf16>), dim3(blocks), dim3(256), 0, st, (const bf16*)src, (bf16*)dst, n);
  else { set_error("cast: bad dtypes %d -> %d", sdt, ddt); return MI_ERR_ARG; }
  MI_LAUNCH_CHECK();
  return MI_OK;
}

// loss[0] += mean|a-b| ; uses loss[1..] as scratch: caller passes a float buffer of >= 1+1024 entries
extern "C" int mi_l1_loss(const void* a, const void* b, void* da, float* loss, int64_t n, float scale, int dtype,
                          void* stream) {
  MI_CHECK_ARG(a && b && loss && n > 0, "l1_loss: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  int blocks = cdiv(n, 256 * 8);
  if (blocks > 1024) blocks = 1024;
  if (dtype == MI_F32)
    hipLaunchKernelGGL((l1_kernel<float>), dim3(blocks), dim3(256), 0, st, (const float*)a, (const float*)b, (float*)da,
                       loss + 1, n, scale);
  else if (dtype == MI_BF16)
    hipLaunchKernelGGL((l1_kernel<bf16>), dim3(blocks), dim3(256), 0, st, (const bf16*)a, (const bf16*)b, (bf16*)da,
                       loss + 1, n, scale);
  else { set_error("l1_loss: bad dtype"); return MI_ERR_ARG; }
  MI_LAUNCH_CHECK();
  return launch_reduce_rows(loss + 1, loss, blocks, 1, 1, 1, 1.0f / (float)n, st);
}
