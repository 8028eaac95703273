// SE(3) diffuser on the device: IGSO(3) tables, prior sampling, forward marginal, reverse step.
//
// Reference (all host numpy/scipy float64 there, with a device->host->device round trip per step):
//   tables           data/so3_diffuser.py:122-180 (+ :9-49 expansion, :52-68 density, :71-117 score)
//   sample_ref       data/se3_diffuser.py:216-268, so3_diffuser.py:215-251, r3_diffuser.py:39-40
//   forward_marginal data/se3_diffuser.py:43-110, so3_diffuser.py:311-328, r3_diffuser.py:81-101
//   reverse          data/se3_diffuser.py:160-214, so3_diffuser.py:330-366 (+ :201-209 g(t)),
//                    r3_diffuser.py:106-146; rotvec composition data/utils.py:184-195 (scipy Rotation)
// Random numbers are INPUTS (standard normal / uniform draws in the reference's call order), so
// the same numpy stream -- or any device generator -- can drive them: identical noised inputs give
// identical frames.  Arithmetic is fp64 like the reference; frames are stored fp32 [.., 7] =
// (qw, qx, qy, qz, tx, ty, tz) in Angstrom.  Rotations are composed as unit quaternions
// (R_t Exp(v) == q_t (x) exp_q(v)), which removes the reference's matrix -> eigh -> quaternion step.
#include "fd_common.h"
#include "../../include/fd_hip.h"

namespace {

// (f, f') of the truncated IGSO(3) series, see fd_frames.hip::igso3_series
__device__ __forceinline__ void igso3_f_df(double om, double sg, int L, double* f, double* df) {
  const double lo = sin(0.5 * om), dlo = 0.5 * cos(0.5 * om);
  const double so = sin(om), co = cos(om);
  double s = lo, c = 2.0 * dlo;
  const double q = exp(-sg * sg);
  double w = 1.0, r = q;
  double F = 0.0, D = 0.0;
  const double ilo = 1.0 / lo, ilo2 = ilo * ilo;
  for (int l = 0; l < L; ++l) {
    const double a = (double)l + 0.5;
    const double cw = (double)(2 * l + 1) * w;
    F += cw * s * ilo;
    D += cw * (lo * a * c - s * dlo) * ilo2;
    w *= r;
    r *= q;
    if (w == 0.0) break;
    const double s2 = s * co + c * so;
    c = c * co - s * so;
    s = s2;
  }
  *f = F; *df = D;
}

// one thread per (sigma, omega) entry
__global__ __launch_bounds__(256) void igso3_tables_kernel(const double* __restrict__ sigma,
                                                           const double* __restrict__ omega, int ns, int no, int L,
                                                           double* __restrict__ pdf, double* __restrict__ score_norms) {
  const long total = (long)ns * no;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int si = (int)(e / no), oi = (int)(e % no);
    const double om = omega[oi];
    double f, df;
    igso3_f_df(om, sigma[si], L, &f, &df);
    pdf[e] = f * (1.0 - cos(om)) / 3.14159265358979323846;   // density(marginal=True) :52-68
    score_norms[e] = df / (f + 1e-4);                        // score() :117
  }
}

// cdf[s, :] = cumsum(pdf[s, :]) / no * pi  (:161-162); one thread per sigma row
__global__ __launch_bounds__(64) void igso3_cdf_kernel(const double* __restrict__ pdf, int ns, int no,
                                                       double* __restrict__ cdf) {
  const int si = (int)(blockIdx.x * 64 + threadIdx.x);
  if (si >= ns) return;
  double acc = 0.0;
  for (int o = 0; o < no; ++o) {
    acc += pdf[(long)si * no + o];
    cdf[(long)si * no + o] = acc / (double)no * 3.14159265358979323846;
  }
}

// scipy Rotation.from_rotvec -> quaternion (w, x, y, z)
__device__ __forceinline__ void rotvec_to_quat(const double* v, double* q) {
  const double a2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
  const double ang = sqrt(a2);
  double sc;
  if (ang <= 1e-3) sc = 0.5 - a2 / 48.0 + a2 * a2 / 3840.0;
  else sc = sin(0.5 * ang) / ang;
  q[0] = cos(0.5 * ang);
  q[1] = sc * v[0]; q[2] = sc * v[1]; q[3] = sc * v[2];
}

__device__ __forceinline__ void quat_mul(const double* p, const double* q, double* o) {
  o[0] = p[0] * q[0] - p[1] * q[1] - p[2] * q[2] - p[3] * q[3];
  o[1] = p[0] * q[1] + p[1] * q[0] + p[2] * q[3] - p[3] * q[2];
  o[2] = p[0] * q[2] - p[1] * q[3] + p[2] * q[0] + p[3] * q[1];
  o[3] = p[0] * q[3] + p[1] * q[2] - p[2] * q[1] + p[3] * q[0];
}

__device__ __forceinline__ void quat_normalize(double* q) {
  const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}

// np.interp(x, xp, fp) for increasing xp
__device__ __forceinline__ double interp(double x, const double* __restrict__ xp, const double* __restrict__ fp,
                                         int n) {
  if (x <= xp[0]) return fp[0];
  if (x >= xp[n - 1]) return fp[n - 1];
  int lo = 0, hi = n - 1;   // xp[lo] <= x < xp[hi]
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (xp[mid] <= x) lo = mid; else hi = mid;
  }
  const double slope = (fp[lo + 1] - fp[lo]) / (xp[lo + 1] - xp[lo]);
  return slope * (x - xp[lo]) + fp[lo];
}

// sampled IGSO(3) rotation vector: normalised Gaussian axis * inverse-CDF angle (so3_diffuser.py:215-248)
__device__ __forceinline__ void sample_rotvec(const double* z, double u, const double* __restrict__ cdf_row,
                                              const double* __restrict__ omega, int no, double* v) {
  const double n = sqrt(z[0] * z[0] + z[1] * z[1] + z[2] * z[2]);
  const double ang = interp(u, cdf_row, omega, no);
  v[0] = z[0] / n * ang; v[1] = z[1] / n * ang; v[2] = z[2] / n * ang;
}

__global__ __launch_bounds__(256) void sample_ref_kernel(const double* __restrict__ z_axis,
                                                         const double* __restrict__ u,
                                                         const double* __restrict__ z_trans,
                                                         const double* __restrict__ cdf_row,
                                                         const double* __restrict__ omega, int no, double cs,
                                                         float* __restrict__ out, long n) {
  for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < n; r += (long)gridDim.x * 256) {
    double v[3], q[4];
    sample_rotvec(z_axis + r * 3, u[r], cdf_row, omega, no, v);
    rotvec_to_quat(v, q);
    float* o = out + r * 7;
    o[0] = (float)q[0]; o[1] = (float)q[1]; o[2] = (float)q[2]; o[3] = (float)q[3];
    o[4] = (float)(z_trans[r * 3 + 0] / cs);
    o[5] = (float)(z_trans[r * 3 + 1] / cs);
    o[6] = (float)(z_trans[r * 3 + 2] / cs);
  }
}

// one residue of forward_marginal at (sigma, beta): sampled rotation, noised frame, DSM targets
__device__ __forceinline__ void forward_marginal_one(
    long r, const float* __restrict__ rig0, const double* __restrict__ z_axis, const double* __restrict__ u,
    const double* __restrict__ z_trans, const double* __restrict__ cdf_row, const double* __restrict__ omega, int no,
    const double* __restrict__ score_row, double sigma, double e1, double cv, double sd, double cs, int L,
    const float* __restrict__ mask, float* __restrict__ rig_t, double* __restrict__ rot_score,
    double* __restrict__ trans_score) {
  const float* q0f = rig0 + r * 7;
  const double m = mask ? (double)mask[r] : 1.0;
  double v[3], qe[4], q0[4] = {q0f[0], q0f[1], q0f[2], q0f[3]}, qt[4];
  sample_rotvec(z_axis + r * 3, u[r], cdf_row, omega, no, v);
  // score of the SAMPLED rotation vector (so3_diffuser.py:324 -> :274-305 with float64 vec)
  const double vn = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  const double om = vn + 1e-6;
  double g;
  if (score_row) {
    // use_cached_score (so3_diffuser.py:293-299): bucketize(om, omega[:-1]) into this sigma's score_norms row
    int lo = 0, hi = no - 1;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (omega[mid] < om) lo = mid + 1; else hi = mid;
    }
    g = score_row[lo];
  } else {
    double f, df;
    igso3_f_df(om, sigma, L, &f, &df);
    g = df / (f + 1e-4);
  }
  const double sc = g / (om + 1e-6);
  rotvec_to_quat(v, qe);
  quat_normalize(q0);
  quat_mul(q0, qe, qt);          // right multiply: R_0 Exp(v)
  quat_normalize(qt);
  float* o = rig_t + r * 7;
  const bool diff = m > 0.5;
  for (int k = 0; k < 4; ++k) o[k] = diff ? (float)qt[k] : q0f[k];
  for (int k = 0; k < 3; ++k) {
    rot_score[r * 3 + k] = diff ? sc * v[k] : 0.0;
    const double x0 = (double)q0f[4 + k] * cs;
    const double xt = e1 * x0 + sd * z_trans[r * 3 + k];
    trans_score[r * 3 + k] = diff ? -(xt - e1 * x0) / cv : 0.0;
    o[4 + k] = diff ? (float)(xt / cs) : q0f[4 + k];
  }
}

__global__ __launch_bounds__(256) void forward_marginal_kernel(
    const float* __restrict__ rig0, const double* __restrict__ z_axis, const double* __restrict__ u,
    const double* __restrict__ z_trans, const double* __restrict__ cdf_row, const double* __restrict__ omega, int no,
    const double* __restrict__ score_row, double sigma, double beta, double cs, int L, const float* __restrict__ mask,
    float* __restrict__ rig_t,
    double* __restrict__ rot_score, double* __restrict__ trans_score, long n) {
  const double e1 = exp(-0.5 * beta), cv = 1.0 - exp(-beta), sd = sqrt(cv);
  for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < n; r += (long)gridDim.x * 256)
    forward_marginal_one(r, rig0, z_axis, u, z_trans, cdf_row, omega, no, score_row, sigma, e1, cv, sd, cs, L, mask,
                         rig_t, rot_score, trans_score);
}

// a training batch: example b = blockIdx.y has its own time, i.e. its own (sigma bin, sigma, marginal beta) triple
__global__ __launch_bounds__(256) void forward_marginal_batch_kernel(
    const float* __restrict__ rig0, const double* __restrict__ z_axis, const double* __restrict__ u,
    const double* __restrict__ z_trans, const double* __restrict__ cdf, const double* __restrict__ omega, int no,
    const double* __restrict__ score_norms, const double* __restrict__ tparams, double cs, int L,
    const float* __restrict__ mask, float* __restrict__ rig_t, double* __restrict__ rot_score,
    double* __restrict__ trans_score, int N) {
  const int b = (int)blockIdx.y;
  const long row = (long)tparams[b * 3 + 0];
  const double sigma = tparams[b * 3 + 1], beta = tparams[b * 3 + 2];
  const double e1 = exp(-0.5 * beta), cv = 1.0 - exp(-beta), sd = sqrt(cv);
  for (int n = (int)(blockIdx.x * 256 + threadIdx.x); n < N; n += (int)gridDim.x * 256)
    forward_marginal_one((long)b * N + n, rig0, z_axis, u, z_trans, cdf + row * no, omega, no,
                         score_norms ? score_norms + row * no : nullptr, sigma, e1, cv, sd, cs, L, mask, rig_t,
                         rot_score, trans_score);
}

// one block per batch element (centering needs the mean over its N residues).  ST = the scores' storage type: the network writes
// float32, the reference widens them to float64 before the step -- a float32 argument widened in registers is the same number
// and saves the two conversion launches of every reverse step.  out may alias rig_t (in place): every read of a row's
// translation for the mean happens before the barrier, after it each thread reads and writes only its own rows.
template <typename RT, typename TT>
__global__ __launch_bounds__(256) void reverse_step_kernel(
    const float* rig_t, const RT* __restrict__ rot_score, const TT* __restrict__ trans_score,
    const double* __restrict__ z_rot, const double* __restrict__ z_trans, const float* __restrict__ mask, int N,
    double g_rot, double b_t, const double* __restrict__ tparams, double dt, double noise_scale, double cs, int center,
    int diffuse_rot, int diffuse_trans, float* out) {
  __shared__ double red[4][3];
  __shared__ double com[3];
  const int b = (int)blockIdx.x;
  const int tid = (int)threadIdx.x;
  if (tparams) { g_rot = tparams[0]; b_t = tparams[1]; }   // time-dependent scalars from HBM (graph replay)
  const double sdt = sqrt(dt), gb = sqrt(b_t);
  double acc[3] = {0.0, 0.0, 0.0};
  // pass 1: translations (stored temporarily in out[4:7] as fp64-rounded-to-fp32 is NOT acceptable for the
  // mean, so the mean is accumulated here in fp64 and x' recomputed in pass 2)
  for (int n = tid; n < N; n += 256) {
    const long r = (long)b * N + n;
    for (int k = 0; k < 3; ++k) {
      const double x = (double)rig_t[r * 7 + 4 + k] * cs;
      const double f = -0.5 * b_t * x;
      const double pert = (f - gb * gb * (double)trans_score[r * 3 + k]) * dt + gb * sdt * (noise_scale * z_trans[r * 3 + k]);
      acc[k] += x - pert;
    }
  }
  for (int k = 0; k < 3; ++k) acc[k] = fd::wave_sum(acc[k]);
  if (fd::lane_id() == 0)
    for (int k = 0; k < 3; ++k) red[fd::wave_id()][k] = acc[k];
  __syncthreads();
  if (tid < 3) com[tid] = center ? (red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid]) / (double)N : 0.0;
  __syncthreads();
  for (int n = tid; n < N; n += 256) {
    const long r = (long)b * N + n;
    float in[7];
    for (int k = 0; k < 7; ++k) in[k] = rig_t[r * 7 + k];
    float* o = out + r * 7;
    const bool diff = mask ? mask[r] > 0.5f : true;
    // rotation: R' = R_t Exp(g^2 s dt + g sqrt(dt) z)
    if (diff && diffuse_rot) {
      double v[3], qe[4], qt[4] = {in[0], in[1], in[2], in[3]}, qn[4];
      for (int k = 0; k < 3; ++k)
        v[k] = g_rot * g_rot * (double)rot_score[r * 3 + k] * dt + g_rot * sdt * (noise_scale * z_rot[r * 3 + k]);
      rotvec_to_quat(v, qe);
      quat_normalize(qt);
      quat_mul(qt, qe, qn);
      quat_normalize(qn);
      for (int k = 0; k < 4; ++k) o[k] = (float)qn[k];
    } else {
      for (int k = 0; k < 4; ++k) o[k] = in[k];
    }
    for (int k = 0; k < 3; ++k) {
      if (diff && diffuse_trans) {
        const double x = (double)in[4 + k] * cs;
        const double f = -0.5 * b_t * x;
        const double pert = (f - gb * gb * (double)trans_score[r * 3 + k]) * dt + gb * sdt * (noise_scale * z_trans[r * 3 + k]);
        o[4 + k] = (float)((x - pert - com[k]) / cs);
      } else {
        o[4 + k] = in[4 + k];
      }
    }
  }
}

// First node of a captured diffusion step (sampler.sample, one hipGraph for all num_t steps): the step index lives in a device
// counter, so the captured launch sequence needs nothing from the host between replays -- it sets the network's time input
// t[B] = all_t[idx], the reverse step's scalars {g_rot(t), b(t)} = all_tp[idx] and copies the step's normal draws (rotation then
// translation, se3_diffuser.py:213-262) out of a buffer that holds K steps' worth (refilled by ONE generator launch every K
// steps), then advances the counter.  Replaces three launches per step (fill_, copy_, normal_) in front of every replay.
__global__ __launch_bounds__(1024) void sample_advance_kernel(int* __restrict__ counter, const float* __restrict__ all_t,
                                                              const double* __restrict__ all_tp,
                                                              const double* __restrict__ z_all, int K, long nz,
                                                              float* __restrict__ t_out, int B, double* __restrict__ tparams,
                                                              double* __restrict__ z_out) {
  const int idx = *counter;                         // (every thread reads it before thread 0 advances it: barrier below)
  const int tid = (int)threadIdx.x;
  if (tid < B) t_out[tid] = all_t[idx];
  if (tid < 2) tparams[tid] = all_tp[2 * idx + tid];
  const double* __restrict__ src = z_all + (long)(idx % K) * nz;
  for (long i = tid; i < nz; i += blockDim.x) z_out[i] = src[i];
  for (int b = tid + (int)blockDim.x; b < B; b += (int)blockDim.x) t_out[b] = all_t[idx];
  __syncthreads();
  if (tid == 0) *counter = idx + 1;
}

}  // namespace

extern "C" int fd_sample_advance(int* counter, const float* all_t, const double* all_tp, const double* z_all, int K, long nz,
                                 float* t_out, int B, double* tparams, double* z_out, void* stream) {
  FD_CHECK_ARG(counter && all_t && all_tp && z_all && t_out && tparams && z_out, "fd_sample_advance: null operand");
  FD_CHECK_ARG(K >= 1 && nz >= 0 && B >= 1, "fd_sample_advance: bad extents");
  hipLaunchKernelGGL(sample_advance_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, counter, all_t, all_tp, z_all, K, nz,
                     t_out, B, tparams, z_out);
  FD_CHECK_LAUNCH("fd_sample_advance");
  return FD_OK;
}

extern "C" int fd_igso3_tables(const double* sigma, const double* omega, int ns, int no, int L, double* pdf,
                               double* cdf, double* score_norms, void* stream) {
  if (ns == 0 || no == 0) return FD_OK;
  long g = ((long)ns * no + 255) / 256;
  hipLaunchKernelGGL(igso3_tables_kernel, dim3((unsigned)(g > 65535 ? 65535 : g)), dim3(256), 0, (hipStream_t)stream,
                     sigma, omega, ns, no, L, pdf, score_norms);
  FD_CHECK_LAUNCH("fd_igso3_tables");
  hipLaunchKernelGGL(igso3_cdf_kernel, dim3((unsigned)((ns + 63) / 64)), dim3(64), 0, (hipStream_t)stream, pdf, ns, no,
                     cdf);
  FD_CHECK_LAUNCH("fd_igso3_tables(cdf)");
  return FD_OK;
}

extern "C" int fd_sample_ref(const double* z_axis, const double* u, const double* z_trans, const double* cdf_row,
                             const double* omega, int no, double coord_scale, float* out, long n, void* stream) {
  if (n == 0) return FD_OK;
  long g = (n + 255) / 256;
  hipLaunchKernelGGL(sample_ref_kernel, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, (hipStream_t)stream,
                     z_axis, u, z_trans, cdf_row, omega, no, coord_scale, out, n);
  FD_CHECK_LAUNCH("fd_sample_ref");
  return FD_OK;
}

extern "C" int fd_forward_marginal(const float* rig0, const double* z_axis, const double* u, const double* z_trans,
                                   const double* cdf_row, const double* omega, int no, const double* score_row,
                                   double sigma, double beta, double coord_scale, int L, const float* mask,
                                   float* rig_t, double* rot_score, double* trans_score, long n, void* stream) {
  if (n == 0) return FD_OK;
  long g = (n + 255) / 256;
  hipLaunchKernelGGL(forward_marginal_kernel, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0,
                     (hipStream_t)stream, rig0, z_axis, u, z_trans, cdf_row, omega, no, score_row, sigma, beta,
                     coord_scale, L, mask, rig_t, rot_score, trans_score, n);
  FD_CHECK_LAUNCH("fd_forward_marginal");
  return FD_OK;
}

extern "C" int fd_forward_marginal_batch(const float* rig0, const double* z_axis, const double* u,
                                         const double* z_trans, const double* cdf, const double* omega, int no,
                                         const double* score_norms, const double* tparams, double coord_scale, int L,
                                         const float* mask, float* rig_t, double* rot_score, double* trans_score,
                                         int B, int N, void* stream) {
  if (B == 0 || N == 0) return FD_OK;
  FD_CHECK_ARG(tparams != nullptr, "fd_forward_marginal_batch: null per-example parameters");
  hipLaunchKernelGGL(forward_marginal_batch_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)B), dim3(256), 0,
                     (hipStream_t)stream, rig0, z_axis, u, z_trans, cdf, omega, no, score_norms, tparams, coord_scale,
                     L, mask, rig_t, rot_score, trans_score, N);
  FD_CHECK_LAUNCH("fd_forward_marginal_batch");
  return FD_OK;
}

extern "C" int fd_se3_reverse_step(const float* rig_t, const double* rot_score, const double* trans_score,
                                   const double* z_rot, const double* z_trans, const float* mask, int B, int N,
                                   double g_rot, double b_t, const double* tparams, double dt, double noise_scale,
                                   double coord_scale, int center, int diffuse_rot, int diffuse_trans, float* out,
                                   void* stream) {
  if (B == 0 || N == 0) return FD_OK;
  hipLaunchKernelGGL(HIP_KERNEL_NAME(reverse_step_kernel<double, double>), dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, rig_t, rot_score,
                     trans_score, z_rot, z_trans, mask, N, g_rot, b_t, tparams, dt, noise_scale, coord_scale, center,
                     diffuse_rot, diffuse_trans, out);
  FD_CHECK_LAUNCH("fd_se3_reverse_step");
  return FD_OK;
}

// the scores exactly as ScoreNetwork.forward returns them (rot_score float64, trans_score float32: score_network.py:199-214 after
// the heads of this package): no conversion launch in front of the step
extern "C" int fd_se3_reverse_step_net(const float* rig_t, const double* rot_score, const float* trans_score,
                                       const double* z_rot, const double* z_trans, const float* mask, int B, int N,
                                       double g_rot, double b_t, const double* tparams, double dt, double noise_scale,
                                       double coord_scale, int center, int diffuse_rot, int diffuse_trans, float* out,
                                       void* stream) {
  if (B == 0 || N == 0) return FD_OK;
  hipLaunchKernelGGL(HIP_KERNEL_NAME(reverse_step_kernel<double, float>), dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, rig_t,
                     rot_score, trans_score, z_rot, z_trans, mask, N, g_rot, b_t, tparams, dt, noise_scale, coord_scale, center,
                     diffuse_rot, diffuse_trans, out);
  FD_CHECK_LAUNCH("fd_se3_reverse_step_net");
  return FD_OK;
}

extern "C" int fd_se3_reverse_step_f32(const float* rig_t, const float* rot_score, const float* trans_score,
                                       const double* z_rot, const double* z_trans, const float* mask, int B, int N,
                                       double g_rot, double b_t, const double* tparams, double dt, double noise_scale,
                                       double coord_scale, int center, int diffuse_rot, int diffuse_trans, float* out,
                                       void* stream) {
  if (B == 0 || N == 0) return FD_OK;
  hipLaunchKernelGGL(HIP_KERNEL_NAME(reverse_step_kernel<float, float>), dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, rig_t, rot_score,
                     trans_score, z_rot, z_trans, mask, N, g_rot, b_t, tparams, dt, noise_scale, coord_scale, center,
                     diffuse_rot, diffuse_trans, out);
  FD_CHECK_LAUNCH("fd_se3_reverse_step_f32");
  return FD_OK;
}
