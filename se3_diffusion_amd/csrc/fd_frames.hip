// Per-residue frame algebra: sequence-transformer row softmax, backbone update, score heads
// (IGSO(3) series in fp64), psi head and idealised backbone atoms -- forward and backward.
//
// Reference:
//   softmax   torch.nn.MultiheadAttention inside nn.TransformerEncoderLayer (ipa_pytorch.py:584-593)
//   bb_update ipa_pytorch.py:530-557,641-644; rigid_utils.py:266-275 (quat_multiply_by_vec),
//             :587-616 (Rotation.compose_q_update_vec), :1039-1063 (Rigid.compose_q_update_vec)
//   scores    ipa_pytorch.py:650-662 -> se3_diffuser.py:115-125; data/utils.py:582-599
//             (quat_to_rotvec); so3_diffuser.py:9-49,71-117,182-213,274-305; r3_diffuser.py:42-43,148-166
//   psi       ipa_pytorch.py:491-507; score_network.py:167,201-203
//   atoms     all_atom.py:152-174, feats.py:165-228, residue_constants.py:127-133,819-824
#include "fd_common.h"
#include "../../include/fd_hip.h"

namespace {

struct Rot { float r[9]; };

__device__ __forceinline__ Rot quat_to_rot(const float* __restrict__ q) {
  const float a = q[0], b = q[1], c = q[2], d = q[3];
  Rot R;
  R.r[0] = a * a + b * b - c * c - d * d;
  R.r[1] = 2.f * (b * c - a * d);
  R.r[2] = 2.f * (b * d + a * c);
  R.r[3] = 2.f * (b * c + a * d);
  R.r[4] = a * a - b * b + c * c - d * d;
  R.r[5] = 2.f * (c * d - a * b);
  R.r[6] = 2.f * (b * d - a * c);
  R.r[7] = 2.f * (c * d + a * b);
  R.r[8] = a * a - b * b - c * c + d * d;
  return R;
}

// dL/dq from dL/dR (G row-major 3x3)
__device__ __forceinline__ void rot_grad_to_quat(const float* __restrict__ q, const float* G, float* dq) {
  const float a = q[0], b = q[1], c = q[2], d = q[3];
  dq[0] += 2.f * a * (G[0] + G[4] + G[8]) + 2.f * (-d * G[1] + c * G[2] + d * G[3] - b * G[5] - c * G[6] + b * G[7]);
  dq[1] += 2.f * b * (G[0] - G[4] - G[8]) + 2.f * (c * G[1] + d * G[2] + c * G[3] - a * G[5] + d * G[6] + a * G[7]);
  dq[2] += 2.f * c * (-G[0] + G[4] - G[8]) + 2.f * (b * G[1] + a * G[2] + b * G[3] + d * G[5] - a * G[6] + d * G[7]);
  dq[3] += 2.f * d * (-G[0] - G[4] + G[8]) + 2.f * (-a * G[1] + b * G[2] + a * G[3] + c * G[5] + b * G[6] + c * G[7]);
}

// ------------------------------------------------------------ row softmax (seq transformer)
// S [rows_b*nh*N, N] in place: softmax_j(S[row][j] + key_add[b][j]); one wave per row.
__global__ __launch_bounds__(256) void row_softmax_fwd_kernel(float* __restrict__ S, const float* __restrict__ key_add,
                                                              long rows, int N, int rows_per_batch) {
  const int lane = fd::lane_id(), wave = fd::wave_id();
  for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
    float* s = S + row * N;
    const float* ka = key_add ? key_add + (row / rows_per_batch) * N : nullptr;
    float mx = -INFINITY;
    for (int j = lane; j < N; j += 64) {
      float v = s[j] + (ka ? ka[j] : 0.f);
      s[j] = v;
      mx = fmaxf(mx, v);
    }
    mx = fd::wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < N; j += 64) {
      float e = expf(s[j] - mx);
      s[j] = e;
      sum += e;
    }
    sum = fd::wave_sum(sum);
    for (int j = lane; j < N; j += 64) s[j] = s[j] / sum;
  }
}

__global__ __launch_bounds__(256) void row_softmax_bwd_kernel(const float* __restrict__ A, float* __restrict__ dA,
                                                              long rows, int N) {
  const int lane = fd::lane_id(), wave = fd::wave_id();
  for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
    const float* a = A + row * N;
    float* d = dA + row * N;
    float dot = 0.f;
    for (int j = lane; j < N; j += 64) dot += a[j] * d[j];
    dot = fd::wave_sum(dot);
    for (int j = lane; j < N; j += 64) d[j] = a[j] * (d[j] - dot);
  }
}

// ------------------------------------------------------------ backbone update
// one wave per residue: upd = W6 (node * d) + b6 ; q' = normalize(q + d * (q (x) (0,u_q))) ; t' = t + d * R(q) u_t
__global__ __launch_bounds__(256) void bb_update_fwd_kernel(const float* __restrict__ node, long ldn, int cs,
                                                            const float* __restrict__ dmask,
                                                            const float* __restrict__ W6, const float* __restrict__ b6,
                                                            const float* __restrict__ quat,
                                                            const float* __restrict__ trans, float* __restrict__ upd,
                                                            float* __restrict__ quat_out, float* __restrict__ trans_out,
                                                            long R_) {
  const int lane = fd::lane_id(), wave = fd::wave_id();
  for (long r = (long)blockIdx.x * 4 + wave; r < R_; r += (long)gridDim.x * 4) {
    const float d = dmask[r];
    float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int c = lane; c < cs; c += 64) {
      const float x = node[r * ldn + c] * d;
#pragma unroll
      for (int k = 0; k < 6; ++k) acc[k] += W6[k * cs + c] * x;
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) acc[k] = fd::wave_sum(acc[k]) + b6[k];
    if (lane == 0) {
      const float* q = quat + r * 4;
      const float w = q[0], x = q[1], y = q[2], z = q[3];
      const float vx = acc[0], vy = acc[1], vz = acc[2];
      float n0 = w + d * (-x * vx - y * vy - z * vz);
      float n1 = x + d * (w * vx + y * vz - z * vy);
      float n2 = y + d * (w * vy - x * vz + z * vx);
      float n3 = z + d * (w * vz + x * vy - y * vx);
      const float nrm = sqrtf(n0 * n0 + n1 * n1 + n2 * n2 + n3 * n3);
      quat_out[r * 4 + 0] = n0 / nrm;
      quat_out[r * 4 + 1] = n1 / nrm;
      quat_out[r * 4 + 2] = n2 / nrm;
      quat_out[r * 4 + 3] = n3 / nrm;
      const Rot R = quat_to_rot(q);
      const float ux = acc[3], uy = acc[4], uz = acc[5];
      trans_out[r * 3 + 0] = trans[r * 3 + 0] + (R.r[0] * ux + R.r[1] * uy + R.r[2] * uz) * d;
      trans_out[r * 3 + 1] = trans[r * 3 + 1] + (R.r[3] * ux + R.r[4] * uy + R.r[5] * uz) * d;
      trans_out[r * 3 + 2] = trans[r * 3 + 2] + (R.r[6] * ux + R.r[7] * uy + R.r[8] * uz) * d;
#pragma unroll
      for (int k = 0; k < 6; ++k) upd[r * 6 + k] = acc[k];
    }
  }
}

// thread per residue.  in: dq' dt' (grad wrt outputs), optional dframe [R,12] (dL/dR, dL/dt of the
// INPUT frame from the IPA kernels).  out: dq dt (grad wrt input frame), dupd [R,6], dupd_scaled = dupd*d.
__global__ __launch_bounds__(256) void bb_update_bwd_kernel(const float* __restrict__ dquat_out,
                                                            const float* __restrict__ dtrans_out,
                                                            const float* __restrict__ dframe,
                                                            const float* __restrict__ dmask,
                                                            const float* __restrict__ upd,
                                                            const float* __restrict__ quat, float* __restrict__ dquat,
                                                            float* __restrict__ dtrans, float* __restrict__ dupd,
                                                            float* __restrict__ dupd_s, long R_) {
  for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < R_; r += (long)gridDim.x * 256) {
    const float d = dmask[r];
    const float* q = quat + r * 4;
    const float w = q[0], x = q[1], y = q[2], z = q[3];
    const float* u = upd + r * 6;
    const float vx = u[0], vy = u[1], vz = u[2];
    const float n0 = w + d * (-x * vx - y * vy - z * vz);
    const float n1 = x + d * (w * vx + y * vz - z * vy);
    const float n2 = y + d * (w * vy - x * vz + z * vx);
    const float n3 = z + d * (w * vz + x * vy - y * vx);
    const float nrm = sqrtf(n0 * n0 + n1 * n1 + n2 * n2 + n3 * n3);
    const float qo0 = n0 / nrm, qo1 = n1 / nrm, qo2 = n2 / nrm, qo3 = n3 / nrm;
    const float* go = dquat_out + r * 4;
    const float dotp = qo0 * go[0] + qo1 * go[1] + qo2 * go[2] + qo3 * go[3];
    const float dn0 = (go[0] - qo0 * dotp) / nrm, dn1 = (go[1] - qo1 * dotp) / nrm;
    const float dn2 = (go[2] - qo2 * dotp) / nrm, dn3 = (go[3] - qo3 * dotp) / nrm;
    const float gw = d * dn0, gx = d * dn1, gy = d * dn2, gz = d * dn3;
    float dq[4];
    dq[0] = dn0 + (vx * gx + vy * gy + vz * gz);
    dq[1] = dn1 + (-vx * gw - vz * gy + vy * gz);
    dq[2] = dn2 + (-vy * gw + vz * gx - vx * gz);
    dq[3] = dn3 + (-vz * gw - vy * gx + vx * gy);
    float du[6];
    du[0] = -x * gw + w * gx + z * gy - y * gz;
    du[1] = -y * gw - z * gx + w * gy + x * gz;
    du[2] = -z * gw + y * gx - x * gy + w * gz;
    const Rot R = quat_to_rot(q);
    const float* gt = dtrans_out + r * 3;
    const float t0 = gt[0] * d, t1 = gt[1] * d, t2 = gt[2] * d;
    du[3] = R.r[0] * t0 + R.r[3] * t1 + R.r[6] * t2;
    du[4] = R.r[1] * t0 + R.r[4] * t1 + R.r[7] * t2;
    du[5] = R.r[2] * t0 + R.r[5] * t1 + R.r[8] * t2;
    float G[9];
    G[0] = t0 * u[3]; G[1] = t0 * u[4]; G[2] = t0 * u[5];
    G[3] = t1 * u[3]; G[4] = t1 * u[4]; G[5] = t1 * u[5];
    G[6] = t2 * u[3]; G[7] = t2 * u[4]; G[8] = t2 * u[5];
    float dt0 = gt[0], dt1 = gt[1], dt2 = gt[2];
    if (dframe) {
      const float* f = dframe + r * 12;
#pragma unroll
      for (int k = 0; k < 9; ++k) G[k] += f[k];
      dt0 += f[9]; dt1 += f[10]; dt2 += f[11];
    }
    rot_grad_to_quat(q, G, dq);
#pragma unroll
    for (int k = 0; k < 4; ++k) dquat[r * 4 + k] = dq[k];
    dtrans[r * 3 + 0] = dt0; dtrans[r * 3 + 1] = dt1; dtrans[r * 3 + 2] = dt2;
#pragma unroll
    for (int k = 0; k < 6; ++k) { dupd[r * 6 + k] = du[k]; dupd_s[r * 6 + k] = du[k] * d; }
  }
}

// ------------------------------------------------------------ IGSO(3) series (fp64)
// f   = sum_l (2l+1) w_l sin(a w)/sin(w/2),  a = l + 1/2,  w_l = exp(-l(l+1) s^2/2)
// f'  = sum_l (2l+1) w_l N/lo^2,  N = lo*dhi - hi*dlo
// f'' = sum_l (2l+1) w_l [ hi (1/4 - a^2)/lo - 2 N dlo / lo^3 ]
// sin/cos((l+1/2)w) by rotation recurrence, weights by ratio recurrence (both fp64).
__device__ __forceinline__ void igso3_series(double om, double sg, int L, double* f, double* df, double* d2f) {
  const double lo = sin(0.5 * om), dlo = 0.5 * cos(0.5 * om);
  const double so = sin(om), co = cos(om);
  double s = lo, c = 2.0 * dlo;      // sin, cos of (l + 1/2) w at l = 0
  const double q = exp(-sg * sg);
  double w = 1.0, r = q;
  double F = 0.0, D = 0.0, D2 = 0.0;
  const double ilo = 1.0 / lo, ilo2 = ilo * ilo, ilo3 = ilo2 * ilo;
  for (int l = 0; l < L; ++l) {
    const double a = (double)l + 0.5;
    const double cw = (double)(2 * l + 1) * w;
    const double hi = s, dhi = a * c;
    const double Nn = lo * dhi - hi * dlo;
    F += cw * hi * ilo;
    D += cw * Nn * ilo2;
    D2 += cw * (hi * (0.25 - a * a) * ilo - 2.0 * Nn * dlo * ilo3);
    w *= r;
    r *= q;
    if (w == 0.0) break;
    const double s2 = s * co + c * so;
    c = c * co - s * so;
    s = s2;
  }
  *f = F; *df = D; *d2f = D2;
}

// sigma(t) = log(t e^{max} + (1-t) e^{min}); bin = digitize(sigma) - 1 (so3_diffuser.py:188-215 t_to_idx)
__device__ __forceinline__ int sigma_index(double t, const double* __restrict__ grid, int ng, double emax,
                                           double emin) {
  double s = log(t * emax + (1.0 - t) * emin);
  s *= (1.0 + 4.5e-16);  // x == grid[k] must land in bin k whatever the last ulp of log() does
  int lo = 0, hi = ng;   // count of grid entries <= s
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (grid[mid] <= s) lo = mid + 1; else hi = mid;
  }
  int idx = lo - 1;
  if (idx < 0) idx = ng - 1;  // numpy negative indexing (t < 0 is rejected upstream)
  return idx;
}

// d/dw log IGSO3 at (w, sigma-bin si): g = f'/(f + 1e-4) and its derivative gp.  With a cached table
// (use_cached_score, so3_diffuser.py:293-299) g = score_norms[si, bucketize(w, omega[:-1])] and the lookup carries
// no gradient (torch.gather of a constant), so gp = 0.
__device__ __forceinline__ int omega_bucket(double om, const double* __restrict__ grid, int no) {
  int lo = 0, hi = no - 1;  // count of grid[0 .. no-2] < om  (torch.bucketize, right=False)
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (grid[mid] < om) lo = mid + 1; else hi = mid;
  }
  return lo;
}

struct HeadConst {
  float atoms[15];  // N, CA, C, CB, O local coordinates (O in the psi frame)
  float Rd[9];      // psi-group default frame rotation
  float td[3];      // psi-group default frame translation
  float coord_scale;
  double exp_max_sigma, exp_min_sigma;
  float min_b, max_b;
  int L;
  const double* score_norms;  // [ng, n_omega] or null
  const double* omega_grid;   // [n_omega]
  int n_omega;
};

__device__ __forceinline__ void igso3_score(float omega, float tb, const double* __restrict__ sigma_grid, int ng,
                                            const HeadConst& hc, double* g, double* gp) {
  const int si = sigma_index((double)tb, sigma_grid, ng, hc.exp_max_sigma, hc.exp_min_sigma);
  if (hc.score_norms) {
    *g = hc.score_norms[(long)si * hc.n_omega + omega_bucket((double)omega, hc.omega_grid, hc.n_omega)];
    *gp = 0.0;
    return;
  }
  double f, df, d2f;
  igso3_series((double)omega, sigma_grid[si], hc.L, &f, &df, &d2f);
  const double fe = f + 1e-4;
  *g = df / fe;
  *gp = d2f / fe - df * df / (fe * fe);
}

// thread per residue
__global__ __launch_bounds__(128) void heads_fwd_kernel(
    const float* __restrict__ rig0 /*[R,7] init (noised) frames, A*/, const float* __restrict__ quatF,
    const float* __restrict__ transF /*nm*/, const float* __restrict__ upsi /*[R,2]*/,
    const float* __restrict__ gt_psi, long gt_stride, const float* __restrict__ fixed,
    const float* __restrict__ mask, const float* __restrict__ t, const double* __restrict__ sigma_grid, int ng,
    HeadConst hc, double* __restrict__ rot_score, float* __restrict__ trans_score, float* __restrict__ rigids,
    float* __restrict__ psi_out, float* __restrict__ atom37, float* __restrict__ atom14, float* __restrict__ sc_ca_out,
    int N, long R_) {
  for (long r = (long)blockIdx.x * 128 + threadIdx.x; r < R_; r += (long)gridDim.x * 128) {
    const int b = (int)(r / N);
    const float m = mask[r];
    const float tb = t[b];
    const float* q0 = rig0 + r * 7;
    const float* qf = quatF + r * 4;
    // ---- rotation score: q0t = inv(qF) (x) q0
    const float nn = qf[0] * qf[0] + qf[1] * qf[1] + qf[2] * qf[2] + qf[3] * qf[3];
    const float cw = qf[0] / nn, cx = -qf[1] / nn, cy = -qf[2] / nn, cz = -qf[3] / nn;
    float pw = cw * q0[0] - cx * q0[1] - cy * q0[2] - cz * q0[3];
    float px = cw * q0[1] + cx * q0[0] + cy * q0[3] - cz * q0[2];
    float py = cw * q0[2] - cx * q0[3] + cy * q0[0] + cz * q0[1];
    float pz = cw * q0[3] + cx * q0[2] - cy * q0[1] + cz * q0[0];
    if (pw < 0.f) { pw = -pw; px = -px; py = -py; pz = -pz; }
    const float mm = sqrtf(px * px + py * py + pz * pz);
    const float ang = 2.f * atan2f(mm, pw);
    const float a2 = ang * ang;
    const float sc = ang <= 1e-3f ? 2.f + a2 / 12.f + 7.f * a2 * a2 / 2880.f : ang / sinf(ang / 2.f + 1e-6f);
    const float vx = sc * px, vy = sc * py, vz = sc * pz;
    const float omega = sqrtf(vx * vx + vy * vy + vz * vz) + 1e-6f;
    double g, gp;
    igso3_score(omega, tb, sigma_grid, ng, hc, &g, &gp);
    const double scal = g / (double)(omega + 1e-6f) * (double)m;
    rot_score[r * 3 + 0] = scal * (double)vx;
    rot_score[r * 3 + 1] = scal * (double)vy;
    rot_score[r * 3 + 2] = scal * (double)vz;
    // ---- translation score (fp32, reference arithmetic order)
    const float cs = hc.coord_scale;
    const float beta = tb * hc.min_b + 0.5f * (tb * tb) * (hc.max_b - hc.min_b);
    const float e1 = expf(-0.5f * beta), cv = 1.f - expf(-beta);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float ta = transF[r * 3 + k] / cs;       // unscale_rigids (nm -> A)
      const float xt = q0[4 + k] * cs, x0 = ta * cs;
      trans_score[r * 3 + k] = (-(xt - e1 * x0) / cv) * m;
      rigids[r * 7 + 4 + k] = ta;
      // sampling: the predicted CA position is the next forward's self-conditioning input (train_se3_diffusion.py:763-765)
      if (sc_ca_out != nullptr) sc_ca_out[r * 3 + k] = ta;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) rigids[r * 7 + k] = qf[k];
    // ---- psi head
    const float u0 = upsi[r * 2], u1 = upsi[r * 2 + 1];
    const float den = sqrtf(fmaxf(u0 * u0 + u1 * u1, 1e-8f));
    const float fm = fixed[r];
    const float ps = (1.f - fm) * (u0 / den) + fm * gt_psi[r * gt_stride];
    const float pc = (1.f - fm) * (u1 / den) + fm * gt_psi[r * gt_stride + 1];
    psi_out[r * 2] = ps;
    psi_out[r * 2 + 1] = pc;
    // ---- backbone atoms (A)
    const Rot R = quat_to_rot(qf);
    const float tx = rigids[r * 7 + 4], ty = rigids[r * 7 + 5], tz = rigids[r * 7 + 6];
    float* a37 = atom37 + r * 111;
    float* a14 = atom14 + r * 42;
    for (int k = 0; k < 111; ++k) a37[k] = 0.f;
    for (int k = 0; k < 42; ++k) a14[k] = 0.f;
    float pos[5][3];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const float lx = hc.atoms[3 * a], ly = hc.atoms[3 * a + 1], lz = hc.atoms[3 * a + 2];
      pos[a][0] = R.r[0] * lx + R.r[1] * ly + R.r[2] * lz + tx;
      pos[a][1] = R.r[3] * lx + R.r[4] * ly + R.r[5] * lz + ty;
      pos[a][2] = R.r[6] * lx + R.r[7] * ly + R.r[8] * lz + tz;
    }
    {
      // O: frame = bb o (default_psi_frame o Rpsi); Rpsi = [[1,0,0],[0,c,-s],[0,s,c]]
      const float ox = hc.atoms[12], oy = hc.atoms[13], oz = hc.atoms[14];
      const float rx = ox, ry = pc * oy - ps * oz, rz = ps * oy + pc * oz;       // Rpsi * pO
      const float wx = hc.Rd[0] * rx + hc.Rd[1] * ry + hc.Rd[2] * rz + hc.td[0];  // bb-local
      const float wy = hc.Rd[3] * rx + hc.Rd[4] * ry + hc.Rd[5] * rz + hc.td[1];
      const float wz = hc.Rd[6] * rx + hc.Rd[7] * ry + hc.Rd[8] * rz + hc.td[2];
      pos[4][0] = R.r[0] * wx + R.r[1] * wy + R.r[2] * wz + tx;
      pos[4][1] = R.r[3] * wx + R.r[4] * wy + R.r[5] * wz + ty;
      pos[4][2] = R.r[6] * wx + R.r[7] * wy + R.r[8] * wz + tz;
    }
    // atom37 order N, CA, C, CB, O ; atom14 order N, CA, C, O, CB
    const int m14[5] = {0, 1, 2, 4, 3};
#pragma unroll
    for (int a = 0; a < 5; ++a)
#pragma unroll
      for (int k = 0; k < 3; ++k) { a37[a * 3 + k] = pos[a][k]; a14[m14[a] * 3 + k] = pos[a][k]; }
  }
}

// thread per residue: grads wrt quatF, transF(nm), upsi.  Any of the incoming grads may be null.
__global__ __launch_bounds__(128) void heads_bwd_kernel(
    const float* __restrict__ rig0, const float* __restrict__ quatF, const float* __restrict__ transF,
    const float* __restrict__ upsi, const float* __restrict__ psi_out, const float* __restrict__ fixed,
    const float* __restrict__ mask, const float* __restrict__ t, const double* __restrict__ sigma_grid, int ng,
    HeadConst hc, const double* __restrict__ d_rot, const float* __restrict__ d_trans_score,
    const float* __restrict__ d_rigids, const float* __restrict__ d_psi, const float* __restrict__ d_atom37,
    float* __restrict__ dquatF, float* __restrict__ dtransF, float* __restrict__ dupsi, int N, long R_) {
  for (long r = (long)blockIdx.x * 128 + threadIdx.x; r < R_; r += (long)gridDim.x * 128) {
    const int b = (int)(r / N);
    const float m = mask[r];
    const float tb = t[b];
    const float* q0 = rig0 + r * 7;
    const float* qf = quatF + r * 4;
    float dq[4] = {0.f, 0.f, 0.f, 0.f};
    float dtA[3] = {0.f, 0.f, 0.f};  // grad wrt translation in Angstrom (= transF / cs)
    const float cs = hc.coord_scale;
    if (d_rigids) {
#pragma unroll
      for (int k = 0; k < 4; ++k) dq[k] += d_rigids[r * 7 + k];
#pragma unroll
      for (int k = 0; k < 3; ++k) dtA[k] += d_rigids[r * 7 + 4 + k];
    }
    if (d_trans_score) {
      const float beta = tb * hc.min_b + 0.5f * (tb * tb) * (hc.max_b - hc.min_b);
      const float e1 = expf(-0.5f * beta), cv = 1.f - expf(-beta);
#pragma unroll
      for (int k = 0; k < 3; ++k) dtA[k] += d_trans_score[r * 3 + k] * m * (e1 * cs / cv);
    }
    if (d_rot) {
      // recompute forward
      const float nn = qf[0] * qf[0] + qf[1] * qf[1] + qf[2] * qf[2] + qf[3] * qf[3];
      const float c0 = qf[0] / nn, c1 = -qf[1] / nn, c2 = -qf[2] / nn, c3 = -qf[3] / nn;
      float pw = c0 * q0[0] - c1 * q0[1] - c2 * q0[2] - c3 * q0[3];
      float px = c0 * q0[1] + c1 * q0[0] + c2 * q0[3] - c3 * q0[2];
      float py = c0 * q0[2] - c1 * q0[3] + c2 * q0[0] + c3 * q0[1];
      float pz = c0 * q0[3] + c1 * q0[2] - c2 * q0[1] + c3 * q0[0];
      const float sgn = pw < 0.f ? -1.f : 1.f;
      pw *= sgn; px *= sgn; py *= sgn; pz *= sgn;
      const float mm = sqrtf(px * px + py * py + pz * pz);
      const float ang = 2.f * atan2f(mm, pw);
      const float a2 = ang * ang;
      float sc, dsc;
      if (ang <= 1e-3f) {
        sc = 2.f + a2 / 12.f + 7.f * a2 * a2 / 2880.f;
        dsc = ang / 6.f + 7.f * a2 * ang / 720.f;
      } else {
        const float sh = sinf(ang / 2.f + 1e-6f), ch = cosf(ang / 2.f + 1e-6f);
        sc = ang / sh;
        dsc = 1.f / sh - ang * ch * 0.5f / (sh * sh);
      }
      const float vx = sc * px, vy = sc * py, vz = sc * pz;
      const float vn = sqrtf(vx * vx + vy * vy + vz * vz);
      const float omega = vn + 1e-6f;
      double g, gp;
      igso3_score(omega, tb, sigma_grid, ng, hc, &g, &gp);
      const double den = (double)(omega + 1e-6f);
      const double gx = d_rot[r * 3] * (double)m, gy = d_rot[r * 3 + 1] * (double)m, gz = d_rot[r * 3 + 2] * (double)m;
      // s = g(w) v / den ; w = |v| + eps ; den = w + eps
      const double vdot = gx * vx + gy * vy + gz * vz;
      const double radial = vn > 0.f ? vdot * (gp / den - g / (den * den)) / (double)vn : 0.0;
      const float dvx = (float)(g / den * gx + radial * vx);
      const float dvy = (float)(g / den * gy + radial * vy);
      const float dvz = (float)(g / den * gz + radial * vz);
      // v = sc(ang) * u ; ang = 2 atan2(|u|, w)
      const float dvu = dvx * px + dvy * py + dvz * pz;
      const float r2 = mm * mm + pw * pw;
      const float dang_dm = 2.f * pw / r2, dang_dw = -2.f * mm / r2;
      const float km = mm > 0.f ? dvu * dsc * dang_dm / mm : 0.f;
      float dpx = sc * dvx + km * px, dpy = sc * dvy + km * py, dpz = sc * dvz + km * pz;
      float dpw = dvu * dsc * dang_dw;
      dpw *= sgn; dpx *= sgn; dpy *= sgn; dpz *= sgn;
      // p = c (x) q0  =>  dc = dp (x) conj(q0)
      const float e0 = q0[0], e1 = -q0[1], e2 = -q0[2], e3 = -q0[3];
      const float dc0 = dpw * e0 - dpx * e1 - dpy * e2 - dpz * e3;
      const float dc1 = dpw * e1 + dpx * e0 + dpy * e3 - dpz * e2;
      const float dc2 = dpw * e2 - dpx * e3 + dpy * e0 + dpz * e1;
      const float dc3 = dpw * e3 + dpx * e2 - dpy * e1 + dpz * e0;
      // c = conj(qF) / nn
      const float dnn = -(dc0 * qf[0] - dc1 * qf[1] - dc2 * qf[2] - dc3 * qf[3]) / (nn * nn);
      dq[0] += dc0 / nn + 2.f * qf[0] * dnn;
      dq[1] += -dc1 / nn + 2.f * qf[1] * dnn;
      dq[2] += -dc2 / nn + 2.f * qf[2] * dnn;
      dq[3] += -dc3 / nn + 2.f * qf[3] * dnn;
    }
    float dps = 0.f, dpc = 0.f;
    if (d_psi) { dps += d_psi[r * 2]; dpc += d_psi[r * 2 + 1]; }
    if (d_atom37) {
      const Rot R = quat_to_rot(qf);
      const float* g37 = d_atom37 + r * 111;
      float G[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const float lx = hc.atoms[3 * a], ly = hc.atoms[3 * a + 1], lz = hc.atoms[3 * a + 2];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float gk = g37[a * 3 + k];
          G[3 * k] += gk * lx; G[3 * k + 1] += gk * ly; G[3 * k + 2] += gk * lz;
          dtA[k] += gk;
        }
      }
      {
        const float ps = psi_out[r * 2], pc = psi_out[r * 2 + 1];
        const float ox = hc.atoms[12], oy = hc.atoms[13], oz = hc.atoms[14];
        const float rx = ox, ry = pc * oy - ps * oz, rz = ps * oy + pc * oz;
        const float wx = hc.Rd[0] * rx + hc.Rd[1] * ry + hc.Rd[2] * rz + hc.td[0];
        const float wy = hc.Rd[3] * rx + hc.Rd[4] * ry + hc.Rd[5] * rz + hc.td[1];
        const float wz = hc.Rd[6] * rx + hc.Rd[7] * ry + hc.Rd[8] * rz + hc.td[2];
        const float g0 = g37[12], g1 = g37[13], g2 = g37[14];
        G[0] += g0 * wx; G[1] += g0 * wy; G[2] += g0 * wz;
        G[3] += g1 * wx; G[4] += g1 * wy; G[5] += g1 * wz;
        G[6] += g2 * wx; G[7] += g2 * wy; G[8] += g2 * wz;
        dtA[0] += g0; dtA[1] += g1; dtA[2] += g2;
        // dw = R^T g ; d(Rpsi pO) = Rd^T dw
        const float dwx = R.r[0] * g0 + R.r[3] * g1 + R.r[6] * g2;
        const float dwy = R.r[1] * g0 + R.r[4] * g1 + R.r[7] * g2;
        const float dwz = R.r[2] * g0 + R.r[5] * g1 + R.r[8] * g2;
        const float dry = hc.Rd[1] * dwx + hc.Rd[4] * dwy + hc.Rd[7] * dwz;
        const float drz = hc.Rd[2] * dwx + hc.Rd[5] * dwy + hc.Rd[8] * dwz;
        // ry = pc*oy - ps*oz ; rz = ps*oy + pc*oz
        dps += -dry * oz + drz * oy;
        dpc += dry * oy + drz * oz;
      }
      rot_grad_to_quat(qf, G, dq);
    }
    // psi_out = (1-fm) * u/den + fm * gt
    {
      const float fm = fixed[r];
      const float u0 = upsi[r * 2], u1 = upsi[r * 2 + 1];
      const float n2 = u0 * u0 + u1 * u1;
      const float g0 = dps * (1.f - fm), g1 = dpc * (1.f - fm);
      if (n2 > 1e-8f) {
        const float den = sqrtf(n2);
        const float p0 = u0 / den, p1 = u1 / den;
        const float dt_ = p0 * g0 + p1 * g1;
        dupsi[r * 2] = (g0 - p0 * dt_) / den;
        dupsi[r * 2 + 1] = (g1 - p1 * dt_) / den;
      } else {
        dupsi[r * 2] = g0 * 1e4f;
        dupsi[r * 2 + 1] = g1 * 1e4f;
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) dquatF[r * 4 + k] = dq[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) dtransF[r * 3 + k] = dtA[k] / cs;
  }
}

// ---------------------------------------------------------------- input frames
// rigids_t [R,7] (quaternion | translation in Angstrom) -> quat [R,4], trans [R,3] * coordinate_scaling (score_network.py:
// 190-193 scale_rigids) and the embedder's timestep argument t * 10000 per example (score_network.py:38,43): one launch
// instead of three element-wise torch kernels at the head of every forward pass
__global__ __launch_bounds__(256) void split_rigids_kernel(const float* __restrict__ rig, float scale,
                                                           const float* __restrict__ t, float tscale,
                                                           float* __restrict__ quat, float* __restrict__ trans,
                                                           float* __restrict__ tscaled, long R_, int B) {
  for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < R_; r += (long)gridDim.x * 256) {
    const float* s = rig + r * 7;
    quat[r * 4 + 0] = s[0]; quat[r * 4 + 1] = s[1]; quat[r * 4 + 2] = s[2]; quat[r * 4 + 3] = s[3];
    trans[r * 3 + 0] = s[4] * scale; trans[r * 3 + 1] = s[5] * scale; trans[r * 3 + 2] = s[6] * scale;
    if (r < B && tscaled != nullptr) tscaled[r] = t[r] * tscale;
  }
}

}  // namespace

static unsigned rows4_grid(long rows, long cap) {
  long g = (rows + 3) / 4;
  return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

extern "C" int fd_split_rigids(const float* rig7, float scale, const float* t, float tscale, float* quat, float* trans,
                               float* tscaled, long R_, int B, void* stream) {
  FD_CHECK_ARG(rig7 && quat && trans, "fd_split_rigids: null operand");
  FD_CHECK_ARG(tscaled == nullptr || (t != nullptr && B <= R_), "fd_split_rigids: t missing or B > rows");
  if (R_ == 0) return FD_OK;
  long g = (R_ + 255) / 256;
  if (g > 1024) g = 1024;
  hipLaunchKernelGGL(split_rigids_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, rig7, scale, t, tscale, quat, trans,
                     tscaled, R_, B);
  FD_CHECK_LAUNCH("fd_split_rigids");
  return FD_OK;
}

extern "C" int fd_row_softmax_fwd(float* S, const float* key_add, long rows, int N, int rows_per_batch,
                                  void* stream) {
  if (rows == 0 || N == 0) return FD_OK;
  FD_CHECK_ARG(rows_per_batch > 0, "fd_row_softmax_fwd: rows_per_batch must be positive");
  hipLaunchKernelGGL(row_softmax_fwd_kernel, dim3(rows4_grid(rows, 16384)), dim3(256), 0, (hipStream_t)stream, S,
                     key_add, rows, N, rows_per_batch);
  FD_CHECK_LAUNCH("fd_row_softmax_fwd");
  return FD_OK;
}

extern "C" int fd_row_softmax_bwd(const float* A, float* dA, long rows, int N, void* stream) {
  if (rows == 0 || N == 0) return FD_OK;
  hipLaunchKernelGGL(row_softmax_bwd_kernel, dim3(rows4_grid(rows, 16384)), dim3(256), 0, (hipStream_t)stream, A, dA,
                     rows, N);
  FD_CHECK_LAUNCH("fd_row_softmax_bwd");
  return FD_OK;
}

extern "C" int fd_bb_update_fwd(const float* node, long ldn, int cs, const float* dmask, const float* W6,
                                const float* b6, const float* quat, const float* trans, float* upd, float* quat_out,
                                float* trans_out, long R_, void* stream) {
  if (R_ == 0) return FD_OK;
  hipLaunchKernelGGL(bb_update_fwd_kernel, dim3(rows4_grid(R_, 8192)), dim3(256), 0, (hipStream_t)stream, node, ldn,
                     cs, dmask, W6, b6, quat, trans, upd, quat_out, trans_out, R_);
  FD_CHECK_LAUNCH("fd_bb_update_fwd");
  return FD_OK;
}

extern "C" int fd_bb_update_bwd(const float* dquat_out, const float* dtrans_out, const float* dframe,
                                const float* dmask, const float* upd, const float* quat, float* dquat, float* dtrans,
                                float* dupd, float* dupd_s, long R_, void* stream) {
  if (R_ == 0) return FD_OK;
  long g = (R_ + 255) / 256;
  hipLaunchKernelGGL(bb_update_bwd_kernel, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, (hipStream_t)stream,
                     dquat_out, dtrans_out, dframe, dmask, upd, quat, dquat, dtrans, dupd, dupd_s, R_);
  FD_CHECK_LAUNCH("fd_bb_update_bwd");
  return FD_OK;
}

static HeadConst make_hc(const FdHeadConst* c) {
  HeadConst hc;
  for (int i = 0; i < 15; ++i) hc.atoms[i] = c->atoms[i];
  for (int i = 0; i < 9; ++i) hc.Rd[i] = c->Rd[i];
  for (int i = 0; i < 3; ++i) hc.td[i] = c->td[i];
  hc.coord_scale = c->coord_scale;
  hc.exp_max_sigma = c->exp_max_sigma;
  hc.exp_min_sigma = c->exp_min_sigma;
  hc.min_b = c->min_b;
  hc.max_b = c->max_b;
  hc.L = c->L;
  hc.score_norms = c->score_norms;
  hc.omega_grid = c->omega_grid;
  hc.n_omega = c->n_omega;
  return hc;
}

extern "C" int fd_heads_fwd(const float* rig0, const float* quatF, const float* transF, const float* upsi,
                            const float* gt_psi, long gt_stride, const float* fixed, const float* mask,
                            const float* t, const double* sigma_grid, int ng, const FdHeadConst* c,
                            double* rot_score, float* trans_score, float* rigids, float* psi_out, float* atom37,
                            float* atom14, float* sc_ca_out, int B, int N, void* stream) {
  FD_CHECK_ARG(c != nullptr, "fd_heads_fwd: null constants");
  FD_CHECK_ARG(!c->score_norms || (c->omega_grid && c->n_omega > 1), "fd_heads_fwd: cached score table without its grid");
  const long R_ = (long)B * N;
  if (R_ == 0) return FD_OK;
  long g = (R_ + 127) / 128;
  hipLaunchKernelGGL(heads_fwd_kernel, dim3((unsigned)(g > 8192 ? 8192 : g)), dim3(128), 0, (hipStream_t)stream, rig0,
                     quatF, transF, upsi, gt_psi, gt_stride, fixed, mask, t, sigma_grid, ng, make_hc(c), rot_score,
                     trans_score, rigids, psi_out, atom37, atom14, sc_ca_out, N, R_);
  FD_CHECK_LAUNCH("fd_heads_fwd");
  return FD_OK;
}

extern "C" int fd_heads_bwd(const float* rig0, const float* quatF, const float* transF, const float* upsi,
                            const float* psi_out, const float* fixed, const float* mask, const float* t,
                            const double* sigma_grid, int ng, const FdHeadConst* c, const double* d_rot,
                            const float* d_trans_score, const float* d_rigids, const float* d_psi,
                            const float* d_atom37, float* dquatF, float* dtransF, float* dupsi, int B, int N,
                            void* stream) {
  FD_CHECK_ARG(c != nullptr, "fd_heads_bwd: null constants");
  FD_CHECK_ARG(!c->score_norms || (c->omega_grid && c->n_omega > 1), "fd_heads_bwd: cached score table without its grid");
  const long R_ = (long)B * N;
  if (R_ == 0) return FD_OK;
  long g = (R_ + 127) / 128;
  hipLaunchKernelGGL(heads_bwd_kernel, dim3((unsigned)(g > 8192 ? 8192 : g)), dim3(128), 0, (hipStream_t)stream, rig0,
                     quatF, transF, upsi, psi_out, fixed, mask, t, sigma_grid, ng, make_hc(c), d_rot, d_trans_score,
                     d_rigids, d_psi, d_atom37, dquatF, dtransF, dupsi, N, R_);
  FD_CHECK_LAUNCH("fd_heads_bwd");
  return FD_OK;
}
