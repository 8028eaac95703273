// Stand-alone idealised backbone builder: frames (A) + psi -> N, CA, C, CB, O.
//
// Drop-in kernel behind data/all_atom.py:compute_backbone (reference all_atom.py:152-174,
// which runs torsion_angles_to_frames for all 8 rigid groups x 14 atoms with CPU-resident
// index tensors -> a device<->host round trip per call).  Here: one thread per residue,
// 5 atoms, no host involvement.  Same arithmetic as the atoms block of fd_heads_fwd.
#include "fd_common.h"
#include "../../include/fd_hip.h"

namespace {

struct AtomConst {
  float atoms[15];
  float Rd[9];
  float td[3];
};

__global__ __launch_bounds__(256) void backbone_atoms_kernel(const float* __restrict__ rigids,
                                                             const float* __restrict__ psi, AtomConst ac,
                                                             float* __restrict__ atom37, float* __restrict__ atom14,
                                                             long R_) {
  for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < R_; r += (long)gridDim.x * 256) {
    const float* q = rigids + r * 7;
    const float a = q[0], b = q[1], c = q[2], d = q[3];
    float Rm[9];
    Rm[0] = a * a + b * b - c * c - d * d; Rm[1] = 2.f * (b * c - a * d); Rm[2] = 2.f * (b * d + a * c);
    Rm[3] = 2.f * (b * c + a * d); Rm[4] = a * a - b * b + c * c - d * d; Rm[5] = 2.f * (c * d - a * b);
    Rm[6] = 2.f * (b * d - a * c); Rm[7] = 2.f * (c * d + a * b); Rm[8] = a * a - b * b - c * c + d * d;
    const float tx = q[4], ty = q[5], tz = q[6];
    float* a37 = atom37 + r * 111;
    float* a14 = atom14 + r * 42;
    for (int k = 0; k < 111; ++k) a37[k] = 0.f;
    for (int k = 0; k < 42; ++k) a14[k] = 0.f;
    float pos[5][3];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const float lx = ac.atoms[3 * n], ly = ac.atoms[3 * n + 1], lz = ac.atoms[3 * n + 2];
      pos[n][0] = Rm[0] * lx + Rm[1] * ly + Rm[2] * lz + tx;
      pos[n][1] = Rm[3] * lx + Rm[4] * ly + Rm[5] * lz + ty;
      pos[n][2] = Rm[6] * lx + Rm[7] * ly + Rm[8] * lz + tz;
    }
    const float ps = psi[r * 2], pc = psi[r * 2 + 1];
    const float ox = ac.atoms[12], oy = ac.atoms[13], oz = ac.atoms[14];
    const float rx = ox, ry = pc * oy - ps * oz, rz = ps * oy + pc * oz;
    const float wx = ac.Rd[0] * rx + ac.Rd[1] * ry + ac.Rd[2] * rz + ac.td[0];
    const float wy = ac.Rd[3] * rx + ac.Rd[4] * ry + ac.Rd[5] * rz + ac.td[1];
    const float wz = ac.Rd[6] * rx + ac.Rd[7] * ry + ac.Rd[8] * rz + ac.td[2];
    pos[4][0] = Rm[0] * wx + Rm[1] * wy + Rm[2] * wz + tx;
    pos[4][1] = Rm[3] * wx + Rm[4] * wy + Rm[5] * wz + ty;
    pos[4][2] = Rm[6] * wx + Rm[7] * wy + Rm[8] * wz + tz;
    const int m14[5] = {0, 1, 2, 4, 3};
#pragma unroll
    for (int n = 0; n < 5; ++n)
#pragma unroll
      for (int k = 0; k < 3; ++k) { a37[n * 3 + k] = pos[n][k]; a14[m14[n] * 3 + k] = pos[n][k]; }
  }
}

}  // namespace

extern "C" int fd_backbone_atoms(const float* rigids, const float* psi, const FdHeadConst* c, float* atom37,
                                 float* atom14, long R_, void* stream) {
  FD_CHECK_ARG(c != nullptr, "fd_backbone_atoms: null constants");
  if (R_ == 0) return FD_OK;
  AtomConst ac;
  for (int i = 0; i < 15; ++i) ac.atoms[i] = c->atoms[i];
  for (int i = 0; i < 9; ++i) ac.Rd[i] = c->Rd[i];
  for (int i = 0; i < 3; ++i) ac.td[i] = c->td[i];
  long g = (R_ + 255) / 256;
  hipLaunchKernelGGL(backbone_atoms_kernel, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, (hipStream_t)stream,
                     rigids, psi, ac, atom37, atom14, R_);
  FD_CHECK_LAUNCH("fd_backbone_atoms");
  return FD_OK;
}
