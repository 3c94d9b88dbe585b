// ppo_chol.hpp -- dense SPD factor + solve of the reduced system on one stream.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "ppo_pairs.hpp"

namespace ppo {
// A: ld x ld column-major (ld = chol_padded_ld(n), pad diagonal = 1), lower triangle valid on entry, overwritten by L.  b: right-hand side (left intact;
// must not alias x), x: solution, workspace: chol_workspace_doubles(n) doubles.  *info (device int) = 0 when A is positive definite, k > 0 when the leading minor of order k is not.
// gate: LM control on the device -- the launches do nothing unless *gate.epoch_word == gate.expect (NULL: always live).
// n_active (optional): the number of rows in front of the padding (<= n): the one-launch solve of small systems skips whole 16-row tiles of padding.
int chol_factor_solve(double* A, int n, const double* b, double* x, int* info, double* workspace, hipStream_t stream, LmGate gate = LmGate{nullptr, 0}, int n_active = -1);
// small systems (at most 96 of at most 192 unknowns, see ppo_chol.hip): the whole solve is ONE launch of one workgroup, the matrix in LDS and registers
// (ppo_chol_small.inc); an overlapped reduction of such a window must be scheduled with fin[0] = ld / 64 (everything reduced ahead)
bool chol_small_applies(int ld, int n_active);
// Overlapped reduction (ppo_chol.hip, StepOverlap): only the pair chunks of the first fin[0] 64-column blocks of A have been reduced when this is
// called; the launches of the factorisation carry the remaining chunks (ordered by column block: cb_start[c] = first chunk of block c, host
// array of ld / 64 + 1 entries) as extra workgroups that add to A and to the right-hand side ctx.bs, which is solved IN PLACE.
// fin: host array of ld / 64 + 1 entries from chol_overlap_schedule().
struct CholOverlap {
  PairCtx ctx;  // ctx.S == A
  PairChunks pc;
  const int32_t* cb_start;
  const int* fin;
  LmGate gate;
};
int chol_factor_solve_overlapped(double* A, int n, double* x, int* info, double* workspace, const CholOverlap& ov, hipStream_t stream, int n_active = -1);
// Deterministic mode (options.deterministic): the reduction summed the system as 2^-24 fixed-point integers into Afix (the shape of A, lower triangle,
// pad diagonal = 1.0 in fixed point) and bfix; the factorisation converts on load and writes A, which needs NO initialisation (ppo_chol.hip: StepFix).
// ov (may be NULL): an overlapped reduction as above whose pair workgroups add integers: ov->ctx.S / ov->ctx.bs are Afix / bfix seen as double*.
struct CholFixed {
  const long long* Afix;
  const long long* bfix;
  double* detbad;  // device word raised by an addend that is not finite or does not fit the fixed-point range (ACC_DETRANGE: the system is not usable)
};
int chol_factor_solve_fixed(double* A, int n, double* x, int* info, double* workspace, const CholFixed& fixed, const CholOverlap* ov, LmGate gate, hipStream_t stream, int n_active = -1);
void chol_overlap_schedule(int nb, const int32_t* cb_start, int first_blocks, double share0, double share, int* fin);
size_t chol_workspace_doubles(int n);
int chol_padded_ld(int n);   // leading dimension the native solver needs (multiple of the panel width)
void chol_release(double* A);  // drop the cached launch graph of a matrix that is about to be freed
int chol_default_backend();
}  // namespace ppo
