// ppo_chol.hpp -- dense SPD factor + solve of the reduced system on one stream.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace ppo {
// A: ld x ld column-major (ld = chol_padded_ld(n), pad diagonal = 1), lower triangle valid on entry, overwritten by L.  b: right-hand side (left intact;
// must not alias x), x: solution, workspace: chol_workspace_doubles(n) doubles.  *info (device int) = 0 when A is positive definite, k > 0 when the leading minor of order k is not.
int chol_factor_solve(double* A, int n, const double* b, double* x, int* info, double* workspace, hipStream_t stream);
size_t chol_workspace_doubles(int n);
int chol_padded_ld(int n);   // leading dimension the native solver needs (multiple of the panel width)
void chol_release(double* A);  // drop the cached launch graph of a matrix that is about to be freed
int chol_default_backend();
}  // namespace ppo
