// Host fp64 symmetric eigensolver (see eigen_sym.cpp).  Internal to libnle_hip.so.
#pragma once

namespace nleh {

// M: n x n column-major symmetric (lower triangle read).  U: n x n column-major
// eigenvectors (columns), D: eigenvalues ASCENDING.  false = QL did not converge.
bool sym_eigen(const double* M, int n, double* U, double* D);

// The reference's eigenDecomposition (src/filter.cpp:204-228): eigenvalues DESCENDING,
// *r = length of the leading run with D >= eps (columns/values beyond *r are still
// filled but not part of the result).
bool eigen_decomposition(const double* M, int n, double eps, double* U, double* D, int* r);

}  // namespace nleh
