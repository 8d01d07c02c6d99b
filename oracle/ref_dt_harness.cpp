// ref_dt_harness.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Thin extern "C" shim over the reference's own DTSource container code
// (compiled from /root/reference/DTSource where it lies, never copied), used
// only to validate the oracle's restatement of the pieces of the hot path that
// live in DTSource:
//   Maximum / Minimum        DTSource/DTDoubleArray.cpp:883-899, 923-939
//   operator+= (array,array) DTSource/DTArrayTemplates.h:157-167
//   Copy / operator=(double) DTSource/DTDoubleArray.cpp:340-347
//   column-major (i,j)       DTSource/DTDoubleArray.h:125-128
// The solver functions of main.cpp are NOT built here: main.cpp includes
// boost/program_options.hpp, which this image lacks.
#include "DTDoubleArray.h"
#include "DTDoubleArrayOperators.h"

#include <cstring>

extern "C" {

double ref_maximum(const double *a, long len)
{
    DTMutableDoubleArray arr(len);
    std::memcpy(arr.Pointer(), a, sizeof(double) * (size_t)len);
    return Maximum(arr);
}

double ref_minimum(const double *a, long len)
{
    DTMutableDoubleArray arr(len);
    std::memcpy(arr.Pointer(), a, sizeof(double) * (size_t)len);
    return Minimum(arr);
}

// a += b through the reference's operator, result copied back into a.
void ref_plus_equals(double *a, const double *b, long len)
{
    DTMutableDoubleArray A(len), B(len);
    std::memcpy(A.Pointer(), a, sizeof(double) * (size_t)len);
    std::memcpy(B.Pointer(), b, sizeof(double) * (size_t)len);
    A += B;
    std::memcpy(a, A.Pointer(), sizeof(double) * (size_t)len);
}

// Flat offset of entry (i,j) of an m x n DTDoubleArray, measured through the
// reference's own operator(): proves the column-major convention.
long ref_offset_of(int m, int n, int i, int j)
{
    DTMutableDoubleArray A(m, n);
    A = 0;
    A(i, j) = 1.0;
    const double *p = A.Pointer();
    for (long k = 0; k < (long)m * n; k++)
        if (p[k] == 1.0) return k;
    return -1;
}

} // extern "C"
