// mgx_multigrid.hpp -- header-only C++ binding of libmgx for a host that holds its data in
// DTSource containers (or anything shaped like them): the body of the reference's
//
//     MGOutputs MultiGrid(gridtype &prob, int Nv, int Ndown, int Nup, double omega,
//                         int coarsest, bool pureJacobi)            (main.cpp:283-357)
//
// as ONE call into the C ABI.  It is a template so that this repository does not depend
// on DTSource: instantiate it with the reference's own `gridtype` and `MGOutputs`
// (main.cpp:29-40) and `DTMutableDoubleArray`, e.g. in main.cpp
//
//     #include "mgx_multigrid.hpp"
//     MGOutputs MultiGrid(gridtype &prob, int Nv, int Ndown, int Nup, double omega, int coarsest, bool pureJacobi = false)
//     { return mgx::MultiGrid<MGOutputs, DTMutableDoubleArray>(prob, Nv, Ndown, Nup, omega, coarsest, pureJacobi); }
//
// Requirements on the types (all met by DTSource):
//   prob.f.DoubleData(), prob.v.DoubleData()  arrays with Pointer(), m(), n(); column-major,
//                                             sharing storage with the mesh (DTDoubleArray.h:38-49,125-128;
//                                             DTMesh2D.cpp:553-559), so the solution lands in prob.v
//   prob.v.Grid().dx()                        the mesh width (dy is never used, main.cpp:180,263)
//   Array(n)                                  a mutable fp64 array of n entries with Pointer()
//   Outputs(Array, Array)                     ResidualNorms and Times, Nv+1 entries each (main.cpp:305-306)
// Errors: the reference prints and exits (main.cpp:408-417); so does this, with libmgx's message.
#ifndef MGX_MULTIGRID_HPP
#define MGX_MULTIGRID_HPP

#include <cstdio>
#include <cstdlib>

#include "mgx.h"

namespace mgx {

template <class Outputs, class Array, class Grid>
Outputs MultiGrid(Grid &prob, int Nv, int Ndown, int Nup, double omega, int coarsest, bool pureJacobi = false,
                  const mgx_options *options = NULL)
{
    auto v = prob.v.DoubleData();  // shares storage with prob.v (main.cpp:288)
    auto f = prob.f.DoubleData();
    Array resnorm(Nv + 1), times(Nv + 1);
    const int rc = mgx_solve2d(f.Pointer(), v.Pointer(), (int)v.m(), (int)v.n(), prob.v.Grid().dx(), Nv, Ndown, Nup, omega,
                               coarsest, pureJacobi ? 1 : 0, resnorm.Pointer(), times.Pointer(), options);
    if (rc != MGX_OK) {
        std::printf("Error: %s\n", mgx_last_error());
        std::exit(1);
    }
    return Outputs(resnorm, times);
}

} // namespace mgx

#endif /* MGX_MULTIGRID_HPP */
