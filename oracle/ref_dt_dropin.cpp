// ref_dt_dropin.cpp -- TEST INFRASTRUCTURE ONLY.
//
// The reference-side binding of libmgx, compiled against the reference's REAL DTSource
// containers (the sources are compiled where they lie, oracle/Makefile; output in
// oracle/_ref/).  It is what a maintainer of the reference would have after swapping the
// body of MultiGrid() (main.cpp:283-357) for one call into the C ABI: the same
// gridtype / MGOutputs aggregates (main.cpp:29-40), DTMatlabDataFile for Input.mat and
// Output.mat (main.cpp:391-396,452-455), a zero initial guess with boundary_func == 0 on
// the boundary (main.cpp:420-439).  Option parsing is positional here because the
// reference's uses boost, which the image lacks:
//     dt_dropin Nv Ndown Nup omega coarsest [ngpus [tail_m]]   (cwd holds Input.mat; writes Output.mat)
// ngpus > 1 sets mgx_options.ngpus: the same call then runs over that many j-slabs.
// tests/test_gpu_dropin.py runs it on the GPU box and compares Output.mat with the oracle.
#include "DTDoubleArray.h"
#include "DTMatlabDataFile.h"
#include "DTMesh2D.h"
#include "DTMesh2DGrid.h"

#include <cstdio>
#include <cstdlib>

#include "mgx_multigrid.hpp"

// the two aggregates MultiGrid() trades in (main.cpp:29-40)
struct gridtype {
    DTMutableMesh2D f;  // right-hand side
    DTMutableMesh2D v;  // initial guess in, solution out
};

struct MGOutputs {
    DTDoubleArray ResidualNorms;
    DTDoubleArray Times;
    MGOutputs(DTDoubleArray r, DTDoubleArray t) : ResidualNorms(r), Times(t) {}
};

// Drop-in for main.cpp:283: same signature, same in-place solution, same two output vectors;
// the body is include/mgx_multigrid.hpp, the binding this repository ships.
static mgx_options g_options;  // zero = defaults; main() may set ngpus / tail_m

MGOutputs MultiGrid(gridtype &prob, int Nv, int Ndown, int Nup, double omega, int coarsest, bool pureJacobi = false)
{
    return mgx::MultiGrid<MGOutputs, DTMutableDoubleArray>(prob, Nv, Ndown, Nup, omega, coarsest, pureJacobi,
                                                           g_options.struct_size ? &g_options : NULL);
}

int main(int argc, char **argv)
{
    if (argc < 6 || argc > 8) {
        fprintf(stderr, "usage: dt_dropin Nv Ndown Nup omega coarsest [ngpus [tail_m]]\n");
        return 2;
    }
    if (argc > 6) {
        g_options.struct_size = (int)sizeof(g_options);
        g_options.device = -1;
        g_options.ngpus = atoi(argv[6]);
        g_options.tail_m = argc > 7 ? atoi(argv[7]) : 0;
    }
    const int Nv = atoi(argv[1]), Ndown = atoi(argv[2]), Nup = atoi(argv[3]), coarsest = atoi(argv[5]);
    const double omega = atof(argv[4]);

    DTMatlabDataFile inputFile("Input.mat", DTFile::ReadOnly);
    DTMesh2D f;
    Read(inputFile, "f", f);
    DTMesh2DGrid grid = f.Grid();
    DTDoubleArray fData = f.DoubleData();
    const int M = (int)fData.m(), N = (int)fData.n();

    DTMutableDoubleArray u(M, N);
    u = 0;  // zero guess, zero Dirichlet data (boundary_func, main.cpp:43-47)

    gridtype problem;
    problem.f = DTMutableMesh2D(grid, fData.Copy());
    problem.v = DTMutableMesh2D(grid, u.Copy());
    MGOutputs output = MultiGrid(problem, Nv, Ndown, Nup, omega, coarsest, false);

    DTMatlabDataFile outputFile("Output.mat", DTFile::NewReadWrite);
    outputFile.Save(problem.v.DoubleData(), "Sol");
    outputFile.Save(output.ResidualNorms, "ResNorms");
    outputFile.Save(output.Times, "Times");
    return 0;
}
