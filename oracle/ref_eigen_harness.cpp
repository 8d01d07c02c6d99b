// ref_eigen_harness.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Exact solve of the 5-point Dirichlet problem on one grid with the sparse
// Cholesky factorisation the reference uses in getSparseSol() (main.cpp:79-128:
// 4 on the diagonal, -1 per interior neighbour, right-hand side -h^2 f plus the
// boundary values, Eigen::SimplicialCholesky).  Eigen is vendored header-only in
// the reference checkout and is compiled from there (oracle/Makefile, output in
// oracle/_ref/); the assembly below is written for flat column-major arrays.
// Used to pin the oracle's own banded Cholesky (mgo_coarse_exact).
#include <vector>

#include <Eigen/Sparse>

extern "C" int ref_poisson_exact(const double *f, double *u, int m, double h)
{
    const int n = m - 2;  // unknowns per side
    if (n < 1) return -1;
    const int N = n * n;
    typedef Eigen::Triplet<double> Trip;
    std::vector<Trip> trips;
    trips.reserve((size_t)5 * N);
    Eigen::VectorXd rhs(N);
    const double h2 = h * h;
    for (int j = 1; j <= n; j++) {
        for (int i = 1; i <= n; i++) {
            const int row = (i - 1) + (j - 1) * n;
            trips.push_back(Trip(row, row, 4.0));
            double b = -h2 * f[i + (size_t)j * m];
            // each neighbour is an unknown (matrix entry) or a boundary value (moved to the right)
            if (i > 1) trips.push_back(Trip(row, row - 1, -1.0)); else b += u[(i - 1) + (size_t)j * m];
            if (i < n) trips.push_back(Trip(row, row + 1, -1.0)); else b += u[(i + 1) + (size_t)j * m];
            if (j > 1) trips.push_back(Trip(row, row - n, -1.0)); else b += u[i + (size_t)(j - 1) * m];
            if (j < n) trips.push_back(Trip(row, row + n, -1.0)); else b += u[i + (size_t)(j + 1) * m];
            rhs[row] = b;
        }
    }
    Eigen::SparseMatrix<double> A(N, N);
    A.setFromTriplets(trips.begin(), trips.end());
    Eigen::SimplicialCholesky<Eigen::SparseMatrix<double> > chol(A);
    if (chol.info() != Eigen::Success) return -2;
    Eigen::VectorXd x = chol.solve(rhs);
    for (int j = 1; j <= n; j++)
        for (int i = 1; i <= n; i++) u[i + (size_t)j * m] = x[(i - 1) + (j - 1) * n];
    return 0;
}
