// reference_cases.cpp -- the reference's own solver tests (test/LibSpec.hs:286-321, README.md:183-241) written
// against the C++ host mirror include/sla_hip.hpp.  Build:
//   g++ -std=c++17 -Iinclude examples/reference_cases.cpp -Lsparse-linear-algebra_amd/lib -lsla_hip \
//       -Wl,-rpath,$PWD/sparse-linear-algebra_amd/lib -o examples/reference_cases
#include <cstdio>
#include <cstdlib>

#include "sla_hip.hpp"

using namespace sla;

static int failures = 0;
#define EXPECT(cond)                                               \
    do {                                                           \
        if (!(cond)) { ++failures; std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #cond); } \
    } while (0)

static SpVector mkSpVR(int64_t n, const std::vector<double> &v) { return fromListDenseSV(n, v); }

// checkLinSolveR (LibSpec.hs:309-321): x0 = 0.1 * ones ; nearZero (norm2 (x ^-^ xhat))
static bool checkLinSolveR(LinSolveMethod m, const SpMatrix &aa, const SpVector &b, const SpVector &x) {
    const int64_t n = aa.ncols();
    SpVector xhat = linSolve0(m, aa, b, mkSpVR(n, std::vector<double>((size_t)n, 0.1)));
    return nearZero(norm2(x - xhat));
}

int main() {
    // README.md:97,183-241
    SpMatrix amat = fromListSM({3, 3}, {{0, 0, 2}, {1, 0, 4}, {1, 1, 3}, {1, 2, 2}, {2, 2, 5}});
    SpVector b = fromListDenseSV(3, {3, 2, 5});
    SpVector x = linSolve(amat, b);                                   // amat <\> b
    auto xd = x.toDenseListSV();
    EXPECT(std::fabs(xd[0] - 1.5) < 1e-10 && std::fabs(xd[1] + 2.0) < 1e-10 && std::fabs(xd[2] - 1.0) < 1e-10);
    auto yd = matVec(amat, x).toDenseListSV();                        // amat #> x = [3,2,5]
    EXPECT(std::fabs(yd[0] - 3) < 1e-9 && std::fabs(yd[1] - 2) < 1e-9 && std::fabs(yd[2] - 5) < 1e-9);
    SpVector x0 = fromListSV(3, {});
    SolverState s = bicgsInit(amat, b, x0);
    bicgstabStep(s, 3);
    EXPECT(nearZero(norm2(s._x() - mkSpVR(3, {1.5, -2, 1}))));
    SolverState c = cgsInit(amat, b, x0);
    cgsStep(c, 3);
    EXPECT(nearZero(norm2(c._x() - mkSpVR(3, {1.5, -2, 1}))));

    // the reference's PURE steps: iterate (bicgstabStep aa r0hat) s0 keeps every element (README.md:222-226)
    {
        SolverState s0 = bicgsInit(amat, b, x0);
        SolverState s1 = s0.stepped(), s3 = s1.stepped().stepped();
        EXPECT(norm2(s0._x() - x0) == 0.0);                                    // s0 is still s0
        EXPECT(nearZero(norm2(s3._x() - mkSpVR(3, {1.5, -2, 1}))));
        EXPECT(norm2(s1._x() - s3._x()) > 1e-6);
    }
    // CscMatrix arrays in and out, transposeSM as a handle (vector/src/Data/Sparse/Internal/CSC.hs:17-24, :121-125: the triplet example at the foot of the file)
    {
        SpMatrix a = SpMatrix::fromCSC(3, 3, {0, 2, 3, 6}, {0, 2, 2, 0, 1, 2}, {1, 4, 5, 2, 3, 6});
        auto l = a.toAscList();
        EXPECT(l.size() == 6 && std::get<1>(l[1]) == 2 && std::get<2>(l[1]) == 2.0 && std::get<0>(l[3]) == 2 && std::get<2>(l[3]) == 4.0);
        std::vector<int64_t> cp, ri;
        std::vector<double> va;
        a.toCSC(cp, ri, va);
        EXPECT((cp == std::vector<int64_t>{0, 2, 3, 6}) && (ri == std::vector<int64_t>{0, 2, 2, 0, 1, 2}) && (va == std::vector<double>{1, 4, 5, 2, 3, 6}));
        auto t = a.transposeSM().toAscList();
        EXPECT(t.size() == 6 && std::get<0>(t[1]) == 0 && std::get<1>(t[1]) == 2 && std::get<2>(t[1]) == 4.0);
    }
    // m1 ## m2 (LibSpec.hs:61-62, fixtures :1263-1271) and the size check of matMat_ (SpMatrix.hs:795)
    {
        SpMatrix m1 = fromListDenseSM(2, {1, 3, 2, 4}), m2 = fromListDenseSM(2, {5, 7, 6, 8});
        auto c = matMat(m1, m2).toAscList();   // m1m2 = fromListDenseSM 2 [19,43,22,50]
        EXPECT(c.size() == 4 && std::get<2>(c[0]) == 19.0 && std::get<2>(c[1]) == 22.0 && std::get<2>(c[2]) == 43.0 && std::get<2>(c[3]) == 50.0);
        bool bad = false;
        try { matMat(m1, fromListSM({3, 2}, {{0, 0, 1.0}})); } catch (const MatVecSizeMismatchException &) { bad = true; }
        EXPECT(bad);
    }

    // specLinSolve (LibSpec.hs:286-300): aa0 2x2 dense, aa2 3x3 SPD tridiagonal
    SpMatrix aa0 = fromListDenseSM(2, {1, 3, 2, 4});
    SpMatrix aa2 = fromListSM({3, 3}, {{0, 0, 2}, {1, 0, -1}, {0, 1, -1}, {1, 1, 2}, {2, 1, -1}, {1, 2, -1}, {2, 2, 2}});
    for (LinSolveMethod m : {LinSolveMethod::BICGSTAB_, LinSolveMethod::CGS_, LinSolveMethod::CGNE_}) {
        EXPECT(checkLinSolveR(m, aa0, mkSpVR(2, {8, 18}), mkSpVR(2, {2, 3})));
        EXPECT(checkLinSolveR(m, aa2, mkSpVR(3, {4, -2, 4}), mkSpVR(3, {3, 2, 3})));
    }
    EXPECT(dot(mkSpVR(2, {5, 6}), mkSpVR(2, {5, 6})) == 61.0);         // tv0 <.> tv0 (LibSpec.hs:45-46)

    // error behaviour (Sparse.hs:1022, :1031)
    bool threw = false;
    try { linSolve0(LinSolveMethod::BICGSTAB_, amat, mkSpVR(4, {1, 1, 1, 1}), x0); } catch (const MatVecSizeMismatchException &) { threw = true; }
    EXPECT(threw);
    threw = false;
    try { linSolve0(LinSolveMethod::GMRES_, amat, b, x0); } catch (const IterationException &) { threw = true; }
    EXPECT(threw);
    threw = false;
    try { fromListSM({2, 2}, {{0, 0, 1.0}, {2, 0, 1.0}}); } catch (const std::out_of_range &) { threw = true; }
    EXPECT(threw);

    // arnoldi (LibSpec.hs:229-232): aa4, kn = 3
    SpMatrix aa4 = fromListDenseSM(3, {3, 2, -2, 2, 2, -1, 6, 5, -4});
    ArnoldiResult ar = arnoldi(aa4, mkSpVR(3, {1, 1, 1}), 3);
    EXPECT(ar.k >= 1 && ar.k <= 3);

    // specTriangularSolve (LibSpec.hs:203-213, fixtures :1409-1434): nearZero || T xhat - b ||
    SpMatrix ltri1 = fromListSM({3, 3}, {{0, 0, 2}, {1, 0, 1}, {1, 1, 4}, {2, 1, 2}, {2, 2, 3}});
    SpMatrix utri1 = fromListSM({3, 3}, {{0, 0, 2}, {0, 1, 1}, {0, 2, 1}, {1, 1, 4}, {1, 2, 2}, {2, 2, 3}});
    SpVector bl = fromListDenseSV(3, {4, 10, 17}), bu = fromListDenseSV(3, {9, 14, 9});
    EXPECT(nearZero(norm2(matVec(ltri1, triLowerSolve(ltri1, bl)) - bl)));
    EXPECT(nearZero(norm2(matVec(utri1, triUpperSolve(utri1, bu)) - bu)));
    threw = false;
    try { triLowerSolve(fromListSM({2, 2}, {{0, 0, 1.0}, {1, 0, 1.0}}), mkSpVR(2, {1, 1})); } catch (const NeedsPivoting &) { threw = true; }
    EXPECT(threw);

    std::printf(failures ? "reference_cases: %d FAILED\n" : "reference_cases: all passed\n", failures);
    return failures ? 1 : 0;
}
