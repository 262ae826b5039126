// sla_hip.hpp -- header-only C++ host mirror of the reference's Numeric.LinearAlgebra.Sparse surface for
// the hot path, on top of the C ABI (sla_hip.h).  The reference is compiled code (Haskell) whose toolchain
// is absent from the authoring image, so the host side above the boundary is provided in C++ with the
// reference's names, argument order and error behaviour:
//
//   fromListSM / fromListDenseSM / fromListSV / fromListDenseSV      (SpMatrix.hs:218-241, SpVector.hs:194,275)
//   matVec (#>)   vecMat (<#)   dot (<.>)   norm2   normalize2       (Common.hs:242-256, SpVector.hs:116-129)
//   operator+ (^+^)  operator- (^-^)  scalar * (.*)                  (SpVector.hs:107-114)
//   linSolve0, LinSolveMethod, bicgsInit/bicgstabStep, cgsInit/cgsStep, cgneInit/cgneStep   (Sparse.hs:855-1072)
//   arnoldi, gmres, linSolve (<\>)                                   (Sparse.hs:630-667, 828-848, 1080-1084)
//
// Exceptions mirror Control/Exception/Common.hs: MatVecSizeMismatchException, IterationException (IterE);
// an out-of-bounds fromListSM throws std::out_of_range (the reference calls `error`).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "sla_hip.h"

namespace sla {

struct MatVecSizeMismatchException : std::runtime_error { using std::runtime_error::runtime_error; };
struct IterationException : std::runtime_error { using std::runtime_error::runtime_error; };
struct NeedsPivoting : std::runtime_error { using std::runtime_error::runtime_error; };  // Control/Exception/Common.hs:58
struct SlaError : std::runtime_error {
    int code;
    SlaError(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

inline void check(int rc) {
    if (rc == SLA_OK) return;
    const std::string msg = sla_last_error();
    switch (rc) {
        case SLA_ERR_DIM_MISMATCH: throw MatVecSizeMismatchException(msg);
        case SLA_ERR_UNSUPPORTED_METHOD: throw IterationException(msg);
        case SLA_ERR_OOB: throw std::out_of_range(msg);
        case SLA_ERR_NEEDS_PIVOTING: throw NeedsPivoting(msg);
        default: throw SlaError(rc, msg);
    }
}

enum class LinSolveMethod { GMRES_ = 0, CGNE_ = 1, BCG_ = 2, CGS_ = 3, BICGSTAB_ = 4 };  // Sparse.hs:1007-1011

class Context {
  public:
    explicit Context(int device = 0) { check(sla_ctx_create(device, &h_)); }
    ~Context() { sla_ctx_destroy(h_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    sla_ctx_t get() const { return h_; }
    // typed knob entry (sla_ctx_set_option): e.g. set_option("bicg_fuse45", "0") for the reference's literal K4 / K5 split
    Context &set_option(const std::string &name, const std::string &value) {
        check(sla_ctx_set_option(h_, name.c_str(), value.c_str()));
        return *this;
    }
    static Context &instance() {
        static Context c(0);
        return c;
    }

  private:
    sla_ctx_t h_ = nullptr;
};

// SpVector Double: dense on the device (solver inputs / outputs are dense in the reference's own usage)
class SpVector {
  public:
    SpVector() = default;
    explicit SpVector(int64_t n) : n_(n) {
        sla_vec_t v;
        check(sla_vec_create(Context::instance().get(), n, nullptr, &v));
        h_.reset(v, sla_vec_destroy);
    }
    SpVector(int64_t n, const std::vector<double> &dense) : n_(n) {
        std::vector<double> d(dense);
        d.resize((size_t)n, 0.0);
        sla_vec_t v;
        check(sla_vec_create(Context::instance().get(), n, d.data(), &v));
        h_.reset(v, sla_vec_destroy);
    }
    int64_t dim() const { return n_; }
    sla_vec_t get() const { return h_.get(); }
    std::vector<double> toDenseListSV() const {  // SpVector.hs:300
        std::vector<double> out((size_t)n_);
        if (n_) check(sla_vec_to_host(h_.get(), out.data()));
        return out;
    }
    SpVector clone() const {
        SpVector c(n_);
        check(sla_vec_copy(h_.get(), c.get()));
        return c;
    }

  private:
    int64_t n_ = 0;
    std::shared_ptr<sla_vec> h_;
};

inline SpVector fromListDenseSV(int64_t d, const std::vector<double> &ll) { return SpVector(d, ll); }
// fromListSV d iix: the FIRST duplicate wins (foldr insert), out-of-bounds entries are dropped (SpVector.hs:275-278)
inline SpVector fromListSV(int64_t d, const std::vector<std::pair<int64_t, double>> &iix) {
    std::vector<double> dense((size_t)d, 0.0);
    std::vector<char> seen((size_t)d, 0);
    for (const auto &e : iix)
        if (e.first >= 0 && e.first < d && !seen[(size_t)e.first]) { dense[(size_t)e.first] = e.second; seen[(size_t)e.first] = 1; }
    return SpVector(d, dense);
}

// (^+^) (^-^) (.*): new vectors, like the pure reference operators
inline SpVector axpby(double a, const SpVector &x, double b, const SpVector &y) {
    SpVector out = y.clone();
    check(sla_axpby(a, x.get(), b, out.get()));
    return out;
}
inline SpVector operator+(const SpVector &x, const SpVector &y) { return axpby(1.0, x, 1.0, y); }
inline SpVector operator-(const SpVector &x, const SpVector &y) { return axpby(1.0, x, -1.0, y); }
inline SpVector operator*(double a, const SpVector &x) {
    SpVector out = x.clone();
    check(sla_scal(a, out.get()));
    return out;
}
inline double dot(const SpVector &x, const SpVector &y) {
    double out;
    check(sla_dot(x.get(), y.get(), &out));
    return out;
}
inline double norm2(const SpVector &x) {
    double out;
    check(sla_nrm2(x.get(), &out));
    return out;
}
inline SpVector normalize2(const SpVector &x) { return (1.0 / norm2(x)) * x; }  // Class.hs:94-95
inline bool nearZero(double a) { return std::fabs(a) <= 1e-12; }                // Eps.hs:41-42

// SpMatrix Double, lowered once to the device CSR at construction
class SpMatrix {
  public:
    using Triple = std::tuple<int64_t, int64_t, double>;
    SpMatrix(int64_t m, int64_t n, const std::vector<Triple> &iix) : m_(m), n_(n) {
        std::vector<int64_t> r, c;
        std::vector<double> v;
        for (const auto &t : iix) { r.push_back(std::get<0>(t)); c.push_back(std::get<1>(t)); v.push_back(std::get<2>(t)); }
        sla_csr_t a;
        check(sla_csr_from_coo(Context::instance().get(), m, n, (int64_t)r.size(), r.data(), c.data(), v.data(), SLA_DUP_LAST_WINS, &a));
        h_.reset(a, sla_csr_destroy);
    }
    SpMatrix(int64_t m, int64_t n, sla_csr_t owned) : m_(m), n_(n) { h_.reset(owned, sla_csr_destroy); }   // adopt a library handle
    int64_t nrows() const { return m_; }
    int64_t ncols() const { return n_; }
    sla_csr_t get() const { return h_.get(); }
    // toListSM in ascending (row, col) order (the reference's is the reverse, SpMatrix.hs:251-253)
    std::vector<Triple> toAscList() const {
        int64_t nnz = 0, rows = 0;
        check(sla_csr_dims(h_.get(), nullptr, nullptr, &nnz, &rows));
        std::vector<int64_t> rp((size_t)rows + 1), ci((size_t)std::max<int64_t>(nnz, 1));
        std::vector<double> va((size_t)std::max<int64_t>(nnz, 1));
        check(sla_csr_export(h_.get(), rp.data(), ci.data(), va.data()));
        std::vector<Triple> out;
        for (int64_t i = 0; i < rows; ++i)
            for (int64_t k = rp[(size_t)i]; k < rp[(size_t)i + 1]; ++k) out.emplace_back(i, ci[(size_t)k], va[(size_t)k]);
        return out;
    }
    bool isDiagonalSM() const {  // SpMatrix.hs:411-415
        int d;
        check(sla_csr_is_diagonal(h_.get(), &d));
        return d != 0;
    }
    // CscMatrix arrays (vector/src/Data/Sparse/Internal/CSC.hs:17-24) in and out
    static SpMatrix fromCSC(int64_t m, int64_t n, const std::vector<int64_t> &colptr, const std::vector<int64_t> &rowidx, const std::vector<double> &val) {
        sla_csr_t a;
        check(sla_csr_from_csc(Context::instance().get(), m, n, colptr.data(), rowidx.data(), val.data(), &a));
        return SpMatrix(m, n, a);
    }
    void toCSC(std::vector<int64_t> &colptr, std::vector<int64_t> &rowidx, std::vector<double> &val) const {
        int64_t nnz = 0;
        check(sla_csr_dims(h_.get(), nullptr, nullptr, &nnz, nullptr));
        colptr.assign((size_t)n_ + 1, 0);
        rowidx.assign((size_t)std::max<int64_t>(nnz, 1), 0);
        val.assign((size_t)std::max<int64_t>(nnz, 1), 0.0);
        check(sla_csr_export_csc(h_.get(), colptr.data(), rowidx.data(), val.data()));
        rowidx.resize((size_t)nnz);
        val.resize((size_t)nnz);
    }
    SpMatrix transposeSM() const {  // SpMatrix.hs:717
        sla_csr_t t;
        check(sla_csr_transpose(h_.get(), &t));
        return SpMatrix(n_, m_, t);
    }

  private:
    int64_t m_, n_;
    std::shared_ptr<sla_csr> h_;
};

inline SpMatrix fromListSM(std::pair<int64_t, int64_t> dims, const std::vector<SpMatrix::Triple> &iix) {
    return SpMatrix(dims.first, dims.second, iix);
}
// fromListDenseSM m ll: column-major, entry k -> (k mod m, k div m)   (SpMatrix.hs:239-241)
inline SpMatrix fromListDenseSM(int64_t m, const std::vector<double> &ll) {
    const int64_t n = (int64_t)ll.size() / m;
    std::vector<SpMatrix::Triple> t;
    for (int64_t k = 0; k < m * n; ++k) t.emplace_back(k % m, k / m, ll[(size_t)k]);
    return SpMatrix(m, n, t);
}

// ilu0Pre aa (Sparse.hs:696-706): (L, U); exact = the reference's complete-LU-then-filter definition (<= 4096 rows)
inline std::pair<SpMatrix, SpMatrix> ilu0Pre(const SpMatrix &A, bool exact = true) {
    sla_csr_t l, u;
    check(sla_ilu0_pre(A.get(), exact ? 1 : 0, &l, &u, nullptr));
    return {SpMatrix(A.nrows(), A.ncols(), l), SpMatrix(A.nrows(), A.ncols(), u)};
}

// m1 ## m2 and m1 ##^ m2 (matMat_ AB / ABt, SpMatrix.hs:768-811); size mismatch throws like `error "matMat : ..."`
inline SpMatrix matMat(const SpMatrix &A, const SpMatrix &B) {
    sla_csr_t c;
    check(sla_csr_matmat(A.get(), B.get(), 0, &c));
    return SpMatrix(A.nrows(), B.ncols(), c);
}
inline SpMatrix matMatT(const SpMatrix &A, const SpMatrix &B) {
    sla_csr_t c;
    check(sla_csr_matmat(A.get(), B.get(), 1, &c));
    return SpMatrix(A.nrows(), B.nrows(), c);
}

inline SpVector matVec(const SpMatrix &A, const SpVector &x) {  // A #> x
    SpVector y(A.nrows());
    check(sla_spmv(A.get(), x.get(), y.get()));
    return y;
}
inline SpVector vecMat(const SpVector &x, const SpMatrix &A) {  // x <# A
    SpVector y(A.ncols());
    check(sla_spmv_t(A.get(), x.get(), y.get()));
    return y;
}

// forward / backward substitution with one triangle of T (Sparse.hs:750-811); throws NeedsPivoting
inline SpVector triLowerSolve(const SpMatrix &ll, const SpVector &b) {
    SpVector x(ll.nrows());
    check(sla_tri_solve(ll.get(), 0, b.get(), x.get(), nullptr));
    return x;
}
inline SpVector triUpperSolve(const SpMatrix &uu, const SpVector &w) {
    SpVector x(uu.nrows());
    check(sla_tri_solve(uu.get(), 1, w.get(), x.get(), nullptr));
    return x;
}

// solver state records
class SolverState {
  public:
    SolverState(LinSolveMethod m, const SpMatrix &A, const SpVector &b, const SpVector &x0) : A_(A) {
        sla_solver_t s;
        check(sla_solver_init((int)m, A.get(), b.get(), x0.get(), &s));
        h_.reset(s, sla_solver_destroy);
    }
    SolverState &step(int k = 1) {   // k steps in place (the fast path)
        check(sla_solver_step(h_.get(), k));
        return *this;
    }
    // the reference's pure step (bicgstabStep aa r0hat s / cgsStep aa rhat s, Sparse.hs:928, :972): a NEW record, *this untouched
    SolverState stepped(const SpVector *r0hat = nullptr, int k = 1) const {
        SolverState out(*this);
        sla_solver_t t;
        check(sla_solver_clone(h_.get(), &t));
        out.h_.reset(t, sla_solver_destroy);
        if (r0hat) check(sla_solver_set_shadow(t, r0hat->get()));
        check(sla_solver_step(t, k));
        return out;
    }
    SpVector field(int f, int64_t n) const {
        SpVector out(n);
        check(sla_solver_get(h_.get(), f, out.get()));
        return out;
    }
    SpVector _x() const { return field(SLA_STATE_X, A_.ncols()); }
    SpVector _r() const { return field(SLA_STATE_R, A_.nrows()); }
    SpVector _p() const { return field(SLA_STATE_P, A_.ncols()); }
    SpVector _u() const { return field(SLA_STATE_U, A_.nrows()); }
    SpVector _rHat() const { return field(SLA_STATE_RHAT, A_.nrows()); }   // BCG records only (Sparse.hs:886-887)
    SpVector _pHat() const { return field(SLA_STATE_PHAT, A_.nrows()); }

  private:
    SpMatrix A_;
    std::shared_ptr<sla_solver> h_;
};
inline SolverState bicgsInit(const SpMatrix &A, const SpVector &b, const SpVector &x0) { return SolverState(LinSolveMethod::BICGSTAB_, A, b, x0); }
inline SolverState cgsInit(const SpMatrix &A, const SpVector &b, const SpVector &x0) { return SolverState(LinSolveMethod::CGS_, A, b, x0); }
inline SolverState cgneInit(const SpMatrix &A, const SpVector &b, const SpVector &x0) { return SolverState(LinSolveMethod::CGNE_, A, b, x0); }
// extension: the reference's bcgInit / bcgStep are commented out (Sparse.hs:889-909); linSolve0(BCG_) throws like the reference
inline SolverState bcgInit(const SpMatrix &A, const SpVector &b, const SpVector &x0) { return SolverState(LinSolveMethod::BCG_, A, b, x0); }
inline SolverState &bcgStep(SolverState &s, int k = 1) { return s.step(k); }
inline SolverState &bicgstabStep(SolverState &s, int k = 1) { return s.step(k); }
inline SolverState &cgsStep(SolverState &s, int k = 1) { return s.step(k); }
inline SolverState &cgneStep(SolverState &s, int k = 1) { return s.step(k); }

// linSolve0 method aa b x0   (Sparse.hs:1016-1072)
inline SpVector linSolve0(LinSolveMethod method, const SpMatrix &aa, const SpVector &b, const SpVector &x0,
                          sla_solve_info *info = nullptr, const sla_solve_opts *opts = nullptr) {
    SpVector x(aa.ncols());
    check(sla_linsolve0((int)method, aa.get(), b.get(), x0.get(), opts, x.get(), info));
    return x;
}

struct ArnoldiResult {
    int k;                  // H columns produced (< kn after a breakdown)
    std::vector<double> Q;  // n x (k+1), column-major
    std::vector<double> H;  // (kn+1) x kn, column-major, leading dimension kn+1
};
inline ArnoldiResult arnoldi(const SpMatrix &aa, const SpVector &b, int kn) {  // Sparse.hs:630-667
    ArnoldiResult r;
    r.Q.assign((size_t)aa.ncols() * (size_t)(kn + 1), 0.0);
    r.H.assign((size_t)(kn + 1) * (size_t)kn, 0.0);
    check(sla_arnoldi(aa.get(), b.get(), kn, r.Q.data(), r.H.data(), &r.k));
    r.Q.resize((size_t)aa.ncols() * (size_t)(r.k + 1));
    return r;
}
inline SpVector gmres(const SpMatrix &aa, const SpVector &b, const SpVector &x0, int restart = 30, sla_solve_info *info = nullptr) {
    SpVector x(aa.ncols());
    check(sla_gmres(aa.get(), b.get(), x0.get(), restart, nullptr, x.get(), info));
    return x;
}
inline SpVector linSolve(const SpMatrix &aa, const SpVector &b, sla_solve_info *info = nullptr) {  // aa <\> b
    SpVector x(aa.ncols());
    check(sla_linsolve(aa.get(), b.get(), x.get(), info));
    return x;
}

}  // namespace sla
