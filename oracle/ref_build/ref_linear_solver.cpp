// ref_linear_solver.cpp -- TEST INFRASTRUCTURE ONLY (oracle/_ref).  Stand-in for the one part of the reference that cannot be
// built here: cuba::SparseLinearSolver (/root/reference/src/cuda_linear_solver.h:28-39), implemented upstream by NVIDIA
// cuSOLVER's sparse Cholesky (src/cuda_linear_solver.cpp, closed source).  The contract is "solve Hsc x = bsc exactly":
// this version downloads the scalar CSR values the reference hands over, expands them to a dense symmetric matrix and
// factorises it with a plain Cholesky on the host (the checker graphs are small).  A non-positive pivot reports failure
// like the reference's "factorize failed" path (:406-410) -- without its sticky-flag quirk.
#include <cmath>
#include <cstdio>
#include <vector>

#include <cuda_runtime.h>      // the name shim

#include "cuda_linear_solver.h"

namespace cuba
{

class HostDenseCholesky : public SparseLinearSolver
{
public:
	void initialize(const HschurSparseBlockMatrix& Hsc) override
	{
		n_ = Hsc.rows();
		nnz_ = Hsc.nnzSymm();
		rowPtr_.assign(Hsc.rowPtr(), Hsc.rowPtr() + n_ + 1);
		colInd_.assign(Hsc.colInd(), Hsc.colInd() + nnz_);
	}

	bool solve(const Scalar* d_A, const Scalar* d_b, Scalar* d_x) override
	{
		std::vector<Scalar> vals(nnz_), b(n_);
		if (hipMemcpy(vals.data(), d_A, sizeof(Scalar) * nnz_, hipMemcpyDeviceToHost) != hipSuccess) return false;
		if (hipMemcpy(b.data(), d_b, sizeof(Scalar) * n_, hipMemcpyDeviceToHost) != hipSuccess) return false;
		const size_t n = (size_t)n_;
		std::vector<double> A(n * n, 0.0), x(n);
		for (int r = 0; r < n_; r++)
			for (int k = rowPtr_[r]; k < rowPtr_[r + 1]; k++) A[(size_t)r * n + colInd_[k]] = (double)vals[k];
		// Cholesky A = L L^T in place (lower triangle, row-major)
		for (size_t j = 0; j < n; j++)
		{
			double d = A[j * n + j];
			for (size_t k = 0; k < j; k++) d -= A[j * n + k] * A[j * n + k];
			if (!(d > 0)) { std::fprintf(stderr, "[ref stand-in solver] factorize failed\n"); return false; }
			const double l = std::sqrt(d);
			A[j * n + j] = l;
			for (size_t i = j + 1; i < n; i++)
			{
				double s = A[i * n + j];
				const double* ai = &A[i * n]; const double* aj = &A[j * n];
				for (size_t k = 0; k < j; k++) s -= ai[k] * aj[k];
				A[i * n + j] = s / l;
			}
		}
		for (size_t i = 0; i < n; i++)
		{
			double s = (double)b[i];
			for (size_t k = 0; k < i; k++) s -= A[i * n + k] * x[k];
			x[i] = s / A[i * n + i];
		}
		for (size_t ii = n; ii-- > 0;)
		{
			double s = x[ii];
			for (size_t k = ii + 1; k < n; k++) s -= A[k * n + ii] * x[k];
			x[ii] = s / A[ii * n + ii];
		}
		std::vector<Scalar> xs(x.begin(), x.end());
		return hipMemcpy(d_x, xs.data(), sizeof(Scalar) * n_, hipMemcpyHostToDevice) == hipSuccess;
	}

private:
	int n_ = 0, nnz_ = 0;
	std::vector<int> rowPtr_, colInd_;
};

SparseLinearSolver::Ptr SparseLinearSolver::create() { return std::make_unique<HostDenseCholesky>(); }
SparseLinearSolver::~SparseLinearSolver() {}

}  // namespace cuba
