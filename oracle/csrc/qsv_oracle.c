/*
 * TEST INFRASTRUCTURE ONLY -- plain-C restatement of the reference's gate-application path, used as the
 * checker at sizes where NumPy temporaries are too slow and as bench.py's `cpu_baseline` ("port").
 *
 * The reference computes  state' = expand_gate(matrix, N, indices) @ state
 * (simulators/dv_simulator/gates.py:44-54, numpy_quantum.py:243-247).  The product with the expanded operator
 * touches, for every assignment of the non-target qubits, only the 2^k amplitudes that differ in the target
 * qubits; the loops below do exactly that.  Qubit q of n is bit (n-1-q) of the flat index (qubit 0 = MSB);
 * leg 0 of a 2-qubit matrix is the more significant one (kron(gate, I, ...) + `targets` order).
 *
 * Pinned against the reference's golden vectors by tests/test_oracle_golden.py::test_c_oracle_*.
 * Never linked into libqsv.so; built into oracle/libqsv_oracle.so by oracle/Makefile.
 */
#include <complex.h>
#include <omp.h>
#include <stdint.h>

typedef double _Complex cplx;

static inline uint64_t insert_zero(uint64_t w, int p) {
    return ((w >> p) << (p + 1)) | (w & ((1ull << p) - 1ull));
}

/* m: 2x2 row-major interleaved complex; state: 2^n interleaved complex, updated in place */
void oracle_apply_1q(double *state_, int n, int q, const double *m_) {
    cplx *state = (cplx *)state_;
    const cplx *m = (const cplx *)m_;
    const int bit = n - 1 - q;
    const uint64_t s = 1ull << bit, pairs = 1ull << (n - 1);
#pragma omp parallel for schedule(static)
    for (uint64_t w = 0; w < pairs; ++w) {
        const uint64_t i0 = insert_zero(w, bit), i1 = i0 | s;
        const cplx a0 = state[i0], a1 = state[i1];
        state[i0] = m[0] * a0 + m[1] * a1;
        state[i1] = m[2] * a0 + m[3] * a1;
    }
}

/* m: 4x4 row-major interleaved complex, row/col index = (bit of q0) * 2 + (bit of q1) */
void oracle_apply_2q(double *state_, int n, int q0, int q1, const double *m_) {
    cplx *state = (cplx *)state_;
    const cplx *m = (const cplx *)m_;
    const int b0 = n - 1 - q0, b1 = n - 1 - q1;
    const int lo = b0 < b1 ? b0 : b1, hi = b0 < b1 ? b1 : b0;
    const uint64_t s0 = 1ull << b0, s1 = 1ull << b1, groups = 1ull << (n - 2);
#pragma omp parallel for schedule(static)
    for (uint64_t w = 0; w < groups; ++w) {
        const uint64_t base = insert_zero(insert_zero(w, lo), hi);
        const uint64_t idx[4] = {base, base | s1, base | s0, base | s0 | s1};
        cplx a[4], r[4];
        for (int c = 0; c < 4; ++c) a[c] = state[idx[c]];
        for (int row = 0; row < 4; ++row) {
            cplx acc = 0;
            for (int c = 0; c < 4; ++c) acc += m[row * 4 + c] * a[c];
            r[row] = acc;
        }
        for (int c = 0; c < 4; ++c) state[idx[c]] = r[c];
    }
}

/* The literal dense algorithm at small N (calibration only): out = U_full @ in with U_full given dense,
 * as `gate @ state` at gates.py:50.  u: 2^n x 2^n row-major interleaved. */
void oracle_dense_matvec(const double *u_, const double *in_, double *out_, int n) {
    const cplx *u = (const cplx *)u_, *in = (const cplx *)in_;
    cplx *out = (cplx *)out_;
    const uint64_t dim = 1ull << n;
#pragma omp parallel for schedule(static)
    for (uint64_t r = 0; r < dim; ++r) {
        cplx acc = 0;
        for (uint64_t c = 0; c < dim; ++c) acc += u[r * dim + c] * in[c];
        out[r] = acc;
    }
}

/* Set (if requested > 0) and report the number of OpenMP threads the loops above use: bench.py states it as
 * cpu_baseline.cores.  (Setting OMP_NUM_THREADS from Python is too late once an OpenMP runtime is loaded.) */
int oracle_threads(int requested) {
    if (requested > 0) omp_set_num_threads(requested);
    return omp_get_max_threads();
}

double oracle_norm2(const double *state, int n) {
    const uint64_t len = 2ull << n;
    double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static)
    for (uint64_t i = 0; i < len; ++i) s += state[i] * state[i];
    return s;
}
