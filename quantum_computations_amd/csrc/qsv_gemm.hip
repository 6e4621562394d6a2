// Large-grid single-axis contraction through rocBLAS.
//
// The reference's CV simulator works on position grids of d = 1000 points with bond dimensions up to 100
// (SURVEY.md 8a row a9): applying a d x d operator to one axis of an (L, d, R) site tensor is then a plain
// complex GEMM per l -- out[l] (d_out x R) = M (d_out x d_in) . in[l] (d_in x R) -- at 8 L d^2 R flops, far
// above what the LDS-tiled vector kernel of qsv_qudit.hip sustains (8 TFLOP/s against ~60 from the f64 MFMA
// pipeline).  Plain library GEMMs are what rocBLAS is for, so this file routes that case to
// rocblas_zgemm_strided_batched.  rocBLAS is bound with dlopen on first use: libqsv.so keeps loading (and the
// qubit path keeps working) on a machine without it, and inside a PyTorch process the copy PyTorch already
// loaded is reused instead of a second one.  When it cannot be loaded the caller falls back to its own HIP
// kernels; `QSV_NO_ROCBLAS=1` forces that for comparisons.
#include <dlfcn.h>

#include <cstdlib>
#include <mutex>
#include <vector>

#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>

#include "qsv_internal.h"

namespace {

struct RocblasApi {
    decltype(&rocblas_create_handle) create = nullptr;
    decltype(&rocblas_set_stream) set_stream = nullptr;
    decltype(&rocblas_zgemm_strided_batched) zgemm = nullptr;
    decltype(&rocsolver_zgesvd) zgesvd = nullptr;      // null when rocSOLVER is absent: SVD entry points fail loudly
    decltype(&rocsolver_zgeqrf) zgeqrf = nullptr;
    decltype(&rocsolver_zungqr) zungqr = nullptr;
    rocblas_handle handle[16] = {};
    bool tried = false, ok = false;
    std::mutex lock;
};

RocblasApi &api() {
    static RocblasApi a;
    return a;
}

bool load_locked(RocblasApi &a) {
    if (a.tried) return a.ok;
    a.tried = true;
    const char *off = std::getenv("QSV_NO_ROCBLAS");
    if (off && off[0] == '1') return false;
    void *lib = nullptr;
    for (const char *name : {"librocblas.so.5", "librocblas.so", "/opt/rocm/lib/librocblas.so.5",
                             "/opt/rocm/lib/librocblas.so"}) {
        lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (lib) break;
    }
    if (!lib) return false;
    a.create = reinterpret_cast<decltype(a.create)>(dlsym(lib, "rocblas_create_handle"));
    a.set_stream = reinterpret_cast<decltype(a.set_stream)>(dlsym(lib, "rocblas_set_stream"));
    a.zgemm = reinterpret_cast<decltype(a.zgemm)>(dlsym(lib, "rocblas_zgemm_strided_batched"));
    a.ok = a.create && a.set_stream && a.zgemm;
    if (a.ok) {
        void *solver = nullptr;
        for (const char *name : {"librocsolver.so.0", "librocsolver.so", "/opt/rocm/lib/librocsolver.so.0",
                                 "/opt/rocm/lib/librocsolver.so"}) {
            solver = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (solver) break;
        }
        if (solver) {
            a.zgesvd = reinterpret_cast<decltype(a.zgesvd)>(dlsym(solver, "rocsolver_zgesvd"));
            a.zgeqrf = reinterpret_cast<decltype(a.zgeqrf)>(dlsym(solver, "rocsolver_zgeqrf"));
            a.zungqr = reinterpret_cast<decltype(a.zungqr)>(dlsym(solver, "rocsolver_zungqr"));
        }
    }
    return a.ok;
}

// Handle of `device` bound to `stream`, or null (and *rc set) when the libraries cannot be used.
rocblas_handle handle_for(RocblasApi &a, int device, hipStream_t stream, int *rc) {
    *rc = QSV_OK;
    if (device < 0 || device >= 16) {
        *rc = qsv_fail(QSV_EINVAL, "device index out of range");
        return nullptr;
    }
    if (!load_locked(a)) {
        *rc = qsv_fail(QSV_EHIP, "rocBLAS could not be loaded (librocblas.so.5): tensor-network entry points need it");
        return nullptr;
    }
    if (!a.handle[device] && a.create(&a.handle[device]) != rocblas_status_success) {
        a.handle[device] = nullptr;
        *rc = qsv_fail(QSV_EHIP, "rocblas_create_handle failed");
        return nullptr;
    }
    if (a.set_stream(a.handle[device], stream) != rocblas_status_success) {
        *rc = qsv_fail(QSV_EHIP, "rocblas_set_stream failed");
        return nullptr;
    }
    return a.handle[device];
}

rocblas_operation op_of(int op) {
    return op == 0 ? rocblas_operation_none : op == 1 ? rocblas_operation_transpose
                                                      : rocblas_operation_conjugate_transpose;
}

// m1[row, i] = sqrt(s_i) * vt[row * k + i]  (i < r): compacts the (rows x k) factor to (rows x r)
__global__ __launch_bounds__(QSV_BLOCK) void k_scale_columns(const amp_t *__restrict__ vt, amp_t *__restrict__ m1,
                                                            uint64_t rows, uint64_t k, uint64_t r,
                                                            const double *__restrict__ s) {
    const uint64_t total = rows * r;
    for (uint64_t o = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x; o < total;
         o += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        const uint64_t row = o / r, i = o % r;
        const double w = sqrt(s[i]);
        const amp_t v = vt[row * k + i];
        m1[o] = amp_t{w * v.x, w * v.y};
    }
}

// m2[i, c] = sqrt(s_i) * u[i * cols + c]  (i < r)
__global__ __launch_bounds__(QSV_BLOCK) void k_scale_rows(const amp_t *__restrict__ u, amp_t *__restrict__ m2,
                                                         uint64_t cols, uint64_t r, const double *__restrict__ s) {
    const uint64_t total = r * cols;
    for (uint64_t o = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x; o < total;
         o += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        const double w = sqrt(s[o / cols]);
        const amp_t v = u[o];
        m2[o] = amp_t{w * v.x, w * v.y};
    }
}

int blocks_for(uint64_t items) {
    const uint64_t b = (items + QSV_BLOCK - 1) / QSV_BLOCK;
    return static_cast<int>(b < 1 ? 1 : (b > 65536 ? 65536 : b));
}

struct DeviceBuffers {   // frees whatever was allocated when it goes out of scope
    void *p[12] = {};
    int n = 0;
    template <class T>
    bool alloc(T **out, size_t bytes) {
        if (hipMalloc(reinterpret_cast<void **>(out), bytes ? bytes : 16) != hipSuccess) return false;
        p[n++] = *out;
        return true;
    }
    ~DeviceBuffers() {
        for (int i = 0; i < n; ++i) (void)hipFree(p[i]);
    }
};

// out (column-major n x m, ld n) = in (row-major n x m): LDS-tiled transpose of the element order
__global__ __launch_bounds__(256) void k_to_column_major(const amp_t *__restrict__ in, amp_t *__restrict__ out,
                                                        uint64_t n, uint64_t m) {
    __shared__ amp_t tile[16][17];
    const uint64_t tiles_m = (m + 15) / 16, tiles = tiles_m * ((n + 15) / 16);
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    for (uint64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const uint64_t r0 = (t / tiles_m) * 16, c0 = (t % tiles_m) * 16;
        __syncthreads();
        if (r0 + ty < n && c0 + tx < m) tile[ty][tx] = in[(r0 + ty) * m + c0 + tx];
        __syncthreads();
        if (c0 + ty < m && r0 + tx < n) out[(c0 + ty) * n + r0 + tx] = tile[tx][ty];
    }
}

// out[a, b] (row-major A x B) = sqrt(s[by_row ? a : b]) * in[a * sa + b * sb]
__global__ __launch_bounds__(QSV_BLOCK) void k_scale_strided(const amp_t *__restrict__ in, amp_t *__restrict__ out,
                                                            uint64_t A, uint64_t B, uint64_t sa, uint64_t sb,
                                                            const double *__restrict__ s, int by_row) {
    const uint64_t total = A * B;
    for (uint64_t o = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x; o < total;
         o += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        const uint64_t a = o / B, b = o % B;
        const double w = sqrt(s[by_row ? a : b]);
        const amp_t v = in[a * sa + b * sb];
        out[o] = amp_t{w * v.x, w * v.y};
    }
}

// The reference's truncation rule (mps.py:83-86) on singular values sorted in decreasing order.
uint64_t kept_rank(const std::vector<double> &sv, int64_t max_bond_dim, double abs_err, double rel_err) {
    double total = 0.0;
    for (double v : sv) total += v;
    double allowed = total * rel_err;
    if (abs_err > allowed) allowed = abs_err;
    if (allowed < 0.0) allowed = 0.0;
    uint64_t r = 0;
    double tail = 0.0;
    for (size_t i = sv.size(); i-- > 0;) {
        tail += sv[i];
        if (tail > allowed) ++r;
    }
    if (max_bond_dim >= 0 && r > static_cast<uint64_t>(max_bond_dim)) r = static_cast<uint64_t>(max_bond_dim);
    if (r > sv.size()) r = sv.size();
    return r;
}

}  // namespace

// C (m x n) = op(A) . op(B) on row-major complex128 device buffers with tight leading dimensions.
// Row-major C = op(A) op(B) is the column-major product C^T = op(B)^T op(A)^T, and a row-major buffer read as
// column-major IS the transpose, so the operands swap places and keep their op flags.
int qsvg_gemm(int device, hipStream_t stream, int op_a, int op_b, uint64_t m, uint64_t n, uint64_t k,
              const amp_t *a_ptr, const amp_t *b_ptr, amp_t *c_ptr) {
    const uint64_t lim = 0x7fffffffull;
    if (m > lim || n > lim || k > lim) return qsv_fail(QSV_EINVAL, "matrix dimension exceeds 2^31 - 1");
    RocblasApi &a = api();
    std::lock_guard<std::mutex> guard(a.lock);
    int rc;
    rocblas_handle h = handle_for(a, device, stream, &rc);
    if (!h) return rc;
    const rocblas_double_complex one{1.0, 0.0}, zero{0.0, 0.0};
    const rocblas_int lda = static_cast<rocblas_int>(op_a == 0 ? k : m);   // columns of the row-major A buffer
    const rocblas_int ldb = static_cast<rocblas_int>(op_b == 0 ? n : k);
    const rocblas_status s =
        a.zgemm(h, op_of(op_b), op_of(op_a), static_cast<rocblas_int>(n), static_cast<rocblas_int>(m),
                static_cast<rocblas_int>(k), &one, reinterpret_cast<const rocblas_double_complex *>(b_ptr), ldb, 0,
                reinterpret_cast<const rocblas_double_complex *>(a_ptr), lda, 0, &zero,
                reinterpret_cast<rocblas_double_complex *>(c_ptr), static_cast<rocblas_int>(n), 0, 1);
    if (s != rocblas_status_success) return qsv_fail(QSV_EHIP, "rocblas_zgemm failed");
    return QSV_OK;
}

// tensor_svd (cv_simulator/mps.py:52-97) of a row-major (rows x cols) device matrix:
//   theta = U S Vh,  r from the truncation rule,  m1 = U[:, :r] sqrt(S[:r]),  m2 = sqrt(S[:r]) Vh[:r, :].
// rocSOLVER is column-major, so it factors theta^T = U' S V'^H (cols x rows); then U = (V'^H)^T and Vh = U'^T, i.e.
// the column-major U' buffer is the row-major Vh and the column-major V'^H buffer is the row-major U: no transposes.
int qsvg_svd_split(int device, hipStream_t stream, amp_t *theta, uint64_t rows, uint64_t cols, int64_t max_bond_dim,
                   double abs_err, double rel_err, amp_t *m1, amp_t *m2, uint64_t capacity, uint64_t *rank_out,
                   double *s_host) {
    const uint64_t lim = 0x7fffffffull;
    if (rows > lim || cols > lim) return qsv_fail(QSV_EINVAL, "matrix dimension exceeds 2^31 - 1");
    RocblasApi &a = api();
    std::lock_guard<std::mutex> guard(a.lock);
    int rc;
    rocblas_handle h = handle_for(a, device, stream, &rc);
    if (!h) return rc;
    if (!a.zgesvd) return qsv_fail(QSV_EHIP, "rocSOLVER could not be loaded (librocsolver.so.0): no SVD available");
    const uint64_t k = rows < cols ? rows : cols;
    DeviceBuffers buf;
    double *dS = nullptr, *dE = nullptr;
    amp_t *dU = nullptr, *dV = nullptr;
    rocblas_int *dinfo = nullptr;
    if (!buf.alloc(&dS, sizeof(double) * k) || !buf.alloc(&dE, sizeof(double) * k) ||
        !buf.alloc(&dU, sizeof(amp_t) * cols * k) || !buf.alloc(&dV, sizeof(amp_t) * k * rows) ||
        !buf.alloc(&dinfo, sizeof(rocblas_int)))
        return qsv_fail(QSV_ENOMEM, "device allocation of the SVD factors failed");
    const rocblas_status s = a.zgesvd(
        h, rocblas_svect_singular, rocblas_svect_singular, static_cast<rocblas_int>(cols),
        static_cast<rocblas_int>(rows), reinterpret_cast<rocblas_double_complex *>(theta),
        static_cast<rocblas_int>(cols), dS, reinterpret_cast<rocblas_double_complex *>(dU),
        static_cast<rocblas_int>(cols), reinterpret_cast<rocblas_double_complex *>(dV), static_cast<rocblas_int>(k), dE,
        rocblas_outofplace, dinfo);
    if (s != rocblas_status_success) return qsv_fail(QSV_EHIP, "rocsolver_zgesvd failed");
    std::vector<double> sv(k);
    rocblas_int info = 0;
    QSV_HIP(hipMemcpyAsync(sv.data(), dS, sizeof(double) * k, hipMemcpyDeviceToHost, stream));
    QSV_HIP(hipMemcpyAsync(&info, dinfo, sizeof(info), hipMemcpyDeviceToHost, stream));
    QSV_HIP(hipStreamSynchronize(stream));
    if (info != 0) return qsv_fail(QSV_EHIP, "rocsolver_zgesvd did not converge");
    const uint64_t r = kept_rank(sv, max_bond_dim, abs_err, rel_err);
    if (r > capacity) return qsv_fail(QSV_EINVAL, "output buffers are smaller than the kept bond dimension");
    if (r > 0) {
        hipLaunchKernelGGL(k_scale_columns, dim3(blocks_for(rows * r)), dim3(QSV_BLOCK), 0, stream, dV, m1, rows, k, r,
                           dS);
        hipLaunchKernelGGL(k_scale_rows, dim3(blocks_for(r * cols)), dim3(QSV_BLOCK), 0, stream, dU, m2, cols, r, dS);
        QSV_HIP(hipGetLastError());
        QSV_HIP(hipStreamSynchronize(stream));   // the factors are freed on return
    }
    if (s_host)
        for (uint64_t i = 0; i < k; ++i) s_host[i] = sv[i];
    *rank_out = r;
    return QSV_OK;
}

// tensor_svd on its randomized branch (mps.py:5-50,78-79; Halko, Martinsson & Tropp 2010): range finder with
// l = k + 10 Gaussian probes (drawn by the caller so that the reference's random stream is reproduced) and q power
// iterations re-orthonormalised by Householder QR, SVD of the small l x m' projection, first k triplets kept, then the
// same truncation rule.  Everything is column-major here; A is the tall orientation of theta (the reference
// transposes a wide matrix first) and `omega` is (m' x l) column-major, m' = min(rows, cols).
int qsvg_rsvd_split(int device, hipStream_t stream, const amp_t *theta, uint64_t rows, uint64_t cols, int64_t k_keep,
                    int l, int q, const amp_t *omega, double abs_err, double rel_err, amp_t *m1, amp_t *m2,
                    uint64_t capacity, uint64_t *rank_out, double *s_host) {
    const uint64_t lim = 0x7fffffffull;
    if (rows > lim || cols > lim) return qsv_fail(QSV_EINVAL, "matrix dimension exceeds 2^31 - 1");
    RocblasApi &a = api();
    std::lock_guard<std::mutex> guard(a.lock);
    int rc;
    rocblas_handle h = handle_for(a, device, stream, &rc);
    if (!h) return rc;
    if (!a.zgesvd || !a.zgeqrf || !a.zungqr)
        return qsv_fail(QSV_EHIP, "rocSOLVER could not be loaded (librocsolver.so.0): no SVD available");
    const bool wide = rows < cols;                      // the reference works on theta^T then
    const uint64_t n = wide ? cols : rows, m = wide ? rows : cols;   // A is n x m, n >= m
    const uint64_t L = static_cast<uint64_t>(l), kk = L < m ? L : m;
    if (k_keep < 1 || L < static_cast<uint64_t>(k_keep) || L > m)
        return qsv_fail(QSV_EINVAL, "need 1 <= k <= l <= min(rows, cols)");
    DeviceBuffers buf;
    amp_t *A = nullptr, *Qn = nullptr, *Qm = nullptr, *tau = nullptr, *B = nullptr, *UB = nullptr, *VB = nullptr;
    double *dS = nullptr, *dE = nullptr;
    rocblas_int *dinfo = nullptr;
    if (!buf.alloc(&Qn, sizeof(amp_t) * n * L) || !buf.alloc(&Qm, sizeof(amp_t) * m * L) ||
        !buf.alloc(&tau, sizeof(amp_t) * L) || !buf.alloc(&B, sizeof(amp_t) * L * m) ||
        !buf.alloc(&UB, sizeof(amp_t) * L * kk) || !buf.alloc(&VB, sizeof(amp_t) * kk * m) ||
        !buf.alloc(&dS, sizeof(double) * (2 * kk + 2)))
        return qsv_fail(QSV_ENOMEM, "device allocation of the randomized-SVD workspace failed");
    dE = dS + kk;
    dinfo = reinterpret_cast<rocblas_int *>(dS + 2 * kk);
    if (wide) {
        // theta row-major (rows x cols) read column-major is theta^T (cols x rows) = A already
        A = const_cast<amp_t *>(theta);
    } else {
        if (!buf.alloc(&A, sizeof(amp_t) * n * m))
            return qsv_fail(QSV_ENOMEM, "device allocation of the randomized-SVD workspace failed");
        const uint64_t tiles = ((n + 15) / 16) * ((m + 15) / 16);
        hipLaunchKernelGGL(k_to_column_major, dim3(static_cast<unsigned>(tiles < 65536 ? tiles : 65536)), dim3(256), 0,
                           stream, theta, A, n, m);
        QSV_HIP(hipGetLastError());
    }
    const rocblas_double_complex one{1.0, 0.0}, zero{0.0, 0.0};
    auto Z = [](const amp_t *p) { return reinterpret_cast<const rocblas_double_complex *>(p); };
    auto W = [](amp_t *p) { return reinterpret_cast<rocblas_double_complex *>(p); };
    const rocblas_int ni = static_cast<rocblas_int>(n), mi = static_cast<rocblas_int>(m), li = static_cast<rocblas_int>(L);
    auto gemm = [&](rocblas_operation ta, rocblas_operation tb, rocblas_int M_, rocblas_int N_, rocblas_int K_,
                    const amp_t *pa, rocblas_int lda, const amp_t *pb, rocblas_int ldb, amp_t *pc, rocblas_int ldc) {
        return a.zgemm(h, ta, tb, M_, N_, K_, &one, Z(pa), lda, 0, Z(pb), ldb, 0, &zero, W(pc), ldc, 0, 1) ==
               rocblas_status_success;
    };
    auto orthonormalise = [&](amp_t *Y, rocblas_int rows_) {      // Y <- Q of its reduced QR
        return a.zgeqrf(h, rows_, li, W(Y), rows_, W(tau)) == rocblas_status_success &&
               a.zungqr(h, rows_, li, li, W(Y), rows_, W(tau)) == rocblas_status_success;
    };
    const rocblas_operation N = rocblas_operation_none, Cc = rocblas_operation_conjugate_transpose;
    bool ok = gemm(N, N, ni, li, mi, A, ni, omega, mi, Qn, ni) && orthonormalise(Qn, ni);       // Y = A O
    for (int it = 0; ok && it < q; ++it) {
        ok = gemm(Cc, N, mi, li, ni, A, ni, Qn, ni, Qm, mi) && orthonormalise(Qm, mi) &&         // Y = A^H Q
             gemm(N, N, ni, li, mi, A, ni, Qm, mi, Qn, ni) && orthonormalise(Qn, ni);            // Y = A Q
    }
    ok = ok && gemm(Cc, N, li, mi, ni, Qn, ni, A, ni, B, li);                                    // B = Q^H A  (l x m)
    if (!ok) return qsv_fail(QSV_EHIP, "rocBLAS / rocSOLVER call failed in the randomized range finder");
    if (a.zgesvd(h, rocblas_svect_singular, rocblas_svect_singular, li, mi, W(B), li, dS, W(UB), li, W(VB),
                 static_cast<rocblas_int>(kk), dE, rocblas_outofplace, dinfo) != rocblas_status_success)
        return qsv_fail(QSV_EHIP, "rocsolver_zgesvd failed");
    const uint64_t k = static_cast<uint64_t>(k_keep);
    std::vector<double> sv(k);
    rocblas_int info = 0;
    QSV_HIP(hipMemcpyAsync(sv.data(), dS, sizeof(double) * k, hipMemcpyDeviceToHost, stream));
    QSV_HIP(hipMemcpyAsync(&info, dinfo, sizeof(info), hipMemcpyDeviceToHost, stream));
    QSV_HIP(hipStreamSynchronize(stream));
    if (info != 0) return qsv_fail(QSV_EHIP, "rocsolver_zgesvd did not converge");
    const uint64_t r = kept_rank(sv, k_keep, abs_err, rel_err);
    if (r > capacity) return qsv_fail(QSV_EINVAL, "output buffers are smaller than the kept bond dimension");
    if (r > 0) {
        // U_A = Q U_B[:, :r]  (n x r, column-major)
        amp_t *UA = nullptr;
        if (!buf.alloc(&UA, sizeof(amp_t) * n * r))
            return qsv_fail(QSV_ENOMEM, "device allocation of the randomized-SVD workspace failed");
        if (!gemm(N, N, ni, static_cast<rocblas_int>(r), li, Qn, ni, UB, li, UA, ni))
            return qsv_fail(QSV_EHIP, "rocblas_zgemm failed");
        // A = U_A S Vh_B.  Tall theta: theta = A; wide theta: theta = A^T = Vh_B^T S U_A^T.
        const amp_t *u_src = wide ? VB : UA, *v_src = wide ? UA : VB;
        // m1[row, i] = sqrt(s_i) u[row, i];  m2[i, c] = sqrt(s_i) vh[i, c]
        const uint64_t u_sa = wide ? kk : 1, u_sb = wide ? 1 : n;          // wide: u[row, i] = VB[i + row * kk]
        const uint64_t v_sa = wide ? n : 1, v_sb = wide ? 1 : kk;          // wide: vh[i, c] = UA[c + i * n]
        hipLaunchKernelGGL(k_scale_strided, dim3(blocks_for(rows * r)), dim3(QSV_BLOCK), 0, stream, u_src, m1, rows, r,
                           u_sa, u_sb, dS, 0);
        hipLaunchKernelGGL(k_scale_strided, dim3(blocks_for(r * cols)), dim3(QSV_BLOCK), 0, stream, v_src, m2, r, cols,
                           v_sa, v_sb, dS, 1);
        QSV_HIP(hipGetLastError());
    }
    QSV_HIP(hipStreamSynchronize(stream));   // the workspace is freed on return
    if (s_host)
        for (uint64_t i = 0; i < k; ++i) s_host[i] = sv[i];
    *rank_out = r;
    return QSV_OK;
}

// 1 = done by rocBLAS, 0 = not available / shape not worth it (caller uses its own kernels), < 0 = error.
int qsvg_axis_gemm(int device, hipStream_t stream, const amp_t *in, amp_t *out, uint64_t L, uint64_t d_in,
                   uint64_t d_out, uint64_t R, const double *dev_m) {
    const uint64_t lim = 0x7fffffffull;
    if (device < 0 || device >= 16 || d_in > lim || d_out > lim || R > lim || L > lim) return 0;
    RocblasApi &a = api();
    std::lock_guard<std::mutex> guard(a.lock);
    if (!load_locked(a)) return 0;
    if (!a.handle[device]) {
        if (a.create(&a.handle[device]) != rocblas_status_success) {
            a.handle[device] = nullptr;
            a.ok = false;
            return 0;
        }
    }
    rocblas_handle h = a.handle[device];
    if (a.set_stream(h, stream) != rocblas_status_success) return qsv_fail(QSV_EHIP, "rocblas_set_stream failed");
    const rocblas_double_complex one{1.0, 0.0}, zero{0.0, 0.0};
    const auto *A = reinterpret_cast<const rocblas_double_complex *>(in);
    const auto *M = reinterpret_cast<const rocblas_double_complex *>(dev_m);
    auto *C = reinterpret_cast<rocblas_double_complex *>(out);
    rocblas_status s;
    if (R == 1) {
        // out (L x d_out, row-major) = in (L x d_in) . M^T; in column-major terms out^T = M . in^T
        s = a.zgemm(h, rocblas_operation_transpose, rocblas_operation_none, static_cast<rocblas_int>(d_out),
                    static_cast<rocblas_int>(L), static_cast<rocblas_int>(d_in), &one, M,
                    static_cast<rocblas_int>(d_in), 0, A, static_cast<rocblas_int>(d_in), 0, &zero, C,
                    static_cast<rocblas_int>(d_out), 0, 1);
    } else {
        // per l: out_l^T (R x d_out, column-major) = in_l^T (R x d_in) . M^T (d_in x d_out); the row-major
        // buffers are exactly those column-major matrices
        s = a.zgemm(h, rocblas_operation_none, rocblas_operation_none, static_cast<rocblas_int>(R),
                    static_cast<rocblas_int>(d_out), static_cast<rocblas_int>(d_in), &one, A,
                    static_cast<rocblas_int>(R), static_cast<rocblas_stride>(d_in * R), M,
                    static_cast<rocblas_int>(d_in), 0, &zero, C, static_cast<rocblas_int>(R),
                    static_cast<rocblas_stride>(d_out * R), static_cast<rocblas_int>(L));
    }
    if (s != rocblas_status_success) return qsv_fail(QSV_EHIP, "rocblas_zgemm_strided_batched failed");
    return 1;
}
