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

#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <string>
#include <vector>

#include "qsv_linalg.h"

namespace qsvl {

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
            a.zgesdd = reinterpret_cast<decltype(a.zgesdd)>(dlsym(solver, "rocsolver_zgesdd"));
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

Pool &pool_of(int device) {
    static Pool pools[16];
    return pools[device];
}

}  // namespace qsvl

using namespace qsvl;

// C (m x n) = op(A) . op(B) on row-major complex128 device buffers with tight leading dimensions.
// Row-major C = op(A) op(B) is the column-major product C^T = op(B)^T op(A)^T, and a row-major buffer read as
// column-major IS the transpose, so the operands swap places and keep their op flags.
// A handful of output entries over a long inner dimension -- the environment matrices of the MPS read-out
// (site_register.py: (chi x chi) = X^H T with chi = 1..2 and an inner dimension of chi d = 1000..4000): rocBLAS runs such a
// product as ONE 64 x 64 macro tile walking the whole inner dimension (170-210 us each, 0.15 s of the GKP Grover run).
// Here every output entry is a workgroup-wide reduction; the sum over p is taken in a fixed order.
__global__ __launch_bounds__(256) void k_gemm_few_outputs(const amp_t *__restrict__ A, const amp_t *__restrict__ B,
                                                         amp_t *__restrict__ C, int op_a, int op_b, uint64_t m, uint64_t n,
                                                         uint64_t k) {
    __shared__ double red[4][2];
    const uint64_t i = blockIdx.x / n, j = blockIdx.x % n;
    double re = 0.0, im = 0.0;
    for (uint64_t p = threadIdx.x; p < k; p += 256) {
        amp_t x = op_a == 0 ? A[i * k + p] : A[p * m + i];
        amp_t y = op_b == 0 ? B[p * n + j] : B[j * k + p];
        if (op_a == 2) x.y = -x.y;
        if (op_b == 2) y.y = -y.y;
        re += x.x * y.x - x.y * y.y;
        im += x.x * y.y + x.y * y.x;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        re += __shfl_xor(re, o, 64);
        im += __shfl_xor(im, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        red[threadIdx.x >> 6][0] = re;
        red[threadIdx.x >> 6][1] = im;
    }
    __syncthreads();
    if (threadIdx.x == 0)
        C[blockIdx.x] = amp_t{(red[0][0] + red[1][0]) + (red[2][0] + red[3][0]), (red[0][1] + red[1][1]) + (red[2][1] + red[3][1])};
}

int qsvg_gemm(int device, hipStream_t stream, int op_a, int op_b, uint64_t m, uint64_t n, uint64_t k,
              const amp_t *a_ptr, const amp_t *b_ptr, amp_t *c_ptr) {
    const uint64_t lim = 0x7fffffffull;
    if (m > lim || n > lim || k > lim) return qsv_fail(QSV_EINVAL, "matrix dimension exceeds 2^31 - 1");
    if (m * n <= 64 && k >= 256 && m > 0 && n > 0) {
        QSV_HIP(hipSetDevice(device));
        hipLaunchKernelGGL(k_gemm_few_outputs, dim3(static_cast<unsigned>(m * n)), dim3(256), 0, stream, a_ptr, b_ptr, c_ptr,
                           op_a, op_b, m, n, k);
        QSV_HIP(hipGetLastError());
        return QSV_OK;
    }
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

// Give the decomposition scratch pool of `device` back to the driver (it is re-grown on demand).
int qsvg_release_workspace(int device) {
    if (device < 0 || device >= 16) return qsv_fail(QSV_EINVAL, "device index out of range");
    RocblasApi &a = api();
    std::lock_guard<std::mutex> guard(a.lock);
    Pool &pool = pool_of(device);
    if (pool.base) {
        QSV_HIP(hipSetDevice(device));
        QSV_HIP(hipDeviceSynchronize());
        QSV_HIP(hipFree(pool.base));
        pool.base = nullptr;
        pool.capacity = 0;
    }
    return QSV_OK;
}
