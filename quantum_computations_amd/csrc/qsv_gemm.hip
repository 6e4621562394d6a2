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

#include <rocblas/rocblas.h>

#include "qsv_internal.h"

namespace {

struct RocblasApi {
    decltype(&rocblas_create_handle) create = nullptr;
    decltype(&rocblas_set_stream) set_stream = nullptr;
    decltype(&rocblas_zgemm_strided_batched) zgemm = nullptr;
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
    return a.ok;
}

}  // namespace

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
