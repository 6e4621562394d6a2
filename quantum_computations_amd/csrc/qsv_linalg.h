// Internal: the rocBLAS / rocSOLVER binding and the scratch pool shared by qsv_gemm.hip (plain GEMMs) and qsv_decomp.hip
// (splits, panel kernels).  Both libraries are bound with dlopen on first use: libqsv.so keeps loading -- and the qubit path
// keeps working -- on a machine without them, and inside a PyTorch process the copies PyTorch already loaded are reused.
#pragma once

#include <mutex>

#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>

#include "qsv_internal.h"

namespace qsvl {

struct RocblasApi {
    decltype(&rocblas_create_handle) create = nullptr;
    decltype(&rocblas_set_stream) set_stream = nullptr;
    decltype(&rocblas_zgemm_strided_batched) zgemm = nullptr;
    decltype(&rocsolver_zgesvd) zgesvd = nullptr;      // null when rocSOLVER is absent: SVD entry points fail loudly
    decltype(&rocsolver_zgesdd) zgesdd = nullptr;
    decltype(&rocsolver_zgeqrf) zgeqrf = nullptr;
    decltype(&rocsolver_zungqr) zungqr = nullptr;
    rocblas_handle handle[16] = {};
    bool tried = false, ok = false;
    std::mutex lock;      // serialises every call that uses the handles or the pool
};

RocblasApi &api();
bool load_locked(RocblasApi &a);
// Handle of `device` bound to `stream`, or null (and *rc set) when the libraries cannot be used.
rocblas_handle handle_for(RocblasApi &a, int device, hipStream_t stream, int *rc);
rocblas_operation op_of(int op);

// Scratch memory of the decompositions: one grow-only pool per device (a split needs a copy of theta plus panels --
// gigabytes -- and hipMalloc / hipFree of that size on every call costs milliseconds and synchronises the device).
// A DeviceBuffers object carves from the pool; requests the pool cannot hold fall back to hipMalloc and are freed when
// the object goes out of scope.  Calls are serialised by the library lock and end with a stream synchronisation, so the
// pool is never in use by two calls.
struct Pool {
    char *base = nullptr;
    size_t capacity = 0;
};

Pool &pool_of(int device);

struct DeviceBuffers {
    Pool *pool = nullptr;
    size_t used = 0;
    void *extra[12] = {};
    int n = 0;

    // Make the pool of `device` at least `bytes` large (no-op when it already is).  Call before the first alloc.
    void reserve(int device, size_t bytes) {
        pool = &pool_of(device);
        if (pool->capacity >= bytes) return;
        if (pool->base) {
            (void)hipDeviceSynchronize();
            (void)hipFree(pool->base);
            pool->base = nullptr;
            pool->capacity = 0;
        }
        const size_t want = bytes + bytes / 8;
        if (hipMalloc(reinterpret_cast<void **>(&pool->base), want) == hipSuccess) pool->capacity = want;
        else pool->base = nullptr;
    }

    template <class T>
    bool alloc(T **out, size_t bytes) {
        const size_t need = (bytes + 255) / 256 * 256;
        if (pool && pool->base && used + need <= pool->capacity) {
            *out = reinterpret_cast<T *>(pool->base + used);
            used += need;
            return true;
        }
        if (n >= 12 || hipMalloc(reinterpret_cast<void **>(out), bytes ? bytes : 16) != hipSuccess) return false;
        extra[n++] = *out;
        return true;
    }

    ~DeviceBuffers() {
        for (int i = 0; i < n; ++i) (void)hipFree(extra[i]);
    }
};

}  // namespace qsvl
