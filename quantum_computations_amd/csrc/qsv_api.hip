// C ABI of libqsv.so (include/qsv.h): argument validation, qubit -> bit mapping, kernel selection.
// The device work lives in qsv_kernels.hip (qubits) and qsv_qudit.hip (d-level modes).

#include "qsv_internal.h"

#include <sys/mman.h>

#include <thread>

#include <cmath>
#include <cstring>
#include <vector>

namespace {

thread_local std::string g_last_error;

bool valid(const qsv_state *st) { return st != nullptr; }

// reference convention: qubit q of n <-> bit n-1-q of the flat index
inline int bit_of(const qsv_state *st, int q) { return st->n - 1 - q; }

int check_qubits(const qsv_state *st, int k, const int *qs) {
    if (st->kind != 0) return qsv_fail(QSV_ESTATE, "this call needs a qubit register");
    for (int i = 0; i < k; ++i) {
        if (qs[i] < 0) return qsv_fail(QSV_EINVAL, "Non-negative index");
        if (qs[i] >= st->n)
            return qsv_fail(QSV_EINVAL, "qubit index " + std::to_string(qs[i]) + " out of range for a " +
                                            std::to_string(st->n) + "-qubit register");
        for (int j = 0; j < i; ++j)
            if (qs[i] == qs[j]) return qsv_fail(QSV_EINVAL, "Indices must be distinct.");
    }
    return QSV_OK;
}

int alloc_workspace(qsv_state *st) {
    QSV_HIP(hipMalloc(reinterpret_cast<void **>(&st->partials), sizeof(double) * 2 * QSV_REDUCE_BLOCKS));
    QSV_HIP(hipHostMalloc(reinterpret_cast<void **>(&st->partials_host), sizeof(double) * 2 * QSV_REDUCE_BLOCKS,
                          hipHostMallocDefault));
    QSV_HIP(hipEventCreate(&st->ev_start));
    QSV_HIP(hipEventCreate(&st->ev_stop));
    return QSV_OK;
}

int create_common(int kind, int n, int d, uint64_t amps, int device, void *dev_amps, uint64_t capacity,
                  void *stream, qsv_state **out) {
    if (!out) return qsv_fail(QSV_EINVAL, "null output handle");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return qsv_fail(QSV_EHIP, std::string("no HIP device available: ") +
                                      (e != hipSuccess ? hipGetErrorString(e) : "device count is 0"));
    if (device < 0 || device >= count) return qsv_fail(QSV_EINVAL, "device ordinal out of range");
    QSV_HIP(hipSetDevice(device));
    qsv_state *st = new qsv_state();
    st->device = device;
    st->kind = kind;
    st->n = n;
    st->d = d;
    st->amps = amps;
    st->stream = static_cast<hipStream_t>(stream);
    if (dev_amps) {
        if (capacity < amps) {
            delete st;
            return qsv_fail(QSV_ENOMEM, "view capacity is smaller than the register");
        }
        st->data = static_cast<amp_t *>(dev_amps);
        st->capacity = capacity;
        st->owns_data = false;
    } else {
        if (hipMalloc(reinterpret_cast<void **>(&st->data), sizeof(amp_t) * amps) != hipSuccess) {
            delete st;
            return qsv_fail(QSV_ENOMEM, "device allocation of " + std::to_string(amps * sizeof(amp_t)) +
                                            " bytes for the register failed");
        }
        st->capacity = amps;
        st->owns_data = true;
    }
    int rc = alloc_workspace(st);
    if (rc) {
        qsv_destroy(st);
        return rc;
    }
    if (st->owns_data) {
        rc = qsvk_set_basis(st, 0);
        if (rc) {
            qsv_destroy(st);
            return rc;
        }
    }
    *out = st;
    return QSV_OK;
}

bool is_one(double re, double im) { return re == 1.0 && im == 0.0; }
bool is_zero(double re, double im) { return re == 0.0 && im == 0.0; }

bool matrix_is_diagonal(int D, const double *m) {
    for (int r = 0; r < D; ++r)
        for (int c = 0; c < D; ++c)
            if (r != c && !is_zero(m[2 * (r * D + c)], m[2 * (r * D + c) + 1])) return false;
    return true;
}

bool matrix_equals(int D, const double *m, const double *ref_real) {
    for (int i = 0; i < D * D; ++i)
        if (m[2 * i] != ref_real[i] || m[2 * i + 1] != 0.0) return false;
    return true;
}

// A control only saves HBM traffic when skipping its control = 0 amplitudes skips whole 128-byte lines, i.e.
// when the control bit is >= 3 (8 amplitudes).  Below that a "controlled" kernel still moves every line and
// the predicated lanes cost time (sweep: 2.0 ms vs 1.35 ms at n = 28), so such controls are folded back into
// the matrix and the gate runs at full traffic.
constexpr int QSV_MIN_CTRL_BIT = 3;

// 1-qubit diagonal with the traffic-saving special cases: d0 == 1 touches only the bit = 1 half.
int diag_1q_bits(qsv_state *st, int bit, const double d[4]) {
    if (st->specialize && is_one(d[0], d[1])) {
        if (is_one(d[2], d[3])) return QSV_OK;  // identity
        if (bit >= QSV_MIN_CTRL_BIT) return qsvk_phase(st, 1, &bit, d[2], d[3]);
    }
    return qsvk_diag(st, 1, &bit, 0, nullptr, d);
}

int diag_2q_bits(qsv_state *st, int b0, int b1, const double d[8]) {
    if (st->specialize) {
        const bool one0 = is_one(d[0], d[1]), one1 = is_one(d[2], d[3]), one2 = is_one(d[4], d[5]);
        const bool c0 = b0 >= QSV_MIN_CTRL_BIT, c1 = b1 >= QSV_MIN_CTRL_BIT;
        if (one0 && one1 && one2) {  // controlled phase: CZ touches a quarter of the register
            if (is_one(d[6], d[7])) return QSV_OK;
            const double dd[4] = {1.0, 0.0, d[6], d[7]};
            if (c0 && c1) {
                const int both[2] = {b0, b1};
                return qsvk_phase(st, 2, both, d[6], d[7]);
            }
            if (c0) return qsvk_diag(st, 1, &b1, 1, &b0, dd);  // half traffic: control on the wide bit
            if (c1) return qsvk_diag(st, 1, &b0, 1, &b1, dd);
        } else if (one0 && one1 && c0) {
            return qsvk_diag(st, 1, &b1, 1, &b0, d + 4);  // control on leg 0
        } else if (one0 && one2 && d[2] == d[6] && d[3] == d[7]) {
            const double dd[4] = {1.0, 0.0, d[2], d[3]};  // acts on leg 1 only
            return diag_1q_bits(st, b1, dd);
        }
    }
    const int bits[2] = {b0, b1};
    return qsvk_diag(st, 2, bits, 0, nullptr, d);
}

// Controlled 2x2 gate; one narrow control (bit < 3) is folded into a 4x4 matrix (see QSV_MIN_CTRL_BIT).
int controlled_1q_bits(qsv_state *st, int nctrl, const int *cbits, int tbit, const double u[8]) {
    int fold = -1;
    for (int i = 0; i < nctrl && st->n >= QSV_LANE_BITS; ++i)
        if (cbits[i] < QSV_MIN_CTRL_BIT) fold = i;
    if (fold < 0) return qsvk_dense(st, 1, &tbit, nctrl, cbits, u);
    double m[32] = {0};
    m[0] = 1.0;            // |00><00|
    m[2 * (1 * 4 + 1)] = 1.0;  // |01><01|
    for (int r = 0; r < 2; ++r)
        for (int c = 0; c < 2; ++c) {
            m[2 * ((2 + r) * 4 + 2 + c)] = u[2 * (r * 2 + c)];
            m[2 * ((2 + r) * 4 + 2 + c) + 1] = u[2 * (r * 2 + c) + 1];
        }
    std::vector<int> rest;
    for (int i = 0; i < nctrl; ++i)
        if (i != fold) rest.push_back(cbits[i]);
    const int bits[2] = {cbits[fold], tbit};
    return qsvk_dense(st, 2, bits, static_cast<int>(rest.size()), rest.data(), m);
}

}  // namespace

int qsv_fail(int code, const std::string &msg) {
    g_last_error = msg;
    return code;
}

extern "C" {

int qsv_version(void) { return QSV_VERSION; }

const char *qsv_last_error(void) { return g_last_error.c_str(); }

int qsv_device_count(int *count) {
    if (!count) return qsv_fail(QSV_EINVAL, "null pointer");
    *count = 0;
    hipError_t e = hipGetDeviceCount(count);
    if (e != hipSuccess) {
        *count = 0;
        return qsv_fail(QSV_EHIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    }
    return QSV_OK;
}

int qsv_create(int n_qubits, int device, qsv_state **out) {
    if (n_qubits < 0 || n_qubits > 40) return qsv_fail(QSV_EINVAL, "n_qubits must be in 0..40");
    return create_common(0, n_qubits, 2, 1ull << n_qubits, device, nullptr, 0, nullptr, out);
}

int qsv_create_view(int n_qubits, int device, void *dev_amps, uint64_t capacity_amps, void *hip_stream,
                    qsv_state **out) {
    if (n_qubits < 0 || n_qubits > 40) return qsv_fail(QSV_EINVAL, "n_qubits must be in 0..40");
    if (!dev_amps) return qsv_fail(QSV_EINVAL, "null device pointer");
    if (reinterpret_cast<uintptr_t>(dev_amps) % 16) return qsv_fail(QSV_EINVAL, "device pointer must be 16-byte aligned");
    return create_common(0, n_qubits, 2, 1ull << n_qubits, device, dev_amps, capacity_amps, hip_stream, out);
}

int qsv_rebind_view(qsv_state *st, int n_qubits, void *dev_amps, uint64_t capacity_amps) {
    if (!valid(st)) return qsv_fail(QSV_EINVAL, "null state");
    if (st->kind != 0 || st->owns_data) return qsv_fail(QSV_ESTATE, "only a qubit view register can be re-pointed");
    if (n_qubits < 0 || n_qubits > 40) return qsv_fail(QSV_EINVAL, "n_qubits must be in 0..40");
    if (!dev_amps) return qsv_fail(QSV_EINVAL, "null device pointer");
    if (reinterpret_cast<uintptr_t>(dev_amps) % 16) return qsv_fail(QSV_EINVAL, "device pointer must be 16-byte aligned");
    if (capacity_amps < (1ull << n_qubits)) return qsv_fail(QSV_EINVAL, "capacity is smaller than the register");
    st->data = static_cast<amp_t *>(dev_amps);
    st->capacity = capacity_amps;
    st->n = n_qubits;
    st->amps = 1ull << n_qubits;
    return QSV_OK;
}

int qsv_destroy(qsv_state *st) {
    if (!st) return QSV_OK;
    (void)hipSetDevice(st->device);
    if (st->stream || st->data) (void)hipStreamSynchronize(st->stream);
    if (st->owns_data && st->data) (void)hipFree(st->data);
    if (st->spare) (void)hipFree(st->spare);
    if (st->partials) (void)hipFree(st->partials);
    if (st->partials_host) (void)hipHostFree(st->partials_host);
    if (st->dev_matrix) (void)hipFree(st->dev_matrix);
    if (st->stage_dev) (void)hipFree(st->stage_dev);
    if (st->stage_host) (void)hipHostFree(st->stage_host);
    for (hipEvent_t ev : st->stage_done)
        if (ev) (void)hipEventDestroy(ev);
    if (st->ev_start) (void)hipEventDestroy(st->ev_start);
    if (st->ev_stop) (void)hipEventDestroy(st->ev_stop);
    for (hipEvent_t ev : st->marks)
        if (ev) (void)hipEventDestroy(ev);
    delete st;
    return QSV_OK;
}

int qsv_set_stream(qsv_state *st, void *hip_stream) {
    if (!valid(st)) return qsv_fail(QSV_EINVAL, "null state");
    QSV_HIP(hipStreamSynchronize(st->stream));
    st->stream = static_cast<hipStream_t>(hip_stream);
    return QSV_OK;
}

int qsv_set_option(qsv_state *st, int option, int64_t value) {
    if (!valid(st)) return qsv_fail(QSV_EINVAL, "null state");
    switch (option) {
        case QSV_OPT_SPECIALIZE: st->specialize = value != 0; return QSV_OK;
        case QSV_OPT_UNROLL:
            if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8)
                return qsv_fail(QSV_EINVAL, "unroll must be 0, 1, 2, 4 or 8");
            st->unroll = static_cast<int>(value);
            return QSV_OK;
        case QSV_OPT_GRID_CAP:
            if (value < 0 || value > (1 << 30)) return qsv_fail(QSV_EINVAL, "bad grid cap");
            st->grid_cap = static_cast<int>(value);
            return QSV_OK;
        case QSV_OPT_NONTEMPORAL: st->nontemporal = value != 0; return QSV_OK;
        case QSV_OPT_TILE_REGIONS:
            if (value < -1 || value > (1 << 20)) return qsv_fail(QSV_EINVAL, "tile regions must be -1, 0 or a count");
            st->remap = static_cast<int>(value);
            return QSV_OK;
        case QSV_OPT_ITEM_STRIDE_BIT:
            if (value < 8 || value > 24) return qsv_fail(QSV_EINVAL, "item stride bit must be in 8..24");
            st->ubit = static_cast<int>(value);
            return QSV_OK;
        case QSV_OPT_PLANE_KERNEL: st->plane_kernel = value != 0; return QSV_OK;
        case QSV_OPT_READOUT_VARIANT:
            if (value < 0 || value > 2) return qsv_fail(QSV_EINVAL, "read-out variant must be 0, 1 or 2");
            st->readout_variant = static_cast<int>(value);
            return QSV_OK;
        case QSV_OPT_COMPLEX_PRODUCT:
            if (value != 0 && value != 3 && value != 4) return qsv_fail(QSV_EINVAL, "complex product must be 0, 3 or 4");
            st->complex_product = static_cast<int>(value);
            return QSV_OK;
        case QSV_OPT_TILE_SEQUENCE_GATES:
            if (value < -1 || value > 48) return qsv_fail(QSV_EINVAL, "tile sequence limit must be -1 .. 48");
            st->tile_sequence_gates = static_cast<int>(value);
            return QSV_OK;
        case QSV_OPT_SEQUENCE_WORK:
            if (value < -1 || value > (1 << 20)) return qsv_fail(QSV_EINVAL, "sequence work limit out of range");
            st->sequence_work = static_cast<int>(value);
            return QSV_OK;
        case QSV_OPT_KQ_VARIANT:
            if (value < 0 || value > 6) return qsv_fail(QSV_EINVAL, "k-qubit kernel variant must be 0 .. 6");
            st->kq_variant = static_cast<int>(value);
            return QSV_OK;
        default: return qsv_fail(QSV_EINVAL, "unknown option");
    }
}

int qsv_num_qubits(const qsv_state *st, int *n_qubits) {
    if (!valid(st) || !n_qubits) return qsv_fail(QSV_EINVAL, "null pointer");
    if (st->kind != 0) return qsv_fail(QSV_ESTATE, "this call needs a qubit register");
    *n_qubits = st->n;
    return QSV_OK;
}

int qsv_num_amps(const qsv_state *st, uint64_t *n_amps) {
    if (!valid(st) || !n_amps) return qsv_fail(QSV_EINVAL, "null pointer");
    *n_amps = st->amps;
    return QSV_OK;
}

int qsv_device_ptr(qsv_state *st, void **dev_amps) {
    if (!valid(st) || !dev_amps) return qsv_fail(QSV_EINVAL, "null pointer");
    *dev_amps = st->data;
    return QSV_OK;
}

int qsv_sync(qsv_state *st) {
    if (!valid(st)) return qsv_fail(QSV_EINVAL, "null state");
    QSV_HIP(hipSetDevice(st->device));
    QSV_HIP(hipStreamSynchronize(st->stream));
    return QSV_OK;
}

int qsv_set_basis(qsv_state *st, uint64_t index) {
    if (!valid(st)) return qsv_fail(QSV_EINVAL, "null state");
    if (index >= st->amps) return qsv_fail(QSV_EINVAL, "basis index out of range");
    QSV_HIP(hipSetDevice(st->device));
    return qsvk_set_basis(st, index);
}

int qsv_upload(qsv_state *st, const double *host, uint64_t offset, uint64_t count) {
    if (!valid(st) || (!host && count)) return qsv_fail(QSV_EINVAL, "null pointer");
    if (offset > st->amps || count > st->amps - offset) return qsv_fail(QSV_EINVAL, "upload range out of bounds");
    QSV_HIP(hipSetDevice(st->device));
    QSV_HIP(hipMemcpyAsync(st->data + offset, host, sizeof(amp_t) * count, hipMemcpyHostToDevice, st->stream));
    QSV_HIP(hipStreamSynchronize(st->stream));
    return QSV_OK;
}

// Make the pages of [p, p + bytes) present and writable, with up to `threads` threads.  A freshly allocated destination
// (np.empty of 4 GiB is an untouched mmap) otherwise takes its one million page faults inside the copy, one after the
// other: that, not PCIe, was the 16.8 GB/s of round 1's downloads.  Writing back the byte just read keeps the content.
// Nothing here may leave through the C ABI as an exception: std::thread throws std::system_error when the process is at
// its thread limit (the GPU runner enforces one), so every creation is guarded -- whatever threads did start are joined,
// the range they did not cover is touched by the calling thread.  No madvise: the mapping belongs to the caller.
static void prefault(char *p, size_t bytes, int threads) noexcept {
    const size_t page = 4096;
    char *first = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(p) + page - 1) / page * page);
    char *last = p + bytes;
    if (first >= last) return;
    const size_t pages = static_cast<size_t>(last - first + page - 1) / page;
    auto touch = [first](size_t from, size_t to) {
        constexpr size_t page = 4096;
        for (size_t i = from; i < to; ++i) {
            volatile char *c = first + i * page;
            *c = *c;
        }
    };
    std::thread pool[16];
    threads = std::max(1, std::min(threads, 16));
    int started = 0;
    for (int t = 1; t < threads; ++t) {                  // slice 0 is the caller's own
        try {
            pool[started] = std::thread(touch, pages * t / threads, pages * (t + 1) / threads);
            ++started;
        } catch (...) {
            touch(pages * t / threads, pages);               // no more threads: the rest is done here
            break;
        }
    }
    touch(0, pages / threads);
    for (int t = 0; t < started; ++t) pool[t].join();
}

int qsv_download(qsv_state *st, double *host, uint64_t offset, uint64_t count) {
    if (!valid(st) || (!host && count)) return qsv_fail(QSV_EINVAL, "null pointer");
    if (offset > st->amps || count > st->amps - offset) return qsv_fail(QSV_EINVAL, "download range out of bounds");
    QSV_HIP(hipSetDevice(st->device));
    const size_t bytes = sizeof(amp_t) * count, piece = 256ull << 20;
    if (bytes < piece) {
        QSV_HIP(hipMemcpyAsync(host, st->data + offset, bytes, hipMemcpyDeviceToHost, st->stream));
        QSV_HIP(hipStreamSynchronize(st->stream));
        return QSV_OK;
    }
    // large downloads: pieces of 256 MiB; while piece c crosses PCIe, eight threads fault in the pages of piece c + 1
    char *dst = reinterpret_cast<char *>(host);
    const char *src = reinterpret_cast<const char *>(st->data + offset);
    const int threads = 8;
    prefault(dst, std::min(piece, bytes), threads);
    for (size_t done = 0; done < bytes; done += piece) {
        const size_t len = std::min(piece, bytes - done);
        std::thread ahead;
        bool ahead_running = false;
        if (done + len < bytes) {
            try {
                ahead = std::thread(prefault, dst + done + len, std::min(piece, bytes - done - len), threads);
                ahead_running = true;
            } catch (...) {
                // at the thread limit: the copy below takes the faults itself (slower, still correct)
            }
        }
        const hipError_t e = hipMemcpyAsync(dst + done, src + done, len, hipMemcpyDeviceToHost, st->stream);
        const hipError_t e2 = e == hipSuccess ? hipStreamSynchronize(st->stream) : e;
        if (ahead_running) ahead.join();
        if (e2 != hipSuccess) return qsv_fail(QSV_EHIP, std::string("download: ") + hipGetErrorString(e2));
    }
    return QSV_OK;
}

int qsv_copy(qsv_state *dst, const qsv_state *src) {
    if (!valid(dst) || !valid(src)) return qsv_fail(QSV_EINVAL, "null state");
    if (dst->kind != src->kind || dst->d != src->d) return qsv_fail(QSV_ESTATE, "registers of different kinds");
    if (src->amps > dst->capacity) return qsv_fail(QSV_ENOMEM, "destination register too small");
    QSV_HIP(hipSetDevice(dst->device));
    QSV_HIP(hipStreamSynchronize(src->stream));
    if (dst->device == src->device) {
        snprintf(dst->last_kernel, sizeof(dst->last_kernel), "k_copy");
        const int rc = qsvk_copy(dst->data, src->data, src->amps, dst->stream);
        if (rc) return rc;
    } else {
        QSV_HIP(hipMemcpyAsync(dst->data, src->data, sizeof(amp_t) * src->amps, hipMemcpyDeviceToDevice, dst->stream));
    }
    dst->n = src->n;
    dst->amps = src->amps;
    return QSV_OK;
}

int qsv_fill_random(qsv_state *st, uint64_t seed, uint64_t index_offset, double *norm2) {
    if (!valid(st)) return qsv_fail(QSV_EINVAL, "null state");
    QSV_HIP(hipSetDevice(st->device));
    return qsvk_fill_random(st, seed, index_offset, norm2);
}

int qsv_scale(qsv_state *st, double re, double im) {
    if (!valid(st)) return qsv_fail(QSV_EINVAL, "null state");
    QSV_HIP(hipSetDevice(st->device));
    return qsvk_scale(st, re, im);
}

// ---- gates ----------------------------------------------------------------------------------------

int qsv_apply_1q(qsv_state *st, int q, const double m[8]) {
    if (!valid(st) || !m) return qsv_fail(QSV_EINVAL, "null pointer");
    int rc = check_qubits(st, 1, &q);
    if (rc) return rc;
    QSV_HIP(hipSetDevice(st->device));
    const int bit = bit_of(st, q);
    if (st->specialize && matrix_is_diagonal(2, m)) {
        const double d[4] = {m[0], m[1], m[6], m[7]};
        return diag_1q_bits(st, bit, d);
    }
    return qsvk_dense(st, 1, &bit, 0, nullptr, m);
}

int qsv_apply_2q(qsv_state *st, int q0, int q1, const double m[32]) {
    if (!valid(st) || !m) return qsv_fail(QSV_EINVAL, "null pointer");
    const int qs[2] = {q0, q1};
    int rc = check_qubits(st, 2, qs);
    if (rc) return rc;
    QSV_HIP(hipSetDevice(st->device));
    const int bits[2] = {bit_of(st, q0), bit_of(st, q1)};
    if (st->specialize) {
        if (matrix_is_diagonal(4, m)) {
            const double d[8] = {m[0], m[1], m[10], m[11], m[20], m[21], m[30], m[31]};
            return diag_2q_bits(st, bits[0], bits[1], d);
        }
        static const double CX[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 1, 0, 0, 1, 0};
        static const double XC[16] = {1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 1, 0, 0, 1, 0, 0};  // control = leg 1
        static const double SW[16] = {1, 0, 0, 0, 0, 0, 1, 0, 0, 1, 0, 0, 0, 0, 0, 1};
        if (matrix_equals(4, m, CX)) return qsv_apply_cx(st, q0, q1);
        if (matrix_equals(4, m, XC)) return qsv_apply_cx(st, q1, q0);
        if (matrix_equals(4, m, SW)) return qsv_apply_swap(st, q0, q1);
        // controlled-U with the control on leg 0: rows/cols 0,1 are the identity block
        bool ctl0 = true;
        for (int r = 0; r < 4 && ctl0; ++r)
            for (int c = 0; c < 4; ++c) {
                if (r >= 2 && c >= 2) continue;
                const double want = (r == c) ? 1.0 : 0.0;
                if (m[2 * (r * 4 + c)] != want || m[2 * (r * 4 + c) + 1] != 0.0) {
                    ctl0 = false;
                    break;
                }
            }
        if (ctl0) {
            const double u[8] = {m[20], m[21], m[22], m[23], m[28], m[29], m[30], m[31]};
            if (bits[0] >= QSV_MIN_CTRL_BIT) return qsvk_dense(st, 1, &bits[1], 1, &bits[0], u);
        }
    }
    return qsvk_dense(st, 2, bits, 0, nullptr, m);
}

int qsv_apply_diag_1q(qsv_state *st, int q, const double d[4]) {
    if (!valid(st) || !d) return qsv_fail(QSV_EINVAL, "null pointer");
    int rc = check_qubits(st, 1, &q);
    if (rc) return rc;
    QSV_HIP(hipSetDevice(st->device));
    return diag_1q_bits(st, bit_of(st, q), d);
}

int qsv_apply_diag_2q(qsv_state *st, int q0, int q1, const double d[8]) {
    if (!valid(st) || !d) return qsv_fail(QSV_EINVAL, "null pointer");
    const int qs[2] = {q0, q1};
    int rc = check_qubits(st, 2, qs);
    if (rc) return rc;
    QSV_HIP(hipSetDevice(st->device));
    return diag_2q_bits(st, bit_of(st, q0), bit_of(st, q1), d);
}

int qsv_apply_cx(qsv_state *st, int control, int target) {
    if (!valid(st)) return qsv_fail(QSV_EINVAL, "null state");
    const int qs[2] = {control, target};
    int rc = check_qubits(st, 2, qs);
    if (rc) return rc;
    QSV_HIP(hipSetDevice(st->device));
    const int cbit = bit_of(st, control), tbit = bit_of(st, target);
    static const double X[8] = {0, 0, 1, 0, 1, 0, 0, 0};
    return controlled_1q_bits(st, 1, &cbit, tbit, X);
}

int qsv_apply_swap(qsv_state *st, int q0, int q1) {
    if (!valid(st)) return qsv_fail(QSV_EINVAL, "null state");
    const int qs[2] = {q0, q1};
    int rc = check_qubits(st, 2, qs);
    if (rc) return rc;
    QSV_HIP(hipSetDevice(st->device));
    const int b0 = bit_of(st, q0), b1 = bit_of(st, q1);
    if (b0 >= QSV_LANE_BITS && b1 >= QSV_LANE_BITS && st->n >= QSV_LANE_BITS) return qsvk_pair_exchange(st, b0, b1);
    static const double SW[32] = {1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0,
                                  0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0};
    const int bits[2] = {b0, b1};
    return qsvk_dense(st, 2, bits, 0, nullptr, SW);
}

int qsv_apply_controlled_1q(qsv_state *st, int n_controls, const int *controls, int target, const double m[8]) {
    if (!valid(st) || !m || (n_controls > 0 && !controls)) return qsv_fail(QSV_EINVAL, "null pointer");
    if (n_controls < 0 || n_controls >= 64) return qsv_fail(QSV_EINVAL, "bad control count");
    std::vector<int> qs(controls, controls + n_controls);
    qs.push_back(target);
    int rc = check_qubits(st, static_cast<int>(qs.size()), qs.data());
    if (rc) return rc;
    QSV_HIP(hipSetDevice(st->device));
    std::vector<int> cbits(n_controls);
    for (int i = 0; i < n_controls; ++i) cbits[i] = bit_of(st, controls[i]);
    const int tbit = bit_of(st, target);
    if (st->n < QSV_LANE_BITS && n_controls + 1 > QSV_MAX_K)
        return qsv_fail(QSV_EINVAL, "too many controls for a tiny register");
    return controlled_1q_bits(st, n_controls, cbits.data(), tbit, m);
}

int qsv_apply_mcphase(qsv_state *st, int n_qubits, const int *qubits, double re, double im) {
    if (!valid(st) || (n_qubits > 0 && !qubits)) return qsv_fail(QSV_EINVAL, "null pointer");
    if (n_qubits < 0 || n_qubits > 64) return qsv_fail(QSV_EINVAL, "bad qubit count");
    int rc = check_qubits(st, n_qubits, qubits);
    if (rc) return rc;
    QSV_HIP(hipSetDevice(st->device));
    std::vector<int> cbits(n_qubits);
    for (int i = 0; i < n_qubits; ++i) cbits[i] = bit_of(st, qubits[i]);
    // narrow bits (< 3) become diagonal targets instead of controls (QSV_MIN_CTRL_BIT)
    std::vector<int> wide, narrow;
    for (int b : cbits) (b >= QSV_MIN_CTRL_BIT || st->n < QSV_LANE_BITS ? wide : narrow).push_back(b);
    if (narrow.empty()) return qsvk_phase(st, n_qubits, cbits.data(), re, im);
    while (narrow.size() > 2) {
        wide.push_back(narrow.back());
        narrow.pop_back();
    }
    double d[8] = {1, 0, 1, 0, 1, 0, 1, 0};
    const int k = static_cast<int>(narrow.size());
    d[2 * ((1 << k) - 1)] = re;
    d[2 * ((1 << k) - 1) + 1] = im;
    return qsvk_diag(st, k, narrow.data(), static_cast<int>(wide.size()), wide.data(), d);
}

int qsv_apply_kq(qsv_state *st, int k, const int *qubits, const double *m) {
    if (!valid(st) || !qubits || !m) return qsv_fail(QSV_EINVAL, "null pointer");
    if (k < 1 || k > QSV_MAX_K) return qsv_fail(QSV_EINVAL, "k must be in 1..6");
    int rc = check_qubits(st, k, qubits);
    if (rc) return rc;
    if (k == 1) return qsv_apply_1q(st, qubits[0], m);
    if (k == 2) return qsv_apply_2q(st, qubits[0], qubits[1], m);
    QSV_HIP(hipSetDevice(st->device));
    std::vector<int> bits(k);
    for (int j = 0; j < k; ++j) bits[j] = bit_of(st, qubits[j]);
    if (st->specialize && matrix_is_diagonal(1 << k, m)) {
        std::vector<double> d(2ull << k);
        const int D = 1 << k;
        for (int i = 0; i < D; ++i) {
            d[2 * i] = m[2 * (i * D + i)];
            d[2 * i + 1] = m[2 * (i * D + i) + 1];
        }
        return qsvk_diag(st, k, bits.data(), 0, nullptr, d.data());
    }
    // no host wait: the matrix has been copied into the staging ring (qsvk_stage) before this returns -- a drain of the
    // stream here (a leftover of round 1's synchronous upload, removed in round 3) left the GPU idle between the fused
    // blocks of a circuit while the host prepared the next one
    return qsvk_generic(st, k, bits.data(), m);
}

int qsv_apply_sequence(qsv_state *st, int k, const int *qubits, int n_gates, const int *arity, const int *legs,
                       const double *matrices, int *handled) {
    if (!valid(st) || !qubits || !arity || !legs || !matrices || !handled) return qsv_fail(QSV_EINVAL, "null pointer");
    *handled = 0;
    if (k < 1 || k > QSV_MAX_K) return qsv_fail(QSV_EINVAL, "k must be in 1..6");
    int rc = check_qubits(st, k, qubits);
    if (rc) return rc;
    if (n_gates < 1) return qsv_fail(QSV_EINVAL, "a gate sequence needs at least one gate");
    for (int g = 0; g < n_gates; ++g)
        if (arity[g] < 1 || arity[g] > k) return qsv_fail(QSV_EINVAL, "gate sequence: arity outside 1..k");
    QSV_HIP(hipSetDevice(st->device));
    int bits[QSV_MAX_K];
    for (int j = 0; j < k; ++j) bits[j] = bit_of(st, qubits[j]);
    // 6-qubit blocks with a short gate list: the list on LDS tiles (k_seq_tile; 1.67-1.88 ms against 2.0-2.1 for the dense
    // product on the benchmark circuit's blocks).  5-qubit blocks only on request: 1.79 ms there against 1.6 dense.
    static const int builtin_limit = [] {
        const char *e = getenv("QSV_TILE_SEQUENCE_GATES");
        return e ? atoi(e) : 12;
    }();
    const bool on_request = st->tile_sequence_gates >= 0;
    const int limit = on_request ? st->tile_sequence_gates : builtin_limit;
    if ((k == 6 || (k == 5 && on_request)) && n_gates <= limit) {
        rc = qsvk_sequence_tile(st, k, bits, n_gates, arity, legs, matrices);
        if (rc == QSV_OK) {
            *handled = 1;
            return QSV_OK;
        }
        if (rc != QSV_UNHANDLED_KQ) return rc;
    }
    if (k != 5 || st->n < QSV_LANE_BITS) return QSV_OK;       // only 5-qubit blocks have the register form (handled = 0)
    rc = qsvk_sequence5(st, bits, n_gates, arity, legs, matrices);
    if (rc == QSV_UNHANDLED_KQ) return QSV_OK;
    if (rc) return rc;
    *handled = 1;
    return QSV_OK;
}

int qsv_permute(qsv_state *st, const int *new_ordering) {
    if (!valid(st) || (st->n > 0 && !new_ordering)) return qsv_fail(QSV_EINVAL, "null pointer");
    if (st->kind != 0) return qsv_fail(QSV_ESTATE, "this call needs a qubit register");
    const int n = st->n;
    std::vector<int> seen(n, 0);
    for (int j = 0; j < n; ++j) {
        if (new_ordering[j] < 0 || new_ordering[j] >= n || seen[new_ordering[j]]++)
            return qsv_fail(QSV_EINVAL, "new_ordering must be a permutation of all qubits");
    }
    QSV_HIP(hipSetDevice(st->device));
    // qubit at position j moves to position new_ordering[j]:
    // destination bit (n-1-new_ordering[j]) takes source bit (n-1-j)
    std::vector<int> src_of_dst(n);
    for (int j = 0; j < n; ++j) src_of_dst[n - 1 - new_ordering[j]] = n - 1 - j;
    return qsvk_permute(st, src_of_dst.data());
}

// ---- measurement / insertion ------------------------------------------------------------------------

int qsv_measure_probs(qsv_state *st, int q, const double eig0[4], const double eig1[4], double *p0, double *p1) {
    if (!valid(st) || !eig0 || !eig1 || !p0 || !p1) return qsv_fail(QSV_EINVAL, "null pointer");
    int rc = check_qubits(st, 1, &q);
    if (rc) return rc;
    QSV_HIP(hipSetDevice(st->device));
    return qsvk_measure_probs(st, bit_of(st, q), eig0, eig1, p0, p1);
}

int qsv_collapse(qsv_state *st, int q, const double eig[4], double scale) {
    if (!valid(st) || !eig) return qsv_fail(QSV_EINVAL, "null pointer");
    int rc = check_qubits(st, 1, &q);
    if (rc) return rc;
    QSV_HIP(hipSetDevice(st->device));
    return qsvk_collapse(st, bit_of(st, q), eig, scale);
}

int qsv_measure(qsv_state *st, int q, const double eig0[4], const double eig1[4], int forced, double u01,
                int *outcome, double *p0, double *p1) {
    if (!valid(st) || !eig0 || !eig1 || !outcome) return qsv_fail(QSV_EINVAL, "null pointer");
    if (forced < -1 || forced > 1) return qsv_fail(QSV_EINVAL, "Measurement results must be from 0 or 1");
    double a = 0.0, b = 0.0;
    int rc = qsv_measure_probs(st, q, eig0, eig1, &a, &b);
    if (rc) return rc;
    if (p0) *p0 = a;
    if (p1) *p1 = b;
    int s = forced;
    if (s < 0) s = (u01 * (a + b) < a) ? 0 : 1;
    const double pn = s ? b : a;
    // the reference divides by the norm of the chosen branch; a zero-probability forced branch yields
    // inf/nan there as well
    const double scale = 1.0 / std::sqrt(pn);
    rc = qsvk_collapse(st, bit_of(st, q), s ? eig1 : eig0, scale);
    if (rc) return rc;
    *outcome = s;
    return QSV_OK;
}

int qsv_insert(qsv_state *st, int q, const double amp[4]) {
    if (!valid(st) || !amp) return qsv_fail(QSV_EINVAL, "null pointer");
    if (st->kind != 0) return qsv_fail(QSV_ESTATE, "this call needs a qubit register");
    if (q < 0 || q > st->n) return qsv_fail(QSV_EINVAL, "new_ordering must be a permutation of all qubits");
    if (st->n + 1 > 40) return qsv_fail(QSV_EINVAL, "register too large");
    QSV_HIP(hipSetDevice(st->device));
    // after insertion the register has n+1 qubits and the new one is qubit q: bit (n+1)-1-q
    return qsvk_insert(st, st->n - q, amp);
}

// ---- read-out ---------------------------------------------------------------------------------------

int qsv_norm2(qsv_state *st, double *out) {
    if (!valid(st) || !out) return qsv_fail(QSV_EINVAL, "null pointer");
    QSV_HIP(hipSetDevice(st->device));
    return qsvk_norm2(st, out);
}

int qsv_probabilities(qsv_state *st, const uint64_t *indices, int count, double *out) {
    if (!valid(st) || (count > 0 && (!indices || !out))) return qsv_fail(QSV_EINVAL, "null pointer");
    for (int i = 0; i < count; ++i)
        if (indices[i] >= st->amps) return qsv_fail(QSV_EINVAL, "amplitude index out of range");
    QSV_HIP(hipSetDevice(st->device));
    return qsvk_probabilities(st, indices, count, out);
}

int qsv_expect_pauli(qsv_state *st, int k, const int *qubits, const char *paulis, double *re, double *im) {
    if (!valid(st) || !re || !im || (k > 0 && (!qubits || !paulis))) return qsv_fail(QSV_EINVAL, "null pointer");
    if (k < 0 || k > 64) return qsv_fail(QSV_EINVAL, "bad Pauli string length");
    int rc = check_qubits(st, k, qubits);
    if (rc) return rc;
    uint64_t xmask = 0, zmask = 0;
    int n_y = 0;
    for (int j = 0; j < k; ++j) {
        const uint64_t bit = 1ull << bit_of(st, qubits[j]);
        switch (paulis[j]) {
            case 'I': case 'i': break;
            case 'X': case 'x': xmask |= bit; break;
            case 'Z': case 'z': zmask |= bit; break;
            case 'Y': case 'y': xmask |= bit; zmask |= bit; ++n_y; break;
            default: return qsv_fail(QSV_EINVAL, "Pauli letters must be I, X, Y or Z");
        }
    }
    QSV_HIP(hipSetDevice(st->device));
    return qsvk_expect_pauli(st, xmask, zmask, n_y, re, im);
}

int qsv_sample(qsv_state *st, int shots, const double *u, uint64_t *out) {
    if (!valid(st) || (shots > 0 && (!u || !out))) return qsv_fail(QSV_EINVAL, "null pointer");
    if (shots < 0 || shots > (1 << 24)) return qsv_fail(QSV_EINVAL, "shots must be in 0..2^24");
    if (shots == 0) return QSV_OK;
    QSV_HIP(hipSetDevice(st->device));
    return qsvk_sample(st, shots, u, out);
}

int qsv_inner(qsv_state *a, qsv_state *b, double *re, double *im) {
    if (!valid(a) || !valid(b) || !re || !im) return qsv_fail(QSV_EINVAL, "null pointer");
    if (a->amps != b->amps) return qsv_fail(QSV_EINVAL, "registers of different sizes");
    if (a->device != b->device) return qsv_fail(QSV_EINVAL, "registers on different devices");
    QSV_HIP(hipSetDevice(a->device));
    return qsvk_inner(a, b, re, im);
}

int qsv_reduced_density(qsv_state *st, int k, const int *qubits, double *rho) {
    if (!valid(st) || !qubits || !rho) return qsv_fail(QSV_EINVAL, "null pointer");
    if (st->kind != 0) return qsv_fail(QSV_ESTATE, "this call needs a qubit register");
    if (k < 1 || k > QSV_MAX_K) return qsv_fail(QSV_EINVAL, "keep between 1 and 6 qubits");
    int rc = check_qubits(st, k, qubits);
    if (rc) return rc;
    QSV_HIP(hipSetDevice(st->device));
    std::vector<int> bits(k);
    for (int j = 0; j < k; ++j) bits[j] = bit_of(st, qubits[j]);
    return qsvk_reduced_density(st, k, bits.data(), rho);
}

int qsv_expect_density(qsv_state *ket, qsv_state *rho, double *re, double *im) {
    if (!valid(ket) || !valid(rho) || !re || !im) return qsv_fail(QSV_EINVAL, "null pointer");
    if (ket->kind != 0 || rho->kind != 0) return qsv_fail(QSV_ESTATE, "this call needs qubit registers");
    if (rho->n != 2 * ket->n) return qsv_fail(QSV_EINVAL, "the density register must have twice the ket's qubits");
    if (ket->device != rho->device) return qsv_fail(QSV_EINVAL, "registers on different devices");
    QSV_HIP(hipSetDevice(rho->device));
    return qsvk_expect_density(ket, rho, re, im);
}

// ---- d-level modes ----------------------------------------------------------------------------------

static int qudit_amps(int n_modes, int d, uint64_t *amps) {
    if (n_modes < 0 || d < 2) return qsv_fail(QSV_EINVAL, "need n_modes >= 0 and d >= 2");
    uint64_t a = 1;
    for (int i = 0; i < n_modes; ++i) {
        if (a > (1ull << 40) / static_cast<uint64_t>(d)) return qsv_fail(QSV_EINVAL, "register too large");
        a *= static_cast<uint64_t>(d);
    }
    *amps = a;
    return QSV_OK;
}

int qsv_create_qudit(int n_modes, int d, int device, qsv_state **out) {
    uint64_t amps = 0;
    int rc = qudit_amps(n_modes, d, &amps);
    if (rc) return rc;
    return create_common(1, n_modes, d, amps, device, nullptr, 0, nullptr, out);
}

int qsv_create_qudit_view(int n_modes, int d, int device, void *dev_amps, uint64_t capacity_amps, void *hip_stream,
                          qsv_state **out) {
    uint64_t amps = 0;
    int rc = qudit_amps(n_modes, d, &amps);
    if (rc) return rc;
    if (!dev_amps) return qsv_fail(QSV_EINVAL, "null device pointer");
    return create_common(1, n_modes, d, amps, device, dev_amps, capacity_amps, hip_stream, out);
}

int qsv_qudit_shape(const qsv_state *st, int *n_modes, int *d) {
    if (!valid(st) || !n_modes || !d) return qsv_fail(QSV_EINVAL, "null pointer");
    if (st->kind != 1) return qsv_fail(QSV_ESTATE, "this call needs a qudit register");
    *n_modes = st->n;
    *d = st->d;
    return QSV_OK;
}

static int check_modes(const qsv_state *st, int k, const int *modes) {
    if (st->kind != 1) return qsv_fail(QSV_ESTATE, "this call needs a qudit register");
    for (int i = 0; i < k; ++i) {
        if (modes[i] < 0 || modes[i] >= st->n) return qsv_fail(QSV_EINVAL, "mode index out of range");
        for (int j = 0; j < i; ++j)
            if (modes[i] == modes[j]) return qsv_fail(QSV_EINVAL, "Indices must be distinct.");
    }
    return QSV_OK;
}

int qsv_apply_mode1(qsv_state *st, int mode, const double *m) {
    if (!valid(st) || !m) return qsv_fail(QSV_EINVAL, "null pointer");
    int rc = check_modes(st, 1, &mode);
    if (rc) return rc;
    QSV_HIP(hipSetDevice(st->device));
    return qsvq_mode1(st, mode, m, false);
}

int qsv_apply_mode1_diag(qsv_state *st, int mode, const double *diag) {
    if (!valid(st) || !diag) return qsv_fail(QSV_EINVAL, "null pointer");
    int rc = check_modes(st, 1, &mode);
    if (rc) return rc;
    QSV_HIP(hipSetDevice(st->device));
    return qsvq_mode1(st, mode, diag, true);
}

int qsv_apply_mode2(qsv_state *st, int mode0, int mode1, const double *m) {
    if (!valid(st) || !m) return qsv_fail(QSV_EINVAL, "null pointer");
    const int ms[2] = {mode0, mode1};
    int rc = check_modes(st, 2, ms);
    if (rc) return rc;
    QSV_HIP(hipSetDevice(st->device));
    return qsvq_mode2(st, mode0, mode1, m, false);
}

int qsv_apply_mode2_diag(qsv_state *st, int mode0, int mode1, const double *diag) {
    if (!valid(st) || !diag) return qsv_fail(QSV_EINVAL, "null pointer");
    const int ms[2] = {mode0, mode1};
    int rc = check_modes(st, 2, ms);
    if (rc) return rc;
    QSV_HIP(hipSetDevice(st->device));
    return qsvq_mode2(st, mode0, mode1, diag, true);
}

int qsv_apply_mode2_gather(qsv_state *st, int mode0, int mode1, int nnz, const int32_t *cols, const double *vals) {
    if (!valid(st) || !cols || !vals) return qsv_fail(QSV_EINVAL, "null pointer");
    if (nnz < 1 || nnz > 64) return qsv_fail(QSV_EINVAL, "nnz must be in 1..64");
    const int ms[2] = {mode0, mode1};
    int rc = check_modes(st, 2, ms);
    if (rc) return rc;
    QSV_HIP(hipSetDevice(st->device));
    return qsvq_mode2_gather(st, mode0, mode1, nnz, cols, vals);
}

int qsv_apply_mode2_blocks(qsv_state *st, int mode0, int mode1, int nblocks, const int32_t *sizes,
                           const int32_t *plane_indices, const double *mats) {
    if (!valid(st) || !sizes || !plane_indices || !mats) return qsv_fail(QSV_EINVAL, "null pointer");
    if (nblocks < 1 || nblocks > (1 << 20)) return qsv_fail(QSV_EINVAL, "bad block count");
    const int ms[2] = {mode0, mode1};
    int rc = check_modes(st, 2, ms);
    if (rc) return rc;
    QSV_HIP(hipSetDevice(st->device));
    return qsvq_mode2_blocks(st, mode0, mode1, nblocks, sizes, plane_indices, mats);
}

int qsv_mode_marginal(qsv_state *st, int mode, double *probs) {
    if (!valid(st) || !probs) return qsv_fail(QSV_EINVAL, "null pointer");
    int rc = check_modes(st, 1, &mode);
    if (rc) return rc;
    QSV_HIP(hipSetDevice(st->device));
    return qsvq_mode_marginal(st, mode, probs);
}

int qsv_mode_project(qsv_state *st, int mode, int level, double scale) {
    if (!valid(st)) return qsv_fail(QSV_EINVAL, "null state");
    int rc = check_modes(st, 1, &mode);
    if (rc) return rc;
    if (level < 0 || level >= st->d) return qsv_fail(QSV_EINVAL, "level out of range");
    QSV_HIP(hipSetDevice(st->device));
    return qsvq_mode_project(st, mode, level, scale);
}

int qsv_mode_insert(qsv_state *st, int mode, const double *vec) {
    if (!valid(st) || !vec) return qsv_fail(QSV_EINVAL, "null pointer");
    if (st->kind != 1) return qsv_fail(QSV_ESTATE, "this call needs a qudit register");
    if (mode < 0 || mode > st->n) return qsv_fail(QSV_EINVAL, "Cannot insert mode at this index");
    if (st->amps > (1ull << 40) / static_cast<uint64_t>(st->d)) return qsv_fail(QSV_EINVAL, "register too large");
    QSV_HIP(hipSetDevice(st->device));
    return qsvq_mode_insert(st, mode, vec);
}

int qsv_tensor_apply_axis(int device, void *hip_stream, const void *dev_in, void *dev_out, uint64_t L, uint64_t d_in,
                          uint64_t d_out, uint64_t R, const double *m) {
    if (!dev_in || !dev_out || !m) return qsv_fail(QSV_EINVAL, "null pointer");
    if (dev_in == dev_out) return qsv_fail(QSV_EINVAL, "in-place contraction is not supported: pass distinct buffers");
    if (L == 0 || d_in == 0 || d_out == 0 || R == 0) return qsv_fail(QSV_EINVAL, "empty tensor");
    return qsvq_tensor_axis(device, static_cast<hipStream_t>(hip_stream), static_cast<const amp_t *>(dev_in),
                            static_cast<amp_t *>(dev_out), L, d_in, d_out, R, m);
}

int qsv_tensor_apply_axis_dev(int device, void *hip_stream, const void *dev_in, void *dev_out, uint64_t L,
                              uint64_t d_in, uint64_t d_out, uint64_t R, const void *dev_m) {
    if (!dev_in || !dev_out || !dev_m) return qsv_fail(QSV_EINVAL, "null pointer");
    if (dev_in == dev_out) return qsv_fail(QSV_EINVAL, "in-place contraction is not supported: pass distinct buffers");
    if (L == 0 || d_in == 0 || d_out == 0 || R == 0) return qsv_fail(QSV_EINVAL, "empty tensor");
    return qsvq_tensor_axis_dev(device, static_cast<hipStream_t>(hip_stream), static_cast<const amp_t *>(dev_in),
                                static_cast<amp_t *>(dev_out), L, d_in, d_out, R, static_cast<const double *>(dev_m));
}

// ---- matrix-product-state sites ---------------------------------------------------------------------

namespace {
inline hipStream_t as_stream(void *s) { return static_cast<hipStream_t>(s); }
inline const amp_t *camp(const void *p) { return static_cast<const amp_t *>(p); }
inline amp_t *amp(void *p) { return static_cast<amp_t *>(p); }
inline bool empty_site(uint64_t L, uint64_t d, uint64_t R) { return L == 0 || d == 0 || R == 0; }
}  // namespace

int qsv_tensor_gemm(int device, void *hip_stream, int op_a, int op_b, uint64_t m, uint64_t n, uint64_t k,
                    const void *dev_a, const void *dev_b, void *dev_c) {
    if (!dev_a || !dev_b || !dev_c) return qsv_fail(QSV_EINVAL, "null pointer");
    if (op_a < 0 || op_a > 2 || op_b < 0 || op_b > 2) return qsv_fail(QSV_EINVAL, "op must be 0, 1 or 2");
    if (m == 0 || n == 0 || k == 0) return qsv_fail(QSV_EINVAL, "empty matrix");
    return qsvg_gemm(device, as_stream(hip_stream), op_a, op_b, m, n, k, camp(dev_a), camp(dev_b), amp(dev_c));
}

int qsv_tensor_svd_split(int device, void *hip_stream, void *dev_theta, uint64_t rows, uint64_t cols,
                         int64_t max_bond_dim, double abs_err, double rel_err, void *dev_m1, void *dev_m2,
                         uint64_t capacity, uint64_t *rank, double *singular_values) {
    if (!dev_theta || !dev_m1 || !dev_m2 || !rank) return qsv_fail(QSV_EINVAL, "null pointer");
    if (rows == 0 || cols == 0) return qsv_fail(QSV_EINVAL, "empty matrix");
    return qsvg_svd_split(device, as_stream(hip_stream), amp(dev_theta), rows, cols, max_bond_dim, abs_err, rel_err,
                          amp(dev_m1), amp(dev_m2), capacity, rank, singular_values);
}

int qsv_tensor_rsvd_split(int device, void *hip_stream, const void *dev_theta, uint64_t rows, uint64_t cols,
                          int64_t max_bond_dim, int probes, int power_iterations, const void *dev_omega,
                          double abs_err, double rel_err, void *dev_m1, void *dev_m2, uint64_t capacity,
                          uint64_t *rank, double *singular_values) {
    if (!dev_theta || !dev_m1 || !dev_m2 || !rank) return qsv_fail(QSV_EINVAL, "null pointer");   // dev_omega may be NULL
    if (rows == 0 || cols == 0) return qsv_fail(QSV_EINVAL, "empty matrix");
    if (max_bond_dim < 1 || probes < max_bond_dim || power_iterations < 0)
        return qsv_fail(QSV_EINVAL, "need max_bond_dim >= 1, probes >= max_bond_dim, power_iterations >= 0");
    return qsvg_rsvd_split(device, as_stream(hip_stream), camp(dev_theta), rows, cols, max_bond_dim, probes,
                           power_iterations, camp(dev_omega), abs_err, rel_err, amp(dev_m1), amp(dev_m2), capacity,
                           rank, singular_values);
}

int qsv_tensor_skinny_gemm(int device, void *hip_stream, int op, uint64_t n, uint64_t m, int l, const void *dev_a,
                           const void *dev_q, void *dev_y) {
    if (!dev_a || !dev_q || !dev_y) return qsv_fail(QSV_EINVAL, "null pointer");
    if (n == 0 || m == 0) return qsv_fail(QSV_EINVAL, "empty matrix");
    if (op < 0 || op > 3) return qsv_fail(QSV_EINVAL, "op must be 0 (A), 1 (A^H), 2 (A^T) or 3 (conj A)");
    return qsvg_skinny_gemm(device, as_stream(hip_stream), op, n, m, l, camp(dev_a), camp(dev_q), amp(dev_y));
}

int qsv_tensor_release_workspace(int device) { return qsvg_release_workspace(device); }

int qsv_tensor_scale_axis(int device, void *hip_stream, void *dev_t, uint64_t L, uint64_t d, uint64_t R,
                          const void *dev_diag) {
    if (!dev_t || !dev_diag) return qsv_fail(QSV_EINVAL, "null pointer");
    if (empty_site(L, d, R)) return qsv_fail(QSV_EINVAL, "empty tensor");
    return qsvq_tensor_scale_axis(device, as_stream(hip_stream), amp(dev_t), L, d, R,
                                  static_cast<const double *>(dev_diag));
}

int qsv_tensor_plane_diag(int device, void *hip_stream, void *dev_theta, uint64_t L, uint64_t d, uint64_t R,
                          const void *dev_plane) {
    if (!dev_theta || !dev_plane) return qsv_fail(QSV_EINVAL, "null pointer");
    if (empty_site(L, d, R)) return qsv_fail(QSV_EINVAL, "empty tensor");
    return qsvq_tensor_plane_diag(device, as_stream(hip_stream), amp(dev_theta), L, d, R,
                                  static_cast<const double *>(dev_plane));
}

int qsv_tensor_plane_gather(int device, void *hip_stream, const void *dev_in, void *dev_out, uint64_t L, uint64_t d,
                            uint64_t R, int per_point, const int32_t *dev_cols, const void *dev_vals) {
    if (!dev_in || !dev_out || !dev_cols || !dev_vals) return qsv_fail(QSV_EINVAL, "null pointer");
    if (dev_in == dev_out) return qsv_fail(QSV_EINVAL, "in-place resampling is not supported: pass distinct buffers");
    if (empty_site(L, d, R) || per_point < 1) return qsv_fail(QSV_EINVAL, "empty tensor or table");
    if (d > 46340) return qsv_fail(QSV_EINVAL, "plane indices are 32-bit: d must be <= 46340");
    return qsvq_tensor_plane_gather(device, as_stream(hip_stream), camp(dev_in), amp(dev_out), L, d, R, per_point,
                                    dev_cols, static_cast<const double *>(dev_vals));
}

int qsv_tensor_plane_phase(int device, void *hip_stream, void *dev_theta, uint64_t L, uint64_t d, uint64_t R,
                           const void *dev_grid, double strength) {
    if (!dev_theta || !dev_grid) return qsv_fail(QSV_EINVAL, "null pointer");
    if (empty_site(L, d, R)) return qsv_fail(QSV_EINVAL, "empty tensor");
    return qsvq_tensor_plane_phase(device, as_stream(hip_stream), amp(dev_theta), L, d, R,
                                   static_cast<const double *>(dev_grid), strength);
}

int qsv_tensor_plane_affine(int device, void *hip_stream, const void *dev_in, void *dev_out, uint64_t L, uint64_t d,
                            uint64_t R, const void *dev_grid, const double *a) {
    if (!dev_in || !dev_out || !dev_grid || !a) return qsv_fail(QSV_EINVAL, "null pointer");
    if (dev_in == dev_out) return qsv_fail(QSV_EINVAL, "in-place resampling is not supported: pass distinct buffers");
    if (empty_site(L, d, R) || d < 2) return qsv_fail(QSV_EINVAL, "need a grid of at least two points");
    return qsvq_tensor_plane_affine(device, as_stream(hip_stream), camp(dev_in), amp(dev_out), L, d, R,
                                    static_cast<const double *>(dev_grid), a);
}

int qsv_tensor_take_level(int device, void *hip_stream, const void *dev_in, void *dev_out, uint64_t L, uint64_t d,
                          uint64_t R, uint64_t level, double scale) {
    if (!dev_in || !dev_out) return qsv_fail(QSV_EINVAL, "null pointer");
    if (empty_site(L, d, R) || level >= d) return qsv_fail(QSV_EINVAL, "level out of range");
    return qsvq_tensor_take_level(device, as_stream(hip_stream), camp(dev_in), amp(dev_out), L, d, R, level, scale);
}

int qsv_tensor_insert_axis(int device, void *hip_stream, const void *dev_in, void *dev_out, uint64_t L, uint64_t d,
                           uint64_t R, const void *dev_vec) {
    if (!dev_in || !dev_out || !dev_vec) return qsv_fail(QSV_EINVAL, "null pointer");
    if (empty_site(L, d, R)) return qsv_fail(QSV_EINVAL, "empty tensor");
    return qsvq_tensor_insert_axis(device, as_stream(hip_stream), camp(dev_in), amp(dev_out), L, d, R,
                                   static_cast<const double *>(dev_vec));
}

int qsv_tensor_outer(int device, void *hip_stream, const void *dev_p, const void *dev_q, void *dev_out, uint64_t X,
                     uint64_t Y, uint64_t Z, uint64_t W, int swap_last) {
    if (!dev_p || !dev_q || !dev_out) return qsv_fail(QSV_EINVAL, "null pointer");
    if (X == 0 || Y == 0 || Z == 0 || W == 0) return qsv_fail(QSV_EINVAL, "empty tensor");
    return qsvq_tensor_outer(device, as_stream(hip_stream), camp(dev_p), camp(dev_q), amp(dev_out), X, Y, Z, W,
                             swap_last);
}

int qsv_tensor_axis_overlap(int device, void *hip_stream, const void *dev_z, const void *dev_t, uint64_t L, uint64_t d,
                            uint64_t R, void *dev_out) {
    if (!dev_z || !dev_t || !dev_out) return qsv_fail(QSV_EINVAL, "null pointer");
    if (empty_site(L, d, R)) return qsv_fail(QSV_EINVAL, "empty tensor");
    return qsvq_tensor_axis_overlap(device, as_stream(hip_stream), camp(dev_z), camp(dev_t), L, d, R,
                                    static_cast<double *>(dev_out));
}

// ---- timing -----------------------------------------------------------------------------------------

int qsv_timer_start(qsv_state *st) {
    if (!valid(st)) return qsv_fail(QSV_EINVAL, "null state");
    QSV_HIP(hipSetDevice(st->device));
    QSV_HIP(hipEventRecord(st->ev_start, st->stream));
    return QSV_OK;
}

int qsv_timer_stop(qsv_state *st, float *elapsed_ms) {
    if (!valid(st) || !elapsed_ms) return qsv_fail(QSV_EINVAL, "null pointer");
    QSV_HIP(hipSetDevice(st->device));
    QSV_HIP(hipEventRecord(st->ev_stop, st->stream));
    QSV_HIP(hipEventSynchronize(st->ev_stop));
    QSV_HIP(hipEventElapsedTime(elapsed_ms, st->ev_start, st->ev_stop));
    return QSV_OK;
}

int qsv_last_kernel(const qsv_state *st, char *buf, size_t buf_len) {
    if (!valid(st) || !buf || buf_len == 0) return qsv_fail(QSV_EINVAL, "null pointer");
    std::strncpy(buf, st->last_kernel, buf_len - 1);
    buf[buf_len - 1] = '\0';
    return QSV_OK;
}

int qsv_event_record(qsv_state *st, int slot) {
    if (!valid(st)) return qsv_fail(QSV_EINVAL, "null state");
    if (slot < 0 || slot >= 16384) return qsv_fail(QSV_EINVAL, "event slot out of range");
    QSV_HIP(hipSetDevice(st->device));
    if (st->marks.size() <= static_cast<size_t>(slot)) st->marks.resize(slot + 1, nullptr);
    if (!st->marks[slot]) QSV_HIP(hipEventCreate(&st->marks[slot]));
    QSV_HIP(hipEventRecord(st->marks[slot], st->stream));
    return QSV_OK;
}

int qsv_event_elapsed_ms(qsv_state *st, int slot_a, int slot_b, float *elapsed_ms) {
    if (!valid(st) || !elapsed_ms) return qsv_fail(QSV_EINVAL, "null pointer");
    const int hi = slot_a > slot_b ? slot_a : slot_b;
    if (slot_a < 0 || slot_b < 0 || static_cast<size_t>(hi) >= st->marks.size() || !st->marks[slot_a] ||
        !st->marks[slot_b])
        return qsv_fail(QSV_EINVAL, "event slot was never recorded");
    QSV_HIP(hipSetDevice(st->device));
    QSV_HIP(hipEventSynchronize(st->marks[slot_b]));
    QSV_HIP(hipEventElapsedTime(elapsed_ms, st->marks[slot_a], st->marks[slot_b]));
    return QSV_OK;
}

}  // extern "C"
