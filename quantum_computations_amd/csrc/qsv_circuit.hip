// Whole circuits in ONE launch, on registers that fit a workgroup's LDS (n <= 13 qubits: 2^13 x 16 B = 128 KiB).
//
// The reference's own register sizes are 4..12 qubits (its dense operators stop there), and its drivers sweep thousands
// of such circuits through a multiprocessing.Pool (impact_.../randomised_benchmarking.py:60-76,
// average_clifford_fidelity.py:212).  At those sizes the per-gate path is bound by its launches (12 us per gate through
// Simulator.run); here one workgroup keeps the register in LDS, streams the circuit's gate list from memory and runs the
// whole `for gate in circuit` loop of dv_simulator/simulator.py:40-52 -- matrix gates, M (forced or sampled from a
// host-drawn uniform), Insert, ClassicalControl -- without leaving the kernel.  One workgroup per circuit instance: a
// batch of circuits fills the 256 CUs (qsv_run_programs; Simulator.run_batch).
//
// Program format (64-bit words; doubles are stored by bit pattern).  Every op starts with a header word
//     op[0:8] | k[8:12] | len[12:28] (words, header included) | b0[28:34] | b1[34:40] | ... | b5[58:64]
// followed by its payload.  b_j = bit position of matrix leg j (leg 0 most significant, as in expand_gate,
// numpy_quantum.py:243-247) in the register's CURRENT size.  No op straddles a PROG_CHUNK boundary (the host pads with
// NOPs), so a chunk staged in LDS is always self-contained.
//     DENSE   k = 1..4   payload 2 * 4^k doubles: row-major complex matrix
//     MEASURE            b0; payload eig0 (4 doubles), eig1 (4 doubles), forced (int64: -1 / 0 / 1), u01 (double)
//     INSERT             b0 = bit of the new qubit in the grown register; payload a0, a1 (4 doubles)
//     CCTRL              payload pos_mask, neg_mask over the measurement record: the NEXT op runs iff all pos bits are 1
//                        and all neg bits are 0 (ClassicalControl.eval, simulator.py:16-17)
//     NOP, END
#include "qsv_internal.h"

#include <algorithm>
#include <cstring>
#include <mutex>

namespace {

constexpr int PROG_CHUNK = 2048;      // words (16 KiB) of program staged in LDS at a time
constexpr int RUN_MAX_QUBITS = 13;
// workgroup size: 256 threads up to 10 qubits (512 pairs: two per thread), 1024 above (n = 13: four pairs per thread;
// with 256 threads a gate at n = 12 took 1.46 us, the loop over 2048 pairs being 8 trips of LDS latency)

enum : uint32_t { OP_END = 0, OP_NOP = 1, OP_DENSE = 2, OP_MEASURE = 3, OP_INSERT = 4, OP_CCTRL = 5 };

struct cplx {
    double re, im;
};
__device__ __forceinline__ amp_t cmul(cplx m, amp_t a) { return amp_t{m.re * a.x - m.im * a.y, m.re * a.y + m.im * a.x}; }
__device__ __forceinline__ amp_t cfma(cplx m, amp_t a, amp_t acc) {
    acc.x = fma(m.re, a.x, acc.x);
    acc.x = fma(-m.im, a.y, acc.x);
    acc.y = fma(m.re, a.y, acc.y);
    acc.y = fma(m.im, a.x, acc.y);
    return acc;
}
__device__ __forceinline__ uint32_t insert_zero(uint32_t w, int p) {
    const uint32_t low = w & ((1u << p) - 1u);
    return ((w >> p) << (p + 1)) | low;
}
__device__ __forceinline__ double as_double(uint64_t w) { return __longlong_as_double(static_cast<long long>(w)); }

struct RunArgs {
    const uint64_t *prog;          // all programs
    const uint64_t *prog_off;      // [count + 1] word offsets
    const int32_t *n0;             // [count] qubits of the initial register
    const amp_t *states_in;        // initial kets, concatenated
    const uint64_t *state_off;     // [count + 1] amplitude offsets
    amp_t *states_out;             // final kets, concatenated
    const uint64_t *out_off;       // [count + 1]
    int32_t *results;              // measurement outcomes, concatenated
    double *probs;                 // (p0, p1) per measurement, same offsets x 2
    const uint64_t *result_off;    // [count + 1]
    uint32_t reg_amps;             // LDS slots of the register
};

// sum of (x, y) over the workgroup, the same value in every thread (so that every thread takes the same branch on it)
template <int RUN_THREADS>
__device__ __forceinline__ void block_sum2(double &x, double &y, double *red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        x += __shfl_xor(x, o, 64);
        y += __shfl_xor(y, o, 64);
    }
    const int wave = threadIdx.x >> 6;
    __syncthreads();                       // red[] may still be read from the previous reduction
    if ((threadIdx.x & 63) == 0) {
        red[2 * wave] = x;
        red[2 * wave + 1] = y;
    }
    __syncthreads();
    x = y = 0.0;
#pragma unroll
    for (int w = 0; w < RUN_THREADS / 64; ++w) {   // fixed order: deterministic
        x += red[2 * w];
        y += red[2 * w + 1];
    }
}

template <int K, int RUN_THREADS>
__device__ __forceinline__ void dense_k(amp_t *reg, int n, const uint32_t (&b)[6], const uint64_t *payload) {
    constexpr int D = 1 << K;
    // sorted target bits for the enumeration of the groups, per-column offsets in the caller's leg order
    int sorted[K];
#pragma unroll
    for (int j = 0; j < K; ++j) sorted[j] = static_cast<int>(b[j]);
#pragma unroll
    for (int i = 0; i < K; ++i)
#pragma unroll
        for (int j = i + 1; j < K; ++j)
            if (sorted[j] < sorted[i]) {
                const int t = sorted[i];
                sorted[i] = sorted[j];
                sorted[j] = t;
            }
    uint32_t off[D];
#pragma unroll
    for (int c = 0; c < D; ++c) {
        off[c] = 0;
#pragma unroll
        for (int leg = 0; leg < K; ++leg)
            if ((c >> (K - 1 - leg)) & 1) off[c] |= 1u << b[leg];
    }
    const uint32_t groups = 1u << (n - K);
    if constexpr (K <= 2) {
        cplx m[D * D];                    // the whole matrix in registers (K = 2: 64 VGPRs), one LDS broadcast each
#pragma unroll
        for (int e = 0; e < D * D; ++e) m[e] = cplx{as_double(payload[2 * e]), as_double(payload[2 * e + 1])};
        for (uint32_t g = threadIdx.x; g < groups; g += RUN_THREADS) {
            uint32_t base = g;
#pragma unroll
            for (int j = 0; j < K; ++j) base = insert_zero(base, sorted[j]);
            amp_t x[D];
#pragma unroll
            for (int c = 0; c < D; ++c) x[c] = reg[base | off[c]];
#pragma unroll
            for (int r = 0; r < D; ++r) {
                amp_t acc = {0.0, 0.0};
#pragma unroll
                for (int c = 0; c < D; ++c) acc = cfma(m[r * D + c], x[c], acc);
                reg[base | off[r]] = acc;   // a group belongs to one thread: in place
            }
        }
    } else {
        for (uint32_t g = threadIdx.x; g < groups; g += RUN_THREADS) {
            uint32_t base = g;
#pragma unroll
            for (int j = 0; j < K; ++j) base = insert_zero(base, sorted[j]);
            amp_t x[D];
#pragma unroll
            for (int c = 0; c < D; ++c) x[c] = reg[base | off[c]];
#pragma unroll 1
            for (int r = 0; r < D; ++r) {
                amp_t acc = {0.0, 0.0};
#pragma unroll
                for (int c = 0; c < D; ++c)
                    acc = cfma(cplx{as_double(payload[2 * (r * D + c)]), as_double(payload[2 * (r * D + c) + 1])}, x[c], acc);
                uint32_t o = 0;
#pragma unroll
                for (int leg = 0; leg < K; ++leg)
                    if ((r >> (K - 1 - leg)) & 1) o |= 1u << b[leg];
                reg[base | o] = acc;
            }
        }
    }
}

template <int RUN_THREADS>
__global__ __launch_bounds__(RUN_THREADS) void k_run_programs(const RunArgs a) {
    constexpr int OWN_MAX = (1 << RUN_MAX_QUBITS) / 2 / RUN_THREADS;   // pairs a thread owns in the two-phase ops
    extern __shared__ __attribute__((aligned(16))) char smem_run[];
    amp_t *reg = reinterpret_cast<amp_t *>(smem_run);
    uint64_t *prog = reinterpret_cast<uint64_t *>(reg + a.reg_amps);
    double *red = reinterpret_cast<double *>(prog + PROG_CHUNK);
    const int inst = blockIdx.x;
    const uint32_t tid = threadIdx.x;
    int n = a.n0[inst];
    {
        const amp_t *src = a.states_in + a.state_off[inst];
        for (uint32_t i = tid; i < (1u << n); i += RUN_THREADS) reg[i] = src[i];
    }
    const uint64_t *p = a.prog + a.prog_off[inst];
    const uint64_t plen = a.prog_off[inst + 1] - a.prog_off[inst];
    uint64_t record = 0;          // measurement outcomes so far, bit i = outcome i (the first 64 can be controls)
    int n_results = 0;
    bool skip = false, done = false;
    for (uint64_t c0 = 0; c0 < plen && !done; c0 += PROG_CHUNK) {
        __syncthreads();          // the previous chunk has been consumed; the register's last writes are visible
        const uint32_t chunk = static_cast<uint32_t>(plen - c0 < PROG_CHUNK ? plen - c0 : PROG_CHUNK);
        for (uint32_t i = tid; i < chunk; i += RUN_THREADS) prog[i] = p[c0 + i];
        __syncthreads();
        uint32_t pc = 0;
        while (pc < chunk) {
            const uint64_t h = prog[pc];
            // the header is the same in every thread; tell the compiler (scalar branches, no divergence handling)
            const uint32_t h_lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(h));
            const uint32_t h_hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(h >> 32));
            const uint64_t hh = (static_cast<uint64_t>(h_hi) << 32) | h_lo;
            const uint32_t op = h_lo & 0xffu, k = (h_lo >> 8) & 0xfu, len = (h_lo >> 12) & 0xffffu;
            uint32_t b[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) b[j] = static_cast<uint32_t>(hh >> (28 + 6 * j)) & 0x3fu;
            const uint64_t *payload = prog + pc + 1;
            if (op == OP_END) {
                done = true;
                break;
            }
            const bool run = !skip;
            skip = false;
            if (op == OP_CCTRL) {
                const uint64_t pos = payload[0], neg = payload[1];
                skip = !(((record & pos) == pos) && ((record & neg) == 0));
            } else if (run && op == OP_DENSE) {
                switch (k) {
                    case 1: dense_k<1, RUN_THREADS>(reg, n, b, payload); break;
                    case 2: dense_k<2, RUN_THREADS>(reg, n, b, payload); break;
                    case 3: dense_k<3, RUN_THREADS>(reg, n, b, payload); break;
                    default: dense_k<4, RUN_THREADS>(reg, n, b, payload); break;
                }
                __syncthreads();
            } else if (run && op == OP_MEASURE) {
                // M.apply (gates.py:165-186): both branch kets r_s = e_s[0] a0 + e_s[1] a1 (unconjugated eigenvectors),
                // outcome from the host-drawn uniform as np.random.choice(p = [p0, p1]) picks it, collapse + renormalise
                const int bit = static_cast<int>(b[0]);
                const cplx e00{as_double(payload[0]), as_double(payload[1])}, e01{as_double(payload[2]), as_double(payload[3])};
                const cplx e10{as_double(payload[4]), as_double(payload[5])}, e11{as_double(payload[6]), as_double(payload[7])};
                const long long forced = static_cast<long long>(payload[8]);
                const double u01 = as_double(payload[9]);
                const uint32_t pairs = 1u << (n - 1);
                double p0 = 0.0, p1 = 0.0;
                for (uint32_t q = tid; q < pairs; q += RUN_THREADS) {
                    const uint32_t i0 = insert_zero(q, bit);
                    const amp_t a0 = reg[i0], a1 = reg[i0 | (1u << bit)];
                    const amp_t r0 = cfma(e01, a1, cmul(e00, a0)), r1 = cfma(e11, a1, cmul(e10, a0));
                    p0 += r0.x * r0.x + r0.y * r0.y;
                    p1 += r1.x * r1.x + r1.y * r1.y;
                }
                block_sum2<RUN_THREADS>(p0, p1, red);
                const int s = forced >= 0 ? static_cast<int>(forced) : (u01 < p0 / (p0 + p1) ? 0 : 1);
                const cplx ea = s ? e10 : e00, eb = s ? e11 : e01;
                const double scale = 1.0 / sqrt(s ? p1 : p0);
                amp_t keep[OWN_MAX];
#pragma unroll
                for (int j = 0; j < OWN_MAX; ++j) {
                    const uint32_t q = tid + j * RUN_THREADS;
                    if (q < pairs) {
                        const uint32_t i0 = insert_zero(q, bit);
                        const amp_t r = cfma(eb, reg[i0 | (1u << bit)], cmul(ea, reg[i0]));
                        keep[j] = amp_t{r.x * scale, r.y * scale};
                    }
                }
                __syncthreads();
#pragma unroll
                for (int j = 0; j < OWN_MAX; ++j) {
                    const uint32_t q = tid + j * RUN_THREADS;
                    if (q < pairs) reg[q] = keep[j];
                }
                if (tid == 0) {
                    a.results[a.result_off[inst] + n_results] = s;
                    a.probs[2 * (a.result_off[inst] + n_results)] = p0;
                    a.probs[2 * (a.result_off[inst] + n_results) + 1] = p1;
                }
                if (n_results < 64) record |= static_cast<uint64_t>(s) << n_results;
                ++n_results;
                --n;
                __syncthreads();
            } else if (run && op == OP_INSERT) {
                // Insert.apply (gates.py:145-153): kron with the new qubit, which lands on bit b0 of the grown register
                const int bit = static_cast<int>(b[0]);
                const cplx a0{as_double(payload[0]), as_double(payload[1])}, a1{as_double(payload[2]), as_double(payload[3])};
                const uint32_t old_amps = 1u << n;
                amp_t keep[OWN_MAX];
#pragma unroll
                for (int j = 0; j < OWN_MAX; ++j) {
                    const uint32_t q = tid + j * RUN_THREADS;
                    if (q < old_amps) keep[j] = reg[q];
                }
                __syncthreads();
#pragma unroll
                for (int j = 0; j < OWN_MAX; ++j) {
                    const uint32_t q = tid + j * RUN_THREADS;
                    if (q < old_amps) {
                        const uint32_t i0 = insert_zero(q, bit);
                        reg[i0] = cmul(a0, keep[j]);
                        reg[i0 | (1u << bit)] = cmul(a1, keep[j]);
                    }
                }
                ++n;
                __syncthreads();
            }
            pc += len;
        }
    }
    __syncthreads();
    amp_t *dst = a.states_out + a.out_off[inst];
    for (uint32_t i = tid; i < (1u << n); i += RUN_THREADS) dst[i] = reg[i];
}

// grow-only device + pinned host staging of the executor, one per device
struct RunPool {
    char *dev = nullptr, *host = nullptr;
    size_t bytes = 0;
    hipStream_t stream = nullptr;
};
RunPool g_pools[16];
std::mutex g_pool_mutex;

size_t pad256(size_t x) { return (x + 255) / 256 * 256; }

}  // namespace

extern "C" int qsv_run_programs(int device, int count, int max_qubits, const uint64_t *programs, const uint64_t *prog_offsets,
                                const int *n_initial, const double *states_in, const uint64_t *state_offsets,
                                double *states_out, const uint64_t *out_offsets, int *results, double *probabilities,
                                const uint64_t *result_offsets) {
    if (count <= 0) return count == 0 ? QSV_OK : qsv_fail(QSV_EINVAL, "negative instance count");
    if (!programs || !prog_offsets || !n_initial || !states_in || !state_offsets || !states_out || !out_offsets || !result_offsets)
        return qsv_fail(QSV_EINVAL, "null pointer");
    if (max_qubits < 0 || max_qubits > RUN_MAX_QUBITS)
        return qsv_fail(QSV_EINVAL, "the single-launch executor holds registers of at most 13 qubits");
    if (device < 0 || device >= 16) return qsv_fail(QSV_EINVAL, "device ordinal out of range");
    const uint64_t n_meas = result_offsets[count];
    if (n_meas && (!results || !probabilities)) return qsv_fail(QSV_EINVAL, "null pointer");
    for (int i = 0; i < count; ++i)
        if (n_initial[i] < 0 || n_initial[i] > max_qubits || state_offsets[i + 1] - state_offsets[i] != (1ull << n_initial[i]))
            return qsv_fail(QSV_EINVAL, "initial register size does not match its qubit count");
    int dev_count = 0;
    hipError_t e = hipGetDeviceCount(&dev_count);
    if (e != hipSuccess || dev_count <= 0)
        return qsv_fail(QSV_EHIP, std::string("no HIP device available: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0"));
    if (device >= dev_count) return qsv_fail(QSV_EINVAL, "device ordinal out of range");
    QSV_HIP(hipSetDevice(device));
    // one blob in, one blob out
    const size_t b_prog = sizeof(uint64_t) * prog_offsets[count], b_tab = sizeof(uint64_t) * (count + 1);
    const size_t b_n0 = sizeof(int32_t) * count, b_in = sizeof(amp_t) * state_offsets[count];
    const size_t b_out = sizeof(amp_t) * out_offsets[count], b_res = sizeof(int32_t) * n_meas, b_prob = 2 * sizeof(double) * n_meas;
    size_t o = 0;
    const size_t o_prog = o; o += pad256(b_prog);
    const size_t o_poff = o; o += pad256(b_tab);
    const size_t o_soff = o; o += pad256(b_tab);
    const size_t o_ooff = o; o += pad256(b_tab);
    const size_t o_roff = o; o += pad256(b_tab);
    const size_t o_n0 = o; o += pad256(b_n0);
    const size_t o_in = o; o += pad256(b_in);
    const size_t in_bytes = o;
    const size_t o_out = o; o += pad256(b_out);
    const size_t o_prob = o; o += pad256(b_prob);
    const size_t o_res = o; o += pad256(b_res);
    const size_t total = o;
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    RunPool &pool = g_pools[device];
    if (!pool.stream) QSV_HIP(hipStreamCreateWithFlags(&pool.stream, hipStreamNonBlocking));
    if (pool.bytes < total) {
        if (pool.dev) (void)hipFree(pool.dev);
        if (pool.host) (void)hipHostFree(pool.host);
        pool.dev = pool.host = nullptr;
        pool.bytes = 0;
        const size_t want = std::max<size_t>(total + total / 2, 1u << 20);
        if (hipMalloc(reinterpret_cast<void **>(&pool.dev), want) != hipSuccess)
            return qsv_fail(QSV_ENOMEM, "device staging of the circuit executor");
        if (hipHostMalloc(reinterpret_cast<void **>(&pool.host), want, hipHostMallocDefault) != hipSuccess) {
            (void)hipFree(pool.dev);
            pool.dev = nullptr;
            return qsv_fail(QSV_ENOMEM, "pinned staging of the circuit executor");
        }
        pool.bytes = want;
    }
    std::memcpy(pool.host + o_prog, programs, b_prog);
    std::memcpy(pool.host + o_poff, prog_offsets, b_tab);
    std::memcpy(pool.host + o_soff, state_offsets, b_tab);
    std::memcpy(pool.host + o_ooff, out_offsets, b_tab);
    std::memcpy(pool.host + o_roff, result_offsets, b_tab);
    std::memcpy(pool.host + o_n0, n_initial, b_n0);
    std::memcpy(pool.host + o_in, states_in, b_in);
    QSV_HIP(hipMemcpyAsync(pool.dev, pool.host, in_bytes, hipMemcpyHostToDevice, pool.stream));
    RunArgs a;
    a.prog = reinterpret_cast<const uint64_t *>(pool.dev + o_prog);
    a.prog_off = reinterpret_cast<const uint64_t *>(pool.dev + o_poff);
    a.n0 = reinterpret_cast<const int32_t *>(pool.dev + o_n0);
    a.states_in = reinterpret_cast<const amp_t *>(pool.dev + o_in);
    a.state_off = reinterpret_cast<const uint64_t *>(pool.dev + o_soff);
    a.states_out = reinterpret_cast<amp_t *>(pool.dev + o_out);
    a.out_off = reinterpret_cast<const uint64_t *>(pool.dev + o_ooff);
    a.results = reinterpret_cast<int32_t *>(pool.dev + o_res);
    a.probs = reinterpret_cast<double *>(pool.dev + o_prob);
    a.result_off = reinterpret_cast<const uint64_t *>(pool.dev + o_roff);
    a.reg_amps = 1u << max_qubits;
    const size_t lds = sizeof(amp_t) * a.reg_amps + sizeof(uint64_t) * PROG_CHUNK + sizeof(double) * 2 * 16;
    if (max_qubits <= 10) {
        QSV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_run_programs<256>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    static_cast<int>(lds)));
        hipLaunchKernelGGL(k_run_programs<256>, dim3(static_cast<unsigned>(count)), dim3(256), lds, pool.stream, a);
    } else {
        QSV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_run_programs<1024>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    static_cast<int>(lds)));
        hipLaunchKernelGGL(k_run_programs<1024>, dim3(static_cast<unsigned>(count)), dim3(1024), lds, pool.stream, a);
    }
    e = hipGetLastError();
    if (e != hipSuccess) return qsv_fail(QSV_EHIP, std::string("kernel launch: ") + hipGetErrorString(e));
    QSV_HIP(hipMemcpyAsync(pool.host + o_out, pool.dev + o_out, total - o_out, hipMemcpyDeviceToHost, pool.stream));
    QSV_HIP(hipStreamSynchronize(pool.stream));
    std::memcpy(states_out, pool.host + o_out, b_out);
    if (n_meas) {
        std::memcpy(results, pool.host + o_res, b_res);
        std::memcpy(probabilities, pool.host + o_prob, b_prob);
    }
    return QSV_OK;
}
