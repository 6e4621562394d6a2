// Internal declarations shared by the HIP translation units of libqsv.so (not part of the C ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "qsv.h"

// One amplitude = complex128 = one 16-byte vector: exactly one dwordx4 per lane, 1 KiB per wave64 access.
typedef double amp_t __attribute__((ext_vector_type(2)));

constexpr int QSV_BLOCK = 256;    // threads per workgroup (4 waves, one per SIMD)
constexpr int QSV_LANE_BITS = 6;  // index bits 0..5 are always spread over the 64 lanes of a wave
constexpr int QSV_MAX_INS = 48;   // max inserted bit positions (high targets + high controls)
constexpr int QSV_MAX_K = 6;      // generic k-qubit gates: 2^k x 2^k matrix, k <= 6

// Arguments of the dense / phase kernels, passed by value (kernarg segment -> SGPRs).
//
// Work item w in [0, W) enumerates the "free" index bits.  deposit(w) re-inserts a zero at each position
// pos[0] < pos[1] < ... (all >= QSV_LANE_BITS, so the six lane bits pass through and a wave always touches
// 64 consecutive amplitudes) and ORs in or_mask (high controls fixed to 1).  The 2^KH amplitudes a thread
// owns sit at deposit(w) + hoff[h]; the 2^KL "low" target bits are lane bits and are resolved by wave64
// shuffles with lane masks lxor[x].
struct GateArgs {
    uint64_t W;
    uint64_t or_mask;
    uint64_t hoff[4];
    int32_t nins;
    uint32_t lane_ctrl;  // controls below QSV_LANE_BITS: a lane takes part iff (lane & lane_ctrl) == lane_ctrl
    int32_t remap;       // R > 1: tiles are walked as R contiguous regions side by side (R = 8: one per XCD)
    int32_t ubit;        // the U work items of a thread are 2^ubit items apart (8 = consecutive 256-item tiles)
    int32_t lbit[2];     // positions of the low target bits (bit j of l)
    int32_t lxor[4];     // lane xor mask of low-bit combination x
    uint32_t pos[QSV_MAX_INS];   // 32-bit on purpose: a dword array in the kernarg segment is indexed with scalar loads
    double m[32];        // matrix in kernel order: row = (h << KL) | l, interleaved complex, row-major
};

// Diagonal gate on up to 2 target bits + controls: every enumerated amplitude is multiplied by
// d[(bit(b0) << 1) | bit(b1)] (b1 < 0: single target, d[bit(b0)]).
struct DiagArgs {
    uint64_t W;
    uint64_t or_mask;
    int32_t nins;
    uint32_t lane_ctrl;
    int32_t b0, b1;
    int32_t remap;
    uint32_t pos[QSV_MAX_INS];   // 32-bit on purpose: a dword array in the kernarg segment is indexed with scalar loads
    double d[8];
};

struct qsv_state {
    int device = 0;
    int kind = 0;            // 0 = qubits, 1 = qudits
    int n = 0;               // qubits (kind 0) or modes (kind 1)
    int d = 2;               // local dimension
    uint64_t amps = 1;       // current number of amplitudes
    uint64_t capacity = 0;   // slots available at `data`
    amp_t *data = nullptr;
    bool owns_data = false;
    amp_t *spare = nullptr;  // second buffer of the out-of-place operations (ping-pong with `data`)
    uint64_t spare_capacity = 0;
    hipStream_t stream = nullptr;
    // workspace
    double *partials = nullptr;       // device: reduction partials (2 doubles per workgroup)
    double *partials_host = nullptr;  // pinned host mirror
    double *dev_matrix = nullptr;     // device: matrix / table of the generic kernels
    size_t dev_matrix_bytes = 0;
    // staging ring for gate matrices and tables (qsvk_stage): pinned host slots mirrored by device slots
    char *stage_host = nullptr;       // QSV_STAGE_SLOTS x QSV_STAGE_BYTES, pinned
    char *stage_dev = nullptr;        // the same on the device
    hipEvent_t stage_done[8] = {};    // kernel that read slot i has finished
    bool stage_busy[8] = {};
    unsigned stage_next = 0;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    std::vector<hipEvent_t> marks;    // lazily created events of qsv_event_record
    // options
    int specialize = 1;
    int unroll = 0;
    int grid_cap = 0;
    int nontemporal = 1;
    int ubit = 8;
    int remap = -1;                   // tile order: -1 = per-kernel default, 0 = plain, R = regions
    int kq_variant = 0;               // k = 3..5 gates: 0 = per-case choice, 1 = wave shuffles (k_dense_big<K, KL>),
                                      // 2 = no transpose (per-thread strided access), 3 = line-granular (k_dense_lds)
    int last_passes = 0;              // k_seq_tile: LDS passes of the last gate list (diagnostics)
    int tile_sequence_gates = -1;     // qsv_apply_sequence: longest gate list applied on LDS tiles (-1 = built-in, 0 = never)
    int sequence_work = -1;           // qsv_apply_sequence: multiply-add limit per 32 amplitudes (-1 = built-in, 0 = never)
    int complex_product = 0;          // complex 5- / 6-qubit blocks: 0 = three real multiplications per entry (3M),
                                      // 4 = the four-multiplication form (measurement variant)
    int readout_variant = 0;          // measurement / insertion / permutation / table diagonals: 0 = streaming forms,
                                      // 1 = round-1 grid-stride forms
    int plane_kernel = 1;             // block-diagonal two-mode operators on the last two modes: 1 = workgroup-per-plane
                                      // form (k_mode2_plane), 0 = plane-per-thread form (k_mode2_blocks<64>)
    char last_kernel[96] = "";        // name of the most recent gate kernel launched (qsv_last_kernel)
};

constexpr int QSV_REDUCE_BLOCKS = 1024;
constexpr int QSV_STAGE_SLOTS = 8;
constexpr size_t QSV_STAGE_BYTES = 512u << 10;

// A gate's matrix / tables on their way to the device without a host-side wait (see qsvk_stage in qsv_kernels.hip).
struct StageRef {
    char *dev = nullptr;   // device address of part A; part B follows at pad16(bytes of A)
    int slot = -1;         // ring slot, or -1: the data went through st->dev_matrix with a synchronous copy
};
int qsvk_stage(qsv_state *st, const void *a, size_t bytes_a, const void *b, size_t bytes_b, StageRef *out);
int qsvk_stage_done(qsv_state *st, const StageRef &ref);   // call right after the launch that reads the slot
inline size_t qsv_pad16(size_t x) { return (x + 15) / 16 * 16; }

// error plumbing (qsv_api.hip)
int qsv_fail(int code, const std::string &msg);
#define QSV_HIP(expr)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess)                                                                           \
            return qsv_fail(QSV_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e));               \
    } while (0)

// launchers (qsv_kernels.hip / qsv_qudit.hip); all enqueue on st->stream
int qsvk_copy(amp_t *dst, const amp_t *src, uint64_t amps, hipStream_t stream);   // nontemporal copy kernel
int qsvk_dense(qsv_state *st, int k, const int *bits, int nctrl, const int *cbits, const double *m_user);
int qsvk_pair_exchange(qsv_state *st, int bit_a, int bit_b);  // SWAP on two high bits (moves 1/2 of the state)
int qsvk_diag(qsv_state *st, int k, const int *bits, int nctrl, const int *cbits, const double *d_user);
int qsvk_phase(qsv_state *st, int nctrl, const int *cbits, double re, double im);
int qsvk_generic(qsv_state *st, int k, const int *bits, const double *m_user);
int qsvk_sequence5(qsv_state *st, const int *bits, int n_gates, const int *arity, const int *legs, const double *mats);
int qsvk_sequence_tile(qsv_state *st, int k, const int *bits, int n_gates, const int *arity, const int *legs, const double *mats);
constexpr int QSV_UNHANDLED_KQ = 1 << 20;  // internal: "not this kernel's case"
int qsvk_measure_probs(qsv_state *st, int bit, const double eig0[4], const double eig1[4], double *p0, double *p1);
int qsvk_collapse(qsv_state *st, int bit, const double eig[4], double scale);
int qsvk_insert(qsv_state *st, int bit, const double amp[4]);
int qsvk_permute(qsv_state *st, const int *src_bit_of_dst_bit);
int qsvk_norm2(qsv_state *st, double *out);
int qsvk_inner(qsv_state *a, qsv_state *b, double *re, double *im);
int qsvk_probabilities(qsv_state *st, const uint64_t *indices, int count, double *out);
int qsvk_expect_pauli(qsv_state *st, uint64_t xmask, uint64_t zmask, int n_y, double *re, double *im);
int qsvk_reduced_density(qsv_state *st, int k, const int *bits, double *rho_out);
int qsvk_expect_density(qsv_state *ket, qsv_state *rho, double *re, double *im);
int qsvk_sample(qsv_state *st, int shots, const double *u, uint64_t *out);
int qsvk_fill_random(qsv_state *st, uint64_t seed, uint64_t index_offset, double *norm2);
int qsvk_scale(qsv_state *st, double re, double im);
int qsvk_set_basis(qsv_state *st, uint64_t index);
int qsvk_ensure_matrix(qsv_state *st, size_t bytes);
int qsvk_scratch(qsv_state *st, uint64_t amps, amp_t **out);  // the state's persistent spare buffer (never freed by callers)
int qsvk_adopt(qsv_state *st, uint64_t new_amps);              // make the spare buffer the register

int qsvq_mode1(qsv_state *st, int mode, const double *m, bool diag);
int qsvq_mode2(qsv_state *st, int mode0, int mode1, const double *m, bool diag);
int qsvq_mode2_gather(qsv_state *st, int mode0, int mode1, int nnz, const int32_t *cols, const double *vals);
int qsvq_mode2_blocks(qsv_state *st, int mode0, int mode1, int nblocks, const int32_t *sizes,
                      const int32_t *plane_indices, const double *mats);
int qsvq_mode_marginal(qsv_state *st, int mode, double *probs);
int qsvq_mode_project(qsv_state *st, int mode, int level, double scale);
int qsvq_mode_insert(qsv_state *st, int mode, const double *vec);
int qsvg_gemm(int device, hipStream_t stream, int op_a, int op_b, uint64_t m, uint64_t n, uint64_t k, const amp_t *a,
              const amp_t *b, amp_t *c);
int qsvg_svd_split(int device, hipStream_t stream, amp_t *theta, uint64_t rows, uint64_t cols, int64_t max_bond_dim,
                   double abs_err, double rel_err, amp_t *m1, amp_t *m2, uint64_t capacity, uint64_t *rank_out,
                   double *s_host);
int qsvg_rsvd_split(int device, hipStream_t stream, const amp_t *theta, uint64_t rows, uint64_t cols, int64_t k_keep,
                    int l, int q, const amp_t *omega, double abs_err, double rel_err, amp_t *m1, amp_t *m2,
                    uint64_t capacity, uint64_t *rank_out, double *s_host);
int qsvg_skinny_gemm(int device, hipStream_t stream, int op, uint64_t n, uint64_t m, int l, const amp_t *A,
                     const amp_t *Q, amp_t *Y);
int qsvg_release_workspace(int device);
int qsvq_tensor_scale_axis(int device, hipStream_t stream, amp_t *t, uint64_t L, uint64_t d, uint64_t R,
                           const double *dev_diag);
int qsvq_tensor_plane_diag(int device, hipStream_t stream, amp_t *t, uint64_t L, uint64_t d, uint64_t R,
                           const double *dev_plane);
int qsvq_tensor_plane_gather(int device, hipStream_t stream, const amp_t *in, amp_t *out, uint64_t L, uint64_t d,
                             uint64_t R, int nnz, const int32_t *dev_cols, const double *dev_vals);
int qsvq_tensor_plane_phase(int device, hipStream_t stream, amp_t *t, uint64_t L, uint64_t d, uint64_t R,
                            const double *dev_qs, double strength);
int qsvq_tensor_plane_affine(int device, hipStream_t stream, const amp_t *in, amp_t *out, uint64_t L, uint64_t d,
                             uint64_t R, const double *dev_qs, const double *a);
int qsvq_tensor_outer(int device, hipStream_t stream, const amp_t *p, const amp_t *q, amp_t *out, uint64_t X, uint64_t Y,
                      uint64_t Z, uint64_t W, int swap_last);
int qsvq_tensor_take_level(int device, hipStream_t stream, const amp_t *in, amp_t *out, uint64_t L, uint64_t d,
                           uint64_t R, uint64_t level, double scale);
int qsvq_tensor_insert_axis(int device, hipStream_t stream, const amp_t *in, amp_t *out, uint64_t L, uint64_t d,
                            uint64_t R, const double *dev_vec);
int qsvq_tensor_axis_overlap(int device, hipStream_t stream, const amp_t *z, const amp_t *t, uint64_t L, uint64_t d,
                             uint64_t R, double *dev_out);
// qsv_gemm.hip: 1 = done by rocBLAS, 0 = unavailable (use the HIP kernels), < 0 = error
int qsvg_axis_gemm(int device, hipStream_t stream, const amp_t *in, amp_t *out, uint64_t L, uint64_t d_in,
                   uint64_t d_out, uint64_t R, const double *dev_m);
int qsvq_tensor_axis_dev(int device, hipStream_t stream, const amp_t *in, amp_t *out, uint64_t L, uint64_t d_in,
                         uint64_t d_out, uint64_t R, const double *dev_m);
int qsvq_tensor_axis(int device, hipStream_t stream, const amp_t *in, amp_t *out, uint64_t L, uint64_t d_in,
                     uint64_t d_out, uint64_t R, const double *m_host);
