/*
 * qsv.h -- C ABI of libqsv.so: MI355X (gfx950) state-vector gate application.
 *
 * This is the drop-in boundary for the gate-application hot path of the reference's
 * simulators/{dv,cv}_simulator.  The reference has no FFI layer: its boundary is the duck-typed
 * Python protocol `gate.apply(state)` (simulators/dv_simulator/simulator.py:47-52,
 * simulators/cv_simulator/simulator.py:66-70).  Each entry point below names the reference
 * code it replaces.  Python binds this header with ctypes (quantum_computations_amd/_lib.py);
 * INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *  - All functions return an int status: QSV_OK (0) or a negative QSV_E* code; the message of
 *    the most recent failure on the calling thread is returned by qsv_last_error().
 *  - Amplitudes are complex128, interleaved (re, im) doubles -- the memory layout of a NumPy
 *    complex128 array.  Matrices are row-major, interleaved complex.
 *  - Qubit numbering is the reference's: qubit q of an n-qubit register is bit (n-1-q) of the
 *    flat amplitude index (qubit 0 = most significant; `X(0)|000>` -> index 4).  For a k-qubit
 *    matrix, qubits[0] is the most significant leg, matching expand_gate's kron(gate, I, ...) +
 *    `targets` order (simulators/dv_simulator/numpy_quantum.py:243-247).
 *  - A qsv_state is owned by the library and used from one host thread at a time; calls enqueue
 *    work on the state's HIP stream and return without waiting unless they return data.
 *    Distinct handles may be used concurrently from distinct threads.
 *  - Host buffers passed in are only read/written during the call.  Randomness stays with the
 *    caller (qsv_measure takes the uniform draw), so seeded runs and forced results reproduce.
 */
#ifndef QSV_H
#define QSV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QSV_VERSION 100 /* 0.1.0 */

enum {
    QSV_OK = 0,
    QSV_EINVAL = -1,  /* bad qubit / duplicate index / bad size -> Python ValueError (gates.py:9-19) */
    QSV_ENOMEM = -2,  /* device allocation failed or view capacity too small -> MemoryError */
    QSV_EHIP = -3,    /* HIP runtime error (no device, launch failure, ...) -> RuntimeError */
    QSV_ESTATE = -4   /* wrong kind of state for this call (qubit vs qudit) -> TypeError */
};

/* Options for qsv_set_option. */
enum {
    QSV_OPT_SPECIALIZE = 1, /* 1 (default): diagonal / controlled / permutation matrices take the
                               reduced-traffic kernels; 0: every gate runs the dense kernel */
    QSV_OPT_UNROLL = 2,     /* work items in flight per thread (1,2,4,8); 0 = built-in default */
    QSV_OPT_GRID_CAP = 3,   /* max workgroups per launch; 0 = one tile per workgroup */
    QSV_OPT_NONTEMPORAL = 4, /* 1 (default): nontemporal loads/stores in the streaming kernels */
    QSV_OPT_ITEM_STRIDE_BIT = 5, /* log2 of the distance (in work items) between the items one thread keeps in
                                    flight when unroll > 1; default 8 = consecutive 4 KiB tiles */
    QSV_OPT_TILE_REGIONS = 6, /* tile order of the streaming kernels: R > 1 walks R contiguous regions of the register
                                side by side (8 = one per XCD), 0 = plain order, -1 (default) = per-kernel choice */
    QSV_OPT_KQ_VARIANT = 7,  /* qsv_apply_kq, k = 3..5, how target bits below 6 reach the registers: 0 (default) =
                                the measured best per case, 1 = wave-shuffle butterflies, 2 = no exchange (per-thread
                                strided access), 3 = line-granular exchange (address arithmetic for bits 3..5, LDS for
                                bits 0..2), 4 = workgroup tile staged through LDS (k = 3..5, every target bit >= 3; other
                                placements as 0), 5 = f64 matrix cores for k = 5 (k = 3, 4 as 0), 6 = f64 matrix cores fed from the LDS
                                tile, matrix in registers, complex k = 5 on bits >= 3 (other cases as 0).  Values 1 and 2 also keep
                                1- and 2-qubit gates on the register kernels (k_dense) instead of the workgroup-tile
                                form (k_dense_tile12).  Same results; for measurements */
    QSV_OPT_PLANE_KERNEL = 8, /* qsv_apply_mode2_blocks on the last two modes with real blocks: 1 (default) = one
                                workgroup per (d x d) plane staged through LDS, 0 = one thread per plane */
    QSV_OPT_READOUT_VARIANT = 9, /* qsv_measure_probs / collapse / insert / permute / k-qubit diagonals: 0 (default) =
                                streaming kernels (whole 1 KiB segments per wave whatever the bit), 1 = plain
                                grid-stride kernels (always used below 14 qubits), 2 = as 0 but reduced density matrices
                                with round 2's per-lane row loads (k_rdm) instead of the LDS-staged workgroup tile
                                (k_rdm_tile).  Same results; for measurements */
    QSV_OPT_COMPLEX_PRODUCT = 10, /* complex 32 x 32 / 64 x 64 blocks (qsv_apply_kq, k = 5, 6): 0 (default) = three real
                                multiplications per matrix entry (Ar xr, Ai xi, (Ar + Ai)(xr + xi)), 4 = four.  Equal
                                to rounding (normwise); for measurements */
    QSV_OPT_SEQUENCE_WORK = 11, /* qsv_apply_sequence: the largest gate sequence applied as a sequence, in multiply-adds
                                per 32 amplitudes (256 per one-qubit gate, 512 per two-qubit gate; the dense block costs
                                4096).  Longer sequences report handled = 0.  -1 (default) = $QSV_SEQUENCE_WORK or 0, 0 = never: the
                                sequence form measured no faster than the dense block; kept for measurements */
    QSV_OPT_TILE_SEQUENCE_GATES = 12 /* qsv_apply_sequence: blocks made of at most this many 1- and 2-qubit gates are
                                applied as that gate list on LDS-resident tiles (k_seq_tile) instead of their dense
                                product.  -1 (default) = 6-qubit blocks of at most 12 gates ($QSV_TILE_SEQUENCE_GATES);
                                an explicit value also admits 5-qubit blocks (slower than their dense product: for
                                measurements); 0 = never */
};

typedef struct qsv_state qsv_state;

/* ---- library ---------------------------------------------------------------------------- */
int qsv_version(void);
const char *qsv_last_error(void);
int qsv_device_count(int *count);

/* ---- lifecycle -------------------------------------------------------------------------- */
/* Allocate an n-qubit register on `device` initialised to |0...0> (n = 0 is the 1-element
 * register [1.0] that parse_state(None) returns, simulators/dv_simulator/simulator.py:22). */
int qsv_create(int n_qubits, int device, qsv_state **out);
/* Wrap caller-owned device memory (e.g. a torch tensor's data_ptr()) holding `capacity_amps`
 * complex128 slots; `hip_stream` is a hipStream_t (NULL = the default stream).  The register's
 * amplitudes always start at dev_amps; measure/insert keep them there. */
int qsv_create_view(int n_qubits, int device, void *dev_amps, uint64_t capacity_amps, void *hip_stream,
                    qsv_state **out);
/* Re-point a view register (one made by qsv_create_view) at another window of caller-owned device memory, with a new
 * size: no allocation, no synchronisation -- the register's stream orders the launches on either side.  This is how the
 * sharded register applies a gate slice by slice inside an exchange step (one handle walks the landed slices), the
 * part of the reference's `for gate in circuit: state = gate.apply(state)` loop (dv_simulator/simulator.py:40-52) that
 * overlaps with the transfer.  QSV_ESTATE on a register that owns its memory. */
int qsv_rebind_view(qsv_state *st, int n_qubits, void *dev_amps, uint64_t capacity_amps);
int qsv_destroy(qsv_state *st);
int qsv_set_stream(qsv_state *st, void *hip_stream);
int qsv_set_option(qsv_state *st, int option, int64_t value);
int qsv_num_qubits(const qsv_state *st, int *n_qubits);
int qsv_num_amps(const qsv_state *st, uint64_t *n_amps);
int qsv_device_ptr(qsv_state *st, void **dev_amps);
int qsv_sync(qsv_state *st);

/* ---- data movement ---------------------------------------------------------------------- */
int qsv_set_basis(qsv_state *st, uint64_t index);
int qsv_upload(qsv_state *st, const double *host_interleaved, uint64_t offset_amps, uint64_t count_amps);
int qsv_download(qsv_state *st, double *host_interleaved, uint64_t offset_amps, uint64_t count_amps);
int qsv_copy(qsv_state *dst, const qsv_state *src);
/* Fill with pseudo-random complex normal amplitudes keyed by (seed, global amplitude index +
 * index_offset) and leave the squared norm of what was written in *norm2 (may be NULL);
 * qsv_scale() then normalises.  Used for states too large to come from the host. */
int qsv_fill_random(qsv_state *st, uint64_t seed, uint64_t index_offset, double *norm2);
int qsv_scale(qsv_state *st, double re, double im);

/* ---- gate application: replaces Gate.apply -> expand_gate -> `gate @ state`
 *      (simulators/dv_simulator/gates.py:44-54, numpy_quantum.py:243-247) ------------------- */
int qsv_apply_1q(qsv_state *st, int q, const double m[8]);
int qsv_apply_2q(qsv_state *st, int q0, int q1, const double m[32]);
/* Diagonal gates (Z, RZ, P, Pdg, T, Tdg, CZ: gates.py:79-114,128-130): d = the diagonal. */
int qsv_apply_diag_1q(qsv_state *st, int q, const double d[4]);
int qsv_apply_diag_2q(qsv_state *st, int q0, int q1, const double d[8]);
/* CX (gates.py:116-126) and SWAP (gates.py:132-134) as pure amplitude moves. */
int qsv_apply_cx(qsv_state *st, int control, int target);
int qsv_apply_swap(qsv_state *st, int q0, int q1);
/* `m` (2x2) on `target` for amplitudes whose `controls` are all 1 (add_control, numpy_quantum.py:250). */
int qsv_apply_controlled_1q(qsv_state *st, int n_controls, const int *controls, int target, const double m[8]);
/* Multiply the amplitudes whose `qubits` are all 1 by (re, im): multi-controlled phase / Z. */
int qsv_apply_mcphase(qsv_state *st, int n_qubits, const int *qubits, double re, double im);
/* Generic k-qubit matrix (2^k x 2^k), 1 <= k <= 6: Gate(indices, matrix).apply (gates.py:7-54). */
int qsv_apply_kq(qsv_state *st, int k, const int *qubits, const double *m);
/* A fused block given as the SEQUENCE of the gates it was made of instead of their product: the same state as
 * qsv_apply_kq with the product matrix (to rounding), i.e. consecutive iterations of the reference's loop
 * `for gate in circuit: state = gate.apply(state)` (dv_simulator/simulator.py:40-52) in ONE pass over the register.
 * Gate g acts on the block's legs legs[2 g] (and legs[2 g + 1] if arity[g] == 2) -- positions in `qubits` -- with the
 * 2 x 2 / 4 x 4 row-major complex matrix that follows the previous gate's in `matrices`.  A dense 5-qubit block is the one
 * gate shape bound by arithmetic (4 x 1024 FMAs per amplitude group); its few source gates cost a fraction of that on the
 * amplitudes the thread already holds -- but measured no faster (two waves per SIMD; profiles/r03_sequence_blocks.txt), so
 * the form is OFF unless QSV_OPT_SEQUENCE_WORK / $QSV_SEQUENCE_WORK allow it.  *handled = 0: no sequence form for this block
 * / register (switched off, over the work limit, k != 5, gates on more than two qubits, tiny registers) -- nothing was
 * applied, call qsv_apply_kq with the product matrix. */
int qsv_apply_sequence(qsv_state *st, int k, const int *qubits, int n_gates, const int *arity, const int *legs,
                       const double *matrices, int *handled);
/* Qubit-axis permutation of the ket: permute_tensor_product (numpy_quantum.py:227-240); the qubit
 * at position j moves to position new_ordering[j]. */
int qsv_permute(qsv_state *st, const int *new_ordering);

/* ---- measurement / insertion: replace M.apply (gates.py:165-186), Insert.apply (:145-153) - */
/* Projects qubit q on eig0 / eig1 (each 2 complex numbers, used UNCONJUGATED exactly as the
 * reference does), leaves p0 = |res0|^2, p1 = |res1|^2, picks outcome = forced if forced is 0/1,
 * else (u01 < p0/(p0+p1) ? 0 : 1), and shrinks the register to the normalised (n-1)-qubit ket. */
int qsv_measure(qsv_state *st, int q, const double eig0[4], const double eig1[4], int forced, double u01,
                int *outcome, double *p0, double *p1);
/* The two halves of qsv_measure, for callers that draw the outcome themselves between them (the Python
 * layer calls np.random.choice exactly as gates.py:183 does): the reduction without collapsing, and the
 * collapse onto `eig` with the result multiplied by `scale` (1/norm of the chosen branch). */
int qsv_measure_probs(qsv_state *st, int q, const double eig0[4], const double eig1[4], double *p0, double *p1);
int qsv_collapse(qsv_state *st, int q, const double eig[4], double scale);
/* Grows the register: new qubit with amplitudes amp[0..1] (complex) at position q. */
int qsv_insert(qsv_state *st, int q, const double amp[4]);

/* ---- read-out (npq.norm, numpy_quantum.py:131-132; |amp|^2 of chosen indices) -------------- */
int qsv_norm2(qsv_state *st, double *out);
int qsv_probabilities(qsv_state *st, const uint64_t *indices, int count, double *out);
/* <a|b> with a conjugated: the ket-ket branch of npq.fidelity (numpy_quantum.py:151-152). */
int qsv_inner(qsv_state *a, qsv_state *b, double *re, double *im);
/* <psi| P |psi> for the Pauli string P = paulis[0] on qubits[0] (x) paulis[1] on qubits[1] ... ('I','X','Y','Z'):
 * npq.expect (numpy_quantum.py:194-201) for tensor products of npq.PAULIS without building the 2^N operator. */
int qsv_expect_pauli(qsv_state *st, int k, const int *qubits, const char *paulis, double *re, double *im);
/* Reduced density matrix of the k <= 6 qubits `qubits` (all others traced out) in one read pass over the register:
 * rho[i][j] = sum_rest psi[i, rest] conj(psi[j, rest]), written row-major as 4^k complex numbers, qubits[0] the most
 * significant bit of i and j.  What a caller of the reference gets from npq.ket2dm (numpy_quantum.py:110-113)
 * followed by a partial trace -- without the 2^N x 2^N matrix. */
int qsv_reduced_density(qsv_state *st, int k, const int *qubits, double *rho);
/* <a| rho |a> for an n-qubit ket `ket` and a density matrix `rho` held row-major as a 2n-qubit register (the layout
 * Gate.apply uses for U rho U^dagger): the ket / density-matrix branches of npq.fidelity (numpy_quantum.py:153-156).
 * npq.purity (numpy_quantum.py:164-166) of such a register is qsv_norm2: tr(rho rho) = sum |rho_ij|^2 for a
 * hermitian rho. */
int qsv_expect_density(qsv_state *ket, qsv_state *rho, double *re, double *im);
/* Draw `shots` computational-basis outcomes from |amp|^2 by inverse-CDF sampling: out[s] is the smallest basis
 * index whose cumulative probability exceeds u[s] * norm^2 (u[s] in [0, 1), drawn by the caller).  The register
 * is not collapsed.  Equivalent to measuring every qubit with MZ (gates.py:188-190) on independent copies. */
int qsv_sample(qsv_state *st, int shots, const double *u, uint64_t *out);

/* ---- d-level mode registers: the "d x d (d^2 x d^2) operator along mode axes" contraction of
 *      cv_simulator (np.tensordot at simulators/cv_simulator/utils.py:15,37; gates.py:73,160) --- */
int qsv_create_qudit(int n_modes, int d, int device, qsv_state **out);
int qsv_create_qudit_view(int n_modes, int d, int device, void *dev_amps, uint64_t capacity_amps,
                          void *hip_stream, qsv_state **out);
int qsv_qudit_shape(const qsv_state *st, int *n_modes, int *d);
int qsv_apply_mode1(qsv_state *st, int mode, const double *m /* d x d */);
int qsv_apply_mode1_diag(qsv_state *st, int mode, const double *diag /* d */);
int qsv_apply_mode2(qsv_state *st, int mode0, int mode1, const double *m /* d^2 x d^2 */);
int qsv_apply_mode2_diag(qsv_state *st, int mode0, int mode1, const double *diag /* d^2 */);
/* Sparse two-mode map: output plane point (i0, i1) = sum_k vals[(i0*d+i1)*nnz + k] * input plane point
 * cols[(i0*d+i1)*nnz + k] (= j0*d+j1; negative = unused slot).  The bilinear resampling of the (q1, q2)
 * plane that BS and CX do per bond pair with RegularGridInterpolator (cv_simulator/gates.py:74-80,187-189)
 * is nnz = 4; SWAP (gates.py:48-55) is nnz = 1. */
int qsv_apply_mode2_gather(qsv_state *st, int mode0, int mode1, int nnz, const int32_t *cols, const double *vals);
/* Block-diagonal two-mode operator, in place: `nblocks` disjoint sets of (d, d)-plane points, set k holding
 * sizes[k] (<= 32) points listed in plane_indices (= j0*d+j1 in (mode0, mode1) order, concatenated), mixed by
 * the dense sizes[k] x sizes[k] complex matrix found at the matching position of `mats` (row-major, concatenated).
 * Plane points in no block are left alone.  A Fock-basis beam splitter (it conserves n_a + n_b) is 2d-1 blocks. */
int qsv_apply_mode2_blocks(qsv_state *st, int mode0, int mode1, int nblocks, const int32_t *sizes,
                           const int32_t *plane_indices, const double *mats);
/* Homodyne read-out (Mq.apply, cv_simulator/gates.py:90-117): probs[j] = sum over the other modes of
 * |amp|^2 at level j of `mode` (the diagonal of partial_density_mps, mps.py:176-190, without the dq factors);
 * project keeps level `level` of `mode`, multiplies by `scale` and removes the mode. */
int qsv_mode_marginal(qsv_state *st, int mode, double *probs /* d */);
int qsv_mode_project(qsv_state *st, int mode, int level, double scale);
/* New mode with amplitudes vec[0..d) (complex) at position `mode` (Insert.apply, gates.py:24-45). */
int qsv_mode_insert(qsv_state *st, int mode, const double *vec /* d */);
/* out[l, :, r] = M @ in[l, :, r] on a raw (L, d_in, R) device tensor -> (L, d_out, R): the exact
 * tensordot+moveaxis of whittaker_shannon / rotation (cv_simulator/utils.py:9-39) on an MPS site. */
int qsv_tensor_apply_axis(int device, void *hip_stream, const void *dev_in, void *dev_out, uint64_t L,
                          uint64_t d_in, uint64_t d_out, uint64_t R, const double *m /* d_out x d_in */);
/* Same with the operator already in device memory (the reference's gates build their d x d matrix once in
 * __init__, cv_simulator/gates.py:48-52,61-66, and reuse it for every apply): nothing is uploaded and the call
 * is asynchronous on `hip_stream`.  Grids of >= 64 points go to rocBLAS zgemm when it can be loaded. */
int qsv_tensor_apply_axis_dev(int device, void *hip_stream, const void *dev_in, void *dev_out, uint64_t L,
                              uint64_t d_in, uint64_t d_out, uint64_t R, const void *dev_m /* d_out x d_in */);

/* ---- matrix-product-state sites (cv_simulator/mps.py:102-201) ------------------------------
 * The reference keeps a CV register as a list of (chi_l, d, chi_r) site tensors and, for every two-mode gate,
 * contracts two neighbours, maps the (q_left, q_right) plane and splits the result again with a truncated SVD
 * (cv_simulator/gates.py:48-84,151-192).  These entry points are that update on raw device tensors: every pointer
 * is device memory (row-major complex128 unless noted), every call runs on `hip_stream` of `device`; calls that
 * return data to the host synchronise the stream.  GEMM and SVD come from rocBLAS / rocSOLVER (bound by dlopen);
 * without them these calls return QSV_EHIP. */
/* C (m x n) = op(A) . op(B); op: 0 = as stored, 1 = transpose, 2 = conjugate transpose; A is (m x k) or, with an
 * op, (k x m); likewise B.  np.tensordot(m1, m2, (2, 0)) (gates.py:68) and the environment recursions of
 * MPS.norm / partial_density_mps (mps.py:166-190). */
int qsv_tensor_gemm(int device, void *hip_stream, int op_a, int op_b, uint64_t m, uint64_t n, uint64_t k,
                    const void *dev_a, const void *dev_b, void *dev_c);
/* tensor_svd (mps.py:52-97) of the (rows x cols) matrix `dev_theta` (destroyed): m1 = U[:, :r] sqrt(S[:r]) as
 * (rows x r), m2 = sqrt(S[:r]) Vh[:r, :] as (r x cols), r chosen by the reference's rule -- drop the longest tail of
 * singular values whose sum stays <= max(abs_err, rel_err * sum(S)), then cap at max_bond_dim (< 0 = no cap).
 * `capacity` = the r the output buffers can hold; *rank = r; singular_values (host, min(rows, cols) doubles) may be
 * NULL. */
int qsv_tensor_svd_split(int device, void *hip_stream, void *dev_theta, uint64_t rows, uint64_t cols,
                         int64_t max_bond_dim, double abs_err, double rel_err, void *dev_m1, void *dev_m2,
                         uint64_t capacity, uint64_t *rank, double *singular_values);
/* The same split on the reference's randomized branch (taken when max_bond_dim * 10 < min(rows, cols), mps.py:78):
 * Halko-Martinsson-Tropp range finder with `probes` = max_bond_dim + 10 Gaussian test vectors and
 * `power_iterations` passes (mps.py:5-50), SVD of the small projection, first max_bond_dim triplets, then the
 * truncation rule above.  `dev_omega` holds the test matrix the reference would draw,
 * rng.normal(size=(min(rows, cols), probes)), as complex128 in COLUMN-major order, so that a seeded run consumes
 * the same random stream and lands on the same subspace.  `dev_theta` is not modified.
 * `dev_omega` may be NULL on a first call: converting and uploading 10^5 normal deviates costs more than a split that
 * never reads them (under a loose tolerance a verified low-rank route with fixed probes decides most splits).  A caller
 * whose generator is shared with the rest of the simulation still has to DRAW them, as mps.py:14-15 does -- it can do so
 * while this call runs.  *rank = QSV_RANK_NEEDS_OMEGA then means nothing was computed: call again with the test matrix. */
#define QSV_RANK_NEEDS_OMEGA UINT64_MAX
int qsv_tensor_rsvd_split(int device, void *hip_stream, const void *dev_theta, uint64_t rows, uint64_t cols,
                          int64_t max_bond_dim, int probes, int power_iterations, const void *dev_omega,
                          double abs_err, double rel_err, void *dev_m1, void *dev_m2, uint64_t capacity,
                          uint64_t *rank, double *singular_values /* max_bond_dim doubles or NULL */);
/* Tall-skinny product on the f64 matrix cores, COLUMN-major with tight leading dimensions: Y = op(A) . Q with A (n x m)
 * and a panel of 1 <= l <= 256 columns (one pass over A per 64 of them); op: 0 = A, 3 = conj(A) (Q is m x l, Y is n x l); 1 = A^H, 2 = A^T (Q is n x l,
 * Y is m x l).  The building block of the range finder (A @ O, A^H @ Q, A @ Q of mps.py:15-21), where A is gigabytes
 * and the panel a few dozen columns: l is tiled in steps of 16 instead of the library's 64-wide macro tile, and the
 * transposed / conjugated forms let a row-major theta be used in either orientation without a re-ordered copy. */
int qsv_tensor_skinny_gemm(int device, void *hip_stream, int op, uint64_t n, uint64_t m, int l, const void *dev_a,
                           const void *dev_q, void *dev_y);
/* The splits keep their scratch memory (a copy-free pass needs panels only, the exact SVD its factors) in a grow-only
 * pool per device; this returns the pool of `device` to the driver.  It is re-grown on demand. */
int qsv_tensor_release_workspace(int device);
/* t[l, j, r] *= diag[j] in place: Z and P on a site (gates.py:223,245). */
int qsv_tensor_scale_axis(int device, void *hip_stream, void *dev_t, uint64_t L, uint64_t d, uint64_t R,
                          const void *dev_diag /* d */);
/* theta[a, j, l, b] *= plane[j, l] in place: the CZ phases on a two-site tensor (gates.py:159-160). */
int qsv_tensor_plane_diag(int device, void *hip_stream, void *dev_theta, uint64_t L, uint64_t d, uint64_t R,
                          const void *dev_plane /* d x d */);
/* out[a, p, b] = sum_e vals[p, e] in[a, cols[p, e], b] over the d*d plane points p: the bilinear (q1, q2)-plane
 * resampling of BS / CX for every bond pair at once (gates.py:74-80,187-189); cols < 0 are padding. */
int qsv_tensor_plane_gather(int device, void *hip_stream, const void *dev_in, void *dev_out, uint64_t L, uint64_t d,
                            uint64_t R, int per_point, const int32_t *dev_cols, const void *dev_vals);
/* The same two maps with nothing tabulated: theta[a, j, l, b] *= exp(i strength q_j q_l) (CZ), and
 * out[a, i0, i1, b] = bilinear interpolation of in[a, :, :, b] at (a00 q_i0 + a01 q_i1, a10 q_i0 + a11 q_i1), zero
 * outside the grid (BS: a rotation; CX: a shear).  `dev_grid` = the d grid points (doubles, ascending). */
int qsv_tensor_plane_phase(int device, void *hip_stream, void *dev_theta, uint64_t L, uint64_t d, uint64_t R,
                           const void *dev_grid, double strength);
int qsv_tensor_plane_affine(int device, void *hip_stream, const void *dev_in, void *dev_out, uint64_t L, uint64_t d,
                            uint64_t R, const void *dev_grid, const double *a /* a00 a01 a10 a11, host */);
/* out[l, r] = scale * in[l, level, r]: the site after a homodyne outcome (gates.py:108). */
int qsv_tensor_take_level(int device, void *hip_stream, const void *dev_in, void *dev_out, uint64_t L, uint64_t d,
                          uint64_t R, uint64_t level, double scale);
/* out[l, j, r] = vec[j] * in[l, r]: np.einsum("i,ajb -> aijb") of Insert.apply (gates.py:39). */
int qsv_tensor_insert_axis(int device, void *hip_stream, const void *dev_in, void *dev_out, uint64_t L, uint64_t d,
                           uint64_t R, const void *dev_vec /* d */);
/* out[x, y, z, w] = p[x, z] * q[y, w] (out[x, y, w, z] when swap_last != 0), p is (X x Z), q is (Y x W): the outer
 * products that attach the two halves of a GKP Bell pair to their neighbouring sites with the bond legs adjacent
 * (InsertBell.apply, gkp_simulator/insert_bell.py:80-90), ready for qsv_tensor_svd_split. */
int qsv_tensor_outer(int device, void *hip_stream, const void *dev_p, const void *dev_q, void *dev_out, uint64_t X,
                     uint64_t Y, uint64_t Z, uint64_t W, int swap_last);
/* out[j] = Re sum_{l, r} z[l, j, r] conj(t[l, j, r]) (d doubles in device memory): the diagonal of the reduced
 * density matrix once both environments are contracted into z (mps.py:188-189). */
int qsv_tensor_axis_overlap(int device, void *hip_stream, const void *dev_z, const void *dev_t, uint64_t L, uint64_t d,
                            uint64_t R, void *dev_out);

/* ---- whole circuits in one launch (registers of at most 13 qubits) ------------------------------------------------
 * Replaces the caller loop itself -- `for gate in self.circuit: ... gate.apply(state)` with its measurement record and
 * ClassicalControl (dv_simulator/simulator.py:40-52, :6-17) -- for the register sizes the reference can run (4..12
 * qubits) and the batches of circuits its drivers push through a multiprocessing.Pool
 * (impact_.../randomised_benchmarking.py:60-76, average_clifford_fidelity.py:212): one workgroup per instance keeps the
 * register in LDS and walks its gate list; `count` instances run side by side on the CUs.  Everything is host memory,
 * the call returns when the results are in place.
 *   programs / prog_offsets[count + 1]: the instances' gate lists as 64-bit words (format: csrc/qsv_circuit.hip;
 *       built by quantum_computations_amd.dv_simulator.program), word offsets per instance.
 *   n_initial[count]; states_in + state_offsets[count + 1]: initial kets (interleaved complex128), amplitude offsets.
 *   states_out + out_offsets[count + 1]: final kets (their sizes follow from the programs: M removes, Insert adds a qubit).
 *   results / probabilities + result_offsets[count + 1]: outcome and the two branch norms (p0, p1) of every
 *       measurement, in program order.  A measurement op carries either a forced outcome (gates.py:183 `result=`) or
 *       the uniform number the host drew for it; outcome 0 iff u < p0 / (p0 + p1), as np.random.choice picks it.
 *   max_qubits: the largest register any instance reaches (<= 13). */
int qsv_run_programs(int device, int count, int max_qubits, const uint64_t *programs, const uint64_t *prog_offsets,
                     const int *n_initial, const double *states_in, const uint64_t *state_offsets, double *states_out,
                     const uint64_t *out_offsets, int *results, double *probabilities, const uint64_t *result_offsets);

/* ---- timing on the state's stream (HIP events), for bench.py's roofline figures ----------- */
int qsv_timer_start(qsv_state *st);
int qsv_timer_stop(qsv_state *st, float *elapsed_ms); /* records, synchronises the event, returns ms */
/* Name of the gate kernel the most recent qsv_apply_* call launched on this state, spelled as rocprofv3
 * prints it (e.g. "k_dense<1, 0, 1, true>"); "" if none.  Lets bench.py attribute its event timings. */
int qsv_last_kernel(const qsv_state *st, char *buf, size_t buf_len);
/* Non-blocking marks: record HIP event number `slot` (0 <= slot < 16384) on the state's stream;
 * qsv_event_elapsed_ms waits for mark `slot_b` and returns the device time between two marks. */
int qsv_event_record(qsv_state *st, int slot);
int qsv_event_elapsed_ms(qsv_state *st, int slot_a, int slot_b, float *elapsed_ms);

#ifdef __cplusplus
}
#endif
#endif /* QSV_H */
