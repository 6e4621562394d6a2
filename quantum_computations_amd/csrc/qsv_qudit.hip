// d-level mode kernels of libqsv.so (gfx950): apply a d x d (or d^2 x d^2) operator along the mode axes of a
// dense register viewed as (L, d, R) -- the np.tensordot(M, T, [1, axis]) + moveaxis contraction the reference's
// cv_simulator performs on every MPS site (simulators/cv_simulator/utils.py:15,37; gates.py:73,160,222,246).
//
// Mode 0 is the most significant (slowest) axis, like qubit 0 of the qubit registers.

#include "qsv_internal.h"

#include <algorithm>
#include <cstring>
#include <vector>

namespace {

struct cplx {
    double re, im;
};

__device__ __forceinline__ amp_t cmul(cplx m, amp_t a) {
    return amp_t{m.re * a.x - m.im * a.y, m.re * a.y + m.im * a.x};
}

__device__ __forceinline__ amp_t cfma(cplx m, amp_t a, amp_t acc) {
    return amp_t{fma(m.re, a.x, fma(-m.im, a.y, acc.x)), fma(m.re, a.y, fma(m.im, a.x, acc.y))};
}

// ----------------------------------------------------------------------------------------------------
// out[l, i, r] = sum_j M[i, j] in[l, j, r]   (L, d_in, R) -> (L, d_out, R)
//
// A workgroup owns a tile of TR consecutive r values of one l: the (d_in x TR) input tile is staged in LDS
// with fully coalesced loads (TR * 16 B contiguous per row), then every wave produces output rows
// i = wave, wave + 4, ...  four at a time, so each LDS read feeds four complex FMAs.  M is read with
// wave-uniform addresses (scalar loads, served by the scalar cache / L2).
// ----------------------------------------------------------------------------------------------------
constexpr int TR = 64;      // r values per tile = one wave-width
constexpr int ROWS = 4;     // output rows per wave per pass

__global__ __launch_bounds__(QSV_BLOCK) void k_axis_tile(const amp_t *__restrict__ in, amp_t *__restrict__ out,
                                                        uint64_t L, int d_in, int d_out, uint64_t R,
                                                        uint64_t r_tiles, const double *__restrict__ M) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    amp_t *tile = reinterpret_cast<amp_t *>(smem_raw);  // [d_in][TR]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t total = L * r_tiles;
    for (uint64_t t = blockIdx.x; t < total; t += gridDim.x) {
        const uint64_t l = t / r_tiles, r0 = (t % r_tiles) * TR;
        const bool r_ok = r0 + lane < R;
        __syncthreads();  // previous tile fully consumed
        for (int j = wave; j < d_in; j += QSV_BLOCK / 64)
            tile[j * TR + lane] = r_ok ? in[(l * d_in + j) * R + r0 + lane] : amp_t{0.0, 0.0};
        __syncthreads();
        for (int i0 = wave * ROWS; i0 < d_out; i0 += (QSV_BLOCK / 64) * ROWS) {
            amp_t acc[ROWS];
#pragma unroll
            for (int k = 0; k < ROWS; ++k) acc[k] = amp_t{0.0, 0.0};
            for (int j = 0; j < d_in; ++j) {
                const amp_t x = tile[j * TR + lane];
#pragma unroll
                for (int k = 0; k < ROWS; ++k) {
                    const int i = min(i0 + k, d_out - 1);
                    const cplx m = {M[2 * (static_cast<size_t>(i) * d_in + j)],
                                    M[2 * (static_cast<size_t>(i) * d_in + j) + 1]};
                    acc[k] = cfma(m, x, acc[k]);
                }
            }
#pragma unroll
            for (int k = 0; k < ROWS; ++k)
                if (i0 + k < d_out && r_ok) out[(l * d_out + i0 + k) * R + r0 + lane] = acc[k];
        }
    }
}

// Small-R case (the mode is the last or a late axis: fibres are short contiguous runs).  One thread per
// output element, inputs re-read through the caches.  Correct for every shape; used when R < TR.
__global__ __launch_bounds__(QSV_BLOCK) void k_axis_simple(const amp_t *__restrict__ in, amp_t *__restrict__ out,
                                                          uint64_t L, int d_in, int d_out, uint64_t R,
                                                          const double *__restrict__ M) {
    const uint64_t total = L * d_out * R;
    for (uint64_t o = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x; o < total;
         o += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        const uint64_t r = o % R, i = (o / R) % d_out, l = o / (R * d_out);
        amp_t acc = {0.0, 0.0};
        for (int j = 0; j < d_in; ++j) {
            const cplx m = {M[2 * (i * d_in + j)], M[2 * (i * d_in + j) + 1]};
            acc = cfma(m, in[(l * d_in + j) * R + r], acc);
        }
        out[o] = acc;
    }
}

// Thin fibres (R <= 8: the sites of a matrix-product state with bonds of 1-2, d = 1000 -- every single-mode gate of the
// GKP runs): a WAVE per output row (l, i), lanes striding over j, so the operator row is read as whole 1 KiB segments
// (k_axis_simple reads it one 16-byte entry per thread, rows apart: 224 us for a 1000 x 1000 operator on a (1, 1000, 2)
// site, 16 MB that HBM delivers in a few microseconds).  Per-lane partial sums, then a shuffle tree.
__global__ __launch_bounds__(QSV_BLOCK) void k_axis_rows(const amp_t *__restrict__ in, amp_t *__restrict__ out, uint64_t L,
                                                        int d_in, int d_out, int R, const double *__restrict__ M) {
    constexpr int RMAX = 8;
    const uint64_t row = static_cast<uint64_t>(blockIdx.x) * (QSV_BLOCK / 64) + (threadIdx.x >> 6);
    if (row >= L * static_cast<uint64_t>(d_out)) return;      // whole waves leave together
    const int lane = threadIdx.x & 63;
    const uint64_t l = row / d_out, i = row % d_out;
    amp_t acc[RMAX];
#pragma unroll
    for (int r = 0; r < RMAX; ++r) acc[r] = amp_t{0.0, 0.0};
    const amp_t *mrow = reinterpret_cast<const amp_t *>(M) + i * static_cast<uint64_t>(d_in);
    for (int j = lane; j < d_in; j += 64) {
        const amp_t mv = mrow[j];
        const cplx m = {mv.x, mv.y};
        const amp_t *x = in + (l * d_in + j) * static_cast<uint64_t>(R);
#pragma unroll
        for (int r = 0; r < RMAX; ++r)
            if (r < R) acc[r] = cfma(m, x[r], acc[r]);
    }
#pragma unroll
    for (int r = 0; r < RMAX; ++r)
        if (r < R) {
            double re = acc[r].x, im = acc[r].y;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                re += __shfl_xor(re, o, 64);
                im += __shfl_xor(im, o, 64);
            }
            if (lane == 0) out[row * R + r] = amp_t{re, im};
        }
}

// in[l, j, r] *= diag[j]
__global__ __launch_bounds__(QSV_BLOCK) void k_axis_diag(amp_t *__restrict__ a, uint64_t total, int d, uint64_t R,
                                                        const double *__restrict__ diag) {
    for (uint64_t o = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x; o < total;
         o += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        const uint64_t j = (o / R) % d;
        a[o] = cmul(cplx{diag[2 * j], diag[2 * j + 1]}, a[o]);
    }
}

// Two-mode dense: view (L, d, Mid, d, R); out[l,i0,m,i1,r] = sum_{j0,j1} G[(i0,i1),(j0,j1)] in[l,j0,m,j1,r].
__global__ __launch_bounds__(QSV_BLOCK) void k_mode2_simple(const amp_t *__restrict__ in, amp_t *__restrict__ out,
                                                           uint64_t L, int d, uint64_t Mid, uint64_t R,
                                                           const double *__restrict__ G) {
    const uint64_t total = L * d * Mid * d * R;
    const uint64_t s1 = R, sm = R * d, s0 = R * d * Mid, sl = R * d * Mid * d;
    const int d2 = d * d;
    for (uint64_t o = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x; o < total;
         o += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        const uint64_t r = o % R, i1 = (o / s1) % d, m = (o / sm) % Mid, i0 = (o / s0) % d, l = o / sl;
        const uint64_t base = l * sl + m * sm + r;
        const size_t row = (i0 * d + i1) * static_cast<size_t>(d2);
        amp_t acc = {0.0, 0.0};
        for (int j0 = 0; j0 < d; ++j0)
            for (int j1 = 0; j1 < d; ++j1) {
                const cplx g = {G[2 * (row + j0 * d + j1)], G[2 * (row + j0 * d + j1) + 1]};
                acc = cfma(g, in[base + j0 * s0 + j1 * s1], acc);
            }
        out[o] = acc;
    }
}

__global__ __launch_bounds__(QSV_BLOCK) void k_mode2_diag(amp_t *__restrict__ a, uint64_t total, int d, uint64_t Mid,
                                                         uint64_t R, const double *__restrict__ diag) {
    const uint64_t s1 = R, s0 = R * d * Mid;
    for (uint64_t o = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x; o < total;
         o += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        const uint64_t i1 = (o / s1) % d, i0 = (o / s0) % d;
        const uint64_t k = i0 * d + i1;
        a[o] = cmul(cplx{diag[2 * k], diag[2 * k + 1]}, a[o]);
    }
}

// Two-mode sparse row operator ("gather"): every output plane point (i0, i1) is a weighted sum of `nnz` input plane
// points.  This is the MI355X form of the reference's per-bond-pair RegularGridInterpolator loop for BS / CX
// (bilinear resampling of the (q1, q2) plane: 4 weights per point, cv_simulator/gates.py:74-80,187-189), and of
// SWAP (nnz = 1).  View (L, d, Mid, d, R); cols/vals are indexed [(i0 * d + i1) * nnz + k], col = j0 * d + j1.
__global__ __launch_bounds__(QSV_BLOCK) void k_mode2_gather(const amp_t *__restrict__ in, amp_t *__restrict__ out,
                                                           uint64_t L, int d, uint64_t Mid, uint64_t R, int nnz,
                                                           const int32_t *__restrict__ cols,
                                                           const double *__restrict__ vals) {
    const uint64_t total = L * d * Mid * d * R;
    const uint64_t s1 = R, sm = R * d, s0 = R * d * Mid, sl = R * d * Mid * d;
    for (uint64_t o = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x; o < total;
         o += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        const uint64_t r = o % R, i1 = (o / s1) % d, m = (o / sm) % Mid, i0 = (o / s0) % d, l = o / sl;
        const uint64_t base = l * sl + m * sm + r;
        const size_t row = (i0 * d + i1) * static_cast<size_t>(nnz);
        amp_t acc = {0.0, 0.0};
        for (int k = 0; k < nnz; ++k) {
            const int32_t c = cols[row + k];
            if (c < 0) continue;  // padding entry
            const cplx w = {vals[2 * (row + k)], vals[2 * (row + k) + 1]};
            acc = cfma(w, in[base + (c / d) * s0 + (c % d) * s1], acc);
        }
        out[o] = acc;
    }
}

// Block-diagonal two-mode operator, in place: the (d x d) plane splits into `nblocks` disjoint index sets, each mixed
// by its own small dense matrix (a beam splitter conserves n_a + n_b: its blocks are the anti-diagonals of the Fock
// plane, sizes 1..d).  A thread owns one plane (all other indices fixed; consecutive lanes = consecutive memory),
// walks the blocks, holds a block's <= 32 amplitudes in registers and reads the block matrix and the plane offsets
// with wave-uniform (scalar) loads.  sum_b s_b^2 complex FMAs per plane instead of d^4 for the dense operator.
constexpr int MAX_BLOCK = 32;

// One work item = (plane, block): a thread loads the block's s amplitudes of its plane, multiplies by the block matrix
// (wave-uniform scalar loads: the block index is the same for the whole workgroup) and stores them back.  Splitting the
// planes' 2d-1 blocks over separate workgroups instead of walking them in one thread gives 2d-1 times more waves in
// flight and short dependency chains.  Workgroup order: all blocks of one group of planes are dealt to the same XCD
// back to back, so when the plane group is not contiguous in memory (R < 8: neighbouring lanes are up to 16 KiB apart and
// every 128-byte line is shared by several anti-diagonals) the re-touched lines are served by that XCD's L2.
// Blocks come in pairs (`pairs`: a large block with a small one, at most MAX_BLOCK elements together, second = -1 if
// alone) that one workgroup handles one after the other: a beam splitter's 63 anti-diagonals of 1..32 elements become 32
// work items of exactly 32 elements each, instead of 63 whose smallest move 16 bytes per thread.
// The block matrix is stored with its rows padded to NC = 4 ceil(s / 4) entries (zeros), and the row loop is compiled
// for each NC without any guard inside: the row's coefficients are wave-uniform, so they arrive through a few wide
// scalar loads issued together and feed the FMAs as SGPR operands.  (Round 2's first version guarded every column with
// `c < s`; the compiler then issued one s_load_dwordx2 + s_waitcnt per pair of FMAs -- ~100 cycles of exposed scalar-cache
// latency for 8 cycles of arithmetic, which is what the kernel's 7 ms were made of.)
// Block elements that are a constant step apart in memory (a beam splitter's anti-diagonals, a two-mode squeezer's
// diagonals) are addressed as first + c * step; arbitrary index sets go through the offset table, four entries at a time.
template <int NC, bool REALM, bool NT, bool STRIDED>
__device__ __forceinline__ void block_rows(amp_t *__restrict__ base, int s, const uint64_t *__restrict__ off,
                                           uint64_t first, int64_t step, const double *__restrict__ M) {
    amp_t x[NC];
    if constexpr (STRIDED) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const amp_t *src = base + first + static_cast<int64_t>(c) * step;
            if (c < NC - 3 || c < s) x[c] = NT ? __builtin_nontemporal_load(src) : *src;
            else x[c] = amp_t{0.0, 0.0};
        }
    } else {
#pragma unroll
        for (int c0 = 0; c0 < NC; c0 += 4) {
            uint64_t o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = off[c0 + j];   // the table is padded: reading past the block is safe
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = c0 + j;
                if (c < NC - 3 || c < s) x[c] = NT ? __builtin_nontemporal_load(base + o[j]) : base[o[j]];
                else x[c] = amp_t{0.0, 0.0};
            }
        }
    }
    constexpr int RS = (REALM ? 1 : 2) * NC;   // doubles per matrix row
    for (int row = 0; row < s; ++row) {
        const double *mr = M + static_cast<size_t>(row) * RS;
        amp_t acc = {0.0, 0.0};
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if constexpr (REALM) {
                acc.x = fma(mr[c], x[c].x, acc.x);
                acc.y = fma(mr[c], x[c].y, acc.y);
            } else {
                acc = cfma(cplx{mr[2 * c], mr[2 * c + 1]}, x[c], acc);
            }
        }
        amp_t *dst = STRIDED ? base + first + static_cast<int64_t>(row) * step : base + off[row];
        if (NT) __builtin_nontemporal_store(acc, dst);
        else *dst = acc;
    }
}

// (The tables are separate `const * __restrict__` kernel arguments on purpose: only then does the compiler know that
// nothing the kernel stores can alias them, which is what lets it fetch them through the scalar cache.)
template <int THREADS, bool REALM, bool STRIDED>
__global__ __launch_bounds__(THREADS) void k_mode2_blocks(amp_t *__restrict__ a, uint64_t L, int d, uint64_t Mid,
                                                         uint64_t R, int nblocks /* pairs */, uint64_t groups,
                                                         const int32_t *__restrict__ t_pairs,      // [2 * items], -1 = none
                                                         const int32_t *__restrict__ t_sizes,      // [blocks]
                                                         const int64_t *__restrict__ t_mat_start,  // [blocks] in doubles
                                                         const int32_t *__restrict__ t_idx_start,  // [blocks]
                                                         const uint64_t *__restrict__ t_plane_off,  // element offsets
                                                         const uint64_t *__restrict__ t_first,     // [blocks] STRIDED
                                                         const int64_t *__restrict__ t_step,       // [blocks] STRIDED
                                                         const double *__restrict__ t_mats) {
    constexpr bool NT = THREADS == QSV_BLOCK;   // contiguous plane groups touch every amplitude once: stream past the caches
    const uint64_t planes = L * Mid * R;
    const uint64_t items = (groups + 7) / 8 * 8 * nblocks;      // plane groups padded to a multiple of the 8 XCDs
    for (uint64_t w = blockIdx.x; w < items; w += gridDim.x) {
        // workgroup w runs on XCD w % 8; its sequence number there selects (group, block), block fastest
        const uint64_t seq = w / 8;
        const uint64_t g = (seq / nblocks) * 8 + w % 8;
        const int item = static_cast<int>(seq % nblocks);
        if (g >= groups) continue;
        const uint64_t p = g * THREADS + threadIdx.x;
        if (p >= planes) continue;
        const uint64_t r = p % R, m = (p / R) % Mid, l = p / (R * Mid);
        amp_t *base = a + l * (R * d * Mid * d) + m * (R * d) + r;
        for (int half = 0; half < 2; ++half) {
            const int b = t_pairs[2 * item + half];
            if (b < 0) continue;
            const int s = t_sizes[b];
            const uint64_t *off = t_plane_off + t_idx_start[b];
            const double *M = t_mats + t_mat_start[b];
            const uint64_t first = STRIDED ? t_first[b] : 0;
            const int64_t step = STRIDED ? t_step[b] : 0;
            switch ((s + 3) >> 2) {
            case 1: block_rows<4, REALM, NT, STRIDED>(base, s, off, first, step, M); break;
            case 2: block_rows<8, REALM, NT, STRIDED>(base, s, off, first, step, M); break;
            case 3: block_rows<12, REALM, NT, STRIDED>(base, s, off, first, step, M); break;
            case 4: block_rows<16, REALM, NT, STRIDED>(base, s, off, first, step, M); break;
            case 5: block_rows<20, REALM, NT, STRIDED>(base, s, off, first, step, M); break;
            case 6: block_rows<24, REALM, NT, STRIDED>(base, s, off, first, step, M); break;
            case 7: block_rows<28, REALM, NT, STRIDED>(base, s, off, first, step, M); break;
            default: block_rows<32, REALM, NT, STRIDED>(base, s, off, first, step, M); break;
            }
        }
    }
}


// Block-diagonal two-mode operator when the two modes are the LAST two (R = 1): the (d x d) plane is d^2 consecutive
// amplitudes (16 KiB for d = 32) and a block's elements lie 16 * (d - 1) bytes apart -- every 128-byte line holds
// amplitudes of eight different blocks.  The plane-per-thread form above then touches 16 bytes of 64 different lines per
// wave-instruction, and lines written piecemeal by different workgroups leave the L2 half filled (measured in round 1:
// 2.5x the algorithmic bytes written).  Here a 1024-thread workgroup owns whole planes instead:
//   * thread t loads amplitude t of the plane batch (1 KiB per wave-instruction, three batches ahead) into LDS;
//   * thread t is ALSO the owner of one output element e(t): it keeps that element's row of its block matrix in
//     registers for the whole launch (real matrices: <= 32 doubles) and per plane reads the block's inputs from LDS --
//     input k of a block sits a constant `stride` elements after input k-1, so the address is one register plus an
//     immediate; threads of the same block read the same address (a broadcast, no bank conflict);
//   * the result goes to a second LDS tile at its own place, and after a barrier thread t stores amplitude t of that
//     tile: every global access of the kernel is a whole 1 KiB segment, every line is written exactly once.
// Elements are dealt to threads in order of decreasing block size, so the waves of a workgroup have near-uniform trip
// counts (sum over waves = 342 iterations per plane for d = 32 against 512 for the natural order).
// Only for real block matrices (a complex row would need 128 VGPRs at 4 waves per SIMD); complex ones keep the form above.
constexpr int PLANE_THREADS = 1024;

struct PlaneArgs {
    uint64_t batches;     // plane batches in the register
    uint32_t batch_amps;  // amplitudes per batch (planes per batch x d^2), <= PLANE_THREADS
    int32_t stride;       // elements between consecutive inputs of a block (may be negative)
};

template <int STRIDE, bool NT>  // STRIDE != 0: compile-time stride (LDS offsets become immediates); 0: g.stride
__global__ __launch_bounds__(PLANE_THREADS) void k_mode2_plane(amp_t *__restrict__ a, const PlaneArgs g,
                                                               const double *__restrict__ coef,   // [threads][MAX_BLOCK]
                                                               const int32_t *__restrict__ rd0,   // [threads] first input
                                                               const int32_t *__restrict__ wr,    // [threads] my output
                                                               const int32_t *__restrict__ wave_trip,
                                                               const int32_t *__restrict__ count) {  // [threads] my block's size
    __shared__ amp_t in_t[PLANE_THREADS];
    __shared__ amp_t out_t[PLANE_THREADS];
    const int tid = threadIdx.x;
    const bool active = static_cast<uint32_t>(tid) < g.batch_amps;
    // lanes whose block is smaller than their wave's trip count do not read past their block: input k0 + j >= my_n would
    // lie outside the block -- up to 35 strides past it, i.e. outside in_t for the blocks near the end of the plane
    const int my_n = count[tid];
    in_t[tid] = amp_t{0.0, 0.0};
    out_t[tid] = amp_t{0.0, 0.0};
    double c[MAX_BLOCK];
#pragma unroll
    for (int k = 0; k < MAX_BLOCK; ++k) c[k] = coef[static_cast<size_t>(tid) * MAX_BLOCK + k];
    const int nk = __builtin_amdgcn_readfirstlane(wave_trip[tid >> 6]);
    const int stride = STRIDE != 0 ? STRIDE : g.stride;
    const amp_t *src = in_t + rd0[tid];
    amp_t *dst = out_t + wr[tid];
    const uint64_t step = gridDim.x;
    auto fetch = [&](uint64_t b) -> amp_t {
        amp_t v = {0.0, 0.0};
        if (active && b < g.batches) {
            const amp_t *p = a + b * g.batch_amps + tid;
            v = NT ? __builtin_nontemporal_load(p) : *p;
        }
        return v;
    };
    amp_t g0 = fetch(blockIdx.x), g1 = fetch(blockIdx.x + step), g2 = fetch(blockIdx.x + 2 * step);
    for (uint64_t b = blockIdx.x; b < g.batches; b += step) {
        const amp_t cur = g0;
        g0 = g1;
        g1 = g2;
        g2 = fetch(b + 3 * step);
        if (active) in_t[tid] = cur;
        __syncthreads();
        amp_t acc = {0.0, 0.0};
#pragma unroll
        for (int k0 = 0; k0 < MAX_BLOCK; k0 += 4) {
            if (k0 < nk) {  // wave-uniform; four LDS reads in flight per step
                amp_t x[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) x[j] = k0 + j < my_n ? src[(k0 + j) * stride] : amp_t{0.0, 0.0};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc.x = fma(c[k0 + j], x[j].x, acc.x);
                    acc.y = fma(c[k0 + j], x[j].y, acc.y);
                }
            }
        }
        if (active) *dst = acc;
        __syncthreads();
        if (active) {
            amp_t *p = a + b * g.batch_amps + tid;
            if (NT) __builtin_nontemporal_store(out_t[tid], p);
            else *p = out_t[tid];
        }
    }
}

// probs[j] += sum over everything but the mode axis of |a[l, j, r]|^2.  One workgroup per (j, slice of l):
// deterministic two-level sum (block partials [j][slice] summed on the host in index order).
__global__ __launch_bounds__(QSV_BLOCK) void k_mode_marginal(const amp_t *__restrict__ a, uint64_t L, int d, uint64_t R,
                                                            int slices, double *__restrict__ partials) {
    __shared__ double red[QSV_BLOCK / 64];
    const int j = blockIdx.x / slices, slice = blockIdx.x % slices;
    const uint64_t fibre = L * R;  // number of (l, r) pairs
    const uint64_t lo = fibre * slice / slices, hi = fibre * (slice + 1) / slices;
    double s = 0.0;
    for (uint64_t f = lo + threadIdx.x; f < hi; f += blockDim.x) {
        const uint64_t l = f / R, r = f % R;
        const amp_t v = a[(l * d + j) * R + r];
        s += v.x * v.x + v.y * v.y;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < QSV_BLOCK / 64; ++i) t += red[i];
        partials[blockIdx.x] = t;
    }
}

// out[l, r] = scale * a[l, level, r]: the register after a homodyne outcome (cv_simulator/gates.py:108-115).
__global__ __launch_bounds__(QSV_BLOCK) void k_mode_project(const amp_t *__restrict__ a, amp_t *__restrict__ out,
                                                           uint64_t L, int d, uint64_t R, int level, double scale) {
    const uint64_t total = L * R;
    for (uint64_t o = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x; o < total;
         o += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        const uint64_t l = o / R, r = o % R;
        amp_t v = a[(l * d + level) * R + r];
        v.x *= scale;
        v.y *= scale;
        out[o] = v;
    }
}

// out[l, j, r] = vec[j] * a[l, r]: a new mode in a product state (cv_simulator/gates.py:24-35 without the SVD).
__global__ __launch_bounds__(QSV_BLOCK) void k_mode_insert(const amp_t *__restrict__ a, amp_t *__restrict__ out,
                                                          uint64_t L, int d, uint64_t R,
                                                          const double *__restrict__ vec) {
    const uint64_t total = L * d * R;
    for (uint64_t o = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x; o < total;
         o += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        const uint64_t r = o % R, j = (o / R) % d, l = o / (R * d);
        out[o] = cmul(cplx{vec[2 * j], vec[2 * j + 1]}, a[l * R + r]);
    }
}

int check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return qsv_fail(QSV_EHIP, std::string("kernel launch: ") + hipGetErrorString(e));
    return QSV_OK;
}

uint64_t ipow(uint64_t b, int e) {
    uint64_t r = 1;
    while (e-- > 0) r *= b;
    return r;
}

int grid_of(uint64_t items, int per_block, int cap) {
    uint64_t b = (items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > static_cast<uint64_t>(cap)) b = cap;
    return static_cast<int>(b);
}

static bool axis_rows_enabled() {       // QSV_AXIS_ROWS=0: thin fibres through k_axis_simple, for comparisons
    static const bool on = [] {
        const char *e = getenv("QSV_AXIS_ROWS");
        return !(e && e[0] == '0');
    }();
    return on;
}

int launch_axis(int device, hipStream_t stream, const amp_t *in, amp_t *out, uint64_t L, uint64_t d_in,
                uint64_t d_out, uint64_t R, const double *dev_m) {
    // grids of >= 64 points with enough work to fill the chip are plain GEMMs: rocBLAS (f64 MFMA), qsv_gemm.hip
    if (d_in >= 64 && d_out >= 64 && 8.0 * L * d_in * d_out * R >= 2e8) {
        const int rc = qsvg_axis_gemm(device, stream, in, out, L, d_in, d_out, R, dev_m);
        if (rc != 0) return rc < 0 ? rc : QSV_OK;
    }
    const size_t lds = sizeof(amp_t) * d_in * TR;
    if (R >= TR && lds <= 160 * 1024 - 256) {
        const uint64_t r_tiles = (R + TR - 1) / TR;
        const int grid = grid_of(L * r_tiles, 1, 1 << 20);
        if (lds > 64 * 1024)
            QSV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_axis_tile),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
        hipLaunchKernelGGL(k_axis_tile, dim3(grid), dim3(QSV_BLOCK), lds, stream, in, out, L, static_cast<int>(d_in),
                           static_cast<int>(d_out), R, r_tiles, dev_m);
    } else if (R <= 8 && d_in >= 128 && L * d_out <= (1ull << 31) && axis_rows_enabled()) {
        const uint64_t rows = L * d_out, per_block = QSV_BLOCK / 64;
        hipLaunchKernelGGL(k_axis_rows, dim3(static_cast<unsigned>((rows + per_block - 1) / per_block)), dim3(QSV_BLOCK), 0,
                           stream, in, out, L, static_cast<int>(d_in), static_cast<int>(d_out), static_cast<int>(R), dev_m);
    } else {
        const int grid = grid_of(L * d_out * R, QSV_BLOCK, 1 << 16);
        hipLaunchKernelGGL(k_axis_simple, dim3(grid), dim3(QSV_BLOCK), 0, stream, in, out, L,
                           static_cast<int>(d_in), static_cast<int>(d_out), R, dev_m);
    }
    return check_launch();
}

// The spare buffer (`fresh`) now holds the register: swap it in (qsv_kernels.hip::qsvk_adopt).
int adopt(qsv_state *st, amp_t *fresh) {
    (void)fresh;
    return qsvk_adopt(st, st->amps);
}

}  // namespace

int qsvq_mode1(qsv_state *st, int mode, const double *m, bool diag) {
    const uint64_t d = st->d, R = ipow(d, st->n - 1 - mode), L = ipow(d, mode);
    // d = 2^K levels: the register IS a (K * n_modes)-qubit register and the mode gate a K-qubit gate on K
    // consecutive bits -- use the in-place, register-blocked qubit kernels (k_dense / k_dense_big) instead of the
    // out-of-place contraction below.
    int K = 0;
    while ((1ull << K) < d) ++K;
    if (!diag && (1ull << K) == d && K >= 1 && K <= 5) {
        int bits[5];
        for (int j = 0; j < K; ++j) bits[j] = K * (st->n - 1 - mode) + (K - 1 - j);  // leg 0 = top bit of the level
        const int modes = st->n;
        st->n = K * modes;  // the qubit kernels read the register size from the state
        const int rc = K <= 2 ? qsvk_dense(st, K, bits, 0, nullptr, m) : qsvk_generic(st, K, bits, m);
        st->n = modes;
        return rc;
    }
    const size_t bytes = sizeof(double) * 2 * (diag ? d : d * d);
    int rc = qsvk_ensure_matrix(st, bytes);
    if (rc) return rc;
    QSV_HIP(hipMemcpyAsync(st->dev_matrix, m, bytes, hipMemcpyHostToDevice, st->stream));
    // the caller's buffer is pageable and may be freed as soon as we return: HIP does not promise to have
    // staged an async pageable copy by then, so wait for it here (a few microseconds next to a full pass)
    QSV_HIP(hipStreamSynchronize(st->stream));
    if (diag) {
        const int grid = grid_of(st->amps, QSV_BLOCK * 4, 1 << 16);
        hipLaunchKernelGGL(k_axis_diag, dim3(grid), dim3(QSV_BLOCK), 0, st->stream, st->data, st->amps,
                           static_cast<int>(d), R, st->dev_matrix);
        return check_launch();
    }
    amp_t *fresh = nullptr;
    rc = qsvk_scratch(st, st->amps, &fresh);
    if (rc) return rc;
    rc = launch_axis(st->device, st->stream, st->data, fresh, L, d, d, R, st->dev_matrix);
    if (rc) return rc;
    return adopt(st, fresh);
}

int qsvq_mode2(qsv_state *st, int mode0, int mode1, const double *m, bool diag) {
    // G is given for legs (mode0, mode1); the kernels want (earlier axis, later axis)
    const uint64_t d = st->d, d2 = d * d;
    const int a = mode0 < mode1 ? mode0 : mode1, b = mode0 < mode1 ? mode1 : mode0;
    const uint64_t L = ipow(d, a), Mid = ipow(d, b - a - 1), R = ipow(d, st->n - 1 - b);
    std::vector<double> g;
    const double *src = m;
    if (mode0 > mode1) {  // transpose the two legs on rows and columns (or on the diagonal)
        if (diag) {
            g.resize(2 * d2);
            for (uint64_t i0 = 0; i0 < d; ++i0)
                for (uint64_t i1 = 0; i1 < d; ++i1) {
                    g[2 * (i1 * d + i0)] = m[2 * (i0 * d + i1)];
                    g[2 * (i1 * d + i0) + 1] = m[2 * (i0 * d + i1) + 1];
                }
        } else {
            g.resize(2 * d2 * d2);
            for (uint64_t i0 = 0; i0 < d; ++i0)
                for (uint64_t i1 = 0; i1 < d; ++i1)
                    for (uint64_t j0 = 0; j0 < d; ++j0)
                        for (uint64_t j1 = 0; j1 < d; ++j1) {
                            const uint64_t dst = (i1 * d + i0) * d2 + (j1 * d + j0);
                            const uint64_t s = (i0 * d + i1) * d2 + (j0 * d + j1);
                            g[2 * dst] = m[2 * s];
                            g[2 * dst + 1] = m[2 * s + 1];
                        }
        }
        src = g.data();
    }
    const size_t bytes = sizeof(double) * 2 * (diag ? d2 : d2 * d2);
    int rc = qsvk_ensure_matrix(st, bytes);
    if (rc) return rc;
    QSV_HIP(hipMemcpyAsync(st->dev_matrix, src, bytes, hipMemcpyHostToDevice, st->stream));
    QSV_HIP(hipStreamSynchronize(st->stream));  // `g` dies at return
    if (diag) {
        const int grid = grid_of(st->amps, QSV_BLOCK * 4, 1 << 16);
        hipLaunchKernelGGL(k_mode2_diag, dim3(grid), dim3(QSV_BLOCK), 0, st->stream, st->data, st->amps,
                           static_cast<int>(d), Mid, R, st->dev_matrix);
        return check_launch();
    }
    amp_t *fresh = nullptr;
    rc = qsvk_scratch(st, st->amps, &fresh);
    if (rc) return rc;
    const int grid = grid_of(st->amps, QSV_BLOCK, 1 << 16);
    hipLaunchKernelGGL(k_mode2_simple, dim3(grid), dim3(QSV_BLOCK), 0, st->stream, st->data, fresh, L,
                       static_cast<int>(d), Mid, R, st->dev_matrix);
    rc = check_launch();
    if (rc) return rc;
    return adopt(st, fresh);
}

int qsvq_mode2_gather(qsv_state *st, int mode0, int mode1, int nnz, const int32_t *cols, const double *vals) {
    const uint64_t d = st->d, d2 = d * d;
    const int a = mode0 < mode1 ? mode0 : mode1, b = mode0 < mode1 ? mode1 : mode0;
    const uint64_t L = ipow(d, a), Mid = ipow(d, b - a - 1), R = ipow(d, st->n - 1 - b);
    // kernel order is (earlier axis, later axis); transpose rows and columns when mode0 is the later one
    std::vector<int32_t> c(d2 * nnz);
    std::vector<double> v(2 * d2 * nnz);
    for (uint64_t i0 = 0; i0 < d; ++i0)
        for (uint64_t i1 = 0; i1 < d; ++i1)
            for (int k = 0; k < nnz; ++k) {
                const uint64_t src = (i0 * d + i1) * nnz + k;
                const uint64_t dst = (mode0 < mode1 ? (i0 * d + i1) : (i1 * d + i0)) * nnz + k;
                int32_t col = cols[src];
                if (col >= static_cast<int32_t>(d2)) return qsv_fail(QSV_EINVAL, "gather column out of range");
                if (col >= 0 && mode0 > mode1) col = static_cast<int32_t>((col % d) * d + col / d);
                c[dst] = col;
                v[2 * dst] = vals[2 * src];
                v[2 * dst + 1] = vals[2 * src + 1];
            }
    const size_t vbytes = sizeof(double) * v.size(), cbytes = sizeof(int32_t) * c.size();
    int rc = qsvk_ensure_matrix(st, vbytes + cbytes);
    if (rc) return rc;
    int32_t *dcols = reinterpret_cast<int32_t *>(reinterpret_cast<char *>(st->dev_matrix) + vbytes);
    QSV_HIP(hipMemcpyAsync(st->dev_matrix, v.data(), vbytes, hipMemcpyHostToDevice, st->stream));
    QSV_HIP(hipMemcpyAsync(dcols, c.data(), cbytes, hipMemcpyHostToDevice, st->stream));
    QSV_HIP(hipStreamSynchronize(st->stream));  // c and v die at return
    amp_t *fresh = nullptr;
    rc = qsvk_scratch(st, st->amps, &fresh);
    if (rc) return rc;
    const int grid = grid_of(st->amps, QSV_BLOCK, 1 << 16);
    hipLaunchKernelGGL(k_mode2_gather, dim3(grid), dim3(QSV_BLOCK), 0, st->stream, st->data, fresh, L,
                       static_cast<int>(d), Mid, R, nnz, dcols, st->dev_matrix);
    rc = check_launch();
    if (rc) return rc;
    return adopt(st, fresh);
}


constexpr int QSV_UNHANDLED = 1 << 20;  // internal: "this fast path does not apply, take the general one"

// The R = 1 workgroup-per-plane form (k_mode2_plane).  `off[idx_start[b] + j]` is the plane offset of element j of block
// b (R = 1: an index into the d^2 consecutive amplitudes of the plane).  Needs the same element-to-element stride in
// every block; returns QSV_UNHANDLED otherwise.
static int launch_plane(qsv_state *st, uint64_t d, uint64_t L, int nblocks, const int32_t *sizes,
                        const std::vector<int64_t> &mat_start, const std::vector<int32_t> &idx_start,
                        const std::vector<uint64_t> &off, const double *mats) {
    const int plane = static_cast<int>(d * d);
    int64_t stride = 0;
    for (int b = 0; b < nblocks; ++b)
        for (int j = 1; j < sizes[b]; ++j) {
            const int64_t step = static_cast<int64_t>(off[idx_start[b] + j]) - static_cast<int64_t>(off[idx_start[b] + j - 1]);
            if (stride == 0) stride = step;
            if (step != stride) return QSV_UNHANDLED;
        }
    // every plane element: (block, row inside the block); elements outside all blocks are 1 x 1 identities
    struct Elem { int e, block, row, size; };
    std::vector<Elem> elems;
    std::vector<uint8_t> covered(plane, 0);
    for (int b = 0; b < nblocks; ++b)
        for (int j = 0; j < sizes[b]; ++j) {
            const int e = static_cast<int>(off[idx_start[b] + j]);
            elems.push_back({e, b, j, sizes[b]});
            covered[e] = 1;
        }
    for (int e = 0; e < plane; ++e)
        if (!covered[e]) elems.push_back({e, -1, 0, 1});
    std::stable_sort(elems.begin(), elems.end(), [](const Elem &x, const Elem &y) { return x.size > y.size; });
    const int per_batch = PLANE_THREADS / plane;             // planes per batch
    if (L % per_batch != 0) return QSV_UNHANDLED;
    const int used = per_batch * plane;
    // thread t: sorted position t / per_batch of plane slot t % per_batch, so that a wave holds elements of equal rank
    std::vector<double> coef(static_cast<size_t>(PLANE_THREADS) * MAX_BLOCK, 0.0);
    std::vector<int32_t> rd0(PLANE_THREADS, 0), wr(PLANE_THREADS, 0), trip(PLANE_THREADS / 64, 0), cnt(PLANE_THREADS, 0);
    for (int t = 0; t < used; ++t) {
        const Elem &el = elems[t / per_batch];
        const int slot = t % per_batch;
        wr[t] = slot * plane + el.e;
        if (el.block < 0) {
            coef[static_cast<size_t>(t) * MAX_BLOCK] = 1.0;
            rd0[t] = wr[t];
        } else {
            const double *row = mats + 2 * (mat_start[el.block] + static_cast<int64_t>(el.row) * el.size);
            for (int k = 0; k < el.size; ++k) coef[static_cast<size_t>(t) * MAX_BLOCK + k] = row[2 * k];
            rd0[t] = slot * plane + static_cast<int>(off[idx_start[el.block]]);
        }
        trip[t / 64] = std::max(trip[t / 64], el.size);
        cnt[t] = el.size;
    }
    // one image [coef | rd0 | wr | trip | cnt] through the staging ring (no host wait: see qsvk_stage)
    const size_t b_c = qsv_pad16(sizeof(double) * coef.size()), b_i = qsv_pad16(sizeof(int32_t) * PLANE_THREADS);
    std::vector<char> image(b_c + 4 * b_i, 0);
    memcpy(image.data(), coef.data(), sizeof(double) * coef.size());
    memcpy(image.data() + b_c, rd0.data(), sizeof(int32_t) * PLANE_THREADS);
    memcpy(image.data() + b_c + b_i, wr.data(), sizeof(int32_t) * PLANE_THREADS);
    memcpy(image.data() + b_c + 2 * b_i, trip.data(), sizeof(int32_t) * trip.size());
    memcpy(image.data() + b_c + 3 * b_i, cnt.data(), sizeof(int32_t) * PLANE_THREADS);
    StageRef staged;
    int rc = qsvk_stage(st, image.data(), image.size(), nullptr, 0, &staged);
    if (rc) return rc;
    char *p = staged.dev;
    double *d_c = reinterpret_cast<double *>(p);
    int32_t *d_r = reinterpret_cast<int32_t *>(p + b_c), *d_w = reinterpret_cast<int32_t *>(p + b_c + b_i),
            *d_t = reinterpret_cast<int32_t *>(p + b_c + 2 * b_i), *d_n = reinterpret_cast<int32_t *>(p + b_c + 3 * b_i);
    PlaneArgs g;
    g.batches = L / per_batch;
    g.batch_amps = static_cast<uint32_t>(used);
    g.stride = static_cast<int32_t>(stride);
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, st->device);  // 256 if the query fails
    const unsigned grid = static_cast<unsigned>(std::min<uint64_t>(g.batches, static_cast<uint64_t>(cus)));
    const bool nt = st->nontemporal != 0;
    const int cstride = (stride == 31 || stride == -31) ? static_cast<int>(stride) : 0;
    snprintf(st->last_kernel, sizeof(st->last_kernel), "k_mode2_plane<%d, %s>", cstride, nt ? "true" : "false");
#define QSV_LAUNCH_PLANE(S, N) \
    hipLaunchKernelGGL((k_mode2_plane<S, N>), dim3(grid), dim3(PLANE_THREADS), 0, st->stream, st->data, g, d_c, d_r, d_w, d_t, d_n)
    if (cstride == 31) { if (nt) QSV_LAUNCH_PLANE(31, true); else QSV_LAUNCH_PLANE(31, false); }
    else if (cstride == -31) { if (nt) QSV_LAUNCH_PLANE(-31, true); else QSV_LAUNCH_PLANE(-31, false); }
    else { if (nt) QSV_LAUNCH_PLANE(0, true); else QSV_LAUNCH_PLANE(0, false); }
#undef QSV_LAUNCH_PLANE
    rc = check_launch();
    if (rc) return rc;
    return qsvk_stage_done(st, staged);
}

int qsvq_mode2_blocks(qsv_state *st, int mode0, int mode1, int nblocks, const int32_t *sizes,
                      const int32_t *plane_indices, const double *mats) {
    const uint64_t d = st->d;
    const int a = mode0 < mode1 ? mode0 : mode1, b = mode0 < mode1 ? mode1 : mode0;
    const uint64_t L = ipow(d, a), Mid = ipow(d, b - a - 1), R = ipow(d, st->n - 1 - b);
    const uint64_t stride0 = mode0 < mode1 ? R * d * Mid : R, stride1 = mode0 < mode1 ? R : R * d * Mid;
    std::vector<int64_t> mat_start(nblocks);
    std::vector<int32_t> idx_start(nblocks);
    int64_t mtot = 0;
    int32_t itot = 0;
    std::vector<uint8_t> seen(d * d, 0);
    for (int k = 0; k < nblocks; ++k) {
        if (sizes[k] < 1 || sizes[k] > MAX_BLOCK)
            return qsv_fail(QSV_EINVAL, "block sizes must be in 1.." + std::to_string(MAX_BLOCK));
        mat_start[k] = mtot;
        idx_start[k] = itot;
        mtot += static_cast<int64_t>(sizes[k]) * sizes[k];
        itot += sizes[k];
    }
    std::vector<uint64_t> off(itot + MAX_BLOCK, 0);  // padded: the kernel's unrolled loads may look one block ahead
    for (int32_t i = 0; i < itot; ++i) {
        const int32_t pi = plane_indices[i];
        if (pi < 0 || pi >= static_cast<int32_t>(d * d) || seen[pi]++)
            return qsv_fail(QSV_EINVAL, "plane indices must be distinct and inside the (d, d) plane");
        off[i] = (pi / d) * stride0 + (pi % d) * stride1;
    }
    bool real = true;
    for (int64_t i = 0; i < mtot && real; ++i) real = mats[2 * i + 1] == 0.0;
    if (real && R == 1 && Mid == 1 && d * d <= static_cast<uint64_t>(PLANE_THREADS) && st->plane_kernel) {
        const int rc_plane = launch_plane(st, d, L, nblocks, sizes, mat_start, idx_start, off, mats);
        if (rc_plane != QSV_UNHANDLED) return rc_plane;
    }
    // padded matrices: row stride 4 ceil(s / 4) entries, one double per entry when every block is real
    std::vector<int64_t> pad_start(nblocks);
    int64_t ptot = 0;
    const int per = real ? 1 : 2;
    for (int k = 0; k < nblocks; ++k) {
        pad_start[k] = ptot;
        ptot += static_cast<int64_t>(per) * sizes[k] * ((sizes[k] + 3) / 4 * 4);
    }
    std::vector<double> pad_mats(ptot + 2 * MAX_BLOCK, 0.0);
    for (int k = 0; k < nblocks; ++k) {
        const int sz = sizes[k], s4 = (sz + 3) / 4 * 4;
        for (int r = 0; r < sz; ++r)
            for (int c = 0; c < sz; ++c) {
                const double *src = mats + 2 * (mat_start[k] + static_cast<int64_t>(r) * sz + c);
                double *dst = pad_mats.data() + pad_start[k] + static_cast<int64_t>(per) * (r * s4 + c);
                dst[0] = src[0];
                if (!real) dst[1] = src[1];
            }
    }
    // constant step between the elements of every block?
    bool strided = true;
    std::vector<uint64_t> first(nblocks);
    std::vector<int64_t> step(nblocks, 0);
    for (int k = 0; k < nblocks && strided; ++k) {
        const uint64_t *o = off.data() + idx_start[k];
        first[k] = o[0];
        if (sizes[k] > 1) step[k] = static_cast<int64_t>(o[1]) - static_cast<int64_t>(o[0]);
        for (int c = 2; c < sizes[k] && strided; ++c)
            strided = static_cast<int64_t>(o[c]) - static_cast<int64_t>(o[c - 1]) == step[k];
    }
    // pair blocks: largest with the smallest that still fits MAX_BLOCK elements together
    std::vector<int> order(nblocks);
    for (int k = 0; k < nblocks; ++k) order[k] = k;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return sizes[x] > sizes[y]; });
    std::vector<int32_t> pairs;
    for (int lo = 0, hi = nblocks - 1; lo <= hi; ++lo) {
        pairs.push_back(order[lo]);
        if (lo < hi && sizes[order[lo]] + sizes[order[hi]] <= MAX_BLOCK) pairs.push_back(order[hi--]);
        else pairs.push_back(-1);
    }
    const int nitems = static_cast<int>(pairs.size() / 2);
    // one device buffer: [mats | plane_off | first | step | mat_start | sizes | idx_start | pairs], sections 16-byte aligned
    auto pad16 = [](size_t x) { return (x + 15) / 16 * 16; };
    struct Section { const void *src; size_t bytes; size_t at; };
    Section sec[8] = {{pad_mats.data(), sizeof(double) * pad_mats.size(), 0},
                      {off.data(), sizeof(uint64_t) * off.size(), 0},
                      {first.data(), sizeof(uint64_t) * nblocks, 0},
                      {step.data(), sizeof(int64_t) * nblocks, 0},
                      {pad_start.data(), sizeof(int64_t) * nblocks, 0},
                      {sizes, sizeof(int32_t) * nblocks, 0},
                      {idx_start.data(), sizeof(int32_t) * nblocks, 0},
                      {pairs.data(), sizeof(int32_t) * pairs.size(), 0}};
    size_t total = 0;
    for (auto &x : sec) {
        x.at = total;
        total += pad16(x.bytes);
    }
    // one image, one transfer through the staging ring (no host wait: see qsvk_stage)
    std::vector<char> image(total, 0);
    for (auto &x : sec) memcpy(image.data() + x.at, x.src, x.bytes);
    StageRef staged;
    int rc = qsvk_stage(st, image.data(), total, nullptr, 0, &staged);
    if (rc) return rc;
    char *p = staged.dev;
    const double *t_mats = reinterpret_cast<const double *>(p + sec[0].at);
    const uint64_t *t_off = reinterpret_cast<const uint64_t *>(p + sec[1].at);
    const uint64_t *t_first = reinterpret_cast<const uint64_t *>(p + sec[2].at);
    const int64_t *t_step = reinterpret_cast<const int64_t *>(p + sec[3].at);
    const int64_t *t_start = reinterpret_cast<const int64_t *>(p + sec[4].at);
    const int32_t *t_sizes = reinterpret_cast<const int32_t *>(p + sec[5].at);
    const int32_t *t_idx = reinterpret_cast<const int32_t *>(p + sec[6].at);
    const int32_t *t_pairs = reinterpret_cast<const int32_t *>(p + sec[7].at);
    const uint64_t planes = L * Mid * R;
    // R >= 8: a 128-byte line holds amplitudes of one plane point only, so every line is touched by exactly one block
    // (stream, nontemporal).  R < 8: lines are shared by up to 8 anti-diagonals; one wave per workgroup keeps a plane
    // group (64 planes) small enough for the XCD's L2 to serve the re-touched lines.
    const int threads = R >= 8 ? QSV_BLOCK : 64;
    const uint64_t groups = (planes + threads - 1) / threads;
    const uint64_t items = (groups + 7) / 8 * 8 * nitems;
    const unsigned grid = static_cast<unsigned>(items < 0x00ffffffull ? items : 0x00ffffffull);
    snprintf(st->last_kernel, sizeof(st->last_kernel), "k_mode2_blocks<%d, %s, %s>", threads, real ? "true" : "false",
             strided ? "true" : "false");
    const int di = static_cast<int>(d);
#define QSV_LAUNCH_BLOCKS(T, RM, SD)                                                                                   \
    hipLaunchKernelGGL((k_mode2_blocks<T, RM, SD>), dim3(grid), dim3(T), 0, st->stream, st->data, L, di, Mid, R, nitems, \
                       groups, t_pairs, t_sizes, t_start, t_idx, t_off, t_first, t_step, t_mats)
    if (threads == QSV_BLOCK) {
        if (real && strided) QSV_LAUNCH_BLOCKS(QSV_BLOCK, true, true);
        else if (real) QSV_LAUNCH_BLOCKS(QSV_BLOCK, true, false);
        else if (strided) QSV_LAUNCH_BLOCKS(QSV_BLOCK, false, true);
        else QSV_LAUNCH_BLOCKS(QSV_BLOCK, false, false);
    } else {
        if (real && strided) QSV_LAUNCH_BLOCKS(64, true, true);
        else if (real) QSV_LAUNCH_BLOCKS(64, true, false);
        else if (strided) QSV_LAUNCH_BLOCKS(64, false, true);
        else QSV_LAUNCH_BLOCKS(64, false, false);
    }
#undef QSV_LAUNCH_BLOCKS
    rc = check_launch();
    if (rc) return rc;
    return qsvk_stage_done(st, staged);
}

int qsvq_mode_marginal(qsv_state *st, int mode, double *probs) {
    const uint64_t d = st->d, R = ipow(d, st->n - 1 - mode), L = ipow(d, mode);
    int slices = static_cast<int>(std::min<uint64_t>(std::max<uint64_t>(1, (L * R) / (QSV_BLOCK * 8)), 256));
    while (static_cast<uint64_t>(slices) * d > (1u << 20)) slices = (slices + 1) / 2;
    const size_t bytes = sizeof(double) * d * slices;
    int rc = qsvk_ensure_matrix(st, bytes);
    if (rc) return rc;
    hipLaunchKernelGGL(k_mode_marginal, dim3(static_cast<unsigned>(d * slices)), dim3(QSV_BLOCK), 0, st->stream,
                       st->data, L, static_cast<int>(d), R, slices, st->dev_matrix);
    rc = check_launch();
    if (rc) return rc;
    std::vector<double> host(d * slices);
    QSV_HIP(hipMemcpyAsync(host.data(), st->dev_matrix, bytes, hipMemcpyDeviceToHost, st->stream));
    QSV_HIP(hipStreamSynchronize(st->stream));
    for (uint64_t j = 0; j < d; ++j) {
        double s = 0.0;
        for (int k = 0; k < slices; ++k) s += host[j * slices + k];
        probs[j] = s;
    }
    return QSV_OK;
}

// Replace the register by `fresh` holding new_amps amplitudes (the mode count changed).
static int adopt_resized(qsv_state *st, amp_t *fresh, uint64_t new_amps) {
    (void)fresh;
    return qsvk_adopt(st, new_amps);
}

int qsvq_mode_project(qsv_state *st, int mode, int level, double scale) {
    const uint64_t d = st->d, R = ipow(d, st->n - 1 - mode), L = ipow(d, mode);
    amp_t *fresh = nullptr;
    int rc = qsvk_scratch(st, L * R, &fresh);
    if (rc) return rc;
    const int grid = grid_of(L * R, QSV_BLOCK * 4, 1 << 16);
    hipLaunchKernelGGL(k_mode_project, dim3(grid), dim3(QSV_BLOCK), 0, st->stream, st->data, fresh, L,
                       static_cast<int>(d), R, level, scale);
    rc = check_launch();
    if (rc) return rc;
    st->n -= 1;
    return adopt_resized(st, fresh, L * R);
}

int qsvq_mode_insert(qsv_state *st, int mode, const double *vec) {
    const uint64_t d = st->d, R = ipow(d, st->n - mode), L = ipow(d, mode);  // the register has st->n modes now
    const uint64_t out_amps = st->amps * d;
    if (!st->owns_data && out_amps > st->capacity)
        return qsv_fail(QSV_ENOMEM, "insert: the caller-owned buffer has no room for one more mode");
    int rc = qsvk_ensure_matrix(st, sizeof(double) * 2 * d);
    if (rc) return rc;
    QSV_HIP(hipMemcpyAsync(st->dev_matrix, vec, sizeof(double) * 2 * d, hipMemcpyHostToDevice, st->stream));
    amp_t *fresh = nullptr;
    rc = qsvk_scratch(st, out_amps, &fresh);
    if (rc) return rc;
    const int grid = grid_of(out_amps, QSV_BLOCK * 4, 1 << 16);
    hipLaunchKernelGGL(k_mode_insert, dim3(grid), dim3(QSV_BLOCK), 0, st->stream, st->data, fresh, L,
                       static_cast<int>(d), R, st->dev_matrix);
    rc = check_launch();
    if (rc) return rc;
    QSV_HIP(hipStreamSynchronize(st->stream));  // `vec` was read by the async copy
    st->n += 1;
    return adopt_resized(st, fresh, out_amps);
}

int qsvq_tensor_axis_dev(int device, hipStream_t stream, const amp_t *in, amp_t *out, uint64_t L, uint64_t d_in,
                         uint64_t d_out, uint64_t R, const double *dev_m) {
    QSV_HIP(hipSetDevice(device));
    return launch_axis(device, stream, in, out, L, d_in, d_out, R, dev_m);
}

int qsvq_tensor_axis(int device, hipStream_t stream, const amp_t *in, amp_t *out, uint64_t L, uint64_t d_in,
                     uint64_t d_out, uint64_t R, const double *m_host) {
    QSV_HIP(hipSetDevice(device));
    double *dev_m = nullptr;
    const size_t bytes = sizeof(double) * 2 * d_in * d_out;
    if (hipMalloc(reinterpret_cast<void **>(&dev_m), bytes) != hipSuccess)
        return qsv_fail(QSV_ENOMEM, "device allocation of the operator failed");
    hipError_t e = hipMemcpyAsync(dev_m, m_host, bytes, hipMemcpyHostToDevice, stream);
    int rc = e == hipSuccess ? launch_axis(device, stream, in, out, L, d_in, d_out, R, dev_m)
                             : qsv_fail(QSV_EHIP, std::string("operator upload: ") + hipGetErrorString(e));
    (void)hipStreamSynchronize(stream);
    (void)hipFree(dev_m);
    return rc;
}

// ====================================================================================================
// Raw-tensor entry points for matrix-product-state sites (cv_simulator/mps.py:102-201 keeps the state as a list of
// (chi_l, d, chi_r) tensors).  Same kernels as the register path, addressed by (L, d, R) instead of through a
// qsv_state; every operand is a device pointer and every call is asynchronous on `stream`.
// ====================================================================================================
namespace {

// out[j] partial sums of Re( z[l, j, r] * conj(t[l, j, r]) ) over (l, r): the last step of the reduced density
// diagonal (mps.py:188-189 restricted to i == j).
__global__ __launch_bounds__(QSV_BLOCK) void k_axis_overlap(const amp_t *__restrict__ z, const amp_t *__restrict__ t,
                                                           uint64_t L, int d, uint64_t R,
                                                           double *__restrict__ out) {
    __shared__ double red[QSV_BLOCK / 64];
    const int j = blockIdx.x;
    const uint64_t fibre = L * R;
    double s = 0.0;
    for (uint64_t f = threadIdx.x; f < fibre; f += blockDim.x) {
        const uint64_t l = f / R, r = f % R;
        const amp_t a = z[(l * d + j) * R + r], b = t[(l * d + j) * R + r];
        s += a.x * b.x + a.y * b.y;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double acc = 0.0;
        for (int i = 0; i < QSV_BLOCK / 64; ++i) acc += red[i];
        out[j] = acc;
    }
}

// theta[a, j, l, b] *= exp(i s q_j q_l): the CZ phases (gates.py:159) evaluated in the kernel -- no (d, d) table.
__global__ __launch_bounds__(QSV_BLOCK) void k_plane_phase(amp_t *__restrict__ t, uint64_t total, int d, uint64_t R,
                                                          const double *__restrict__ qs, double strength) {
    for (uint64_t o = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x; o < total;
         o += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        const uint64_t i1 = (o / R) % d, i0 = (o / (R * d)) % d;
        double sn, cs;
        sincos(strength * (qs[i0] * qs[i1]), &sn, &cs);
        t[o] = cmul(cplx{cs, sn}, t[o]);
    }
}

// Lower cell of x on the sorted grid qs the way RegularGridInterpolator finds it (searchsorted(grid, x) - 1, clipped
// to [0, d - 2]) and the normalised distance into that cell.
__device__ __forceinline__ int grid_cell(const double *__restrict__ qs, int d, double inv_dq, double x, double *frac) {
    int i = static_cast<int>(floor((x - qs[0]) * inv_dq));
    i = i < 0 ? 0 : (i > d - 2 ? d - 2 : i);
    while (i > 0 && !(qs[i] < x)) --i;                 // largest i with qs[i] < x ...
    while (i < d - 2 && qs[i + 1] < x) ++i;            // ... within the clip range
    *frac = (x - qs[i]) / (qs[i + 1] - qs[i]);
    return i;
}

// out[a, i0, i1, b] = bilinear interpolation of the plane in[a, :, :, b] at the affine image of (q_i0, q_i1),
// zero outside the grid: the RegularGridInterpolator loop of BS / CX (gates.py:74-80,187-189) for every bond pair at
// once, with the four source points and weights computed in the kernel instead of read from a 48-byte-per-point table.
__global__ __launch_bounds__(QSV_BLOCK) void k_plane_affine(const amp_t *__restrict__ in, amp_t *__restrict__ out,
                                                           uint64_t L, int d, uint64_t R,
                                                           const double *__restrict__ qs, double a00, double a01,
                                                           double a10, double a11) {
    const uint64_t total = L * d * d * R;
    const uint64_t s1 = R, s0 = R * d, sl = R * d * d;
    const double lo = qs[0], hi = qs[d - 1], inv_dq = (d - 1) / (hi - lo);
    for (uint64_t o = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x; o < total;
         o += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        const uint64_t r = o % R, i1 = (o / s1) % d, i0 = (o / s0) % d, l = o / sl;
        const double x = qs[i0], y = qs[i1];
        const double xs = a00 * x + a01 * y, ys = a10 * x + a11 * y;
        amp_t acc = {0.0, 0.0};
        if (xs >= lo && xs <= hi && ys >= lo && ys <= hi) {
            double f0, f1;
            const int j0 = grid_cell(qs, d, inv_dq, xs, &f0), j1 = grid_cell(qs, d, inv_dq, ys, &f1);
            const amp_t *p = in + l * sl + r + j0 * s0 + j1 * s1;
            const amp_t v00 = p[0], v01 = p[s1], v10 = p[s0], v11 = p[s0 + s1];
            const double w00 = (1 - f0) * (1 - f1), w01 = (1 - f0) * f1, w10 = f0 * (1 - f1), w11 = f0 * f1;
            acc.x = w00 * v00.x + w01 * v01.x + w10 * v10.x + w11 * v11.x;
            acc.y = w00 * v00.y + w01 * v01.y + w10 * v10.y + w11 * v11.y;
        }
        out[o] = acc;
    }
}

// out[x, y, z, w] = p[x, z] * q[y, w]  (or out[x, y, w, z] with swap_last): the two outer products of InsertBell.apply
// (gkp_simulator/insert_bell.py:80,87 -- "aib,kd -> aikbd" and "dl,bjc -> bdljc") with the bond legs kept adjacent.
__global__ __launch_bounds__(QSV_BLOCK) void k_outer(const amp_t *__restrict__ p, const amp_t *__restrict__ q,
                                                    amp_t *__restrict__ out, uint64_t X, uint64_t Y, uint64_t Z,
                                                    uint64_t W, int swap_last) {
    const uint64_t total = X * Y * Z * W;
    for (uint64_t o = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x; o < total;
         o += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        const uint64_t last = swap_last ? Z : W, mid = swap_last ? W : Z;
        const uint64_t i3 = o % last, i2 = (o / last) % mid, y = (o / (last * mid)) % Y, x = o / (last * mid * Y);
        const uint64_t z = swap_last ? i3 : i2, w = swap_last ? i2 : i3;
        const amp_t a = p[x * Z + z], b = q[y * W + w];
        out[o] = cmul(cplx{a.x, a.y}, b);
    }
}

}  // namespace

int qsvq_tensor_scale_axis(int device, hipStream_t stream, amp_t *t, uint64_t L, uint64_t d, uint64_t R,
                           const double *dev_diag) {
    QSV_HIP(hipSetDevice(device));
    const uint64_t total = L * d * R;
    const int grid = grid_of(total, QSV_BLOCK * 4, 1 << 16);
    hipLaunchKernelGGL(k_axis_diag, dim3(grid), dim3(QSV_BLOCK), 0, stream, t, total, static_cast<int>(d), R, dev_diag);
    return check_launch();
}

int qsvq_tensor_plane_diag(int device, hipStream_t stream, amp_t *t, uint64_t L, uint64_t d, uint64_t R,
                           const double *dev_plane) {
    QSV_HIP(hipSetDevice(device));
    const uint64_t total = L * d * d * R;
    const int grid = grid_of(total, QSV_BLOCK * 4, 1 << 16);
    hipLaunchKernelGGL(k_mode2_diag, dim3(grid), dim3(QSV_BLOCK), 0, stream, t, total, static_cast<int>(d),
                       static_cast<uint64_t>(1), R, dev_plane);
    return check_launch();
}

int qsvq_tensor_plane_gather(int device, hipStream_t stream, const amp_t *in, amp_t *out, uint64_t L, uint64_t d,
                             uint64_t R, int nnz, const int32_t *dev_cols, const double *dev_vals) {
    QSV_HIP(hipSetDevice(device));
    const int grid = grid_of(L * d * d * R, QSV_BLOCK, 1 << 16);
    hipLaunchKernelGGL(k_mode2_gather, dim3(grid), dim3(QSV_BLOCK), 0, stream, in, out, L, static_cast<int>(d),
                       static_cast<uint64_t>(1), R, nnz, dev_cols, dev_vals);
    return check_launch();
}

int qsvq_tensor_take_level(int device, hipStream_t stream, const amp_t *in, amp_t *out, uint64_t L, uint64_t d,
                           uint64_t R, uint64_t level, double scale) {
    QSV_HIP(hipSetDevice(device));
    const int grid = grid_of(L * R, QSV_BLOCK, 1 << 16);
    hipLaunchKernelGGL(k_mode_project, dim3(grid), dim3(QSV_BLOCK), 0, stream, in, out, L, static_cast<int>(d), R,
                       static_cast<int>(level), scale);
    return check_launch();
}

int qsvq_tensor_insert_axis(int device, hipStream_t stream, const amp_t *in, amp_t *out, uint64_t L, uint64_t d,
                            uint64_t R, const double *dev_vec) {
    QSV_HIP(hipSetDevice(device));
    const int grid = grid_of(L * d * R, QSV_BLOCK, 1 << 16);
    hipLaunchKernelGGL(k_mode_insert, dim3(grid), dim3(QSV_BLOCK), 0, stream, in, out, L, static_cast<int>(d), R,
                       dev_vec);
    return check_launch();
}

int qsvq_tensor_axis_overlap(int device, hipStream_t stream, const amp_t *z, const amp_t *t, uint64_t L, uint64_t d,
                             uint64_t R, double *dev_out) {
    QSV_HIP(hipSetDevice(device));
    hipLaunchKernelGGL(k_axis_overlap, dim3(static_cast<unsigned>(d)), dim3(QSV_BLOCK), 0, stream, z, t, L,
                       static_cast<int>(d), R, dev_out);
    return check_launch();
}

int qsvq_tensor_plane_phase(int device, hipStream_t stream, amp_t *t, uint64_t L, uint64_t d, uint64_t R,
                            const double *dev_qs, double strength) {
    QSV_HIP(hipSetDevice(device));
    const uint64_t total = L * d * d * R;
    const int grid = grid_of(total, QSV_BLOCK * 4, 1 << 16);
    hipLaunchKernelGGL(k_plane_phase, dim3(grid), dim3(QSV_BLOCK), 0, stream, t, total, static_cast<int>(d), R, dev_qs,
                       strength);
    return check_launch();
}

int qsvq_tensor_plane_affine(int device, hipStream_t stream, const amp_t *in, amp_t *out, uint64_t L, uint64_t d,
                             uint64_t R, const double *dev_qs, const double *a) {
    QSV_HIP(hipSetDevice(device));
    const int grid = grid_of(L * d * d * R, QSV_BLOCK, 1 << 18);
    hipLaunchKernelGGL(k_plane_affine, dim3(grid), dim3(QSV_BLOCK), 0, stream, in, out, L, static_cast<int>(d), R,
                       dev_qs, a[0], a[1], a[2], a[3]);
    return check_launch();
}

int qsvq_tensor_outer(int device, hipStream_t stream, const amp_t *p, const amp_t *q, amp_t *out, uint64_t X, uint64_t Y,
                      uint64_t Z, uint64_t W, int swap_last) {
    QSV_HIP(hipSetDevice(device));
    const int grid = grid_of(X * Y * Z * W, QSV_BLOCK, 1 << 16);
    hipLaunchKernelGGL(k_outer, dim3(grid), dim3(QSV_BLOCK), 0, stream, p, q, out, X, Y, Z, W, swap_last);
    return check_launch();
}
