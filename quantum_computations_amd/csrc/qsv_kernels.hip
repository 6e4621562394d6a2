// Qubit-register kernels of libqsv.so, written for gfx950 (MI355X, wave64) only.
//
// Every gate is one streaming pass over the complex128 register in HBM: the path is bandwidth-bound
// (0.44-0.94 flop/B), so the kernels are organised around memory access, not arithmetic:
//   * a wave always touches 64 consecutive amplitudes (1 KiB) per load/store instruction: index bits 0..5 are
//     lane bits, whatever the target qubit;
//   * target bits >= 6 are resolved in registers (a thread owns the 2 or 4 amplitudes of its group);
//   * target bits < 6 lie inside a wavefront and are resolved with wave64 shuffles: each lane keeps its own
//     amplitude, fetches its partners' with __shfl_xor and computes only its own row of the matrix;
//   * control bits >= 6 are removed from the enumeration (amplitudes with control = 0 are never touched);
//     control bits < 6 predicate lanes;
//   * each thread keeps U independent work items in flight (8 x 16 B loads per lane) to cover HBM latency.
// This replaces Gate.apply -> expand_gate -> dense mat-vec of the reference
// (simulators/dv_simulator/gates.py:44-54, numpy_quantum.py:243-247).

#include "qsv_internal.h"

#include <algorithm>
#include <array>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

namespace {

struct cplx {
    double re, im;
};

__device__ __forceinline__ amp_t cmul(cplx m, amp_t a) {
    amp_t r;
    r.x = m.re * a.x - m.im * a.y;
    r.y = m.re * a.y + m.im * a.x;
    return r;
}

// acc + m * a
__device__ __forceinline__ amp_t cfma(cplx m, amp_t a, amp_t acc) {
    amp_t r;
    r.x = fma(m.re, a.x, fma(-m.im, a.y, acc.x));
    r.y = fma(m.re, a.y, fma(m.im, a.x, acc.y));
    return r;
}

template <bool NT>
__device__ __forceinline__ amp_t ld(const amp_t *p) {
    if constexpr (NT)
        return __builtin_nontemporal_load(p);
    else
        return *p;
}

template <bool NT>
__device__ __forceinline__ void st(amp_t *p, amp_t v) {
    if constexpr (NT)
        __builtin_nontemporal_store(v, p);
    else
        *p = v;
}

__device__ __forceinline__ amp_t shfl_xor_amp(amp_t v, int lane_mask) {
    amp_t r;
    r.x = __shfl_xor(v.x, lane_mask, 64);
    r.y = __shfl_xor(v.y, lane_mask, 64);
    return r;
}

__device__ __forceinline__ uint64_t insert_zero(uint64_t w, int p) {
    const uint64_t low = w & ((1ull << p) - 1ull);
    return ((w >> p) << (p + 1)) | low;
}

// The positions are 32-bit words on purpose.  The argument struct lives in the kernarg segment; a run-time index into
// a BYTE array there makes the compiler fetch the byte with a vector load (gfx950 has no sub-dword scalar loads)
// followed by s_waitcnt vmcnt(0) in front of every amplitude load -- which also drains every amplitude load already in
// flight (k_rdm ran at 1.3-2.3 TB/s that way; rocprof: 60-70 % of the wave cycles parked).  A dword array is indexed
// with s_load_dword.
template <class Args>
__device__ __forceinline__ uint64_t deposit(uint64_t w, const Args &g) {
    for (int j = 0; j < g.nins; ++j) w = insert_zero(w, static_cast<int>(g.pos[j]));
    return w | g.or_mask;
}

// ----------------------------------------------------------------------------------------------------
// Dense 1- and 2-qubit gates (optionally controlled).
// ----------------------------------------------------------------------------------------------------
template <int KH, int KL, int U, bool NT>
__device__ __forceinline__ void dense_body(amp_t *__restrict__ a, const GateArgs &g) {
    constexpr int NH = 1 << KH, NL = 1 << KL, D = NH * NL;
    const int lane = threadIdx.x & 63;
    const bool lane_ok = (static_cast<uint32_t>(lane) & g.lane_ctrl) == g.lane_ctrl;

    // This lane's value of the low target bits.
    int l = 0;
#pragma unroll
    for (int j = 0; j < KL; ++j) l |= ((lane >> g.lbit[j]) & 1) << j;

    // coef[h][hp][x] = M[(h, l)][(hp, l ^ x)]: the matrix row(s) this lane computes.  For KL == 0 the
    // values are wave-uniform and stay in SGPRs; otherwise they are selected per lane once, up front.
    cplx coef[NH][NH][NL];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int hp = 0; hp < NH; ++hp)
#pragma unroll
            for (int x = 0; x < NL; ++x) {
                cplx c = {0.0, 0.0};
#pragma unroll
                for (int lc = 0; lc < NL; ++lc) {
                    const int row = (h << KL) | lc, col = (hp << KL) | (lc ^ x);
                    if (NL == 1 || l == lc) {
                        c.re = g.m[2 * (row * D + col)];
                        c.im = g.m[2 * (row * D + col) + 1];
                    }
                }
                coef[h][hp][x] = c;
            }

    constexpr uint64_t TILE = static_cast<uint64_t>(QSV_BLOCK) * U;
    const uint64_t ntiles = (g.W + TILE - 1) / TILE;
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        amp_t v[U][NH];
        uint64_t base[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // the U items of a thread sit 2^ubit work items apart (ubit = 8: back to back tiles of 256)
            // tile order: with `remap` = R the launch walks R contiguous regions of the register side by side
            // (workgroups are dealt round-robin over the 8 XCDs, so R = 8 gives every XCD its own region)
            const uint64_t tile_eff = (g.remap > 1 && ntiles % g.remap == 0)
                                          ? (tile % g.remap) * (ntiles / g.remap) + tile / g.remap
                                          : tile;
            const uint64_t t = tile_eff * QSV_BLOCK + threadIdx.x;
            const uint64_t w = U == 1 ? t
                                      : (((t >> g.ubit) * U + u) << g.ubit) | (t & ((1ull << g.ubit) - 1ull));
            ok[u] = w < g.W;  // W is a multiple of 64: uniform over the wave
            base[u] = deposit(w, g);
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                v[u][h] = amp_t{0.0, 0.0};
                if (ok[u] && lane_ok) v[u][h] = ld<NT>(a + base[u] + g.hoff[h]);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!ok[u]) continue;
            amp_t out[NH];
#pragma unroll
            for (int h = 0; h < NH; ++h) out[h] = amp_t{0.0, 0.0};
#pragma unroll
            for (int x = 0; x < NL; ++x)
#pragma unroll
                for (int hp = 0; hp < NH; ++hp) {
                    const amp_t p = (x == 0) ? v[u][hp] : shfl_xor_amp(v[u][hp], g.lxor[x]);
#pragma unroll
                    for (int h = 0; h < NH; ++h) out[h] = cfma(coef[h][hp][x], p, out[h]);
                }
            if (lane_ok) {
#pragma unroll
                for (int h = 0; h < NH; ++h) st<NT>(a + base[u] + g.hoff[h], out[h]);
            }
        }
    }
}

// k_dense: every amplitude is read and written once (2 * 16 * 2^n bytes).  k_dense_ctrl: the same body on a
// sub-space (controlled gates, SWAP as a pair exchange) -- a separate symbol so that profiles keep the
// full-traffic launches apart from the reduced-traffic ones.
template <int KH, int KL, int U, bool NT>
__global__ __launch_bounds__(QSV_BLOCK) void k_dense(amp_t *__restrict__ a, const GateArgs g) {
    dense_body<KH, KL, U, NT>(a, g);
}

template <int KH, int KL, int U, bool NT>
__global__ __launch_bounds__(QSV_BLOCK) void k_dense_ctrl(amp_t *__restrict__ a, const GateArgs g) {
    dense_body<KH, KL, U, NT>(a, g);
}

// ----------------------------------------------------------------------------------------------------
// Diagonal gates: one multiply per touched amplitude, natural (fully coalesced) enumeration.
// ----------------------------------------------------------------------------------------------------
template <int U, bool NT>
__global__ __launch_bounds__(QSV_BLOCK) void k_diag(amp_t *__restrict__ a, const DiagArgs g) {
    const int lane = threadIdx.x & 63;
    const bool lane_ok = (static_cast<uint32_t>(lane) & g.lane_ctrl) == g.lane_ctrl;
    const cplx d0 = {g.d[0], g.d[1]}, d1 = {g.d[2], g.d[3]}, d2 = {g.d[4], g.d[5]}, d3 = {g.d[6], g.d[7]};
    constexpr uint64_t TILE = static_cast<uint64_t>(QSV_BLOCK) * U;
    const uint64_t ntiles = (g.W + TILE - 1) / TILE;
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        amp_t v[U];
        uint64_t idx[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t tile_eff = (g.remap > 1 && ntiles % g.remap == 0)
                                          ? (tile % g.remap) * (ntiles / g.remap) + tile / g.remap
                                          : tile;
            const uint64_t w = tile_eff * TILE + static_cast<uint64_t>(u) * QSV_BLOCK + threadIdx.x;
            ok[u] = (w < g.W) && lane_ok;
            idx[u] = deposit(w, g);
            if (ok[u]) v[u] = ld<NT>(a + idx[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!ok[u]) continue;
            const int s0 = static_cast<int>((idx[u] >> g.b0) & 1ull);
            cplx d;
            if (g.b1 < 0) {
                d = s0 ? d1 : d0;
            } else {
                const int s1 = static_cast<int>((idx[u] >> g.b1) & 1ull);
                d = s0 ? (s1 ? d3 : d2) : (s1 ? d1 : d0);
            }
            st<NT>(a + idx[u], cmul(d, v[u]));
        }
    }
}

// Generic-K diagonal: table of 2^K complex numbers in device memory, staged through LDS.
__global__ __launch_bounds__(QSV_BLOCK) void k_diag_table(amp_t *__restrict__ a, uint64_t amps, int K,
                                                         const uint8_t *__restrict__ bitpos /*K, device*/,
                                                         const double *__restrict__ table) {
    __shared__ double tab[2 << QSV_MAX_K];
    __shared__ int bp[QSV_MAX_K];
    for (int i = threadIdx.x; i < (2 << K); i += blockDim.x) tab[i] = table[i];
    if (threadIdx.x < K) bp[threadIdx.x] = bitpos[threadIdx.x];
    __syncthreads();
    for (uint64_t i = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x; i < amps;
         i += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        int sel = 0;
        for (int j = 0; j < K; ++j) sel |= static_cast<int>((i >> bp[j]) & 1ull) << (K - 1 - j);
        const cplx d = {tab[2 * sel], tab[2 * sel + 1]};
        a[i] = cmul(d, a[i]);
    }
}

// ----------------------------------------------------------------------------------------------------
// Generic k-qubit dense gate (k <= 6) and the tiny-register path (n < 6): one thread per group, gathers
// with arbitrary strides.  Correct for every layout; not bandwidth-tuned (DESIGN.md "kernels").
// ----------------------------------------------------------------------------------------------------
struct GenericArgs {
    uint64_t W;
    int32_t K;
    uint8_t sorted_pos[QSV_MAX_K];  // ascending target bit positions (for deposit)
    uint8_t leg_pos[QSV_MAX_K];     // bit position of matrix leg j (leg 0 = most significant)
};

template <int K>
__global__ __launch_bounds__(QSV_BLOCK) void k_generic(amp_t *__restrict__ a, const GenericArgs g,
                                                      const double *__restrict__ M) {
    constexpr int D = 1 << K;
    uint64_t off[D];
#pragma unroll
    for (int c = 0; c < D; ++c) {
        uint64_t o = 0;
#pragma unroll
        for (int j = 0; j < K; ++j)
            if ((c >> (K - 1 - j)) & 1) o |= 1ull << g.leg_pos[j];
        off[c] = o;
    }
    for (uint64_t w = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x; w < g.W;
         w += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        uint64_t base = w;
#pragma unroll
        for (int j = 0; j < K; ++j) base = insert_zero(base, g.sorted_pos[j]);
        amp_t in[D];
#pragma unroll
        for (int c = 0; c < D; ++c) in[c] = a[base + off[c]];
#pragma unroll 1
        for (int r = 0; r < D; ++r) {
            amp_t acc = {0.0, 0.0};
#pragma unroll
            for (int c = 0; c < D; ++c) {
                const cplx m = {M[2 * (r * D + c)], M[2 * (r * D + c) + 1]};
                acc = cfma(m, in[c], acc);
            }
            // rows are written as they are produced: all inputs are already in registers
            uint64_t o = 0;
            for (int j = 0; j < K; ++j)
                if ((r >> (K - 1 - j)) & 1) o |= 1ull << g.leg_pos[j];
            a[base + o] = acc;
        }
    }
}

// ----------------------------------------------------------------------------------------------------
// Register-blocked k-qubit dense gate (k = 3..5: fused gate blocks, Gate(indices, matrix) with many legs).
// A thread owns the 2^k amplitudes of one group in registers (k = 5: 128 VGPRs), the six lowest NON-target bits
// are the lane bits, rows are produced one at a time with the matrix read through scalar loads (wave-uniform
// addresses -> s_load_dwordx16, no LDS) and stored in place.  2^k complex FMAs per amplitude: k = 5 is
// 8 flop/B, close to the fp64 ridge of the chip, so this kernel is bounded by HBM *and* the fp64 pipe.
// Device table layout behind M: [2^k x 2^k complex matrix][2^k uint64 amplitude offsets].
// ----------------------------------------------------------------------------------------------------
//
// Target bits below 6 (KL of them) are lane bits, and a wave must keep touching 64 consecutive amplitudes per
// instruction.  So the thread loads 2^k coalesced amplitudes over the KH high targets and KL *stand-in* high
// bits E_j instead, and the wave then swaps the roles of lane bit L_j and register bit e_j with KL butterfly
// stages of wave64 shuffles (a distributed transpose): afterwards every thread holds one complete group.  The
// same stages, applied to the outputs, restore the memory layout before the coalesced stores.
struct BigArgs {
    uint64_t W;
    int32_t nins;
    uint32_t pos[2 * QSV_MAX_K];  // ascending: high targets and stand-in bits
    uint64_t or_mask;            // unused (0); lets deposit() serve this struct too
    uint64_t w0;                 // first work item of this launch (registers beyond 2^32 work items take several)
    uint32_t regions;            // tile order (see GateArgs::remap)
    int32_t lbit[QSV_MAX_K];     // lane-bit position of low target j (register index bit j)
};

template <int D, int KL>
__device__ __forceinline__ void wave_transpose(amp_t (&x)[D], const BigArgs &g, int lane) {
#pragma unroll
    for (int j = 0; j < KL; ++j) {
        const int lb = g.lbit[j];
        const bool up = (lane >> lb) & 1;
#pragma unroll
        for (int c = 0; c < D; ++c) {
            if (c & (1 << j)) continue;
            const amp_t r0 = x[c], r1 = x[c | (1 << j)];
            const amp_t recv = shfl_xor_amp(up ? r0 : r1, 1 << lb);  // I keep entry e_j == my bit, trade the other
            x[c] = up ? recv : r0;
            x[c | (1 << j)] = up ? r1 : recv;
        }
    }
}

// Complex matrix rows in three real multiplications per entry instead of four ("3M"): with As = Ar + Ai prepared by the
// host and xs = xr + xi formed once per input amplitude,
//     S1 = sum Ar xr,  S2 = sum Ai xi,  S3 = sum As xs   ->   re = S1 - S2,  im = S3 - S1 - S2.
// A 32 x 32 complex product per amplitude is 8 flop/B -- at the chip's fp64 ridge -- so a quarter fewer FMAs is time
// (round 3; normwise error bound as for the four-multiplication form).  A row of the matrix is three planes of D doubles.
template <int D>
__device__ __forceinline__ amp_t row_product_3m(const double *__restrict__ row, const amp_t (&x)[D], const double (&xs)[D]) {
    double s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
    for (int c = 0; c < D; ++c) {
        s1 = fma(row[c], x[c].x, s1);
        s2 = fma(row[D + c], x[c].y, s2);
        s3 = fma(row[2 * D + c], xs[c], s3);
    }
    return amp_t{s1 - s2, s3 - s1 - s2};
}

template <int K, int KL, bool NT, bool M3 = false>
__global__ __launch_bounds__(QSV_BLOCK) void k_dense_big(amp_t *__restrict__ a, const BigArgs g,
                                                        const double *__restrict__ M,
                                                        const uint64_t *__restrict__ hoff) {
    constexpr int D = 1 << K;
    // straight-line body (a grid-stride loop here cost the k = 5 transposed variants 4x: registers live across
    // the back edge); registers beyond 2^32 work items are covered by several launches with a work-item offset
    // tile order as in k_dense: `regions` contiguous pieces of this launch's range walked side by side
    const uint64_t tile = (g.regions > 1 && gridDim.x % g.regions == 0)
                              ? (blockIdx.x % g.regions) * (gridDim.x / g.regions) + blockIdx.x / g.regions
                              : blockIdx.x;
    const uint64_t w = g.w0 + tile * static_cast<uint64_t>(QSV_BLOCK) + threadIdx.x;
    if (w >= g.W) return;  // W and w0 are multiples of 64 whenever KL > 0: whole waves leave together
    const uint64_t base = deposit(w, g);
    amp_t x[D];
#pragma unroll
    for (int c = 0; c < D; ++c) x[c] = ld<NT>(a + base + hoff[c]);
    if constexpr (KL == 0) {
        double xs[M3 ? D : 1];
        if constexpr (M3) {
#pragma unroll
            for (int c = 0; c < D; ++c) xs[c] = x[c].x + x[c].y;
        }
#pragma unroll 1
        for (int r = 0; r < D; ++r) {
            amp_t acc = {0.0, 0.0};
            if constexpr (M3) {
                acc = row_product_3m<D>(M + 3 * D * r, x, xs);
            } else {
                const double *row = M + 2 * D * r;
#pragma unroll
                for (int c = 0; c < D; ++c) acc = cfma(cplx{row[2 * c], row[2 * c + 1]}, x[c], acc);
            }
            st<NT>(a + base + hoff[r], acc);  // in place: every input of this group is already in registers
        }
    } else {
        static_assert(!M3, "the shuffle form keeps two register arrays: no room for a third");
        const int lane = threadIdx.x & 63;
        wave_transpose<D, KL>(x, g, lane);
        amp_t y[D];
#pragma unroll
        for (int r = 0; r < D; ++r) {
            const double *row = M + 2 * D * r;
            amp_t acc = {0.0, 0.0};
#pragma unroll
            for (int c = 0; c < D; ++c) acc = cfma(cplx{row[2 * c], row[2 * c + 1]}, x[c], acc);
            y[r] = acc;
        }
        wave_transpose<D, KL>(y, g, lane);
#pragma unroll
        for (int c = 0; c < D; ++c) st<NT>(a + base + hoff[c], y[c]);
    }
}


// ----------------------------------------------------------------------------------------------------
// Register-blocked k-qubit dense gate with target bits below 6, second form: the low target bits are brought into
// registers WITHOUT wave shuffles and without a second register array for the results.
//
// A low target bit L still gets a stand-in high bit E (as above), so that whatever a wave-instruction touches is made
// of whole 128-byte lines.  But only the bits 0..2 of the index live INSIDE a line.  For a target on lane bit 3, 4 or
// 5 ("A" bits) the exchange of roles between L and E is pure address arithmetic: lane l fetches, for register value
// v, the amplitude in row E := (bit L of l), column (l with bit L := v).  Each wave-instruction then reads 8 whole
// lines from 2^|A| rows instead of 8 consecutive ones -- the same number of lines -- and the thread owns both values
// of bit L at once.  Nothing moves between lanes.
// Targets on lane bits 0..2 ("B" bits, at most three) need a real transpose among the 2^|B| neighbouring lanes that
// share a line.  It goes through LDS, 2^|B| rows of 64 amplitudes (at most 8 KiB per wave) at a time: lane l writes
// row s at column l ^ dep(s) and reads row (its own B bits) at column l ^ dep(t) -- an XOR swizzle that makes both
// directions conflict-free for 16-byte accesses (a 16-lane group of ds_read_b128 always sees 16 different columns
// mod 16, and rows are 64 slots apart).  Results take the same road back one row group at a time, straight from the
// accumulator, so the thread holds the 2^K inputs (K = 5: 128 VGPRs) and nothing else: 3 waves per SIMD instead of
// one, and the next wave's loads overlap this wave's arithmetic.
// ----------------------------------------------------------------------------------------------------
struct LdsArgs {
    uint64_t W;
    int32_t nins;
    uint32_t pos[2 * QSV_MAX_K];  // ascending: high targets and stand-in bits
    uint64_t or_mask;            // unused (0); lets deposit() serve this struct too
    uint64_t w0;                 // first work item of this launch
    uint32_t regions;            // tile order (see GateArgs::remap)
    uint32_t amask;              // lane bits of the A targets
    int32_t abit[3], aE[3];      // A target j: lane bit, stand-in bit
    int32_t na;                  // number of A targets
    uint32_t bdep[8];            // dep(v): the KB bits of v spread onto the lane bits of the B targets
    uint32_t bmask;              // lane bits of the B targets
};

// The LDS rows are private to a wave (a wave exchanges data with itself only), and the LDS executes one wave's
// instructions in order: no workgroup barrier is needed, only the compiler must keep the program order of the
// accesses (it does: they may alias) -- wave_sync() marks the spots.
__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_wave_barrier(); }

template <int K, int KB, bool NT, bool REALM, int BLOCK, bool M3 = false>
__global__ __launch_bounds__(BLOCK) void k_dense_lds(amp_t *__restrict__ a, const LdsArgs g,
                                                     const double *__restrict__ M,
                                                     const uint64_t *__restrict__ hoff) {
    constexpr int D = 1 << K, NB = 1 << KB;
    __shared__ amp_t tiles[KB > 0 ? NB * BLOCK : 1];
    amp_t *tile = tiles + (KB > 0 ? NB * 64 * (threadIdx.x >> 6) : 0);
    const uint64_t tile_id = (g.regions > 1 && gridDim.x % g.regions == 0)
                                 ? (blockIdx.x % g.regions) * (gridDim.x / g.regions) + blockIdx.x / g.regions
                                 : blockIdx.x;
    const uint64_t w = g.w0 + tile_id * static_cast<uint64_t>(BLOCK) + threadIdx.x;
    if (w >= g.W) return;  // W and w0 are multiples of 64: whole waves leave together
    const uint32_t lane = threadIdx.x & 63;
    // row of this lane: its A bits move from the column to the stand-in bits
    uint64_t base = deposit(w, g) & ~static_cast<uint64_t>(g.amask);
    for (int j = 0; j < g.na; ++j) base |= static_cast<uint64_t>((lane >> g.abit[j]) & 1u) << g.aE[j];
    amp_t x[D];
#pragma unroll
    for (int c = 0; c < D; ++c) x[c] = ld<NT>(a + base + hoff[c]);
    uint32_t my_row = 0;  // this lane's B bits, as a row number
    if constexpr (KB > 0) {
#pragma unroll
        for (int v = 1; v < NB; ++v)
            if ((lane & g.bmask) == g.bdep[v]) my_row = v;
#pragma unroll
        for (int o = 0; o < D / NB; ++o) {
            wave_sync();
#pragma unroll
            for (int s = 0; s < NB; ++s) tile[s * 64 + (lane ^ g.bdep[s])] = x[o * NB + s];
            wave_sync();
#pragma unroll
            for (int t = 0; t < NB; ++t) x[o * NB + t] = tile[my_row * 64 + (lane ^ g.bdep[t])];
        }
    }
    // every thread now owns one complete group; rows of the matrix come through scalar loads
    double xs[M3 ? D : 1];
    if constexpr (M3) {
        static_assert(!REALM, "a real matrix needs two multiplications per entry anyway");
#pragma unroll
        for (int c = 0; c < D; ++c) xs[c] = x[c].x + x[c].y;
    }
#pragma unroll 1
    for (int o = 0; o < D / NB; ++o) {
        if constexpr (KB > 0) wave_sync();
#pragma unroll
        for (int t = 0; t < NB; ++t) {
            const int r = o * NB + t;
            amp_t acc = {0.0, 0.0};
            if constexpr (M3) {
                acc = row_product_3m<D>(M + 3 * D * r, x, xs);
            } else if constexpr (REALM) {
                const double *row = M + D * r;
#pragma unroll
                for (int c = 0; c < D; ++c) {
                    acc.x = fma(row[c], x[c].x, acc.x);
                    acc.y = fma(row[c], x[c].y, acc.y);
                }
            } else {
                const double *row = M + 2 * D * r;
#pragma unroll
                for (int c = 0; c < D; ++c) acc = cfma(cplx{row[2 * c], row[2 * c + 1]}, x[c], acc);
            }
            if constexpr (KB > 0) tile[my_row * 64 + (lane ^ g.bdep[t])] = acc;
            else st<NT>(a + base + hoff[r], acc);  // in place: every input of this group is already in registers
        }
        if constexpr (KB > 0) {
            wave_sync();
#pragma unroll
            for (int s = 0; s < NB; ++s) st<NT>(a + base + hoff[o * NB + s], tile[s * 64 + (lane ^ g.bdep[s])]);
        }
    }
}


// ----------------------------------------------------------------------------------------------------
// Fused 5-qubit blocks as a SEQUENCE of their source gates (round 3).  A dense 32 x 32 complex block costs 4 x 1024 real
// FMAs per amplitude group -- the one shape of the gate path that is bound by arithmetic (the vector pipe is 81 % busy at
// the clock its power draw leaves it: profiles/r03_k5_sq_counters.txt).  But a fused block IS a product of a few 1- and
// 2-qubit gates (5.9 on average on the benchmark circuit): applied one after the other to the 32 amplitudes a thread
// already holds in registers they cost 256 FMAs per 1-qubit gate and 512 per 2-qubit gate, ~2 400 per block instead of
// 4 096, and nothing but the gates' own small matrices comes through the scalar cache.  Loads, the exchange of low target bits
// (address arithmetic for lane bits 3..5, the XOR-swizzled LDS rows for bits 0..2) and stores are k_dense_big's /
// k_dense_lds's; only the arithmetic in the middle differs.  The register index is runtime data of the gate list, the
// register ARRAY must be indexed statically: one unrolled body per register bit (1-qubit gates) and per pair of register
// bits (2-qubit gates, leg 0 canonicalised onto the higher bit by the host), selected by a wave-uniform switch.
// ----------------------------------------------------------------------------------------------------
struct SeqGate {
    int32_t code;      // 0..4: 1-qubit gate on register bit `code`; 5 + p: 2-qubit gate on the p-th pair (hi, lo), hi > lo
    int32_t pad[3];
    double m[32];      // 2 x 2 or 4 x 4 row-major complex; 2-qubit: matrix index bit 1 <-> register bit hi
};
constexpr int SEQ_MAX_GATES = 48;
static int seq_max_work() {
    static const int v = [] {
        const char *e = getenv("QSV_SEQUENCE_WORK");
        return e ? atoi(e) : 0;      // measured (profiles/r03_sequence_blocks.txt): no faster than the dense block
    }();
    return v;
}

template <int J, int D = 32>
__device__ __forceinline__ void seq_apply1(amp_t (&x)[D], const double *__restrict__ m) {
    const cplx m00{m[0], m[1]}, m01{m[2], m[3]}, m10{m[4], m[5]}, m11{m[6], m[7]};
#pragma unroll
    for (int c = 0; c < D; ++c) {
        if (c & (1 << J)) continue;
        const amp_t a0 = x[c], a1 = x[c | (1 << J)];
        x[c] = cfma(m01, a1, cmul(m00, a0));
        x[c | (1 << J)] = cfma(m11, a1, cmul(m10, a0));
    }
}

template <int HI, int LO, int D = 32>
__device__ __forceinline__ void seq_apply2(amp_t (&x)[D], const double *__restrict__ m) {
#pragma unroll
    for (int c = 0; c < D; ++c) {
        if (c & ((1 << HI) | (1 << LO))) continue;
        const amp_t in[4] = {x[c], x[c | (1 << LO)], x[c | (1 << HI)], x[c | (1 << HI) | (1 << LO)]};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            amp_t acc = cmul(cplx{m[8 * r], m[8 * r + 1]}, in[0]);
#pragma unroll
            for (int cc = 1; cc < 4; ++cc) acc = cfma(cplx{m[8 * r + 2 * cc], m[8 * r + 2 * cc + 1]}, in[cc], acc);
            x[c | ((r >> 1) << HI) | ((r & 1) << LO)] = acc;
        }
    }
}

__device__ __forceinline__ void seq_run(amp_t (&x)[32], const SeqGate *__restrict__ gates, int n_gates) {
#pragma unroll 1
    for (int g = 0; g < n_gates; ++g) {
        const double *m = gates[g].m;
        switch (gates[g].code) {     // wave-uniform (scalar loads)
            case 0: seq_apply1<0>(x, m); break;
            case 1: seq_apply1<1>(x, m); break;
            case 2: seq_apply1<2>(x, m); break;
            case 3: seq_apply1<3>(x, m); break;
            case 4: seq_apply1<4>(x, m); break;
            case 5: seq_apply2<1, 0>(x, m); break;
            case 6: seq_apply2<2, 0>(x, m); break;
            case 7: seq_apply2<2, 1>(x, m); break;
            case 8: seq_apply2<3, 0>(x, m); break;
            case 9: seq_apply2<3, 1>(x, m); break;
            case 10: seq_apply2<3, 2>(x, m); break;
            case 11: seq_apply2<4, 0>(x, m); break;
            case 12: seq_apply2<4, 1>(x, m); break;
            case 13: seq_apply2<4, 2>(x, m); break;
            default: seq_apply2<4, 3>(x, m); break;
        }
    }
}

// every target on bit 6 or higher: k_dense_big<5, 0>'s loads and in-place stores around the gate sequence
template <bool NT>
__global__ __launch_bounds__(QSV_BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_seq_big(amp_t *__restrict__ a, const BigArgs g, const SeqGate *__restrict__ gates,
                                                      int n_gates, const uint64_t *__restrict__ hoff) {
    constexpr int D = 32;
    const uint64_t tile = (g.regions > 1 && gridDim.x % g.regions == 0)
                              ? (blockIdx.x % g.regions) * (gridDim.x / g.regions) + blockIdx.x / g.regions
                              : blockIdx.x;
    const uint64_t w = g.w0 + tile * static_cast<uint64_t>(QSV_BLOCK) + threadIdx.x;
    if (w >= g.W) return;
    const uint64_t base = deposit(w, g);
    amp_t x[D];
#pragma unroll
    for (int c = 0; c < D; ++c) x[c] = ld<NT>(a + base + hoff[c]);
    seq_run(x, gates, n_gates);
#pragma unroll
    for (int c = 0; c < D; ++c) st<NT>(a + base + hoff[c], x[c]);
}

// targets below bit 6: k_dense_lds<5, KB>'s addressing and LDS exchange around the gate sequence
template <int KB, bool NT>
__global__ __launch_bounds__(QSV_BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_seq_lds(amp_t *__restrict__ a, const LdsArgs g, const SeqGate *__restrict__ gates,
                                                      int n_gates, const uint64_t *__restrict__ hoff) {
    constexpr int D = 32, NB = 1 << KB;
    __shared__ amp_t tiles[KB > 0 ? NB * QSV_BLOCK : 1];
    amp_t *tile = tiles + (KB > 0 ? NB * 64 * (threadIdx.x >> 6) : 0);
    const uint64_t tile_id = (g.regions > 1 && gridDim.x % g.regions == 0)
                                 ? (blockIdx.x % g.regions) * (gridDim.x / g.regions) + blockIdx.x / g.regions
                                 : blockIdx.x;
    const uint64_t w = g.w0 + tile_id * static_cast<uint64_t>(QSV_BLOCK) + threadIdx.x;
    if (w >= g.W) return;  // W and w0 are multiples of 64: whole waves leave together
    const uint32_t lane = threadIdx.x & 63;
    uint64_t base = deposit(w, g) & ~static_cast<uint64_t>(g.amask);
    for (int j = 0; j < g.na; ++j) base |= static_cast<uint64_t>((lane >> g.abit[j]) & 1u) << g.aE[j];
    amp_t x[D];
#pragma unroll
    for (int c = 0; c < D; ++c) x[c] = ld<NT>(a + base + hoff[c]);
    uint32_t my_row = 0;
    if constexpr (KB > 0) {
#pragma unroll
        for (int v = 1; v < NB; ++v)
            if ((lane & g.bmask) == g.bdep[v]) my_row = v;
#pragma unroll
        for (int o = 0; o < D / NB; ++o) {
            wave_sync();
#pragma unroll
            for (int s = 0; s < NB; ++s) tile[s * 64 + (lane ^ g.bdep[s])] = x[o * NB + s];
            wave_sync();
#pragma unroll
            for (int t = 0; t < NB; ++t) x[o * NB + t] = tile[my_row * 64 + (lane ^ g.bdep[t])];
        }
    }
    seq_run(x, gates, n_gates);
    if constexpr (KB > 0) {
#pragma unroll
        for (int o = 0; o < D / NB; ++o) {
            wave_sync();
#pragma unroll
            for (int t = 0; t < NB; ++t) tile[my_row * 64 + (lane ^ g.bdep[t])] = x[o * NB + t];
            wave_sync();
#pragma unroll
            for (int s = 0; s < NB; ++s) st<NT>(a + base + hoff[o * NB + s], tile[s * 64 + (lane ^ g.bdep[s])]);
        }
    } else {
#pragma unroll
        for (int c = 0; c < D; ++c) st<NT>(a + base + hoff[c], x[c]);
    }
}

// ----------------------------------------------------------------------------------------------------
// 6-qubit dense gates.  2^6 complex FMAs per amplitude are 16 flop/B: above the fp64 ridge of the chip (~10 flop/B),
// so this one kernel of the gate path is bounded by arithmetic, not by HBM, and a 64 x 64 matrix times a (64 x groups)
// panel is a GEMM: it runs on the f64 matrix cores (v_mfma_f64_16x16x4_f64: A[i = lane & 15][k = lane >> 4],
// B[k = lane >> 4][j = lane & 15], D col = lane & 15, row = (lane >> 4) + 4 reg; same peak as the vector pipe, but 64
// VGPRs of inputs per lane instead of 256, so two waves per SIMD overlap loads with arithmetic -- the register-blocked
// vector form ran at 8.5 TFLOP/s, one wave per SIMD, stalled on its 1 KiB matrix rows).
// A wave owns 16 groups (the 16 lowest free index values): lane (li, lk) holds amplitudes 4 s + lk, s = 0..15, of group
// li -- lk runs over the two lowest target bits, so whatever the targets are a wave-instruction touches whole 128-byte
// lines (half lines when bits 0, 1 and 2 are all targets).
// The matrix sits in LDS column-major (real and imaginary planes), a complex product is four real MFMAs, a real
// matrix needs two.  Loads and stores are four 256-byte runs per wave-instruction: whole 128-byte lines.
// ----------------------------------------------------------------------------------------------------
typedef double f64x4 __attribute__((ext_vector_type(4)));

struct Mfma6Args {
    uint64_t W;          // groups = amps / 64
    uint64_t or_mask;    // unused (0); lets deposit() serve this struct too
    int32_t nins;        // 6
    uint32_t pos[8];     // ascending target bits (all >= 4)
};

template <int K, bool NT, bool REALM, bool M3 = false>
__global__ __launch_bounds__(QSV_BLOCK) __attribute__((amdgpu_waves_per_eu(2, (K == 6 && !REALM) ? 2 : 3))) void k_dense_mfma(
    amp_t *__restrict__ a, const Mfma6Args g, const double *__restrict__ Mcol,  // [plane][col][row]
    const uint64_t *__restrict__ hoff) {
    constexpr int D = 1 << K, SL = D / 4 /* k-slices */, RT = D / 16 /* row tiles */;
    extern __shared__ __attribute__((aligned(16))) char smem6[];
    double *mre = reinterpret_cast<double *>(smem6);
    double *mim = mre + D * D;
    uint64_t *off = reinterpret_cast<uint64_t *>(mre + (REALM ? 1 : 2) * D * D);
    for (int i = threadIdx.x; i < (REALM ? 1 : 2) * D * D; i += QSV_BLOCK) mre[i] = Mcol[i];
    if (threadIdx.x < D) off[threadIdx.x] = hoff[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    const uint64_t wave = blockIdx.x * (QSV_BLOCK / 64) + (threadIdx.x >> 6);
    const uint64_t waves = static_cast<uint64_t>(gridDim.x) * (QSV_BLOCK / 64);
    const uint64_t tiles = g.W / 16;
    if (wave >= tiles) return;
    auto fetch = [&](amp_t (&x)[SL], uint64_t tile) {
        const uint64_t base = deposit(tile * 16 + li, g);
#pragma unroll
        for (int s = 0; s < SL; ++s) x[s] = ld<NT>(a + base + off[4 * s + lk]);
    };
    // complex matrices in three real MFMAs per (slice, row tile) instead of four (see row_product_3m; As = Ar + Ai and
    // xs = xr + xi are one VALU add each, next to 64-cycle MFMAs): S1 += Ar xr, S2 += Ai xi, S3 += As xs.  The row tiles
    // are taken in two halves so that the 3 x RT / 2 accumulators stay at 48 registers.
    auto apply3 = [&](const amp_t (&x)[SL], uint64_t tile) {
        constexpr int HT = RT / 2;
        const uint64_t base = deposit(tile * 16 + li, g);
#pragma unroll 1
        for (int h = 0; h < 2; ++h) {   // not unrolled: interleaved halves need the registers of both
            f64x4 s1[HT], s2[HT], s3[HT];
#pragma unroll
            for (int t = 0; t < HT; ++t) s1[t] = s2[t] = s3[t] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < SL; ++s) {
                double are[HT], aim[HT], asum[HT];
                const double xsum = x[s].x + x[s].y;   // once per half: cheaper than 32 registers held across the tile
#pragma unroll
                for (int t = 0; t < HT; ++t) {
                    are[t] = mre[(4 * s + lk) * D + 16 * (h * HT + t) + li];
                    aim[t] = mim[(4 * s + lk) * D + 16 * (h * HT + t) + li];
                    asum[t] = are[t] + aim[t];
                }
#pragma unroll
                for (int t = 0; t < HT; ++t) s1[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(are[t], x[s].x, s1[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < HT; ++t) s2[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(aim[t], x[s].y, s2[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < HT; ++t) s3[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(asum[t], xsum, s3[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            // in place: the wave's loads of this tile are complete (their values are MFMA operands above) before the
            // first store issues; the second half reads only registers
#pragma unroll
            for (int t = 0; t < HT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    st<NT>(a + base + off[16 * (h * HT + t) + lk + 4 * r],
                           amp_t{s1[t][r] - s2[t][r], s3[t][r] - s1[t][r] - s2[t][r]});
        }
    };
    auto apply4 = [&](const amp_t (&x)[SL], uint64_t tile) {
        f64x4 cre[RT], cim[RT];
#pragma unroll
        for (int t = 0; t < RT; ++t) cre[t] = cim[t] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < SL; ++s) {
            double are[RT], aim[RT];
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                are[t] = mre[(4 * s + lk) * D + 16 * t + li];
                if constexpr (!REALM) aim[t] = mim[(4 * s + lk) * D + 16 * t + li];
            }
#pragma unroll
            for (int t = 0; t < RT; ++t) {  // dependent updates of one accumulator stay 2 RT instructions apart
                cre[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(are[t], x[s].x, cre[t], 0, 0, 0);
                cim[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(are[t], x[s].y, cim[t], 0, 0, 0);
            }
            if constexpr (!REALM) {
#pragma unroll
                for (int t = 0; t < RT; ++t) {
                    cre[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(-aim[t], x[s].y, cre[t], 0, 0, 0);
                    cim[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(aim[t], x[s].x, cim[t], 0, 0, 0);
                }
            }
            // keep the matrix reads of slice s + 1 behind the MFMAs of slice s: hoisted to the top (the scheduler's
            // choice without this fence) the slices' operands need up to 256 VGPRs and spill
            __builtin_amdgcn_sched_barrier(0);
        }
        // in place: the wave has read every amplitude of its 16 groups before the first of these stores can issue
        const uint64_t base = deposit(tile * 16 + li, g);
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                st<NT>(a + base + off[16 * t + lk + 4 * r], amp_t{cre[t][r], cim[t][r]});
    };
    auto apply = [&](const amp_t (&x)[SL], uint64_t tile) {
        if constexpr (M3) apply3(x, tile);
        else apply4(x, tile);
    };
    // A ring of input buffers, no copies between them: the next tiles' loads are in flight while this tile's MFMAs run
    // (K = 6: 256 MFMAs = 7 us per tile, one tile ahead; K = 5: 64 MFMAs = 1.7 us, two ahead).  Every fetch is
    // unconditional -- past the end a wave re-reads its first tile and drops it -- because the compiler cannot count
    // loads issued under a branch and would wait for all of them (see k_rdm).
    // (a real 64 x 64 matrix halves the MFMAs: that variant is bound by memory and runs three waves per SIMD without a
    // ring -- 1.74-2.04 ms by placement, the same as two waves with one (1.74-2.01), in 156 instead of 230 registers)
    constexpr int NBUF = K == 6 ? (REALM ? 1 : 2) : 3;
    amp_t x[NBUF][SL];
    if constexpr (NBUF == 1) {
        for (uint64_t tile = wave; tile < tiles; tile += waves) {
            fetch(x[0], tile);
            apply(x[0], tile);
        }
        return;
    }
#pragma unroll
    for (int b = 0; b < NBUF - 1; ++b) {
        const uint64_t t = wave + b * waves;
        fetch(x[b], t < tiles ? t : wave);
    }
    for (uint64_t tile = wave; tile < tiles;) {
#pragma unroll
        for (int b = 0; b < NBUF; ++b) {
            const uint64_t t = tile + (NBUF - 1) * waves;
            fetch(x[(b + NBUF - 1) % NBUF], t < tiles ? t : wave);
            apply(x[b], tile);
            tile += waves;
            if (tile >= tiles) break;
        }
    }
}

// ----------------------------------------------------------------------------------------------------
// Reductions, measurement, insertion, permutation, fills.
// ----------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Block-wide sum of two doubles; result valid in thread 0.
__device__ __forceinline__ void block_sum2(double &x, double &y) {
    __shared__ double sx[QSV_BLOCK / 64], sy[QSV_BLOCK / 64];
    x = wave_sum(x);
    y = wave_sum(y);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        sx[wave] = x;
        sy[wave] = y;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        x = 0.0;
        y = 0.0;
        for (int i = 0; i < QSV_BLOCK / 64; ++i) {
            x += sx[i];
            y += sy[i];
        }
    }
}

// partials[2*block + s] = sum over this block's pairs of |eig_s[0] a0 + eig_s[1] a1|^2
__global__ __launch_bounds__(QSV_BLOCK) void k_measure_probs(const amp_t *__restrict__ a, uint64_t pairs, int bit,
                                                            cplx e00, cplx e01, cplx e10, cplx e11,
                                                            double *__restrict__ partials) {
    double p0 = 0.0, p1 = 0.0;
    const uint64_t s = 1ull << bit;
    for (uint64_t w = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x; w < pairs;
         w += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        const uint64_t i0 = insert_zero(w, bit);
        const amp_t a0 = a[i0], a1 = a[i0 + s];
        const amp_t r0 = cfma(e01, a1, cmul(e00, a0));
        const amp_t r1 = cfma(e11, a1, cmul(e10, a0));
        p0 += r0.x * r0.x + r0.y * r0.y;
        p1 += r1.x * r1.x + r1.y * r1.y;
    }
    block_sum2(p0, p1);
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x] = p0;
        partials[2 * blockIdx.x + 1] = p1;
    }
}

// out[w] = scale * (e0 a[i0] + e1 a[i1]): the (n-1)-qubit post-measurement ket.
__global__ __launch_bounds__(QSV_BLOCK) void k_collapse(const amp_t *__restrict__ a, amp_t *__restrict__ out,
                                                       uint64_t pairs, int bit, cplx e0, cplx e1, double scale) {
    const uint64_t s = 1ull << bit;
    for (uint64_t w = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x; w < pairs;
         w += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        const uint64_t i0 = insert_zero(w, bit);
        amp_t r = cfma(e1, a[i0 + s], cmul(e0, a[i0]));
        r.x *= scale;
        r.y *= scale;
        out[w] = r;
    }
}

// out[j] = amp[bit(j)] * a[j with the bit removed]: kron(state, new) + move (gates.py:149-152).
__global__ __launch_bounds__(QSV_BLOCK) void k_insert(const amp_t *__restrict__ a, amp_t *__restrict__ out,
                                                     uint64_t out_amps, int bit, cplx c0, cplx c1) {
    for (uint64_t j = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x; j < out_amps;
         j += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        const uint64_t low = j & ((1ull << bit) - 1ull);
        const uint64_t src = ((j >> (bit + 1)) << bit) | low;
        out[j] = cmul(((j >> bit) & 1ull) ? c1 : c0, a[src]);
    }
}

struct PermArgs {
    int32_t n;
    uint8_t src_bit[64];  // bit j of the destination index comes from bit src_bit[j] of the source index
};

__global__ __launch_bounds__(QSV_BLOCK) void k_permute(const amp_t *__restrict__ a, amp_t *__restrict__ out,
                                                      uint64_t amps, const PermArgs g) {
    for (uint64_t j = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x; j < amps;
         j += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        uint64_t src = 0;
        for (int b = 0; b < g.n; ++b) src |= ((j >> b) & 1ull) << g.src_bit[b];
        out[j] = a[src];
    }
}

__global__ __launch_bounds__(QSV_BLOCK) void k_norm2(const amp_t *__restrict__ a, uint64_t amps,
                                                    double *__restrict__ partials) {
    double s = 0.0, unused = 0.0;
    for (uint64_t i = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x; i < amps;
         i += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        const amp_t v = a[i];
        s += v.x * v.x + v.y * v.y;
    }
    block_sum2(s, unused);
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x] = s;
        partials[2 * blockIdx.x + 1] = 0.0;
    }
}

__global__ __launch_bounds__(QSV_BLOCK) void k_inner(const amp_t *__restrict__ a, const amp_t *__restrict__ b,
                                                    uint64_t amps, double *__restrict__ partials) {
    double re = 0.0, im = 0.0;
    for (uint64_t i = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x; i < amps;
         i += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        const amp_t x = a[i], y = b[i];
        re += x.x * y.x + x.y * y.y;  // conj(x) * y
        im += x.x * y.y - x.y * y.x;
    }
    block_sum2(re, im);
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x] = re;
        partials[2 * blockIdx.x + 1] = im;
    }
}

// <psi| P |psi> for a Pauli string: P|i> = i^{nY} (-1)^{popcount(i & zmask)} |i ^ xmask>  (Y = i X Z).
// partials[2b], [2b+1] = real and imaginary part of this block's share of sum_i conj(psi[i ^ xmask]) sign(i) psi[i];
// the factor i^{nY} is applied on the host.
__global__ __launch_bounds__(QSV_BLOCK) void k_expect_pauli(const amp_t *__restrict__ a, uint64_t amps,
                                                           uint64_t xmask, uint64_t zmask,
                                                           double *__restrict__ partials) {
    double re = 0.0, im = 0.0;
    for (uint64_t i = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x; i < amps;
         i += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        const amp_t x = a[i ^ xmask], y = a[i];
        const double s = (__popcll(i & zmask) & 1) ? -1.0 : 1.0;
        re += s * (x.x * y.x + x.y * y.y);  // conj(x) * y
        im += s * (x.x * y.y - x.y * y.x);
    }
    block_sum2(re, im);
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x] = re;
        partials[2 * blockIdx.x + 1] = im;
    }
}

// Sampling, pass 1: chunk_sums[c] = sum of |amp|^2 over chunk c (SAMPLE_CHUNK consecutive amplitudes per workgroup).
constexpr int SAMPLE_CHUNK = 4096;

__global__ __launch_bounds__(QSV_BLOCK) void k_chunk_sums(const amp_t *__restrict__ a, uint64_t amps,
                                                         double *__restrict__ chunk_sums) {
    const uint64_t chunks = (amps + SAMPLE_CHUNK - 1) / SAMPLE_CHUNK;
    for (uint64_t c = blockIdx.x; c < chunks; c += gridDim.x) {
        double s = 0.0, unused = 0.0;
        // thread t owns SAMPLE_CHUNK / QSV_BLOCK consecutive amplitudes: the same split pass 2 walks
        constexpr int PER = SAMPLE_CHUNK / QSV_BLOCK;
        const uint64_t first = c * SAMPLE_CHUNK + static_cast<uint64_t>(threadIdx.x) * PER;
        for (int k = 0; k < PER; ++k)
            if (first + k < amps) {
                const amp_t v = a[first + k];
                s += v.x * v.x + v.y * v.y;
            }
        __syncthreads();  // block_sum2 reuses its shared scratch across iterations
        block_sum2(s, unused);
        if (threadIdx.x == 0) chunk_sums[c] = s;
    }
}

// Sampling, pass 2: one workgroup per shot walks its chunk and returns the first index whose running sum of
// |amp|^2 exceeds `residual` (clamped to the chunk's last amplitude against rounding).
__global__ __launch_bounds__(QSV_BLOCK) void k_sample_in_chunk(const amp_t *__restrict__ a, uint64_t amps,
                                                              const uint64_t *__restrict__ chunk_of_shot,
                                                              const double *__restrict__ residual_of_shot,
                                                              uint64_t *__restrict__ out) {
    __shared__ double part[QSV_BLOCK];
    constexpr int PER = SAMPLE_CHUNK / QSV_BLOCK;
    const uint64_t c = chunk_of_shot[blockIdx.x];
    const double residual = residual_of_shot[blockIdx.x];
    const uint64_t first = c * SAMPLE_CHUNK + static_cast<uint64_t>(threadIdx.x) * PER;
    double mine[PER];
    double s = 0.0;
    for (int k = 0; k < PER; ++k) {
        double p = 0.0;
        if (first + k < amps) {
            const amp_t v = a[first + k];
            p = v.x * v.x + v.y * v.y;
        }
        mine[k] = p;
        s += p;
    }
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double run = 0.0;
        int t = 0;
        for (; t < QSV_BLOCK - 1; ++t) {
            if (run + part[t] > residual) break;
            run += part[t];
        }
        part[0] = run;                       // sum before thread t
        part[1] = static_cast<double>(t);    // the thread that holds the crossing
    }
    __syncthreads();
    const int owner = static_cast<int>(part[1]);
    if (threadIdx.x == owner) {
        double run = part[0];
        int k = 0;
        for (; k < PER - 1; ++k) {
            if (run + mine[k] > residual) break;
            run += mine[k];
        }
        uint64_t idx = first + k;
        if (idx >= amps) idx = amps - 1;
        out[blockIdx.x] = idx;
    }
}

__global__ void k_gather_prob(const amp_t *__restrict__ a, const uint64_t *__restrict__ idx, int count,
                              double *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) {
        const amp_t v = a[idx[i]];
        out[i] = v.x * v.x + v.y * v.y;
    }
}

__global__ __launch_bounds__(QSV_BLOCK) void k_scale(amp_t *__restrict__ a, uint64_t amps, cplx c) {
    for (uint64_t i = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x; i < amps;
         i += static_cast<uint64_t>(gridDim.x) * blockDim.x)
        a[i] = cmul(c, a[i]);
}

// Register-to-register copy (qsv_copy; also the "what does a plain copy reach on this box" leg of bench.py), every
// wave-instruction a whole 1 KiB segment, nontemporal both ways (hipMemcpy D2D: 4.9 TB/s).  Forms (QSV_COPY_MODE, for
// measurements; profiles/r03_copy_kernel.txt): 0 = four amplitudes per thread through registers, 1 = one amplitude per
// thread through registers, 2 = one per thread, HBM -> LDS directly (global_load_lds_dwordx4, as the gate kernels
// load) and LDS -> HBM.
constexpr int COPY_ITEMS = 4;
template <int MODE>
__global__ __launch_bounds__(QSV_BLOCK) void k_copy(amp_t *__restrict__ dst, const amp_t *__restrict__ src, uint32_t regions) {
    const uint64_t tile = (regions > 1 && gridDim.x % regions == 0)
                              ? (blockIdx.x % regions) * static_cast<uint64_t>(gridDim.x / regions) + blockIdx.x / regions
                              : blockIdx.x;
    if constexpr (MODE == 0) {
        const uint64_t base = tile * (QSV_BLOCK * COPY_ITEMS) + threadIdx.x;
        amp_t v[COPY_ITEMS];
#pragma unroll
        for (int u = 0; u < COPY_ITEMS; ++u) v[u] = __builtin_nontemporal_load(src + base + u * QSV_BLOCK);
#pragma unroll
        for (int u = 0; u < COPY_ITEMS; ++u) __builtin_nontemporal_store(v[u], dst + base + u * QSV_BLOCK);
    } else if constexpr (MODE == 1) {
        const uint64_t i = tile * QSV_BLOCK + threadIdx.x;
        __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
    } else {
        __shared__ amp_t lds[QSV_BLOCK];
        const uint64_t i = tile * QSV_BLOCK + threadIdx.x;
#if defined(__HIP_DEVICE_COMPILE__)   // the builtin exists in the device pass only
        __builtin_amdgcn_global_load_lds(src + i, lds + (threadIdx.x & ~63u), 16, 0, 2);
#endif
        __syncthreads();
        __builtin_nontemporal_store(lds[threadIdx.x], dst + i);
    }
}

__global__ __launch_bounds__(QSV_BLOCK) void k_zero(amp_t *__restrict__ a, uint64_t amps, uint64_t one_at) {
    for (uint64_t i = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x; i < amps;
         i += static_cast<uint64_t>(gridDim.x) * blockDim.x)
        a[i] = amp_t{i == one_at ? 1.0 : 0.0, 0.0};
}

// splitmix64: counter-based, so a sharded register can be filled shard by shard from global indices.
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__global__ __launch_bounds__(QSV_BLOCK) void k_fill_random(amp_t *__restrict__ a, uint64_t amps, uint64_t seed,
                                                          uint64_t index_offset, double *__restrict__ partials) {
    double s = 0.0, unused = 0.0;
    const uint64_t key = splitmix64(seed);
    for (uint64_t i = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x; i < amps;
         i += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        const uint64_t g = i + index_offset;
        const uint64_t r1 = splitmix64(key ^ (2 * g)), r2 = splitmix64(key ^ (2 * g + 1));
        const double u1 = (static_cast<double>(r1 >> 11) + 0.5) * 0x1.0p-53;  // (0, 1)
        const double u2 = (static_cast<double>(r2 >> 11) + 0.5) * 0x1.0p-53;
        const double rad = sqrt(-2.0 * log(u1));
        double sn, cs;
        sincos(6.283185307179586476925286766559 * u2, &sn, &cs);
        const amp_t v = {rad * cs, rad * sn};
        a[i] = v;
        s += v.x * v.x + v.y * v.y;
    }
    block_sum2(s, unused);
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x] = s;
        partials[2 * blockIdx.x + 1] = 0.0;
    }
}


// ----------------------------------------------------------------------------------------------------
// Streaming forms of the read-out / reshaping kernels (registers of at least 2^14 amplitudes; the plain grid-stride
// forms above stay for smaller ones).  Common shape: one work item = 64 consecutive amplitudes per wave-instruction,
// ITEMS independent items per thread in flight, nontemporal accesses (every amplitude is touched once), and a target
// bit below 6 is resolved inside the wave -- the partner amplitude comes from __shfl_xor, compaction / expansion by
// one qubit is a lane gather -- so that every global access is a whole 1 KiB segment whatever the bit.
// ----------------------------------------------------------------------------------------------------
constexpr int RO_ITEMS = 4;          // the reductions (k_measure_probs_s): four items per thread and trip
// amplitudes per thread of the kernels that move the register (collapse, insert, permute, table diagonals): one -- the
// plain copy kernel reaches 6.55 TB/s with one amplitude per thread and 6.05 with four (profiles/r03_copy_kernel.txt),
// and these kernels follow it (profiles/r03_readout_kernels.csv).  QSV_RO_ITEMS = 1 / 2 / 4 for measurements.
// items for a launch over `count` amplitudes: an AQL dispatch counts work-items in 32 bits (2^24 - 1 workgroups of 256),
// so registers beyond 2^32 amplitudes per launch take two or four per thread
static int ro_fit_items(int items, uint64_t count) {
    while (items < 4 && count / (static_cast<uint64_t>(QSV_BLOCK) * items) > 0x00ffffffull) items *= 2;
    return items;
}
static int ro_move_items() {
    static const int items = [] {
        const char *e = getenv("QSV_RO_ITEMS");
        const int v = e ? atoi(e) : 1;
        return v == 2 || v == 4 ? v : 1;   // (the table diagonals take two: 6.2 TB/s against 5.85 with one or four)
    }();
    return items;
}
#define QSV_RO_DISPATCH(items, CALL)  \
    do {                              \
        if ((items) == 4) { constexpr int IT = 4; CALL; } \
        else if ((items) == 2) { constexpr int IT = 2; CALL; } \
        else { constexpr int IT = 1; CALL; } \
    } while (0)
constexpr int RO_MIN_QUBITS = 14;  // below this the plain grid-stride forms run (tiles of 2^10 amplitudes must divide)

__device__ __forceinline__ amp_t shfl_amp(amp_t v, int src_lane) {
    amp_t r;
    r.x = __shfl(v.x, src_lane, 64);
    r.y = __shfl(v.y, src_lane, 64);
    return r;
}

// partials[2*block + s] = sum over this block's pairs of |eig_s[0] a0 + eig_s[1] a1|^2   (M.apply, gates.py:173-183)
template <bool LOW>
__global__ __launch_bounds__(QSV_BLOCK) void k_measure_probs_s(const amp_t *__restrict__ a, uint64_t amps, int bit,
                                                               cplx e00, cplx e01, cplx e10, cplx e11,
                                                               double *__restrict__ partials) {
    double p0 = 0.0, p1 = 0.0;
    const uint64_t s = 1ull << bit;
    if constexpr (LOW) {
        // natural order: a lane whose bit is 0 holds a0 and fetches a1 from its partner (and accumulates outcome 0),
        // a lane whose bit is 1 holds a1, fetches a0 and accumulates outcome 1: no lane idles, no amplitude is read twice
        const bool up = (threadIdx.x >> bit) & 1;
        const cplx mine = up ? e11 : e00, theirs = up ? e10 : e01;
        double acc = 0.0;
        const uint64_t stride = static_cast<uint64_t>(gridDim.x) * QSV_BLOCK * RO_ITEMS;
        for (uint64_t i0 = blockIdx.x * static_cast<uint64_t>(QSV_BLOCK) * RO_ITEMS + threadIdx.x; i0 < amps; i0 += stride) {
            amp_t v[RO_ITEMS];
#pragma unroll
            for (int u = 0; u < RO_ITEMS; ++u) v[u] = __builtin_nontemporal_load(a + i0 + u * QSV_BLOCK);
#pragma unroll
            for (int u = 0; u < RO_ITEMS; ++u) {
                const amp_t r = cfma(theirs, shfl_xor_amp(v[u], 1 << bit), cmul(mine, v[u]));
                acc += r.x * r.x + r.y * r.y;
            }
        }
        p0 = up ? 0.0 : acc;
        p1 = up ? acc : 0.0;
    } else {
        const uint64_t pairs = amps >> 1;
        const uint64_t stride = static_cast<uint64_t>(gridDim.x) * QSV_BLOCK * RO_ITEMS;
        for (uint64_t w0 = blockIdx.x * static_cast<uint64_t>(QSV_BLOCK) * RO_ITEMS + threadIdx.x; w0 < pairs; w0 += stride) {
            amp_t lo[RO_ITEMS], hi[RO_ITEMS];
#pragma unroll
            for (int u = 0; u < RO_ITEMS; ++u) {
                const uint64_t i = insert_zero(w0 + u * QSV_BLOCK, bit);
                lo[u] = __builtin_nontemporal_load(a + i);
                hi[u] = __builtin_nontemporal_load(a + i + s);
            }
#pragma unroll
            for (int u = 0; u < RO_ITEMS; ++u) {
                const amp_t r0 = cfma(e01, hi[u], cmul(e00, lo[u]));
                const amp_t r1 = cfma(e11, hi[u], cmul(e10, lo[u]));
                p0 += r0.x * r0.x + r0.y * r0.y;
                p1 += r1.x * r1.x + r1.y * r1.y;
            }
        }
    }
    block_sum2(p0, p1);
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x] = p0;
        partials[2 * blockIdx.x + 1] = p1;
    }
}

// out[w] = scale * (e0 a[i0] + e1 a[i0 + s]), i0 = w with a zero inserted at `bit`.
template <bool LOW, int ITEMS>
__global__ __launch_bounds__(QSV_BLOCK) void k_collapse_s(const amp_t *__restrict__ a, amp_t *__restrict__ out,
                                                          uint64_t pairs, int bit, cplx e0, cplx e1, double scale) {
    const cplx f0 = {e0.re * scale, e0.im * scale}, f1 = {e1.re * scale, e1.im * scale};
    const uint64_t w0 = (blockIdx.x * static_cast<uint64_t>(QSV_BLOCK) + (threadIdx.x & ~63u)) * ITEMS + (threadIdx.x & 63);
    if constexpr (LOW) {
        // a wave turns 2 * ITEMS rows of 64 amplitudes into ITEMS rows of 64 results: the pair sum lands in the
        // lanes whose bit is 0, and output lane l gathers it from lane insert_zero(l & 31, bit) of row l >> 5
        const int lane = threadIdx.x & 63;
        const int src = static_cast<int>(insert_zero(static_cast<uint64_t>(lane & 31), bit));
        amp_t v[2 * ITEMS];
#pragma unroll
        for (int u = 0; u < 2 * ITEMS; ++u) v[u] = __builtin_nontemporal_load(a + 2 * (w0 - lane) + u * 64 + lane);
#pragma unroll
        for (int u = 0; u < ITEMS; ++u) {
            const amp_t ra = cfma(f1, shfl_xor_amp(v[2 * u], 1 << bit), cmul(f0, v[2 * u]));
            const amp_t rb = cfma(f1, shfl_xor_amp(v[2 * u + 1], 1 << bit), cmul(f0, v[2 * u + 1]));
            const amp_t ga = shfl_amp(ra, src), gb = shfl_amp(rb, src);
            __builtin_nontemporal_store(lane < 32 ? ga : gb, out + w0 + u * 64);
        }
    } else {
        const uint64_t s = 1ull << bit;
        amp_t lo[ITEMS], hi[ITEMS];
#pragma unroll
        for (int u = 0; u < ITEMS; ++u) {
            const uint64_t i = insert_zero(w0 + u * 64, bit);
            lo[u] = __builtin_nontemporal_load(a + i);
            hi[u] = __builtin_nontemporal_load(a + i + s);
        }
#pragma unroll
        for (int u = 0; u < ITEMS; ++u)
            __builtin_nontemporal_store(cfma(f1, hi[u], cmul(f0, lo[u])), out + w0 + u * 64);
    }
}

// out[j] = amp[bit(j)] * a[j with the bit removed]
template <bool LOW, int ITEMS>
__global__ __launch_bounds__(QSV_BLOCK) void k_insert_s(const amp_t *__restrict__ a, amp_t *__restrict__ out,
                                                        uint64_t in_amps, int bit, cplx c0, cplx c1) {
    const uint64_t w0 = (blockIdx.x * static_cast<uint64_t>(QSV_BLOCK) + (threadIdx.x & ~63u)) * ITEMS + (threadIdx.x & 63);
    amp_t v[ITEMS];
#pragma unroll
    for (int u = 0; u < ITEMS; ++u) v[u] = __builtin_nontemporal_load(a + w0 + u * 64);
    if constexpr (LOW) {
        // a row of 64 inputs becomes two rows of 64 outputs: output lane l of row h reads input lane 32 h + (l without
        // its `bit`) and takes the factor of its own bit
        const int lane = threadIdx.x & 63;
        const int from = static_cast<int>(((static_cast<uint32_t>(lane) >> (bit + 1)) << bit) | (lane & ((1 << bit) - 1)));
        const cplx c = ((lane >> bit) & 1) ? c1 : c0;
#pragma unroll
        for (int u = 0; u < ITEMS; ++u) {
            const uint64_t o = 2 * (w0 - lane + u * 64) + lane;
            __builtin_nontemporal_store(cmul(c, shfl_amp(v[u], from)), out + o);
            __builtin_nontemporal_store(cmul(c, shfl_amp(v[u], 32 + from)), out + o + 64);
        }
    } else {
        const uint64_t s = 1ull << bit;
#pragma unroll
        for (int u = 0; u < ITEMS; ++u) {
            const uint64_t o = insert_zero(w0 + u * 64, bit);
            __builtin_nontemporal_store(cmul(c0, v[u]), out + o);
            __builtin_nontemporal_store(cmul(c1, v[u]), out + o + s);
        }
    }
}

// Qubit permutation: out[j] = a[p(j)], p moves index bits.  A wave owns a tile of 64 amplitudes that is 8 whole
// 128-byte lines on BOTH sides: the tile's six index bits are the destination bits 0..2 (inside a line), the destination
// bits fed by source bits 0..2, and filler bits.  Stores are consecutive within each line; loads hit 8 whole source lines
// in some lane order -- no LDS and no shuffle, the coalescer sees complete lines either way.  The tile's base addresses
// are wave-uniform: the destination base is the tile number with zeros inserted at the tile's bit positions, the source
// base is that number pushed through the bit permutation one byte at a time (256-entry tables, scalar loads).
struct PermTileArgs {
    uint64_t tiles;          // amps / 64
    int32_t bytes;           // ceil(n / 8): lookup tables
    uint8_t tile_dst[6];     // destination bit of lane bit k, ascending (tile_dst[0..2] = 0, 1, 2)
    uint8_t tile_src[6];     // source bit that feeds it
};

template <int ITEMS>
__global__ __launch_bounds__(QSV_BLOCK) void k_permute_s(const amp_t *__restrict__ a, amp_t *__restrict__ out,
                                                         const PermTileArgs g,
                                                         const uint64_t *__restrict__ lut /*[bytes][256]*/) {
    const int lane = threadIdx.x & 63;
    uint64_t dst_lane = 0, src_lane = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const uint64_t b = (lane >> k) & 1;
        dst_lane |= b << g.tile_dst[k];
        src_lane |= b << g.tile_src[k];
    }
    // wave-uniform on purpose (readfirstlane): the per-tile address arithmetic below then runs on the scalar unit
    const uint64_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (QSV_BLOCK / 64) + (threadIdx.x >> 6));
    const uint64_t waves = static_cast<uint64_t>(gridDim.x) * (QSV_BLOCK / 64);
    for (uint64_t t0 = wave * ITEMS; t0 < g.tiles; t0 += waves * ITEMS) {
        amp_t v[ITEMS];
        uint64_t dst[ITEMS];
#pragma unroll
        for (int u = 0; u < ITEMS; ++u) {
            uint64_t d = t0 + u;
#pragma unroll
            for (int k = 0; k < 6; ++k) d = insert_zero(d, g.tile_dst[k]);
            uint64_t sidx = 0;
            for (int b = 0; b < g.bytes; ++b) sidx |= lut[b * 256 + ((d >> (8 * b)) & 255)];
            dst[u] = d | dst_lane;
            v[u] = amp_t{0.0, 0.0};
            if (t0 + u < g.tiles) v[u] = __builtin_nontemporal_load(a + (sidx | src_lane));
        }
#pragma unroll
        for (int u = 0; u < ITEMS; ++u)
            if (t0 + u < g.tiles) __builtin_nontemporal_store(v[u], out + dst[u]);
    }
}

// K-qubit diagonal (K <= 6): a[i] *= table[bits of i at bitpos], natural order, table in LDS.
template <int ITEMS>
__global__ __launch_bounds__(QSV_BLOCK) void k_diag_table_s(amp_t *__restrict__ a, uint64_t amps, int K,
                                                           const uint8_t *__restrict__ bitpos,
                                                           const double *__restrict__ table) {
    __shared__ amp_t tab[1 << QSV_MAX_K];
    __shared__ int bp[QSV_MAX_K];
    if (threadIdx.x < (1 << K)) tab[threadIdx.x] = amp_t{table[2 * threadIdx.x], table[2 * threadIdx.x + 1]};
    if (threadIdx.x < K) bp[threadIdx.x] = bitpos[threadIdx.x];
    __syncthreads();
    const uint64_t i0 = blockIdx.x * static_cast<uint64_t>(QSV_BLOCK) * ITEMS + threadIdx.x;
    amp_t v[ITEMS];
#pragma unroll
    for (int u = 0; u < ITEMS; ++u) v[u] = __builtin_nontemporal_load(a + i0 + u * QSV_BLOCK);
#pragma unroll
    for (int u = 0; u < ITEMS; ++u) {
        const uint64_t i = i0 + u * QSV_BLOCK;
        int sel = 0;
        for (int j = 0; j < K; ++j) sel |= static_cast<int>((i >> bp[j]) & 1ull) << (K - 1 - j);
        const amp_t d = tab[sel];
        __builtin_nontemporal_store(cmul(cplx{d.x, d.y}, v[u]), a + i);
    }
}


// ----------------------------------------------------------------------------------------------------
// Reduced density matrix of k <= 6 kept qubits in ONE read pass:  rho[i][j] = sum_g psi[i, g] conj(psi[j, g]),
// g running over the 2^(n-k) settings of the other qubits.  That is X X^H for the (2^k x 2^(n-k)) matrix X -- a rank
// update with a tiny result -- and runs on the f64 matrix cores: a lane (i = lane & 15, kk = lane >> 4) loads ONE
// amplitude, row i of group 4 q + kk, and the same register pair serves as A[i][kk] and as B[kk][j] of
// v_mfma_f64_16x16x4_f64 (re = xr xr^T + xi xi^T, im = xi xr^T - xr xi^T).  T = 1, 2, 4 row tiles of 16 cover
// 2^k <= 16, 32, 64; only the upper triangle of tiles is accumulated (rho is Hermitian).  The sum is deterministic:
// waves add into their workgroup's LDS tile one after the other, workgroups write partials, a second launch adds
// the partials in index order.
// ----------------------------------------------------------------------------------------------------
struct RdmArgs {
    uint64_t W;          // groups = amps >> k
    uint64_t or_mask;    // unused (0); lets deposit() serve this struct too
    int32_t nins;        // k
    int32_t D;           // 2^k
    uint32_t pos[8];     // ascending kept bits
};

// RDM_U quads of groups are loaded before the first MFMA of an iteration (a wave with a single 16-byte load in flight
// spends its life waiting for HBM: 0.25-1.3 TB/s in the first version of this kernel).  When 2^k < 16 the sixteen rows
// of the tile are shared by S = 16 / 2^k groups (row i = r + 2^k s): every lane still loads a different amplitude, the
// tile then holds S x S blocks of which only the S diagonal ones (same group on both sides) mean anything; the host adds
// those up.
constexpr int RDM_LOADS = 4;   // 16-byte loads per lane and buffer: RDM_LOADS / T quads of groups

template <int T>
__global__ __launch_bounds__(QSV_BLOCK) void k_rdm(const amp_t *__restrict__ a, const RdmArgs g,
                                                   const uint64_t *__restrict__ hoff,  // [16 T] row offsets
                                                   double *__restrict__ partials) {    // [grid][P][2][256]
    constexpr int P = T * (T + 1) / 2;
    constexpr int RDM_U = RDM_LOADS / T;
    __shared__ double red[P * 2 * 256];
    const int lane = threadIdx.x & 63, i = lane & 15, kk = lane >> 4;
    const int S = T == 1 ? 16 / g.D : 1;                 // groups sharing the 16 rows of a tile (D = 1 never occurs)
    const uint64_t sub = T == 1 ? static_cast<uint64_t>(i / g.D) : 0;
    uint64_t row_off[T];
#pragma unroll
    for (int t = 0; t < T; ++t) row_off[t] = hoff[T == 1 ? i % g.D : 16 * t + i];
    // C independent accumulator sets: consecutive MFMAs never wait for one another's result (with one set per tile
    // pair the two updates of `re` and of `im` in a step are back-to-back dependent issues of a 64-cycle instruction)
    constexpr int C = RDM_U >= 2 ? 2 : 1;
    f64x4 re[C][P], im[C][P];
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
        for (int p = 0; p < P; ++p) re[c][p] = im[c][p] = f64x4{0.0, 0.0, 0.0, 0.0};
    // a step = 4 S groups = one MFMA k-slice per tile pair.  The host sizes the grid so that waves * RDM_U divides the
    // step count: every load below is unconditional.
    const uint64_t steps = g.W / (4 * static_cast<uint64_t>(S));
    const uint64_t wave = blockIdx.x * (QSV_BLOCK / 64) + (threadIdx.x >> 6);
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * (QSV_BLOCK / 64) * RDM_U;
    auto fetch = [&](amp_t (&x)[RDM_U][T], uint64_t q0) {
#pragma unroll
        for (int u = 0; u < RDM_U; ++u) {
            const uint64_t base = deposit(((q0 + u) * 4 + kk) * S + sub, g);
#pragma unroll
            for (int t = 0; t < T; ++t) x[u][t] = __builtin_nontemporal_load(a + base + row_off[t]);
        }
    };
    auto update = [&](const amp_t (&x)[RDM_U][T]) {
        // first halves of every sum, then second halves: 2 C P independent instructions between dependent ones
#pragma unroll
        for (int u = 0; u < RDM_U; ++u) {
            int p = 0;
#pragma unroll
            for (int ti = 0; ti < T; ++ti)
#pragma unroll
                for (int tj = ti; tj < T; ++tj, ++p) {
                    re[u % C][p] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[u][ti].x, x[u][tj].x, re[u % C][p], 0, 0, 0);
                    im[u % C][p] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[u][ti].y, x[u][tj].x, im[u % C][p], 0, 0, 0);
                }
        }
#pragma unroll
        for (int u = 0; u < RDM_U; ++u) {
            int p = 0;
#pragma unroll
            for (int ti = 0; ti < T; ++ti)
#pragma unroll
                for (int tj = ti; tj < T; ++tj, ++p) {
                    re[u % C][p] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[u][ti].y, x[u][tj].y, re[u % C][p], 0, 0, 0);
                    im[u % C][p] = __builtin_amdgcn_mfma_f64_16x16x4f64(-x[u][ti].x, x[u][tj].y, im[u % C][p], 0, 0, 0);
                }
        }
        __builtin_amdgcn_sched_barrier(0);   // the other buffer's loads stay where they are written: behind these MFMAs
    };
    // A ring of NBUF buffers, no copies between them: while one feeds the matrix cores the loads of the NBUF - 1 others
    // are in flight, and the wait in front of the MFMAs is for the OLDEST buffer only (a register copy at the loop end
    // made the compiler wait for every outstanding load in the middle of the MFMAs: one buffer in flight per wave,
    // 4.3 TB/s).  Every fetch is unconditional (past the end a wave re-reads its first chunk and drops it): with loads
    // under a branch the compiler cannot count how many younger loads are in flight and waits for all of them.
    // T = 4 runs one wave per SIMD (160 accumulator registers) and an update is 40 MFMAs = 1 us: three buffers ahead
    // cover the HBM latency; T = 1 has four waves per SIMD and needs one.
    constexpr int NBUF = T == 1 ? 2 : T == 2 ? 3 : 4;
    amp_t x[NBUF][RDM_U][T];
    const uint64_t first = wave * RDM_U;
#pragma unroll
    for (int b = 0; b < NBUF - 1; ++b) {
        const uint64_t q = first + b * stride;
        fetch(x[b], q < steps ? q : first);
    }
    for (uint64_t q0 = first; q0 < steps;) {
#pragma unroll
        for (int b = 0; b < NBUF; ++b) {
            const uint64_t q = q0 + (NBUF - 1) * stride;
            fetch(x[(b + NBUF - 1) % NBUF], q < steps ? q : first);
            update(x[b]);
            q0 += stride;
            if (q0 >= steps) break;
        }
    }
#pragma unroll
    for (int c = 1; c < C; ++c)
#pragma unroll
        for (int p = 0; p < P; ++p) {
            re[0][p] += re[c][p];
            im[0][p] += im[c][p];
        }
    // deterministic sum over the four waves of the workgroup, then one partial per workgroup
    for (int wv = 0; wv < QSV_BLOCK / 64; ++wv) {
        if ((threadIdx.x >> 6) == wv) {
#pragma unroll
            for (int p = 0; p < P; ++p)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double *slot_re = red + ((p * 2 + 0) * 4 + r) * 64 + lane;
                    double *slot_im = red + ((p * 2 + 1) * 4 + r) * 64 + lane;
                    *slot_re = (wv == 0 ? 0.0 : *slot_re) + re[0][p][r];
                    *slot_im = (wv == 0 ? 0.0 : *slot_im) + im[0][p][r];
                }
        }
        __syncthreads();
    }
    double *out = partials + static_cast<size_t>(blockIdx.x) * (P * 2 * 256);
    for (int e = threadIdx.x; e < P * 2 * 256; e += QSV_BLOCK) out[e] = red[e];
}

// Round 3 form of the same rank update: a WORKGROUP tile staged through LDS, so that HBM is read in whole 1 KiB
// wave-instructions whatever the kept bits are.  k_rdm above lets lane i of the MFMA operand fetch row i itself: the 16
// rows of a group are 2^(kept bit) apart, so with kept bits outside the lowest four every lane touches a different
// 128-byte line (scattered kept bits [0, 5, 12, 25] at n = 28: 2.9 TB/s; k = 6: 1.9).  Here a tile is (16 T rows) x (64
// groups).  A wave-load is 64 CONSECUTIVE amplitudes for one setting c_h of the kept bits >= 6: the lane bits carry the
// kept bits < 6 (rows) and 6 - l free bits (groups); the 16 T loads of a tile (each wave issues 4 T of them, one tile
// ahead, into registers) are written into the LDS tile at [row][group ^ (row & 15)] and read back as MFMA operands --
// lane (i, kk), row tile t, step m reads [16 t + i][(4 m + kk) ^ i]: 16 distinct 16-byte columns per 16-lane group.
// The arithmetic is cut from four to three MFMAs per (tile pair, step): with a = xr_i, b = xi_i, c = xr_j, d = xi_j
//     P1 += a c^T,  P2 += b d^T,  P3 += (a + b)(c - d)^T      re = P1 + P2,   im = b c^T - a d^T = P3 - P1 + P2,
// 7.5 instead of 10 MFMAs per KiB at k = 6 (the kernel that is bound by the matrix cores).  The (pair, step) units of a
// tile are dealt to the four waves: T <= 2: four steps each, all pairs; T = 4: five pairs x eight steps each.
struct RdmTileArgs {
    uint64_t tiles;      // tiles of 64 S groups
    uint64_t or_mask;    // unused (0); lets deposit() serve this struct too
    int32_t nins;        // h: kept bits >= 6
    uint32_t pos[8];     // those bits, ascending
    int32_t k, l, h;     // kept bits, of which below / from bit 6
    uint32_t lmask;      // lane bits that are kept bits
    int32_t log_s;       // 2^k < 16: log2 of the groups sharing the 16 rows
    uint32_t regions;    // tile order: R > 1 walks R contiguous regions of the register side by side
    uint64_t hoff[64];   // offset of setting c_h of the kept bits >= 6 (kernel arguments: scalar loads, no upload)
};

template <int T>
__global__ __launch_bounds__(QSV_BLOCK) __attribute__((amdgpu_waves_per_eu(2, T == 4 ? 2 : 4))) void k_rdm_tile(
    const amp_t *__restrict__ a, const RdmTileArgs g, double *__restrict__ partials) {   // [grid][P][2][256]
    // Work split over the four waves.  T <= 2: four of the sixteen steps each, every tile pair.  T = 4 (ten pairs: all of
    // them would be 240 accumulator registers): five pairs x eight steps.  Both halves run the SAME code on the pair list
    // A = {(0,0), (2,2), (0,1), (2,3), (0,2)} of operand slots; the second half fills slot tt with row tile tt + 1 mod 4
    // and so computes (1,1), (3,3), (1,2), (3,0), (1,3) -- the complement (the cyclic shift maps A onto it), with (3,0)
    // standing for (0,3) as its conjugate transpose (the host reads that block mirrored).  (Two code paths with their
    // own pair lists made the compiler hold both sets of operands: 256 registers and 100 spilled.)
    constexpr int P = T * (T + 1) / 2, NL = 4 * T /* loads per wave and tile */;
    constexpr int MY_P = T == 4 ? 5 : P, MY_M = T == 4 ? 8 : 4;
    constexpr int TILE_BYTES = 16 * T * 64 * 16, RED_BYTES = P * 2 * 256 * 8;
    __shared__ __attribute__((aligned(16))) char smem[TILE_BYTES > RED_BYTES ? TILE_BYTES : RED_BYTES];
    amp_t *tile = reinterpret_cast<amp_t *>(smem);
    double *red = reinterpret_cast<double *>(smem);
    const int lane = threadIdx.x & 63, i = lane & 15, kk = lane >> 4;
    const uint32_t wave_s = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // this lane's place in a wave-load: row bits (kept bits < 6) and group bits (the other lane bits)
    uint32_t r_low = 0, f_low = 0;
    {
        int rb = 0, fb = 0;
        for (int b = 0; b < 6; ++b) {
            if ((g.lmask >> b) & 1u) r_low |= ((lane >> b) & 1u) << rb++;
            else f_low |= ((lane >> b) & 1u) << fb++;
        }
    }
    // the NL loads of this wave: load j = NL wave + jj -> c_h = j mod 2^h, block bb = j >> h (s = bb >> l, b = bb mod 2^l).
    // Everything about a load except the lane's own row / group bits is the same for the whole wave: scalar registers.
    const uint32_t per_tile = 1u << (g.l + g.log_s);     // consecutive w values a tile consumes
    auto lds_slot = [&](int jj) {                         // where my amplitude of load jj goes in the tile
        const uint32_t j = NL * wave_s + jj, c_h = j & ((1u << g.h) - 1u), bb = j >> g.h;
        const uint32_t sblk = bb >> g.l, b = bb & ((1u << g.l) - 1u);
        const uint32_t row = (sblk << g.k) | (c_h << g.l) | r_low, slot = (b << (6 - g.l)) | f_low;
        return row * 64 + (slot ^ (row & 15u));
    };
    auto fetch = [&](amp_t (&x)[NL], uint64_t t) {
#pragma unroll
        for (int jj = 0; jj < NL; ++jj) {
            const uint32_t j = NL * wave_s + jj;
            const uint64_t w = t * per_tile + (j >> g.h);
            const uint64_t c_off = g.hoff[j & ((1u << g.h) - 1u)];    // scalar load
            x[jj] = __builtin_nontemporal_load(a + deposit(w << 6, g) + c_off + lane);
        }
    };
    f64x4 p1[MY_P], p2[MY_P], p3[MY_P];
#pragma unroll
    for (int p = 0; p < MY_P; ++p) p1[p] = p2[p] = p3[p] = f64x4{0.0, 0.0, 0.0, 0.0};
    const uint32_t shift = T == 4 ? (wave_s & 1u) : 0u;                       // row-tile rotation of this wave
    const int m_first = MY_M * (T == 4 ? wave_s >> 1 : wave_s);
    // operand slots of pair q (T = 4: the list A above; otherwise every pair ti <= tj in order)
    constexpr int A_I[5] = {0, 2, 0, 2, 0}, A_J[5] = {0, 2, 1, 3, 2};
    amp_t x[NL];
    // tile order: with R regions, the workgroups in flight together read from R places of the register instead of one
    // narrow window per kept-bit setting (kept bits on the top address bits: every stream would sit in the same channels)
    const uint64_t per_region = g.regions > 1 ? g.tiles / g.regions : 0;
    auto place = [&](uint64_t s) { return g.regions > 1 ? (s % g.regions) * per_region + s / g.regions : s; };
    uint64_t t = blockIdx.x;
    fetch(x, place(t));                                   // gridDim.x <= tiles: every workgroup owns at least one
    for (; t < g.tiles; t += gridDim.x) {
#pragma unroll
        for (int jj = 0; jj < NL; ++jj) tile[lds_slot(jj)] = x[jj];
        __syncthreads();
        const uint64_t nxt = t + gridDim.x;
        fetch(x, place(nxt < g.tiles ? nxt : blockIdx.x));   // unconditional (see k_rdm): past the end re-read and drop
        auto step = [&](int mm) {
            const int m = m_first + mm;
            amp_t v[T];
#pragma unroll
            for (int tt = 0; tt < T; ++tt) v[tt] = tile[(16 * ((tt + shift) & (T - 1)) + i) * 64 + ((4 * m + kk) ^ i)];
            if constexpr (T == 4) {
#pragma unroll
                for (int q = 0; q < MY_P; ++q) {
                    const amp_t vi = v[A_I[q]], vj = v[A_J[q]];
                    p1[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(vi.x, vj.x, p1[q], 0, 0, 0);
                    p2[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(vi.y, vj.y, p2[q], 0, 0, 0);
                    p3[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(vi.x + vi.y, vj.x - vj.y, p3[q], 0, 0, 0);
                }
            } else {
                int q = 0;
#pragma unroll
                for (int ti = 0; ti < T; ++ti)
#pragma unroll
                    for (int tj = ti; tj < T; ++tj, ++q) {
                        p1[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(v[ti].x, v[tj].x, p1[q], 0, 0, 0);
                        p2[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(v[ti].y, v[tj].y, p2[q], 0, 0, 0);
                        p3[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(v[ti].x + v[ti].y, v[tj].x - v[tj].y, p3[q], 0, 0, 0);
                    }
            }
            // the next step's LDS reads stay behind these MFMAs: hoisted, the operands of all steps are live at once
            __builtin_amdgcn_sched_barrier(0);
        };
        if constexpr (T == 4) {       // a real loop: unrolled, the eight steps of 15 MFMAs push the allocator into spills
#pragma unroll 1
            for (int mm = 0; mm < MY_M; ++mm) step(mm);
        } else {
#pragma unroll
            for (int mm = 0; mm < MY_M; ++mm) step(mm);
        }
        __syncthreads();                                  // every wave is done with the tile before it is overwritten
    }
    // re = P1 + P2, im = P3 - P1 + P2; deterministic sum over the waves that share a pair, one partial per workgroup.
    // Pair index p of slot pair q: T = 4, first half (0,0) (2,2) (0,1) (2,3) (0,2) = 0 7 1 8 2; second half (1,1) (3,3)
    // (1,2) (3,0) (1,3) = 4 9 5 3 6, where 3 = (0,3) holds the block of (3,0): its conjugate transpose.
    for (uint32_t wv = 0; wv < 4; ++wv) {
        if (wave_s == wv) {
            const bool first = T == 4 ? wv < 2 : wv == 0;
#pragma unroll
            for (int q = 0; q < MY_P; ++q) {
                constexpr int HALF0[5] = {0, 7, 1, 8, 2}, HALF1[5] = {4, 9, 5, 3, 6};
                const int p = T == 4 ? (shift ? HALF1[q] : HALF0[q]) : q;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double *slot_re = red + ((p * 2 + 0) * 4 + r) * 64 + lane;
                    double *slot_im = red + ((p * 2 + 1) * 4 + r) * 64 + lane;
                    *slot_re = (first ? 0.0 : *slot_re) + (p1[q][r] + p2[q][r]);
                    *slot_im = (first ? 0.0 : *slot_im) + (p3[q][r] - p1[q][r] + p2[q][r]);
                }
            }
        }
        __syncthreads();
    }
    double *out = partials + static_cast<size_t>(blockIdx.x) * (P * 2 * 256);
    for (int e = threadIdx.x; e < P * 2 * 256; e += QSV_BLOCK) out[e] = red[e];
}

// out[e] = sum over blocks of partials[block][e].  16 entries x 16 slices per workgroup: slice s adds blocks s, s+16, ...
// in order, the 16 slice sums are added in slice order through LDS -- a fixed summation tree, so the result does not
// depend on scheduling (one thread per entry walking every block took 0.24 ms: a chain of ~1000 dependent-latency loads).
__global__ __launch_bounds__(QSV_BLOCK) void k_sum_partials(const double *__restrict__ partials, int blocks, int entries,
                                                            double *__restrict__ out) {
    __shared__ double part[16][17];
    const int le = threadIdx.x & 15, slice = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + le;
    double s = 0.0;
    if (e < entries)
        for (int b = slice; b < blocks; b += 16) s += partials[static_cast<size_t>(b) * entries + e];
    part[slice][le] = s;
    __syncthreads();
    if (slice == 0 && e < entries) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += part[i][le];
        out[e] = t;
    }
}

// Small registers: one thread per entry (i, j) of rho walks every group (2^n amplitudes < 2^14: microseconds).
__global__ __launch_bounds__(QSV_BLOCK) void k_rdm_small(const amp_t *__restrict__ a, const RdmArgs g,
                                                         const uint64_t *__restrict__ hoff, double *__restrict__ out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= g.D * g.D) return;
    const int i = e / g.D, j = e % g.D;
    double sr = 0.0, si = 0.0;
    for (uint64_t w = 0; w < g.W; ++w) {
        const uint64_t base = deposit(w, g);
        const amp_t x = a[base + hoff[i]], y = a[base + hoff[j]];
        sr += x.x * y.x + x.y * y.y;   // x conj(y)
        si += x.y * y.x - x.x * y.y;
    }
    out[2 * e] = sr;
    out[2 * e + 1] = si;
}

// <a| rho |a> for a ket `a` (2^n amplitudes) and a density matrix stored row-major as a 2n-qubit register:
// partials[2b], [2b+1] = this block's share of sum_ij conj(a_i) rho_ij a_j.  rho is streamed once; the ket stays in L2.
__global__ __launch_bounds__(QSV_BLOCK) void k_expect_density(const amp_t *__restrict__ ket, const amp_t *__restrict__ rho,
                                                              uint64_t dim_bits, double *__restrict__ partials) {
    double re = 0.0, im = 0.0;
    const uint64_t total = 1ull << (2 * dim_bits), mask = (1ull << dim_bits) - 1;
    for (uint64_t e = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x; e < total;
         e += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        const amp_t r = __builtin_nontemporal_load(rho + e);
        const amp_t ai = ket[e >> dim_bits], aj = ket[e & mask];
        // conj(ai) * aj
        const double cr = ai.x * aj.x + ai.y * aj.y, ci = ai.x * aj.y - ai.y * aj.x;
        re += cr * r.x - ci * r.y;
        im += cr * r.y + ci * r.x;
    }
    block_sum2(re, im);
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x] = re;
        partials[2 * blockIdx.x + 1] = im;
    }
}

// ----------------------------------------------------------------------------------------------------
// host-side helpers
// ----------------------------------------------------------------------------------------------------
int grid_for(uint64_t items, int per_block, int cap) {
    uint64_t blocks = (items + per_block - 1) / per_block;
    if (blocks < 1) blocks = 1;
    if (cap > 0 && blocks > static_cast<uint64_t>(cap)) blocks = cap;
    // an AQL dispatch counts work-ITEMS in 32 bits: at most 2^32 / 256 workgroups of 256 threads per launch
    // (a 33-qubit register would need 2^25); every kernel launched through here loops over the remainder
    if (blocks > 0x00ffffffull) blocks = 0x00ffffffull;
    return static_cast<int>(blocks);
}

int check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return qsv_fail(QSV_EHIP, std::string("kernel launch: ") + hipGetErrorString(e));
    return QSV_OK;
}

// Sum the first `blocks` pairs of partials on the host, in index order (deterministic).
int sum_partials(qsv_state *st, int blocks, double *x, double *y) {
    QSV_HIP(hipMemcpyAsync(st->partials_host, st->partials, sizeof(double) * 2 * blocks, hipMemcpyDeviceToHost,
                           st->stream));
    QSV_HIP(hipStreamSynchronize(st->stream));
    double sx = 0.0, sy = 0.0;
    for (int i = 0; i < blocks; ++i) {
        sx += st->partials_host[2 * i];
        sy += st->partials_host[2 * i + 1];
    }
    *x = sx;
    if (y) *y = sy;
    return QSV_OK;
}

template <int KH, int KL, int U>
void launch_dense_nt(qsv_state *st, const GateArgs &g, int grid) {
    const bool sub = g.nins > KH || g.lane_ctrl != 0;
    const dim3 gd(grid), bd(QSV_BLOCK);
    snprintf(st->last_kernel, sizeof(st->last_kernel), "k_dense%s<%d, %d, %d, %s>", sub ? "_ctrl" : "", KH, KL, U,
             st->nontemporal ? "true" : "false");
    if (st->nontemporal) {
        if (sub) hipLaunchKernelGGL((k_dense_ctrl<KH, KL, U, true>), gd, bd, 0, st->stream, st->data, g);
        else hipLaunchKernelGGL((k_dense<KH, KL, U, true>), gd, bd, 0, st->stream, st->data, g);
    } else {
        if (sub) hipLaunchKernelGGL((k_dense_ctrl<KH, KL, U, false>), gd, bd, 0, st->stream, st->data, g);
        else hipLaunchKernelGGL((k_dense<KH, KL, U, false>), gd, bd, 0, st->stream, st->data, g);
    }
}

// Work items in flight per thread, from the MI355X sweeps under profiles/ (n = 28): with nontemporal
// accesses one item per thread streams best (6.0-6.4 TB/s) until the pair stride reaches 16 MiB (bit 20),
// where four items per thread hold 5.8 TB/s and one item drops to 5.4.
template <int KH, int KL>
int default_unroll(const GateArgs &g) {
    if (KH == 0) return KL == 2 ? 2 : 1;
    bool far = false;  // a pair stride of 16 MiB .. 512 MiB (bits 20..25)
    for (int h = 1; h < (1 << KH); h <<= 1) {
        int bit = 0;
        while ((g.hoff[h] >> bit) > 1) ++bit;
        far = far || (bit >= 20 && bit <= 25);
    }
    if (KH == 1 && KL == 1) return 1;   // round 2 re-sweep: one item per thread at every stride (1.35-1.44 against 1.50-1.53 ms on bits 20..25)
    if (KH == 1 && KL == 0 && far) {
        // single far pair stride: only 16 MiB (bit 20) wants four items per thread (profiles/r01_sweep_far_bits.txt)
        int bit = 0;
        while ((g.hoff[1] >> bit) > 1) ++bit;
        return bit == 20 ? 4 : 1;
    }
    return far ? 4 : 1;
}

template <int KH, int KL>
int launch_dense(qsv_state *st, const GateArgs &g) {
    int U = st->unroll > 0 ? st->unroll : default_unroll<KH, KL>(g);
    // never more unrolling than there is work for one tile
    while (U > 1 && g.W < static_cast<uint64_t>(QSV_BLOCK) * U) U >>= 1;
    GateArgs ga = g;
    ga.ubit = st->ubit;
    // measured on MI355X at n = 28 (profiles/r01_sweep_tile_order.txt): natural-order kernels like 32 regions,
    // pair kernels 8 (one per XCD), except at pair strides of 16..512 MiB where the plain order is best
    if (st->remap >= 0) {
        ga.remap = st->remap;
    } else {
        bool far = false;
        int top = 0;
        for (int h = 1; h < (1 << KH); h <<= 1) {
            int bit = 0;
            while ((g.hoff[h] >> bit) > 1) ++bit;
            far = far || (bit >= 20 && bit <= 25);
            top = bit > top ? bit : top;
        }
        const bool sub = g.nins > KH || g.lane_ctrl != 0;  // controlled / pair-exchange launches: plain order
        if (sub) ga.remap = 0;
        else if (KH == 0) ga.remap = KL == 2 ? 0 : 32;   // round 2 re-sweep (tools/probe_low_pairs.py): 1.35 against 1.41 ms
        else if (KH == 1 && KL == 0) ga.remap = top == 20 ? 0 : top == 24 ? 2 : 8;  // per-stride winners of the sweeps
        else if (KH == 1) ga.remap = (top == 20 || top == 24) ? 32 : 8;   // KL = 1; round 2 re-sweep (tools/probe_retune.py)
        else ga.remap = top < 20 ? 8 : 0;
    }
    while (ga.ubit > 8 && (g.W >> ga.ubit) < static_cast<uint64_t>(U)) --ga.ubit;  // small registers
    const int grid = grid_for(g.W, QSV_BLOCK * U, st->grid_cap);
    switch (U) {
        case 1: launch_dense_nt<KH, KL, 1>(st, ga, grid); break;
        case 2: launch_dense_nt<KH, KL, 2>(st, ga, grid); break;
        case 4: launch_dense_nt<KH, KL, 4>(st, ga, grid); break;
        default: launch_dense_nt<KH, KL, 8>(st, ga, grid); break;
    }
    return check_launch();
}

int launch_diag(qsv_state *st, const DiagArgs &g0) {
    DiagArgs g = g0;
    // full traffic: one item per thread, 32 regions; sub-space launches (Z, CZ, multi-controlled phases):
    // four items per thread, one region per XCD (profiles/r01_sweep_sub_kernels.txt)
    const bool sub = g0.nins > 0 || g0.lane_ctrl != 0;
    g.remap = st->remap >= 0 ? st->remap : (sub ? 8 : 32);
    int U = st->unroll > 0 ? st->unroll : (sub ? 4 : 1);
    while (U > 1 && g.W < static_cast<uint64_t>(QSV_BLOCK) * U) U >>= 1;
    const dim3 gd(grid_for(g.W, QSV_BLOCK * U, st->grid_cap)), bd(QSV_BLOCK);
    snprintf(st->last_kernel, sizeof(st->last_kernel), "k_diag<%d, %s>%s", U, st->nontemporal ? "true" : "false",
             (g.nins > 0 || g.lane_ctrl) ? " [sub-space]" : "");
#define QSV_LAUNCH_DIAG(UU)                                                                  \
    if (st->nontemporal)                                                                      \
        hipLaunchKernelGGL((k_diag<UU, true>), gd, bd, 0, st->stream, st->data, g);            \
    else                                                                                      \
        hipLaunchKernelGGL((k_diag<UU, false>), gd, bd, 0, st->stream, st->data, g)
    switch (U) {
        case 1: QSV_LAUNCH_DIAG(1); break;
        case 2: QSV_LAUNCH_DIAG(2); break;
        case 4: QSV_LAUNCH_DIAG(4); break;
        default: QSV_LAUNCH_DIAG(8); break;
    }
#undef QSV_LAUNCH_DIAG
    return check_launch();
}

int dispatch_dense(qsv_state *st, int KH, int KL, const GateArgs &g) {
    if (KH == 1 && KL == 0) return launch_dense<1, 0>(st, g);
    if (KH == 0 && KL == 1) return launch_dense<0, 1>(st, g);
    if (KH == 2 && KL == 0) return launch_dense<2, 0>(st, g);
    if (KH == 1 && KL == 1) return launch_dense<1, 1>(st, g);
    if (KH == 0 && KL == 2) return launch_dense<0, 2>(st, g);
    return qsv_fail(QSV_EINVAL, "dense kernel: unsupported target split");
}

// Fill pos/or_mask/lane_ctrl/W from target and control bit positions.
template <class Args>
int fill_enumeration(const qsv_state *st, Args &g, const std::vector<int> &removed_high, int nctrl,
                     const int *cbits) {
    std::vector<int> ins(removed_high);
    g.or_mask = 0;
    g.lane_ctrl = 0;
    for (int i = 0; i < nctrl; ++i) {
        if (cbits[i] >= QSV_LANE_BITS) {
            ins.push_back(cbits[i]);
            g.or_mask |= 1ull << cbits[i];
        } else {
            g.lane_ctrl |= 1u << cbits[i];
        }
    }
    std::sort(ins.begin(), ins.end());
    if (ins.size() > static_cast<size_t>(QSV_MAX_INS)) return qsv_fail(QSV_EINVAL, "too many controls");
    g.nins = static_cast<int>(ins.size());
    for (size_t i = 0; i < ins.size(); ++i) g.pos[i] = static_cast<uint32_t>(ins[i]);
    g.W = st->amps >> ins.size();
    return QSV_OK;
}

// Expand a controlled k-qubit matrix to the full (k + nctrl)-qubit matrix (controls as leading legs).
std::vector<double> expand_controls(int k, int nctrl, const double *m) {
    const int D = 1 << k, F = 1 << (k + nctrl);
    std::vector<double> full(2ull * F * F, 0.0);
    for (int r = 0; r < F; ++r) full[2 * (r * F + r)] = 1.0;
    const int b0 = F - D;  // all controls = 1
    for (int r = 0; r < D; ++r)
        for (int c = 0; c < D; ++c) {
            full[2 * ((b0 + r) * F + b0 + c)] = m[2 * (r * D + c)];
            full[2 * ((b0 + r) * F + b0 + c) + 1] = m[2 * (r * D + c) + 1];
        }
    return full;
}

}  // namespace

// ----------------------------------------------------------------------------------------------------
// launchers
// ----------------------------------------------------------------------------------------------------
int qsvk_ensure_matrix(qsv_state *st, size_t bytes) {
    if (st->dev_matrix_bytes >= bytes) return QSV_OK;
    if (st->dev_matrix) {
        QSV_HIP(hipStreamSynchronize(st->stream));
        QSV_HIP(hipFree(st->dev_matrix));
        st->dev_matrix = nullptr;
        st->dev_matrix_bytes = 0;
    }
    if (hipMalloc(reinterpret_cast<void **>(&st->dev_matrix), bytes) != hipSuccess)
        return qsv_fail(QSV_ENOMEM, "device allocation of the gate-matrix buffer failed");
    st->dev_matrix_bytes = bytes;
    return QSV_OK;
}

// Gate matrices and index tables come from host memory that dies when the call returns.  Copying them to the device
// with hipMemcpyAsync + hipStreamSynchronize makes every such gate wait for the previous kernel before its own launch
// can even be queued: 72 us of idle GPU between the fused blocks of a circuit (4 % of the pass at n = 28, two thirds of
// it at n = 22).  Instead the data is copied into the next slot of a pinned ring by the CPU, a transfer to the slot's
// device mirror is queued on the register's stream in front of the kernel, and an event recorded behind the
// kernel says when the slot may be overwritten -- the host only ever waits when it is a whole ring ahead of the GPU.
// Payloads larger than a slot take the synchronous road through st->dev_matrix.
// The transfer itself is a small kernel that reads the pinned slot over PCIe (pinned host memory is mapped into the
// device's address space): a copy-engine transfer queued between two kernels starts ~45 us after the first kernel ends
// (the dependency crosses from the compute queue to the SDMA queue), a kernel behind a kernel on the same queue ~5 us.
__global__ __launch_bounds__(QSV_BLOCK) void k_stage_copy(uint4 *__restrict__ dst, const uint4 *__restrict__ src, uint32_t n16) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += gridDim.x * blockDim.x) dst[i] = src[i];
}

int qsvk_stage(qsv_state *st, const void *a, size_t bytes_a, const void *b, size_t bytes_b, StageRef *out) {
    const size_t off_b = qsv_pad16(bytes_a), total = off_b + bytes_b;
    if (total > QSV_STAGE_BYTES) {
        const int rc = qsvk_ensure_matrix(st, total);
        if (rc) return rc;
        char *dev = reinterpret_cast<char *>(st->dev_matrix);
        if (bytes_a) QSV_HIP(hipMemcpyAsync(dev, a, bytes_a, hipMemcpyHostToDevice, st->stream));
        if (bytes_b) QSV_HIP(hipMemcpyAsync(dev + off_b, b, bytes_b, hipMemcpyHostToDevice, st->stream));
        QSV_HIP(hipStreamSynchronize(st->stream));  // the sources are pageable host memory that dies at return
        out->dev = dev;
        out->slot = -1;
        return QSV_OK;
    }
    if (!st->stage_dev) {
        // events first, then the buffers; stage_dev is published last, so a failure half way leaves the ring absent
        // (not half built) and the next call starts over
        hipEvent_t events[QSV_STAGE_SLOTS] = {};
        for (int i = 0; i < QSV_STAGE_SLOTS; ++i)
            if (hipEventCreate(&events[i]) != hipSuccess) {
                for (int j = 0; j < i; ++j) (void)hipEventDestroy(events[j]);
                return qsv_fail(QSV_EHIP, "event creation for the gate-matrix staging ring failed");
            }
        char *host = nullptr, *dev = nullptr;
        if (hipHostMalloc(reinterpret_cast<void **>(&host), QSV_STAGE_SLOTS * QSV_STAGE_BYTES, 0) != hipSuccess ||
            hipMalloc(reinterpret_cast<void **>(&dev), QSV_STAGE_SLOTS * QSV_STAGE_BYTES) != hipSuccess) {
            if (host) (void)hipHostFree(host);
            for (hipEvent_t ev : events) (void)hipEventDestroy(ev);
            return qsv_fail(QSV_ENOMEM, "allocation of the gate-matrix staging ring failed");
        }
        for (int i = 0; i < QSV_STAGE_SLOTS; ++i) st->stage_done[i] = events[i];
        st->stage_host = host;
        st->stage_dev = dev;
    }
    const int slot = static_cast<int>(st->stage_next++ % QSV_STAGE_SLOTS);
    if (st->stage_busy[slot]) {
        QSV_HIP(hipEventSynchronize(st->stage_done[slot]));
        st->stage_busy[slot] = false;
    }
    char *host = st->stage_host + slot * QSV_STAGE_BYTES, *dev = st->stage_dev + slot * QSV_STAGE_BYTES;
    if (bytes_a) std::memcpy(host, a, bytes_a);
    if (bytes_b) std::memcpy(host + off_b, b, bytes_b);
    const uint32_t n16 = static_cast<uint32_t>((total + 15) / 16);
    hipLaunchKernelGGL(k_stage_copy, dim3((n16 + QSV_BLOCK - 1) / QSV_BLOCK < 16 ? (n16 + QSV_BLOCK - 1) / QSV_BLOCK : 16), dim3(QSV_BLOCK), 0,
                       st->stream, reinterpret_cast<uint4 *>(dev), reinterpret_cast<const uint4 *>(host), n16);
    QSV_HIP(hipGetLastError());
    // the slot is protected from here on: whatever the caller does next (its launch may fail, it may return early), the
    // pinned slot is not rewritten before this copy kernel has read it.  qsvk_stage_done moves the mark behind the consumer.
    QSV_HIP(hipEventRecord(st->stage_done[slot], st->stream));
    st->stage_busy[slot] = true;
    out->dev = dev;
    out->slot = slot;
    return QSV_OK;
}

int qsvk_stage_done(qsv_state *st, const StageRef &ref) {
    if (ref.slot < 0) return QSV_OK;
    QSV_HIP(hipEventRecord(st->stage_done[ref.slot], st->stream));   // re-record: now behind the kernel that reads the slot
    st->stage_busy[ref.slot] = true;
    return QSV_OK;
}

// Out-of-place operations (measure, insert, permute, the mode contractions) write into the state's spare
// buffer and then swap it in (qsvk_adopt).  The spare is kept between calls: a circuit that alternates such
// operations ping-pongs between two allocations instead of paying hipMalloc/hipFree of the register per gate
// (measured: ~0.44 s per gate on a 16 GiB register).
int qsvk_scratch(qsv_state *st, uint64_t amps, amp_t **out) {
    if (amps == 0) amps = 1;
    if (st->spare_capacity < amps) {
        if (st->spare) {
            QSV_HIP(hipStreamSynchronize(st->stream));
            QSV_HIP(hipFree(st->spare));
            st->spare = nullptr;
            st->spare_capacity = 0;
        }
        if (hipMalloc(reinterpret_cast<void **>(&st->spare), sizeof(amp_t) * amps) != hipSuccess)
            return qsv_fail(QSV_ENOMEM, "device allocation of the spare register failed");
        st->spare_capacity = amps;
    }
    *out = st->spare;
    return QSV_OK;
}

// The spare buffer now holds the register (new_amps amplitudes): swap it in when the library owns the memory,
// copy it back when the caller does (a view's pointer must stay valid).
int qsvk_adopt(qsv_state *st, uint64_t new_amps) {
    if (st->owns_data) {
        std::swap(st->data, st->spare);
        std::swap(st->capacity, st->spare_capacity);
    } else {
        QSV_HIP(hipMemcpyAsync(st->data, st->spare, sizeof(amp_t) * new_amps, hipMemcpyDeviceToDevice, st->stream));
    }
    st->amps = new_amps;
    return QSV_OK;
}

// k = 3..5 on any register with at least k qubits.
template <int K, int KL>
static void launch_big_kernel(qsv_state *st, bool nt, dim3 gd, const BigArgs &g, const double *dev_m, const uint64_t *dev_off,
                              bool m3 = false) {
    const dim3 bd(QSV_BLOCK);
    if constexpr (K == 5 && KL == 0) {
        if (m3) {
            if (nt) hipLaunchKernelGGL((k_dense_big<K, KL, true, true>), gd, bd, 0, st->stream, st->data, g, dev_m, dev_off);
            else hipLaunchKernelGGL((k_dense_big<K, KL, false, true>), gd, bd, 0, st->stream, st->data, g, dev_m, dev_off);
            return;
        }
    }
    if (nt) hipLaunchKernelGGL((k_dense_big<K, KL, true>), gd, bd, 0, st->stream, st->data, g, dev_m, dev_off);
    else hipLaunchKernelGGL((k_dense_big<K, KL, false>), gd, bd, 0, st->stream, st->data, g, dev_m, dev_off);
}

// ---- k-qubit dense gate, targets on bits >= 3, staged through LDS: "tile" form -------------------------------------
// k_dense_big gives every thread a whole 2^K-amplitude column (128 VGPRs at K = 5: two or three waves per SIMD that
// load, compute and store in lock step).  Here a workgroup of 2^K / ROWS waves owns 64 columns: wave q loads inputs
// q ROWS .. q ROWS + ROWS - 1 of every column (1 KiB per wave-instruction: the 64 columns are the 64 lowest free index
// values, so lanes 8j..8j+7 cover one whole 128-byte line whatever the target bits >= 3 are), parks them in a
// [2^K][64] LDS tile, and after one barrier computes outputs q ROWS .. q ROWS + ROWS - 1 of its lane's column: the 2^K
// inputs come back from LDS one 16-byte read each (lane-contiguous, conflict-free), the ROWS x 2^K slice of the matrix
// is wave-uniform and arrives as SGPR operands.  ROWS accumulators instead of 2^K amplitudes per thread: five
// workgroups per CU (LDS-bound at K = 5), whose load / compute / store phases overlap.
template <int K, int ROWS, bool REAL, bool NT, bool M3 = false>
__global__ __launch_bounds__((1 << K) / ROWS * 64) void k_dense_tile(amp_t *__restrict__ a, const BigArgs g,
                                                                     const double *__restrict__ mat,   // [q][c][ROWS]
                                                                     const uint64_t *__restrict__ off) {
    static_assert(!(REAL && M3), "a real matrix needs two multiplications per entry anyway");
    constexpr int PER = REAL ? 1 : M3 ? 3 : 2;   // doubles per matrix entry: re | (re, im) | (re, im, re + im)
    constexpr int D = 1 << K;
    __shared__ amp_t tile[D * 64];
    const int lane = threadIdx.x & 63;
    const int q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // tile order as in k_dense: `regions` contiguous pieces of this launch's range walked side by side
    const uint64_t tile_id = (g.regions > 1 && gridDim.x % g.regions == 0)
                                 ? (blockIdx.x % g.regions) * (gridDim.x / g.regions) + blockIdx.x / g.regions
                                 : blockIdx.x;
    const uint64_t base = deposit(g.w0 + tile_id * 64 + lane, g);
    uint64_t o[ROWS];
#pragma unroll
    for (int i = 0; i < ROWS; ++i) o[i] = off[q * ROWS + i];
    // HBM -> LDS without a stop in the registers (global_load_lds_dwordx4: lane l of the wave lands at the row's base + 16 l)
#if defined(__HIP_DEVICE_COMPILE__)   // the builtin exists in the device pass only
#pragma unroll
    for (int i = 0; i < ROWS; ++i)
        __builtin_amdgcn_global_load_lds(a + base + o[i], tile + (q * ROWS + i) * 64, 16, 0, NT ? 2 : 0);
#endif
    __syncthreads();
    amp_t acc[ROWS];
    double s3[M3 ? ROWS : 1];   // 3M form (see row_product_3m): acc.x = sum Ar xr, acc.y = sum Ai xi, s3 = sum As xs
#pragma unroll
    for (int i = 0; i < ROWS; ++i) acc[i] = amp_t{0.0, 0.0};
    if constexpr (M3) {
#pragma unroll
        for (int i = 0; i < ROWS; ++i) s3[i] = 0.0;
    }
    const double *m = mat + static_cast<size_t>(q) * D * ROWS * PER;
#pragma unroll 8
    for (int c = 0; c < D; ++c) {
        const amp_t v = tile[c * 64 + lane];
        const double *mc = m + c * ROWS * PER;
        [[maybe_unused]] const double vs = v.x + v.y;
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            if constexpr (REAL) {
                acc[i].x = fma(mc[i], v.x, acc[i].x);
                acc[i].y = fma(mc[i], v.y, acc[i].y);
            } else if constexpr (M3) {
                acc[i].x = fma(mc[3 * i], v.x, acc[i].x);
                acc[i].y = fma(mc[3 * i + 1], v.y, acc[i].y);
                s3[i] = fma(mc[3 * i + 2], vs, s3[i]);
            } else {
                acc[i] = cfma(cplx{mc[2 * i], mc[2 * i + 1]}, v, acc[i]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
        amp_t out = acc[i];
        if constexpr (M3) out = amp_t{acc[i].x - acc[i].y, s3[i] - acc[i].x - acc[i].y};
        if (NT) __builtin_nontemporal_store(out, a + base + o[i]);
        else a[base + o[i]] = out;
    }
}

// ---- a fused block as the SEQUENCE of its source gates on an LDS-resident tile (round 3) -------------------------------
// Dense 5- and 6-qubit blocks are the gate shapes bound by arithmetic (4 x 32 / 3 x 64 real multiply-adds per amplitude at
// a power-limited clock: 1.6 / 1.9 ms against the 1.31 ms of a pass over HBM), yet a fused block is the product of a
// handful of 1- and 2-qubit gates worth 4-8 multiply-adds each.  Here a workgroup brings a 4096-amplitude tile into LDS
// -- the block's K target bits plus the 12 - K lowest other bits: every row of the tile is 64 contiguous amplitudes (bits
// 0-5 are always in the tile when K <= 6), so HBM sees whole 1 KiB runs in both directions, straight into LDS on the way
// in -- and applies the source gates one after the other on that 12-qubit register (a barrier between gates), as the
// single-launch executor does with whole registers (qsv_circuit.hip).  Two workgroups per CU: one computes while the
// other loads or stores.
// The gate list is cut into PASSES over at most four tile bits each (consecutive gates whose legs fit four bits together):
// a thread takes the 16 amplitudes of one group of the pass's four bits into registers, applies the pass's gates there
// (seq_apply*: compile-time register indices behind a wave-uniform switch) and puts them back -- one LDS round trip and one
// barrier per pass instead of one per gate.
struct TilePass {
    int32_t first, count;      // gates [first, first + count) of the SeqGate list (codes relative to the pass's four bits)
    int32_t q[4];              // the pass's tile bits, ascending
    int32_t pad[2];
};

constexpr int TILE_SEQ_BITS = 12, TILE_SEQ_ROWS = 1 << (TILE_SEQ_BITS - 6), TILE_SEQ_THREADS = 256, TILE_SEQ_MAX_PASSES = 24;
constexpr int TILE_SEQ_ROWS_PER_WAVE = TILE_SEQ_ROWS / (TILE_SEQ_THREADS / 64);

struct TileSeqArgs {
    BigArgs g;              // pos[] = ALL tile bits: the tile number is deposited around them
    uint32_t lane_bit[6];   // address bit of lane bit j (the six lowest tile bits: 0..5 unless a 6-qubit block sits above them)
};

__device__ __forceinline__ void seq_run16(amp_t (&x)[16], const SeqGate *__restrict__ gates, int n_gates) {
#pragma unroll 1
    for (int g = 0; g < n_gates; ++g) {
        const double *m = gates[g].m;
        switch (gates[g].code) {     // wave-uniform (scalar loads): 0..3 one-qubit gates, 5 + p two-qubit gates as in SeqGate
            case 0: seq_apply1<0, 16>(x, m); break;
            case 1: seq_apply1<1, 16>(x, m); break;
            case 2: seq_apply1<2, 16>(x, m); break;
            case 3: seq_apply1<3, 16>(x, m); break;
            case 5: seq_apply2<1, 0, 16>(x, m); break;
            case 6: seq_apply2<2, 0, 16>(x, m); break;
            case 7: seq_apply2<2, 1, 16>(x, m); break;
            case 8: seq_apply2<3, 0, 16>(x, m); break;
            case 9: seq_apply2<3, 1, 16>(x, m); break;
            default: seq_apply2<3, 2, 16>(x, m); break;
        }
    }
}

template <bool NT>
__global__ __launch_bounds__(TILE_SEQ_THREADS) void k_seq_tile(amp_t *__restrict__ a, const TileSeqArgs ta,
                                                              const SeqGate *__restrict__ gates,
                                                              const TilePass *__restrict__ passes, int n_passes,
                                                              const uint64_t *__restrict__ off) {   // [rows] row offsets
    const BigArgs &g = ta.g;
    extern __shared__ __attribute__((aligned(16))) char seq_smem[];
    amp_t *tile = reinterpret_cast<amp_t *>(seq_smem);        // [rows][64 lanes] = the tile register, index row * 64 + lane
    const int lane = threadIdx.x & 63;
    const int q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // wave q: rows q RPW .. q RPW + RPW - 1
    constexpr int RPW = TILE_SEQ_ROWS_PER_WAVE;
    const uint64_t tile_id = (g.regions > 1 && gridDim.x % g.regions == 0)
                                 ? (blockIdx.x % g.regions) * (gridDim.x / g.regions) + blockIdx.x / g.regions
                                 : blockIdx.x;
    uint64_t base = deposit(g.w0 + tile_id, g);
#pragma unroll
    for (int j = 0; j < 6; ++j) base |= static_cast<uint64_t>((lane >> j) & 1) << ta.lane_bit[j];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int i = 0; i < RPW; ++i)
        __builtin_amdgcn_global_load_lds(a + base + off[q * RPW + i], tile + (q * RPW + i) * 64, 16, 0, NT ? 2 : 0);
#endif
    __syncthreads();
#pragma unroll 1
    for (int p = 0; p < n_passes; ++p) {
        const TilePass &ps = passes[p];
        // the groups of the pass: their index bits with zeros inserted at the pass's four tile bits
        for (uint32_t grp = threadIdx.x; grp < (1u << (TILE_SEQ_BITS - 4)); grp += TILE_SEQ_THREADS) {
            uint32_t g0 = grp;
#pragma unroll
            for (int j = 0; j < 4; ++j) g0 = static_cast<uint32_t>(insert_zero(g0, ps.q[j]));
            amp_t x[16];
#pragma unroll
            for (int c = 0; c < 16; ++c)
                x[c] = tile[g0 | ((c & 1) << ps.q[0]) | (((c >> 1) & 1) << ps.q[1]) | (((c >> 2) & 1) << ps.q[2]) | (((c >> 3) & 1) << ps.q[3])];
            seq_run16(x, gates + ps.first, ps.count);
#pragma unroll
            for (int c = 0; c < 16; ++c)
                tile[g0 | ((c & 1) << ps.q[0]) | (((c >> 1) & 1) << ps.q[1]) | (((c >> 2) & 1) << ps.q[2]) | (((c >> 3) & 1) << ps.q[3])] = x[c];
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < RPW; ++i) st<NT>(a + base + off[q * RPW + i], tile[(q * RPW + i) * 64 + lane]);
}

// Complex 5-qubit blocks: the LDS tile of k_dense_tile feeding the f64 MATRIX CORES (round 3).  The vector kernels spend
// 4 x 32 FP64 FMAs per amplitude; rocprofv3's SQ counters put their vector pipe at 81 % busy over the whole launch, at a
// clock that the FP64 load pulls down to ~1.65 GHz (a 1-qubit gate runs at 2.3): they are bound by arithmetic, and the
// three-multiplication form does not help them because its third plane of matrix rows (24 KiB per gate) no longer fits
// the scalar cache the rows stream through.  Here the matrix lives in REGISTERS for the whole launch -- lane (i, kk)
// holds M[16 t + i][4 s + kk] of every (row tile t, slice s): 32 complex values + their sums, 96 registers -- as the A
// operand of v_mfma_f64_16x16x4_f64, and a complex product is three MFMAs (S1 += Ar xr, S2 += Ai xi, S3 += (Ar + Ai)(xr +
// xi)): 48 MFMAs per 16 groups instead of 4096 wave-FMAs.  The HBM side is k_dense_tile's: a workgroup owns 64 columns,
// wave q brings input rows 8 q .. 8 q + 7 straight into the [32][64] LDS tile (1 KiB per wave-instruction), then takes
// the 16 columns 16 q .. 16 q + 15 through the matrix cores (B operand: lane (j, kk) reads [4 s + kk][16 q + j]), puts
// the results back into its own columns of the tile, and after a barrier stores rows 8 q .. again as whole 1 KiB runs.
// Persistent workgroups (the matrix registers are loaded once).
template <bool NT>
__global__ __launch_bounds__(QSV_BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_dense_mtile5(
    amp_t *__restrict__ a, const BigArgs g, const double *__restrict__ mat /* [r][c] (re, im) */,
    const uint64_t *__restrict__ off) {
    constexpr int D = 32, ROWS = 8;
    __shared__ amp_t tiles[2][D * 64];   // two tiles: the next one is on its way from HBM while this one is computed and stored
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    const int q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double are[8][2], aim[8][2], asum[8][2];
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const double *e = mat + 2 * ((16 * t + li) * D + 4 * s + lk);
            are[s][t] = e[0];
            aim[s][t] = e[1];
            asum[s][t] = e[0] + e[1];
        }
    uint64_t o[ROWS];
#pragma unroll
    for (int i = 0; i < ROWS; ++i) o[i] = off[q * ROWS + i];
    const uint64_t n_tiles = g.W / 64;
    auto base_of = [&](uint64_t t0) {
        const uint64_t tile_id = (g.regions > 1 && n_tiles % g.regions == 0) ? (t0 % g.regions) * (n_tiles / g.regions) + t0 / g.regions : t0;
        return deposit(g.w0 + tile_id * 64 + lane, g);
    };
    auto fetch = [&](uint64_t base, amp_t *tile) {
#if defined(__HIP_DEVICE_COMPILE__)   // the builtin exists in the device pass only
#pragma unroll
        for (int i = 0; i < ROWS; ++i)
            __builtin_amdgcn_global_load_lds(a + base + o[i], tile + (q * ROWS + i) * 64, 16, 0, NT ? 2 : 0);
#endif
    };
    // Barriers are raw s_barrier + counted waits: __syncthreads() carries a fence that drains the vector-memory counter, i.e.
    // waits for the NEXT tile's loads (issued just before) and for this tile's stores.  The counter retires loads, stores and
    // LDS-DMA in issue order, so at the top of an iteration "all but the 8 youngest" = the previous tile's 8 stores may stay
    // in flight while this tile's 8 loads (older) are complete.
    uint64_t t0 = blockIdx.x;
    uint64_t base = base_of(t0);
    fetch(base, tiles[0]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int it = 0; t0 < n_tiles; t0 += gridDim.x, ++it) {
        amp_t *tile = tiles[it & 1];
        // this tile's loads have landed (every wave waits for its own, then the barrier), and nobody reads the other tile
        // any more (the LDS reads of the previous store phase have returned)
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const uint64_t next = t0 + gridDim.x;
        const uint64_t next_base = base_of(next < n_tiles ? next : blockIdx.x);   // past the end: re-read the first tile, unused
        fetch(next_base, tiles[(it & 1) ^ 1]);
        f64x4 s1[2], s2[2], s3[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) s1[t] = s2[t] = s3[t] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const amp_t x = tile[(4 * s + lk) * 64 + 16 * q + li];
            const double xs = x.x + x.y;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                s1[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(are[s][t], x.x, s1[t], 0, 0, 0);
                s2[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(aim[s][t], x.y, s2[t], 0, 0, 0);
                s3[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(asum[s][t], xs, s3[t], 0, 0, 0);
            }
        }
        // this wave has read everything it needs from its 16 columns (the reads are MFMA operands above): results in place
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                tile[(16 * t + lk + 4 * r) * 64 + 16 * q + li] = amp_t{s1[t][r] - s2[t][r], s3[t][r] - s1[t][r] - s2[t][r]};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my results are in the tile; the next tile's loads stay in flight
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            const amp_t v = tile[(q * ROWS + i) * 64 + lane];
            if (NT) __builtin_nontemporal_store(v, a + base + o[i]);
            else a[base + o[i]] = v;
        }
        base = next_base;
    }
}

// 1- and 2-qubit gates in the same form: one wave per input row (2 / 4 waves per workgroup), the matrix and the row
// offsets in the kernel arguments.  On the benchmark circuit's placements 4-8 % faster than the register form (k_dense):
// 1.28-1.34 ms against 1.31-1.48 per 1-qubit gate at n = 28, 1.36 against 1.48 on average over all pairs of bits >= 6.
struct SmallGate {
    double m[32];      // [row][col] (re, im), kernel index bit i <-> leg i
    uint64_t off[4];   // amplitude offset of input / output row c
};

template <int K, bool NT>
__device__ __forceinline__ void tile12_body(amp_t *__restrict__ a, const BigArgs &g, const SmallGate &sg) {
    constexpr int D = 1 << K;
    __shared__ amp_t tile[D * 64];
    const int lane = threadIdx.x & 63;
    const int q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t tile_id = (g.regions > 1 && gridDim.x % g.regions == 0)
                                 ? (blockIdx.x % g.regions) * (gridDim.x / g.regions) + blockIdx.x / g.regions
                                 : blockIdx.x;
    amp_t *p = a + deposit(g.w0 + tile_id * 64 + lane, g) + sg.off[q];
#if defined(__HIP_DEVICE_COMPILE__)   // the builtin exists in the device pass only
    __builtin_amdgcn_global_load_lds(p, tile + q * 64, 16, 0, NT ? 2 : 0);
#endif
    __syncthreads();
    amp_t acc = {0.0, 0.0};
#pragma unroll
    for (int c = 0; c < D; ++c)
        acc = cfma(cplx{sg.m[2 * (q * D + c)], sg.m[2 * (q * D + c) + 1]}, tile[c * 64 + lane], acc);
    if (NT) __builtin_nontemporal_store(acc, p);
    else *p = acc;
}

template <int K, bool NT>
__global__ __launch_bounds__((1 << K) * 64) void k_dense_tile12(amp_t *__restrict__ a, const BigArgs g, const SmallGate sg) {
    tile12_body<K, NT>(a, g, sg);
}

// the same body on a sub-space (controls removed from the enumeration and forced to 1): a symbol of its own, so that
// profiles keep full- and reduced-traffic launches apart (as k_dense / k_dense_ctrl do)
template <int K, bool NT>
__global__ __launch_bounds__((1 << K) * 64) void k_dense_tile12_ctrl(amp_t *__restrict__ a, const BigArgs g, const SmallGate sg) {
    tile12_body<K, NT>(a, g, sg);
}

template <int K, int ROWS>
static int launch_tile_kernel(qsv_state *st, bool nt, bool realm, dim3 gd, const BigArgs &g, const double *m,
                              const uint64_t *dev_off, bool m3 = false) {
    const dim3 bd((1 << K) / ROWS * 64);
    if constexpr (K == 5) {
        if (m3) {
            if (nt) hipLaunchKernelGGL((k_dense_tile<K, ROWS, false, true, true>), gd, bd, 0, st->stream, st->data, g, m, dev_off);
            else hipLaunchKernelGGL((k_dense_tile<K, ROWS, false, false, true>), gd, bd, 0, st->stream, st->data, g, m, dev_off);
            return check_launch();
        }
    }
    if (nt) {
        if (realm) hipLaunchKernelGGL((k_dense_tile<K, ROWS, true, true>), gd, bd, 0, st->stream, st->data, g, m, dev_off);
        else hipLaunchKernelGGL((k_dense_tile<K, ROWS, false, true>), gd, bd, 0, st->stream, st->data, g, m, dev_off);
    } else {
        if (realm) hipLaunchKernelGGL((k_dense_tile<K, ROWS, true, false>), gd, bd, 0, st->stream, st->data, g, m, dev_off);
        else hipLaunchKernelGGL((k_dense_tile<K, ROWS, false, false>), gd, bd, 0, st->stream, st->data, g, m, dev_off);
    }
    return check_launch();
}

// Tile order of k_dense_tile / k_dense_tile12 by target placement.  Which DRAM channels the workgroups in flight hit
// together depends on the target bits; no single order wins everywhere: contiguous windows (the d = 2^K modes of the CV
// path) have a clear best order per position, scattered targets (fused qubit gates) are served well by 4 regions
// (K = 3) / 2 (K = 4) / 8 (K = 5).
// The rule is keyed on ABSOLUTE bit positions -- on the physical address bits a target toggles -- not on the distance
// from the register's top bit: the same bits want the same order on registers of 25, 26, 27, 28, 29 and 31 qubits
// (the shard sizes of the strong- and weak-scaling runs and of config 3; profiles/r03_tile_order_by_size.txt: bits 3-6
// want 2 regions, 17 / 21 / 25 eight, 20 / 22 / 23 four at every size, and a pair (lo >= 17, 27) wants four regions
// whether bit 27 is the top bit (n = 28) or not (n = 29, 31)).  Against the best of {0, 2, 4, 8, 16} regions per
// placement the rule is within 0.7-2.5 % on the sum over the sampled placements at every size.
static uint32_t tile_regions(int k, const std::vector<int> &sorted_bits) {
    if (k == 1) {   // per target bit
        const int b = sorted_bits[0];
        return b <= 6 ? 2 : b == 7 ? 8 : b <= 16 ? 0 : b == 17 ? 8 : b <= 19 ? 0 : b == 20 ? 4 : b == 21 ? 8
             : b <= 23 ? 4 : b == 25 ? 8 : 0;
    }
    if (k == 2) {   // pairs of bits >= 6 (lower targets stay on the register form)
        const int lo = sorted_bits[0], hi = sorted_bits[1];
        if (lo <= 8) {
            // 8 regions, except where both strides are short: (7|8, <= 17), (6, <= 11) and (6..8, 24) run 3-12 % faster in
            // plain order at every size
            if ((lo >= 7 && hi <= 17) || (lo == 6 && hi <= 11) || hi == 24) return 0;
            return 8;
        }
        if (hi == 20 && lo >= 12 && lo <= 16) return 0;
        if (hi >= 20 && hi <= 21) return 8;
        if (hi == 22) return lo >= 13 ? 4 : 8;
        if (hi == 23 && lo >= 20) return 8;
        if (hi == 24 && lo >= 18) return 4;
        if (hi == 26 && lo == 22) return 4;
        if ((hi == 27 && lo >= 17) || (hi == 28 && lo >= 20)) return 4;
        return 0;
    }
    const int lo = sorted_bits.front(), top = sorted_bits.back();
    const bool window = top - lo == static_cast<int>(sorted_bits.size()) - 1;
    if (k == 3) {
        if (!window) return 4;
        return top <= 8 ? 4 : top <= 19 ? 2 : top <= 22 ? 8 : top <= 24 ? 2 : 0;
    }
    if (k == 4) {
        if (!window) return 2;
        return top <= 10 ? 4 : top <= 15 ? 2 : top <= 19 ? 0 : top == 20 ? 2 : top <= 23 ? 8 : 2;
    }
    if (!window) return 8;
    return top <= 11 ? 8 : top <= 13 ? 4 : top <= 20 ? 0 : 2;
}

// the dispatches of one tile-form gate (registers beyond 2^24 tiles take several); `sub`: the reduced-traffic symbol
static int launch_tile12_kernels(qsv_state *st, int k, bool sub, BigArgs g, const SmallGate &sg) {
    const bool nt = st->nontemporal != 0;
    snprintf(st->last_kernel, sizeof(st->last_kernel), "k_dense_tile12%s<%d, %s>", sub ? "_ctrl" : "", k, nt ? "true" : "false");
    // columns per dispatch: a power of two, so that every dispatch of a split launch has a tile count the region
    // order divides (with 2^24 - 1 tiles per dispatch a 31-qubit shard ran its 1-qubit gates in plain order: 12.0 ms
    // on bits 3..6 against 10.9 for the register form, profiles/r03_tile_order_by_size.txt)
    const uint64_t per_launch = (1ull << 23) * 64;
    for (g.w0 = 0; g.w0 < g.W; g.w0 += per_launch) {
        const dim3 gd(static_cast<unsigned>(std::min(per_launch, g.W - g.w0) / 64)), bd((1 << k) * 64);
        if (sub) {
            if (nt) hipLaunchKernelGGL((k_dense_tile12_ctrl<1, true>), gd, bd, 0, st->stream, st->data, g, sg);
            else hipLaunchKernelGGL((k_dense_tile12_ctrl<1, false>), gd, bd, 0, st->stream, st->data, g, sg);
        } else if (k == 1) {
            if (nt) hipLaunchKernelGGL((k_dense_tile12<1, true>), gd, bd, 0, st->stream, st->data, g, sg);
            else hipLaunchKernelGGL((k_dense_tile12<1, false>), gd, bd, 0, st->stream, st->data, g, sg);
        } else {
            if (nt) hipLaunchKernelGGL((k_dense_tile12<2, true>), gd, bd, 0, st->stream, st->data, g, sg);
            else hipLaunchKernelGGL((k_dense_tile12<2, false>), gd, bd, 0, st->stream, st->data, g, sg);
        }
        const int rc = check_launch();
        if (rc) return rc;
    }
    return QSV_OK;
}

static int launch_tile12(qsv_state *st, int k, const int *bits, int nctrl, const int *cbits, const double *m_user) {
    const int D = 1 << k;
    if (k + nctrl > 2 * QSV_MAX_K || (nctrl && k != 1)) return QSV_UNHANDLED_KQ;   // controlled 4 x 4 gates only arise with a folded narrow control
    for (int i = 0; i < nctrl; ++i)
        if (cbits[i] < 3) return QSV_UNHANDLED_KQ;   // a control inside a 128-byte line cannot be skipped
    const uint64_t W = st->amps >> (k + nctrl);
    const int lowest = k == 1 ? bits[0] : std::min(bits[0], bits[1]);
    // 2-qubit gates with a target inside a wavefront (bits 3..5) are better off with k_dense<1, 1> (1.29-1.37 ms)
    if (lowest < (k == 1 ? 3 : QSV_LANE_BITS) || W < 64 || W % 64) return QSV_UNHANDLED_KQ;
    SmallGate sg;
    std::memset(&sg, 0, sizeof(sg));
    int ui[4];
    for (int c = 0; c < D; ++c) {
        ui[c] = 0;
        for (int leg = 0; leg < k; ++leg) {
            if ((c >> leg) & 1) sg.off[c] |= 1ull << bits[leg];
            ui[c] |= ((c >> leg) & 1) << (k - 1 - leg);
        }
    }
    for (int r = 0; r < D; ++r)
        for (int c = 0; c < D; ++c) {
            sg.m[2 * (r * D + c)] = m_user[2 * (ui[r] * D + ui[c])];
            sg.m[2 * (r * D + c) + 1] = m_user[2 * (ui[r] * D + ui[c]) + 1];
        }
    std::vector<int> sorted(bits, bits + k);
    std::sort(sorted.begin(), sorted.end());
    std::vector<int> ins(sorted);
    ins.insert(ins.end(), cbits, cbits + nctrl);
    std::sort(ins.begin(), ins.end());
    BigArgs g;
    std::memset(&g, 0, sizeof(g));
    g.W = W;
    g.nins = k + nctrl;
    for (int j = 0; j < k + nctrl; ++j) g.pos[j] = static_cast<uint32_t>(ins[j]);
    for (int i = 0; i < nctrl; ++i) g.or_mask |= 1ull << cbits[i];
    // controlled launches (CX on 40 random control / target pairs: 0.686 ms with 8 regions, 0.734 for k_dense_ctrl)
    g.regions = st->remap >= 0 ? static_cast<uint32_t>(st->remap) : nctrl ? 8 : tile_regions(k, sorted);
    return launch_tile12_kernels(st, k, nctrl != 0, g, sg);
}

template <int K>
static int dispatch_big(qsv_state *st, int KL, bool nt, dim3 gd, const BigArgs &g, const double *dev_m, const uint64_t *dev_off,
                        bool m3 = false) {
    switch (KL) {
        case 0: launch_big_kernel<K, 0>(st, nt, gd, g, dev_m, dev_off, m3); break;
        case 1: launch_big_kernel<K, 1>(st, nt, gd, g, dev_m, dev_off); break;
        case 2: launch_big_kernel<K, 2>(st, nt, gd, g, dev_m, dev_off); break;
        case 3: launch_big_kernel<K, 3>(st, nt, gd, g, dev_m, dev_off); break;
        case 4:
            if constexpr (K >= 4) launch_big_kernel<K, 4>(st, nt, gd, g, dev_m, dev_off);
            break;
        default:
            if constexpr (K >= 5) launch_big_kernel<K, 5>(st, nt, gd, g, dev_m, dev_off);
            break;
    }
    return check_launch();
}



template <int K, int KB, int BLOCK>
static void launch_lds_kernel(qsv_state *st, bool nt, bool realm, dim3 gd, const LdsArgs &g, const double *dev_m,
                              const uint64_t *dev_off, bool m3 = false) {
    const dim3 bd(BLOCK);
    if constexpr (K == 5) {
        if (m3) {
            if (nt) hipLaunchKernelGGL((k_dense_lds<K, KB, true, false, BLOCK, true>), gd, bd, 0, st->stream, st->data, g, dev_m, dev_off);
            else hipLaunchKernelGGL((k_dense_lds<K, KB, false, false, BLOCK, true>), gd, bd, 0, st->stream, st->data, g, dev_m, dev_off);
            return;
        }
    }
    if (nt) {
        if (realm) hipLaunchKernelGGL((k_dense_lds<K, KB, true, true, BLOCK>), gd, bd, 0, st->stream, st->data, g, dev_m, dev_off);
        else hipLaunchKernelGGL((k_dense_lds<K, KB, true, false, BLOCK>), gd, bd, 0, st->stream, st->data, g, dev_m, dev_off);
    } else {
        if (realm) hipLaunchKernelGGL((k_dense_lds<K, KB, false, true, BLOCK>), gd, bd, 0, st->stream, st->data, g, dev_m, dev_off);
        else hipLaunchKernelGGL((k_dense_lds<K, KB, false, false, BLOCK>), gd, bd, 0, st->stream, st->data, g, dev_m, dev_off);
    }
}

template <int K, int BLOCK>
static int dispatch_lds(qsv_state *st, int KB, bool nt, bool realm, dim3 gd, const LdsArgs &g, const double *dev_m,
                        const uint64_t *dev_off, bool m3 = false) {
    switch (KB) {
        case 0: launch_lds_kernel<K, 0, BLOCK>(st, nt, realm, gd, g, dev_m, dev_off, m3); break;
        case 1: launch_lds_kernel<K, 1, BLOCK>(st, nt, realm, gd, g, dev_m, dev_off, m3); break;
        case 2: launch_lds_kernel<K, 2, BLOCK>(st, nt, realm, gd, g, dev_m, dev_off, m3); break;
        default: launch_lds_kernel<K, 3, BLOCK>(st, nt, realm, gd, g, dev_m, dev_off, m3); break;
    }
    return check_launch();
}

static bool mtile_default() {
    static const bool on = [] { const char *e = getenv("QSV_MTILE"); return e ? atoi(e) != 0 : false; }();
    return on;
}

// k = 3..5 on any register with at least k qubits.  bits[j] = bit position of matrix leg j (leg 0 most significant).
static int launch_dense_big(qsv_state *st, int k, const int *bits, const double *m_user) {
    const int D = 1 << k;
    std::vector<int> high, low;
    for (int j = 0; j < k; ++j) (bits[j] >= QSV_LANE_BITS ? high : low).push_back(bits[j]);
    std::sort(low.begin(), low.end());  // bits 0..2 (inside a 128-byte line) first, then bits 3..5
    int KL = static_cast<int>(low.size());
    // stand-in bits for the low targets: the lowest free bits >= 6 (needs n >= k + 6)
    std::vector<int> standin;
    for (int b = QSV_LANE_BITS; b < st->n && static_cast<int>(standin.size()) < KL; ++b)
        if (std::find(high.begin(), high.end(), b) == high.end()) standin.push_back(b);
    bool all_from_bit3 = true;
    for (int j = 0; j < k; ++j) all_from_bit3 = all_from_bit3 && bits[j] >= 3;
    const bool tile_ok = k >= 3 && k <= 5 && all_from_bit3 && (st->amps >> k) >= 64 && (st->amps >> k) % 64 == 0;
    bool real_matrix = true;
    for (int i = 0; i < D * D && real_matrix; ++i) real_matrix = m_user[2 * i + 1] == 0.0;
    // shipped choice: k = 3, 4, and k = 5 with a real matrix (a complex 32 x 32 product per column keeps the FP64 pipe busy
    // for 0.9 of the 1.4 ms the memory traffic takes; the tile form's extra LDS round trip then costs more than it hides.
    // A persistent form with two LDS tiles and the next tile's loads in flight during the arithmetic was measured too:
    // 1.86 ms -- two workgroups per CU leave the FMA chains exposed to the scalar-load and LDS latencies)
    // complex 5-qubit blocks on bits >= 3: the tile-fed matrix-core kernel (k_dense_mtile5; QSV_OPT_KQ_VARIANT = 6 forces it)
    const bool use_mtile = tile_ok && k == 5 && !real_matrix && st->complex_product != 4 &&
                           (st->kq_variant == 6 || (st->kq_variant == 0 && mtile_default()));
    const bool use_tile = use_mtile || (tile_ok && (st->kq_variant == 4 || (st->kq_variant == 0 && (k <= 4 || real_matrix))));
    const bool transposed = KL > 0 && static_cast<int>(standin.size()) == KL && st->kq_variant != 2 && !use_tile;
    if (!transposed) {  // all targets high, or a register too small to transpose: lanes = lowest free bits
        high.assign(bits, bits + k);
        low.clear();
        standin.clear();
        KL = 0;
    }
    const int KH = k - KL;
    // Which form (MI355X, n = 28, profiles/r02_sweep_kq_kernels.txt): k = 5 with low targets -> the line-granular
    // kernel (4.9-5.2 TB/s at every placement; the shuffle form drops to 2.1-4.4 there); k = 5 real matrices ->
    // the same kernel's two-FMA arithmetic (5.4-5.8 TB/s); k = 3, 4 and k = 5 on high bits -> the shuffle form
    // (its butterflies are cheap up to 16 amplitudes per thread: 5.6-6.0 TB/s).  QSV_OPT_KQ_VARIANT overrides.
    const bool fits = (st->amps >> k) >= 64 && (st->amps >> k) % 64 == 0;
    const bool use_lds = !use_tile && fits && (k == 6 || st->kq_variant == 3 ||
                                  (st->kq_variant == 0 && k == 5 && (KL > 0 || real_matrix)));
    if (k == 6 && !use_lds) return QSV_UNHANDLED_KQ;  // only the line-granular kernel is built for 64 x 64 matrices
    int KB = 0;
    for (int b : low) KB += b < 3;
    // register index c = (h << KL) | t: h bit i <-> high[i], t bit j <-> low[j] (stored at stand-in bit standin[j])
    std::vector<uint64_t> off(D, 0);
    for (int c = 0; c < D; ++c) {
        for (int i = 0; i < KH; ++i)
            if ((c >> (KL + i)) & 1) off[c] |= 1ull << high[i];
        for (int j = 0; j < KL; ++j)
            if ((c >> j) & 1) off[c] |= 1ull << ((use_lds && low[j] >= 3) ? low[j] : standin[j]);
    }
    auto user_index = [&](int c) {  // kernel register index -> index of the caller's matrix
        int u = 0;
        for (int leg = 0; leg < k; ++leg) {
            int v = 0;
            for (int i = 0; i < KH; ++i)
                if (high[i] == bits[leg]) v = (c >> (KL + i)) & 1;
            for (int j = 0; j < KL; ++j)
                if (low[j] == bits[leg]) v = (c >> j) & 1;
            u |= v << (k - 1 - leg);
        }
        return u;
    };
    const bool realm = (use_lds || use_tile) && real_matrix;
    // complex 5-qubit blocks in three real multiplications per entry (row_product_3m): a measurement variant only
    // (QSV_OPT_COMPLEX_PRODUCT = 3).  On the vector pipe it does not pay (profiles/r03_complex_product.txt): a quarter
    // fewer FMAs, but a third plane of matrix rows through the scalar cache and xr + xi in 64 more registers -- 1.64-1.71
    // ms against 1.61-1.77 for k_dense_big<5, 0>, 1.80-1.99 against 1.66-1.73 for k_dense_lds with targets inside a line
    // (two waves per SIMD instead of three).  These kernels are not waiting for the FP64 pipe.  On the matrix cores (k = 6)
    // the same trick is worth 13 %: see launch_dense_mfma.
    const bool m3 = !use_mtile && k == 5 && !real_matrix && st->complex_product == 3 && (use_tile || use_lds || KL == 0);
    std::vector<double> m(realm ? static_cast<size_t>(D) * D : (m3 ? 3ull : 2ull) * D * D);
    std::vector<int> ui(D);
    for (int c = 0; c < D; ++c) ui[c] = user_index(c);
    for (int r = 0; r < D; ++r)
        for (int c = 0; c < D; ++c) {
            const int ur = ui[r], uc = ui[c];
            if (realm) {
                m[r * D + c] = m_user[2 * (ur * D + uc)];
            } else if (m3 && !use_tile) {   // a row = three planes of D doubles: Ar | Ai | Ar + Ai
                const double re = m_user[2 * (ur * D + uc)], im = m_user[2 * (ur * D + uc) + 1];
                m[3 * D * r + c] = re;
                m[3 * D * r + D + c] = im;
                m[3 * D * r + 2 * D + c] = re + im;
            } else if (m3) {                // tile form: (Ar, Ai, Ar + Ai) per entry, regrouped below
                const double re = m_user[2 * (ur * D + uc)], im = m_user[2 * (ur * D + uc) + 1];
                m[3 * (r * D + c)] = re;
                m[3 * (r * D + c) + 1] = im;
                m[3 * (r * D + c) + 2] = re + im;
            } else {
                m[2 * (r * D + c)] = m_user[2 * (ur * D + uc)];
                m[2 * (r * D + c) + 1] = m_user[2 * (ur * D + uc) + 1];
            }
        }
    std::vector<int> ins(high);
    ins.insert(ins.end(), standin.begin(), standin.end());
    std::sort(ins.begin(), ins.end());
    const uint64_t W = st->amps >> k;
    if (use_tile && !use_mtile) {  // matrix slice of wave q, input c: ROWS consecutive entries  [q][c][i] = m[q ROWS + i][c]
        const int rows = k == 5 ? 8 : k == 4 ? 4 : 2, per = realm ? 1 : m3 ? 3 : 2;
        std::vector<double> mt(m.size());
        for (int r = 0; r < D; ++r)
            for (int c = 0; c < D; ++c)
                for (int e = 0; e < per; ++e)
                    mt[per * ((static_cast<size_t>(r / rows) * D + c) * rows + r % rows) + e] = m[per * (r * D + c) + e];
        m.swap(mt);
    }
    // matrix and offsets ride the staging ring to the device: queued on the stream in front of the kernel, no host wait
    StageRef staged;
    int rc = qsvk_stage(st, m.data(), sizeof(double) * m.size(), off.data(), sizeof(uint64_t) * D, &staged);
    if (rc) return rc;
    const double *dev_m = reinterpret_cast<const double *>(staged.dev);
    const uint64_t *dev_off = reinterpret_cast<const uint64_t *>(staged.dev + qsv_pad16(sizeof(double) * m.size()));
    if (use_mtile) {
        BigArgs g;
        std::memset(&g, 0, sizeof(g));
        g.W = W;
        g.nins = static_cast<int>(ins.size());
        for (size_t j = 0; j < ins.size(); ++j) g.pos[j] = static_cast<uint32_t>(ins[j]);
        const bool nt = st->nontemporal != 0;
        g.regions = st->remap >= 0 ? static_cast<uint32_t>(st->remap) : tile_regions(k, ins);
        snprintf(st->last_kernel, sizeof(st->last_kernel), "k_dense_mtile5<%s>", nt ? "true" : "false");
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, st->device);  // 256 if the query fails
        const unsigned grid = static_cast<unsigned>(std::min<uint64_t>(W / 64, 2ull * cus));
        if (nt) hipLaunchKernelGGL(k_dense_mtile5<true>, dim3(grid), dim3(QSV_BLOCK), 0, st->stream, st->data, g, dev_m, dev_off);
        else hipLaunchKernelGGL(k_dense_mtile5<false>, dim3(grid), dim3(QSV_BLOCK), 0, st->stream, st->data, g, dev_m, dev_off);
        const int rc2 = check_launch();
        if (rc2) return rc2;
        return qsvk_stage_done(st, staged);
    }
    if (use_tile) {
        const int rows = k == 5 ? 8 : k == 4 ? 4 : 2;
        BigArgs g;
        std::memset(&g, 0, sizeof(g));
        g.W = W;
        g.nins = static_cast<int>(ins.size());
        for (size_t j = 0; j < ins.size(); ++j) g.pos[j] = static_cast<uint32_t>(ins[j]);
        const bool nt = st->nontemporal != 0;  // every wave-instruction touches whole 128-byte lines
        g.regions = st->remap >= 0 ? static_cast<uint32_t>(st->remap) : tile_regions(k, ins);
        snprintf(st->last_kernel, sizeof(st->last_kernel), "k_dense_tile<%d, %d, %s, %s%s>", k, rows, realm ? "true" : "false",
                 nt ? "true" : "false", m3 ? ", true" : "");
        const uint64_t per_launch = (1ull << 23) * 64;  // columns per dispatch (a power of two: see launch_tile12_kernels)
        for (g.w0 = 0; g.w0 < g.W; g.w0 += per_launch) {
            const dim3 gd(static_cast<unsigned>(std::min(per_launch, g.W - g.w0) / 64));
            const int rc2 = k == 5 ? launch_tile_kernel<5, 8>(st, nt, realm, gd, g, dev_m, dev_off, m3)
                          : k == 4 ? launch_tile_kernel<4, 4>(st, nt, realm, gd, g, dev_m, dev_off)
                                   : launch_tile_kernel<3, 2>(st, nt, realm, gd, g, dev_m, dev_off);
            if (rc2) return rc2;
        }
        return qsvk_stage_done(st, staged);
    }
    if (use_lds) {
        LdsArgs g;
        std::memset(&g, 0, sizeof(g));
        g.W = W;
        g.nins = static_cast<int>(ins.size());
        for (size_t j = 0; j < ins.size(); ++j) g.pos[j] = static_cast<uint32_t>(ins[j]);
        for (int j = 0; j < KL; ++j) {
            if (low[j] >= 3) {
                g.abit[g.na] = low[j];
                g.aE[g.na] = standin[j];
                g.amask |= 1u << low[j];
                ++g.na;
            } else {
                g.bmask |= 1u << low[j];
            }
        }
        for (int v = 0; v < (1 << KB); ++v)
            for (int j = 0; j < KB; ++j)
                if ((v >> j) & 1) g.bdep[v] |= 1u << low[j];
        const bool nt = st->nontemporal != 0;  // every wave-instruction touches whole 128-byte lines
        snprintf(st->last_kernel, sizeof(st->last_kernel), "k_dense_lds<%d, %d, %s, %s%s>", k, KB, nt ? "true" : "false",
                 realm ? "true" : "false", m3 ? ", 256, true" : "");
        g.regions = st->remap >= 0 ? static_cast<uint32_t>(st->remap) : 8;
        const uint64_t per_launch = 0x00ffffffull * QSV_BLOCK;  // an AQL dispatch counts work-items in 32 bits
        for (g.w0 = 0; g.w0 < g.W; g.w0 += per_launch) {
            const dim3 gd(grid_for(std::min(per_launch, g.W - g.w0), QSV_BLOCK, 0));
            const int rc2 = k == 3 ? dispatch_lds<3, QSV_BLOCK>(st, KB, nt, realm, gd, g, dev_m, dev_off)
                          : k == 4 ? dispatch_lds<4, QSV_BLOCK>(st, KB, nt, realm, gd, g, dev_m, dev_off)
                          : k == 5 ? dispatch_lds<5, QSV_BLOCK>(st, KB, nt, realm, gd, g, dev_m, dev_off, m3)
                                   : dispatch_lds<6, QSV_BLOCK>(st, KB, nt, realm, gd, g, dev_m, dev_off);
            if (rc2) return rc2;
        }
        return qsvk_stage_done(st, staged);
    }
    BigArgs g;
    std::memset(&g, 0, sizeof(g));
    for (int j = 0; j < KL; ++j) g.lbit[j] = low[j];
    g.W = W;
    g.nins = static_cast<int>(ins.size());
    for (size_t j = 0; j < ins.size(); ++j) g.pos[j] = static_cast<uint32_t>(ins[j]);
    // partial-line nontemporal accesses are slow: use them only when every access is a full 1 KiB per wave
    bool coalesced = true;
    for (int b : ins) coalesced = coalesced && b >= QSV_LANE_BITS;
    const bool nt = st->nontemporal != 0 && coalesced;
    snprintf(st->last_kernel, sizeof(st->last_kernel), "k_dense_big<%d, %d, %s%s>", k, KL, nt ? "true" : "false", m3 ? ", true" : "");
    g.regions = st->remap >= 0 ? static_cast<uint32_t>(st->remap) : 8;
    const uint64_t per_launch = 0x00ffffffull * QSV_BLOCK;  // an AQL dispatch counts work-items in 32 bits
    for (g.w0 = 0; g.w0 < g.W; g.w0 += per_launch) {
        const dim3 gd(grid_for(std::min(per_launch, g.W - g.w0), QSV_BLOCK, 0));
        const int rc2 = k == 3 ? dispatch_big<3>(st, KL, nt, gd, g, dev_m, dev_off)
                      : k == 4 ? dispatch_big<4>(st, KL, nt, gd, g, dev_m, dev_off)
                               : dispatch_big<5>(st, KL, nt, gd, g, dev_m, dev_off, m3);
        if (rc2) return rc2;
    }
    return qsvk_stage_done(st, staged);
}


// A fused 5-qubit block as the sequence of its source gates (k_seq_big / k_seq_lds).  bits[j] = bit position of block leg j;
// gate g acts on block legs legs[2 g] (and legs[2 g + 1] when arity[g] == 2), its matrix follows the previous gate's in
// `mats` (8 or 32 doubles).  QSV_UNHANDLED_KQ: the register or the sequence does not fit this form (the caller applies
// the block's product matrix instead).
int qsvk_sequence5(qsv_state *st, const int *bits, int n_gates, const int *arity, const int *legs, const double *mats) {
    const int k = 5, D = 32;
    if (n_gates < 1 || n_gates > SEQ_MAX_GATES || st->n < k) return QSV_UNHANDLED_KQ;
    // multiply-adds per 32 amplitudes: 256 per one-qubit gate, 512 per two-qubit gate, 4096 for the dense block
    int work = 0;
    for (int gi = 0; gi < n_gates; ++gi) work += arity[gi] == 1 ? 256 : 512;
    if (work > (st->sequence_work >= 0 ? st->sequence_work : seq_max_work())) return QSV_UNHANDLED_KQ;
    const uint64_t W = st->amps >> k;
    if (W < 64 || W % 64) return QSV_UNHANDLED_KQ;
    std::vector<int> high, low;
    for (int j = 0; j < k; ++j) (bits[j] >= QSV_LANE_BITS ? high : low).push_back(bits[j]);
    std::sort(low.begin(), low.end());
    const int KL = static_cast<int>(low.size());
    std::vector<int> standin;
    for (int b = QSV_LANE_BITS; b < st->n && static_cast<int>(standin.size()) < KL; ++b)
        if (std::find(high.begin(), high.end(), b) == high.end()) standin.push_back(b);
    if (static_cast<int>(standin.size()) != KL) return QSV_UNHANDLED_KQ;
    const int KH = k - KL;
    int KB = 0;
    for (int b : low) KB += b < 3;
    // register index c = (h << KL) | t: h bit i <-> high[i], t bit j <-> low[j] (as launch_dense_big)
    std::vector<uint64_t> off(D, 0);
    for (int c = 0; c < D; ++c) {
        for (int i = 0; i < KH; ++i)
            if ((c >> (KL + i)) & 1) off[c] |= 1ull << high[i];
        for (int j = 0; j < KL; ++j)
            if ((c >> j) & 1) off[c] |= 1ull << (low[j] >= 3 ? low[j] : standin[j]);
    }
    auto reg_bit = [&](int phys) {
        for (int j = 0; j < KL; ++j)
            if (low[j] == phys) return j;
        for (int i = 0; i < KH; ++i)
            if (high[i] == phys) return KL + i;
        return -1;
    };
    std::vector<SeqGate> rec(n_gates);
    const double *m = mats;
    for (int gi = 0; gi < n_gates; ++gi) {
        SeqGate &r = rec[gi];
        std::memset(&r, 0, sizeof(r));
        if (arity[gi] == 1) {
            const int leg = legs[2 * gi];
            if (leg < 0 || leg >= k) return qsv_fail(QSV_EINVAL, "gate sequence: leg outside the block");
            r.code = reg_bit(bits[leg]);
            std::memcpy(r.m, m, sizeof(double) * 8);
            m += 8;
        } else if (arity[gi] == 2) {
            const int l0 = legs[2 * gi], l1 = legs[2 * gi + 1];
            if (l0 < 0 || l0 >= k || l1 < 0 || l1 >= k || l0 == l1) return qsv_fail(QSV_EINVAL, "gate sequence: legs outside the block");
            const int j0 = reg_bit(bits[l0]), j1 = reg_bit(bits[l1]);
            const int hi = std::max(j0, j1), lo = std::min(j0, j1);
            r.code = 5 + hi * (hi - 1) / 2 + lo;
            for (int rr = 0; rr < 4; ++rr)
                for (int cc = 0; cc < 4; ++cc) {
                    // record index bit 1 <-> register bit hi; the caller's index bit 1 <-> leg 0
                    const int ur = j0 > j1 ? rr : ((rr & 1) << 1) | (rr >> 1), uc = j0 > j1 ? cc : ((cc & 1) << 1) | (cc >> 1);
                    r.m[2 * (rr * 4 + cc)] = m[2 * (ur * 4 + uc)];
                    r.m[2 * (rr * 4 + cc) + 1] = m[2 * (ur * 4 + uc) + 1];
                }
            m += 32;
        } else {
            return QSV_UNHANDLED_KQ;
        }
    }
    std::vector<int> ins(high);
    ins.insert(ins.end(), standin.begin(), standin.end());
    std::sort(ins.begin(), ins.end());
    StageRef staged;
    int rc = qsvk_stage(st, rec.data(), sizeof(SeqGate) * rec.size(), off.data(), sizeof(uint64_t) * D, &staged);
    if (rc) return rc;
    const SeqGate *dev_g = reinterpret_cast<const SeqGate *>(staged.dev);
    const uint64_t *dev_off = reinterpret_cast<const uint64_t *>(staged.dev + qsv_pad16(sizeof(SeqGate) * rec.size()));
    const bool nt = st->nontemporal != 0;
    const uint64_t per_launch = 0x00ffffffull * QSV_BLOCK;
    if (KL == 0) {
        BigArgs g;
        std::memset(&g, 0, sizeof(g));
        g.W = W;
        g.nins = static_cast<int>(ins.size());
        for (size_t j = 0; j < ins.size(); ++j) g.pos[j] = static_cast<uint32_t>(ins[j]);
        g.regions = st->remap >= 0 ? static_cast<uint32_t>(st->remap) : 8;
        snprintf(st->last_kernel, sizeof(st->last_kernel), "k_seq_big<%s>", nt ? "true" : "false");
        for (g.w0 = 0; g.w0 < g.W; g.w0 += per_launch) {
            const dim3 gd(grid_for(std::min(per_launch, g.W - g.w0), QSV_BLOCK, 0)), bd(QSV_BLOCK);
            if (nt) hipLaunchKernelGGL(k_seq_big<true>, gd, bd, 0, st->stream, st->data, g, dev_g, n_gates, dev_off);
            else hipLaunchKernelGGL(k_seq_big<false>, gd, bd, 0, st->stream, st->data, g, dev_g, n_gates, dev_off);
            rc = check_launch();
            if (rc) return rc;
        }
        return qsvk_stage_done(st, staged);
    }
    LdsArgs g;
    std::memset(&g, 0, sizeof(g));
    g.W = W;
    g.nins = static_cast<int>(ins.size());
    for (size_t j = 0; j < ins.size(); ++j) g.pos[j] = static_cast<uint32_t>(ins[j]);
    for (int j = 0; j < KL; ++j) {
        if (low[j] >= 3) {
            g.abit[g.na] = low[j];
            g.aE[g.na] = standin[j];
            g.amask |= 1u << low[j];
            ++g.na;
        } else {
            g.bmask |= 1u << low[j];
        }
    }
    for (int v = 0; v < (1 << KB); ++v)
        for (int j = 0; j < KB; ++j)
            if ((v >> j) & 1) g.bdep[v] |= 1u << low[j];
    g.regions = st->remap >= 0 ? static_cast<uint32_t>(st->remap) : 8;
    snprintf(st->last_kernel, sizeof(st->last_kernel), "k_seq_lds<%d, %s>", KB, nt ? "true" : "false");
    for (g.w0 = 0; g.w0 < g.W; g.w0 += per_launch) {
        const dim3 gd(grid_for(std::min(per_launch, g.W - g.w0), QSV_BLOCK, 0)), bd(QSV_BLOCK);
#define QSV_LAUNCH_SEQ(KBV)                                                                                           \
    do {                                                                                                              \
        if (nt) hipLaunchKernelGGL((k_seq_lds<KBV, true>), gd, bd, 0, st->stream, st->data, g, dev_g, n_gates, dev_off); \
        else hipLaunchKernelGGL((k_seq_lds<KBV, false>), gd, bd, 0, st->stream, st->data, g, dev_g, n_gates, dev_off);   \
    } while (0)
        switch (KB) {
            case 0: QSV_LAUNCH_SEQ(0); break;
            case 1: QSV_LAUNCH_SEQ(1); break;
            case 2: QSV_LAUNCH_SEQ(2); break;
            default: QSV_LAUNCH_SEQ(3); break;
        }
#undef QSV_LAUNCH_SEQ
        rc = check_launch();
        if (rc) return rc;
    }
    return qsvk_stage_done(st, staged);
}

// A fused block of k = 1..6 qubits as the list of its source gates on LDS tiles (k_seq_tile).  Arguments as
// qsvk_sequence5.  QSV_UNHANDLED_KQ: the register is too small for a tile, a gate acts on more than two qubits, or the
// list needs more passes than the kernel's table holds.
int qsvk_sequence_tile(qsv_state *st, int k, const int *bits, int n_gates, const int *arity, const int *legs,
                       const double *mats) {
    if (k < 1 || k > 6 || n_gates < 1 || n_gates > SEQ_MAX_GATES || st->n < TILE_SEQ_BITS) return QSV_UNHANDLED_KQ;
    // the tile's bits: the targets and the lowest other bits, 12 in all, ascending
    std::vector<int> tile_bits(bits, bits + k);
    for (int b = 0; b < st->n && static_cast<int>(tile_bits.size()) < TILE_SEQ_BITS; ++b)
        if (std::find(bits, bits + k, b) == bits + k) tile_bits.push_back(b);
    std::sort(tile_bits.begin(), tile_bits.end());
    auto position = [&](int bit) { return static_cast<int>(std::find(tile_bits.begin(), tile_bits.end(), bit) - tile_bits.begin()); };
    // passes: consecutive gates whose legs fit four tile bits together
    struct Draft {
        std::vector<int> q;      // tile bits of the pass
        std::vector<int> gate;   // indices into the gate list
    };
    std::vector<Draft> drafts;
    std::vector<std::array<int, 2>> where(n_gates);
    for (int gi = 0; gi < n_gates; ++gi) {
        if (arity[gi] != 1 && arity[gi] != 2) return QSV_UNHANDLED_KQ;
        for (int j = 0; j < 2; ++j) {
            const int leg = legs[2 * gi + (j < arity[gi] ? j : 0)];
            if (leg < 0 || leg >= k) return qsv_fail(QSV_EINVAL, "gate sequence: leg outside the block");
            where[gi][j] = position(bits[leg]);
        }
        if (arity[gi] == 2 && where[gi][0] == where[gi][1]) return qsv_fail(QSV_EINVAL, "gate sequence: legs outside the block");
        std::vector<int> merged = drafts.empty() ? std::vector<int>() : drafts.back().q;
        for (int j = 0; j < arity[gi]; ++j)
            if (std::find(merged.begin(), merged.end(), where[gi][j]) == merged.end()) merged.push_back(where[gi][j]);
        if (drafts.empty() || merged.size() > 4) {
            drafts.push_back(Draft{});
            merged.assign(where[gi].begin(), where[gi].begin() + arity[gi]);
        }
        drafts.back().q = merged;
        drafts.back().gate.push_back(gi);
    }
    if (drafts.size() > static_cast<size_t>(TILE_SEQ_MAX_PASSES)) return QSV_UNHANDLED_KQ;
    std::vector<TilePass> passes(drafts.size());
    std::vector<SeqGate> rec(n_gates);
    std::vector<size_t> mat_at(n_gates);
    {
        size_t at = 0;
        for (int gi = 0; gi < n_gates; ++gi) {
            mat_at[gi] = at;
            at += arity[gi] == 1 ? 8 : 32;
        }
    }
    int next = 0;
    for (size_t p = 0; p < drafts.size(); ++p) {
        Draft &d = drafts[p];
        for (int b = 0; d.q.size() < 4; ++b)          // fewer than four bits in use: any other tile bits complete the group
            if (std::find(d.q.begin(), d.q.end(), b) == d.q.end()) d.q.push_back(b);
        std::sort(d.q.begin(), d.q.end());
        TilePass &ps = passes[p];
        std::memset(&ps, 0, sizeof(ps));
        ps.first = next;
        ps.count = static_cast<int32_t>(d.gate.size());
        for (int j = 0; j < 4; ++j) ps.q[j] = d.q[j];
        for (int gi : d.gate) {
            SeqGate &r = rec[next++];
            std::memset(&r, 0, sizeof(r));
            const double *m = mats + mat_at[gi];
            auto local = [&](int tile_bit) { return static_cast<int>(std::find(d.q.begin(), d.q.end(), tile_bit) - d.q.begin()); };
            if (arity[gi] == 1) {
                r.code = local(where[gi][0]);
                std::memcpy(r.m, m, sizeof(double) * 8);
            } else {
                const int j0 = local(where[gi][0]), j1 = local(where[gi][1]);
                const int hi = std::max(j0, j1), lo = std::min(j0, j1);
                r.code = 5 + hi * (hi - 1) / 2 + lo;
                for (int rr = 0; rr < 4; ++rr)
                    for (int cc = 0; cc < 4; ++cc) {
                        // record index bit 1 <-> register bit hi; the caller's index bit 1 <-> leg 0
                        const int ur = j0 > j1 ? rr : ((rr & 1) << 1) | (rr >> 1), uc = j0 > j1 ? cc : ((cc & 1) << 1) | (cc >> 1);
                        r.m[2 * (rr * 4 + cc)] = m[2 * (ur * 4 + uc)];
                        r.m[2 * (rr * 4 + cc) + 1] = m[2 * (ur * 4 + uc) + 1];
                    }
            }
        }
    }
    std::vector<uint64_t> off(TILE_SEQ_ROWS, 0);
    for (int row = 0; row < TILE_SEQ_ROWS; ++row)
        for (int j = 0; j < TILE_SEQ_BITS - 6; ++j)
            if ((row >> j) & 1) off[row] |= 1ull << tile_bits[6 + j];
    // one image: [gates | passes | row offsets]
    const size_t gates_bytes = qsv_pad16(sizeof(SeqGate) * rec.size()), passes_bytes = qsv_pad16(sizeof(TilePass) * passes.size());
    std::vector<char> image(gates_bytes + passes_bytes, 0);
    std::memcpy(image.data(), rec.data(), sizeof(SeqGate) * rec.size());
    std::memcpy(image.data() + gates_bytes, passes.data(), sizeof(TilePass) * passes.size());
    StageRef staged;
    int rc = qsvk_stage(st, image.data(), image.size(), off.data(), sizeof(uint64_t) * off.size(), &staged);
    if (rc) return rc;
    const SeqGate *dev_g = reinterpret_cast<const SeqGate *>(staged.dev);
    const TilePass *dev_p = reinterpret_cast<const TilePass *>(staged.dev + gates_bytes);
    const uint64_t *dev_off = reinterpret_cast<const uint64_t *>(staged.dev + qsv_pad16(image.size()));
    TileSeqArgs ta;
    std::memset(&ta, 0, sizeof(ta));
    BigArgs &g = ta.g;
    g.W = st->amps >> TILE_SEQ_BITS;      // tiles: the index with every tile bit taken out
    g.nins = TILE_SEQ_BITS;
    for (int j = 0; j < TILE_SEQ_BITS; ++j) g.pos[j] = static_cast<uint32_t>(tile_bits[j]);
    for (int j = 0; j < 6; ++j) ta.lane_bit[j] = static_cast<uint32_t>(tile_bits[j]);
    g.regions = st->remap >= 0 ? static_cast<uint32_t>(st->remap) : 8;
    const bool nt = st->nontemporal != 0;
    const size_t lds = sizeof(amp_t) << TILE_SEQ_BITS;
    static bool raised = false;
    if (!raised) {
        QSV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_seq_tile<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    static_cast<int>(lds)));
        QSV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_seq_tile<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    static_cast<int>(lds)));
        raised = true;
    }
    snprintf(st->last_kernel, sizeof(st->last_kernel), "k_seq_tile<%s>", nt ? "true" : "false");
    st->last_passes = static_cast<int>(passes.size());
    const uint64_t per_launch = 1ull << 23;       // tiles per dispatch (a power of two: the tile order stays whole)
    for (g.w0 = 0; g.w0 < g.W; g.w0 += per_launch) {
        const uint64_t tiles = std::min(per_launch, g.W - g.w0);
        const dim3 gd(static_cast<unsigned>(tiles)), bd(TILE_SEQ_THREADS);
        if (nt) hipLaunchKernelGGL(k_seq_tile<true>, gd, bd, lds, st->stream, st->data, ta, dev_g, dev_p, static_cast<int>(passes.size()), dev_off);
        else hipLaunchKernelGGL(k_seq_tile<false>, gd, bd, lds, st->stream, st->data, ta, dev_g, dev_p, static_cast<int>(passes.size()), dev_off);
        rc = check_launch();
        if (rc) return rc;
    }
    return qsvk_stage_done(st, staged);
}

// k = 5 (complex matrices) and k = 6 on the matrix cores (k_dense_mfma).  bits[j] = bit position of matrix leg j (leg 0
// most significant).
static int launch_dense_mfma(qsv_state *st, int k, const int *bits, const double *m_user) {
    const int D = 1 << k;
    // A wave's 16 lanes li are the 16 lowest free index values and its 4 lanes lk the two lowest target bits, so a
    // wave-instruction covers whole 128-byte lines wherever the targets sit, unless bits 0, 1 AND 2 are all targets
    // (then it covers half lines, and the other half follows in the next instruction of the same wave): 2.9-3.1 ms at
    // every placement.  (Round 2 first moved low targets away with a qubit permutation before and after: 6.2 ms.)
    const std::vector<int> tb(bits, bits + k);
    std::vector<int> sorted(tb);
    std::sort(sorted.begin(), sorted.end());
    // register / matrix index c: bit i <-> sorted[i]
    std::vector<uint64_t> off(D, 0);
    for (int c = 0; c < D; ++c)
        for (int i = 0; i < k; ++i)
            if ((c >> i) & 1) off[c] |= 1ull << sorted[i];
    auto user_index = [&](int c) {
        int u = 0;
        for (int leg = 0; leg < k; ++leg)
            for (int i = 0; i < k; ++i)
                if (sorted[i] == tb[leg]) u |= ((c >> i) & 1) << (k - 1 - leg);
        return u;
    };
    bool real_matrix = true;
    for (int i = 0; i < D * D && real_matrix; ++i) real_matrix = m_user[2 * i + 1] == 0.0;
    std::vector<double> m(real_matrix ? D * D : 2 * D * D);  // [plane][col][row]
    std::vector<int> ui(D);
    for (int c = 0; c < D; ++c) ui[c] = user_index(c);
    for (int r = 0; r < D; ++r)
        for (int c = 0; c < D; ++c) {
            const int ur = ui[r], uc = ui[c];
            m[c * D + r] = m_user[2 * (ur * D + uc)];
            if (!real_matrix) m[D * D + c * D + r] = m_user[2 * (ur * D + uc) + 1];
        }
    StageRef staged;
    int rc = qsvk_stage(st, m.data(), sizeof(double) * m.size(), off.data(), sizeof(uint64_t) * D, &staged);
    if (rc) return rc;
    const double *dev_m = reinterpret_cast<const double *>(staged.dev);
    const uint64_t *dev_off = reinterpret_cast<const uint64_t *>(staged.dev + qsv_pad16(sizeof(double) * m.size()));
    Mfma6Args g;
    std::memset(&g, 0, sizeof(g));
    g.W = st->amps >> k;
    g.nins = k;
    for (int i = 0; i < k; ++i) g.pos[i] = static_cast<uint32_t>(sorted[i]);
    const bool nt = st->nontemporal != 0;
    const size_t lds = sizeof(double) * (real_matrix ? 1 : 2) * D * D + sizeof(uint64_t) * D;
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, st->device);  // 256 if the query fails
    const uint64_t wave_tiles = g.W / 16;
    const unsigned grid = static_cast<unsigned>(std::min<uint64_t>((wave_tiles + 3) / 4, ((k == 6 && !real_matrix) ? 2ull : 3ull) * cus));
    const bool m3 = !real_matrix && st->complex_product != 4;   // three real MFMAs per complex entry
    snprintf(st->last_kernel, sizeof(st->last_kernel), "k_dense_mfma<%d, %s, %s%s>", k, nt ? "true" : "false",
             real_matrix ? "true" : "false", m3 ? ", true" : "");
#define QSV_LAUNCH_MFMA(KK, N, ...)                                                                                \
    do {                                                                                                           \
        QSV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_dense_mfma<KK, N, __VA_ARGS__>),              \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));           \
        hipLaunchKernelGGL((k_dense_mfma<KK, N, __VA_ARGS__>), dim3(grid), dim3(QSV_BLOCK), lds, st->stream,       \
                           st->data, g, dev_m, dev_off);                                                           \
    } while (0)
    if (k == 6 && m3) {
        if (nt) QSV_LAUNCH_MFMA(6, true, false, true);
        else QSV_LAUNCH_MFMA(6, false, false, true);
    } else if (k == 6) {
        if (nt) { if (real_matrix) QSV_LAUNCH_MFMA(6, true, true); else QSV_LAUNCH_MFMA(6, true, false); }
        else { if (real_matrix) QSV_LAUNCH_MFMA(6, false, true); else QSV_LAUNCH_MFMA(6, false, false); }
    } else if (m3) {
        if (nt) QSV_LAUNCH_MFMA(5, true, false, true);
        else QSV_LAUNCH_MFMA(5, false, false, true);
    } else {
        if (nt) { if (real_matrix) QSV_LAUNCH_MFMA(5, true, true); else QSV_LAUNCH_MFMA(5, true, false); }
        else { if (real_matrix) QSV_LAUNCH_MFMA(5, false, true); else QSV_LAUNCH_MFMA(5, false, false); }
    }
#undef QSV_LAUNCH_MFMA
    rc = check_launch();
    if (rc) return rc;
    return qsvk_stage_done(st, staged);
}

int qsvk_generic(qsv_state *st, int k, const int *bits, const double *m_user) {
    if (k < 1 || k > QSV_MAX_K) return qsv_fail(QSV_EINVAL, "generic gate: k must be in 1..6");
    // matrix-core form: the 2^(n-k) groups must fill whole waves of 16
    const bool mfma_ok = st->n >= k && (st->amps >> k) >= 16;
    if (k == 6 && mfma_ok && st->kq_variant == 0) return launch_dense_mfma(st, 6, bits, m_user);
    if (k == 5 && mfma_ok && st->kq_variant == 5) return launch_dense_mfma(st, 5, bits, m_user);
    // (k = 5 on the matrix cores is a measurement variant only: on the benchmark circuit's fused blocks it wins where its
    // wave-instructions cover >= 512 contiguous bytes and the other targets are low (1.50 against 1.66 ms), loses with
    // targets above bit 18 (1.8-1.9 against 1.65), and over the whole circuit ties with the vector kernels: 28.3 ms both)
    if (k >= 3 && k <= 6 && st->n >= k) {
        const int rc_big = launch_dense_big(st, k, bits, m_user);
        if (rc_big != QSV_UNHANDLED_KQ) return rc_big;
    }
    const size_t bytes = sizeof(double) * 2ull << (2 * k);
    StageRef staged;
    int rc = qsvk_stage(st, m_user, bytes, nullptr, 0, &staged);
    if (rc) return rc;
    const double *dev_m = reinterpret_cast<const double *>(staged.dev);
    GenericArgs g;
    std::memset(&g, 0, sizeof(g));
    g.K = k;
    g.W = st->amps >> k;
    std::vector<int> sorted(bits, bits + k);
    std::sort(sorted.begin(), sorted.end());
    for (int j = 0; j < k; ++j) {
        g.sorted_pos[j] = static_cast<uint8_t>(sorted[j]);
        g.leg_pos[j] = static_cast<uint8_t>(bits[j]);
    }
    const int grid = grid_for(g.W, QSV_BLOCK, 4096);
    snprintf(st->last_kernel, sizeof(st->last_kernel), "k_generic<%d>", k);
    switch (k) {
        case 1: hipLaunchKernelGGL((k_generic<1>), dim3(grid), dim3(QSV_BLOCK), 0, st->stream, st->data, g, dev_m); break;
        case 2: hipLaunchKernelGGL((k_generic<2>), dim3(grid), dim3(QSV_BLOCK), 0, st->stream, st->data, g, dev_m); break;
        case 3: hipLaunchKernelGGL((k_generic<3>), dim3(grid), dim3(QSV_BLOCK), 0, st->stream, st->data, g, dev_m); break;
        case 4: hipLaunchKernelGGL((k_generic<4>), dim3(grid), dim3(QSV_BLOCK), 0, st->stream, st->data, g, dev_m); break;
        case 5: hipLaunchKernelGGL((k_generic<5>), dim3(grid), dim3(QSV_BLOCK), 0, st->stream, st->data, g, dev_m); break;
        default: hipLaunchKernelGGL((k_generic<6>), dim3(grid), dim3(QSV_BLOCK), 0, st->stream, st->data, g, dev_m); break;
    }
    rc = check_launch();
    if (rc) return rc;
    return qsvk_stage_done(st, staged);
}

// bits[j] = bit position of matrix leg j (leg 0 most significant); k in {1, 2}.
int qsvk_dense(qsv_state *st, int k, const int *bits, int nctrl, const int *cbits, const double *m_user) {
    if (k < 1 || k > 2) return qsv_fail(QSV_EINVAL, "dense kernel handles 1- and 2-qubit matrices");
    if (st->n < QSV_LANE_BITS) {
        // tiny register: fold the controls into a (k + nctrl)-qubit matrix for the gather kernel
        if (nctrl == 0) return qsvk_generic(st, k, bits, m_user);
        std::vector<int> legs(cbits, cbits + nctrl);
        legs.insert(legs.end(), bits, bits + k);
        const std::vector<double> full = expand_controls(k, nctrl, m_user);
        return qsvk_generic(st, k + nctrl, legs.data(), full.data());
    }
    // dense, every target on bit 3 (1 qubit) / 6 (2 qubits) or higher, controls on bit 3 or higher: the workgroup-tile
    // form (k_dense_tile12 / k_dense_tile12_ctrl); QSV_OPT_KQ_VARIANT 1 or 2, or an explicit QSV_OPT_UNROLL, keep k_dense
    if (st->unroll == 0 && st->kq_variant != 1 && st->kq_variant != 2) {
        const int rc_tile = launch_tile12(st, k, bits, nctrl, cbits, m_user);
        if (rc_tile != QSV_UNHANDLED_KQ) return rc_tile;
    }
    GateArgs g;
    std::memset(&g, 0, sizeof(g));
    std::vector<int> high, low;
    for (int j = 0; j < k; ++j) (bits[j] >= QSV_LANE_BITS ? high : low).push_back(bits[j]);
    const int KH = static_cast<int>(high.size()), KL = static_cast<int>(low.size());
    int rc = fill_enumeration(st, g, high, nctrl, cbits);
    if (rc) return rc;
    for (int h = 0; h < (1 << KH); ++h) {
        uint64_t o = 0;
        for (int i = 0; i < KH; ++i)
            if ((h >> i) & 1) o |= 1ull << high[i];
        g.hoff[h] = o;
    }
    for (int j = 0; j < KL; ++j) g.lbit[j] = low[j];
    for (int x = 0; x < (1 << KL); ++x) {
        int mask = 0;
        for (int j = 0; j < KL; ++j)
            if ((x >> j) & 1) mask |= 1 << low[j];
        g.lxor[x] = mask;
    }
    // kernel order: index = (h << KL) | l, h bit i <-> high[i], l bit i <-> low[i]
    const int D = 1 << k;
    auto user_index = [&](int kidx) {
        const int h = kidx >> KL, l = kidx & ((1 << KL) - 1);
        int u = 0;
        for (int j = 0; j < k; ++j) {
            int bitval = 0;
            for (int i = 0; i < KH; ++i)
                if (high[i] == bits[j]) bitval = (h >> i) & 1;
            for (int i = 0; i < KL; ++i)
                if (low[i] == bits[j]) bitval = (l >> i) & 1;
            u |= bitval << (k - 1 - j);
        }
        return u;
    };
    for (int r = 0; r < D; ++r)
        for (int c = 0; c < D; ++c) {
            const int ur = user_index(r), uc = user_index(c);
            g.m[2 * (r * D + c)] = m_user[2 * (ur * D + uc)];
            g.m[2 * (r * D + c) + 1] = m_user[2 * (ur * D + uc) + 1];
        }
    return dispatch_dense(st, KH, KL, g);
}

// SWAP of two bits >= QSV_LANE_BITS: exchange a[base | Sa] <-> a[base | Sb]; the 00 and 11 quarters stay put.
int qsvk_pair_exchange(qsv_state *st, int bit_a, int bit_b) {
    const uint64_t quarter = st->amps >> 2;
    if (st->unroll == 0 && st->kq_variant != 1 && st->kq_variant != 2 && bit_a >= 3 && bit_b >= 3 && quarter >= 64 &&
        quarter % 64 == 0) {
        // tile form: an X "gate" between the amplitudes with (a, b) = (1, 0) and (0, 1); both bits leave the enumeration
        SmallGate sg;
        std::memset(&sg, 0, sizeof(sg));
        sg.m[2] = 1.0;   // m[0][1]
        sg.m[4] = 1.0;   // m[1][0]
        sg.off[0] = 1ull << bit_a;
        sg.off[1] = 1ull << bit_b;
        BigArgs t;
        std::memset(&t, 0, sizeof(t));
        t.W = quarter;
        t.nins = 2;
        t.pos[0] = static_cast<uint32_t>(std::min(bit_a, bit_b));
        t.pos[1] = static_cast<uint32_t>(std::max(bit_a, bit_b));
        t.regions = st->remap >= 0 ? static_cast<uint32_t>(st->remap) : 8;
        return launch_tile12_kernels(st, 1, true, t, sg);
    }
    GateArgs g;
    std::memset(&g, 0, sizeof(g));
    std::vector<int> removed = {bit_a, bit_b};
    int rc = fill_enumeration(st, g, removed, 0, nullptr);
    if (rc) return rc;
    g.hoff[0] = 1ull << bit_a;
    g.hoff[1] = 1ull << bit_b;
    const double x[8] = {0, 0, 1, 0, 1, 0, 0, 0};
    std::memcpy(g.m, x, sizeof(x));
    return dispatch_dense(st, 1, 0, g);
}

// bits[j] = position of diagonal leg j (leg 0 most significant); d has 2^k complex entries.
int qsvk_diag(qsv_state *st, int k, const int *bits, int nctrl, const int *cbits, const double *d_user) {
    if (k < 1 || k > QSV_MAX_K) return qsv_fail(QSV_EINVAL, "diagonal gate: k must be in 1..6");
    if (k <= 2 && st->n >= QSV_LANE_BITS) {
        DiagArgs g;
        std::memset(&g, 0, sizeof(g));
        int rc = fill_enumeration(st, g, {}, nctrl, cbits);
        if (rc) return rc;
        g.b0 = bits[0];
        g.b1 = k == 2 ? bits[1] : -1;
        std::memcpy(g.d, d_user, sizeof(double) * (2 << k));
        return launch_diag(st, g);
    }
    // table path (k > 2 or tiny register); controls are folded into the table
    std::vector<int> legs(cbits, cbits + nctrl);
    legs.insert(legs.end(), bits, bits + k);
    const int K = k + nctrl;
    if (K > QSV_MAX_K) return qsv_fail(QSV_EINVAL, "diagonal gate: too many legs for the table kernel");
    std::vector<double> table(2ull << K, 0.0);
    for (int i = 0; i < (1 << K); ++i) {
        table[2 * i] = 1.0;
        if ((i >> k) == (1 << nctrl) - 1) {
            table[2 * i] = d_user[2 * (i & ((1 << k) - 1))];
            table[2 * i + 1] = d_user[2 * (i & ((1 << k) - 1)) + 1];
        }
    }
    const size_t tbytes = sizeof(double) * table.size();
    uint8_t pos[8] = {0};
    for (int j = 0; j < K; ++j) pos[j] = static_cast<uint8_t>(legs[j]);
    StageRef staged;
    int rc = qsvk_stage(st, table.data(), tbytes, pos, 8, &staged);
    if (rc) return rc;
    const double *dtable = reinterpret_cast<const double *>(staged.dev);
    const uint8_t *dpos = reinterpret_cast<const uint8_t *>(staged.dev + qsv_pad16(tbytes));
    if (st->n >= RO_MIN_QUBITS && st->readout_variant != 1) {
        snprintf(st->last_kernel, sizeof(st->last_kernel), "k_diag_table_s");
        QSV_RO_DISPATCH(ro_fit_items(getenv("QSV_RO_ITEMS") ? ro_move_items() : 2, st->amps), hipLaunchKernelGGL((k_diag_table_s<IT>), dim3(static_cast<unsigned>(st->amps / (QSV_BLOCK * IT))), dim3(QSV_BLOCK), 0,
                           st->stream, st->data, st->amps, K, dpos, dtable));
    } else {
        const int grid = grid_for(st->amps, QSV_BLOCK, 4096);
        snprintf(st->last_kernel, sizeof(st->last_kernel), "k_diag_table");
        hipLaunchKernelGGL(k_diag_table, dim3(grid), dim3(QSV_BLOCK), 0, st->stream, st->data, st->amps, K, dpos, dtable);
    }
    rc = check_launch();
    if (rc) return rc;
    return qsvk_stage_done(st, staged);
}

int qsvk_phase(qsv_state *st, int nctrl, const int *cbits, double re, double im) {
    if (st->n < QSV_LANE_BITS || nctrl == 0) {
        if (nctrl == 0) return qsvk_scale(st, re, im);
        if (nctrl > QSV_MAX_K) return qsv_fail(QSV_EINVAL, "phase on a tiny register: too many qubits");
        // all qubits are "controls": table with the phase at the all-ones entry
        std::vector<double> d = {1.0, 0.0, re, im};
        return qsvk_diag(st, 1, cbits + nctrl - 1, nctrl - 1, cbits, d.data());
    }
    DiagArgs g;
    std::memset(&g, 0, sizeof(g));
    int rc = fill_enumeration(st, g, {}, nctrl, cbits);
    if (rc) return rc;
    // every enumerated amplitude already has all controls = 1: multiply by the phase whatever bit b0 is
    g.b0 = 0;
    g.b1 = -1;
    g.d[0] = g.d[2] = re;
    g.d[1] = g.d[3] = im;
    return launch_diag(st, g);
}

int qsvk_measure_probs(qsv_state *st, int bit, const double e0[4], const double e1[4], double *p0, double *p1) {
    if (st->n >= RO_MIN_QUBITS && st->readout_variant != 1) {
        const int grid = grid_for(st->amps >> (bit < QSV_LANE_BITS ? 0 : 1), QSV_BLOCK * RO_ITEMS, QSV_REDUCE_BLOCKS);
        const cplx a{e0[0], e0[1]}, b{e0[2], e0[3]}, c{e1[0], e1[1]}, d{e1[2], e1[3]};
        snprintf(st->last_kernel, sizeof(st->last_kernel), "k_measure_probs_s<%s>", bit < QSV_LANE_BITS ? "true" : "false");
        if (bit < QSV_LANE_BITS)
            hipLaunchKernelGGL(k_measure_probs_s<true>, dim3(grid), dim3(QSV_BLOCK), 0, st->stream, st->data, st->amps, bit,
                               a, b, c, d, st->partials);
        else
            hipLaunchKernelGGL(k_measure_probs_s<false>, dim3(grid), dim3(QSV_BLOCK), 0, st->stream, st->data, st->amps, bit,
                               a, b, c, d, st->partials);
        int rc = check_launch();
        if (rc) return rc;
        return sum_partials(st, grid, p0, p1);
    }
    snprintf(st->last_kernel, sizeof(st->last_kernel), "k_measure_probs");
    const uint64_t pairs = st->amps >> 1;
    const int grid = grid_for(pairs, QSV_BLOCK * 8, QSV_REDUCE_BLOCKS);
    hipLaunchKernelGGL(k_measure_probs, dim3(grid), dim3(QSV_BLOCK), 0, st->stream, st->data, pairs, bit,
                       cplx{e0[0], e0[1]}, cplx{e0[2], e0[3]}, cplx{e1[0], e1[1]}, cplx{e1[2], e1[3]},
                       st->partials);
    int rc = check_launch();
    if (rc) return rc;
    return sum_partials(st, grid, p0, p1);
}

static int adopt(qsv_state *st, amp_t *fresh, uint64_t new_amps) {
    (void)fresh;  // == st->spare
    return qsvk_adopt(st, new_amps);
}

int qsvk_collapse(qsv_state *st, int bit, const double e[4], double scale) {
    const uint64_t pairs = st->amps >> 1;
    amp_t *fresh = nullptr;
    int rc = qsvk_scratch(st, pairs, &fresh);
    if (rc) return rc;
    if (st->n >= RO_MIN_QUBITS && st->readout_variant != 1) {
        const int items = ro_fit_items(ro_move_items(), pairs);
        const dim3 gd(static_cast<unsigned>(pairs / (QSV_BLOCK * items))), bd(QSV_BLOCK);
        snprintf(st->last_kernel, sizeof(st->last_kernel), "k_collapse_s<%s>", bit < QSV_LANE_BITS ? "true" : "false");
        if (bit < QSV_LANE_BITS)
            QSV_RO_DISPATCH(items, hipLaunchKernelGGL((k_collapse_s<true, IT>), gd, bd, 0, st->stream, st->data, fresh, pairs, bit,
                                                      cplx{e[0], e[1]}, cplx{e[2], e[3]}, scale));
        else
            QSV_RO_DISPATCH(items, hipLaunchKernelGGL((k_collapse_s<false, IT>), gd, bd, 0, st->stream, st->data, fresh, pairs, bit,
                                                      cplx{e[0], e[1]}, cplx{e[2], e[3]}, scale));
    } else {
        const int grid = grid_for(pairs, QSV_BLOCK * 4, 8192);
        snprintf(st->last_kernel, sizeof(st->last_kernel), "k_collapse");
        hipLaunchKernelGGL(k_collapse, dim3(grid), dim3(QSV_BLOCK), 0, st->stream, st->data, fresh, pairs, bit,
                           cplx{e[0], e[1]}, cplx{e[2], e[3]}, scale);
    }
    rc = check_launch();
    if (rc) return rc;
    st->n -= 1;
    return adopt(st, fresh, pairs);
}

int qsvk_insert(qsv_state *st, int bit, const double amp[4]) {
    const uint64_t out_amps = st->amps << 1;
    if (!st->owns_data && out_amps > st->capacity)
        return qsv_fail(QSV_ENOMEM, "insert: the caller-owned buffer has no room for one more qubit");
    amp_t *fresh = nullptr;
    int rc = qsvk_scratch(st, out_amps, &fresh);
    if (rc) return rc;
    if (st->n >= RO_MIN_QUBITS && st->readout_variant != 1) {
        const int items = ro_fit_items(ro_move_items(), st->amps);
        const dim3 gd(static_cast<unsigned>(st->amps / (QSV_BLOCK * items))), bd(QSV_BLOCK);
        snprintf(st->last_kernel, sizeof(st->last_kernel), "k_insert_s<%s>", bit < QSV_LANE_BITS ? "true" : "false");
        if (bit < QSV_LANE_BITS)
            QSV_RO_DISPATCH(items, hipLaunchKernelGGL((k_insert_s<true, IT>), gd, bd, 0, st->stream, st->data, fresh, st->amps, bit,
                                                      cplx{amp[0], amp[1]}, cplx{amp[2], amp[3]}));
        else
            QSV_RO_DISPATCH(items, hipLaunchKernelGGL((k_insert_s<false, IT>), gd, bd, 0, st->stream, st->data, fresh, st->amps, bit,
                                                      cplx{amp[0], amp[1]}, cplx{amp[2], amp[3]}));
    } else {
        const int grid = grid_for(out_amps, QSV_BLOCK * 4, 8192);
        snprintf(st->last_kernel, sizeof(st->last_kernel), "k_insert");
        hipLaunchKernelGGL(k_insert, dim3(grid), dim3(QSV_BLOCK), 0, st->stream, st->data, fresh, out_amps, bit,
                           cplx{amp[0], amp[1]}, cplx{amp[2], amp[3]});
    }
    rc = check_launch();
    if (rc) return rc;
    st->n += 1;
    return adopt(st, fresh, out_amps);
}

int qsvk_permute(qsv_state *st, const int *src_bit_of_dst_bit) {
    PermArgs g;
    std::memset(&g, 0, sizeof(g));
    g.n = st->n;
    for (int b = 0; b < st->n; ++b) g.src_bit[b] = static_cast<uint8_t>(src_bit_of_dst_bit[b]);
    amp_t *fresh = nullptr;
    int rc = qsvk_scratch(st, st->amps, &fresh);
    if (rc) return rc;
    if (st->n >= RO_MIN_QUBITS && st->readout_variant != 1) {
        // the tile: destination bits 0..2, the destinations of source bits 0..2, then the lowest other bits up to six
        PermTileArgs t;
        std::memset(&t, 0, sizeof(t));
        t.tiles = st->amps >> 6;
        t.bytes = (st->n + 7) / 8;
        std::vector<int> tile = {0, 1, 2};
        for (int j = 3; j < st->n; ++j)
            if (src_bit_of_dst_bit[j] < 3) tile.push_back(j);
        for (int j = 3; j < st->n && tile.size() < 6; ++j)
            if (std::find(tile.begin(), tile.end(), j) == tile.end()) tile.push_back(j);
        std::sort(tile.begin(), tile.end());
        for (int k = 0; k < 6; ++k) {
            t.tile_dst[k] = static_cast<uint8_t>(tile[k]);
            t.tile_src[k] = static_cast<uint8_t>(src_bit_of_dst_bit[tile[k]]);
        }
        // lut[b][v]: where the destination-index bits 8b .. 8b+7 (value v) come from in the source index
        std::vector<uint64_t> lut(static_cast<size_t>(t.bytes) * 256, 0);
        for (int b = 0; b < t.bytes; ++b)
            for (int v = 0; v < 256; ++v)
                for (int i = 0; i < 8 && 8 * b + i < st->n; ++i)
                    if ((v >> i) & 1) lut[b * 256 + v] |= 1ull << src_bit_of_dst_bit[8 * b + i];
        rc = qsvk_ensure_matrix(st, sizeof(uint64_t) * lut.size());
        if (rc) return rc;
        QSV_HIP(hipMemcpyAsync(st->dev_matrix, lut.data(), sizeof(uint64_t) * lut.size(), hipMemcpyHostToDevice, st->stream));
        QSV_HIP(hipStreamSynchronize(st->stream));  // `lut` dies at return
        const int items = ro_move_items();
        const int grid = grid_for(t.tiles, (QSV_BLOCK / 64) * items, 1 << 22);
        snprintf(st->last_kernel, sizeof(st->last_kernel), "k_permute_s");
        QSV_RO_DISPATCH(items, hipLaunchKernelGGL((k_permute_s<IT>), dim3(grid), dim3(QSV_BLOCK), 0, st->stream, st->data, fresh, t,
                                                  reinterpret_cast<const uint64_t *>(st->dev_matrix)));
    } else {
        const int grid = grid_for(st->amps, QSV_BLOCK * 4, 8192);
        snprintf(st->last_kernel, sizeof(st->last_kernel), "k_permute");
        hipLaunchKernelGGL(k_permute, dim3(grid), dim3(QSV_BLOCK), 0, st->stream, st->data, fresh, st->amps, g);
    }
    rc = check_launch();
    if (rc) return rc;
    return adopt(st, fresh, st->amps);
}

int qsvk_norm2(qsv_state *st, double *out) {
    const int grid = grid_for(st->amps, QSV_BLOCK * 8, QSV_REDUCE_BLOCKS);
    hipLaunchKernelGGL(k_norm2, dim3(grid), dim3(QSV_BLOCK), 0, st->stream, st->data, st->amps, st->partials);
    int rc = check_launch();
    if (rc) return rc;
    return sum_partials(st, grid, out, nullptr);
}

int qsvk_inner(qsv_state *a, qsv_state *b, double *re, double *im) {
    QSV_HIP(hipStreamSynchronize(b->stream));
    const int grid = grid_for(a->amps, QSV_BLOCK * 8, QSV_REDUCE_BLOCKS);
    hipLaunchKernelGGL(k_inner, dim3(grid), dim3(QSV_BLOCK), 0, a->stream, a->data, b->data, a->amps, a->partials);
    int rc = check_launch();
    if (rc) return rc;
    return sum_partials(a, grid, re, im);
}


// rho_out: 4^k complex, row-major, row / column index bit (k-1-j) <-> bits[j] (bits[0] = most significant leg).
int qsvk_reduced_density(qsv_state *st, int k, const int *bits, double *rho_out) {
    if (k < 1 || k > QSV_MAX_K || k > st->n) return qsv_fail(QSV_EINVAL, "reduced density matrix: keep 1..6 qubits");
    const int D = 1 << k;
    std::vector<int> sorted(bits, bits + k);
    std::sort(sorted.begin(), sorted.end());
    const int T = D <= 16 ? 1 : D <= 32 ? 2 : 4, P = T * (T + 1) / 2;
    std::vector<uint64_t> off(16 * T, 0);
    for (int r = 0; r < D; ++r)
        for (int i = 0; i < k; ++i)
            if ((r >> i) & 1) off[r] |= 1ull << sorted[i];
    RdmArgs g;
    std::memset(&g, 0, sizeof(g));
    g.W = st->amps >> k;
    g.nins = k;
    g.D = D;
    for (int i = 0; i < k; ++i) g.pos[i] = static_cast<uint32_t>(sorted[i]);
    const int S = D < 16 ? 16 / D : 1;                   // groups sharing a 16-row tile (k_rdm)
    const bool old_form = st->readout_variant == 2;      // round 2's per-lane row loads (measurement variant)
    const bool big = st->n >= RO_MIN_QUBITS && (old_form ? g.W % (4ull * S * 4 * (RDM_LOADS / T)) == 0   // whole iterations
                                                         : g.W % (64ull * S) == 0 && g.W >= 64ull * S);   // whole tiles
    RdmTileArgs gt;
    std::memset(&gt, 0, sizeof(gt));
    if (big && !old_form) {
        gt.k = k;
        for (int i = 0; i < k; ++i) {
            if (sorted[i] < 6) {
                gt.lmask |= 1u << sorted[i];
                ++gt.l;
            } else {
                gt.pos[gt.h++] = static_cast<uint32_t>(sorted[i]);
            }
        }
        gt.nins = gt.h;
        gt.log_s = S == 1 ? 0 : S == 2 ? 1 : S == 4 ? 2 : 3;
        gt.tiles = g.W / (64ull * S);
        for (int c = 0; c < (1 << gt.h); ++c)
            for (int j = 0; j < gt.h; ++j)
                if ((c >> j) & 1) gt.hoff[c] |= 1ull << gt.pos[j];
        // 8 regions where a kept bit sits on the high address bits (k <= 4 on bits 24..27: 1.14 -> 0.82 ms at n = 28) and
        // for 64-row tiles; plain order otherwise (within 5 % either way: profiles/r03_rdm.txt)
        const uint32_t want = st->remap >= 0 ? static_cast<uint32_t>(st->remap) : ((sorted.back() >= 20 || k == 6) ? 8u : 0u);
        gt.regions = want > 1 && gt.tiles % want == 0 ? want : 0;
    }
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, st->device);  // 256 if the query fails
    // a power-of-two grid, so that (waves in the grid) x RDM_U divides the (power-of-two) number of steps
    int blocks = 0;
    if (big && old_form) {
        const uint64_t most = std::min<uint64_t>((T <= 2 ? 4ull : 2ull) * cus, std::max<uint64_t>(1, g.W / (4ull * S) / (4 * (RDM_LOADS / T))));
        blocks = 1;
        while (2ull * blocks <= most) blocks *= 2;
    } else if (big) {   // persistent workgroups, as many as the LDS tiles (16 T KiB) and the accumulators let a CU hold
        blocks = static_cast<int>(std::min<uint64_t>(gt.tiles, static_cast<uint64_t>(T == 4 ? 2 : T == 2 ? 4 : 8) * cus));
    }
    const int entries = big ? P * 2 * 256 : 2 * D * D;
    const size_t b_off = sizeof(uint64_t) * off.size(), b_out = sizeof(double) * entries,
                 b_part = sizeof(double) * static_cast<size_t>(blocks) * entries;
    int rc = qsvk_ensure_matrix(st, b_off + b_out + b_part + 64);
    if (rc) return rc;
    char *p = reinterpret_cast<char *>(st->dev_matrix);
    uint64_t *d_off = reinterpret_cast<uint64_t *>(p);
    double *d_out = reinterpret_cast<double *>(p + b_off), *d_part = reinterpret_cast<double *>(p + b_off + b_out);
    if (!(big && !old_form)) QSV_HIP(hipMemcpyAsync(d_off, off.data(), b_off, hipMemcpyHostToDevice, st->stream));
    std::vector<double> raw(entries);
    if (big && !old_form) {
        snprintf(st->last_kernel, sizeof(st->last_kernel), "k_rdm_tile<%d>", T);
        if (T == 1) hipLaunchKernelGGL(k_rdm_tile<1>, dim3(blocks), dim3(QSV_BLOCK), 0, st->stream, st->data, gt, d_part);
        else if (T == 2) hipLaunchKernelGGL(k_rdm_tile<2>, dim3(blocks), dim3(QSV_BLOCK), 0, st->stream, st->data, gt, d_part);
        else hipLaunchKernelGGL(k_rdm_tile<4>, dim3(blocks), dim3(QSV_BLOCK), 0, st->stream, st->data, gt, d_part);
        rc = check_launch();
        if (rc) return rc;
        hipLaunchKernelGGL(k_sum_partials, dim3((entries + 15) / 16), dim3(QSV_BLOCK), 0, st->stream, d_part,
                           blocks, entries, d_out);
    } else if (big) {
        snprintf(st->last_kernel, sizeof(st->last_kernel), "k_rdm<%d>", T);
        if (T == 1) hipLaunchKernelGGL(k_rdm<1>, dim3(blocks), dim3(QSV_BLOCK), 0, st->stream, st->data, g, d_off, d_part);
        else if (T == 2) hipLaunchKernelGGL(k_rdm<2>, dim3(blocks), dim3(QSV_BLOCK), 0, st->stream, st->data, g, d_off, d_part);
        else hipLaunchKernelGGL(k_rdm<4>, dim3(blocks), dim3(QSV_BLOCK), 0, st->stream, st->data, g, d_off, d_part);
        rc = check_launch();
        if (rc) {
            (void)hipStreamSynchronize(st->stream);   // the offsets' upload reads a local vector
            return rc;
        }
        hipLaunchKernelGGL(k_sum_partials, dim3((entries + 15) / 16), dim3(QSV_BLOCK), 0, st->stream, d_part,
                           blocks, entries, d_out);
    } else {
        snprintf(st->last_kernel, sizeof(st->last_kernel), "k_rdm_small");
        hipLaunchKernelGGL(k_rdm_small, dim3((D * D + QSV_BLOCK - 1) / QSV_BLOCK), dim3(QSV_BLOCK), 0, st->stream, st->data, g,
                           d_off, d_out);
    }
    rc = check_launch();
    if (rc) {
        (void)hipStreamSynchronize(st->stream);
        return rc;
    }
    QSV_HIP(hipMemcpyAsync(raw.data(), d_out, b_out, hipMemcpyDeviceToHost, st->stream));
    QSV_HIP(hipStreamSynchronize(st->stream));
    // kernel order (row bit i <-> sorted[i]) -> caller order (bit k-1-j <-> bits[j])
    auto user_index = [&](int c) {
        int u = 0;
        for (int leg = 0; leg < k; ++leg)
            for (int i = 0; i < k; ++i)
                if (sorted[i] == bits[leg]) u |= ((c >> i) & 1) << (k - 1 - leg);
        return u;
    };
    for (int r = 0; r < D; ++r)
        for (int c = 0; c < D; ++c) {
            double vr, vi;
            if (big) {
                // tile pair p = (ti <= tj); D layout: col = lane & 15, row = (lane >> 4) + 4 reg.  Entries below the
                // diagonal are the conjugates of those above it and the diagonal is real, exactly (the matrix cores
                // sum the two mirror entries of a diagonal tile in different orders: equal up to rounding only)
                const bool upper = r <= c;
                const int rr = upper ? r : c, cc = upper ? c : r;
                const int ti = rr / 16, tj = cc / 16;
                // k_rdm_tile<4> holds pair (0, 3) as the block of (3, 0): read it mirrored, imaginary part negated
                const bool mirrored = !old_form && T == 4 && ti == 0 && tj == 3;
                int pidx = 0;
                for (int a = 0; a < ti; ++a) pidx += T - a;
                pidx += tj - ti;
                vr = vi = 0.0;
                for (int sgrp = 0; sgrp < S; ++sgrp) {   // 2^k < 16: the diagonal blocks of the shared tile add up
                    int row = rr % 16 + D * sgrp * (D < 16), col = cc % 16 + D * sgrp * (D < 16);
                    if (mirrored) std::swap(row, col);
                    const int reg = row / 4, lane = (row % 4) * 16 + col;
                    vr += raw[((pidx * 2 + 0) * 4 + reg) * 64 + lane];
                    vi += (mirrored ? -1.0 : 1.0) * raw[((pidx * 2 + 1) * 4 + reg) * 64 + lane];
                }
                if (!upper) vi = -vi;  // rho[r][c] = conj(rho[c][r])
                if (r == c) vi = 0.0;
            } else {
                vr = raw[2 * (r * D + c)];
                vi = raw[2 * (r * D + c) + 1];
            }
            const int ur = user_index(r), uc = user_index(c);
            rho_out[2 * (ur * D + uc)] = vr;
            rho_out[2 * (ur * D + uc) + 1] = vi;
        }
    return QSV_OK;
}

int qsvk_expect_density(qsv_state *ket, qsv_state *rho, double *re, double *im) {
    QSV_HIP(hipStreamSynchronize(ket->stream));
    const int grid = grid_for(rho->amps, QSV_BLOCK * 8, QSV_REDUCE_BLOCKS);
    hipLaunchKernelGGL(k_expect_density, dim3(grid), dim3(QSV_BLOCK), 0, rho->stream, ket->data, rho->data,
                       static_cast<uint64_t>(ket->n), rho->partials);
    int rc = check_launch();
    if (rc) return rc;
    return sum_partials(rho, grid, re, im);
}

int qsvk_expect_pauli(qsv_state *st, uint64_t xmask, uint64_t zmask, int n_y, double *re, double *im) {
    const int grid = grid_for(st->amps, QSV_BLOCK * 8, QSV_REDUCE_BLOCKS);
    hipLaunchKernelGGL(k_expect_pauli, dim3(grid), dim3(QSV_BLOCK), 0, st->stream, st->data, st->amps, xmask, zmask,
                       st->partials);
    int rc = check_launch();
    if (rc) return rc;
    double sr = 0.0, si = 0.0;
    rc = sum_partials(st, grid, &sr, &si);
    if (rc) return rc;
    switch (n_y & 3) {  // times i^{nY}
        case 0: *re = sr; *im = si; break;
        case 1: *re = -si; *im = sr; break;
        case 2: *re = -sr; *im = -si; break;
        default: *re = si; *im = -sr; break;
    }
    return QSV_OK;
}

// Inverse-CDF sampling of basis states: out[s] = smallest index i with sum_{j <= i} |amp_j|^2 > u[s] * total.
int qsvk_sample(qsv_state *st, int shots, const double *u, uint64_t *out) {
    const uint64_t chunks = (st->amps + SAMPLE_CHUNK - 1) / SAMPLE_CHUNK;
    const size_t sums_bytes = sizeof(double) * chunks, shot_bytes = sizeof(uint64_t) * shots;
    int rc = qsvk_ensure_matrix(st, sums_bytes + 3 * shot_bytes + 64);
    if (rc) return rc;
    char *base = reinterpret_cast<char *>(st->dev_matrix);
    double *d_sums = reinterpret_cast<double *>(base);
    uint64_t *d_chunk = reinterpret_cast<uint64_t *>(base + (sums_bytes + 15) / 16 * 16);
    double *d_resid = reinterpret_cast<double *>(d_chunk + shots);
    uint64_t *d_out = reinterpret_cast<uint64_t *>(d_resid + shots);
    hipLaunchKernelGGL(k_chunk_sums, dim3(grid_for(chunks, 1, 1 << 16)), dim3(QSV_BLOCK), 0, st->stream, st->data,
                       st->amps, d_sums);
    rc = check_launch();
    if (rc) return rc;
    std::vector<double> sums(chunks);
    QSV_HIP(hipMemcpyAsync(sums.data(), d_sums, sums_bytes, hipMemcpyDeviceToHost, st->stream));
    QSV_HIP(hipStreamSynchronize(st->stream));
    std::vector<double> cum(chunks + 1, 0.0);
    for (uint64_t c = 0; c < chunks; ++c) cum[c + 1] = cum[c] + sums[c];
    const double total = cum[chunks];
    if (!(total > 0.0)) return qsv_fail(QSV_EINVAL, "cannot sample from a register of zero norm");
    std::vector<uint64_t> chunk(shots);
    std::vector<double> resid(shots);
    for (int s = 0; s < shots; ++s) {
        if (!(u[s] >= 0.0 && u[s] < 1.0)) return qsv_fail(QSV_EINVAL, "uniform draws must lie in [0, 1)");
        const double target = u[s] * total;
        uint64_t c = std::upper_bound(cum.begin(), cum.end(), target) - cum.begin();  // first cum > target
        c = c == 0 ? 0 : c - 1;
        while (c + 1 < chunks && sums[c] == 0.0) ++c;  // never land in an empty chunk
        if (c >= chunks) c = chunks - 1;
        chunk[s] = c;
        resid[s] = target - cum[c];
    }
    QSV_HIP(hipMemcpyAsync(d_chunk, chunk.data(), shot_bytes, hipMemcpyHostToDevice, st->stream));
    QSV_HIP(hipMemcpyAsync(d_resid, resid.data(), sizeof(double) * shots, hipMemcpyHostToDevice, st->stream));
    hipLaunchKernelGGL(k_sample_in_chunk, dim3(shots), dim3(QSV_BLOCK), 0, st->stream, st->data, st->amps, d_chunk,
                       d_resid, d_out);
    rc = check_launch();
    if (rc) return rc;
    QSV_HIP(hipMemcpyAsync(out, d_out, shot_bytes, hipMemcpyDeviceToHost, st->stream));
    QSV_HIP(hipStreamSynchronize(st->stream));  // also covers the two pageable uploads above
    return QSV_OK;
}

int qsvk_probabilities(qsv_state *st, const uint64_t *indices, int count, double *out) {
    if (count <= 0) return QSV_OK;
    const size_t ibytes = sizeof(uint64_t) * count, obytes = sizeof(double) * count;
    int rc = qsvk_ensure_matrix(st, ibytes + obytes);
    if (rc) return rc;
    uint64_t *didx = reinterpret_cast<uint64_t *>(st->dev_matrix);
    double *dout = reinterpret_cast<double *>(reinterpret_cast<char *>(st->dev_matrix) + ibytes);
    QSV_HIP(hipMemcpyAsync(didx, indices, ibytes, hipMemcpyHostToDevice, st->stream));
    hipLaunchKernelGGL(k_gather_prob, dim3((count + 255) / 256), dim3(256), 0, st->stream, st->data, didx, count,
                       dout);
    rc = check_launch();
    if (rc) return rc;
    QSV_HIP(hipMemcpyAsync(out, dout, obytes, hipMemcpyDeviceToHost, st->stream));
    QSV_HIP(hipStreamSynchronize(st->stream));
    return QSV_OK;
}

int qsvk_fill_random(qsv_state *st, uint64_t seed, uint64_t index_offset, double *norm2) {
    const int grid = grid_for(st->amps, QSV_BLOCK * 8, QSV_REDUCE_BLOCKS);
    hipLaunchKernelGGL(k_fill_random, dim3(grid), dim3(QSV_BLOCK), 0, st->stream, st->data, st->amps, seed,
                       index_offset, st->partials);
    int rc = check_launch();
    if (rc) return rc;
    double s = 0.0;
    rc = sum_partials(st, grid, &s, nullptr);
    if (norm2) *norm2 = s;
    return rc;
}

int qsvk_scale(qsv_state *st, double re, double im) {
    const int grid = grid_for(st->amps, QSV_BLOCK * 8, 8192);
    hipLaunchKernelGGL(k_scale, dim3(grid), dim3(QSV_BLOCK), 0, st->stream, st->data, st->amps, cplx{re, im});
    return check_launch();
}

int qsvk_copy(amp_t *dst, const amp_t *src, uint64_t amps, hipStream_t stream) {
    static const int mode = [] { const char *e = getenv("QSV_COPY_MODE"); return e ? atoi(e) : 1; }();
    static const int regions = [] { const char *e = getenv("QSV_COPY_REGIONS"); return e ? atoi(e) : 0; }();
    const uint64_t per_block = mode == 0 ? QSV_BLOCK * COPY_ITEMS : QSV_BLOCK;
    const uint64_t bulk = amps / per_block * per_block;
    for (uint64_t done = 0; done < bulk;) {   // an AQL dispatch counts work-items in 32 bits
        const uint64_t blocks = std::min<uint64_t>((bulk - done) / per_block, 1ull << 23);
        const dim3 gd(static_cast<unsigned>(blocks)), bd(QSV_BLOCK);
        if (mode == 0) hipLaunchKernelGGL(k_copy<0>, gd, bd, 0, stream, dst + done, src + done, regions);
        else if (mode == 1) hipLaunchKernelGGL(k_copy<1>, gd, bd, 0, stream, dst + done, src + done, regions);
        else hipLaunchKernelGGL(k_copy<2>, gd, bd, 0, stream, dst + done, src + done, regions);
        done += blocks * per_block;
    }
    if (bulk < amps) {
        hipError_t e = hipMemcpyAsync(dst + bulk, src + bulk, sizeof(amp_t) * (amps - bulk), hipMemcpyDeviceToDevice, stream);
        if (e != hipSuccess) return qsv_fail(QSV_EHIP, std::string("hipMemcpyAsync: ") + hipGetErrorString(e));
    }
    return check_launch();
}

int qsvk_set_basis(qsv_state *st, uint64_t index) {
    const int grid = grid_for(st->amps, QSV_BLOCK * 8, 8192);
    hipLaunchKernelGGL(k_zero, dim3(grid), dim3(QSV_BLOCK), 0, st->stream, st->data, st->amps, index);
    return check_launch();
}
