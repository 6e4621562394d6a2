/*
 * A plain C client of include/qsv.h: no Python, no torch, no HIP headers -- only the C ABI.
 * Builds an n-qubit GHZ state with H + CX (gates.py:83-85,116-126 of the reference), reads it back, measures qubit 0
 * with a forced outcome and checks the collapsed register.  Exit code 0 = all checks passed, 3 = no GPU available
 * (qsv_create reported QSV_EHIP), anything else = failure.
 *
 *   gcc -std=c11 -Iinclude tests/c_abi/ghz.c -Lquantum_computations_amd -lqsv -lm -o ghz
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "qsv.h"

#define CHECK(call)                                                                              \
    do {                                                                                         \
        int rc_ = (call);                                                                        \
        if (rc_ != QSV_OK) {                                                                     \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, qsv_last_error());                     \
            return rc_ == QSV_EHIP ? 3 : 1;                                                      \
        }                                                                                        \
    } while (0)

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 12;
    const double r = 1.0 / sqrt(2.0);
    const double H[8] = {r, 0, r, 0, r, 0, -r, 0};
    qsv_state *st = NULL;
    if (qsv_version() != QSV_VERSION) return 2;
    CHECK(qsv_create(n, 0, &st));
    CHECK(qsv_apply_1q(st, 0, H));
    for (int q = 0; q + 1 < n; ++q) CHECK(qsv_apply_cx(st, q, q + 1));

    const uint64_t amps = 1ull << n;
    double *ket = malloc(sizeof(double) * 2 * amps);
    CHECK(qsv_download(st, ket, 0, amps));
    for (uint64_t i = 0; i < amps; ++i) {
        const double want = (i == 0 || i == amps - 1) ? r : 0.0;
        if (fabs(ket[2 * i] - want) > 1e-14 || fabs(ket[2 * i + 1]) > 1e-14) {
            fprintf(stderr, "amplitude %llu = (%g, %g), expected %g\n", (unsigned long long)i, ket[2 * i],
                    ket[2 * i + 1], want);
            return 1;
        }
    }
    double norm2 = 0.0;
    CHECK(qsv_norm2(st, &norm2));
    if (fabs(norm2 - 1.0) > 1e-14) return 1;

    /* an invalid call must fail with QSV_EINVAL and leave the register usable */
    if (qsv_apply_cx(st, 1, 1) != QSV_EINVAL || qsv_apply_1q(st, n, H) != QSV_EINVAL) return 1;

    /* measure qubit 0 in Z with the outcome forced to 1: the other n-1 qubits collapse to |1...1> */
    const double e0[4] = {1, 0, 0, 0}, e1[4] = {0, 0, 1, 0};
    int outcome = -1, left = -1;
    double p0 = 0, p1 = 0;
    CHECK(qsv_measure(st, 0, e0, e1, 1, 0.0, &outcome, &p0, &p1));
    CHECK(qsv_num_qubits(st, &left));
    if (outcome != 1 || left != n - 1 || fabs(p0 - 0.5) > 1e-14 || fabs(p1 - 0.5) > 1e-14) return 1;
    const uint64_t last = (amps >> 1) - 1;
    double prob = 0.0;
    CHECK(qsv_probabilities(st, &last, 1, &prob));
    if (fabs(prob - 1.0) > 1e-14) return 1;

    free(ket);
    CHECK(qsv_destroy(st));
    printf("ghz ok: n=%d\n", n);
    return 0;
}
