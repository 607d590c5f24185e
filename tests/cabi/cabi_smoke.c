/* cabi_smoke.c -- the C ABI (include/ssq_hip.h) used from plain C, no Python, no torch:
 * device memory through ssq_malloc / ssq_memcpy_*, then
 *   ssq_pad_signal   against the reflect rule          (utils/common.py:54-158)
 *   ssq_phase_cwt    against |Im(dWx/Wx)| / 2pi        (algos.py:706-740)
 *   ssq_ssqueeze     against the reference's loop nest (algos.py:859-924, 'log' grid)
 *   ssq_colsum       against the row-ordered sum       (_cwt.py:472-476)
 * on small seeded inputs. Built and run by tests/test_gpu_cabi_c.py:
 *   gcc -std=c99 -Iinclude tests/cabi/cabi_smoke.c -Lssqueezepy_amd -lssq_hip -lm
 * Prints "PASS" and returns 0 when every result matches. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "ssq_hip.h"

#define CHECK(call) do { int rc_ = (call); if (rc_) { \
    fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, ssq_last_error()); return 1; } } while (0)

static uint32_t lcg_state = 12345u;
static float frand(void) {              /* uniform in (-1, 1) */
    lcg_state = lcg_state * 1664525u + 1013904223u;
    return (float)((lcg_state >> 8) & 0xFFFFFF) / 8388608.0f - 1.0f;
}

int main(void) {
    int ndev = 0;
    CHECK(ssq_device_count(&ndev));
    if (ndev < 1) { fprintf(stderr, "no device\n"); return 1; }
    CHECK(ssq_set_device(0));
    char name[64]; int cus = 0; int64_t hbm = 0;
    CHECK(ssq_device_info(0, name, 64, &cus, &hbm));
    printf("device %s, %d CUs, %.0f GB\n", name, cus, (double)hbm / 1e9);

    /* ---- pad_signal: reflect */
    enum { N = 37, N1 = 9, N2 = 5 };
    float x[N], xp[N1 + N + N2];
    for (int i = 0; i < N; ++i) x[i] = frand();
    void *dx, *dxp;
    CHECK(ssq_malloc(&dx, sizeof x)); CHECK(ssq_malloc(&dxp, sizeof xp));
    CHECK(ssq_memcpy_h2d(dx, x, sizeof x, NULL));
    CHECK(ssq_pad_signal(SSQ_F32, dx, dxp, 1, N, N1, N2, SSQ_PAD_REFLECT, NULL));
    CHECK(ssq_memcpy_d2h(xp, dxp, sizeof xp, NULL));
    CHECK(ssq_stream_synchronize(NULL));
    for (int i = 0; i < N1 + N + N2; ++i) {
        int s = i - N1;                        /* reflect without repeating the edge */
        if (s < 0) s = -s;
        if (s >= N) s = 2 * (N - 1) - s;
        if (xp[i] != x[s]) { fprintf(stderr, "pad mismatch at %d\n", i); return 1; }
    }

    /* ---- phase transform, reassignment, column sum on a (NA, NN) transform */
    enum { NA = 24, NN = 200 };
    static float Wx[NA * NN * 2], dWx[NA * NN * 2], w[NA * NN], Tx[NA * NN * 2], cs[NN];
    for (int i = 0; i < NA * NN * 2; ++i) { Wx[i] = frand(); dWx[i] = 3.0f * frand(); }
    const double gamma = 1e-3;
    void *dW, *dD, *dw, *dT, *dc, *dcs;
    CHECK(ssq_malloc(&dW, sizeof Wx)); CHECK(ssq_malloc(&dD, sizeof dWx));
    CHECK(ssq_malloc(&dw, sizeof w)); CHECK(ssq_malloc(&dT, sizeof Tx));
    CHECK(ssq_malloc(&dc, NA * sizeof(float))); CHECK(ssq_malloc(&dcs, sizeof cs));
    CHECK(ssq_memcpy_h2d(dW, Wx, sizeof Wx, NULL));
    CHECK(ssq_memcpy_h2d(dD, dWx, sizeof dWx, NULL));
    CHECK(ssq_phase_cwt(SSQ_F32, dW, dD, dw, 1, NA, NN, gamma, NULL));
    CHECK(ssq_memcpy_d2h(w, dw, sizeof w, NULL));
    CHECK(ssq_stream_synchronize(NULL));
    for (int q = 0; q < NA * NN; ++q) {
        float c = Wx[2 * q], d = Wx[2 * q + 1], a = dWx[2 * q], b = dWx[2 * q + 1];
        double ref = fabs((double)(b * c - a * d) / ((double)(c * c + d * d) * 6.283185307179586));
        if (hypotf(c, d) < (float)gamma) { if (!isinf(w[q])) { fprintf(stderr, "w inf\n"); return 1; } }
        else if (fabs(w[q] - ref) > 1e-6 * ref + 1e-12) { fprintf(stderr, "w mismatch at %d\n", q); return 1; }
    }

    /* 'log' frequency grid f_k = f0 * 2^(k/8), k < NA; weights ln2/8; flipud */
    double params[5] = {log2(1e-3), 1.0 / 8, 0, 0, 0};
    float cst[NA];
    for (int i = 0; i < NA; ++i) cst[i] = (float)(log(2.0) / 8);
    CHECK(ssq_memcpy_h2d(dc, cst, sizeof cst, NULL));
    CHECK(ssq_ssqueeze(SSQ_F32, dW, dD, NULL, dT, dc, 0, 1, NA, NN, gamma, SSQ_GRID_LOG, params, 1,
                       NULL, NULL));
    CHECK(ssq_memcpy_d2h(Tx, dT, sizeof Tx, NULL));
    CHECK(ssq_colsum(SSQ_F32, dT, NULL, dcs, 1, NA, NN, NULL));
    CHECK(ssq_memcpy_d2h(cs, dcs, sizeof cs, NULL));
    CHECK(ssq_stream_synchronize(NULL));
    static float Tref[NA * NN * 2];
    memset(Tref, 0, sizeof Tref);
    for (int j = 0; j < NN; ++j)
        for (int i = 0; i < NA; ++i) {                 /* rows in order: the reference's sum order */
            int q = i * NN + j;
            float c = Wx[2 * q], d = Wx[2 * q + 1], a = dWx[2 * q], b = dWx[2 * q + 1];
            if (!((double)hypotf(c, d) > gamma)) continue;
            double wv = fabs((double)(b * c - a * d) / ((double)(c * c + d * d) * 6.283185307179586));
            double t = (log2(wv) - params[0]) / params[1];
            long k = t > 0 ? (t >= NA - 1 ? NA - 1 : (long)rint(t)) : 0;
            if (k > NA - 1) k = NA - 1;
            k = NA - 1 - k;                            /* flipud */
            Tref[2 * (k * NN + j)] += c * cst[i];
            Tref[2 * (k * NN + j) + 1] += d * cst[i];
        }
    for (int q = 0; q < NA * NN * 2; ++q)
        if (Tx[q] != Tref[q]) { fprintf(stderr, "Tx mismatch at %d: %g vs %g\n", q, Tx[q], Tref[q]); return 1; }
    for (int j = 0; j < NN; ++j) {
        float acc = 0.f;
        for (int i = 0; i < NA; ++i) acc = acc + Tx[2 * (i * NN + j)];
        if (cs[j] != acc) { fprintf(stderr, "colsum mismatch at %d\n", j); return 1; }
    }
    ssq_free(dx); ssq_free(dxp); ssq_free(dW); ssq_free(dD); ssq_free(dw); ssq_free(dT);
    ssq_free(dc); ssq_free(dcs);
    printf("PASS\n");
    return 0;
}
