/* A plain C caller of libblueice_hip: the reference's smallest configuration (2 sources, one shape parameter with
 * 3 anchors, 40 bins -- BASELINE.json configs[0]) built from closed-form templates, evaluated at a few points.
 * It prints one "z rate0 rate1 loglikelihood status" line per point; tests/test_capi_loads.py::test_c_program
 * compiles it with gcc, runs it on the GPU and checks the numbers against the oracle.
 *
 *   gcc -O2 -Iinclude examples/c_abi_demo.c -o c_abi_demo -Lblueice_amd/lib -lblueice_hip -lm \
 *       -Wl,-rpath,$PWD/blueice_amd/lib
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "blueice_hip.h"

#define S 2
#define B 40
#define A 3

static void die(bi_ctx* ctx, const char* what, int rc) {
    fprintf(stderr, "%s failed (%d): %s\n", what, rc, bi_last_error(ctx));
    exit(1);
}

int main(void) {
    bi_ctx* ctx = NULL;
    int rc = bi_create(0, &ctx);
    if (rc) die(NULL, "bi_create", rc);

    /* anchors of the shape parameter and the model at each of them: a Gaussian whose mean moves with z next to a
     * flat background; expected events per source */
    const int32_t n_anchor[1] = {A};
    const double anchor_z[A] = {-1.0, 0.0, 1.0};
    static double ps[A][S][B], mus[A][S];
    for (int a = 0; a < A; ++a) {
        double norm = 0.0;
        for (int b = 0; b < B; ++b) {
            const double x = (b + 0.5) / B * 10.0 - 5.0, mean = 0.8 * anchor_z[a];
            ps[a][0][b] = exp(-0.5 * (x - mean) * (x - mean));
            norm += ps[a][0][b];
            ps[a][1][b] = 1.0 / B;
        }
        for (int b = 0; b < B; ++b) ps[a][0][b] /= norm;
        mus[a][0] = 1000.0 * (1.0 + 0.05 * anchor_z[a]);
        mus[a][1] = 500.0;
    }
    if ((rc = bi_upload_model(ctx, 1, n_anchor, anchor_z, S, B, &ps[0][0][0], &mus[0][0], NULL, -1))) die(ctx, "bi_upload_model", rc);

    /* data: the rounded expectation at z = 0.25 */
    double counts[B];
    for (int b = 0; b < B; ++b) {
        const double t = 0.25;
        const double p0 = (1 - t) * ps[1][0][b] + t * ps[2][0][b];
        const double m0 = (1 - t) * mus[1][0] + t * mus[2][0];
        counts[b] = floor(m0 * p0 + 500.0 / B + 0.5);
    }
    if ((rc = bi_upload_counts(ctx, 1, counts))) die(ctx, "bi_upload_counts", rc);

    const double z[6] = {0.25, -1.0, 1.0, 0.0, -0.4, 3.0};           /* the last one lies outside the anchors */
    const double rate_scale[6][S] = {{1, 1}, {1, 1}, {0.9, 1.2}, {1, 0}, {1.1, 1}, {1, 1}};
    double ll[6];
    int32_t status[6];
    if ((rc = bi_eval(ctx, 6, z, &rate_scale[0][0], NULL, ll, status))) die(ctx, "bi_eval", rc);
    for (int i = 0; i < 6; ++i) printf("%.17g %.17g %.17g %.17g %d\n", z[i], rate_scale[i][0], rate_scale[i][1], ll[i], (int)status[i]);

    double one;
    int32_t st;
    if ((rc = bi_eval(ctx, 1, z, &rate_scale[0][0], NULL, &one, &st))) die(ctx, "bi_eval(P=1)", rc);
    if (one != ll[0]) { fprintf(stderr, "single call and batch disagree: %.17g vs %.17g\n", one, ll[0]); return 1; }
    bi_destroy(ctx);
    return 0;
}
