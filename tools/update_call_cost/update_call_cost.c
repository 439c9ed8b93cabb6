/* Cost of the per-instance entry point: N calls of svdf_update_csr (one (user, item, rating) instance each) + finish_round, from plain C.
 * What the reference CLI adds on top of this is its own loader thread, iterator and virtual call.  usage: update_call_cost [N=20000000] [stage_window knob] */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include "svdfeature_amd.h"
static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
int main(int argc, char **argv) {
    long n = argc > 1 ? atol(argv[1]) : 20000000L;
    svdf_seed(10);
    svdf_trainer *t = svdf_create(0, 0, 0, 0, -1);
    if (!t) { fprintf(stderr, "%s\n", svdf_last_error()); return 1; }
    svdf_set_param(t, "num_user", "1000000"); svdf_set_param(t, "num_item", "100000"); svdf_set_param(t, "num_global", "0");
    svdf_set_param(t, "num_factor", "64"); svdf_set_param(t, "learning_rate", "0.005"); svdf_set_param(t, "wd_user", "0.004");
    svdf_set_param(t, "wd_item", "0.004"); svdf_set_param(t, "base_score", "3");
    svdf_init_model(t); svdf_init_trainer(t);
    if (argc > 2) svdf_set_knob(t, "stage_window", atol(argv[2]));
    unsigned *u = malloc(n * 4), *it = malloc(n * 4);
    unsigned x = 12345u;
    for (long r = 0; r < n; r++) { x = x * 1664525u + 1013904223u; u[r] = (x >> 8) % 1000000u; x = x * 1664525u + 1013904223u; it[r] = (x >> 8) % 100000u; }
    const float val[2] = {1.0f, 1.0f};
    for (int round = 0; round < 3; round++) {
        double t0 = now_s();
        for (long r = 0; r < n; r++) {
            unsigned idx[2] = {u[r], it[r]};
            svdf_update_csr(t, 3.0f, 0, 1, 1, idx, val);
        }
        double t1 = now_s();
        svdf_finish_round(t);
        svdf_synchronize(t);
        double t2 = now_s();
        printf("round %d: %ld calls in %.3f s = %.1f ns per call; finish_round + synchronize %.3f s; %.1f M inst/s\n", round, n, t1 - t0, (t1 - t0) / n * 1e9,
               t2 - t1, n / (t2 - t0) / 1e6);
    }
    svdf_destroy(t);
    return 0;
}
