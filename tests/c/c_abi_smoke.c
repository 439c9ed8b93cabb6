/* A plain-C consumer of include/svdfeature_amd.h (no C++, no Python): what a C / cgo / JNI caller sees.
 * usage: c_abi_smoke <device: -2 host-only | -1 current GPU> <model_out>
 * Mirrors svd_feature.cpp:194-283: create -> set_param* -> init_model -> init_trainer -> save_model ->
 * (GPU only) per-instance update + predict -> save_model. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <svdfeature_amd.h>

static void put(svdf_trainer *t, const char *k, const char *v) {
    if (svdf_set_param(t, k, v) != 0) { fprintf(stderr, "set_param failed: %s\n", svdf_last_error()); exit(2); }
}

int main(int argc, char **argv) {
    if (argc < 3) return 64;
    int device = atoi(argv[1]);
    svdf_set_error_mode(1);
    svdf_trainer *t = svdf_create(0, 0, 0, 0, device);
    if (!t) { fprintf(stderr, "create failed: %s\n", svdf_last_error()); return 3; }
    put(t, "base_score", "3"); put(t, "learning_rate", "0.005"); put(t, "wd_item", "0.004"); put(t, "wd_user", "0.004");
    put(t, "num_user", "50"); put(t, "num_item", "40"); put(t, "num_global", "0"); put(t, "num_factor", "12");
    put(t, "some_unknown_key_of_another_component", "ignored");   /* svd_feature.cpp:145-150 broadcasts every key */
    svdf_seed(10);
    if (svdf_init_model(t) != 0 || svdf_init_trainer(t) != 0) { fprintf(stderr, "init failed: %s\n", svdf_last_error()); return 4; }
    if (device != -2) {
        unsigned idx[2];
        float val[2] = {1.0f, 1.0f};
        for (int r = 0; r < 500; r++) {
            idx[0] = (unsigned)((r * 7) % 50); idx[1] = (unsigned)((r * 13) % 40);
            if (svdf_update_csr(t, (float)(1 + r % 5), 0, 1, 1, idx, val) != 0) { fprintf(stderr, "update: %s\n", svdf_last_error()); return 5; }
        }
        idx[0] = 3; idx[1] = 4;
        float p = svdf_predict_csr(t, 0.0f, 0, 1, 1, idx, val);
        printf("pred %.9g instances %ld\n", p, (long)svdf_counter(t, 0));
        idx[0] = 50;   /* out of range: must be reported, not executed */
        if (svdf_update_csr(t, 1.0f, 0, 1, 1, idx, val) == 0) { fprintf(stderr, "bound check missing\n"); return 6; }
        printf("error text: %s\n", svdf_last_error());
    }
    FILE *fo = fopen(argv[2], "wb");
    if (!fo) return 7;
    unsigned char mtype[4] = {0, 0, 0, 0};
    fwrite(mtype, 1, 4, fo);                       /* the caller writes SVDTypeParam, svd_feature.cpp:188 */
    if (svdf_save_model(t, fo) != 0) { fprintf(stderr, "save: %s\n", svdf_last_error()); return 8; }
    fclose(fo);
    svdf_destroy(t);
    printf("ok %s\n", svdf_version());
    return 0;
}
