/* svdf_train_bulk.c -- the round loop of svd_feature.cpp:260-288 with whole passes instead of one virtual call per instance:
 * what a maintainer's patched SVDTrainTask::run_task looks like ("Whole passes without per-instance calls"),
 * as a stand-alone plain-C program over include/svdfeature_amd.h.  It is NOT the reference's CLI: it understands the subset of
 * the config file that the loop needs (every `name = value` pair goes to svdf_set_param exactly like configure_trainer does,
 * svd_feature.cpp:145-150; buffer_feature, model_out_folder, num_round, input_type, format_type / model_type, active_type,
 * extend_type, seed are read here like set_param_inner does, :96-111) and writes the same NNNN.model files.
 *
 *   svdf_train_bulk <config> [name=value ...]
 *
 * input_type 0 / 1: one resident data set for all rounds (svdf_dataset_from_buffer_file);  input_type 2: a new rank-pair pass
 * per round (svdf_dataset_from_rank_buffer_file, pairs drawn on the device when the file's rows are plain). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <svdfeature_amd.h>

#define MAXP 512
static char names[MAXP][128], vals[MAXP][512];
static int np_ = 0;

static void add_pair(const char *n, const char *v) {
    if (np_ < MAXP) { snprintf(names[np_], sizeof(names[0]), "%s", n); snprintf(vals[np_], sizeof(vals[0]), "%s", v); np_++; }
}
static void trim(char *s) {
    char *b = s;
    while (*b == ' ' || *b == '\t' || *b == '"') b++;
    memmove(s, b, strlen(b) + 1);
    size_t n = strlen(s);
    while (n > 0 && (s[n - 1] == ' ' || s[n - 1] == '\t' || s[n - 1] == '\n' || s[n - 1] == '\r' || s[n - 1] == '"')) s[--n] = 0;
}
static int read_config(const char *path) {   /* apex-utils/apex_config.h:31-124: name = value, # comments, optional quotes */
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    char line[1024];
    while (fgets(line, sizeof(line), f)) {
        char *hash = strchr(line, '#');
        if (hash) *hash = 0;
        char *eq = strchr(line, '=');
        if (!eq) continue;
        *eq = 0;
        char n[512], v[512];
        snprintf(n, sizeof(n), "%s", line);
        snprintf(v, sizeof(v), "%s", eq + 1);
        trim(n); trim(v);
        if (n[0]) add_pair(n, v);
    }
    fclose(f);
    return 0;
}
static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
static const char *get(const char *name, const char *dflt) {
    const char *r = dflt;
    for (int i = 0; i < np_; i++) if (!strcmp(names[i], name)) r = vals[i];   /* later pairs win (command line after the file) */
    return r;
}
/* svd_feature.cpp:184-191, with the write BESIDE the next pass: svdf_save_model_begin snapshots the model in HBM and returns, a writer thread streams the
 * snapshot into the file; the file of round r is completed (svdf_save_model_end, fclose) before round r + 1's is opened.  The same NNNN.model bytes. */
static FILE *pending_fo = NULL;
static int save_end(svdf_trainer *t) {
    if (!pending_fo) return 0;
    int rc = svdf_save_model_end(t);
    if (fclose(pending_fo) != 0) rc = -1;
    pending_fo = NULL;
    return rc;
}
static int save(svdf_trainer *t, const char *folder, int round, const unsigned char mtype[4]) {
    char path[1024];
    if (save_end(t) != 0) return -1;
    snprintf(path, sizeof(path), "%s/%04d.model", folder, round);
    FILE *fo = fopen(path, "wb");
    if (!fo) return -1;
    fwrite(mtype, 1, 4, fo);
    if (getenv("SVDF_BULK_SYNC_SAVE")) {   /* A/B: the reference's protocol, the file complete before the next pass starts */
        int rc = svdf_save_model(t, fo);
        fclose(fo);
        return rc;
    }
    pending_fo = fo;
    return svdf_save_model_begin(t, fo);
}

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: svdf_train_bulk <config> [name=value ...]\n"); return 64; }
    if (read_config(argv[1]) != 0) { fprintf(stderr, "cannot open %s\n", argv[1]); return 2; }
    for (int i = 2; i < argc; i++) {
        char buf[1024];
        snprintf(buf, sizeof(buf), "%s", argv[i]);
        char *eq = strchr(buf, '=');
        if (eq) { *eq = 0; add_pair(buf, eq + 1); }
    }
    const int input_type = atoi(get("input_type", "0"));
    const int extend_type = atoi(get("extend_type", "0"));
    int format_type = atoi(get("format_type", get("model_type", "2")));   /* 2 = AUTO_DETECT (apex_svd_model.h:279-286) */
    if (format_type == 2) format_type = (input_type != 0) ? 1 : (extend_type == 0 ? 0 : 1);
    const unsigned char mtype[4] = {(unsigned char)format_type, (unsigned char)atoi(get("active_type", "0")), (unsigned char)extend_type,
                                    (unsigned char)atoi(get("variant_type", "0"))};
    const char *buffer = get("buffer_feature", "NULL"), *folder = get("model_out_folder", "models");
    const int num_round = atoi(get("num_round", "10"));
    svdf_seed((unsigned)atoi(get("seed", "10")));   /* svd_feature.cpp:98,293 */
    svdf_trainer *t = svdf_create(mtype[0], mtype[1], mtype[2], mtype[3], -1);
    for (int i = 0; i < np_; i++) svdf_set_param(t, names[i], vals[i]);
    svdf_init_model(t);
    svdf_init_trainer(t);
    if (save(t, folder, 0, mtype) != 0) { fprintf(stderr, "cannot write to %s\n", folder); return 3; }
    svdf_dataset *ds = NULL;
    if (input_type != 2) ds = svdf_dataset_from_buffer_file(t, buffer, format_type == 1);
    long trained = 0;
    double t_second = 0.0;   /* wall clock at the start of round 2: round 1 carries one-off work (the pass is captured as a hipGraph) */
    for (int r = 1; r <= num_round; r++) {   /* svd_feature.cpp:272-283 */
        if (r == 2) t_second = now_s();
        svdf_set_round(t, r - 1);
        if (input_type == 2) ds = svdf_dataset_from_rank_buffer_file(t, buffer);   /* this round's pairs */
        svdf_train_dataset(t, ds);
        svdf_finish_round(t);
        trained += (long)svdf_dataset_info(ds, 0);
        if (input_type == 2) { svdf_dataset_destroy(ds); ds = NULL; }
        save(t, folder, r, mtype);
    }
    if (save_end(t) != 0) { fprintf(stderr, "cannot complete the last model file\n"); return 3; }
    if (ds) svdf_dataset_destroy(ds);
    printf("svdf_train_bulk: %d rounds, %ld instances, %s\n", num_round, trained, svdf_version());
    if (num_round >= 2) printf("svdf_train_bulk: seconds per round (rounds 2..%d, model files written beside the next pass): %.6f\n", num_round, (now_s() - t_second) / (num_round - 1));
    svdf_destroy(t);
    return 0;
}
