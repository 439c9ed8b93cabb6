// svdf_types.h -- PODs shared by the host engine and the HIP kernels.
//
// On-disk layouts restate the reference's structs byte for byte (no reference code is included):
//   ModelParam  = SVDModelParam  (apex_svd_model.h:373-435), 1056 bytes
//   TypeParam   = SVDTypeParam   (apex_svd_model.h:242-261), 4 bytes
// TrainParam restates SVDTrainParam (apex_svd_model.h:291-344).
#ifndef SVDF_TYPES_H_
#define SVDF_TYPES_H_

#include <stdint.h>

namespace svdf {

struct TypeParam {
    uint8_t format_type, active_type, extend_type, variant_type;
};
static_assert(sizeof(TypeParam) == 4, "SVDTypeParam is 4 bytes");

struct ModelParam {
    int num_user, num_item, num_factor, num_global;
    float u_init_sigma, i_init_sigma, base_score;
    int no_user_bias, num_ufeedback;
    float ufeedback_init_sigma;
    int num_randinit_ufactor, num_randinit_ifactor;
    int common_latent_space, user_nonnegative, common_feedback_space, extend_flag, item_nonnegative;
    int reserved[247];
};
static_assert(sizeof(ModelParam) == 1056, "SVDModelParam is 1056 bytes on disk");

struct TrainParam {
    float learning_rate;
    int decay_learning_rate;
    float decay_rate, min_learning_rate;
    float wd_user, wd_item, wd_user_bias, wd_item_bias;
    int reg_method;
    float wd_global;
    int reg_global;
    unsigned num_regfree_global;
    float scale_lr_ufeedback, wd_ufeedback_user, wd_ufeedback, wd_ufeedback_bias;
};

// active_type constants (apex_svd_model.h:61-79)
enum { ACT_LINEAR = 0, ACT_SIGMOID_L2 = 1, ACT_SIGMOID_LIKELIHOOD = 2, ACT_SIGMOID_RANK = 3,
       ACT_HINGE_SMOOTH = 5, ACT_HINGE_L2 = 6, ACT_SIGMOID_QSGRAD = 7 };
// svdpp_tag (apex_svd_data.h:353-371)
enum { TAG_DEFAULT = 0, TAG_START = 1, TAG_END = 2, TAG_MIDDLE = 3 };

// ---- device views -------------------------------------------------------------------------
// ParameterSet (apex_svd_base.h:33-75) flattened: wd[j] applies to ids <= bound[j].
struct DevRanges {
    const unsigned *bound;
    const float *wd;
    int n;
};
// SparseFeatureArray<float> (apex-utils/apex_utils.h:140-196) as CSR in HBM.
struct DevSideTable {
    const unsigned *row_ptr;
    const unsigned *index;
    const float *value;
    unsigned num_row;
};

// Everything a kernel needs, passed by value in the kernarg segment.
// HBM layout of the model: ONE row-major matrix W[n_uiset][pitch] holding
// W_ufeedback | W_user | W_item back to back exactly like the reference's W_uiset
// (apex_svd_model.h:511-556), pitch = ceil(k/4)*4 floats so every row is float4 aligned, pad
// floats kept at 0; ONE bias vector bias[n_uiset] with the same row numbering; g_bias[num_global].
struct DevParams {
    float *W;
    float *bias;
    float *g_bias;
    float *svdpp_state;       // 2*pitch + 4 floats: tmp_fb, old_fb, norm, tmp_bias, old_bias
    int pitch, k;
    unsigned user_off, item_off, fb_off;   // first row of W_user / W_item / W_ufeedback in W
    int num_user, num_item, num_global, num_ufeedback;
    float base_score;
    int active_type, no_user_bias, user_nonnegative, user_group;
    unsigned *ref_ui;         // lazy decay (reg_method 4/5): sample_counter of the last touch per W_uiset row
    unsigned *ref_global;     // the same per global id (reg_global 4/5)
    // opt-in relaxed mode for SHARED ids (not the reference's semantics): ids at or above these
    // thresholds (and all global ids when relax_global) are left out of the conflict schedule and updated with atomic adds
    unsigned relax_user_from, relax_item_from;
    int relax_global;
    int relax_feedback;       // relaxed mode for the implicit-feedback rows of user-group trainers (atomic scatter)
    int g_stride;             // floats between consecutive global biases in device memory (1, or 32 in relaxed-global mode)
    int hot_reduce;           // knob: workgroup pre-reduction of a relaxed shared user row (k_fused HOTU)
    int xcd_remap;            // 1: consecutive tiles of a batch go to the same XCD (blockIdx%8), see k_basicmf
    unsigned imfb_disable;    // extend_type 2: bit l = ufeedback_disable_level l (apex_multi_imfb.h:58-67)
    int imfb_deep;            // extend_type 2: the data nested deeper than IMFB_DEPTH levels at some point: k_imfb<..., IMFB_DEPTH_MAX>
    int fewrow_fast;          // knob: 1 = few-row data sets in the usual configuration run k_fewrow_fast instead of k_fused
    int store_mode;           // row-store cache policy of k_basicmf: 0 plain, 1 nontemporal, 2 sc1 write-through
    int fewrow_i16;           // k = 128 few-row kernel: sixteen lanes per row, 4 instances per wave (k_fewrow_i16)
    int svdpp_helpers;        // waves per user in k_svdpp_wave (1 = one wave per user; 4: helper waves gather / scatter the feedback rows through LDS)
    int small_blocks;         // one-wave workgroups for levels of <= 8192 waves in the grid-stride kernels (launch_shape, svdf_device.h)
    int basic_i8;             // k = 64 basicMF: eight lanes per row, 8 instances per wave instruction (k_basicmf_i8)
    int load_mode;            // row-load cache policy of k_basicmf / k_fewrow_fast: 1 nontemporal hint -- a level reads each row once; measured
                              // -2.2 ... -4.2 % per pass at k = 64 (two cache lines per row), flat at 128 / 256, +16 % at k = 32 (one line): the
                              // engine picks 1 for rows of >= 256 bytes, 0 below (load_mode knob 2 = auto)
    // SVDTrainParam
    float lr, wd_user, wd_item, wd_user_bias, wd_item_bias, wd_global;
    int reg_method, reg_global;
    unsigned num_regfree_global;
    float scale_lr_ufeedback, wd_ufeedback, wd_ufeedback_bias;
    DevRanges u_rng, i_rng, g_rng;
    DevSideTable feat_user, feat_item;
};

// One conflict-free batch of the basicMF schedule: instance s of the batch is
// (user[s], item[s], label[s]) with optional feature values.
struct BasicSchedule {
    const unsigned *user;
    const unsigned *item;
    const float *label;
    const float *uval;   // nullptr when every value is 1.0f
    const float *ival;
};

// One conflict-free level of RUNS (svdf_k_runs.hip): run s = item[s] with up to 8 ratings (user[j][s], label[j][s]); user[j][s] == SLOT_ABSENT ends it
struct RunSchedule {
    const unsigned *item;
    const unsigned *user[8];
    const float *label[8];
};

// USER-RUN UNITS of rank pairs (svdf_k_wave.hip: k_pair_units; svdf_punit.cpp; round 6): the reference's own pair order (apex_svd_data.cpp:946-965) emits a
// user's pairs back to back.  A unit = up to `cap` CONSECUTIVE pairs of one user whose item ids are pairwise distinct; a wave keeps the user's row in
// registers and walks the unit's pairs in file order.  Pair columns stay in FILE order: lo / hi = the lower / higher item id of the pair (the entry order of the
// reference's merged row, :828-860), vlo = the sign of the lower entry (+1: it is the positive item), the higher entry carries -vlo; labels are 1.
struct PairUnit { unsigned user; int begin; int count; int pad; };
struct PairUnitSchedule {
    const PairUnit *units;   // launch order: level by level
    const unsigned *lo, *hi;
    const float *vlo;
};

// One conflict-free batch of "few-row" instances for the fused kernel: at most 2 user ids and 2 item ids
// per instance (slot value 0xFFFFFFFF = absent), any number of global features (CSR over the batch order),
// no id repeated inside an instance.  Covers pairwise-rank pairs (nu=1, ni=2), neighbourhood rows
// (globals + 1 + 1) and basicMF with side ids.
struct FusedSchedule {
    const float *label;
    const unsigned *uidx[2];
    const float *uval[2];
    const unsigned *iidx[2];
    const float *ival[2];
    const int *gptr;          // [n+1], nullptr when the data set has no global feature
    const unsigned *gidx;
    const float *gval;
    const unsigned *gsi[4];   // inline global slots (SLOT_ABSENT = none), nullptr unless every instance has <= 4 distinct global ids
    const float *gsv[4];
};
enum { SLOT_ABSENT = 0xFFFFFFFFu };

// Instance stream in SVDFeatureCSR layout (apex_svd_data.h:109-127) resident in HBM.
struct DevCSR {
    const float *row_label;
    const int *row_ptr;        // 3*num_row+1
    const unsigned *feat_index;
    const float *feat_value;
    int unit_values;                  // user-unit streams only: 1 = every row of every fast-path unit has feature values 1.0
    const unsigned char *row_fresh;   // user-unit streams only, may be null: 1 = this row's item already occurred earlier in
                                      // the same fast-path unit, so its row / bias must be read when the row is reached
};

// One SVD++ unit = all rows of one user (START..END blocks concatenated).
struct DevUnit {
    int fb_begin, fb_end;      // range in fb_index / fb_value
    int row_begin, row_end;    // range of rows in the DevCSR
    int flags;                 // bit0: starts here (prepare_ufeedback), bit1: ends here (update_ufeedback),
                               // bit2: save state at exit, bit3: load state at entry, bit4: UNIT_SIMPLE fast path
};
// A unit as k_svdpp_wave wants it: in LAUNCH order (entry s of a level's range, no order[] indirection) and with what the wave would
// otherwise fetch through two more dependent loads -- the first entry of its rows (row_ptr[3 row_begin]) and its user id.
struct DevUnitX {
    DevUnit u;
    int e0;
    unsigned user;
    int pad;
};
// extend_type 2 (multi-level implicit feedback): one block of a unit.  [fb_begin, fb_end) is the list a DEFAULT / START block
// prepares its level from, [sc_begin, sc_end) the list a DEFAULT / END block scatters through (its own list, which may differ
// from the one the level was opened with, apex_multi_imfb.h:186-190).
struct DevBlk {
    int fb_begin, fb_end, sc_begin, sc_end;
    int row_begin, row_end;
    int tag;
};
#define IMFB_DEPTH 4        // nested implicit-feedback levels of the usual build of k_imfb (all in registers)
#define IMFB_DEPTH_MAX 16   // ... of the build the engine switches to for deeper data; beyond that the data is refused
enum { UNIT_START = 1, UNIT_END = 2, UNIT_SAVE = 4, UNIT_LOAD = 8,
       UNIT_SIMPLE = 16 };   // host-verified: rows are (0,1,1) with one user id, distinct feedback ids (repeated items: row_fresh)

// Window-minibatch data set (svdf_k_window.hip): the instances of one exchange window grouped by user.
// urec is in LAUNCH order (users sorted by instance count, descending); a user's instances are entries [begin, begin + count)
// of the user-grouped columns, in file order; slot[s] is where instance s's item-side contribution goes: slots are laid out item
// by item (iptr), inside an item in file order.
struct WinUser { unsigned user; int begin; int count; int pad; };
struct WindowSchedule {
    const WinUser *urec;
    long nusers;
    const unsigned *item;
    const float *label;         // nullptr: every label is 1.0f (rank pairs)
    const int *slot;
    const float *uval, *ival;   // nullptr when every feature value is 1.0f
    const int *iptr;            // [num_item + 1]
    float *contrib;             // [slots][pitch] scratch, owned by the trainer (one slot per item entry)
    float *cbias;               // [slots]
    // rank pairs (user, positive item, negative item): a second item entry per instance; entry 0 is the lower item id like in
    // the reference's merged row (apex_svd_data.cpp:828-860), ival / ival1 carry the signs.  nullptr for one-item instances.
    const unsigned *item1;
    const int *slot1;
    const float *ival1;
    int contrib_bf16;           // 1: contribution rows are bfloat16 (amd:contrib = bf16), sums stay fp32
    // ORDERED SUB-STEPS for hot items (round 6; one-GPU window sequences, in-place sums only; 0 = off).  An item with MORE than hot_sub slots in the
    // window is "hot" there: the users' walk stores, instead of the would-be change, what it was computed FROM -- the user's vector (tmp_u), bias
    // and the label -- and k_window_apply's hot lane applies the item's slots in file order in sub-steps of hot_sub: every sub-step's changes are computed
    // against the row as the previous sub-step left it (oracle/svdf_oracle.c: svdo_update_window_substeps).  Needs unit values, one item entry,
    // fp32 contribution rows.
    int hot_sub;
    float *clabel;              // [slots] scratch: the label of a hot entry
};

// Window-minibatch data set of USER UNITS (svdf_k_wunit.hip): user-group (SVD++) blocks and rows with global
// features.  A unit = one user's part of the window: `seg_count` segments walked in file order by one lane group, a segment = one
// DEFAULT block or START..END span (its feedback list + its rows) of a user-group pass, or all the user's rows of a random-order window
// (no feedback list).  The unit record carries its FIRST segment inline (most units have one), segs[seg_begin + s] the others.
// Rows are regrouped unit by unit; row r's entries are ent[rptr[2r] .. rptr[2r+1]) = global entries and ent[rptr[2r+1] .. rptr[2r+2]) =
// item entries (rptr == nullptr: the fixed layout, `estride` entries per row, the last one the item entry).
// An entry = (id, value, slot): slot is where the entry's contribution goes -- item entries: a row of contrib / a word of cbias; global
// entries: a word of gcontrib; feedback entries (fbent): the row a segment scatters into at its end.  Row slots are laid out target by
// target (tptr over the replicated rows: feedback rows first, then item rows -- the order of W_uiset and of the wire buffer), inside a
// target in file order; global slots global id by global id (gptr).
struct WinSeg { int fb_begin, fb_count, row_begin, row_count; };
struct WinUnit { unsigned user; int seg_begin; int seg_count; int rows; WinSeg first; };
struct WinEnt { unsigned idx; float val; int slot; int pad; };
// deferred feedback scatter (round 5): slot s of a feedback row holds WHO contributes -- the segment whose scaled delta d the sum kernel reads
// from dvec[seg] -- and the entry's value; the contribution (w + d val) - w is formed by k_wunit_sum against the row it is about to update
struct WinFbRec { int seg; float val; };
// one-GPU windows: the targets that have slots at all (target, first slot, one past its last): the in-place sums walk this list instead of every row
struct WinTouched { int t, b, e; };
struct WUnitSchedule {
    const WinUnit *units;
    long nunits;
    const WinSeg *segs;
    const float *label;
    const float *uval;          // nullptr: every user value is 1.0f
    const int *rptr;            // [2 nrows + 1], or nullptr with estride
    int estride;
    const WinEnt *ent;
    const WinEnt *fbent;
    float *contrib, *cbias, *gcontrib;
    const int *tptr;            // [nfb_rows + nitem_rows + 1]
    const int *gptr;            // [num_global + 1]
    long nfb_rows, nitem_rows, nglobal;
    int contrib_bf16;           // 1: contribution rows are bfloat16 (amd:contrib = bf16); bias / global-bias contributions stay fp32
    const WinFbRec *fbrec;      // nullptr: feedback contributions are rows in `contrib` (written by the walk); else deferred: [tptr[nfb_rows]] records
    float *dvec, *dbias;        // deferred: the scaled delta of every segment (row pitch = the model's), its bias delta
    int user_bias;              // 1 unless no_user_bias
    const WinTouched *touched;  // nullptr: every target is visited (wire buffers need the untouched rows' zeros); else [ntouched]
    long ntouched;
};

}  // namespace svdf
// replicated (item-side) parameter ranges of the multi-GPU exchange, packed back to back: range r covers packed
// positions [off[r], off[r+1])
#define SVDF_MAX_DELTA_RANGES 6
struct DeltaRanges {
    float *base[SVDF_MAX_DELTA_RANGES];
    long off[SVDF_MAX_DELTA_RANGES + 1];       // position in the packed (wire) buffer
    long snap_off[SVDF_MAX_DELTA_RANGES];      // position of the range's first float in the snapshot (full layout)
    int n;
};

#endif
