/* gpc_oracle_train.h -- CPU restatement of the reference's fern TRAINING scoring loop.
 *
 * TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench tools' CPU leg); never shipped.
 *
 * PARITY UNPINNED: the reference has no tests or golden vectors for training, and Fern.hpp /
 * Feature.hpp cannot be compiled here (they need Eigen, which this image lacks), so this
 * restatement is checked against nothing but itself and a plain numpy model in tests/.
 * Every function cites the reference lines it follows.
 */
#ifndef GPC_ORACLE_TRAIN_H
#define GPC_ORACLE_TRAIN_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define GPC_ORACLE_PATCH 729 /* 27 x 27 bytes per patch, file order of storeAllTriplets (Feature.hpp:247-256) */

/* the fields of Feature::params (Feature.hpp:84-89) that scoring reads: linear pixel indices and intercept */
typedef struct {
  int32_t i, j, tau;
} gpc_oracle_split;

/* splitStats (Fern.hpp:52-68) */
typedef struct {
  double prec, rec, hmean, convcomb;
  int32_t tp, fp, fn, tot;
} gpc_oracle_split_stats;

/* triplets: n * 3 * 729 bytes, patches ref, pos, neg of each triplet one after the other.
 * marks:    n bytes, bit 0 = pos.split, bit 1 = neg.split (GPCDescriptor::split, Feature.hpp:65). */

/* Fern::evalSplit (Fern.hpp:209-262) */
void gpc_oracle_eval_split(const uint8_t* triplets, const uint8_t* marks, int n,
                           const gpc_oracle_split* params, int score_until_level, double w1,
                           gpc_oracle_split_stats* s);
/* Fern::markSplitSamples (Fern.hpp:271-291) */
void gpc_oracle_mark_split_samples(const uint8_t* triplets, uint8_t* marks, int n,
                                   const gpc_oracle_split* params, int num_params);
/* Fern::train (Fern.hpp:312-372) with the hyperplane samples injected: cand[level * num_resamples + k]
 * is what Feature::sampleHyperplane would have drawn (i, j; its tau is overwritten by the tau loop).
 * fernparams: max_depth entries out; level_stats: max_depth entries = the stats train() prints. */
void gpc_oracle_train_fern(const uint8_t* triplets, uint8_t* marks, int n, int max_depth,
                           const gpc_oracle_split* cand, int num_resamples, int taulo, int tauhi,
                           int only_score_non_split, double w1, gpc_oracle_split* fernparams,
                           gpc_oracle_split_stats* level_stats);

#ifdef __cplusplus
}
#endif
#endif
