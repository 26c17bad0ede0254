/* gpc_oracle_train.c -- see gpc_oracle_train.h.  TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED. */
#include "gpc_oracle_train.h"

#include <string.h>


#define TRIP (3 * GPC_ORACLE_PATCH)

/* Feature::getDecisions (Feature.hpp:101-109): (int)x(i) - (int)x(j) < tau */
static int decision(const uint8_t* patch, const gpc_oracle_split* p) {
  return ((int)patch[p->i] - (int)patch[p->j]) < p->tau;
}

/* the three code words of one triplet over params[0 .. count) (Fern.hpp:222-236, 275-287):
 * 64-bit words shifted left once per level, so only the last 64 levels survive */
static void codes(const uint8_t* t, const gpc_oracle_split* params, int count, uint64_t* ref,
                  uint64_t* pos, uint64_t* neg) {
  uint64_t r = 0, p = 0, q = 0;
  for (int i = 0; i < count; ++i) {
    r <<= 1;
    p <<= 1;
    q <<= 1;
    if (decision(t, &params[i])) r++;
    if (decision(t + GPC_ORACLE_PATCH, &params[i])) p++;
    if (decision(t + 2 * GPC_ORACLE_PATCH, &params[i])) q++;
  }
  *ref = r;
  *pos = p;
  *neg = q;
}

void gpc_oracle_eval_split(const uint8_t* triplets, const uint8_t* marks, int n,
                           const gpc_oracle_split* params, int score_until_level, double w1,
                           gpc_oracle_split_stats* s) {
  memset(s, 0, sizeof *s);
  for (int k = 0; k < n; ++k) {
    uint64_t ref, pos, neg;
    codes(triplets + (long)k * TRIP, params, score_until_level + 1, &ref, &pos, &neg);
    /* samples that were true positives before are ignored (Fern.hpp:239) */
    if (!((marks[k] & 1) && (marks[k] & 2))) {
      s->tot++;
      if (ref == pos) {
        if (ref != neg) s->tp++;
        else s->fn++;
      } else {
        if (ref != neg) s->fn++;
        else s->fp++;
      }
    }
  }
  /* Fern.hpp:255-261 */
  const double w2 = 1. - w1;
  s->prec = ((s->tp + s->fp) == 0) ? 0. : (double)s->tp / (s->tp + s->fp);
  s->rec = ((s->tp + s->fn) == 0) ? 0. : (double)s->tp / (s->tp + s->fn);
  s->hmean = (s->prec + s->rec == 0.) ? 0. : s->prec * s->rec / ((1. - w2) * s->prec + w2 * s->rec);
  s->convcomb = (1. - w2) * s->prec + w2 * s->rec;
}

void gpc_oracle_mark_split_samples(const uint8_t* triplets, uint8_t* marks, int n,
                                   const gpc_oracle_split* params, int num_params) {
  for (int k = 0; k < n; ++k) {
    uint64_t ref, pos, neg;
    codes(triplets + (long)k * TRIP, params, num_params, &ref, &pos, &neg);
    if (ref == pos) marks[k] |= 1;
    if (ref != neg) marks[k] |= 2;
  }
}

void gpc_oracle_train_fern(const uint8_t* triplets, uint8_t* marks, int n, int max_depth,
                           const gpc_oracle_split* cand, int num_resamples, int taulo, int tauhi,
                           int only_score_non_split, double w1, gpc_oracle_split* fernparams,
                           gpc_oracle_split_stats* level_stats) {
  gpc_oracle_split_stats stats;
  memset(&stats, 0, sizeof stats);
  float max_score = 0.f;
  gpc_oracle_split best = {0, 0, 0};                                 /* SplitParams_t bestParams; (:316) */
  memset(fernparams, 0, sizeof(gpc_oracle_split) * (size_t)max_depth); /* fernparams.resize(maxDepth) (:318) */
  if (only_score_non_split) memset(marks, 0, (size_t)n);              /* resetMarkOnSamples (:333-334) */
  for (int level = 0; level < max_depth; ++level) {
    max_score = 0.f;
    for (int k = 0; k < num_resamples; ++k) {
      fernparams[level] = cand[level * num_resamples + k];            /* sampleHyperplane (:339) */
      for (int tau = taulo; tau < tauhi; ++tau) {
        fernparams[level].tau = tau;
        gpc_oracle_eval_split(triplets, marks, n, fernparams, level, w1, &stats);
        if (stats.hmean > max_score) {                                /* double against float (:346) */
          best = fernparams[level];
          max_score = (float)stats.hmean;
        }
      }
    }
    fernparams[level] = best; /* also when nothing scored above 0: the previous level's best (:352) */
    if (only_score_non_split) gpc_oracle_mark_split_samples(triplets, marks, n, fernparams, level); /* (:355-356) */
    level_stats[level] = stats; /* what train() prints: the LAST evaluated candidate's stats (:357-369) */
  }
}
