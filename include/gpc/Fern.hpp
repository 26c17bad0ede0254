// gpc/Fern.hpp -- MI355X-native mirror of the reference's fern trainer (lib/gpc/Fern.hpp).
//
// Same names, signatures and printed table as the reference; evalSplit, markSplitSamples and the
// level / resample / tau loops of train() run on the GPU (gpc_hip_train_*, include/gpc_hip.h) over a
// device-resident copy of the triplets.  Hyperplanes are drawn on the host with the reference's
// generator, in the reference's order (one sampleHyperplane per level and resample).
#ifndef _GPC_fern
#define _GPC_fern
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <string>
#include <vector>

#include "gpc/Feature.hpp"
#include "gpc_hip.h"

using namespace std;
namespace gpc {
namespace training {

// Fern.hpp:52-68
struct splitStats {
  double prec = 0.;
  double rec = 0.;
  double hmean = 0.;     // weighted harmonic mean of precision and recall
  double convcomb = 0.;  // convex combination of precision and recall
  int tp = 0;
  int fp = 0;
  int fn = 0;
  int tot = 0;
};

// Fern.hpp:70-89
struct OptimizerSettings {
  double w1_;
  int numResamples_;
  int taulo_;
  int tauhi_;
  bool onlyScoreNonSplitSamples_;
  OptimizerSettings(int taulo, int tauhi, int numResamples, bool onlyScoreNonSplitSamples, double w1)
      : w1_(w1), numResamples_(numResamples), taulo_(taulo), tauhi_(tauhi),
        onlyScoreNonSplitSamples_(onlyScoreNonSplitSamples) {}
  OptimizerSettings() {}
};
// Fern.hpp:90-118
struct TauOptimizerSettings : public OptimizerSettings {
  TauOptimizerSettings(int taulo, int tauhi, int numResamples, bool onlyScoreNonSplitSamples, double w1)
      : OptimizerSettings(taulo, tauhi, numResamples, onlyScoreNonSplitSamples, w1) {}
  TauOptimizerSettings() : OptimizerSettings() {}
  TauOptimizerSettings& builder(void) { return *this; }
  TauOptimizerSettings& w1(double w1) { this->w1_ = w1; return *this; }
  TauOptimizerSettings& numResamples(double numResamples) { this->numResamples_ = numResamples; return *this; }
  TauOptimizerSettings& taulo(double taulo) { this->taulo_ = taulo; return *this; }
  TauOptimizerSettings& tauhi(int tauhi) { this->tauhi_ = tauhi; return *this; }
  TauOptimizerSettings& onlyScoreNonSplitSamples(bool v) { this->onlyScoreNonSplitSamples_ = v; return *this; }
};
// Fern.hpp:119-138 (taulo = 0, tauhi = 1: the intercept stays 0)
struct ZeroOptimizerSettings : public OptimizerSettings {
  ZeroOptimizerSettings(int numResamples, bool onlyScoreNonSplitSamples, double w1)
      : OptimizerSettings(0, 1, numResamples, onlyScoreNonSplitSamples, w1) {}
  ZeroOptimizerSettings() : OptimizerSettings() {}
  ZeroOptimizerSettings& builder(void) { return *this; }
  ZeroOptimizerSettings& w1(double w1) { this->w1_ = w1; return *this; }
  ZeroOptimizerSettings& numResamples(double numResamples) { this->numResamples_ = numResamples; return *this; }
  ZeroOptimizerSettings& onlyScoreNonSplitSamples(bool v) { this->onlyScoreNonSplitSamples_ = v; return *this; }
};
// Fern.hpp:151-166
inline OptimizerSettings TauOptimizer(int taulo, int tauhi, int numResamples, bool onlyScoreNonSplitSamples, double w1) {
  return OptimizerSettings(taulo, tauhi, numResamples, onlyScoreNonSplitSamples, w1);
}
inline OptimizerSettings ZeroOptimizer(int numResamples, bool onlyScoreNonSplitSamples, double w1) {
  return OptimizerSettings(0, 1, numResamples, onlyScoreNonSplitSamples, w1);
}
// Fern.hpp:167-172
struct FernSettings {
  const int maxDepth;
  const int scale;
  FernSettings(int maxDepth, int scale) : maxDepth(maxDepth), scale(scale) {}
};

namespace detail {
// A vector of triplets as one device-resident training set of the calling thread's context.
class DeviceTriplets {
 public:
  template <class Triplet>
  explicit DeviceTriplets(std::vector<Triplet>& data) : ctx_(gpc::inference::detail::holder().ctx) {
    const size_t n = data.size();
    std::vector<uint8_t> host(n * 3 * 729), marks(n);
    for (size_t k = 0; k < n; ++k) {
      std::memcpy(&host[(k * 3 + 0) * 729], data[k].ref.feature.data(), 729);
      std::memcpy(&host[(k * 3 + 1) * 729], data[k].pos.feature.data(), 729);
      std::memcpy(&host[(k * 3 + 2) * 729], data[k].neg.feature.data(), 729);
      marks[k] = (uint8_t)((data[k].pos.split ? 1 : 0) | (data[k].neg.split ? 2 : 0));
    }
    check(gpc_hip_train_set_create(ctx_, host.data(), (int)n, &set_), "gpc_hip_train_set_create");
    check(gpc_hip_train_set_marks(ctx_, set_, marks.data(), nullptr), "gpc_hip_train_set_marks");
  }
  ~DeviceTriplets() {
    if (set_) gpc_hip_train_set_destroy(ctx_, set_);
  }
  DeviceTriplets(const DeviceTriplets&) = delete;
  DeviceTriplets& operator=(const DeviceTriplets&) = delete;
  // the marks the device holds -> triplet.pos.split / triplet.neg.split
  template <class Triplet>
  void marksTo(std::vector<Triplet>& data) {
    std::vector<uint8_t> marks(data.size());
    check(gpc_hip_train_set_marks(ctx_, set_, nullptr, marks.data()), "gpc_hip_train_set_marks");
    for (size_t k = 0; k < data.size(); ++k) {
      data[k].pos.split = (marks[k] & 1) != 0;
      data[k].neg.split = (marks[k] & 2) != 0;
    }
  }
  gpc_hip_ctx* ctx() const { return ctx_; }
  gpc_hip_train_set* set() const { return set_; }
  void check(int st, const char* what) const {
    if (st != GPC_OK) {
      gpc::inference::detail::fail(st, ctx_, what);
      std::abort();  // a training run cannot go on without its device set (matching calls return empty results instead)
    }
  }

 private:
  gpc_hip_ctx* ctx_;
  gpc_hip_train_set* set_ = nullptr;
};
}  // namespace detail

// Fern.hpp:178-394
class Fern {
 private:
  typedef gpc::training::Feature Feature_t;
  typedef Feature_t::GPCPatchTriplet GPCTriplet_t;
  typedef Feature_t::params SplitParams_t;
  Feature_t Feature;
  std::vector<SplitParams_t> fernparams;
  FernSettings fernsettings;

  static gpc_split toSplit(const SplitParams_t& p) { return gpc_split{p.i, p.j, p.tau}; }
  static SplitParams_t fromSplit(const gpc_split& s) {
    SplitParams_t p;  // i = (ix+13) + 27*(iy+13) at every scale (Feature.hpp:141-142, 155-156, 170-171)
    p.i = s.i;
    p.j = s.j;
    p.tau = s.tau;
    p.ix = s.i % 27 - 13;
    p.iy = s.i / 27 - 13;
    p.jx = s.j % 27 - 13;
    p.jy = s.j / 27 - 13;
    return p;
  }
  static void toStats(const gpc_split_stats& g, splitStats& s) {
    s.prec = g.prec;
    s.rec = g.rec;
    s.hmean = g.hmean;
    s.convcomb = g.convcomb;
    s.tp = g.tp;
    s.fp = g.fp;
    s.fn = g.fn;
    s.tot = g.tot;
  }

 public:
  Fern(FernSettings fernsettings) : fernsettings(fernsettings) {}
  // extension: reproducible hyperplane sampling (tests)
  void seed(unsigned s) { Feature.seed(s); }

  // Fern.hpp:209-262.  Uploads `data`; callers that score many parameter sets on the same data keep
  // a detail::DeviceTriplets and use the C ABI directly (train() below does).
  void evalSplit(std::vector<GPCTriplet_t>& data, std::vector<SplitParams_t>& params, FernSettings fernsetting,
                 OptimizerSettings optsetting, int scoreUntilLevel, splitStats& s) {
    (void)fernsetting;
    detail::DeviceTriplets dev(data);
    std::vector<gpc_split> p(scoreUntilLevel + 1);
    for (int l = 0; l <= scoreUntilLevel; ++l) p[l] = toSplit(params[l]);
    gpc_split_stats g;
    dev.check(gpc_hip_train_eval_split(dev.ctx(), dev.set(), p.data(), scoreUntilLevel, optsetting.w1_, &g),
              "gpc_hip_train_eval_split");
    toStats(g, s);
  }
  // Fern.hpp:271-291
  void markSplitSamples(std::vector<GPCTriplet_t>& data, std::vector<SplitParams_t>& params, int numParams) {
    detail::DeviceTriplets dev(data);
    std::vector<gpc_split> p(numParams > 0 ? numParams : 1);
    for (int l = 0; l < numParams; ++l) p[l] = toSplit(params[l]);
    dev.check(gpc_hip_train_mark_split_samples(dev.ctx(), dev.set(), p.data(), numParams),
              "gpc_hip_train_mark_split_samples");
    dev.marksTo(data);
  }
  // Fern.hpp:299-304
  void resetMarkOnSamples(std::vector<GPCTriplet_t>& data) {
    for (auto& triplet : data) {
      triplet.pos.split = false;
      triplet.neg.split = false;
    }
  }

  // Fern.hpp:312-372
  void train(std::vector<GPCTriplet_t>& trainingSamples, OptimizerSettings optsetting) {
    fernparams.resize(fernsettings.maxDepth);
    cout << setw(7) << "Level" << setw(10) << "Prec" << setw(10) << "Rec" << setw(10) << "Har" << setw(8) << "Tot"
         << setw(8) << "TP" << setw(8) << "FP" << setw(8) << "FN" << setw(6) << "scale" << setw(5) << "tau"
         << setw(5) << "i" << setw(5) << "j" << endl;
    // the hyperplanes of every level, drawn in the reference's order (:339; scoring does not touch the generator)
    const int depth = fernsettings.maxDepth, nres = optsetting.numResamples_;
    std::vector<gpc_split> cand((size_t)depth * (nres > 0 ? nres : 0));
    for (int level = 0; level < depth; level++)
      for (int k = 0; k < nres; k++) {
        Feature.sampleHyperplane(fernsettings.scale, fernparams[level]);
        cand[(size_t)level * nres + k] = toSplit(fernparams[level]);
      }
    if (const char* dump = std::getenv("GPC_TRAIN_DUMP_CANDIDATES")) {  // diagnostics / tests: what was drawn
      std::ofstream f(dump, std::ios::app);
      for (auto& c : cand) f << c.i << " " << c.j << "\n";
    }
    detail::DeviceTriplets dev(trainingSamples);
    std::vector<gpc_split> chosen(depth);
    std::vector<gpc_split_stats> stats(depth);
    dev.check(gpc_hip_train_fern(dev.ctx(), dev.set(), depth, cand.data(), nres, optsetting.taulo_, optsetting.tauhi_,
                                 optsetting.onlyScoreNonSplitSamples_ ? 1 : 0, optsetting.w1_, chosen.data(),
                                 stats.data()),
              "gpc_hip_train_fern");
    if (optsetting.onlyScoreNonSplitSamples_) dev.marksTo(trainingSamples);  // resetMarkOnSamples + markSplitSamples
    for (int level = 0; level < depth; level++) {
      fernparams[level] = fromSplit(chosen[level]);
      cout << setw(7) << level << setw(10) << stats[level].prec << setw(10) << stats[level].rec << setw(10)
           << stats[level].hmean << setw(8) << stats[level].tot << setw(8) << stats[level].tp << setw(8)
           << stats[level].fp << setw(8) << stats[level].fn << setw(6) << fernsettings.scale << setw(5)
           << fernparams[level].tau << setw(5) << fernparams[level].i << setw(5) << fernparams[level].j << endl;
    }
  }

  std::vector<SplitParams_t> getParameters() { return fernparams; }
  int getScale() { return fernsettings.scale; }
};  // Fern

// Fern.hpp:405-414
inline std::vector<Fern> FernFactory(int num_S, int num_M, int num_L, int maxDepth) {
  std::vector<Fern> ferns;
  for (int i = 0; i < num_S; i++) ferns.push_back(Fern(FernSettings(maxDepth, 2)));
  for (int i = 0; i < num_M; i++) ferns.push_back(Fern(FernSettings(maxDepth, 1)));
  for (int i = 0; i < num_L; i++) ferns.push_back(Fern(FernSettings(maxDepth, 0)));
  return ferns;
}
}  // namespace training
}  // namespace gpc
#endif
