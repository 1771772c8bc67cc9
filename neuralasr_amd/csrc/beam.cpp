// beam.cpp — tf.nn.ctc_beam_search_decoder (networks/tfnetwork.py:61-64: defaults beam_width = 100,
// top_paths = 1, merge_repeated = True) as host code: prefix beam search over per-frame log-softmax, a beam
// entry carrying (p_blank, p_label, p_total) of its prefix, at most `beam_width` live entries per frame
// (SURVEY.md Appendix A.6).  merge_repeated additionally merges adjacent equal labels of the OUTPUT sequence,
// even when a blank separated them (TF's documented quirk).  One thread per utterance.
// Parity status: restated from TF 1.x's documented algorithm; unpinned (no TF here), cross-checked against an
// independent Python restatement (oracle/nasr_oracle.py: ctc_beam_search) in tests/test_beam.py.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <deque>
#include <limits>
#include <memory>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/nasr.h"

namespace {

constexpr float kLogZero = -std::numeric_limits<float>::infinity();

inline float lse(float a, float b) {
  if (a == kLogZero) return b;
  if (b == kLogZero) return a;
  return a > b ? a + std::log1p(std::exp(b - a)) : b + std::log1p(std::exp(a - b));
}

struct Prob {
  float total = kLogZero, blank = kLogZero, label = kLogZero;
  void reset() { total = blank = label = kLogZero; }
};

// One prefix of the beam tree.  Children sit in a flat array indexed by label, allocated from the decoder's arena on the
// first extension (a hash map per entry and a heap allocation per child made the decoder 5-10x slower than this).
struct Entry {
  Entry* parent = nullptr;
  int label = -1;
  Prob oldp, newp;
  Entry** kids = nullptr;
  bool active() const { return newp.total != kLogZero; }
};

struct Arena {
  int C;
  std::deque<Entry> entries;
  std::deque<std::vector<Entry*>> kid_arrays;
  explicit Arena(int c) : C(c) {}
  Entry* child(Entry* p, int lab) {
    if (!p->kids) {
      kid_arrays.emplace_back((size_t)C, nullptr);
      p->kids = kid_arrays.back().data();
    }
    Entry*& slot = p->kids[lab];
    if (!slot) {
      entries.emplace_back();
      slot = &entries.back();
      slot->parent = p;
      slot->label = lab;
    }
    return slot;
  }
};

// bounded "top N by newp.total" container: a min-heap on the total, so the bottom is the front and a push that evicts it
// costs O(log N) (the entries' totals do not change while they sit in it)
struct Leaves {
  size_t cap;
  std::vector<Entry*> v;
  explicit Leaves(size_t c) : cap(c) {}
  static bool better(const Entry* a, const Entry* b) { return a->newp.total > b->newp.total; }
  Entry* bottom() const { return v.front(); }
  void push(Entry* e) {
    if (v.size() < cap) {
      v.push_back(e);
      std::push_heap(v.begin(), v.end(), better);
      return;
    }
    if (e->newp.total > v.front()->newp.total) {
      std::pop_heap(v.begin(), v.end(), better);
      v.back() = e;
      std::push_heap(v.begin(), v.end(), better);
    }
  }
};

void decode_one(const float* logits, size_t frame_stride, int T, int C, int beam_width, bool merge_repeated,
                int32_t* ids_out, int32_t* len_out, float* logp_out) {
  const int blank = C - 1;
  Arena arena(C);
  Entry root;
  root.newp.total = 0.f;
  root.newp.blank = 0.f;
  Leaves leaves((size_t)beam_width);
  leaves.v.push_back(&root);
  std::vector<float> lp((size_t)C);
  for (int t = 0; t < T; ++t) {
    const float* x = logits + (size_t)t * frame_stride;
    float m = x[0];
    for (int c = 1; c < C; ++c) m = std::max(m, x[c]);
    double s = 0.0;
    for (int c = 0; c < C; ++c) s += std::exp((double)x[c] - m);
    const float norm = m + (float)std::log(s);
    for (int c = 0; c < C; ++c) lp[c] = x[c] - norm;

    std::vector<Entry*> branches = leaves.v;
    std::sort(branches.begin(), branches.end(), Leaves::better);
    leaves.v.clear();
    for (Entry* b : branches) b->oldp = b->newp;
    for (Entry* b : branches) {   // extensions that keep the prefix
      if (b->parent) {
        if (b->parent->active()) {
          const float prev = (b->label == b->parent->label) ? b->parent->oldp.blank : b->parent->oldp.total;
          b->newp.label = lse(b->newp.label, prev);
        }
        b->newp.label += lp[b->label];
      }
      b->newp.blank = b->oldp.total + lp[blank];
      b->newp.total = lse(b->newp.blank, b->newp.label);
      leaves.push(b);
    }
    for (Entry* b : branches) {   // grow new leaves
      auto candidate = [&](const Prob& p) {
        return p.total > kLogZero && (leaves.v.size() < leaves.cap || p.total > leaves.bottom()->newp.total);
      };
      if (!candidate(b->oldp)) continue;
      for (int lab = 0; lab < C; ++lab) {
        if (lab == blank) continue;
        // the extension's score needs nothing of the child: test it against the beam's bottom BEFORE the child is looked up
        // or created (an inactive child that fails the test is left as it is: nothing reads it until it becomes a leaf)
        const float prev = (lab == b->label) ? b->oldp.blank : b->oldp.total;
        Prob np;
        np.blank = kLogZero;
        np.label = lp[lab] + prev;
        np.total = np.label;
        if (!candidate(np)) {
          // (TF's decoder clears BOTH probabilities of a child that fails here.  That matters for one kind of child: one that
          //  was a leaf of this frame, was evicted above and is still to come in this loop - cleared, it grows no children)
          Entry* e = b->kids ? b->kids[lab] : nullptr;
          if (e && !e->active()) e->oldp.reset();
          continue;
        }
        Entry* c = arena.child(b, lab);
        if (c->active()) continue;
        c->newp = np;
        if (leaves.v.size() == leaves.cap) leaves.bottom()->newp.reset();
        leaves.push(c);
      }
    }
  }
  Entry* best = *std::max_element(leaves.v.begin(), leaves.v.end(),
                                  [](const Entry* a, const Entry* b) { return a->newp.total < b->newp.total; });
  std::vector<int> seq;
  int prev_label = -1;
  for (const Entry* c = best; c->parent; c = c->parent) {
    if (!merge_repeated || c->label != prev_label) seq.push_back(c->label);
    prev_label = c->label;
  }
  std::reverse(seq.begin(), seq.end());
  *len_out = (int32_t)seq.size();
  for (size_t i = 0; i < seq.size(); ++i) ids_out[i] = seq[i];
  if (logp_out) *logp_out = best->newp.total;
}

}  // namespace

extern "C" int nasr_ctc_beam_search(const float* logits, const int32_t* seq_len, int B, int Tp, int C, int beam_width,
                                    int merge_repeated, int32_t* ids_out, int32_t* lens_out, float* logp_out) {
  if (!logits || !seq_len || !ids_out || !lens_out || B < 1 || Tp < 1 || C < 2 || beam_width < 1) return NASR_ERR_ARG;
  for (int b = 0; b < B; ++b)
    if (seq_len[b] < 0 || seq_len[b] > Tp) return NASR_ERR_ARG;
  const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
  const int nthr = (int)std::min<unsigned>(hw, (unsigned)B);
  std::vector<std::thread> pool;
  for (int w = 0; w < nthr; ++w)
    pool.emplace_back([=]() {
      for (int b = w; b < B; b += nthr)
        decode_one(logits + (size_t)b * C, (size_t)B * C, seq_len[b], C, beam_width, merge_repeated != 0,
                   ids_out + (size_t)b * Tp, lens_out + b, logp_out ? logp_out + b : nullptr);
    });
  for (auto& t : pool) t.join();
  return NASR_OK;
}
