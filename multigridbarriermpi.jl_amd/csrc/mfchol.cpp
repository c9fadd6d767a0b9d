#include "mfchol.hpp"

#include <sched.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

namespace mgb {

namespace {

int hw_threads() {
  static int n = [] {
    const char* e = getenv("MGB_NUM_THREADS");
    int v = 0;
    if (e) {
      v = atoi(e);
    } else {
      // cores this process may actually run on (cgroup / affinity), not the machine's core count
      cpu_set_t set;
      CPU_ZERO(&set);
      if (sched_getaffinity(0, sizeof(set), &set) == 0) v = CPU_COUNT(&set);
      if (v <= 0) v = (int)std::thread::hardware_concurrency();
      if (v > 16) v = 16;   // the factorisation stops scaling beyond that; leave room for sibling ranks
    }
    if (v < 1) v = 1;
    if (v > 64) v = 64;
    return v;
  }();
  return n;
}

// Persistent worker pool: parallel_for hands out indices through an atomic counter; workers sleep on a
// condition variable between calls (thread creation per call cost more than the small fronts).
class Pool {
 public:
  static Pool& get() {
    static Pool p(hw_threads());
    return p;
  }
  int size() const { return (int)workers_.size() + 1; }
  template <class F>
  void run(int n, F&& fn) {
    if (n <= 0) return;
    if (workers_.empty() || n == 1 || busy_.exchange(true)) {   // nested / concurrent use: run inline
      for (int i = 0; i < n; ++i) fn(i);
      return;
    }
    std::function<void(int)> f = fn;
    {
      std::lock_guard<std::mutex> lk(m_);
      fn_ = &f;
      n_ = n;
      next_.store(0);
      pending_ = (int)workers_.size();
      ++epoch_;
    }
    cv_.notify_all();
    for (;;) {
      const int i = next_.fetch_add(1);
      if (i >= n) break;
      f(i);
    }
    std::unique_lock<std::mutex> lk(m_);
    done_.wait(lk, [&] { return pending_ == 0; });
    fn_ = nullptr;
    busy_.store(false);
  }
  ~Pool() {
    {
      std::lock_guard<std::mutex> lk(m_);
      stop_ = true;
      ++epoch_;
    }
    cv_.notify_all();
    for (auto& t : workers_) t.join();
  }

 private:
  explicit Pool(int nthreads) {
    for (int t = 1; t < nthreads; ++t) workers_.emplace_back([this] { loop(); });
  }
  void loop() {
    unsigned long seen = 0;
    for (;;) {
      std::function<void(int)>* f;
      int n;
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return epoch_ != seen; });
        seen = epoch_;
        if (stop_) return;
        f = fn_;
        n = n_;
      }
      for (;;) {
        const int i = next_.fetch_add(1);
        if (i >= n) break;
        (*f)(i);
      }
      std::lock_guard<std::mutex> lk(m_);
      if (--pending_ == 0) done_.notify_one();
    }
  }
  std::vector<std::thread> workers_;
  std::mutex m_;
  std::condition_variable cv_, done_;
  std::function<void(int)>* fn_ = nullptr;
  std::atomic<int> next_{0};
  std::atomic<bool> busy_{false};
  int n_ = 0, pending_ = 0;
  unsigned long epoch_ = 0;
  bool stop_ = false;
};

template <class F>
void parallel_for(int n, int nthreads, F&& fn) {
  if (nthreads <= 1 || n <= 1) {
    for (int i = 0; i < n; ++i) fn(i);
    return;
  }
  Pool::get().run(n, fn);
}

// C(m x m, lower, ld) -= P(m x kw, ld) * P^T for columns [j0, j1) ; C and P column-major.
inline void syrk_cols(int m, int kw, const double* P, int ld, double* C, int j0, int j1) {
  int j = j0;
  for (; j + 4 <= j1; j += 4) {
    double* c0 = C + (size_t)j * ld;
    double* c1 = c0 + ld;
    double* c2 = c1 + ld;
    double* c3 = c2 + ld;
    // 4x4 triangle on the diagonal
    for (int k = 0; k < kw; ++k) {
      const double* p = P + (size_t)k * ld;
      const double a0 = p[j], a1 = p[j + 1], a2 = p[j + 2], a3 = p[j + 3];
      c0[j] -= a0 * a0;
      c0[j + 1] -= a1 * a0;
      c0[j + 2] -= a2 * a0;
      c0[j + 3] -= a3 * a0;
      c1[j + 1] -= a1 * a1;
      c1[j + 2] -= a2 * a1;
      c1[j + 3] -= a3 * a1;
      c2[j + 2] -= a2 * a2;
      c2[j + 3] -= a3 * a2;
      c3[j + 3] -= a3 * a3;
    }
    // rows below, tiled so the four C columns stay in L1 while the panel streams
    for (int i0 = j + 4; i0 < m; i0 += 256) {
      const int i1 = std::min(m, i0 + 256);
      for (int k = 0; k < kw; ++k) {
        const double* p = P + (size_t)k * ld;
        const double a0 = p[j], a1 = p[j + 1], a2 = p[j + 2], a3 = p[j + 3];
        for (int i = i0; i < i1; ++i) {
          const double v = p[i];
          c0[i] -= v * a0;
          c1[i] -= v * a1;
          c2[i] -= v * a2;
          c3[i] -= v * a3;
        }
      }
    }
  }
  for (; j < j1; ++j) {
    double* c = C + (size_t)j * ld;
    for (int k = 0; k < kw; ++k) {
      const double* p = P + (size_t)k * ld;
      const double a = p[j];
      for (int i = j; i < m; ++i) c[i] -= p[i] * a;
    }
  }
}

// Partial Cholesky of the first ns pivots of the nf x nf column-major lower front F.
bool partial_chol(double* F, int nf, int ns, int nthreads) {
  const int NB = 48;
  for (int kb = 0; kb < ns; kb += NB) {
    const int kw = std::min(NB, ns - kb);
    for (int k = kb; k < kb + kw; ++k) {
      double* ck = F + (size_t)k * nf;
      double d = ck[k];
      if (!(d > 0.0) || !std::isfinite(d)) return false;
      d = std::sqrt(d);
      ck[k] = d;
      const double inv = 1.0 / d;
      for (int i = k + 1; i < nf; ++i) ck[i] *= inv;
      for (int j = k + 1; j < kb + kw; ++j) {
        double* cj = F + (size_t)j * nf;
        const double a = ck[j];
        for (int i = j; i < nf; ++i) cj[i] -= ck[i] * a;
      }
    }
    const int r0 = kb + kw;  // trailing block starts here
    const int m = nf - r0;
    if (m <= 0) continue;
    const double* P = F + (size_t)kb * nf + r0;     // rows r0.., columns kb..kb+kw
    double* C = F + (size_t)r0 * nf + r0;
    if (nthreads > 1 && (double)m * m * kw > 4e6) {
      // column strips with roughly equal triangle area
      const int nchunk = nthreads * 4;
      std::vector<int> cut(nchunk + 1);
      for (int c = 0; c <= nchunk; ++c) {
        double frac = 1.0 - std::sqrt(1.0 - (double)c / nchunk);
        cut[c] = std::min(m, (int)(frac * m) / 4 * 4);
      }
      cut[nchunk] = m;
      parallel_for(nchunk, nthreads, [&](int c) {
        if (cut[c + 1] > cut[c]) syrk_cols(m, kw, P, nf, C, cut[c], cut[c + 1]);
      });
    } else {
      syrk_cols(m, kw, P, nf, C, 0, m);
    }
  }
  return true;
}

}  // namespace

int MfChol::threads() { return hw_threads(); }

// Nested dissection of dofs[lo, hi) into `sub` (nodes in postorder, indices local to `sub`; the root is the last node).
// The two halves touch disjoint dofs (and disjoint entries of `label`), so the top `par_depth` levels of the recursion
// build their halves concurrently into private subtrees, which are then appended left, right, parent: the same postorder
// the sequential recursion produces.
int MfChol::build(std::vector<int>& dofs, int lo, int hi, const Csr& A, const double* coords, int dim, int leaf,
                  std::vector<int>& label, std::atomic<int>& next_label, Subtree& sub, int par_depth,
                  const unsigned long long* mask, int rlo, int rhi) {
  std::vector<Node>& nodes_ = sub.nodes;      // shadows the member: this recursion only ever touches `sub`
  std::vector<std::vector<int>>& own = sub.own;
  // (`label` entries of dofs outside [lo, hi) may be rewritten concurrently by a sibling subtree: relaxed atomics; their
  // tags are unique per split, so they never compare equal to this split's tags)
  const int cnt = hi - lo;
  auto make_leaf = [&] {
    nodes_.emplace_back();
    own.emplace_back(dofs.begin() + lo, dofs.begin() + hi);
    return (int)nodes_.size() - 1;
  };
  std::vector<int> Ap, Bv, S;
  const unsigned long long* cmask = nullptr;      // rank masks for the children (null: geometric dissection below here)
  int rmid = rlo;
  bool guided = false;
  if (mask && rhi - rlo > 1) {
    // sharded job, top of the tree: split by who touches the unknown (see analyze() in the header)
    rmid = (rlo + rhi) / 2;
    auto bits = [](int a, int b) { return (b >= 64 ? ~0ull : ((1ull << b) - 1)) & ~((1ull << a) - 1); };
    const unsigned long long low = bits(rlo, rmid), high = bits(rmid, rhi);
    for (int i = lo; i < hi; ++i) {
      const int v = dofs[i];
      ((mask[v] & high) == 0 ? Ap : (mask[v] & low) == 0 ? Bv : S).push_back(v);
    }
    if (Ap.empty() || Bv.empty()) {
      align_failed_.store(true);      // a rank without interior unknowns: this subtree falls back to the geometry
      Ap.clear(); Bv.clear(); S.clear();
    } else {
      guided = true;
      cmask = mask;
    }
  }
  if (!guided) {
  if (cnt <= leaf) return make_leaf();
  // widest axis
  int axis = 0;
  double best = -1;
  for (int d = 0; d < dim; ++d) {
    double mn = 1e300, mx = -1e300;
    for (int i = lo; i < hi; ++i) {
      double c = coords[(size_t)dofs[i] * dim + d];
      mn = std::min(mn, c);
      mx = std::max(mx, c);
    }
    if (mx - mn > best) {
      best = mx - mn;
      axis = d;
    }
  }
  // Sort along the axis and cut BETWEEN two distinct coordinate values (a straight mesh line: co-located dofs of
  // different state variables stay together and the cut does not wander through a column of tied coordinates),
  // trying the few such cuts nearest to the median and both one-sided separators (A-side dofs adjacent to B, or
  // B-side dofs adjacent to A); the smallest separator wins.  Separator sizes set the number of 32-pivot
  // panels on the critical path of the device factorisation (gpuchol.hip), so this is worth a few extra passes.
  auto key = [&](int v) { return coords[(size_t)v * dim + axis]; };
  std::sort(dofs.begin() + lo, dofs.begin() + hi, [&](int a, int b) {
    const double ca = key(a), cb = key(b);
    return ca != cb ? ca < cb : a < b;
  });
  std::vector<int> cand;
  {
    const int mid = lo + cnt / 2, wlo = lo + (int)(0.3 * cnt), whi = lo + (int)(0.7 * cnt);
    for (int step = 0; (int)cand.size() < 4 && (mid - step > wlo || mid + step < whi); ++step) {
      for (int sgn = -1; sgn <= 1; sgn += 2) {
        const int bpos = mid + sgn * step;
        if (step == 0 && sgn == 1) continue;
        if (bpos <= std::max(lo, wlo) || bpos >= std::min(hi, whi)) continue;
        if (key(dofs[bpos - 1]) != key(dofs[bpos])) cand.push_back(bpos);
      }
    }
    if (cand.empty()) cand.push_back(mid);
  }
  const int tagA = next_label.fetch_add(2), tagB = tagA + 1;
  int midp = cand[0], best_size = -1;
  bool sep_in_A = true;
  for (int bpos : cand) {
    for (int i = lo; i < bpos; ++i) __atomic_store_n(&label[dofs[i]], tagA, __ATOMIC_RELAXED);
    for (int i = bpos; i < hi; ++i) __atomic_store_n(&label[dofs[i]], tagB, __ATOMIC_RELAXED);
    int sa = 0, sb = 0;
    for (int i = lo; i < hi; ++i) {
      const int v = dofs[i], other = (i < bpos) ? tagB : tagA;
      bool sep = false;
      for (int k = A.rowptr[v]; k < A.rowptr[v + 1] && !sep; ++k) sep = (__atomic_load_n(&label[A.colidx[k]], __ATOMIC_RELAXED) == other);
      if (sep) (i < bpos ? sa : sb)++;
    }
    const int sz = std::min(sa, sb);
    if (best_size < 0 || sz < best_size) {
      best_size = sz;
      midp = bpos;
      sep_in_A = sa <= sb;
    }
  }
  for (int i = lo; i < midp; ++i) __atomic_store_n(&label[dofs[i]], tagA, __ATOMIC_RELAXED);
  for (int i = midp; i < hi; ++i) __atomic_store_n(&label[dofs[i]], tagB, __ATOMIC_RELAXED);
  // Ap / Bv: the two halves without the separator S (taken from one side)
  for (int i = lo; i < hi; ++i) {
    const int v = dofs[i];
    const bool inA = i < midp;
    bool sep = false;
    if (inA == sep_in_A) {
      const int other = inA ? tagB : tagA;
      for (int k = A.rowptr[v]; k < A.rowptr[v + 1] && !sep; ++k) sep = (__atomic_load_n(&label[A.colidx[k]], __ATOMIC_RELAXED) == other);
    }
    (sep ? S : (inA ? Ap : Bv)).push_back(v);
  }
  if (Ap.empty() || Bv.empty() || (int)S.size() * 2 > cnt) return make_leaf();  // S empty: disconnected halves, empty separator node
  }      // !guided
  std::copy(Ap.begin(), Ap.end(), dofs.begin() + lo);
  std::copy(Bv.begin(), Bv.end(), dofs.begin() + lo + Ap.size());
  std::copy(S.begin(), S.end(), dofs.begin() + lo + Ap.size() + Bv.size());
  const int a_end = lo + (int)Ap.size(), b_end = a_end + (int)Bv.size();
  int cl, cr;
  if (par_depth > 0 && cnt > 4096) {
    Subtree left, right;
    std::thread worker([&] { build(dofs, lo, a_end, A, coords, dim, leaf, label, next_label, left, par_depth - 1, cmask, rlo, rmid); });
    build(dofs, a_end, b_end, A, coords, dim, leaf, label, next_label, right, par_depth - 1, cmask, rmid, rhi);
    worker.join();
    auto append = [&](Subtree& part) {
      const int off = (int)nodes_.size();
      for (Node& nd : part.nodes) {
        if (nd.parent >= 0) nd.parent += off;
        for (int& c : nd.children) c += off;
        nodes_.push_back(std::move(nd));
      }
      for (auto& o : part.own) own.push_back(std::move(o));
      return (int)nodes_.size() - 1;      // the part's root
    };
    cl = append(left);
    cr = append(right);
  } else {
    cl = build(dofs, lo, a_end, A, coords, dim, leaf, label, next_label, sub, 0, cmask, rlo, rmid);
    cr = build(dofs, a_end, b_end, A, coords, dim, leaf, label, next_label, sub, 0, cmask, rmid, rhi);
  }
  nodes_.emplace_back();
  own.emplace_back(S);
  const int t = (int)nodes_.size() - 1;
  nodes_[t].children = {cl, cr};
  nodes_[cl].parent = t;
  nodes_[cr].parent = t;
  return t;
}

void MfChol::analyze(const Csr& Ain, const double* coords, int dim, int leaf_size, const unsigned long long* rank_mask, int world) {
  if (const char* e = std::getenv("MGB_LEAF")) leaf_size = std::max(8, std::atoi(e));      // tuning knob
  if (Ain.rows != Ain.cols) throw ArgError("MfChol: matrix not square");
  n_ = Ain.rows;
  static const bool vt = std::getenv("MGB_VERBOSE_SETUP") != nullptr;
  auto tnow = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double tph = tnow();
  auto phase = [&](const char* what) {
    if (vt) std::fprintf(stderr, "[mgb setup] analyze %-19s %.3f s\n", what, tnow() - tph);
    tph = tnow();
  };
  // symmetric adjacency (pattern + transpose) for ordering and the symbolic phase: counting pass, fill, per-row sort
  // (a global sort of the 2 nnz triplets was a third of the analysis)
  Csr A(n_, n_);
  {
    for (int r = 0; r < n_; ++r)
      for (int k = Ain.rowptr[r]; k < Ain.rowptr[r + 1]; ++k) {
        const int c = Ain.colidx[k];
        A.rowptr[r + 1]++;
        if (c != r) A.rowptr[c + 1]++;
      }
    for (int r = 0; r < n_; ++r) A.rowptr[r + 1] += A.rowptr[r];
    A.colidx.resize(A.rowptr[n_]);
    std::vector<int> pos(A.rowptr.begin(), A.rowptr.end() - 1);
    for (int r = 0; r < n_; ++r)
      for (int k = Ain.rowptr[r]; k < Ain.rowptr[r + 1]; ++k) {
        const int c = Ain.colidx[k];
        A.colidx[pos[r]++] = c;
        if (c != r) A.colidx[pos[c]++] = r;
      }
    for (int r = 0; r < n_; ++r) {
      std::sort(A.colidx.begin() + A.rowptr[r], A.colidx.begin() + A.rowptr[r + 1]);
      // the input holds every unordered pair once, so there is nothing to merge
    }
    A.vals.assign(A.colidx.size(), 1.0);
  }
  phase("adjacency");
  nodes_.clear();
  roots_.clear();
  std::vector<int> dofs(n_), label(n_, -1);
  std::iota(dofs.begin(), dofs.end(), 0);
  std::atomic<int> next_label{0};
  Subtree whole;
  int par_depth = 0;
  while ((1 << par_depth) < hw_threads()) ++par_depth;
  const bool ranked = rank_mask && world > 1 && world <= 64 && (world & (world - 1)) == 0;
  align_failed_.store(false);
  if (n_ > 0)
    roots_.push_back(build(dofs, 0, n_, A, coords, dim, leaf_size, label, next_label, whole, par_depth, ranked ? rank_mask : nullptr,
                           0, ranked ? world : 1));
  aligned_world_ = (ranked && !align_failed_.load()) ? world : 1;
  nodes_ = std::move(whole.nodes);
  std::vector<std::vector<int>>& own = whole.own;
  phase("nested dissection");
  // numbering: nodes are already in postorder
  perm_.resize(n_);
  iperm_.resize(n_);
  std::vector<int> node_of(n_);
  int cnt = 0;
  for (size_t t = 0; t < nodes_.size(); ++t) {
    nodes_[t].first = cnt;
    nodes_[t].ns = (int)own[t].size();
    std::sort(own[t].begin(), own[t].end());
    for (int v : own[t]) {
      perm_[cnt] = v;
      iperm_[v] = cnt;
      node_of[cnt] = (int)t;
      ++cnt;
    }
  }
  // symbolic: boundary index lists
  std::vector<int> stamp(n_, -1);
  size_t total = 0;
  max_front_ = 0;
  flops_ = 0;
  for (size_t t = 0; t < nodes_.size(); ++t) {
    Node& nd = nodes_[t];
    const int last = nd.first + nd.ns;
    std::vector<int>& b = nd.bdry;
    b.clear();
    for (int j = nd.first; j < last; ++j) {
      const int v = perm_[j];
      for (int k = A.rowptr[v]; k < A.rowptr[v + 1]; ++k) {
        const int i = iperm_[A.colidx[k]];
        if (i >= last && stamp[i] != (int)t) {
          stamp[i] = (int)t;
          b.push_back(i);
        }
      }
    }
    for (int c : nd.children)
      for (int i : nodes_[c].bdry)
        if (i >= last && stamp[i] != (int)t) {
          stamp[i] = (int)t;
          b.push_back(i);
        }
    std::sort(b.begin(), b.end());
    nd.off = total;
    const int nf = nd.nf();
    total += (size_t)nf * nf;
    max_front_ = std::max(max_front_, nf);
    for (int k = 0; k < nd.ns; ++k) flops_ += (double)(nf - k) * (nf - k);
  }
  phase("front index lists");
  auto pos_in = [&](const Node& p, int i) -> int {
    if (i < p.first + p.ns) {
      if (i < p.first) throw InternalError("MfChol: index below front");
      return i - p.first;
    }
    auto it = std::lower_bound(p.bdry.begin(), p.bdry.end(), i);
    if (it == p.bdry.end() || *it != i) throw InternalError("MfChol: index missing from parent front");
    return p.ns + (int)(it - p.bdry.begin());
  };
  for (size_t t = 0; t < nodes_.size(); ++t) {
    Node& nd = nodes_[t];
    nd.ea.clear();
    if (nd.parent >= 0)
      for (int i : nd.bdry) nd.ea.push_back(pos_in(nodes_[nd.parent], i));
    else if (!nd.bdry.empty())
      throw InternalError("MfChol: root with boundary");
  }
  phase("extend-add maps");
  // assembly map
  a_idx_.assign(nodes_.size(), {});
  a_pos_.assign(nodes_.size(), {});
  // every unordered pair must appear exactly once in the input (e.g. its lower triangle)
  for (int r = 0; r < n_; ++r)
    for (int k = Ain.rowptr[r]; k < Ain.rowptr[r + 1]; ++k) {
      int i = iperm_[r], j = iperm_[Ain.colidx[k]];
      if (i < j) std::swap(i, j);
      const int t = node_of[j];
      const Node& nd = nodes_[t];
      a_idx_[t].push_back(k);
      a_pos_[t].push_back(pos_in(nd, i) + nd.nf() * (j - nd.first));
    }
  a_nnz_ = Ain.rowptr[n_];
  phase("assembly map");
  // the host fronts (106 MB at fem2d L=7) are only touched by the HOST numeric factorisation: allocated on its first use
  fronts_total_ = total;
  std::vector<double>().swap(fronts_);
}

void MfChol::factor_node(int t, const double* vals, bool& ok) {
  Node& nd = nodes_[t];
  const int nf = nd.nf();
  double* F = fronts_.data() + nd.off;
  std::fill(F, F + (size_t)nf * nf, 0.0);
  const std::vector<int>& ai = a_idx_[t];
  const std::vector<int>& ap = a_pos_[t];
  for (size_t q = 0; q < ai.size(); ++q) F[ap[q]] += vals[ai[q]];
  for (int c : nd.children) {
    const Node& ch = nodes_[c];
    const int cf = ch.nf(), cs = ch.ns, nb = (int)ch.bdry.size();
    const double* G = fronts_.data() + ch.off;
    for (int b = 0; b < nb; ++b) {
      const double* gc = G + (size_t)(cs + b) * cf + cs;
      double* fc = F + (size_t)ch.ea[b] * nf;
      for (int a = b; a < nb; ++a) fc[ch.ea[a]] += gc[a];
    }
  }
  if (!partial_chol(F, nf, nd.ns, (double)nd.ns * nf * nf > 3e7 ? hw_threads() : 1)) ok = false;
}

bool MfChol::factor(const double* vals) {
  if (n_ == 0) return true;
  if (fronts_.size() != fronts_total_) fronts_.assign(fronts_total_, 0.0);
  bool ok = true;
  const int nt = hw_threads();
  const int nn = (int)nodes_.size();
  // subtree sizes (postorder: subtree of t is the contiguous range [t-size+1, t])
  std::vector<double> work(nn, 0.0);
  std::vector<int> sz(nn, 1);
  for (int t = 0; t < nn; ++t) {
    const Node& nd = nodes_[t];
    double w = 0;
    for (int k = 0; k < nd.ns; ++k) w += (double)(nd.nf() - k) * (nd.nf() - k);
    work[t] += w;
    if (nd.parent >= 0) {
      work[nd.parent] += work[t];
      sz[nd.parent] += sz[t];
    }
  }
  std::vector<int> tasks;     // roots of independent subtrees
  std::vector<char> in_task(nn, 0);
  if (nt > 1) {
    const double thresh = work[nn - 1] / (4.0 * nt);
    for (int t = nn - 1; t >= 0; --t) {
      const int p = nodes_[t].parent;
      if (p >= 0 && in_task[p]) {
        in_task[t] = 1;
        continue;
      }
      if (work[t] <= thresh || nodes_[t].children.empty()) {
        in_task[t] = 1;
        tasks.push_back(t);
      }
    }
    std::sort(tasks.begin(), tasks.end(), [&](int a, int b) { return work[a] > work[b]; });
    std::atomic<bool> aok(true);
    parallel_for((int)tasks.size(), nt, [&](int q) {
      const int root = tasks[q];
      bool lok = true;
      for (int t = root - sz[root] + 1; t <= root; ++t) factor_node(t, vals, lok);
      if (!lok) aok = false;
    });
    ok = aok;
  }
  for (int t = 0; t < nn && ok; ++t)
    if (!in_task[t]) factor_node(t, vals, ok);
  return ok;
}

void MfChol::forward_node(int t, double* y) const {      // L y = b restricted to the pivots of node t
  const Node& nd = nodes_[t];
  const int nf = nd.nf(), ns = nd.ns;
  const double* F = fronts_.data() + nd.off;
  double* yo = y + nd.first;
  for (int k = 0; k < ns; ++k) {
    const double* ck = F + (size_t)k * nf;
    const double v = yo[k] / ck[k];
    yo[k] = v;
    for (int i = k + 1; i < ns; ++i) yo[i] -= ck[i] * v;
    for (int i = ns; i < nf; ++i) y[nd.bdry[i - ns]] -= ck[i] * v;
  }
}

void MfChol::backward_node(int t, double* y) const {     // L' x = y restricted to the pivots of node t
  const Node& nd = nodes_[t];
  const int nf = nd.nf(), ns = nd.ns;
  const double* F = fronts_.data() + nd.off;
  double* yo = y + nd.first;
  for (int k = ns - 1; k >= 0; --k) {
    const double* ck = F + (size_t)k * nf;
    double v = yo[k];
    for (int i = k + 1; i < ns; ++i) v -= ck[i] * yo[i];
    for (int i = ns; i < nf; ++i) v -= ck[i] * y[nd.bdry[i - ns]];
    yo[k] = v / ck[k];
  }
}

void MfChol::solve(double* b) const {
  if (n_ == 0) return;
  std::vector<double> y(n_);
  for (int i = 0; i < n_; ++i) y[i] = b[perm_[i]];
  const int nn = (int)nodes_.size();
  for (int t = 0; t < nn; ++t) forward_node(t, y.data());
  for (int t = nn - 1; t >= 0; --t) backward_node(t, y.data());
  for (int i = 0; i < n_; ++i) b[perm_[i]] = y[i];
}

CholPartition MfChol::partition(int world) const {
  CholPartition part;
  const int nn = (int)nodes_.size();
  part.owner.assign(nn, -1);
  if (world < 2 || (world & (world - 1)) || roots_.size() != 1) return part;      // replicated
  std::vector<int> frontier{roots_[0]};
  for (int w = 1; w < world; w *= 2) {
    std::vector<int> next;
    for (int t : frontier) {
      if (nodes_[t].children.size() != 2) return part;      // top is not a complete binary tree: replicate
      next.push_back(nodes_[t].children[0]);
      next.push_back(nodes_[t].children[1]);
    }
    frontier.swap(next);
  }
  // postorder: the subtree of t is the contiguous node range [t - size(t) + 1, t]
  std::vector<int> sz(nn, 1);
  for (int t = 0; t < nn; ++t)
    if (nodes_[t].parent >= 0) sz[nodes_[t].parent] += sz[t];
  for (int j = 0; j < world; ++j) {
    const int r = frontier[j];
    if (nodes_[r].bdry.empty() && nodes_[r].parent >= 0) {}      // a subtree without boundary exchanges nothing: fine
    for (int t = r - sz[r] + 1; t <= r; ++t) part.owner[t] = j;
  }
  part.world = world;
  part.roots = frontier;
  return part;
}

std::vector<int> MfChol::top_value_indices(const CholPartition& part) const {
  std::vector<int> idx;
  if (!part.split()) return idx;
  for (size_t t = 0; t < nodes_.size(); ++t)
    if (part.owner[t] < 0) idx.insert(idx.end(), a_idx_[t].begin(), a_idx_[t].end());
  return idx;
}

bool MfChol::factor_solve_dist(const double* vals_in, double* b, const CholPartition& part, int rank, const Allreduce& allreduce,
                               bool vals_local) {
  if (n_ == 0) return true;
  const int nn = (int)nodes_.size();
  if (vals_local && !(part.split() && rank_aligned(part.world)))
    throw ArgError("MfChol: rank-local values need a tree whose top follows the row partition (analyze with rank masks)");
  const double* vals = vals_in;
  if (!part.split()) {
    const bool ok = factor(vals);
    if (ok) solve(b);
    return ok;
  }
  if ((int)part.owner.size() != nn || rank < 0 || rank >= part.world) throw ArgError("MfChol: partition does not match the tree");
  const std::vector<int> top_idx = vals_local ? top_value_indices(part) : std::vector<int>();
  std::vector<double> vals_sum;      // vals_local: this rank's values with the top entries replaced by their sums over the ranks
  if (fronts_.size() != fronts_total_) fronts_.assign(fronts_total_, 0.0);
  bool ok = true;
  std::vector<double> y(n_), y0(n_);
  for (int i = 0; i < n_; ++i) y0[i] = y[i] = b[perm_[i]];
  // (1) own subtree: factor, forward sweep (its updates of top unknowns are the right-hand-side contribution)
  for (int t = 0; t < nn; ++t)
    if (part.owner[t] == rank) factor_node(t, vals, ok);
  for (int t = 0; t < nn && ok; ++t)
    if (part.owner[t] == rank) forward_node(t, y.data());
  // (2) exchange: [flag | per subtree root: lower triangle of its Schur complement | updates of the top right-hand side]
  std::vector<int> top_dofs;
  for (int t = 0; t < nn; ++t)
    if (part.owner[t] < 0)
      for (int k = 0; k < nodes_[t].ns; ++k) top_dofs.push_back(nodes_[t].first + k);
  std::vector<long long> xoff(part.world + 1, 1);
  for (int j = 0; j < part.world; ++j) {
    const long long nb = (long long)nodes_[part.roots[j]].bdry.size();
    xoff[j + 1] = xoff[j] + nb * (nb + 1) / 2;
  }
  const size_t voff = (size_t)xoff[part.world] + top_dofs.size();
  std::vector<double> xb(voff + top_idx.size(), 0.0);
  xb[0] = ok ? 0.0 : 1.0;
  for (size_t q = 0; q < top_idx.size(); ++q) xb[voff + q] = vals[top_idx[q]];      // partial sums of the top entries
  if (ok) {
    const Node& ch = nodes_[part.roots[rank]];
    const int cf = ch.nf(), cs = ch.ns, nb = (int)ch.bdry.size();
    const double* G = fronts_.data() + ch.off;
    double* dst = xb.data() + xoff[rank];
    for (int bcol = 0; bcol < nb; ++bcol)
      for (int a = bcol; a < nb; ++a) *dst++ = G[(size_t)(cs + bcol) * cf + cs + a];
    for (size_t q = 0; q < top_dofs.size(); ++q) xb[(size_t)xoff[part.world] + q] = y[top_dofs[q]] - y0[top_dofs[q]];
  }
  allreduce(xb.data(), (long long)xb.size());
  if (xb[0] != 0.0) return false;      // some rank hit a non-positive pivot
  if (vals_local) {
    vals_sum.assign(vals, vals + a_nnz_);
    for (size_t q = 0; q < top_idx.size(); ++q) vals_sum[top_idx[q]] = xb[voff + q];
    vals = vals_sum.data();
  }
  for (int j = 0; j < part.world; ++j) {
    Node& ch = nodes_[part.roots[j]];
    const int cf = ch.nf(), cs = ch.ns, nb = (int)ch.bdry.size();
    double* G = fronts_.data() + ch.off;
    const double* src = xb.data() + xoff[j];
    for (int bcol = 0; bcol < nb; ++bcol)
      for (int a = bcol; a < nb; ++a) G[(size_t)(cs + bcol) * cf + cs + a] = *src++;
  }
  for (size_t q = 0; q < top_dofs.size(); ++q) y[top_dofs[q]] = y0[top_dofs[q]] + xb[(size_t)xoff[part.world] + q];
  // (3) top: every rank does the same arithmetic on the same data
  for (int t = 0; t < nn; ++t)
    if (part.owner[t] < 0) factor_node(t, vals, ok);
  if (!ok) return false;      // identical on every rank
  for (int t = 0; t < nn; ++t)
    if (part.owner[t] < 0) forward_node(t, y.data());
  for (int t = nn - 1; t >= 0; --t)
    if (part.owner[t] < 0) backward_node(t, y.data());
  // (4) own subtree back, then assemble x: every unknown is contributed by exactly one rank (the top by rank 0)
  for (int t = nn - 1; t >= 0; --t)
    if (part.owner[t] == rank) backward_node(t, y.data());
  std::vector<double> xs(n_, 0.0);
  for (int t = 0; t < nn; ++t)
    if (part.owner[t] == rank || (part.owner[t] < 0 && rank == 0))
      for (int k = 0; k < nodes_[t].ns; ++k) xs[perm_[nodes_[t].first + k]] = y[nodes_[t].first + k];
  allreduce(xs.data(), n_);
  std::copy(xs.begin(), xs.end(), b);
  return true;
}

}  // namespace mgb
