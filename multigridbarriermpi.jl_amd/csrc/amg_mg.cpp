// Multigrid hierarchy, V-cycle and preconditioned CG of the Newton path (kernels: mg.hip).
//
// Hierarchy: the reference's AMG levels R_l = blockdiag(subspaces[sv][l]) (test/test_d0_construction.jl:82-100) are nested,
// R_l = R_{l+1} P_l, so the Newton matrices A_l = R_l' H R_l (test/test_map_rows_compare.jl:102-123,165-170) are the Galerkin
// operators P_l' A_{l+1} P_l of one another.  A V-cycle over them -- Chebyshev-Jacobi smoothing, dense inverse on the coarsest
// level -- preconditions CG on the Newton system of the top level, whose operator is applied matrix-free:
// H v = B' (Y o (B v)).  The reference solves the same system directly (MultiGridBarrier.solve -> MUMPS,
// test/test_instrumented_solve.jl:25-28,99): solver = "pcg" is an alternative to the device Cholesky, not a change of algorithm.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <thread>

#include "amg.hpp"

namespace mgb {

static double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ------------------------------------------------------------------ host symbolic pieces

bool DevElOpOwned::build(const Csr& B, const DevCsr& Bdev, int block, int K, int nY) {
  view = DevElOp();
  const int rpe = block * K;
  if (block < 1 || K < 1 || K > 8 || rpe > 65535 || B.rows == 0 || B.rows % rpe) return false;
  const int nel = B.rows / rpe;
  std::vector<std::vector<int>> cols(nel);
  int cmax = 0, nzm = 0;
  for (int e = 0; e < nel; ++e) {
    std::vector<int>& c = cols[e];
    c.assign(B.colidx.begin() + B.rowptr[e * rpe], B.colidx.begin() + B.rowptr[(e + 1) * rpe]);
    nzm = std::max(nzm, (int)c.size());
    std::sort(c.begin(), c.end());
    c.erase(std::unique(c.begin(), c.end()), c.end());
    cmax = std::max(cmax, (int)c.size());
  }
  if (cmax < 1 || cmax > 255 || nzm > 65535) return false;
  // threads per element: the smallest power of two covering its rows (8 .. 256); LDS per element: xs, vs, ds, us, ys (doubles)
  // + rowptr / tptr / tk / trow (16 bit) + lcol (8 bit) of its class, padded to doubles
  int tpe = 8;
  while (tpe < 256 && tpe < rpe) tpe <<= 1;
  const int table_bytes = 2 * (rpe + 1) + 2 * (cmax + 1);
  const int slot_doubles = cmax + 3 * nzm + 2 * rpe + block * nY + (table_bytes + 7) / 8;      // xs, vs, pr, ent, ds, us, ys, rp, tp
  while (tpe < 256 && (size_t)(256 / tpe) * slot_doubles * 8 > 60 * 1024) tpe <<= 1;      // fewer elements per pass
  if ((size_t)(256 / tpe) * slot_doubles * 8 > 60 * 1024) return false;
  // structure classes: (relative row offsets, local columns) -> class id
  std::map<std::string, int> ids;
  std::vector<int> cls(nel);
  std::vector<unsigned short> c_rowptr, c_tptr;
  std::vector<unsigned long long> c_ent;
  std::vector<int> ec((size_t)nel * cmax);
  std::string key;
  std::vector<unsigned char> lc;
  for (int e = 0; e < nel; ++e) {
    const std::vector<int>& c = cols[e];
    for (int j = 0; j < cmax; ++j) ec[(size_t)e * cmax + j] = j < (int)c.size() ? c[j] : c[0];
    const int k0 = B.rowptr[e * rpe], k1 = B.rowptr[(e + 1) * rpe];
    lc.resize(k1 - k0);
    for (int k = k0; k < k1; ++k) lc[k - k0] = (unsigned char)(std::lower_bound(c.begin(), c.end(), B.colidx[k]) - c.begin());
    key.assign((const char*)lc.data(), lc.size());
    for (int r = 0; r <= rpe; ++r) {
      const unsigned short off = (unsigned short)(B.rowptr[e * rpe + r] - k0);
      key.append((const char*)&off, sizeof off);
    }
    auto it = ids.find(key);
    if (it == ids.end()) {
      const int id = (int)ids.size();
      it = ids.emplace(key, id).first;
      c_rowptr.resize((size_t)(id + 1) * (rpe + 1));
      c_tptr.resize((size_t)(id + 1) * (cmax + 1));
      c_ent.resize((size_t)(id + 1) * nzm, 0ull);
      for (int r = 0; r <= rpe; ++r) c_rowptr[(size_t)id * (rpe + 1) + r] = (unsigned short)(B.rowptr[e * rpe + r] - k0);
      for (size_t k = 0; k < lc.size(); ++k) c_ent[(size_t)id * nzm + k] |= (unsigned long long)lc[k] << 32;
      // column-wise traversal: rows ascending within a column (a node's K rows are consecutive)
      std::vector<int> cnt(cmax + 1, 0);
      for (unsigned char j : lc) cnt[j + 1]++;
      for (int j = 0; j < cmax; ++j) cnt[j + 1] += cnt[j];
      for (int j = 0; j <= cmax; ++j) c_tptr[(size_t)id * (cmax + 1) + j] = (unsigned short)cnt[j];
      std::vector<int> pos(cnt.begin(), cnt.end() - 1);
      for (int r = 0; r < rpe; ++r)
        for (int k = B.rowptr[e * rpe + r] - k0; k < B.rowptr[e * rpe + r + 1] - k0; ++k) {
          const int pp = pos[lc[k]]++;
          c_ent[(size_t)id * nzm + pp] |= (unsigned long long)(unsigned short)k | (unsigned long long)(unsigned short)r << 16;
        }
    }
    cls[e] = it->second;
  }
  // dof gather lists: slots (e, j) of real columns only, elements ascending
  const int N = B.cols;
  std::vector<int> dptr(N + 1, 0), didx;
  for (int e = 0; e < nel; ++e)
    for (int c : cols[e]) dptr[c + 1]++;
  for (int i = 0; i < N; ++i) dptr[i + 1] += dptr[i];
  didx.resize(dptr[N]);
  {
    std::vector<int> pos(dptr.begin(), dptr.end() - 1);
    for (int e = 0; e < nel; ++e)
      for (int j = 0; j < (int)cols[e].size(); ++j) didx[pos[cols[e][j]]++] = e * cmax + j;
  }
  h_ecols = ec;
  h_cls = cls;
  h_tptr = c_tptr;
  h_ent = c_ent;
  this->ecols.upload(ec.data(), ec.size());
  this->cls.upload(cls.data(), cls.size());
  this->dptr.upload(dptr.data(), dptr.size());
  this->didx.upload(didx.data(), didx.size());
  this->c_rowptr.upload(c_rowptr.data(), c_rowptr.size());
  this->c_tptr.upload(c_tptr.data(), c_tptr.size());
  this->c_ent.upload(c_ent.data(), c_ent.size());
  view.nel = nel;
  view.rows_per_el = rpe;
  view.cmax = cmax;
  view.K = K;
  view.block = block;
  view.nnz_max = nzm;
  view.ncls = (int)ids.size();
  view.N = N;
  view.tpe = tpe;
  view.slot_doubles = slot_doubles;
  view.ecols = this->ecols.p;
  view.cls = this->cls.p;
  view.rowptr = Bdev.rowptr;
  view.vals = Bdev.vals;
  view.c_rowptr = this->c_rowptr.p;
  view.c_tptr = this->c_tptr.p;
  view.c_ent = this->c_ent.p;
  view.dptr = this->dptr.p;
  view.didx = this->didx.p;
  return true;
}

bool DevElAsmOwned::build(DevElOpOwned& EO, const Csr& Apat, const BarrierParams& P) {
  view = DevElAsm();
  const DevElOp& E = EO.view;
  if (!E.valid() || EO.h_cls.empty()) return false;
  const int K = E.K, cmax = E.cmax, nzm = E.nnz_max, ncls = E.ncls, blk = E.block;
  // rows a that some barrier term couples with rows b: act[a] = bit mask of the b (pairs of active rows of one term)
  unsigned act[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int ci = 0; ci < P.ncones; ++ci) {
    unsigned m = 0;
    for (int a = 0; a < P.cone[ci].nact(); ++a) m |= 1u << P.cone[ci].col(a);
    for (int a = 0; a < P.cone[ci].nact(); ++a) act[P.cone[ci].col(a)] |= m;
  }
  // Y entry of (a1, a2) per barrier term: slot of the pair inside the term's packed upper triangle, or -1
  int yslot[kMaxCones][8][8];
  {
    int base = 0;
    for (int ci = 0; ci < kMaxCones; ++ci)
      for (int a = 0; a < 8; ++a)
        for (int b = 0; b < 8; ++b) yslot[ci][a][b] = -1;
    for (int ci = 0; ci < P.ncones; ++ci) {
      const ConeSpec& S = P.cone[ci];
      int slot = base;
      for (int a = 0; a < S.nact(); ++a)
        for (int b = a; b < S.nact(); ++b, ++slot) yslot[ci][S.col(a)][S.col(b)] = yslot[ci][S.col(b)][S.col(a)] = slot;
      base += S.nY();
    }
  }
  const int nY = P.nY();
  if ((long long)blk * nY > 65535) return false;
  // per class: the lower-triangle pairs (i >= j) of local columns that share a node on coupled rows, and for each pair its
  // products (entry of column i, entry of column j on the same node, Y entry)
  std::vector<std::vector<unsigned short>> pairs(ncls);
  std::vector<std::vector<int>> tptr(ncls);
  std::vector<std::vector<unsigned long long>> terms(ncls);
  int npm = 0, ntm = 0;
  for (int c = 0; c < ncls; ++c) {
    const unsigned short* tp = EO.h_tptr.data() + (size_t)c * (cmax + 1);
    const unsigned long long* ent = EO.h_ent.data() + (size_t)c * nzm;
    tptr[c].push_back(0);
    for (int i = 0; i < cmax; ++i)
      for (int j = 0; j <= i; ++j) {
        const size_t before = terms[c].size();
        for (int p1 = tp[i]; p1 < tp[i + 1]; ++p1) {
          const int k1 = (int)(ent[p1] & 0xffffu), r1 = (int)((ent[p1] >> 16) & 0xffffu), node = r1 / K, a1 = r1 % K;
          for (int p2 = tp[j]; p2 < tp[j + 1]; ++p2) {
            const int k2 = (int)(ent[p2] & 0xffffu), r2 = (int)((ent[p2] >> 16) & 0xffffu);
            if (r2 / K != node) continue;
            for (int ci = 0; ci < P.ncones; ++ci) {
              const int sl = yslot[ci][a1][r2 % K];
              if (sl >= 0)
                terms[c].push_back((unsigned long long)k1 | (unsigned long long)k2 << 16 | (unsigned long long)(node * nY + sl) << 32);
            }
          }
        }
        if (terms[c].size() > before) {
          pairs[c].push_back((unsigned short)(i | j << 8));
          tptr[c].push_back((int)terms[c].size());
        }
      }
    npm = std::max(npm, (int)pairs[c].size());
    ntm = std::max(ntm, (int)terms[c].size());
  }
  (void)act;
  if (npm == 0) return false;
  std::vector<int> cn(ncls), ctp((size_t)ncls * (npm + 1), 0);
  std::vector<unsigned long long> cte((size_t)ncls * ntm, 0ull);
  for (int c = 0; c < ncls; ++c) {
    cn[c] = (int)pairs[c].size();
    std::copy(tptr[c].begin(), tptr[c].end(), ctp.begin() + (size_t)c * (npm + 1));
    std::copy(terms[c].begin(), terms[c].end(), cte.begin() + (size_t)c * ntm);
  }
  // gather lists: every element slot goes to one lower-triangle entry of the pattern
  const int nnzA = Apat.nnz();
  if ((long long)E.nel * npm > 2000000000LL) return false;
  std::vector<int> tgt((size_t)E.nel * npm, -1), aptr(nnzA + 1, 0);
  const int nthr = std::max(1, std::min(MfChol::threads(), E.nel / 1024));
  std::vector<char> bad(nthr, 0);
  {
    std::vector<std::thread> th;
    auto work = [&](int t) {
      for (int e = (int)((long long)E.nel * t / nthr); e < (int)((long long)E.nel * (t + 1) / nthr); ++e) {
        const int c = EO.h_cls[e];
        const int* ec = EO.h_ecols.data() + (size_t)e * cmax;
        for (int s = 0; s < cn[c]; ++s) {
          const int I = ec[pairs[c][s] & 0xff], J = ec[pairs[c][s] >> 8];
          const int* b0 = Apat.colidx.data() + Apat.rowptr[I];
          const int* e0 = Apat.colidx.data() + Apat.rowptr[I + 1];
          const int* it = std::lower_bound(b0, e0, J);
          if (it == e0 || *it != J) {
            bad[t] = 1;
            continue;
          }
          tgt[(size_t)e * npm + s] = (int)(it - Apat.colidx.data());
        }
      }
    };
    for (int t = 1; t < nthr; ++t) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
  }
  for (char b : bad)
    if (b) return false;      // an element pair outside the pattern: keep the plan T
  for (size_t k = 0; k < tgt.size(); ++k)
    if (tgt[k] >= 0) aptr[tgt[k] + 1]++;
  for (int a = 0; a < nnzA; ++a) aptr[a + 1] += aptr[a];
  std::vector<int> aidx(aptr[nnzA]), pos(aptr.begin(), aptr.end() - 1);
  for (size_t k = 0; k < tgt.size(); ++k)      // k ascending = elements ascending: the fixed summation order
    if (tgt[k] >= 0) aidx[pos[tgt[k]]++] = (int)k;
  c_npairs.upload(cn.data(), cn.size());
  c_tptr.upload(ctp.data(), ctp.size());
  c_terms.upload(cte.data(), cte.size());
  this->aptr.upload(aptr.data(), aptr.size());
  this->aidx.upload(aidx.data(), aidx.size());
  elmat.alloc((size_t)E.nel * npm);
  hip_check(hipMemset(elmat.p, 0, elmat.n * sizeof(double)), "memset elmat");      // slots beyond a class's pair count stay zero
  view.npm = npm;
  view.ntm = ntm;
  view.nnzA = nnzA;
  view.slot_doubles = nzm + blk * nY;      // vs, ys
  if ((size_t)(256 / E.tpe) * view.slot_doubles * 8 > 60 * 1024) {
    view = DevElAsm();
    return false;
  }
  view.c_npairs = c_npairs.p;
  view.c_tptr = c_tptr.p;
  view.c_terms = c_terms.p;
  view.aptr = this->aptr.p;
  view.aidx = this->aidx.p;
  // the host copies of the element tables are no longer needed
  std::vector<int>().swap(EO.h_ecols);
  std::vector<int>().swap(EO.h_cls);
  std::vector<unsigned short>().swap(EO.h_tptr);
  std::vector<unsigned long long>().swap(EO.h_ent);
  return true;
}

Csr build_prolongation(const Csr& Rf, const Csr& Rc) {
  if (Rf.rows != Rc.rows) throw ArgError("mg: levels live on different node sets");
  const int Nf = Rf.cols;
  std::vector<int> rep(Nf, -1);
  for (int r = 0; r < Rf.rows; ++r) {
    int col = -1, big = 0;
    for (int k = Rf.rowptr[r]; k < Rf.rowptr[r + 1]; ++k)
      if (std::fabs(Rf.vals[k]) > 1e-12) {
        ++big;
        col = std::fabs(Rf.vals[k] - 1.0) < 1e-12 ? Rf.colidx[k] : -1;
      }
    if (big == 1 && col >= 0 && rep[col] < 0) rep[col] = r;
  }
  for (int i = 0; i < Nf; ++i)
    if (rep[i] < 0)
      throw ArgError("mg: the level hierarchy is not nodal (no node of the finest mesh carries unknown " + std::to_string(i) +
                     " of a level alone): the V-cycle needs nested nodal subspaces");
  Csr P(Nf, Rc.cols);
  for (int i = 0; i < Nf; ++i) {
    const int r = rep[i];
    for (int k = Rc.rowptr[r]; k < Rc.rowptr[r + 1]; ++k)
      if (std::fabs(Rc.vals[k]) > 1e-14) {
        P.colidx.push_back(Rc.colidx[k]);
        P.vals.push_back(Rc.vals[k]);
      }
    P.rowptr[i + 1] = (int)P.colidx.size();
  }
  return P;
}

Csr sym_full_pattern(const Csr& lower, std::vector<int>& map, std::vector<int>& diagpos) {
  const int N = lower.rows;
  Csr F(N, N);
  std::vector<int> cnt(N, 0);
  for (int i = 0; i < N; ++i)
    for (int k = lower.rowptr[i]; k < lower.rowptr[i + 1]; ++k) {
      const int j = lower.colidx[k];
      if (j > i) throw ArgError("mg: pattern is not lower triangular");
      cnt[i]++;
      if (j != i) cnt[j]++;
    }
  for (int i = 0; i < N; ++i) F.rowptr[i + 1] = F.rowptr[i] + cnt[i];
  F.colidx.assign(F.rowptr[N], 0);
  F.vals.assign(F.rowptr[N], 0.0);
  map.assign(F.rowptr[N], 0);
  diagpos.assign(N, -1);
  std::vector<int> pos(F.rowptr.begin(), F.rowptr.end() - 1);
  // row i: its lower entries (columns ascending, incl. the diagonal) come first when rows are visited in order, and the
  // transposed entries (i, i') of later rows i' > i are appended in ascending i' -- every row ends up sorted
  for (int i = 0; i < N; ++i)
    for (int k = lower.rowptr[i]; k < lower.rowptr[i + 1]; ++k) {
      const int j = lower.colidx[k];
      const int pi = pos[i]++;
      F.colidx[pi] = j;
      map[pi] = k;
      if (j == i) diagpos[i] = pi;
    }
  for (int i = 0; i < N; ++i)
    for (int k = lower.rowptr[i]; k < lower.rowptr[i + 1]; ++k) {
      const int j = lower.colidx[k];
      if (j == i) continue;
      const int pj = pos[j]++;
      F.colidx[pj] = i;
      map[pj] = k;
    }
  for (int i = 0; i < N; ++i)
    if (diagpos[i] < 0) throw ArgError("mg: Hessian pattern without a diagonal entry");
  return F;
}

// ------------------------------------------------------------------ per-level data

Amg::Level::Mg& Amg::mg_of(Level& lv) {
  if (!lv.mg) lv.mg.reset(new Level::Mg);
  if (!mg_inited_) {
    hip_check(hipSetDevice(ctx_.device), "hipSetDevice");
    mg_device_init();
    mg_scal_.alloc(SC_COUNT);
    hip_check(hipMemset(mg_scal_.p, 0, SC_COUNT * sizeof(double)), "memset mg scalars");
    mg_scratch_.alloc(kReductionHeader + 4096);
    hip_check(hipMemset(mg_scratch_.p, 0, mg_scratch_.n * sizeof(double)), "memset mg scratch");
    h_pcg_.alloc(4);
    mg_fail_.alloc(1);
    hip_check(hipMemset(mg_fail_.p, 0, sizeof(int)), "memset mg flag");
    mg_inited_ = true;
  }
  return *lv.mg;
}

void Amg::mg_ensure_vectors(Level& lv) {
  Level::Mg& m = mg_of(lv);
  if (m.vectors) return;
  const int N = lv.plan.N;
  m.dinv.alloc(N);
  m.x.alloc(N);
  m.b.alloc(N);
  m.r.alloc(N);
  m.d0.alloc(N);
  m.d1.alloc(N);
  m.ev0.alloc(N);
  m.ev1.alloc(N);
  m.coef.alloc(kChebStride);
  // deterministic start vector of the power iteration (warm-started from the last estimate afterwards)
  std::vector<double> ev(N);
  unsigned long long sd = 0x9e3779b97f4a7c15ull;
  for (int i = 0; i < N; ++i) {
    sd = sd * 6364136223846793005ull + 1442695040888963407ull;
    ev[i] = 0.5 + (double)(sd >> 11) * (1.0 / 9007199254740992.0);
  }
  m.ev0.upload(ev.data(), N);
  m.vectors = true;
}

bool Amg::mg_ensure_elop(Level& lv) {
  Level::Mg& m = mg_of(lv);
  if (!m.elop_tried) {
    m.elop_tried = true;
    if (geo_.block >= 1 && n_ % geo_.block == 0 && m.elop.build(lv.plan.B, lv.B.view, geo_.block, P_.K, P_.nY()))
      m.elbuf.alloc((size_t)m.elop.view.nel * m.elop.view.cmax);
  }
  return m.elop.view.valid();
}

void Amg::mg_ensure_assembled(Level& lv) {
  Level::Mg& m = mg_of(lv);
  if (m.assembled) return;
  std::vector<int> map, diagpos;
  Csr F = sym_full_pattern(lv.plan.Apat, map, diagpos);
  m.A.upload(F);
  m.amap.upload(map.data(), map.size());
  m.diagpos.upload(diagpos.data(), diagpos.size());
  m.lo_rowptr.upload(lv.plan.Apat.rowptr.data(), lv.plan.Apat.rowptr.size());
  m.lo_colidx.upload(lv.plan.Apat.colidx.data(), lv.plan.Apat.colidx.size());
  m.assembled = true;
}

const Csr& Amg::prolongation_host(int l) {
  mg_ensure_transfer(l);
  return P_host_.at(l);
}

void Amg::mg_ensure_transfer(int l) {
  if (l < 0 || l + 1 >= (int)levels_.size()) throw ArgError("mg: no finer level to prolong to");
  if (ctx_.world > 1) throw ArgError("mg: the V-cycle runs on single-GPU contexts (sharded contexts use the direct solver)");
  Level& lc = level(l);
  Level::Mg& m = mg_of(lc);
  if (m.transfer) return;
  Level& lf = level(l + 1);
  if ((int)P_host_.size() < (int)levels_.size()) P_host_.resize(levels_.size());
  P_host_[l] = build_prolongation(lf.plan.R, lc.plan.R);
  m.P.upload(P_host_[l]);
  m.PT.upload(transpose(P_host_[l]));
  m.transfer = true;
}

int Amg::mg_coarsest(int top) {
  int c0 = 0;
  for (int l = 0; l <= top; ++l)
    if (levels_[l]->plan.N > 0 && levels_[l]->plan.N <= kDenseMax) c0 = l;
  while (c0 < top && levels_[c0]->plan.N == 0) ++c0;
  return c0;
}

// the top level applies H matrix-free (unless it is itself the coarsest level, which needs its assembled values for the inverse)
bool Amg::mg_top_matrix_free(int top) {
  return !pcg_opt.assembled_top && mg_coarsest(top) != top && mg_ensure_elop(level(top));
}

void Amg::mg_prepare(int top) {
  if (ctx_.world > 1) throw ArgError("pcg: single-GPU contexts only (sharded contexts use the direct solver)");
  hip_check(hipSetDevice(ctx_.device), "hipSetDevice");
  if (pcg_opt.degree < 1 || pcg_opt.degree > kChebMaxDegree) throw ArgError("pcg: smoothing degree must be 1..7");
  const int c0 = mg_coarsest(top);
  for (int l = c0; l <= top; ++l) {
    Level& lv = level(l);
    mg_ensure_vectors(lv);
    if (l < top) mg_ensure_transfer(l);
    if (l == top && mg_top_matrix_free(top)) continue;
    mg_ensure_assembled(lv);
  }
  Level& lc = level(c0);
  if (lc.plan.N <= kDenseMax) {
    if (lc.mg->Ainv.n == 0) lc.mg->Ainv.alloc((size_t)lc.plan.N * lc.plan.N);
  } else {
    ensure_chol(lc);      // coarse meshes too large for the dense inverse: the device Cholesky of that level
  }
  Level& lt = level(top);
  if (pcg_r_.n < (size_t)lt.plan.N) {
    pcg_r_.alloc(lt.plan.N);
    pcg_z_.alloc(lt.plan.N);
    pcg_p_.alloc(lt.plan.N);
    pcg_Ap_.alloc(lt.plan.N);
  }
}

// ------------------------------------------------------------------ numeric setup per Newton matrix (Y_ holds w F2(Dz))

void Amg::mg_apply(Level& lv, bool mf, const MgEpi& epi) {
  if (mf) launch_elop_apply(ctx_.stream, lv.mg->elop.view, P_, Y_.p, lv.mg->elbuf.p, epi);
  else launch_csr_apply(ctx_.stream, lv.mg->A.view, epi);
}

void Amg::mg_level_values(Level& lv, bool mf) {
  Level::Mg& m = *lv.mg;
  if (mf) {
    launch_elop_diaginv(ctx_.stream, m.elop.view, P_, Y_.p, m.elbuf.p, m.dinv.p);
    return;
  }
  // assembled: lower-triangle values from the level's Hessian plan (the reference's recipe evaluated on its fixed pattern,
  // test/test_map_rows_compare.jl:102-123), mirrored into the full symmetric storage the smoother sweeps over
  assemble_values(lv);
  launch_expand_sym(ctx_.stream, m.A.view.nnz, m.amap.p, lv.avals.p, m.A.vals.p, lv.plan.N, m.diagpos.p, m.dinv.p);
}

// lambda_max(Dinv A) by power steps in the D inner product (in which Dinv A is self-adjoint: the quotient never exceeds
// lambda_max and grows monotonically), warm-started; then the level's Chebyshev coefficients.  Entirely on the stream.
void Amg::mg_estimate(Level& lv, bool mf, double lmax_given) {
  Level::Mg& m = *lv.mg;
  const int N = lv.plan.N;
  if (lmax_given > 0) {      // tests: the same interval rule from an eigenvalue computed elsewhere
    hip_check(hipStreamSynchronize(ctx_.stream), "sync lmax");
    hip_check(hipMemcpy(mg_scal_.p + SC_LMAX, &lmax_given, sizeof(double), hipMemcpyHostToDevice), "H2D lmax");
    launch_cheb_coef(ctx_.stream, mg_scal_.p, m.coef.p, std::max(pcg_opt.degree, 1), pcg_opt.lo_frac, pcg_opt.hi_frac);
    return;
  }
  launch_power_start(ctx_.stream, N, m.ev0.p, m.dinv.p, mg_scal_.p, mg_scratch_.p);
  const int nsteps = m.ev_warm ? pcg_opt.power_its : std::max(pcg_opt.power_its, pcg_opt.power_its_cold);
  m.ev_warm = true;
  for (int it = 0; it < nsteps; ++it) {
    MgEpi e;
    e.mode = MG_POWER;
    e.n = N;
    e.v = m.ev0.p;
    e.vscale = mg_scal_.p + SC_VSCALE;
    e.out = m.ev1.p;
    e.dinv = m.dinv.p;
    e.scal = mg_scal_.p;
    e.scratch = mg_scratch_.p;
    mg_apply(lv, mf, e);
    std::swap(m.ev0.p, m.ev1.p);
  }
  launch_cheb_coef(ctx_.stream, mg_scal_.p, m.coef.p, std::max(pcg_opt.degree, 1), pcg_opt.lo_frac, pcg_opt.hi_frac);
}

void Amg::mg_values(int top) {
  const int c0 = mg_coarsest(top);
  const bool mf_top = mg_top_matrix_free(top);
  for (int l = c0; l <= top; ++l) {
    Level& lv = level(l);
    if (lv.plan.N == 0) continue;
    const bool mf = (l == top) && mf_top;
    mg_level_values(lv, mf);
    if (l > c0) mg_estimate(lv, mf, 0.0);
  }
  Level& lc = level(c0);
  if (lc.plan.N <= kDenseMax)
    launch_dense_inverse(ctx_.stream, lc.plan.N, lc.mg->lo_rowptr.p, lc.mg->lo_colidx.p, lc.avals.p, lc.mg->Ainv.p, mg_fail_.p);
}

// ------------------------------------------------------------------ smoothing / V-cycle (enqueue only)

void Amg::mg_smooth_pre(Level& lv, bool mf, const double* b, double* x, double* r, int degree, const double* done) {
  Level::Mg& m = *lv.mg;
  double* D[2] = {m.d0.p, m.d1.p};
  for (int a = 0; a < degree; ++a) {
    MgEpi e;
    e.n = lv.plan.N;
    e.mode = a == 0 ? MG_FIRST : MG_STEP;
    e.b = b;
    e.v = a == 0 ? nullptr : D[(a - 1) & 1];
    e.x = x;
    e.r = r;
    e.dinv = m.dinv.p;
    e.coef = m.coef.p;
    e.has_next = a < degree - 1;
    e.k = a + 1;
    e.d_new = D[a & 1];
    e.done = done;
    mg_apply(lv, mf, e);
  }
}

void Amg::mg_smooth_post(Level& lv, bool mf, const double* b, double* x, double* r, int degree, const double* done) {
  Level::Mg& m = *lv.mg;
  double* D[2] = {m.d0.p, m.d1.p};
  for (int a = 0; a < degree; ++a) {
    MgEpi e;
    e.n = lv.plan.N;
    e.mode = a == 0 ? MG_RESID : MG_STEP;
    e.b = b;
    e.v = a == 0 ? x : D[(a - 1) & 1];
    e.x = x;
    e.r = r;
    e.dinv = m.dinv.p;
    e.coef = m.coef.p;
    e.has_next = 1;
    e.add_new = a == degree - 1;
    e.k = a;
    e.d_new = D[a & 1];
    e.done = done;
    mg_apply(lv, mf, e);
  }
  if (degree == 1) launch_waxpby(ctx_.stream, lv.plan.N, x, 1.0, D[0], x);      // x += d_0 cannot ride in the launch that gathers x
}

// x = V(b): one symmetric V-cycle from x = 0 on the levels c0 .. top.  Level vectors: the top level works on the caller's b / x.
void Amg::mg_vcycle(int top, int l, const double* b, double* x, const double* done) {
  Level& lv = level(l);
  Level::Mg& m = *lv.mg;
  const int c0 = mg_coarsest(top);
  if (l == c0) {
    if (lv.plan.N <= kDenseMax) launch_dense_apply(ctx_.stream, lv.plan.N, m.Ainv.p, b, x, done);
    else lv.gchol.factor_solve(ctx_.stream, lv.avals.p, b, x, nullptr, false, true);
    return;
  }
  const bool mf = (l == top) && mg_top_matrix_free(top);
  mg_smooth_pre(lv, mf, b, x, m.r.p, pcg_opt.degree, done);
  // next level with unknowns below
  int lc = l - 1;
  while (lc > c0 && level(lc).plan.N == 0) --lc;
  Level& lvc = level(lc);
  // restriction / prolongation compose over skipped (empty) levels never happens for the reference's geometries: adjacent only
  if (lc != l - 1) throw ArgError("mg: empty intermediate level");
  launch_spmv(ctx_.stream, lvc.mg->PT.view, m.r.p, nullptr, lvc.mg->b.p);
  mg_vcycle(top, lc, lvc.mg->b.p, lvc.mg->x.p, done);
  launch_spmv(ctx_.stream, lvc.mg->P.view, lvc.mg->x.p, x, x);
  mg_smooth_post(lv, mf, b, x, m.r.p, pcg_opt.degree, done);
}

// ------------------------------------------------------------------ preconditioned CG on the Newton system of level `top`

bool Amg::pcg_run(Level& lv, int top, const double* g, double* x, SolveStats* st, int* iters, double* relres) {
  const int N = lv.plan.N;
  const bool mf = mg_top_matrix_free(top);
  double* scal = mg_scal_.p;
  const double* done = scal + SC_DONE;
  const double t0 = now_s();
  h_pcg_.p[0] = -1.0;
  h_pcg_.p[1] = 0.0;
  launch_pcg_init(ctx_.stream, N, g, x, pcg_r_.p, scal, pcg_opt.rtol, pcg_opt.maxit);
  mg_vcycle(top, top, pcg_r_.p, pcg_z_.p, done);
  // completion signals as everywhere else on a single GPU: the dot launch bumps the sequence number the host polls
  auto sig = [&]() {
    HostSignal s;
    s.seq_dev = seq_dev_.p;
    s.seq_host = h_seq_.p;
    ++seq_expected_;
    return s;
  };
  launch_pcg_dot(ctx_.stream, N, pcg_r_.p, pcg_z_.p, scal, mg_scratch_.p, h_pcg_.p, sig());
  launch_pcg_p(ctx_.stream, N, pcg_p_.p, pcg_z_.p, scal);
  int enq = 0;
  for (;;) {
    for (int c = 0; c < std::max(1, pcg_opt.chunk) && enq < pcg_opt.maxit; ++c, ++enq) {
      MgEpi e;
      e.mode = MG_PAP;
      e.n = N;
      e.v = pcg_p_.p;
      e.out = pcg_Ap_.p;
      e.scal = scal;
      e.scratch = mg_scratch_.p;
      e.done = done;
      mg_apply(lv, mf, e);
      launch_pcg_update(ctx_.stream, N, x, pcg_r_.p, pcg_p_.p, pcg_Ap_.p, scal);
      mg_vcycle(top, top, pcg_r_.p, pcg_z_.p, done);
      launch_pcg_dot(ctx_.stream, N, pcg_r_.p, pcg_z_.p, scal, mg_scratch_.p, h_pcg_.p, sig());
      launch_pcg_p(ctx_.stream, N, pcg_p_.p, pcg_z_.p, scal);
    }
    // wait for the batch's last dot (its signal is the newest one expected)
    {
      const volatile unsigned long long* q = h_seq_.p;
      const double tw = now_s();
      for (unsigned long spins = 0; __atomic_load_n(q, __ATOMIC_ACQUIRE) < seq_expected_; ++spins)
        if ((spins & 0xfffff) == 0xfffff && now_s() - tw > 10.0) {
          hip_check(hipStreamSynchronize(ctx_.stream), "sync pcg");
          if (__atomic_load_n(q, __ATOMIC_ACQUIRE) < seq_expected_) throw InternalError("mgb: completion signal lost in pcg");
        }
    }
    if (h_pcg_.p[1] != 0.0 || enq >= pcg_opt.maxit) break;
  }
  const int it = (int)h_pcg_.p[0];
  const double code = h_pcg_.p[1];
  static const bool dbg = std::getenv("MGB_PCG_DEBUG") != nullptr;
  if (dbg)
    std::fprintf(stderr, "[mgb pcg] level %d N=%d: %d iterations, stop code %g (1 converged, 2 breakdown, 3 maxit), <r,Mr> %.3e of %.3e\n",
                 top, N, it, code, h_pcg_.p[2], h_pcg_.p[3]);
  pcg_last_code_ = code;
  if (iters) *iters = it;
  if (relres) *relres = h_pcg_.p[3] > 0 ? std::sqrt(std::fabs(h_pcg_.p[2]) / h_pcg_.p[3]) : 0.0;
  if (st) {
    st->pcg_solves++;
    st->pcg_iters += std::max(it, 0);
    st->time_pcg += now_s() - t0;
  }
  return code == 1.0;
}

// ------------------------------------------------------------------ fine-grained entry points (tests, probes)

void Amg::eval_Y_at(Level& lv, const double* s_host) {
  lv.s_trial.upload(s_host, lv.plan.N);
  dev_apply(lv, lv.s_trial.p, Dz_.p);
  launch_barrier_f2(ctx_.stream, n_, P_, Dz_.p, w_.p, Y_.p);
}

void Amg::hessian_apply(int l, const double* s_host, const double* v_host, double* out_host, bool matrix_free) {
  if (ctx_.world > 1) throw ArgError("hessian_apply: single-GPU contexts only");
  Level& lv = level(l);
  mg_ensure_vectors(lv);
  eval_Y_at(lv, s_host);
  Level::Mg& m = *lv.mg;
  if (matrix_free && !mg_ensure_elop(lv)) throw ArgError("hessian_apply: the operators of this geometry are not element-local");
  if (!matrix_free) {
    mg_ensure_assembled(lv);
    mg_level_values(lv, false);
  }
  m.b.upload(v_host, lv.plan.N);
  MgEpi e;
  e.mode = MG_PLAIN;
  e.n = lv.plan.N;
  e.v = m.b.p;
  e.out = m.x.p;
  mg_apply(lv, matrix_free, e);
  hip_check(hipStreamSynchronize(ctx_.stream), "sync hessian_apply");
  m.x.download(out_host, lv.plan.N);
}

double Amg::mg_smooth(int l, const double* s_host, const double* b_host, double* x_host, int degree, int sweeps, double lmax,
                      bool matrix_free) {
  if (ctx_.world > 1) throw ArgError("mg_smooth: single-GPU contexts only");
  if (degree < 1 || degree > kChebMaxDegree) throw ArgError("mg_smooth: degree must be 1..7");
  Level& lv = level(l);
  mg_ensure_vectors(lv);
  eval_Y_at(lv, s_host);
  Level::Mg& m = *lv.mg;
  if (matrix_free && !mg_ensure_elop(lv)) throw ArgError("mg_smooth: the operators of this geometry are not element-local");
  if (!matrix_free) mg_ensure_assembled(lv);
  mg_level_values(lv, matrix_free);
  const int keep = pcg_opt.degree;
  pcg_opt.degree = degree;
  mg_estimate(lv, matrix_free, lmax);
  m.b.upload(b_host, lv.plan.N);
  m.x.upload(x_host, lv.plan.N);
  for (int sw = 0; sw < sweeps; ++sw) mg_smooth_post(lv, matrix_free, m.b.p, m.x.p, m.r.p, degree, nullptr);
  pcg_opt.degree = keep;
  double used = 0;
  hip_check(hipMemcpyAsync(&used, m.coef.p + kChebStride - 1, sizeof(double), hipMemcpyDeviceToHost, ctx_.stream), "D2H lmax");
  hip_check(hipStreamSynchronize(ctx_.stream), "sync mg_smooth");
  m.x.download(x_host, lv.plan.N);
  return used;
}

void Amg::mg_prolong(int l, const double* xc_host, double* xf_host) {
  mg_ensure_transfer(l);
  Level &lc = level(l), &lf = level(l + 1);
  mg_ensure_vectors(lc);
  mg_ensure_vectors(lf);
  lc.mg->x.upload(xc_host, lc.plan.N);
  launch_spmv(ctx_.stream, lc.mg->P.view, lc.mg->x.p, nullptr, lf.mg->x.p);
  hip_check(hipStreamSynchronize(ctx_.stream), "sync prolong");
  lf.mg->x.download(xf_host, lf.plan.N);
}

void Amg::mg_restrict(int l, const double* rf_host, double* rc_host) {
  mg_ensure_transfer(l);
  Level &lc = level(l), &lf = level(l + 1);
  mg_ensure_vectors(lc);
  mg_ensure_vectors(lf);
  lf.mg->r.upload(rf_host, lf.plan.N);
  launch_spmv(ctx_.stream, lc.mg->PT.view, lf.mg->r.p, nullptr, lc.mg->b.p);
  hip_check(hipStreamSynchronize(ctx_.stream), "sync restrict");
  lc.mg->b.download(rc_host, lc.plan.N);
}

Amg::MgKernelTimes Amg::time_mg_kernels(int l, int reps, int nrot) {
  if (ctx_.world > 1) throw ArgError("time_mg_kernels: single-GPU contexts only");
  Level& lv = level(l);
  mg_ensure_vectors(lv);
  if (!mg_ensure_elop(lv)) throw ArgError("time_mg_kernels: the operators of this geometry are not element-local");
  mg_ensure_assembled(lv);
  const bool has_coarse = l > 0 && level(l - 1).plan.N > 0;
  if (has_coarse) {
    mg_ensure_transfer(l - 1);
    mg_ensure_vectors(level(l - 1));
  }
  if (nrot < 1) nrot = 1;
  Level::Mg& m = *lv.mg;
  const int N = lv.plan.N, nY = P_.nY();
  const DevElOp& E0 = m.elop.view;
  // the point: s = 0 of the current z
  hip_check(hipMemsetAsync(lv.s_trial.p, 0, (size_t)N * sizeof(double), ctx_.stream), "memset");
  dev_apply(lv, lv.s_trial.p, Dz_.p);
  launch_barrier_f2(ctx_.stream, n_, P_, Dz_.p, w_.p, Y_.p);
  mg_level_values(lv, true);       // dinv
  mg_level_values(lv, false);      // assembled values
  mg_estimate(lv, true, 0.0);      // Chebyshev coefficients
  struct Set {
    DevBuf<double> Bvals, Y, v, out, elbuf, x, r, d, Avals;
    DevCsrOwned P, PT;
    DevBuf<double> xc, bc;
  };
  std::vector<std::unique_ptr<Set>> sets;
  struct View {
    DevElOp E;
    DevCsr A, P, PT;
    double *Y, *v, *out, *elbuf, *x, *r, *d, *xc, *bc;
  };
  std::vector<View> vw;
  Level* lc = has_coarse ? &level(l - 1) : nullptr;
  vw.push_back(View{E0, m.A.view, has_coarse ? lc->mg->P.view : DevCsr(), has_coarse ? lc->mg->PT.view : DevCsr(), Y_.p, m.ev0.p,
                    m.ev1.p, m.elbuf.p, m.x.p, m.r.p, m.d0.p, has_coarse ? lc->mg->x.p : nullptr, has_coarse ? lc->mg->b.p : nullptr});
  for (int r = 1; r < nrot; ++r) {
    auto q = std::make_unique<Set>();
    auto dup = [&](DevBuf<double>& dst, const double* src, size_t cnt) {
      dst.alloc(cnt);
      if (cnt) hip_check(hipMemcpyAsync(dst.p, src, cnt * sizeof(double), hipMemcpyDeviceToDevice, ctx_.stream), "dup");
    };
    dup(q->Bvals, lv.B.view.vals, lv.B.view.nnz);
    dup(q->Y, Y_.p, (size_t)n_ * nY);
    dup(q->v, m.ev0.p, N);
    q->out.alloc(N);
    q->elbuf.alloc(m.elbuf.n);
    dup(q->x, m.ev0.p, N);
    dup(q->r, m.ev0.p, N);
    dup(q->d, m.ev0.p, N);
    dup(q->Avals, m.A.view.vals, m.A.view.nnz);
    View V = vw[0];
    V.E.vals = q->Bvals.p;
    V.A.vals = q->Avals.p;
    V.Y = q->Y.p;
    V.v = q->v.p;
    V.out = q->out.p;
    V.elbuf = q->elbuf.p;
    V.x = q->x.p;
    V.r = q->r.p;
    V.d = q->d.p;
    if (has_coarse) {
      q->P.upload(P_host_[l - 1]);
      q->PT.upload(transpose(P_host_[l - 1]));
      q->xc.alloc(lc->plan.N);
      hip_check(hipMemsetAsync(q->xc.p, 0, q->xc.n * sizeof(double), ctx_.stream), "memset");
      q->bc.alloc(lc->plan.N);
      V.P = q->P.view;
      V.PT = q->PT.view;
      V.xc = q->xc.p;
      V.bc = q->bc.p;
    }
    vw.push_back(V);
    sets.push_back(std::move(q));
  }
  hipEvent_t e0, e1;
  hip_check(hipEventCreate(&e0), "event");
  hip_check(hipEventCreate(&e1), "event");
  auto timeit = [&](auto&& fn) {
    for (int r = 0; r < nrot; ++r) fn(vw[r]);
    hip_check(hipEventRecord(e0, ctx_.stream), "rec");
    for (int r = 0; r < reps; ++r) fn(vw[r % nrot]);
    hip_check(hipEventRecord(e1, ctx_.stream), "rec");
    hip_check(hipEventSynchronize(e1), "evsync");
    float ms = 0;
    hip_check(hipEventElapsedTime(&ms, e0, e1), "elapsed");
    return (double)ms / reps;
  };
  MgKernelTimes kt{};
  kt.ms[0] = timeit([&](View& V) {
    MgEpi e;
    e.mode = MG_PLAIN;
    e.n = N;
    e.v = V.v;
    e.out = V.out;
    launch_elop_apply(ctx_.stream, V.E, P_, V.Y, V.elbuf, e);
  });
  kt.ms[1] = timeit([&](View& V) {
    MgEpi e;
    e.mode = MG_STEP;
    e.n = N;
    e.v = V.d;
    e.x = V.x;
    e.r = V.r;
    e.dinv = m.dinv.p;
    e.coef = m.coef.p;
    e.has_next = 1;
    e.k = 1;
    e.d_new = V.out;
    launch_elop_apply(ctx_.stream, V.E, P_, V.Y, V.elbuf, e);
  });
  kt.ms[2] = timeit([&](View& V) {
    MgEpi e;
    e.mode = MG_PLAIN;
    e.n = N;
    e.v = V.v;
    e.out = V.out;
    launch_csr_apply(ctx_.stream, V.A, e);
  });
  if (has_coarse) {
    kt.ms[3] = timeit([&](View& V) { launch_spmv(ctx_.stream, V.P, V.xc, V.x, V.x); });
    kt.ms[4] = timeit([&](View& V) { launch_spmv(ctx_.stream, V.PT, V.r, nullptr, V.bc); });
  }
  hip_check(hipStreamSynchronize(ctx_.stream), "sync");
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  const double nnzB = lv.B.view.nnz, nel = E0.nel, cm = E0.cmax;
  // matrix-free H v: values once + column ids + class ids + Y + element results written and read + their gather lists + v, out
  const double hv = nnzB * 8 + nel * cm * 4 + nel * 4 + (double)n_ * nY * 8 + nel * cm * (8 + 8 + 4) + (N + 1.0) * 4 + 2.0 * N * 8;
  kt.bytes[0] = hv;
  kt.bytes[1] = hv + 5.0 * N * 8;      // + x, r read and written, dinv read, d_new written (out counted in hv)
  kt.bytes[2] = (double)m.A.view.nnz * 12 + (N + 1.0) * 4 + 2.0 * N * 8;
  // SURVEY.md section 8(d), literally: a sweep on the matrix-free H = "2 x apply_D-class passes + 3 vectors", an apply_D-class pass =
  // nnz * 12 + (rows + 1) * 4 + cols * 8 + rows * 16 on the CSR operand (what KC_APPLY is priced at everywhere else)
  const double pass = nnzB * 12 + ((double)lv.B.view.rows + 1) * 4 + N * 8.0 + (double)lv.B.view.rows * 16;
  kt.alg[0] = 2.0 * pass;
  kt.alg[1] = 2.0 * pass + 3.0 * N * 8;
  kt.alg[2] = kt.bytes[2];
  // the unfused sequence: Dz = B v (write n K), u = Y Dz (read n K + Y, write n K), g = B' u (read n K)
  // the unfused sequence with the tightest CSR accounting: B and B' once each, Y, the two vectors, Dz and u written and read
  kt.bytes[5] = nnzB * 24 + ((double)lv.B.view.rows + N + 2) * 4 + (double)n_ * nY * 8 + 2.0 * N * 8 + 4.0 * n_ * P_.K * 8;
  if (has_coarse) {
    const DevCsr& Pv = lc->mg->P.view;
    kt.bytes[3] = kt.alg[3] = (double)Pv.nnz * 12 + (Pv.rows + 1.0) * 4 + Pv.cols * 8.0 + Pv.rows * 16.0;
    const DevCsr& PTv = lc->mg->PT.view;
    kt.bytes[4] = kt.alg[4] = (double)PTv.nnz * 12 + (PTv.rows + 1.0) * 4 + PTv.cols * 8.0 + PTv.rows * 8.0;
  }
  return kt;
}

bool Amg::pcg_solve_linear(int l, const double* s_host, const double* g_host, double* x_host, int* iters, double* relres) {
  Level& lv = level(l);
  mg_prepare(l);
  eval_Y_at(lv, s_host);
  mg_values(l);
  lv.g_trial.upload(g_host, lv.plan.N);
  hip_check(hipStreamSynchronize(ctx_.stream), "sync");
  seq_expected_ = *h_seq_.p;      //  an earlier failed call must not leave the host counter ahead of the device's
  const bool ok = pcg_run(lv, l, lv.g_trial.p, lv.nstep.p, nullptr, iters, relres);
  hip_check(hipStreamSynchronize(ctx_.stream), "sync pcg");
  lv.nstep.download(x_host, lv.plan.N);
  int fail = 0;
  hip_check(hipMemcpy(&fail, mg_fail_.p, sizeof(int), hipMemcpyDeviceToHost), "D2H mg flag");
  if (fail) {
    hip_check(hipMemset(mg_fail_.p, 0, sizeof(int)), "memset mg flag");
    throw NumericError("pcg: the coarsest-level matrix is not positive definite");
  }
  return ok;
}

}  // namespace mgb
