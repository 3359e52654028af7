// Read-level stitching of per-chunk calls: the host step the reference times as t_merge
// (/root/reference/ravvent_performance_evaluator.py:73-75 -> /root/reference/merger.py:155-248).
//
// Per chunk pair: local alignment of the last 25 merged bases with the first 25 appended bases
// (merger.py:163-172, Biopython pairwise2.align.localms / localds, first alignment of the list), a
// per-column pick by the higher per-base probability (SingleMergerByLogits, merger.py:88-119), splice
// (merger.py:236-245).  The reference rebuilds its Python string per chunk (quadratic in read length);
// here the merged read is one growing buffer and a pair costs a 25x25 DP.
//
// Bio.pairwise2 is a third-party dependency absent from the reference tree; its affine local
// alignment, including the order in which co-optimal tracebacks are produced (algns[0]), is restated
// from the published algorithm of Biopython 1.72-1.81.  Arithmetic is double like CPython's; the
// equality tests use the same rint(x*1000+0.5) buckets and the same exact comparisons.
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/ravvent_merge.h"

namespace {

constexpr int kMaxAlignments = 1000;     // pairwise2.MAX_ALIGNMENTS

struct ScoreSet {
  bool matrix;
  double match, mismatch, open, extend;
  double m[4][4];
};

bool score_set(int id, ScoreSet* s) {    // merger.py:124-147
  static const double m2[4][4] = {{10, -3, -1, -4}, {-3, 9, -5, 0}, {-1, -5, 7, -3}, {-4, 0, -3, 8}};
  memset(s, 0, sizeof *s);
  switch (id) {
    case 0: *s = ScoreSet{false, 1.0, -1.0, -1.0, -0.2, {}}; return true;
    case 1: *s = ScoreSet{false, 5.0, -4.0, -3.0, -0.1, {}}; return true;
    case 2:
      s->matrix = true; s->open = -9.0; s->extend = -2.0;
      memcpy(s->m, m2, sizeof m2);
      return true;
    default: return false;
  }
}

inline int base_index(char c) {
  switch (c) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return -1; }
}

inline long long rint1000(double x) { return (long long)(x * 1000 + 0.5); }   // pairwise2.rint: int() truncates toward zero

inline double affine(int length, double open, double extend) {   // calc_affine_penalty, penalize_extend_when_opening=False
  if (length <= 0) return 0.0;
  double p = open + extend * length;
  p -= extend;
  return p;
}

// The DP swept by anti-diagonals (cells (r, d - r)): no cell of a diagonal depends on another one of it, so the loop over r
// vectorises -- doubles, the same operations in the same order per cell as the row-major loop of the reference, hence
// bit-identical.  Everything is indexed by ROW r = 0 .. nl-1 (nl = rows rounded up to a multiple of 8: a fixed-width loop with no
// remainder; rows without a cell on the diagonal are computed and masked): h1 / h2 = scores of diagonals d-1 / d-2, e1 = the column
// state left by the cell above (diagonal d-1), f = the row state carried along each row, ga / ea = gap-open / gap-extend cost of a
// gap in A for that row (0 in the last row: free end gaps), ia = the row's letter code, ibr = B's letter codes REVERSED, so that the
// letters a diagonal meets are contiguous in r.  Scores and trace codes are stored diagonal-major, [d][r] -- what the sweep writes
// with plain vector stores -- and the traceback reads cell (r, c) at [(r + c) * nl + r].
// (Round 3: the row-major stores, per-diagonal trip counts and the gather of the first form cost 9 of the 13 us a chunk pair took
//  on a 2.1 GHz Xeon -- more than the GPU needs for the chunk.)
// MATRIX: letter codes 0..3 and mrow[k][r] = m[letter of row r][k] (four row-indexed arrays, stride mstride): the substitution score is
// picked by three selects on the column's code.  Identity scoring (the default score set): arbitrary codes, equal -> match, no table at all.
// One group of 8 rows of one diagonal.  Every array comes in as a restrict parameter and the trip count is fixed, so the compiler
// emits straight vector code: no run-time overlap checks, no scalar remainder (they were most of the instructions of the sweep
// when the diagonal was one loop over rotating buffer pointers).
template <bool MATRIX>
__attribute__((always_inline)) static inline void diag_group(
    int r0, int r_lo, int r_hi, int r_lastcol, double first_gap, double extend, double match, double mismatch,
    const double* __restrict__ h1, const double* __restrict__ h2, double* __restrict__ hn, const double* __restrict__ e1, double* __restrict__ en,
    double* __restrict__ f, const double* __restrict__ ga, const double* __restrict__ ea, const int* __restrict__ ia, const int* __restrict__ ibd,
    const double* __restrict__ m0, const double* __restrict__ m1, const double* __restrict__ m2, const double* __restrict__ m3,
    double* __restrict__ sd, int* __restrict__ td, double* __restrict__ lane_max) {
#pragma clang loop vectorize(enable) vectorize_width(8) interleave(disable)
  for (int k = 0; k < 8; ++k) {
    const int r = r0 + k;
    const bool valid = r >= r_lo && r <= r_hi;
    const int cb = ibd[r];
    const double mm = MATRIX ? (cb == 0 ? m0[r] : cb == 1 ? m1[r] : cb == 2 ? m2[r] : m3[r]) : (ia[r] == cb ? match : mismatch);
    const double nogap = h2[r - 1] + mm;
    const double row_open = h1[r] + ga[r], row_extend = f[r] + ea[r];
    const double rs = row_open > row_extend ? row_open : row_extend;
    const bool lastcol = r == r_lastcol;               // column lenB: free end gaps in B
    const double col_open = h1[r - 1] + (lastcol ? 0.0 : first_gap), col_extend = e1[r - 1] + (lastcol ? 0.0 : extend);
    const double cs = col_open > col_extend ? col_open : col_extend;
    double b = cs > rs ? cs : rs;
    b = nogap > b ? nogap : b;
    const double hv = valid ? (b < 0 ? 0.0 : b) : 0.0;
    f[r] = valid ? rs : f[r]; en[r] = cs; hn[r] = hv;
    // pairwise2.rint(x) = int(x * 1000 + 0.5); rint is monotone, so rint(max(x, y)) == max(rint(x), rint(y))
    const int ro = (int)(row_open * 1000 + 0.5), re = (int)(row_extend * 1000 + 0.5);
    const int co = (int)(col_open * 1000 + 0.5), ce = (int)(col_extend * 1000 + 0.5), ng = (int)(nogap * 1000 + 0.5);
    const int rr = ro > re ? ro : re, cr = co > ce ? co : ce;
    int br = rr > cr ? rr : cr; br = ng > br ? ng : br;
    int t = ng == br ? 2 : 0;
    t += rr == br ? (ro == rr ? 1 : 0) + (re == rr ? 8 : 0) : 0;
    t += cr == br ? (co == cr ? 4 : 0) + (ce == cr ? 16 : 0) : 0;
    sd[r] = hv; td[r] = valid ? t : -1;
    lane_max[k] = lane_max[k] > hv ? lane_max[k] : hv;   // per lane, not a reduction: rows without a cell hold 0, the border value
  }
}

template <bool MATRIX>
__attribute__((always_inline)) static inline double fill_diagonals_body(
                             int lenA, int lenB, int nl, double first_gap, double extend, double open2, const int* __restrict__ ia,
                             const int* __restrict__ ibr, const double* __restrict__ mrow, int mstride, double match, double mismatch,
                             double* __restrict__ work, double* __restrict__ score, int* __restrict__ trace, double* __restrict__ diag_max) {
  // work: 8 arrays of nl + 8 doubles; element r of an array lives at [8 + r], so that [r - 1] of row 0 is addressable
  const int ws = nl + 8;
  double* hbuf[3] = {work + 8, work + 8 + ws, work + 8 + 2 * ws};       // scores of three consecutive diagonals, rotating
  double* ebuf[2] = {work + 8 + 3 * ws, work + 8 + 4 * ws};             // column states of two
  double* f = work + 8 + 5 * ws;
  const double* ga = f + ws;
  const double* ea = ga + ws;
  const double* m0 = mrow, *m1 = mrow + mstride, *m2 = mrow + 2 * mstride, *m3 = mrow + 3 * mstride;
  double local_max = 0;
  for (int d = 2; d <= lenA + lenB; ++d) {
    const int r_lo = d - lenB > 1 ? d - lenB : 1, r_hi = lenA < d - 1 ? lenA : d - 1, r_lastcol = d - lenB;
    const double* h1 = hbuf[(d + 2) % 3];     // diagonal d-1
    const double* h2 = hbuf[(d + 1) % 3];     // diagonal d-2
    double* hn = hbuf[d % 3];
    double* e1 = ebuf[(d + 1) & 1];
    double* en = ebuf[d & 1];
    {   // column state of column d-1 before row 1 (used by cell (1, d-1)): calc_affine_penalty(d-1, 2*open, extend)
      double p = open2 + extend * (d - 1); p -= extend; e1[0] = p;
    }
    const int* ibd = ibr + (lenB - d);        // ibd[r] = code of B[d - r - 1]
    double* sd = score + (size_t)d * nl;
    int* td = trace + (size_t)d * nl;
    // rows r_lo .. r_hi hold this diagonal's cells; the sweep covers the enclosing 8-aligned window plus row r_hi + 1 (the border cell
    // (r, 0) of diagonal d = r, read as 0 by the next diagonals); rows outside it are never read again by a valid cell
    const int w_lo = r_lo & ~7, w_end = (r_hi + 9) & ~7, w_hi = w_end < nl ? w_end : nl;
    alignas(64) double lane_max[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int r0 = w_lo; r0 < w_hi; r0 += 8)
      diag_group<MATRIX>(r0, r_lo, r_hi, r_lastcol, first_gap, extend, match, mismatch, h1, h2, hn, e1, en, f, ga, ea, ia, ibd, m0, m1, m2, m3, sd, td, lane_max);
    // best score = max over the clipped cells (it starts at 0 in the reference too)
    double dm = 0;
    for (int k = 0; k < 8; ++k) dm = dm > lane_max[k] ? dm : lane_max[k];
    diag_max[d] = dm;                                      // find_start only looks at diagonals that reach the best score
    local_max = local_max > dm ? local_max : dm;
  }
  return local_max;
}

__attribute__((target_clones("avx512f", "avx2", "default")))
static double fill_diagonals(bool matrix, int lenA, int lenB, int nl, double first_gap, double extend, double open2, const int* __restrict__ ia,
                             const int* __restrict__ ibr, const double* __restrict__ mrow, int mstride, double match, double mismatch,
                             double* __restrict__ work, double* __restrict__ score, int* __restrict__ trace, double* __restrict__ diag_max) {
  return matrix ? fill_diagonals_body<true>(lenA, lenB, nl, first_gap, extend, open2, ia, ibr, mrow, mstride, match, mismatch, work, score, trace, diag_max)
                : fill_diagonals_body<false>(lenA, lenB, nl, first_gap, extend, open2, ia, ibr, mrow, mstride, match, mismatch, work, score, trace, diag_max);
}

struct Start { double score; int row, col; };
struct Item { std::string a, b; int end; bool has_end; int row, col; bool col_gap; int trace; };
struct Aligned { std::string a, b; double score; int begin, end; };

struct Aligner {
  std::string A, B;
  int lenA = 0, lenB = 0;
  ScoreSet sc;
  std::vector<double> score;     // (lenA+1) x (lenB+1)
  std::vector<int> trace;        // -1 = None (border)
  std::vector<double> dbuf, mrow, dmax;  // per-diagonal work arrays of fill(); identity match table; best score of each diagonal
  std::vector<int> tbuf, ia, ib;
  std::vector<Start> starts;
  double best = 0;
  int W = 0;

  int NL = 8;                    // row stride of the diagonal-major matrices
  double& S(int r, int c) { return score[(size_t)(r + c) * NL + r]; }
  int& T(int r, int c) { return trace[(size_t)(r + c) * NL + r]; }

  double match_fn(char a, char b) const {
    if (!sc.matrix) return a == b ? sc.match : sc.mismatch;
    return sc.m[base_index(a)][base_index(b)];
  }

  void fill() {                  // _make_score_matrix_fast: local, penalize_end_gaps (False, False); swept by anti-diagonals
    W = lenB + 1;
    NL = (lenA + 1 + 7) & ~7;
    const int nd = lenA + lenB + 1;
    // (no clearing: the sweep writes every cell and border the traceback can reach -- rows r_lo - 1 .. r_hi + 1 of each diagonal, row 0
    //  while it is a border -- and find_start only reads the valid rows of a diagonal; diagonals 0 and 1 hold borders only)
    if (score.size() < (size_t)nd * NL) { score.resize((size_t)nd * NL); trace.resize((size_t)nd * NL); }
    if (dmax.size() < (size_t)nd) dmax.resize(nd);
    for (int i = 0; i < 2 * NL; ++i) { score[i] = 0.0; trace[i] = -1; }
    dmax[0] = dmax[1] = 0.0;
    const double open = sc.open, extend = sc.extend;
    const double first_gap = affine(1, open, extend);
    const int ws = NL + 8;
    dbuf.assign((size_t)8 * ws + 8, 0.0);
    double* f = &dbuf[8 + 5 * ws], *ga = f + ws, *ea = ga + ws;
    ia.assign(NL, 0);
    ib.assign((size_t)lenB + 2 * NL + 2 * lenA + 16, 0);    // reversed letters of B with room for every row of every diagonal
    int* ibr = ib.data() + NL + lenA + 8;                   // index range used: 1 - lenA .. lenB - 2 + NL
    // letter codes.  Matrix scoring: 0..3, and mrow[k][r] = m[A[r-1]][k] for the sweep's selects.  Identity scoring on arbitrary
    // letters: the byte itself (equal bytes match).  Rows without a cell on a diagonal are masked in the sweep, whatever their codes say
    if (sc.matrix) {
      mrow.assign((size_t)4 * ws, 0.0);
      for (int i = 0; i < lenA; ++i) { const int a = base_index(A[i]); ia[i + 1] = a; for (int k = 0; k < 4; ++k) mrow[(size_t)k * ws + i + 1] = sc.m[a][k]; }
      for (int i = 0; i < lenB; ++i) ibr[lenB - 1 - i] = base_index(B[i]);
    } else {
      for (int i = 0; i < lenA; ++i) ia[i + 1] = (unsigned char)A[i];
      for (int i = 0; i < lenB; ++i) ibr[lenB - 1 - i] = (unsigned char)B[i];
    }
    for (int r = 1; r <= lenA; ++r) {
      f[r] = affine(r, 2 * open, extend);                 // row state before column 1
      ga[r] = r == lenA ? 0.0 : first_gap; ea[r] = r == lenA ? 0.0 : extend;
    }
    best = fill_diagonals(sc.matrix, lenA, lenB, NL, first_gap, extend, 2 * open, ia.data(), ibr, mrow.data(), ws, sc.match, sc.mismatch, dbuf.data(), score.data(),
                          trace.data(), dmax.data());
  }

  void find_start(std::vector<Start>& st) {
    st.clear();
    const double lim = best - 0.001;                       // cheap filter (twice the bucket width); candidates take the exact test below
    auto exact = [&](double s) { const double d = s > best ? s - best : best - s; return d * 1000 + 0.5 < 1.0; };   // rint(abs(s - best)) <= rint(0)
    if (lim > 0.0) {
      // the usual case: a positive best score.  Only the diagonals whose maximum reaches it are scanned (their valid rows); the (few)
      // candidates are then put into the row-major order in which the reference scans its matrix
      for (int d = 2; d <= lenA + lenB; ++d) {
        if (dmax[d] < lim) continue;
        const int r_lo = d - lenB > 1 ? d - lenB : 1, r_hi = lenA < d - 1 ? lenA : d - 1;
        const double* sp = score.data() + (size_t)d * NL;
        for (int r = r_lo; r <= r_hi; ++r)
          if (sp[r] >= lim && exact(sp[r])) st.push_back({sp[r], r, d - r});
      }
      std::sort(st.begin(), st.end(), [](const Start& x, const Start& y) { return x.row != y.row ? x.row < y.row : x.col < y.col; });
      return;
    }
    for (int r = 0; r <= lenA; ++r)                        // row-major order, as the reference scans its matrix
      for (int c = 0; c <= lenB; ++c) {
        const double s = S(r, c);
        if (s >= lim && exact(s)) st.push_back({s, r, c});
      }
  }

  static void rev_append(std::string& dst, const std::string& src, int from /*inclusive*/, int to /*exclusive, going down*/) {
    for (int i = from; i > to; --i) dst.push_back(src[i]);
  }

  void finish_backtrace(std::string& a, std::string& b, int row, int col) {
    if (row) rev_append(a, A, row - 1, -1);
    if (col) rev_append(b, B, col - 1, -1);
    if (row > col) b.append(a.size() - b.size(), '-');
    else if (col > row) a.append(b.size() - a.size(), '-');
  }

  // direction_col: walk left (gap in A) else up (gap in B); target = how far the border is
  bool find_gap_open(Item& it, std::vector<Item>& in_process, bool direction_col, int target) {
    bool dead_end = false;
    const double target_score = S(it.row, it.col);
    for (int n = 0; n < target; ++n) {
      if (direction_col) { it.col -= 1; it.a.push_back('-'); it.b.push_back(B[it.col]); }
      else { it.row -= 1; it.a.push_back(A[it.row]); it.b.push_back('-'); }
      const double actual = S(it.row, it.col) + affine(n + 1, sc.open, sc.extend);
      if (S(it.row, it.col) == best) { dead_end = true; break; }
      const int t = T(it.row, it.col);
      if (rint1000(actual) == rint1000(target_score) && n > 0) {
        if (t <= 0) break;
        Item br = it; br.trace = t;
        in_process.push_back(br);
      }
      if (t <= 0) dead_end = true;
    }
    return dead_end;
  }

  // first alignment of _clean_alignments(_recover_alignments(...)); reverse = second attempt on transposed matrices
  bool recover(const std::vector<Start>& starts, bool reverse, Aligned* out) {
    std::vector<Item> in_process;
    int begin = 0;
    double sc_last = 0;
    for (const Start& st : starts) {
      const double s = st.score; const int row = st.row, col = st.col;
      sc_last = s;
      begin = 0;
      bool zero_ext = false;
      for (const Start& o : starts) if (o.row == row - 1 && o.col == col - 1 && o.score == s) { zero_ext = true; break; }
      if (zero_ext) continue;
      if (s <= 0) continue;
      const int t = T(row, col);
      if (t < 0) continue;
      if ((t - t % 2) % 4 == 2) T(row, col) = 2; else continue;
      Item it;
      it.end = -std::max(lenA - row, lenB - col);
      it.has_end = it.end != 0;
      const int cd = lenB - col, rd = lenA - row;
      if (cd > rd) it.a.append(cd - rd, '-');
      rev_append(it.a, A, lenA - 1, row - 1);
      if (rd > cd) it.b.append(rd - cd, '-');
      rev_append(it.b, B, lenB - 1, col - 1);
      it.row = row; it.col = col; it.col_gap = false; it.trace = T(row, col);
      in_process.push_back(std::move(it));
    }
    int n_tracebacks = 0;
    while (!in_process.empty() && n_tracebacks < kMaxAlignments) {
      bool dead_end = false;
      Item it = std::move(in_process.back());
      in_process.pop_back();
      int trace = it.trace;
      while ((it.row > 0 || it.col > 0) && !dead_end) {
        // state before this move (pushed again when the cell has another way out); the aligned strings only
        // ever grow, so the earlier state is a prefix of the current one
        const size_t c_na = it.a.size(), c_nb = it.b.size();
        const int c_row = it.row, c_col = it.col; const bool c_gap = it.col_gap;
        if (trace <= 0) {
          if (it.col && it.col_gap) dead_end = true;
          else finish_backtrace(it.a, it.b, it.row, it.col);
          break;
        } else if (trace % 2 == 1) {
          trace -= 1;
          if (it.col_gap) dead_end = true;
          else { it.col -= 1; it.a.push_back('-'); it.b.push_back(B[it.col]); it.col_gap = false; }
        } else if (trace % 4 == 2) {
          trace -= 2;
          it.row -= 1; it.col -= 1;
          it.a.push_back(A[it.row]); it.b.push_back(B[it.col]);
          it.col_gap = false;
        } else if (trace % 8 == 4) {
          trace -= 4;
          it.row -= 1;
          it.a.push_back(A[it.row]); it.b.push_back('-');
          it.col_gap = true;
        } else if (trace == 8 || trace == 24) {
          trace -= 8;
          if (it.col_gap) dead_end = true;
          else { it.col_gap = false; dead_end = find_gap_open(it, in_process, true, it.col); }
        } else if (trace == 16) {
          trace -= 16;
          it.col_gap = true;
          dead_end = find_gap_open(it, in_process, false, it.row);
        }
        if (trace) {
          Item cache;
          cache.a.assign(it.a, 0, c_na); cache.b.assign(it.b, 0, c_nb);
          cache.end = it.end; cache.has_end = it.has_end; cache.row = c_row; cache.col = c_col; cache.col_gap = c_gap;
          cache.trace = trace;
          in_process.push_back(std::move(cache));
        }
        trace = T(it.row, it.col);
        if (S(it.row, it.col) == best) dead_end = true;
        else if (S(it.row, it.col) <= 0) { begin = std::max(it.row, it.col); trace = 0; }
      }
      if (!dead_end) {
        ++n_tracebacks;
        std::string ra(it.a.rbegin(), it.a.rend()), rb(it.b.rbegin(), it.b.rend());
        int end = it.has_end ? it.end + (int)ra.size() : (int)ra.size();     // _clean_alignments
        if (begin < end) {
          out->a = reverse ? rb : ra;
          out->b = reverse ? ra : rb;
          out->score = sc_last; out->begin = begin; out->end = end;
          return true;
        }
      }
    }
    return false;
  }

  void transpose() {             // _reverse_matrices + swapped sequences
    static const int rt[32] = {0, 4, 2, 6, 1, 5, 3, 7, 16, 20, 18, 22, 17, 21, 19, 23,
                               8, 12, 10, 14, 9, 13, 11, 15, 24, 28, 26, 30, 25, 29, 27, 31};
    const int nl2 = (lenB + 1 + 7) & ~7;                   // the transposed matrix: rows = columns of this one
    std::vector<double> s2((size_t)(lenA + lenB + 1) * nl2, 0.0);
    std::vector<int> t2(s2.size(), -1);
    for (int c = 0; c <= lenB; ++c)
      for (int r = 0; r <= lenA; ++r) {
        s2[(size_t)(r + c) * nl2 + c] = S(r, c);
        const int t = T(r, c);
        t2[(size_t)(r + c) * nl2 + c] = t < 0 ? -1 : rt[t];
      }
    score.swap(s2); trace.swap(t2);
    std::swap(A, B); std::swap(lenA, lenB);
    W = lenB + 1; NL = nl2;
  }

  // pairwise2.align.local{ms,ds}(a, b, ...)[0]; false = empty list
  bool align(const char* a, int la, const char* b, int lb, Aligned* out) {
    if (la <= 0 || lb <= 0) return false;
    A.assign(a, la); B.assign(b, lb); lenA = la; lenB = lb;
    fill();
    find_start(starts);
    if (recover(starts, false, out)) return true;
    transpose();
    for (Start& s : starts) std::swap(s.row, s.col);
    return recover(starts, true, out);
  }
};

}  // namespace

extern "C" {

int rv_local_align(const char* a, int32_t len_a, const char* b, int32_t len_b, int32_t scores_id, char* out_a, char* out_b,
                   int32_t cap, int32_t* out_len, double* score, int32_t* begin, int32_t* end) {
  Aligner al;
  if (!score_set(scores_id, &al.sc)) return RV_MERGE_EINVAL;
  if (len_a < 0 || len_b < 0 || (!a && len_a) || (!b && len_b) || !out_len) return RV_MERGE_EINVAL;
  if (al.sc.matrix) {
    for (int i = 0; i < len_a; ++i) if (base_index(a[i]) < 0) return RV_MERGE_EALPHABET;
    for (int i = 0; i < len_b; ++i) if (base_index(b[i]) < 0) return RV_MERGE_EALPHABET;
  }
  Aligned r;
  if (!al.align(a, len_a, b, len_b, &r)) { *out_len = 0; return 0; }
  if ((int)r.a.size() > cap || !out_a || !out_b) return RV_MERGE_ESPACE;
  memcpy(out_a, r.a.data(), r.a.size());
  memcpy(out_b, r.b.data(), r.b.size());
  *out_len = (int32_t)r.a.size();
  if (score) *score = r.score;
  if (begin) *begin = r.begin;
  if (end) *end = r.end;
  return 1;
}

}  // extern "C"

// Merger.merge as a resumable loop: the state between two snippets is the merged read so far, merge_flag, and
// whether the early return of merger.py:195-200 has been taken.
struct RvMerger {
  Aligner al;
  int overlap = 25;
  bool started = false, merge_flag = false, stopped = false;
  std::string seq;
  std::vector<float> lg;
  std::string m_seq;
  std::vector<float> m_lg;

  int append(const uint8_t* bases, const float* probs, const int32_t* lengths, int64_t stride, int32_t n_chunks) {
    for (int i = 0; i < n_chunks; ++i) if (lengths[i] < 0 || lengths[i] > stride) return RV_MERGE_EINVAL;
    for (int i = 0; i < n_chunks; ++i) {
      const char* sa = (const char*)bases + (size_t)i * stride;
      const float* la = probs + (size_t)i * stride;
      const int n = lengths[i];
      if (!started) { seq.assign(sa, n); lg.assign(la, la + n); started = true; continue; }   // nuc_pred_snippets[0]
      if (stopped) break;
      const int n1 = (int)std::min<size_t>(seq.size(), overlap);      // seq_merged[-overlap:]
      const int n2 = std::min(n, overlap);                             // seq_appended[:overlap]
      const char* s1 = seq.data() + seq.size() - n1;
      const float* l1 = lg.data() + lg.size() - n1;
      if (al.sc.matrix) {
        for (int k = 0; k < n1; ++k) if (base_index(s1[k]) < 0) return RV_MERGE_EALPHABET;
        for (int k = 0; k < n2; ++k) if (base_index(sa[k]) < 0) return RV_MERGE_EALPHABET;
      }
      Aligned r;
      if (!al.align(s1, n1, sa, n2, &r)) {                             // merger.py:181-200
        if (!merge_flag) { seq.assign(sa, n); lg.assign(la, la + n); continue; }
        stopped = true;
        break;
      }
      merge_flag = true;
      // align_logits + SingleMergerByLogits (merger.py:9-23, 88-119)
      m_seq.clear(); m_lg.clear();
      int i1 = 0, i2 = 0;
      for (size_t c = 0; c < r.a.size(); ++c) {
        const char c1 = r.a[c], c2 = r.b[c];
        const float v1 = c1 == '-' ? -1.f : l1[i1++];
        const float v2 = c2 == '-' ? -1.f : la[i2++];
        if (c1 == '-') { m_seq.push_back(c2); m_lg.push_back(v2); }
        else if (c2 == '-') { m_seq.push_back(c1); m_lg.push_back(v1); }
        else if (v2 > v1) { m_seq.push_back(c2); m_lg.push_back(v2); }
        else { m_seq.push_back(c1); m_lg.push_back(v1); }
      }
      // seq_merged[:-overlap] + merged + seq_appended[overlap:]   (merger.py:236-245)
      seq.resize(seq.size() - n1); lg.resize(lg.size() - n1);
      seq.append(m_seq); lg.insert(lg.end(), m_lg.begin(), m_lg.end());
      if (n > overlap) { seq.append(sa + overlap, n - overlap); lg.insert(lg.end(), la + overlap, la + n); }
    }
    return 0;
  }
};

extern "C" {

int rv_merger_create(int32_t scores_id, int32_t overlap, rv_merger* out) {
  if (!out || overlap < 1) return RV_MERGE_EINVAL;
  RvMerger* m = new RvMerger();
  if (!score_set(scores_id, &m->al.sc)) { delete m; return RV_MERGE_EINVAL; }
  m->overlap = overlap;
  *out = m;
  return 0;
}

void rv_merger_destroy(rv_merger m) { delete m; }

int rv_merger_append(rv_merger m, const uint8_t* bases, const float* probs, const int32_t* lengths, int64_t stride,
                     int32_t n_chunks) {
  if (!m || n_chunks < 0 || stride < 0 || (n_chunks && (!bases || !probs || !lengths))) return RV_MERGE_EINVAL;
  return m->append(bases, probs, lengths, stride, n_chunks);
}

int rv_merger_result(rv_merger m, uint8_t* out_seq, float* out_probs, int64_t out_cap, int64_t* out_len) {
  if (!m || !out_len) return RV_MERGE_EINVAL;
  *out_len = (int64_t)m->seq.size();
  if ((int64_t)m->seq.size() > out_cap) return RV_MERGE_ESPACE;
  if (!m->seq.empty()) {
    if (!out_seq || !out_probs) return RV_MERGE_ESPACE;
    memcpy(out_seq, m->seq.data(), m->seq.size());
    memcpy(out_probs, m->lg.data(), m->lg.size() * sizeof(float));
  }
  return 0;
}

int rv_merge_calls(const uint8_t* bases, const float* probs, const int32_t* lengths, int64_t stride, int32_t n_chunks,
                   int32_t scores_id, int32_t overlap, uint8_t* out_seq, float* out_probs, int64_t out_cap, int64_t* out_len) {
  if (n_chunks < 1 || !bases || !probs || !lengths || !out_len || overlap < 1 || stride < 0) return RV_MERGE_EINVAL;
  RvMerger m;
  if (!score_set(scores_id, &m.al.sc)) return RV_MERGE_EINVAL;
  m.overlap = overlap;
  const int rc = m.append(bases, probs, lengths, stride, n_chunks);
  if (rc != 0) return rc;
  return rv_merger_result(&m, out_seq, out_probs, out_cap, out_len);
}

}  // extern "C"
