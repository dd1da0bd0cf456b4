// cp_seed.h -- the `-s` seed path of the reference (src/seed.c:966-1032, find_seeds) for ONE read, as a
// sequential host/device routine run by one lane of the seed kernel (kernels.hip: k_find_seeds, 64 reads per wave).
//
// What the reference computes per read, from the label string, the count profile and the bases:
//   * anno_repeat (seed.c:482-592): a k-mer position is "unique" when it lies in a maximal stretch of non-R labels
//     that holds at least 2.5 K H/D positions, otherwise "repetitive"; the repetitive stretches, in read
//     coordinates, are the intervals of the .rep mask track;
//   * three seed selections that share one masked-interval list: count MAXimizers among the H-labelled and among
//     the D-labelled k-mers (_find_seeds, seed.c:190-476, window 1000), count MINimizers among the non-E k-mers of
//     the repetitive stretches (_find_seeds_rep, seed.c:667-951, window 200).  A selection compresses the profile
//     into segments of equal count (compress_profile, seed.c:61-110 / :599-665), gives every segment the number of
//     windows in which it is the extreme one (a monotone deque), visits the segments by decreasing window count
//     and marks the sequence minimizers (canonical ntHash, src/nthash.h) of the segments it takes as seeds;
//   * the result per k-mer: 'E' (no seed) or the class of the seed ('H', 'D', or 'R' for a repeat seed),
//     seed.c:1007-1015 -- what the .class.data track carries under -s (ClassPro.c:293).
//
// Re-design rather than a transcription: the per-position state is ONE byte (the output array itself: SD_UNIQ,
// SD_REP, SD_SEED, SD_RSEED, rewritten to letters at the end) instead of an int array; the deque holds segment
// indices; segments are visited through a stable counting sort of their window counts (the reference's qsort is
// glibc's stable merge sort) instead of being moved; hashes are computed only for the positions of chosen segments
// (a fold over the k-mer, nthash.h:215-235, equal to the reference's rolling values) instead of for every k-mer.
// The masked-interval list is restated slot for slot: the reference searches and sorts one slot past the live
// part of that array (seed.c:141,161-166), so its leftovers matter; the defined behaviour is that the
// array is all zeros when a read starts.
#pragma once
#include <stdint.h>
#include "cp_types.h"

#ifndef CP_HDM
#ifdef __HIPCC__
#define CP_HDM __host__ __device__ __forceinline__
#else
#define CP_HDM inline
#endif
#endif

#define CP_SEED_W      1000        // seed.c:23 WSIZE
#define CP_SEED_W_REP  200         // seed.c:24 WSIZE_REP
#define CP_SEED_MOD    2147483647  // seed.c:26
#define CP_SEED_BINS   1012        // window counts -10..1000 and one bin for anything larger

enum { SD_UNIQ = 0, SD_REP = 10, SD_SEED = 2, SD_RSEED = 3 };   // -(sasgn) of seed.c: 0, -10/-11, -2, -3

struct cp_seed_read
  { const char     *seq;           // rlen bases
    const char     *cls;           // plen labels (E/H/D/R): d_labels + K-1
    const uint16_t *prof;          // plen counts
    int             plen, K;
    char           *state;         // plen bytes: working state, then the result letters
    // scratch of capacity `cap` segments (cap >= count runs + label runs + 4)
    int32_t *seg_b, *seg_e, *seg_cnt, *seg_nw;
    int32_t *dq;                   // deque of segment indices
    int32_t *order;                // segments by decreasing window count (stable)
    int32_t *bins;                 // CP_SEED_BINS+1 counters
    int32_t *mi_b, *mi_e;          // masked intervals, cap+3 slots
    int32_t *rep_pairs; int rep_cap;
    int      cap;
    int      overflow;
  };

// ---- canonical ntHash of the k-mer starting at read position j, mod 2^31-1 (seed.c:28-55) -----------------
CP_HDM uint64_t cp_nt_seed(unsigned c)                             // seedTab, nthash.h:26-59: A C G T/U in both cases (and the codes 1..7); else 0
{ switch (c)
    { case 'A': case 'a': case 4: case 5: return 0x3c8bfbb395c60474ull;
      case 'C': case 'c': case 7:         return 0x3193c18562a02b4cull;
      case 'G': case 'g': case 3:         return 0x20323ed082572324ull;
      case 'T': case 't': case 'U': case 'u': case 1: return 0x295549f54be24456ull;
    }
  return 0;
}
CP_HDM uint64_t cp_nt_seed_rc(unsigned c)                          // seedTab[c & cpOff], nthash.h:17,226-235: the complement by the low three bits
{ switch (c & 7u)
    { case 1: return 0x295549f54be24456ull;                        // ..001  A a        -> T
      case 3: return 0x20323ed082572324ull;                        // ..011  C c        -> G
      case 4: case 5: return 0x3c8bfbb395c60474ull;                // ..10x  T t U u    -> A
      case 7: return 0x3193c18562a02b4cull;                        // ..111  G g        -> C
    }
  return 0;
}
CP_HDM uint64_t cp_nt_srol(uint64_t v)                             // rol1 + swapbits033, nthash.h:181-207
{ v = (v << 1) | (v >> 63);
  const uint64_t x = (v ^ (v >> 33)) & 1;
  return v ^ (x | (x << 33));
}
CP_HDM int cp_kmer_hash(const char *seq, int j, int K)
{ uint64_t fh = 0, rh = 0;
  for (int i = 0; i < K; i++)
    { fh = cp_nt_srol(fh) ^ cp_nt_seed((unsigned char)seq[j+i]);
      rh = cp_nt_srol(rh) ^ cp_nt_seed_rc((unsigned char)seq[j+K-1-i]);
    }
  return (int)((rh < fh ? rh : fh) % CP_SEED_MOD);
}

// ---- unique / repetitive stretches and the .rep intervals (seed.c:482-566) ---------------------------------
// Returns the number of repeat intervals; pairs (b,e) in read coordinates go to rep_pairs.
CP_HDM int cp_seed_anno_repeat(cp_seed_read &R)
{ const int plen = R.plen, K = R.K;
  const int min_uniq = (int)(K*2.5);
  const char *cls = R.cls;
  char *st = R.state;
  // maximal non-R stretches; one with >= min_uniq H/D positions is unique, everything else repetitive
  int i = 0;
  while (i < plen)
    { if (cls[i] == 'R') { st[i++] = SD_REP; continue; }
      int j = i, normal = 0;
      while (j < plen && cls[j] != 'R') { normal += (cls[j] == 'H' || cls[j] == 'D'); j++; }
      const char v = (normal >= min_uniq) ? (char)SD_UNIQ : (char)SD_REP;
      for (int q = i; q < j; q++) st[q] = v;
      i = j;
    }
  // repetitive runs in read coordinates (the first K-1 bases of the read belong to the first run's k-mer)
  int n = 0;
  i = 0;
  while (i < plen)
    { if (st[i] != SD_REP) { i++; continue; }
      int j = i;
      while (j < plen && st[j] == SD_REP) j++;
      if (n < R.rep_cap) { R.rep_pairs[2*n] = i+K-1; R.rep_pairs[2*n+1] = j+K-1; }
      else R.overflow = 1;
      n++;
      i = j;
    }
  return n;
}

// ---- segments of one selection (seed.c:61-110 with C = 'H'/'D'; seed.c:599-665 with C = 0) ----------------
// valid(i): the k-mer takes part in this selection.  A valid segment is a run of equal counts that STARTS at a
// valid k-mer (it may run on over k-mers of other classes); k-mers skipped between segments form invalid ones
// (count -1).  A segment that would start at the last k-mer is never made (the reference's loop ends first).
CP_HDM bool cp_seed_valid(const cp_seed_read &R, int i, int C)
{ return C ? (R.cls[i] == C) : (R.state[i] == SD_REP && R.cls[i] != 'E'); }

CP_HDM int cp_seed_segments(cp_seed_read &R, int C)
{ const int plen = R.plen;
  int n = 0, b = 0;
  while (b < plen-1)
    { int e = b+1, cnt;
      if (cp_seed_valid(R,b,C) || (n > 0 && R.seg_cnt[n-1] < 0))      // after an invalid segment the next one is taken as valid
        { while (e < plen && R.prof[e] == R.prof[e-1]) e++;
          cnt = R.prof[e-1];
        }
      else
        { while (e < plen && !cp_seed_valid(R,e,C)) e++;
          cnt = -1;
        }
      if (n >= R.cap) { R.overflow = 1; return n; }
      R.seg_b[n] = b; R.seg_e[n] = e; R.seg_cnt[n] = cnt; R.seg_nw[n] = (cnt < 0) ? -10 : 0;
      n++;
      b = e;
    }
  return n;
}

// ---- window counts: in how many windows of W k-mers is a segment the extreme count (seed.c:218-324 / :694-810) ----
// A monotone deque of segment indices; a segment's count is fixed when it leaves the deque.
CP_HDM void cp_seed_window_counts(cp_seed_read &R, int n, bool rep)
{ const int W = rep ? CP_SEED_W_REP : CP_SEED_W;
  int qf = 0, qn = 0;
  bool last_oor = false;
  int last_oor_pos = 0;
  // `beats(a,b)`: count a displaces count b (larger for H/D maximizers, smaller for repeat minimizers);
  // a displaced segment that was not the front gets its count (H/D) or W_REP - count (repeats) as a stand-in
  for (int i = 0; i < n; i++)
    { const int sb = R.seg_b[i], sc = R.seg_cnt[i];
      if (sc >= 0)
        { if (qn > 0)
            { const int fc = R.seg_cnt[R.dq[qf]];
              if (rep ? (sc < fc) : (sc > fc))                      // the whole deque goes
                { last_oor = false;
                  for (int j = 0; j < qn; j++)
                    { const int s = R.dq[qf+j], c = R.seg_cnt[s];
                      if (c == fc) { const int d = sb-R.seg_b[s]; R.seg_nw[s] = d < W ? d : W; }
                      else R.seg_nw[s] = rep ? (CP_SEED_W_REP-c > 0 ? CP_SEED_W_REP-c : 0) : c;
                    }
                  qn = 0;
                }
            }
          while (qn > 0)
            { const int s = R.dq[qf+qn-1], c = R.seg_cnt[s];
              if (!(rep ? (sc < c) : (sc > c))) break;
              R.seg_nw[s] = rep ? (CP_SEED_W_REP-c > 0 ? CP_SEED_W_REP-c : 0) : c;
              qn--;
            }
          R.dq[qf+qn] = i; qn++;
        }
      while (qn > 0 && R.seg_b[R.dq[qf]] <= sb-W)                    // the front has left the window
        { const int s = R.dq[qf];
          int v = W;
          if (last_oor) { v = R.seg_b[s]-last_oor_pos+1; if (v > W) v = W; }
          R.seg_nw[s] = v;
          if (qn > 1)
            { const int c1 = R.seg_cnt[R.dq[qf+1]], c0 = R.seg_cnt[s];
              if (rep ? (c0 < c1) : (c0 > c1)) last_oor_pos = R.seg_e[s];
            }
          qf++; qn--;
          last_oor = true;
        }
    }
  while (qn > 0)                                                   // end of the read; both selections compare with `>` here
    { const int s = R.dq[qf];
      int v = W;
      if (last_oor) { v = R.seg_b[s]-last_oor_pos+1; if (v > W) v = W; }
      R.seg_nw[s] = v;
      if (qn > 1 && R.seg_cnt[s] > R.seg_cnt[R.dq[qf+1]]) last_oor_pos = R.seg_e[s];
      qf++; qn--;
      last_oor = true;
    }
}

// ---- the masked-interval list, slot for slot (seed.c:120-188) --------------------------------------------------
CP_HDM bool cp_seed_ovlp(int ab, int ae, int bb, int be)
{ const int lo = ab > bb ? ab : bb, hi = (ae < be ? ae : be)-1; return lo <= hi; }

CP_HDM int cp_seed_mi_find(const cp_seed_read &R, int M, int b, int e)      // slots 0..M, M included
{ int l = 0, r = M;
  while (l <= r)
    { const int m = (l+r)/2;
      if (cp_seed_ovlp(R.mi_b[m],R.mi_e[m],b,e)) return m;
      if (R.mi_b[m] < b) l = m+1; else r = m-1;
    }
  return -1;
}
CP_HDM int cp_seed_mi_add(cp_seed_read &R, int M, int b, int e)
{ const int idx = cp_seed_mi_find(R,M,b,e);
  if (idx < 0)
    { M++;
      R.mi_b[M] = b; R.mi_e[M] = e;                                // the new interval waits one slot past the sorted part
      for (int i = 1; i < M; i++)                                  // stable sort of slots [0,M) by begin: nearly sorted
        { const int xb = R.mi_b[i], xe = R.mi_e[i];
          int j = i-1;
          while (j >= 0 && R.mi_b[j] > xb) { R.mi_b[j+1] = R.mi_b[j]; R.mi_e[j+1] = R.mi_e[j]; j--; }
          R.mi_b[j+1] = xb; R.mi_e[j+1] = xe;
        }
      return M;
    }
  int l = idx-1;
  while (l >= 0 && cp_seed_ovlp(R.mi_b[l],R.mi_e[l],b,e)) l--;
  l++;
  int r = idx+1;
  while (r < M && cp_seed_ovlp(R.mi_b[r],R.mi_e[r],b,e)) r++;
  r--;
  if (b < R.mi_b[l]) R.mi_b[l] = b;
  R.mi_e[l] = R.mi_e[r] > e ? R.mi_e[r] : e;
  if (l == r) return M;
  const int d = r-l;
  M -= d;
  for (int i = l+1; i < M; i++) { R.mi_b[i] = R.mi_b[i+d]; R.mi_e[i] = R.mi_e[i+d]; }
  return M;
}

// take segment s as a seed segment: mask it with a margin of W and mark its hash minimizers
CP_HDM int cp_seed_take(cp_seed_read &R, int M, int s, int W, char mark)
{ const int b = R.seg_b[s], e = R.seg_e[s];
  M = cp_seed_mi_add(R,M,b-W > 0 ? b-W : 0,e+W < R.plen ? e+W : R.plen);
  int mh = CP_SEED_MOD;
  for (int j = b; j < e; j++) { const int h = cp_kmer_hash(R.seq,j,R.K); if (h < mh) mh = h; }
  for (int j = b; j < e; j++) if (cp_kmer_hash(R.seq,j,R.K) == mh) R.state[j] = mark;
  return M;
}

// ---- one selection (seed.c:190-476 / :667-951) ---------------------------------------------------------------
CP_HDM void cp_seed_select(cp_seed_read &R, int C)
{ const bool rep = (C == 0);
  const int W = rep ? CP_SEED_W_REP : CP_SEED_W;
  const char mark = rep ? (char)SD_RSEED : (char)SD_SEED;
  const int plen = R.plen;
  const int n = cp_seed_segments(R,C);
  if (R.overflow) return;
  cp_seed_window_counts(R,n,rep);
  int M = 0;                                                       // the skipped stretches are masked from the start
  for (int i = 0; i < n; i++)
    if (R.seg_cnt[i] < 0) { R.mi_b[M] = R.seg_b[i]; R.mi_e[M] = R.seg_e[i]; M++; }
  if (M > 0 && R.mi_b[0] == 0 && R.mi_e[0] == plen) return;
  // stable order by decreasing window count: counting sort over -10..1000, larger values in one last bin
  // that is put in order by insertion (counts above 1000 are rare)
  for (int k = 0; k <= CP_SEED_BINS; k++) R.bins[k] = 0;
  for (int i = 0; i < n; i++)
    { int k = R.seg_nw[i]+10; if (k > CP_SEED_BINS-1) k = CP_SEED_BINS-1; R.bins[k]++; }
  { int acc = 0;
    for (int k = CP_SEED_BINS-1; k >= 0; k--) { const int c = R.bins[k]; R.bins[k] = acc; acc += c; }
  }
  const int nbig = (CP_SEED_BINS >= 2) ? R.bins[CP_SEED_BINS-2] : 0;   // size of the "larger than 1000" bin
  for (int i = 0; i < n; i++)
    { int k = R.seg_nw[i]+10; if (k > CP_SEED_BINS-1) k = CP_SEED_BINS-1; R.order[R.bins[k]++] = i; }
  for (int i = 1; i < nbig; i++)
    { const int x = R.order[i], xn = R.seg_nw[x];
      int j = i-1;
      while (j >= 0 && R.seg_nw[R.order[j]] < xn) { R.order[j+1] = R.order[j]; j--; }
      R.order[j+1] = x;
    }
  int i = 0;
  for (; i < n; i++)                                               // every segment that is extreme over a whole window
    { const int s = R.order[i];
      if (R.seg_nw[s] < W) break;
      M = cp_seed_take(R,M,s,W,mark);
    }
  while (i < n)                                                    // then groups of equal window count, while uncovered
    { const int g = R.seg_nw[R.order[i]];
      int ii = i;
      while (ii < n && R.seg_nw[R.order[ii]] == g) ii++;
      // the segments of a group are tested against the list as it was before the group
      for (int q = i; q < ii; q++)
        { const int s = R.order[q];
          const int idx = cp_seed_mi_find(R,M,R.seg_b[s],R.seg_e[s]);
          const bool inside = idx >= 0 && R.mi_b[idx] <= R.seg_b[s] && R.seg_e[s] <= R.mi_e[idx];
          R.dq[q-i] = inside ? 0 : 1;                              // the deque is idle here: per-group flags
        }
      for (int q = i; q < ii; q++)
        if (R.dq[q-i]) M = cp_seed_take(R,M,R.order[q],W,mark);
      if (M > 0 && R.mi_b[0] == 0 && R.mi_e[0] == plen) break;
      i = ii;
    }
}

// ---- the whole path for one read (seed.c:966-1032).  Returns the number of .rep intervals. ------------------
CP_HDM int cp_find_seeds_read(cp_seed_read &R)
{ const int plen = R.plen;
  if (plen <= 0) return 0;
  for (int i = 0; i < R.cap+3; i++) { R.mi_b[i] = 0; R.mi_e[i] = 0; }     // defined start state of the list
  const int nrep = cp_seed_anno_repeat(R);
  cp_seed_select(R,'H');
  cp_seed_select(R,'D');
  cp_seed_select(R,0);
  for (int i = 0; i < plen; i++)
    { const char s = R.state[i];
      R.state[i] = (s == SD_SEED) ? R.cls[i] : (s == SD_RSEED) ? 'R' : 'E';
    }
  return nrep;
}
