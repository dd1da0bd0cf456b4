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

// out-of-line on the device: the three selections and the minimizer marking are called from several places, and
// inlining them all multiplies the kernel's code (and its compile time) for nothing
#ifdef __HIPCC__
#define CP_HDN __host__ __device__ __attribute__((noinline))
#else
#define CP_HDN inline
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
#ifdef CP_SEED_PROF
    unsigned long long prof_t[8], prof_last;
#endif
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
CP_HDN int cp_kmer_hash(const char *seq, int j, int K)
{ uint64_t fh = 0, rh = 0;
  for (int i = 0; i < K; i++)
    { fh = cp_nt_srol(fh) ^ cp_nt_seed((unsigned char)seq[j+i]);
      rh = cp_nt_srol(rh) ^ cp_nt_seed_rc((unsigned char)seq[j+K-1-i]);
    }
  return (int)((rh < fh ? rh : fh) % CP_SEED_MOD);
}

// ---- the same hash from a register window: the bases of up to CP_SEED_WIN consecutive read positions, packed 2 bits
//      each (+ 1 bit "is A/C/G/T/U"), fetched with a dozen independent aligned word loads instead of 2K dependent byte
//      loads per k-mer.  The window must lie inside the read's bases [0,rlen); it is loaded from the 4-byte aligned
//      address at or below seq+j0, so up to 3 bytes before the read may be touched (they belong to the same buffer:
//      the previous read, or the buffer's own alignment slack is never needed because reads start at offset >= 0 of it).
#define CP_SEED_WINW 12                 // words
#define CP_SEED_WIN  (4*CP_SEED_WINW-3) // usable positions after a start offset of 0..3
struct cp_seed_win
  { uint32_t code[3], ok[2]; int off; };

CP_HDM void cp_seed_win_load(cp_seed_win &w, const char *seq, int j0, int avail)      // avail: bases of the read from j0 on
{ const uintptr_t a = (uintptr_t)(seq+j0);
  const int off0 = (int)(a & 3);
  w.off = off0;
  const uint32_t *p = (const uint32_t *)(a-(uintptr_t)w.off);
  const int nw = (w.off+avail+3) >> 2;                     // words that hold bases of the read
  uint32_t v[CP_SEED_WINW];
#ifdef __HIPCC__
#pragma unroll
#endif
  for (int k = 0; k < CP_SEED_WINW; k++) v[k] = (k < nw) ? p[k] : 0u;
  w.code[0] = w.code[1] = w.code[2] = 0; w.ok[0] = w.ok[1] = 0;
#ifdef __HIPCC__
#pragma unroll
#endif
  for (int q = 0; q < 4*CP_SEED_WINW; q++)
    { unsigned c = (v[q >> 2] >> (8*(q & 3))) & 0xffu;
      if (q < off0 || q-off0 >= avail) c = 0;               // bytes around the read's bases are not looked at
      // forward seed class of seedTab (nthash.h:26-59): A a 4 5 -> 0, C c 7 -> 1, G g 3 -> 2, T t U u 1 -> 3
      unsigned cd = 0, good = 1;
      switch (c)
        { case 'A': case 'a': case 4: case 5: cd = 0; break;
          case 'C': case 'c': case 7: cd = 1; break;
          case 'G': case 'g': case 3: cd = 2; break;
          case 'T': case 't': case 'U': case 'u': case 1: cd = 3; break;
          default: good = 0;
        }
      // the complement goes by the low three bits of the byte (seedTab[c & 7]): keep them too when they matter,
      // i.e. for a byte that is no A/C/G/T/U letter but whose low bits name one (IUPAC letters such as Y, K, M)
      if (!good && ((c & 7u) == 1 || (c & 7u) == 3 || (c & 7u) == 4 || (c & 7u) == 5 || (c & 7u) == 7)) good = 2;
      w.code[q >> 4] |= cd << (2*(q & 15));
      if (good == 1) w.ok[q >> 5] |= 1u << (q & 31);
      if (good == 2) w.off |= 0x100;                       // an odd byte in the window: use the byte-wise hash
    }
}
CP_HDM bool cp_seed_win_plain(const cp_seed_win &w) { return (w.off & 0x100) == 0; }

CP_HDM int cp_kmer_hash_win(const cp_seed_win &w, int rel, int K)                      // k-mer starting `rel` positions into the window
{ const uint64_t SA = 0x3c8bfbb395c60474ull, SC = 0x3193c18562a02b4cull, SG = 0x20323ed082572324ull, ST = 0x295549f54be24456ull;
  uint64_t fh = 0, rh = 0;
  const int o = (w.off & 3)+rel;
  for (int i = 0; i < K; i++)
    { const int qa = o+i, qb = o+K-1-i;
      const unsigned ca = ((qa < 16 ? w.code[0] : qa < 32 ? w.code[1] : w.code[2]) >> (2*(qa & 15))) & 3u;
      const unsigned cb = ((qb < 16 ? w.code[0] : qb < 32 ? w.code[1] : w.code[2]) >> (2*(qb & 15))) & 3u;
      const bool ga = ((qa < 32 ? w.ok[0] : w.ok[1]) >> (qa & 31)) & 1u, gb = ((qb < 32 ? w.ok[0] : w.ok[1]) >> (qb & 31)) & 1u;
      const uint64_t sa = ca == 0 ? SA : ca == 1 ? SC : ca == 2 ? SG : ST;
      const uint64_t sb = cb == 0 ? ST : cb == 1 ? SG : cb == 2 ? SC : SA;          // complement
      fh = cp_nt_srol(fh) ^ (ga ? sa : 0);
      rh = cp_nt_srol(rh) ^ (gb ? sb : 0);
    }
  return (int)((rh < fh ? rh : fh) % CP_SEED_MOD);
}

// canonical hashes of `cnt` (<= 8) consecutive k-mers starting `rel` positions into the window: one fold over the first
// k-mer, then the rolling update of NTC64_c (nthash.h:238-267) with the K-fold rotated seeds computed on the spot
CP_HDM void cp_kmer_hashes_win(const cp_seed_win &w, int K, int cnt, int *h8)
{ const uint64_t SA = 0x3c8bfbb395c60474ull, SC = 0x3193c18562a02b4cull, SG = 0x20323ed082572324ull, ST = 0x295549f54be24456ull;
  const int o = (w.off & 3);
  auto code = [&](int q) -> unsigned { return ((q < 16 ? w.code[0] : q < 32 ? w.code[1] : w.code[2]) >> (2*(q & 15))) & 3u; };
  auto good = [&](int q) -> bool { return ((q < 32 ? w.ok[0] : w.ok[1]) >> (q & 31)) & 1u; };
  auto fw = [&](int q) -> uint64_t { const unsigned c = code(q); return good(q) ? (c == 0 ? SA : c == 1 ? SC : c == 2 ? SG : ST) : 0; };
  auto rc = [&](int q) -> uint64_t { const unsigned c = code(q); return good(q) ? (c == 0 ? ST : c == 1 ? SG : c == 2 ? SC : SA) : 0; };
  uint64_t fh = 0, rh = 0;
  for (int i = 0; i < K; i++)
    { fh = cp_nt_srol(fh) ^ fw(o+i);
      rh = cp_nt_srol(rh) ^ rc(o+K-1-i);
    }
  h8[0] = (int)((rh < fh ? rh : fh) % CP_SEED_MOD);
  for (int q = 1; q < cnt; q++)
    { uint64_t fo = fw(o+q-1), ri = rc(o+q+K-1);              // base leaving the forward hash / entering the reverse one,
      for (int i = 0; i < K; i++) { fo = cp_nt_srol(fo); ri = cp_nt_srol(ri); }     // rotated K times (msTab of nthash.h)
      fh = cp_nt_srol(fh) ^ fw(o+q+K-1) ^ fo;
      rh = rh ^ ri ^ rc(o+q-1);
      rh = (rh >> 1) | (rh << 63);                              // ror1 + swapbits3263, nthash.h:186-213
      { const uint64_t x = ((rh >> 32) ^ (rh >> 63)) & 1; rh ^= (x << 32) | (x << 63); }
      h8[q] = (int)((rh < fh ? rh : fh) % CP_SEED_MOD);
    }
}

// minimum hash over the k-mers [b,e) and a mark on every k-mer that attains it (seed.c:392-397)
CP_HDN void cp_seed_mark_minimizers(cp_seed_read &R, int b, int e, bool rep)
{ const int K = R.K, rlen = R.plen+K-1;
  const char mark_rep = 'R';
  if (e-b == 1) { R.state[b] = rep ? mark_rep : R.cls[b]; return; }
  int span = CP_SEED_WIN-(K-1);                               // k-mers one window serves
  if (span > 8) span = 8;
  if (span > 0 && e-b <= span)                                // a short segment: one window, hashed once
    { cp_seed_win w;
      cp_seed_win_load(w,R.seq,b,rlen-b);
      if (cp_seed_win_plain(w))
        { int h8[8], mh = CP_SEED_MOD;
          cp_kmer_hashes_win(w,K,e-b,h8);
          for (int q = 0; q < e-b; q++) if (h8[q] < mh) mh = h8[q];
          for (int q = 0; q < e-b; q++) if (h8[q] == mh) R.state[b+q] = rep ? mark_rep : R.cls[b+q];
          return;
        }
    }
  int mh = CP_SEED_MOD;
  for (int pass = 0; pass < 2; pass++)                        // minimum first, marks second
    for (int j0 = b; j0 < e; )
      { int cnt = e-j0 < span ? e-j0 : span;
        bool fast = span > 0;
        cp_seed_win w;
        if (fast) { cp_seed_win_load(w,R.seq,j0,rlen-j0); fast = cp_seed_win_plain(w); }
        int h8[8];
        if (fast) cp_kmer_hashes_win(w,K,cnt,h8);
        else { cnt = 1; h8[0] = cp_kmer_hash(R.seq,j0,K); }     // a byte nthash.h treats unevenly: the byte-wise hash
        for (int q = 0; q < cnt; q++)
          { if (pass == 0) { if (h8[q] < mh) mh = h8[q]; }
            else if (h8[q] == mh) R.state[j0+q] = rep ? mark_rep : R.cls[j0+q];
          }
        j0 += cnt;
      }
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
CP_HDN int cp_seed_mi_add(cp_seed_read &R, int M, int b, int e)
{ const int idx = cp_seed_mi_find(R,M,b,e);
  if (idx < 0)
    { M++;
      R.mi_b[M] = b; R.mi_e[M] = e;                                // the new interval waits one slot past the sorted part
      // the reference now sorts slots [0,M) (stable, by begin).  Slots [0,M-1) are in order already (the list is
      // only ever changed by this function: an insertion keeps the order, a merge lowers a begin no further than
      // the end of its left neighbour), so the sort moves one element: slot M-1, the interval that waited there.
      if (M >= 2)
        { const int xb = R.mi_b[M-1], xe = R.mi_e[M-1];
          int lo = 0, hi = M-1;                                    // first slot whose begin is > xb (after equal begins: stable)
          while (lo < hi) { const int m = (lo+hi) >> 1; if (R.mi_b[m] > xb) hi = m; else lo = m+1; }
          int i = M-1;
          for (; i-8 >= lo; i -= 8)                                // shift up by one, eight slots per round trip to memory
            { int tb[8], te[8];
#ifdef __HIPCC__
#pragma unroll
#endif
              for (int k = 0; k < 8; k++) { tb[k] = R.mi_b[i-1-k]; te[k] = R.mi_e[i-1-k]; }
#ifdef __HIPCC__
#pragma unroll
#endif
              for (int k = 0; k < 8; k++) { R.mi_b[i-k] = tb[k]; R.mi_e[i-k] = te[k]; }
            }
          for (; i > lo; i--) { R.mi_b[i] = R.mi_b[i-1]; R.mi_e[i] = R.mi_e[i-1]; }
          R.mi_b[lo] = xb; R.mi_e[lo] = xe;
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
  int i = l+1;
  for (; i+8 <= M; i += 8)                                         // close the gap, eight slots per round trip to memory
    { int tb[8], te[8];
#ifdef __HIPCC__
#pragma unroll
#endif
      for (int k = 0; k < 8; k++) { tb[k] = R.mi_b[i+d+k]; te[k] = R.mi_e[i+d+k]; }
#ifdef __HIPCC__
#pragma unroll
#endif
      for (int k = 0; k < 8; k++) { R.mi_b[i+k] = tb[k]; R.mi_e[i+k] = te[k]; }
    }
  for (; i < M; i++) { R.mi_b[i] = R.mi_b[i+d]; R.mi_e[i] = R.mi_e[i+d]; }
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
CP_HDN void cp_seed_select(cp_seed_read &R, int C)
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


// ===================================================================================================================
//  Fast path.  The routine above is the plain form (a lane that walks a read position by position, several times,
//  through nested data-dependent loops: every step a dependent, uncoalesced memory round trip, the lanes of a wave
//  waiting for each other's inner loops).  The form below does the same computation as ONE flat loop over the
//  positions per selection, all lanes in step:
//    * no per-position state array: the output letters themselves are the state ('E' = unmarked; the kernel prefills
//      them), the repetitive stretches are the short list of .rep intervals;
//    * segments are made by a two-state machine (inside a run of equal counts / skipping to the next valid k-mer)
//      and handed, the moment they close, to the window-count deque, which lives ON CHIP (a ring of CP_SEED_DQ
//      entries per lane in LDS);
//    * the stable order by window count comes from two 5-bit radix passes whose digit counters live on chip;
//    * the masked-interval array is zeroed only as far as it is touched.
//  A read that does not fit these bounds (deque deeper than the ring, a window count above the window) reports
//  "not done" and is redone by the plain form: results are identical either way.
// ===================================================================================================================
#ifdef CP_SEED_PROF
#define SEED_STAMP(k) do { unsigned long long t_ = wall_clock64(); R.prof_t[k] += t_-R.prof_last; R.prof_last = t_; } while (0)
#else
#define SEED_STAMP(k) ((void)0)
#endif
#define CP_SEED_PEND    8
#define CP_SEED_DQ      64          // deepest deque seen on 60x HiFi-like reads: 43 (ring entries are two 32-bit words)

struct cp_seed_fast_host                                   // on-chip part, host build: plain arrays
  { uint32_t ic[CP_SEED_DQ], be[CP_SEED_DQ]; int32_t bins[32];       // ic = segment | count << 16, be = begin << 16 | end (all < 65536)
    uint32_t &dq_ic(int s) { return ic[s]; }  uint32_t &dq_be(int s) { return be[s]; }
    int32_t &bin(int k) { return bins[k]; }
    int32_t pending[CP_SEED_PEND]; int32_t &pend(int k) { return pending[k]; }
  };

CP_HDM void cp_seed_mi_touch(cp_seed_read &R, int &hw, int idx)      // slots beyond the high-water mark read as zero
{ while (hw <= idx) { R.mi_b[hw] = 0; R.mi_e[hw] = 0; hw++; } }

CP_HDM int cp_seed_mi_add_hw(cp_seed_read &R, int M, int b, int e, int &hw)
{ cp_seed_mi_touch(R,hw,M+1);
  return cp_seed_mi_add(R,M,b,e);
}

// one selection, flat.  Returns false when the read needs the plain form.
template <class F>
CP_HDN bool cp_seed_select_fast(cp_seed_read &R, F &f, int C, int nrep, int &hw)
{ const bool rep = (C == 0);
  const int W = rep ? CP_SEED_W_REP : CP_SEED_W;
  const int plen = R.plen, Km1 = R.K-1;
  int n = 0;
  // ---- segments + window counts in one pass over the positions -------------------------------------------------
  int qf = 0, qn = 0;                                      // deque = ring slots (qf+j) % CP_SEED_DQ
  bool last_oor = false; int last_oor_pos = 0;
  int ri = 0;                                              // next .rep interval (repeat selection)
  auto valid = [&](int i, unsigned char cl, unsigned char stc) -> bool      // cl = label, stc = letter written so far at k-mer i
    { if (!rep) return cl == (unsigned char)C;
      while (ri < nrep && R.rep_pairs[2*ri+1]-Km1 <= i) ri++;
      const bool in_rep = ri < nrep && R.rep_pairs[2*ri]-Km1 <= i;
      return in_rep && stc == 'E' && cl != 'E';
    };
  bool ok = true;
  // The deque's entries are addressed by their running number p (qf .. qf+qn-1).  They live in the on-chip ring
  // (slot p % CP_SEED_DQ) until the deque gets deeper than the ring; from then on, for the rest of this selection,
  // in the read's HBM scratch (R.dq / R.order are idle until the sort), slot p.
  bool spilled = false;
  auto q_ic = [&](int p) -> uint32_t { return spilled ? (uint32_t)R.dq[p] : f.dq_ic(p % CP_SEED_DQ); };
  auto q_be = [&](int p) -> uint32_t { return spilled ? (uint32_t)R.order[p] : f.dq_be(p % CP_SEED_DQ); };
  auto feed = [&](int sb, int se, int sc)                  // segment n = [sb,se) with count sc (-1: skipped stretch)
    { if (n >= R.cap) { R.overflow = 1; ok = false; return; }
      if (n >= 65535) { ok = false; return; }              // deque entries hold 16-bit segment numbers
      R.seg_b[n] = (int32_t)(((uint32_t)sb << 16) | (uint32_t)se);     // begin and end in one word (both < 65536 here)
      R.seg_nw[n] = (sc < 0) ? -10 : 0;
      if (sc >= 0)
        { if (qn > 0)
            { const int fc = (int)(q_ic(qf) >> 16);
              if (rep ? (sc < fc) : (sc > fc))
                { last_oor = false;
                  for (int j = 0; j < qn; j++)
                    { const uint32_t ic = q_ic(qf+j);
                      const int c = (int)(ic >> 16);
                      int v;
                      if (c == fc) { v = sb-(int)(q_be(qf+j) >> 16); if (v > W) v = W; }
                      else v = rep ? (CP_SEED_W_REP-c > 0 ? CP_SEED_W_REP-c : 0) : c;
                      R.seg_nw[ic & 0xffffu] = v;
                    }
                  qn = 0;
                }
            }
          while (qn > 0)
            { const uint32_t ic = q_ic(qf+qn-1);
              const int c = (int)(ic >> 16);
              if (!(rep ? (sc < c) : (sc > c))) break;
              R.seg_nw[ic & 0xffffu] = rep ? (CP_SEED_W_REP-c > 0 ? CP_SEED_W_REP-c : 0) : c;
              qn--;
            }
          if (!spilled && qn >= CP_SEED_DQ)                // deeper than the ring: move the deque to HBM
            { for (int j = 0; j < qn; j++)
                { R.dq[qf+j] = (int32_t)f.dq_ic((qf+j) % CP_SEED_DQ); R.order[qf+j] = (int32_t)f.dq_be((qf+j) % CP_SEED_DQ); }
              spilled = true;
            }
          const uint32_t nic = (uint32_t)n | ((uint32_t)sc << 16), nbe = ((uint32_t)sb << 16) | (uint32_t)se;
          if (spilled) { R.dq[qf+qn] = (int32_t)nic; R.order[qf+qn] = (int32_t)nbe; }
          else { f.dq_ic((qf+qn) % CP_SEED_DQ) = nic; f.dq_be((qf+qn) % CP_SEED_DQ) = nbe; }
          qn++;
        }
      while (qn > 0 && (int)(q_be(qf) >> 16) <= sb-W)
        { const uint32_t ic = q_ic(qf), be = q_be(qf);
          const int fb = (int)(be >> 16);
          int v = W;
          if (last_oor) { v = fb-last_oor_pos+1; if (v > W) v = W; }
          R.seg_nw[ic & 0xffffu] = v;
          if (qn > 1)
            { const int c1 = (int)(q_ic(qf+1) >> 16), c0 = (int)(ic >> 16);
              if (rep ? (c0 < c1) : (c0 > c1)) last_oor_pos = (int)(be & 0xffffu);
            }
          qf++; qn--;
          last_oor = true;
        }
      n++;
    };
  if (plen >= 2)
    { int b = 0;
      bool run = valid(0,(unsigned char)R.cls[0],(unsigned char)R.state[0]);   // inside a run of equal counts (else: skipping to a valid k-mer)
      int prev = R.prof[0];
      // eight positions per step: one 16-byte load of counts, one 8-byte load of labels (and of the letters written so
      // far, for the repeat selection) per lane instead of two or three strided 1-2 byte loads per position
      for (int e0 = 1; e0 < plen && ok; e0 += 8)
        { uint16_t c8[8]; unsigned char l8[8], s8[8];
          const int nn = plen-e0 < 8 ? plen-e0 : 8;
          if (nn == 8)
            { __builtin_memcpy(c8,R.prof+e0,16);
              __builtin_memcpy(l8,R.cls+e0,8);
              if (rep) __builtin_memcpy(s8,R.state+e0,8);
            }
          else
            for (int k = 0; k < nn; k++) { c8[k] = R.prof[e0+k]; l8[k] = (unsigned char)R.cls[e0+k]; s8[k] = (unsigned char)R.state[e0+k]; }
#ifdef __HIPCC__
#pragma unroll
#endif
          for (int k = 0; k < 8; k++)
            if (k < nn && ok)
              { const int e = e0+k, cur = c8[k];
                if (run)
                  { if (cur != prev) { feed(b,e,prev); b = e; run = valid(e,l8[k],s8[k]); } }
                else if (valid(e,l8[k],s8[k])) { feed(b,e,-1); b = e; run = true; }
                prev = cur;
              }
        }
      if (ok && b < plen-1) feed(b,plen,run ? prev : -1);
    }
  if (!ok) return false;
  SEED_STAMP(0);
  while (qn > 0)                                           // end of the read; both selections compare with `>` here
    { const uint32_t ic = q_ic(qf), be = q_be(qf);
      int v = W;
      if (last_oor) { v = (int)(be >> 16)-last_oor_pos+1; if (v > W) v = W; }
      R.seg_nw[ic & 0xffffu] = v;
      if (qn > 1 && (int)(ic >> 16) > (int)(q_ic(qf+1) >> 16)) last_oor_pos = (int)(be & 0xffffu);
      qf++; qn--;
      last_oor = true;
    }
  // ---- skipped stretches are masked from the start ---------------------------------------------------------------
  int M = 0, nbig = 0;
  for (int i = 0; i < n; i++)
    { const int nw = R.seg_nw[i];
      if (nw == -10)
        { cp_seed_mi_touch(R,hw,M);
          const uint32_t be = (uint32_t)R.seg_b[i];
          R.mi_b[M] = (int)(be >> 16); R.mi_e[M] = (int)(be & 0xffffu); M++;
        }
      else if (nw < 0) return false;                       // cannot happen (a window count is a distance or a count)
      else if (nw > 1000) nbig++;
    }
  cp_seed_mi_touch(R,hw,0);
  SEED_STAMP(1);
  if (M > 0 && R.mi_b[0] == 0 && R.mi_e[0] == plen) return true;
  // ---- stable order by decreasing window count: two 5-bit radix passes on 1000 - count (the skipped stretches,
  //      -10, come last), digit counters on chip.  The second pass also writes the records in their new order
  //      (obe[] = begin/end, onw[] = window count), so the selection below reads them front to back.
  int32_t *obe = R.seg_e, *onw = R.seg_cnt;
  for (int pass = 0; pass < 2; pass++)
    { for (int k = 0; k < 32; k++) f.bin(k) = 0;
      for (int q = 0; q < n; q++)
        { const int nw = R.seg_nw[pass ? R.dq[q] : q];
          f.bin(((nw > 1000 ? 0 : 1000-nw) >> (5*pass)) & 31)++;
        }
      int acc = 0;
      for (int k = 0; k < 32; k++) { const int c = f.bin(k); f.bin(k) = acc; acc += c; }
      for (int q = 0; q < n; q++)
        { const int i = pass ? R.dq[q] : q;
          const int nw = R.seg_nw[i];
          const int d = f.bin(((nw > 1000 ? 0 : 1000-nw) >> (5*pass)) & 31)++;
          if (pass == 0) R.dq[d] = i;
          else { R.order[d] = i; obe[d] = R.seg_b[i]; onw[d] = nw; }
        }
    }
  if (nbig > 0)                                            // counts above 1000 share the first radix key with 1000: put that
    { int m = 0;                                           // head group in order by insertion (rare, short)
      while (m < n && onw[m] >= 1000) m++;
      for (int a = 1; a < m; a++)
        { const int x = R.order[a], xb = obe[a], xn = onw[a];
          int j = a-1;
          while (j >= 0 && onw[j] < xn) { R.order[j+1] = R.order[j]; obe[j+1] = obe[j]; onw[j+1] = onw[j]; j--; }
          R.order[j+1] = x; obe[j+1] = xb; onw[j+1] = xn;
        }
    }
  SEED_STAMP(2);
  auto take = [&](uint32_t be)
    { const int b = (int)(be >> 16), e = (int)(be & 0xffffu);
      M = cp_seed_mi_add_hw(R,M,b-W > 0 ? b-W : 0,e+W < plen ? e+W : plen,hw);
      cp_seed_mark_minimizers(R,b,e,rep);
    };
  int i = 0;
  for (; i < n; i++)                                       // extreme over a whole window
    { if (onw[i] < W) break;
      take((uint32_t)obe[i]);
    }
  SEED_STAMP(3);
  // then groups of equal window count, while uncovered.  One flat loop over the remaining records (the lanes of a wave
  // stay in step): a segment is tested against the list as it was before its group began; the members that are not
  // inside wait in a short pending list (LDS) and are taken when the group ends.
  { int g = 0x7fffffff, npend = 0, gstart = i;
    bool over = false, done = false;                         // over: more members pending than the list holds
    auto flush = [&](int gend)
      { if (over)
          { for (int q = gstart; q < gend; q++) if (R.order[q] & 0x40000000) take((uint32_t)obe[q]); }
        else
          for (int q = 0; q < npend; q++) take((uint32_t)f.pend(q));
        npend = 0; over = false;
      };
    for (int ii = i; ii < n && !done; ii++)
      { const int nw = onw[ii];
        if (nw != g)
          { if (ii > i)
              { SEED_STAMP(4);
                flush(ii);
                SEED_STAMP(6);
                if (M > 0 && R.mi_b[0] == 0 && R.mi_e[0] == plen) { done = true; break; }
              }
            g = nw; gstart = ii;
          }
        cp_seed_mi_touch(R,hw,M);
        const uint32_t be = (uint32_t)obe[ii];
        const int sb = (int)(be >> 16), se = (int)(be & 0xffffu);
        const int idx = cp_seed_mi_find(R,M,sb,se);
        const bool inside = idx >= 0 && R.mi_b[idx] <= sb && se <= R.mi_e[idx];
        if (!inside)
          { R.order[ii] |= 0x40000000;
            if (npend < CP_SEED_PEND) f.pend(npend++) = (int32_t)be; else over = true;
          }
      }
    if (!done) flush(n);
  }
  SEED_STAMP(4);
  return true;
}

// the whole path, flat.  R.state must hold 'E' at every k-mer on entry.  Returns the number of .rep intervals, or -1
// when the read has to be redone by cp_find_seeds_read (after which R.state is whatever this attempt left: the plain
// form rewrites all of it).
template <class F>
CP_HDM int cp_find_seeds_fast(cp_seed_read &R, F &f)
{ const int plen = R.plen, Km1 = R.K-1;
  if (plen <= 0) return 0;
  if (plen > 65535) return -1;                             // deque entries hold 16-bit positions: the plain form takes the read
  const int min_uniq = (int)(R.K*2.5);
  // unique / repetitive stretches -> .rep intervals (seed.c:482-566), one pass
  int nrep = 0, rs = -1;                                   // rs: start of the open repetitive run
  { int i0 = 0, normal = 0; bool inR = (R.cls[0] == 'R');
    if (inR) rs = 0; else normal = (R.cls[0] == 'H' || R.cls[0] == 'D');
    for (int e = 1; e <= plen; e++)
      { const char c = e < plen ? R.cls[e] : 'R';          // a virtual R closes the last stretch
        if (inR)
          { if (c != 'R') { inR = false; i0 = e; normal = (c == 'H' || c == 'D'); } }
        else if (c == 'R')
          { if (normal >= min_uniq)                        // [i0,e) is unique: it ends the open repetitive run
              { if (rs >= 0 && rs < i0)
                  { if (nrep < R.rep_cap) { R.rep_pairs[2*nrep] = rs+Km1; R.rep_pairs[2*nrep+1] = i0+Km1; } else R.overflow = 1;
                    nrep++;
                  }
                rs = e;
              }
            else if (rs < 0) rs = i0;
            inR = true;
          }
        else normal += (c == 'H' || c == 'D');
      }
    if (rs >= 0 && rs < plen)
      { if (nrep < R.rep_cap) { R.rep_pairs[2*nrep] = rs+Km1; R.rep_pairs[2*nrep+1] = plen+Km1; } else R.overflow = 1;
        nrep++;
      }
  }
  if (R.overflow) return -1;
  SEED_STAMP(5);
  int hw = 0;
  if (!cp_seed_select_fast(R,f,'H',nrep,hw)) return -1;
  if (!cp_seed_select_fast(R,f,'D',nrep,hw)) return -1;
  if (!cp_seed_select_fast(R,f,0,nrep,hw)) return -1;
  return nrep;
}
