/*
 * classpro_oracle_seed.c -- TEST INFRASTRUCTURE ONLY (part of the CPU oracle, see classpro_oracle.c).
 *
 * Plain-C sequential restatement of the reference's `-s` seed path for one read:
 *     find_seeds (src/seed.c:966-1032) = anno_repeat (:482-592) + kmer_hash (:28-55, ntHash from src/nthash.h)
 *     + _find_seeds for 'H' and 'D' (:190-476) + _find_seeds_rep (:667-951) + the final relabelling (:1007-1015).
 * Pinned against the reference's own seed.c compiled where it lies (oracle/ref_driver.c -> ref_find_seeds;
 * tests/test_oracle_golden.py, tests/golden/seeds.npz).
 *
 * Defined behaviour where the reference reads stale memory: `mintvl` is searched and sorted one slot past its
 * live part (seed.c:141 `bs_mintvl(mintvl,0,M,...)`, :161-166 `M++; mintvl[M] = ...; qsort(mintvl,M,...)`), so the
 * reference's result depends on what an earlier read of the same thread left in that array (SURVEY hazard 6).
 * Here, and in ref_find_seeds, the array is all zeros at the start of every read; inside a read the three passes
 * see each other's leftovers exactly as the reference's code does (the array is restated slot for slot).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "classpro_oracle.h"

#define WSIZE 1000                 /* seed.c:23 */
#define WSIZE_REP 200              /* seed.c:24 */
#define BOUNDARY_UNIQ_LEN 2000     /* seed.c:25 */
#define MODV 2147483647            /* seed.c:26 */
#define SMIN(a,b) ((a) < (b) ? (a) : (b))
#define SMAX(a,b) ((a) > (b) ? (a) : (b))

typedef struct { int b, e, cnt, nw, is_seed; } seg_t;          /* ClassPro.h:235-241 */
typedef struct { int seg_id, b, e, cnt; } hmer_t;              /* ClassPro.h:244-249 */
typedef struct { int b, e; } mintvl_t;                         /* ClassPro.h:255-258 */

/* ---- ntHash, canonical, 64 bit (nthash.h:20-24 seeds, :181-268) ---------------------------------------- */
static const uint64_t SEED_A = 0x3c8bfbb395c60474ULL, SEED_C = 0x3193c18562a02b4cULL,
                      SEED_G = 0x20323ed082572324ULL, SEED_T = 0x295549f54be24456ULL;

static uint64_t seed_fw(unsigned char c)                       /* seedTab, nthash.h:26-59: every non-zero entry of the table */
{ switch (c)
    { case 'A': case 'a': case 4: case 5: return SEED_A;
      case 'C': case 'c': case 7:         return SEED_C;
      case 'G': case 'g': case 3:         return SEED_G;
      case 'T': case 't': case 'U': case 'u': case 1: return SEED_T;
    }
  return 0;
}
static uint64_t seed_rc(unsigned char c) { return seed_fw((unsigned char)(c & 0x07)); }   /* cpOff, nthash.h:17: complement */

static uint64_t srol(uint64_t v)                               /* rol1 + swapbits033, nthash.h:181-207 */
{ v = (v << 1) | (v >> 63);
  uint64_t x = (v ^ (v >> 33)) & 1;
  return v ^ (x | (x << 33));
}
static uint64_t sror(uint64_t v)                               /* ror1 + swapbits3263, nthash.h:186-213 */
{ v = (v >> 1) | (v << 63);
  uint64_t x = ((v >> 32) ^ (v >> 63)) & 1;
  return v ^ ((x << 32) | (x << 63));
}
static uint64_t srol_n(uint64_t v, int n) { while (n-- > 0) v = srol(v); return v; }   /* msTab31l|msTab33r[c][k] */

/* kmer_hash, seed.c:28-55: hash[i] of the k-mer ending at read position i+K-1, canonical value mod 2^31-1 */
void cpo_kmer_hash(const char *seq, int plen, int K, int *hash)
{ uint64_t fh = 0, rh = 0;
  for (int i = 0; i < K; i++)                                   /* NTF64_a / NTR64_a, nthash.h:215-235 */
    { fh = srol(fh) ^ seed_fw((unsigned char)seq[i]);
      rh = srol(rh) ^ seed_rc((unsigned char)seq[K-1-i]);
    }
  hash[0] = (int)((rh < fh ? rh : fh) % MODV);
  for (int i = 1; i < plen; i++)                                /* NTC64_c, nthash.h:238-267 */
    { unsigned char out = (unsigned char)seq[i-1], in = (unsigned char)seq[i+K-1];
      fh = srol(fh) ^ seed_fw(in) ^ srol_n(seed_fw(out),K);
      rh = sror(rh ^ srol_n(seed_rc(in),K) ^ seed_rc(out));
      hash[i] = (int)((rh < fh ? rh : fh) % MODV);
    }
}

/* ---- stable sorts (glibc's qsort is a merge sort at these sizes: SURVEY hazard 4) ---------------------- */
static void sort_segs_by_nw_desc(seg_t *v, int n)              /* compare_cprofile, seed.c:112-114 */
{ seg_t *tmp = (seg_t *)malloc(sizeof(seg_t)*(size_t)(n > 0 ? n : 1));
  for (int w = 1; w < n; w *= 2)
    { for (int lo = 0; lo < n; lo += 2*w)
        { int mid = SMIN(lo+w,n), hi = SMIN(lo+2*w,n), a = lo, b = mid, o = lo;
          while (a < mid && b < hi) tmp[o++] = (v[b].nw > v[a].nw) ? v[b++] : v[a++];
          while (a < mid) tmp[o++] = v[a++];
          while (b < hi) tmp[o++] = v[b++];
        }
      memcpy(v,tmp,sizeof(seg_t)*(size_t)n);
    }
  free(tmp);
}
static void sort_mintvl_by_b(mintvl_t *v, int n)               /* compare_intvl, seed.c:116-118; insertion = stable */
{ for (int i = 1; i < n; i++)
    { mintvl_t x = v[i];
      int j = i-1;
      while (j >= 0 && v[j].b > x.b) { v[j+1] = v[j]; j--; }
      v[j+1] = x;
    }
}

/* ---- masked-interval list, seed.c:120-188 (slot for slot, including the slot past the live part) --------- */
static int does_ovlp(int ab, int ae, int bb, int be) { return SMAX(ab,bb) <= SMIN(ae-1,be-1); }
static int bs_mintvl(const mintvl_t *v, int l, int r, int b, int e)
{ while (l <= r)
    { int m = (l+r)/2;
      if (does_ovlp(v[m].b,v[m].e,b,e)) return m;
      if (v[m].b < b) l = m+1; else r = m-1;
    }
  return -1;
}
static int is_contained(const mintvl_t *v, int M, int b, int e)
{ int idx = bs_mintvl(v,0,M,b,e);
  return idx != -1 && v[idx].b <= b && e <= v[idx].e;
}
static int add_intvl(mintvl_t *v, int M, int b, int e)
{ int idx = bs_mintvl(v,0,M,b,e);
  if (idx == -1)
    { M++;
      v[M].b = b; v[M].e = e;
      sort_mintvl_by_b(v,M);
      return M;
    }
  int l = idx-1;
  while (l >= 0 && does_ovlp(v[l].b,v[l].e,b,e)) l--;
  l++;
  int r = idx+1;
  while (r < M && does_ovlp(v[r].b,v[r].e,b,e)) r++;
  r--;
  v[l].b = SMIN(v[l].b,b);
  v[l].e = SMAX(v[r].e,e);
  if (l == r) return M;
  int d = r-l;
  M -= d;
  for (int i = l+1; i < M; i++) v[i] = v[i+d];
  return M;
}

/* ---- compress_profile (seed.c:61-110) and compress_profile_rep (:599-665): `valid(i)` differs ---------- */
static int compress(const uint16_t *profile, const char *cls, const int *sasgn, seg_t *cp, int plen, char C)
{
#define VALID(i) (C ? (cls[i] == C) : (sasgn[i] <= -10 && cls[i] != 'E'))
  int N = 0, b = 0, e = 1;
  int prev_valid = VALID(0);
  while (e < plen)
    { if (!prev_valid)
        { while (e < plen && !VALID(e)) e++;
          cp[N].b = b; cp[N].e = e; cp[N].cnt = -1; cp[N].nw = -10; cp[N].is_seed = 0;
          N++;
          b = e; e++;
          prev_valid = 1;
        }
      else
        { while (e < plen && profile[e] == profile[e-1]) e++;
          cp[N].b = b; cp[N].e = e; cp[N].cnt = profile[e-1]; cp[N].nw = 0; cp[N].is_seed = 0;
          N++;
          b = e; e++;
          prev_valid = (b < plen) ? VALID(b) : 0;          /* the reference reads class[plen] here; its value is never used */
        }
    }
#undef VALID
  return N;
}

/* ---- _find_seeds (seed.c:190-476; C = 'H' or 'D') and _find_seeds_rep (:667-951; C = 0) ------------------ */
static void find_seeds_pass(const uint16_t *profile, const char *cls, const int *hash, int *sasgn,
                            seg_t *cp, hmer_t *Q, mintvl_t *mintvl, int plen, char C)
{ const int rep = (C == 0);
  const int W = rep ? WSIZE_REP : WSIZE;
  const int mark = rep ? -3 : -2;
  int N = compress(profile,cls,sasgn,cp,plen,C);
  int qf = 0, qn = 0;                                           /* deque = Q[qf .. qf+qn) */
#define BETTER(a,b) (rep ? ((a) < (b)) : ((a) > (b)))           /* maximizers for H/D, minimizers in repeats */
#define LOSER_NW(c) (rep ? SMAX(WSIZE_REP-(c),0) : (c))          /* seed.c:245,254 / :723,732 ("ad-hoc") */
  int last_oor = 0, last_oor_pos = 0;
  for (int i = 0; i < N; i++)
    { seg_t seg = cp[i];
      if (seg.cnt >= 0)
        { hmer_t now = { i, seg.b, seg.e, seg.cnt };
          if (qn > 0)
            { hmer_t first = Q[qf];
              if (BETTER(now.cnt,first.cnt))                    /* every element is wiped out (seed.c:232-249) */
                { last_oor = 0;
                  for (int j = 0; j < qn; j++)
                    { hmer_t el = Q[qf+j];
                      if (first.cnt == el.cnt) cp[el.seg_id].nw = SMIN(now.b-el.b,W);
                      else                     cp[el.seg_id].nw = LOSER_NW(el.cnt);
                    }
                  qn = 0;
                }
            }
          while (qn > 0)
            { hmer_t el = Q[qf+qn-1];
              if (BETTER(now.cnt,el.cnt)) { cp[el.seg_id].nw = LOSER_NW(el.cnt); qn--; }
              else break;
            }
          Q[qf+qn] = now; qn++;
        }
      if (qn == 0) continue;
      while (qn > 0 && Q[qf].b <= seg.b-W)                      /* out of range (seed.c:265-288) */
        { hmer_t first = Q[qf];
          cp[first.seg_id].nw = last_oor ? SMIN(first.b-last_oor_pos+1,W) : W;
          if (qn > 1 && BETTER(first.cnt,Q[qf+1].cnt)) last_oor_pos = first.e;
          qf++; qn--;
          last_oor = 1;
        }
    }
  while (qn > 0)                                                /* seed.c:303-324 / :789-810: here BOTH passes test `>` */
    { hmer_t first = Q[qf];
      cp[first.seg_id].nw = last_oor ? SMIN(first.b-last_oor_pos+1,W) : W;
      if (qn > 1 && first.cnt > Q[qf+1].cnt) last_oor_pos = first.e;
      qf++; qn--;
      last_oor = 1;
    }
#undef BETTER
#undef LOSER_NW

  int M = 0;                                                    /* invalid segments (seed.c:346-356) */
  for (int i = 0; i < N; i++)
    if (cp[i].cnt == -1) { mintvl[M].b = cp[i].b; mintvl[M].e = cp[i].e; M++; }
  if (M > 0 && mintvl[0].b == 0 && mintvl[0].e == plen) return;
  sort_segs_by_nw_desc(cp,N);                                   /* seed.c:367 */
#define SEED_SEG(s) do { M = add_intvl(mintvl,M,SMAX(0,(s).b-W),SMIN((s).e+W,plen));                   \
                         int mh_ = MODV;                                                                  \
                         for (int j_ = (s).b; j_ < (s).e; j_++) mh_ = SMIN(hash[j_],mh_);                \
                         for (int j_ = (s).b; j_ < (s).e; j_++) if (hash[j_] == mh_) sasgn[j_] = mark;   \
                       } while (0)
  int i = 0;
  for (; i < N; i++)                                            /* seed.c:378-409 */
    { seg_t seg = cp[i];
      if (seg.nw < W) break;
      SEED_SEG(seg);
      cp[i].is_seed = 1;
    }
  while (i < N)                                                 /* seed.c:410-452 */
    { int ii;
      for (ii = i; ii < N && cp[i].nw == cp[ii].nw; ii++)
        if (!is_contained(mintvl,M,cp[ii].b,cp[ii].e)) cp[ii].is_seed = 1;
      for (ii = i; ii < N && cp[i].nw == cp[ii].nw; ii++)
        if (cp[ii].is_seed) { seg_t seg = cp[ii]; SEED_SEG(seg); }
      if (M > 0 && mintvl[0].b == 0 && mintvl[0].e == plen) break;
      i = ii;
    }
#undef SEED_SEG
}

/* ---- anno_repeat, seed.c:482-592: sasgn = -10 / 0 / -11; repeat intervals in read coordinates ------------- */
static int anno_repeat(int *sasgn, const char *cls, int plen, int K, int *rep_pairs, int rep_cap)
{ const int MIN_UNIQ_LEN = (int)(K*2.5);
  for (int i = 0; i < plen; i++) sasgn[i] = -10;
  int b = 0, e;
  int in_R = (cls[0] == 'R');
  int n_normal = (cls[0] == 'H' || cls[0] == 'D') ? 1 : 0;
  for (e = 1; e < plen; e++)
    { if (in_R)
        { if (cls[e] != 'R') { b = e; in_R = 0; n_normal = (cls[e] == 'H' || cls[e] == 'D') ? 1 : 0; } }
      else
        { if (cls[e] == 'R')
            { if (n_normal >= MIN_UNIQ_LEN) for (int i = b; i < e; i++) sasgn[i] = 0;
              in_R = 1;
            }
          else if (cls[e] == 'H' || cls[e] == 'D') n_normal++;
        }
    }
  if (!in_R && n_normal >= MIN_UNIQ_LEN)
    for (int i = b; i < e; i++) sasgn[i] = 0;
  int n = 0;                                                    /* the .rep track's intervals (seed.c:531-566) */
  in_R = (sasgn[0] == -10);
  b = K-1;
  for (int i = 1; i < plen; i++)
    { if (!in_R && sasgn[i] == -10) { b = i+K-1; in_R = 1; }
      if (in_R && sasgn[i] != -10)
        { if (n < rep_cap) { rep_pairs[2*n] = b; rep_pairs[2*n+1] = i+K-1; }
          n++; in_R = 0;
        }
    }
  if (in_R)
    { if (n < rep_cap) { rep_pairs[2*n] = b; rep_pairs[2*n+1] = plen+K-1; }
      n++;
    }
  int l = BOUNDARY_UNIQ_LEN;                                    /* seed.c:573-583 */
  while (l < plen && sasgn[l] == -10) l++;
  int r = plen-BOUNDARY_UNIQ_LEN;
  while (r >= 0 && sasgn[r] == -10) r--;
  for (int i = l; i < r; i++)
    if (sasgn[i] == -10) sasgn[i] = -11;
  return n;
}

/* find_seeds, seed.c:966-1032.  seq[plen+K-1], cls[plen] in {E,H,D,R}; sasgn[plen] receives 'E'/'H'/'D'/'R'
 * (what the .class.data track carries under -s, ClassPro.c:293); returns the number of repeat-mask intervals
 * (pairs in read coordinates, what the .rep.data track carries). */
int cpo_find_seeds(const char *seq, const char *cls, const uint16_t *profile, int plen, int K,
                   int *sasgn, int *rep_pairs, int rep_cap)
{ seg_t *cp = (seg_t *)malloc(sizeof(seg_t)*(size_t)(plen+1));
  hmer_t *Q = (hmer_t *)malloc(sizeof(hmer_t)*(size_t)(plen+1));
  mintvl_t *mintvl = (mintvl_t *)calloc((size_t)plen+2,sizeof(mintvl_t));      /* zeros at read start: defined behaviour */
  int *hash = (int *)malloc(sizeof(int)*(size_t)(plen+1));
  int nrep = anno_repeat(sasgn,cls,plen,K,rep_pairs,rep_cap);
  cpo_kmer_hash(seq,plen,K,hash);
  find_seeds_pass(profile,cls,hash,sasgn,cp,Q,mintvl,plen,'H');
  find_seeds_pass(profile,cls,hash,sasgn,cp,Q,mintvl,plen,'D');
  find_seeds_pass(profile,cls,hash,sasgn,cp,Q,mintvl,plen,0);
  for (int i = 0; i < plen; i++)
    sasgn[i] = (sasgn[i] == -2) ? cls[i] : (sasgn[i] == -3) ? 'R' : 'E';
  free(cp); free(Q); free(mintvl); free(hash);
  return nrep;
}
