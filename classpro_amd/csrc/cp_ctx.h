// cp_ctx.h -- per-base low-complexity context evaluated ON DEMAND from the read bases.
//
// The reference's calc_seq_context (src/context.c:8-108) makes one sequential pass and materialises
// 6 bytes per base (lctx/rctx x {HP,DS,TS}).  find_wall / find_rel_intvl only ever look at these at
// wall candidates (~1 % of positions), so the HIP path never stores them: it evaluates the same values
// in closed form from a short window of bases around the queried position.
//
//   lctx[i][HP] = length of the homopolymer run ending at i                       (context.c:17,22)
//   lctx[i][DS] = 0 inside a homopolymer, else 1 + floor(r2/2), r2 = number of consecutive
//                 p = i, i-1, ... with seq[p-2] == seq[p]                          (context.c:18-30)
//   lctx[i][TS] = 0 if seq[i-2..i] are equal, else 1 + floor(r3/3), r3 = number of consecutive
//                 p = i, i-1, ... with seq[p-3] == seq[p]                          (context.c:42-50)
//   rctx        = the mirror images (run *starting* at the position); the reference fills them by
//                 copying reversed lctx runs when a run closes                    (context.c:24-25,33-39,52-58,62-84)
// every value capped at 127.  The homopolymer mirror reproduces the reference's reversed-copy of
// *capped* values for runs longer than 127 (cells it never writes read as 0, as in the oracle).
// Equality with the sequential reference is checked cell-for-cell in tests/test_context.py.
#pragma once
#include "cp_types.h"

#ifndef __HIPCC__
#undef  CP_HD
#define CP_HD static inline
#endif

// _lctx[i][t] of ClassPro.c:136-141 (index = read position), base by base
template <class SEQ>
CP_HD int cp_lctx_scan(const SEQ &seq, int rlen, int i, int t)
{ (void)rlen;
  if (t == CP_HP)
    { int n = 1;
      while (n < 127 && i-n >= 0 && seq[i-n] == seq[i]) n++;
      return n;
    }
  if (t == CP_DS)
    { if (i == 0) return 0;                          // preset, ClassPro.c:140
      if (seq[i-1] == seq[i]) return 0;
      int r = 0;
      for (int p = i; p >= 2 && seq[p-2] == seq[p] && r < 252; p--) r++;
      return 1+r/2;
    }
  // TS
  if (i < 2) return 0;                               // presets, ClassPro.c:140
  if (seq[i-2] == seq[i-1] && seq[i-1] == seq[i]) return 0;
  int r = 0;
  for (int p = i; p >= 3 && seq[p-3] == seq[p] && r < 378; p--) r++;
  return 1+r/3;
}

// rctx[j][t] of ClassPro.c:137,142, base by base
template <class SEQ>
CP_HD int cp_rctx_scan(const SEQ &seq, int rlen, int j, int t)
{ if (t == CP_HP)
    { int fwd = 0;                                   // equal bases after j
      while (j+fwd+1 < rlen && seq[j+fwd+1] == seq[j] && fwd < 127) fwd++;
      if (fwd >= 127) return 0;                      // run > 127 and j before its last 127 cells: never written
      int L = fwd+1;                                 // run length, walked back only as far as it matters
      for (int p = j-1; p >= 0 && seq[p] == seq[j] && L < 254; p--) L++;
      if (L <= 127) return fwd+1;
      int v = fwd-126+L;                             // reversed copy of capped lctx values (context.c:24-25)
      return v < 127 ? v : 127;
    }
  if (t == CP_DS)
    { if (j >= rlen-1) return 0;                     // context.c:84
      if (seq[j] == seq[j+1]) return 0;
      int r = 0;
      for (int p = j; p+2 < rlen && seq[p] == seq[p+2] && r < 252; p++) r++;
      return 1+r/2;
    }
  // TS
  if (j >= rlen-2) return 0;                         // context.c:84
  if (seq[j] == seq[j+1] && seq[j+1] == seq[j+2]) return 0;
  int r = 0;
  for (int p = j; p+3 < rlen && seq[p] == seq[p+3] && r < 378; p++) r++;
  return 1+r/3;
}

// ---- the same values from eight bases at a time (round 5) ------------------------------------------------------
// The scans above are loops over single bases whose trip count is the lane's own: on the device every iteration is a
// load (or an LDS read with a bounds test), a compare and a handful of scalar instructions that juggle the execution
// mask -- and they were HALF of k_wall_tasks (a diagnostic build whose contexts are constants: 3.43 -> 1.70 ms per 4 Gbases).
// A context is a count of consecutive equal pairs at distance 1, 2 or 3 from a position: with the eight bases from the
// position on in its direction as one 64-bit word D (byte k = the base k steps away), the pairs are the bytes of
// D ^ (D shifted by 1, 2 or 3 bytes) and the count is the number of trailing zero bytes -- no loop, no branch.  A run
// that reaches the end of the word (or of the bases the caller has at hand: nv of them) is left to the loops above,
// which also own every rule about the ends of the read.  cp_seq_dirword(seq,rlen,pos,dir,&D) is the customisation
// point: how many bases (0 or 8) D holds; the default knows nothing (0: always the loops), plain pointers load eight
// bytes, kernels.hip adds its LDS and register windows.
CP_HD int cp_tzbytes(uint64_t x) { return x ? (int)(__builtin_ctzll(x) >> 3) : 8; }
template <class SEQ>
CP_HD int cp_seq_dirword(const SEQ &, int, int, int, uint64_t *) { return 0; }
CP_HD int cp_seq_dirword(const char *seq, int rlen, int pos, int dir, uint64_t *D)
{ const int lo = dir > 0 ? pos : pos-7;
  if (lo < 0 || lo+8 > rlen) return 0;
  uint64_t x;
  __builtin_memcpy(&x,seq+lo,8);
  *D = dir > 0 ? x : __builtin_bswap64(x);
  return 8;
}
// matches of the bases 1, 2, ... steps away with the base itself; -1: all nv-1 of them match (undecided)
CP_HD int cp_word_hp(uint64_t D, int nv)
{ const int m = cp_tzbytes((D ^ ((D & 0xffull)*0x0101010101010101ull)) >> 8);
  return m >= nv-1 ? -1 : m;
}
// the DS / TS value (u = 2 / 3) in the direction of D; -1: undecided
CP_HD int cp_word_ds(uint64_t D, int nv)
{ if (nv < 2) return -1;
  if (((D >> 8) & 0xff) == (D & 0xff)) return 0;
  const int r = cp_tzbytes(D ^ (D >> 16));
  return r >= nv-2 ? -1 : 1+r/2;
}
CP_HD int cp_word_ts(uint64_t D, int nv)
{ if (nv < 3) return -1;
  if (((D >> 8) & 0xff) == (D & 0xff) && ((D >> 16) & 0xff) == (D & 0xff)) return 0;
  const int r = cp_tzbytes(D ^ (D >> 24));
  return r >= nv-3 ? -1 : 1+r/3;
}

template <class SEQ>
CP_HD int cp_lctx(const SEQ &seq, int rlen, int i, int t)
{ uint64_t D = 0;
  const int nv = cp_seq_dirword(seq,rlen,i,-1,&D);
  if (nv)
    { const int v = (t == CP_HP) ? cp_word_hp(D,nv) : (t == CP_DS) ? cp_word_ds(D,nv) : cp_word_ts(D,nv);
      if (v >= 0) return (t == CP_HP) ? v+1 : v;
    }
  return cp_lctx_scan(seq,rlen,i,t);
}
template <class SEQ>
CP_HD int cp_rctx(const SEQ &seq, int rlen, int j, int t)
{ uint64_t D = 0;
  const int nv = cp_seq_dirword(seq,rlen,j,+1,&D);
  if (nv)
    { if (t == CP_HP)
        { const int fwd = cp_word_hp(D,nv);                // the run is short on this side ...
          if (fwd >= 0)
            { uint64_t L = 0;
              const int nl = cp_seq_dirword(seq,rlen,j,-1,&L);
              if (nl && cp_word_hp(L,nl) >= 0) return fwd+1;   // ... and on the other: far from the 127 that change the value
            }
        }
      else
        { const int v = (t == CP_DS) ? cp_word_ds(D,nv) : cp_word_ts(D,nv);
          if (v >= 0) return v;
        }
    }
  return cp_rctx_scan(seq,rlen,j,t);
}
// all three contexts of a position at once (one word for the three)
template <class SEQ>
CP_HD void cp_lctx3(const SEQ &seq, int rlen, int i, int *l3)
{ uint64_t D = 0;
  const int nv = cp_seq_dirword(seq,rlen,i,-1,&D);
  int h = -1, d = -1, s = -1;
  if (nv) { h = cp_word_hp(D,nv); d = cp_word_ds(D,nv); s = cp_word_ts(D,nv); }
  l3[CP_HP] = h >= 0 ? h+1 : cp_lctx_scan(seq,rlen,i,CP_HP);
  l3[CP_DS] = d >= 0 ? d   : cp_lctx_scan(seq,rlen,i,CP_DS);
  l3[CP_TS] = s >= 0 ? s   : cp_lctx_scan(seq,rlen,i,CP_TS);
}
template <class SEQ>
CP_HD void cp_rctx3(const SEQ &seq, int rlen, int j, int *l3)
{ uint64_t D = 0, L = 0;
  const int nv = cp_seq_dirword(seq,rlen,j,+1,&D);
  int h = -1, d = -1, s = -1;
  if (nv)
    { h = cp_word_hp(D,nv); d = cp_word_ds(D,nv); s = cp_word_ts(D,nv);
      if (h >= 0)
        { const int nl = cp_seq_dirword(seq,rlen,j,-1,&L);
          if (!(nl && cp_word_hp(L,nl) >= 0)) h = -1;
        }
    }
  l3[CP_HP] = h >= 0 ? h+1 : cp_rctx_scan(seq,rlen,j,CP_HP);
  l3[CP_DS] = d >= 0 ? d   : cp_rctx_scan(seq,rlen,j,CP_DS);
  l3[CP_TS] = s >= 0 ? s   : cp_rctx_scan(seq,rlen,j,CP_TS);
}

// Profile-indexed views of ClassPro.c:138-142: ctx[DROP][i] = _lctx[i+K-2], ctx[GAIN][i] = rctx[i].
template <class SEQ>
CP_HD int cp_ctx(const SEQ &seq, int rlen, int K, int w, int i, int t)
{ return (w == CP_DROP) ? cp_lctx(seq,rlen,i+K-2,t) : cp_rctx(seq,rlen,i,t); }
template <class SEQ>
CP_HD void cp_ctx3(const SEQ &seq, int rlen, int K, int w, int i, int *l3)
{ if (w == CP_DROP) cp_lctx3(seq,rlen,i+K-2,l3); else cp_rctx3(seq,rlen,i,l3); }
