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

// _lctx[i][t] of ClassPro.c:136-141 (index = read position)
template <class SEQ>
CP_HD int cp_lctx(const SEQ &seq, int rlen, int i, int t)
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

// rctx[j][t] of ClassPro.c:137,142
template <class SEQ>
CP_HD int cp_rctx(const SEQ &seq, int rlen, int j, int t)
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

// Profile-indexed views of ClassPro.c:138-142: ctx[DROP][i] = _lctx[i+K-2], ctx[GAIN][i] = rctx[i].
template <class SEQ>
CP_HD int cp_ctx(const SEQ &seq, int rlen, int K, int w, int i, int t)
{ return (w == CP_DROP) ? cp_lctx(seq,rlen,i+K-2,t) : cp_rctx(seq,rlen,i,t); }
