// cp_wall.h -- scalar building blocks of wall detection (reference src/wall.c:264-958) as device
// functions.  The HIP kernels in kernels.hip orchestrate them: dense per-position passes are
// lane-parallel, the order-dependent candidate walk (wall.c:590-707) runs on one lane because each
// candidate's outcome depends on the pair/memo state left by the previous ones.
//
// Per-read state lives in HBM scratch owned by the wave that processes the read:
//   wall[plen+1]        flag byte per profile position (bit layout of wall.c:264-269)
//   perror              memoised P(error) per (position, etype, wtype), -inf = unset (wall.c:310-315).
//                       The reference keeps a dense [plen+1][2][2] array; only ~3 positions per wall
//                       candidate are ever touched, so the device keeps a small open-addressing table
//                       per read (cp_perr_hash) that stays cache-resident.  Functions below are generic
//                       over that store (`PE`): get(pos,e,w) -> value or -inf, set(pos,e,w,v).
//   eintvl/ointvl       E-/O-interval lists (ClassPro.h:153-157)
// Index plen is reset like every other index (the reference leaves it stale, SURVEY.md hazard 1).
#pragma once
#include "cp_math.h"
#include "cp_ctx.h"

#ifndef CP_HDM
#ifdef __HIPCC__
#define CP_HDM __host__ __device__ __forceinline__
#else
#define CP_HDM inline
#endif
#endif
#include "cp_bounds.h"

// dense store: the reference's layout (used by tests/host_harness.cpp)
struct cp_perr_dense
  { double *a;
    CP_HDM double get(int pos, int e, int w) const { return a[(size_t)pos*4+e*2+w]; }
    CP_HDM void   set(int pos, int e, int w, double v) { a[(size_t)pos*4+e*2+w] = v; }
    // handle API: the cell of (pos,e,w), created unset if the store is sparse
    CP_HDM long long cell(int pos, int e, int w) { return (long long)pos*4+e*2+w; }
    CP_HDM double    load(long long c) const { return a[c]; }
    CP_HDM void      store(long long c, double v) { a[c] = v; }
  };

// sparse store: open addressing, linear probing, insert-only; keys[] start at -1, a slot's four
// values are set to -inf when the slot is claimed.  Capacity (mask+1) is a power of two >= 4*ncand+16
// while at most 3*ncand+2 positions can ever be touched, so the table never fills.
struct cp_perr_hash
  { int32_t *keys;                 // table of error type e at keys+e*tsize, vals+e*tsize*4
    double  *vals;
    uint32_t mask;                 // tsize-1
    CP_HDM uint32_t slot0(int pos) const { return ((uint32_t)pos*0x9E3779B1u >> 7) & mask; }
    CP_HDM double get(int pos, int e, int w) const
    { const int32_t *ke = keys+(size_t)e*(mask+1);
      uint32_t h = slot0(pos);
      while (true)
        { int32_t k = ke[h];
          if (k == pos) return vals[((size_t)e*(mask+1)+h)*4+e*2+w];
          if (k < 0) return -INFINITY;
          h = (h+1) & mask;
        }
    }
    CP_HDM long long cell(int pos, int e, int w)
    { int32_t *ke = keys+(size_t)e*(mask+1);
      double *ve = vals+(size_t)e*(mask+1)*4;
      uint32_t h = slot0(pos);
      while (true)
        { int32_t k = ke[h];
          if (k == pos) break;
          if (k < 0)
            { ke[h] = pos;
              for (int x = 0; x < 4; x++) ve[(size_t)h*4+x] = -INFINITY;
              break;
            }
          h = (h+1) & mask;
        }
      return ((long long)e*(mask+1)+h)*4+e*2+w;
    }
    CP_HDM double load(long long c) const { return vals[c]; }
    CP_HDM void   store(long long c, double v) { vals[c] = v; }
    CP_HDM void   set(int pos, int e, int w, double v) { store(cell(pos,e,w),v); }
  };

#ifdef __HIP_DEVICE_COMPILE__
// real LDS pointers: a generic (flat) access would also wait for the replay's outstanding global stores
#define CP_LDS_PTR(T) __attribute__((address_space(3))) T *
#else
#define CP_LDS_PTR(T) T *
#endif
struct cp_perr_hybrid
  { cp_perr_hash g;
    CP_LDS_PTR(int32_t) lkeys;    // -1 = empty; the OTHERS table follows the SELF table
    CP_LDS_PTR(double)  lvals;
    int                 lcap0, lcap1;   // slots of the SELF / OTHERS table (powers of two)
    int                 use_lds;        // bit e: pass e keeps its memo on chip
    CP_HDM double get(int pos, int e, int w) const
    { if (!((use_lds >> e) & 1)) return g.get(pos,e,w);
      const int key = pos*2+w, off = e ? lcap0 : 0;
      const uint32_t m = (uint32_t)(e ? lcap1 : lcap0)-1;
      uint32_t h = ((uint32_t)key*0x9E3779B1u >> 9) & m;
      while (true)
        { int32_t k = lkeys[off+h];
          if (k == key) return lvals[off+h];
          if (k < 0) return -INFINITY;
          h = (h+1) & m;
        }
    }
    CP_HDM long long cell(int pos, int e, int w)
    { if (!((use_lds >> e) & 1)) return g.cell(pos,e,w);
      const int key = pos*2+w, off = e ? lcap0 : 0;
      const uint32_t m = (uint32_t)(e ? lcap1 : lcap0)-1;
      uint32_t h = ((uint32_t)key*0x9E3779B1u >> 9) & m;
      while (true)
        { int32_t k = lkeys[off+h];
          if (k == key) break;
          if (k < 0) { lkeys[off+h] = key; lvals[off+h] = -INFINITY; break; }
          h = (h+1) & m;
        }
      return -1-(long long)(off+h);                     // negative: an on-chip cell
    }
    CP_HDM double load(long long c) const { return c < 0 ? lvals[-1-c] : g.load(c); }
    CP_HDM void   store(long long c, double v) { if (c < 0) lvals[-1-c] = v; else g.store(c,v); }
    CP_HDM void   set(int pos, int e, int w, double v) { store(cell(pos,e,w),v); }
  };

// E-/O-interval list (ClassPro.h:153-157 arrays): a plain array here; k_find_wall keeps the list in LDS while it is
// short (kernels.hip: fw_evl, same member functions).  put() returns false when slot i does not exist.
struct cp_ev_ptr
  { cp_eintvl *p;
    CP_HDM cp_ev_ptr(cp_eintvl *q = nullptr) : p(q) {}
    CP_HDM int       b(int i) const { return p[i].b; }
    CP_HDM int       e(int i) const { return p[i].e; }
    CP_HDM double    pe(int i) const { return p[i].pe; }
    CP_HDM cp_eintvl get(int i) const { return p[i]; }
    CP_HDM void      put(int i, const cp_eintvl &v) { p[i] = v; }
  };

// SEQ / PROF / LF: anything indexable like the read's bases, its counts and the log-factorial table
// (plain pointers, or the LDS-window accessors of kernels.hip).  EVL: the interval lists (cp_ev_ptr or fw_evl).
template <class PE, class SEQ = CP_SEQ_T, class PROF = CP_PROF_T, class LF = const double *, class EVL = cp_ev_ptr>
struct cp_read_t
  { const cp_dev_params *P;
    PROF                 prof;
    SEQ                  seq;
    LF                   lf;
    int                  plen, rlen;
    uint8_t             *wall;      // flags written by the OTHERS pass + PAIRED_M/ERROR (and everything, if shared)
    uint8_t             *wall_s;    // flags written by the SELF pass (may alias `wall`: the bits are disjoint)
    PE                   perror;    // memo of both error types (one object: indexing an array of stores by the
                                    // error type at run time would push the whole record into scratch memory)
    EVL                  eintvl, ointvl;
    int                  ecap;
    int                  eidx, oidx;
    int                  overflow;
    // Device replay only (use_win != 0): the "already paired" flag of a pass (wall.c:639) is set by that
    // pass alone, on partners at most K+25 positions ahead in all but pathological low-complexity runs, so
    // the lane that replays the pass keeps it in a 128-position bit window instead of re-reading the flag
    // array (a cold miss per candidate); a partner beyond the window switches the window off for the rest
    // of the read, after which the flags -- always written -- are read as in the reference.
    int                  use_win = 0, win_base = 0;
    uint64_t             win_lo = 0, win_hi = 0;
    // Device replay only: the OTHERS pass's "wall by the count change alone" stores (CP_CF_WALLNOW, most real
    // walls) were issued for all candidates at once before the replay, which then only visits live
    // candidates.  The reference skips such a store when an earlier pair took the position as its partner;
    // the replay restores that by writing the partner's flag byte outright when it accepts a pair.
    int                  spec_wallnow = 0;
  };

#if defined(CP_PROF_WALK) && defined(__HIP_DEVICE_COMPILE__)
extern __device__ unsigned long long g_live_prof[8];
extern __device__ unsigned long long g_emit_prof[8];
#endif
// (the stamps inside per-lane code cost more than what they measure: they are a build of their own, -DCP_PROF_WALK -DCP_PROF_INNER;
//  the per-phase stamps of -DCP_PROF_WALK alone are one clock read per phase and read on lane 0)
#if defined(CP_PROF_WALK) && defined(CP_PROF_INNER) && defined(__HIP_DEVICE_COMPILE__)
#define CP_LT(k) do { unsigned long long t_ = wall_clock64(); if (__ffsll((long long)__ballot(1))-1 == (int)(threadIdx.x & 63)) atomicAdd(&g_live_prof[k],t_-lt_); lt_ = wall_clock64(); } while (0)
#define CP_LT0() unsigned long long lt_ = wall_clock64()
#define CP_ET(k) do { unsigned long long t_ = wall_clock64(); if (__ffsll((long long)__ballot(1))-1 == (int)(threadIdx.x & 63)) atomicAdd(&g_emit_prof[k],t_-et_); et_ = wall_clock64(); } while (0)
#define CP_ET0() unsigned long long et_ = wall_clock64()
#else
#define CP_LT(k) ((void)0)
#define CP_LT0() ((void)0)
#define CP_ET(k) ((void)0)
#define CP_ET0() ((void)0)
#endif

#define CP_PERR(R,i,e,w) ((R)->perror.get(i,e,w))
#define CP_NEG_INF (-INFINITY)

// wall.c:310-315
template <class RD>
CP_HD void cp_update_perror(RD *R, int i, int e, int w, int cout, int cin, double erate, double lpe, double l1mpe)
{ if (CP_PERR(R,i,e,w) == CP_NEG_INF)
    R->perror.set(i,e,w,cp_p_errorin(R->lf,e,erate,lpe,l1mpe,cout,cin));
}

// wall.c:317-322
template <class RD>
CP_HD double cp_logp_diff_pair(const RD *R, int i, int j)
{ const auto &pr = R->prof;
  int n_drop = (int)pr[i-1]-pr[i];
  int n_gain = (int)pr[j]-pr[j-1];
  int cov    = pr[i-1] > pr[j] ? pr[i-1] : pr[j];
  return cp_logp_trans(R->P,i,j,n_drop,n_gain,cov);
}

// eight consecutive counts from position lo (only 2-byte aligned; gfx9 global loads take unaligned addresses)
struct __attribute__((packed, aligned(2))) cp_u16x8 { uint16_t v[8]; };
template <class PROF>
CP_HD cp_u16x8 cp_load_u16x8(const PROF &prof, int lo) { return *reinterpret_cast<const cp_u16x8 *>(CP_SPAN(prof,lo,8)); }

// wall.c:324-329
CP_HD bool cp_cthres_ng(int e, int cin, int ct)
{ return (e == CP_SELF) ? (cin >= ct) : (cin < ct); }

// ---------------------------------------------------------------------------------------------
//  One iteration of the candidate walk (wall.c:590-707) for a position i that passed the scan
//  (min(c[i-1],c[i]) < R and |c[i-1]-c[i]| >= 3), split by what it depends on:
//
//   cp_wall_candidate_pre    what the two error-type passes share: the count pair and the low-
//                            complexity context with the largest error rate (wall.c:612-634).
//   cp_wall_candidate_pure   everything of one pass (error type e) that is a function of the read
//                            alone: the threshold filters (wall.c:643-653, 672-676), the candidate's
//                            own P(error) (the value update_perror would store, wall.c:310-315),
//                            find_gain / find_drop's low-complexity partner j and the value
//                            update_perror(j) would store (wall.c:345-378 / 432-467), and the best
//                            high-complexity partner (wall.c:380-404 / 469-493: six scores that use no
//                            memoised value; "first strict maximum" over them).
//   cp_wall_candidate_replay the part that depends on earlier candidates: the paired flags
//                            (wall.c:639), the perror memo -- a position's entry is whatever the FIRST
//                            request computed, and an earlier candidate may have requested it with its
//                            own error rate -- and the flag / interval-list updates.
//
//  The reference evaluates all of this inline per candidate.  On the device every load of that walk
//  is a cold miss and the Bessel / binomial-tail evaluations are long serial chains, so k_find_wall
//  computes pre+pure for 64 candidates at once (one per lane) and then replays the candidates in
//  order; the SELF and OTHERS replays touch disjoint state (different flag bits, memo entries and
//  interval lists; each depends only on its own earlier iterations) and run on two lanes.
// ---------------------------------------------------------------------------------------------
struct cp_wall_pre
  { int    cng, wtype, cin, cout, maxt, maxl;
    double maxpe, lpe, l1mpe;
  };

#define CP_CF_LIVE     1      // passes the threshold filters: the pass goes on to update_perror(i)
#define CP_CF_WALLNOW  2      // OTHERS pass: a wall by the count change alone (wall.c:672-676)
#define CP_LC_NONE     0      // find_gain/find_drop return false at once (partner on the wrong side)
#define CP_LC_BOUNDARY 1      // partner beyond the read end: pe = perror[i]^2
#define CP_LC_PAIR     2      // partner j inside the read and admissible
#define CP_LC_REJECT   3      // partner j inside the read but filtered out: pe = -inf

struct cp_cand_pure
  { double own_pe;            // p_errorin for (i,e,wtype) with the candidate's own error rate
    double lc_v;              // p_errorin for the low-complexity partner (lc_j,e,1-wtype), same error rate
    double hc_pe;             // best high-complexity score, valid if hc_j >= 0
    int    lc_j, hc_j;
    int    flags, lc_kind;
  };

template <class RD>
CP_HD void cp_wall_candidate_pre(const RD *R, int i, cp_wall_pre *pre)
{ const cp_dev_params *P = R->P;
  const int cim1 = R->prof[i-1], ci = R->prof[i];
  int cng = cim1-ci;
  if (cng < 0) cng = -cng;
  pre->cng = cng;
  if (cim1 > ci) { pre->wtype = CP_DROP; pre->cin = ci;   pre->cout = cim1; }
  else           { pre->wtype = CP_GAIN; pre->cin = cim1; pre->cout = ci;   }
  int maxt = -1, maxl = -1;                              // wall.c:624-634
  double maxpe = CP_NEG_INF;
  int l3[3];
  cp_ctx3(R->seq,R->rlen,P->K,pre->wtype,i,l3);
  for (int t = 0; t < 3; t++)
    { int l = l3[t];
      if (l > P->lmax[t]) l = P->lmax[t];
      double pe = P->pe[t][l];
      if (maxpe < pe)
        { maxpe = pe; maxt = t; maxl = l; }
    }
  pre->maxt = maxt; pre->maxl = maxl; pre->maxpe = maxpe;
  pre->lpe = P->lpe[maxt][maxl];
  pre->l1mpe = P->l1mpe[maxt][maxl];
}

// The threshold filters of one pass (wall.c:643-653, 672-676): 0, CP_CF_WALLNOW or CP_CF_LIVE.
CP_HD int cp_wall_candidate_filter(const cp_dev_params *P, int e, const cp_wall_pre &pre)
{ const int CMAX = P->cmax;
  const int cng = pre.cng, cin = pre.cin, cout = pre.cout;
  int ct_init = 0, ct_final = 0;
  if (cout < CMAX)                                       // wall.c:643-648
    { ct_init  = P->cthres[pre.maxt][pre.maxl][cout][CP_INIT][e];
      ct_final = P->cthres[pre.maxt][pre.maxl][cout][CP_FINAL][e];
      if (!(cng > CP_MAX_CNT_CHANGE || cin < (ct_init > 3 ? ct_init : 3)))
        return 0;
    }
  if (e == CP_SELF)                                      // wall.c:651-653
    { if (cout < CMAX && cin >= ct_final)
        return 0;
    }
  else if (cng >= P->cov[CP_HAPLO] || (cout < CMAX && cin < ct_final))       // wall.c:672-676
    return CP_CF_WALLNOW;
  return CP_CF_LIVE;
}

// The expensive, read-only part of a pass whose filter said CP_CF_LIVE.
template <class RD>
CP_HD void cp_wall_candidate_live(RD *R, int i, int e, const cp_wall_pre &pre, cp_cand_pure *out)
{ const cp_dev_params *P = R->P;
  const auto &pr = R->prof;
  const int plen = R->plen, K = P->K, CMAX = P->cmax;
  const int w = pre.wtype, cin = pre.cin, cout = pre.cout;
  const int t = pre.maxt, l = pre.maxl;
  out->flags = CP_CF_LIVE; out->lc_kind = CP_LC_NONE; out->lc_j = -1; out->hc_j = -1;
  out->lc_v = out->hc_pe = CP_NEG_INF;
  CP_LT0();
  out->own_pe = cp_p_errorin_tl(P,e,t,l,cout,cin);
  CP_LT(0);

  // find_gain (w == DROP: partner GAIN to the right of i) / find_drop (w == GAIN: partner DROP to the
  // left), wall.c:331-507, folded into one routine by mirroring the index arithmetic.
  const int ulen = t+1;
  const bool right = (w == CP_DROP);
  int n, j, cout_j, cin_j;

  // low-complexity partner (wall.c:345-378 / 432-467)
  const int m = ulen*l;
  n = 0;
  while (true)
    { int idx = right ? i+ulen*(n+1) : i-ulen*(n+1);
      if (right ? (idx >= plen) : (idx <= 0))
        break;
      if (cp_ctx(R->seq,R->rlen,K,w,idx,t) != m+n+1)     // ctx[DROP] for find_gain, ctx[GAIN] for find_drop
        break;
      n++;
    }
  CP_LT(1);
  j = right ? (i+K-1)+n-m : (i-K+1)-n+m;
  if (right ? (j <= i) : (j >= i))
    return;                                              // lc_kind NONE: no pair at all (wall.c:355 / 442)
  if (right ? (j >= plen) : (j <= 0))
    { out->lc_kind = CP_LC_BOUNDARY;
      out->lc_j = right ? plen : 0;
    }
  else
    { if (right) { cin_j = pr[j-1]; cout_j = pr[j]; }
      else       { cout_j = pr[j-1]; cin_j = pr[j]; }
      out->lc_j = j;
      if (cin_j <= cout_j
          && !(cout_j < CMAX && cp_cthres_ng(e,cin_j,P->cthres[t][l][cout_j][CP_FINAL][e]))
          && (e == CP_SELF || (right ? cp_logp_diff_pair(R,i,j) : cp_logp_diff_pair(R,j,i)) >= CP_THRES_DIFF_EO))
        { out->lc_kind = CP_LC_PAIR;
          out->lc_v = cp_p_errorin_tl(P,e,t,l,cout_j,cin_j);
        }
      else
        out->lc_kind = CP_LC_REJECT;
    }

  CP_LT(2);
  // high-complexity partners (wall.c:380-404 / 469-493).  The six partners sit at consecutive positions, so their
  // seven counts come from ONE 16-byte load where that stays inside the read, and each partner's two look-ups (its
  // P(error in), and for OTHERS the Skellam term of logp_diff_pair) are independent of the other partners': all of
  // them are issued before the first is used (first pass), then the reference's loop picks the first strict maximum
  // (second pass).  One dependent chain of six round trips became two.
  int    hcj[CP_MAX_N_HC+1], hcin[CP_MAX_N_HC+1], hcout[CP_MAX_N_HC+1];
  bool   hok[CP_MAX_N_HC+1];
  double hpe[CP_MAX_N_HC+1], hdiff[CP_MAX_N_HC+1];
  int    cw[CP_MAX_N_HC+2];                               // counts at j0-1 .. j0+5 (right) / j0-6 .. j0 (left), j0 = the n = 0 partner
  const int j0 = right ? (i+K-1) : (i-K+1);
  const int lo8 = right ? j0-1 : j0-(CP_MAX_N_HC+1);
  const bool wide = lo8 >= 0 && lo8+8 <= plen;
  if (wide)
    { const cp_u16x8 x = cp_load_u16x8(pr,lo8);
#ifdef __HIPCC__
#pragma unroll
#endif
      for (int q = 0; q < CP_MAX_N_HC+2; q++) cw[q] = x.v[q];
    }
  const bool pass_i = !(cout < CMAX && cp_cthres_ng(e,cin,P->cthres[CP_HP][1][cout][CP_FINAL][e]));
#ifdef __HIPCC__
#pragma unroll
#endif
  for (n = 0; n <= CP_MAX_N_HC; n++)
    { j = right ? j0+n : j0-n;
      hcj[n] = j; hok[n] = false; hpe[n] = 0.; hdiff[n] = 0.;
      if (right ? (j >= plen) : (j <= 0))
        { hcj[n] = -1; continue; }                        // the reference's loop ends here
      int a, b;                                           // counts at j-1, j
      if (wide) { a = cw[right ? n : CP_MAX_N_HC-n]; b = cw[right ? n+1 : CP_MAX_N_HC+1-n]; }
      else      { a = pr[j-1]; b = pr[j]; }
      if (right) { cin_j = a; cout_j = b; }
      else       { cout_j = a; cin_j = b; }
      hcin[n] = cin_j; hcout[n] = cout_j;
      if (!(cin_j <= cout_j) || !pass_i
          || (cout_j < CMAX && cp_cthres_ng(e,cin_j,P->cthres[CP_HP][1][cout_j][CP_FINAL][e])))
        continue;
      hok[n] = true;
      if (e == CP_OTHERS)                                 // logp_diff_pair (wall.c:317-322) with the counts already in hand
        { const int pim1 = pr[i-1], pi = pr[i];
          const int n_drop = right ? pim1-pi : a-b;       // (p[i-1]-p[i]) of the DROP position
          const int n_gain = right ? b-a : pi-pim1;       // (p[j]-p[j-1]) of the GAIN position
          const int cv = right ? (pim1 > b ? pim1 : b) : (a > pi ? a : pi);
          hdiff[n] = cp_logp_trans(P,right ? i : j,right ? j : i,n_drop,n_gain,cv);
        }
      hpe[n] = cp_p_errorin_tl(P,e,CP_HP,1,cout_j,cin_j);
    }
  bool   have_pe_i = false;
  double pe_i = 0., max_pe = CP_NEG_INF;
  int    max_j = -1;
#ifdef __HIPCC__
#pragma unroll
#endif
  for (n = 0; n <= CP_MAX_N_HC; n++)
    { if (hcj[n] < 0) break;
      if (!hok[n]) continue;
      if (e == CP_OTHERS && hdiff[n] < CP_THRES_DIFF_EO)
        continue;
      if (!have_pe_i)                                   // same arguments every time (wall.c:398)
        { pe_i = cp_p_errorin_tl(P,e,CP_HP,1,cout,cin);      // HC_ERATE = pe[HP][1], wall.c:180
          have_pe_i = true;
        }
      double pe = pe_i * hpe[n];
      if (max_pe < pe)
        { max_j  = hcj[n];
          max_pe = pe;
        }
    }
  out->hc_j = max_j;
  out->hc_pe = max_pe;
  CP_LT(3);
}

template <class RD>
CP_HD void cp_wall_candidate_pure(RD *R, int i, int e, const cp_wall_pre &pre, cp_cand_pure *out)
{ out->flags = cp_wall_candidate_filter(R->P,e,pre);
  out->lc_kind = CP_LC_NONE; out->lc_j = -1; out->hc_j = -1;
  out->own_pe = out->lc_v = out->hc_pe = CP_NEG_INF;
  if (out->flags & CP_CF_LIVE)
    cp_wall_candidate_live(R,i,e,pre,out);
}

// find_gain / find_drop with the memo state of the moment.  The reference keeps one running maximum
// over the low-complexity partner and then the six high-complexity ones, replacing it only by a
// strictly larger score; starting the high-complexity scan from -inf instead (cp_wall_candidate_pure)
// and comparing its winner with the low-complexity score picks the same partner.
template <class RD>
CP_HD bool cp_find_pair_replay(RD *R, int i, int e, int w, const cp_cand_pure &c, double pe_i, cp_eintvl *out)
{ if (c.lc_kind == CP_LC_NONE)
    return false;
  const bool right = (w == CP_DROP);
  int max_j = -1;
  double pe = CP_NEG_INF, max_pe = CP_NEG_INF;
  if (c.lc_kind == CP_LC_BOUNDARY)
    pe = pe_i * pe_i;                                     // pe_i = perror[i][e][w], read by the caller
  else if (c.lc_kind == CP_LC_PAIR)
    { const long long cj = R->perror.cell(c.lc_j,e,1-w);
      double pe_j = R->perror.load(cj);
      if (pe_j == CP_NEG_INF)                             // update_perror(j), wall.c:310-315
        { pe_j = c.lc_v;
          R->perror.store(cj,pe_j);
        }
      pe = right ? pe_i*pe_j : pe_j*pe_i;                 // perror[.][DROP] * perror[.][GAIN]
    }
  if (max_pe < pe)
    { max_j  = c.lc_j;
      max_pe = pe;
    }
  if (c.hc_j >= 0 && max_pe < c.hc_pe)
    { max_j  = c.hc_j;
      max_pe = c.hc_pe;
    }
  if (max_j == -1)
    return false;
  if (right) { out->b = i;     out->e = max_j; }
  else       { out->b = max_j; out->e = i;     }
  out->pe = max_pe;
  return true;
}

template <class RD>
CP_HD bool cp_win_paired(RD *R, int i)                  // advance the window to position i; is i paired?
{ const int d = i-R->win_base;
  if (d >= 128)     { R->win_lo = R->win_hi = 0; }
  else if (d >= 64) { R->win_lo = R->win_hi >> (d-64); R->win_hi = 0; }
  else if (d > 0)   { R->win_lo = (R->win_lo >> d) | (R->win_hi << (64-d)); R->win_hi >>= d; }
  R->win_base = i;
  return (R->win_lo & 1) != 0;
}
template <class RD>
CP_HD void cp_win_mark(RD *R, int j)                    // position j is now paired
{ const int d = j-R->win_base;
  if (d <= 0) return;                                   // behind the walk: never asked again
  if (d < 64)       R->win_lo |= 1ull << d;
  else if (d < 128) R->win_hi |= 1ull << (d-64);
  else              R->use_win = 0;                     // out of reach: fall back to the flag array from here on
}

template <class RD>
CP_HD void cp_wall_candidate_replay(RD *R, int i, int e, int wtype, const cp_cand_pure &c)
{ // wall.c:639.  With the window, "not paired" also means the pass has not written position i yet.
  const bool win = R->use_win != 0;
  if (win ? cp_win_paired(R,i)
          : (e == CP_SELF ? (R->wall_s[i] & CP_W_PAIRED_S) != 0 : (R->wall[i] & CP_W_PAIRED_O) != 0))
    return;
  if (c.flags == 0)
    return;
  if (c.flags & CP_CF_WALLNOW)
    { if (win) R->wall[i] = CP_W_WALL_O; else R->wall[i] |= CP_W_WALL_O;
      return;
    }
  const long long ci = R->perror.cell(i,e,wtype);
  double pe_i = R->perror.load(ci);
  if (pe_i == CP_NEG_INF)                                  // update_perror(i), wall.c:310-315
    { pe_i = c.own_pe;
      R->perror.store(ci,pe_i);
    }
  cp_eintvl I;
  if (e == CP_SELF)                                      // wall.c:651-670
    { if (pe_i < CP_PE_THRES_FINAL)
        return;
      if (cp_find_pair_replay(R,i,e,wtype,c,pe_i,&I) && I.pe >= CP_PE_THRES_FINAL)
        { if (win)                                       // the SELF array only ever holds these two bits
            { R->wall_s[I.b] = (CP_W_WALL_S|CP_W_PAIRED_S);
              R->wall_s[I.e] = (CP_W_WALL_S|CP_W_PAIRED_S);
              cp_win_mark(R,I.b == i ? I.e : I.b);
            }
          else
            { R->wall_s[I.b] |= (CP_W_WALL_S|CP_W_PAIRED_S);
              R->wall_s[I.e] |= (CP_W_WALL_S|CP_W_PAIRED_S);
            }
          if (R->eidx < R->ecap) R->eintvl.put(R->eidx++,I);
          else R->overflow = 1;
        }
    }
  else                                                   // wall.c:671-690
    { if (pe_i < CP_PE_THRES_FINAL)
        { if (win) R->wall[i] = CP_W_WALL_O; else R->wall[i] |= CP_W_WALL_O;
          return;
        }
      if (cp_find_pair_replay(R,i,e,wtype,c,pe_i,&I) && I.pe >= CP_PE_THRES_FINAL)
        { const int j = (I.b == i) ? I.e : I.b;
          R->wall[i] |= CP_W_PAIRED_O;
          if (R->spec_wallnow && j > i) R->wall[j] = CP_W_PAIRED_O;    // not reached yet: the reference has 0 there
          else                          R->wall[j] |= CP_W_PAIRED_O;
          if (win) cp_win_mark(R,j);
          if (R->oidx < R->ecap) R->ointvl.put(R->oidx++,I);
          else R->overflow = 1;
          return;
        }
      if (win) R->wall[i] = CP_W_WALL_O; else R->wall[i] |= CP_W_WALL_O;
    }
}

template <class RD>
CP_HD void cp_wall_candidate_e(RD *R, int i, int e)
{ cp_wall_pre pre;
  cp_cand_pure c;
  cp_wall_candidate_pre(R,i,&pre);
  cp_wall_candidate_pure(R,i,e,pre,&c);
  cp_wall_candidate_replay(R,i,e,pre.wtype,c);
}

template <class RD>
CP_HD void cp_wall_candidate(RD *R, int i)
{ cp_wall_candidate_e(R,i,CP_SELF);
  cp_wall_candidate_e(R,i,CP_OTHERS);
}

// wall.c:519-528: order by (b,e); the pe term truncates to 0, and glibc's qsort is a stable merge
// sort, so ties keep their insertion order.
CP_HD bool cp_eintvl_before(const cp_eintvl &a, int ia, const cp_eintvl &b, int ib)
{ if (a.b != b.b) return a.b < b.b;
  if (a.e != b.e) return a.e < b.e;
  return ia < ib;
}

// wall.c:530-546
template <class EVL>
CP_HD int cp_bs_eintvl(const EVL &v, int l, int r, int b, int e)
{ while (l <= r)
    { int m = (l+r)/2;
      const int vb = v.b(m);
      if (vb == b)
        { const int ve = v.e(m);
          if (ve == e) return m;
          else if (e > ve) l = m+1;
          else r = m-1;
        }
      else if (b > vb) l = m+1;
      else r = m-1;
    }
  return -1;
}

// wall.c:548-568 after the sort: keep the first of each run of equal (b,e)
CP_HD int cp_dedupe_sorted(cp_eintvl *v, int N)
{ if (N < 2) return N;
  int i = 1;
  while (i < N)
    { if (v[i-1].b == v[i].b && v[i-1].e == v[i].e)
        break;
      i++;
    }
  for (int j = i+1; j < N; j++)
    if (!(v[i-1].b == v[j].b && v[i-1].e == v[j].e))
      { v[i] = v[j];
        i++;
      }
  return i;
}

// wall.c:763-860 for one O-only wall i: look up to 200 positions right (DROP) / left (GAIN) for
// partner walls that would make [i,j) an error interval by multiple errors; boundary intervals at
// plen / 0.  `NS` = number of sorted unique E-intervals, `*midx` = append cursor.
template <class RD>
CP_HD void cp_wall_mult(RD *R, int i, int NS, int *midx)
{ uint8_t *wall = R->wall;
  const uint8_t *wall_s = R->wall_s;
  const int plen = R->plen;
  auto &ev = R->eintvl;
  auto ev_put = [&](int k, int b, int e, double p) { cp_eintvl x; x.b = b; x.e = e; x.pe = p; ev.put(k,x); };
  double pe, pe_i, pe_j;
  for (int w = CP_DROP; w <= CP_GAIN; w++)
    { if ((pe_i = CP_PERR(R,i,CP_SELF,w)) < CP_PE_THRES_FINAL)
        continue;
      if (w == CP_DROP)
        { int jend = (i+CP_MULT_WINDOW < plen+1) ? i+CP_MULT_WINDOW : plen+1;
          for (int j = i+1; j < jend; j++)
            { if (j == plen)
                { if ((pe = pe_i * pe_i) < CP_PE_THRES_FINAL)
                    continue;
                  if (*midx >= R->ecap) { R->overflow = 1; return; }
                  ev_put(*midx,i,plen,pe);
                  wall[i] |= CP_W_PAIRED_M;
                  (*midx)++;
                  if (*midx >= plen) { R->overflow = 8; return; }        // the reference exits here: "# E-intvls >= plen" (wall.c:783-788)
                }
              if (!((wall[j] & CP_W_WALL_O) | (wall_s[j] & CP_W_WALL_S)))
                continue;
              if (cp_bs_eintvl(ev,0,NS-1,i,j) == -1)
                { pe_j = CP_PERR(R,j,CP_SELF,CP_GAIN);
                  if ((pe = pe_i * pe_j) >= CP_PE_THRES_FINAL)
                    { if (*midx >= R->ecap) { R->overflow = 1; return; }
                      ev_put(*midx,i,j,pe);
                      wall[i] |= CP_W_PAIRED_M;
                      wall[j] |= CP_W_PAIRED_M;
                      (*midx)++;
                      if (*midx >= plen) { R->overflow = 8; return; }
                    }
                }
              if (wall[j] & CP_W_WALL_O)
                break;
            }
        }
      else
        { int jend = (i-CP_MULT_WINDOW > 0) ? i-CP_MULT_WINDOW : 0;
          for (int j = i-1; j >= jend; j--)
            { if (j == 0)
                { if ((pe = pe_i * pe_i) < CP_PE_THRES_FINAL)
                    continue;
                  if (*midx >= R->ecap) { R->overflow = 1; return; }
                  ev_put(*midx,0,i,pe);
                  wall[i] |= CP_W_PAIRED_M;
                  (*midx)++;
                  if (*midx >= plen) { R->overflow = 8; return; }        // the reference exits here: "# E-intvls >= plen" (wall.c:783-788)
                }
              if (!((wall[j] & CP_W_WALL_O) | (wall_s[j] & CP_W_WALL_S)))
                continue;
              if (cp_bs_eintvl(ev,0,NS-1,j,i) == -1)
                { pe_j = CP_PERR(R,j,CP_SELF,CP_DROP);
                  if ((pe = pe_i * pe_j) >= CP_PE_THRES_FINAL)
                    { if (*midx >= R->ecap) { R->overflow = 1; return; }
                      ev_put(*midx,j,i,pe);
                      wall[i] |= CP_W_PAIRED_M;
                      wall[j] |= CP_W_PAIRED_M;
                      (*midx)++;
                      if (*midx >= plen) { R->overflow = 8; return; }
                    }
                }
              if (wall[j] & CP_W_WALL_O)
                break;
            }
        }
    }
}

// wall.c:878-909: append the union of every chain of overlapping E-intervals (list sorted by (b,e)).
// The reference bounds both loops by NS while NS grows with every appended union, so the scan runs on
// into the entries it has just appended (which are not in sorted position); reproduced literally.
template <class RD>
CP_HD int cp_merge_eintvl(RD *R, int NS, int i0 = 0)     // i0: where the scan starts (kernels.hip does the chains before that in parallel)
{ auto &ev = R->eintvl;
  int i = i0, j;
  while (i < NS-1)
    { int    max_e  = ev.e(i);
      double max_pe = ev.pe(i);
      j = i;
      while (j < NS-1)
        { if (ev.b(j+1) <= ev.e(j))
            { const int e1 = ev.e(j+1);
              const double p1 = ev.pe(j+1);
              if (max_e < e1) max_e = e1;
              if (!(max_pe > p1)) max_pe = p1;
              j++;
            }
          else
            break;
        }
      if (i < j)
        { if (NS >= R->ecap) { R->overflow = 1; return NS; }
          cp_eintvl x; x.b = ev.b(i); x.e = max_e; x.pe = max_pe;
          ev.put(NS,x);
          NS++;
          if (NS >= R->plen) { R->overflow = 8; return NS; }      // the reference exits here too ("# E-intvls >= plen", wall.c:900-905)
        }
      i = j+1;
    }
  return NS;
}

// wall.c:928-946: the record of interval [b,e) given the final sorted E-interval list.
template <class RD>
CP_HD void cp_make_interval(const RD *R, int NS, int b, int e, cp_intvl *out)
{ int idx = cp_bs_eintvl(R->eintvl,0,NS-1,b,e);
  out->b = b;
  out->e = e;
  out->cb = R->prof[b];
  out->ce = R->prof[e-1];
  out->ccb = 0;
  out->cce = 0;
  out->is_rel = 0;
  out->asgn = CP_N_STATE;
  for (int k = 0; k < 6; k++) out->_pad[k] = 0;
  out->pe = (idx != -1) ? cp_log(R->eintvl.pe(idx)) : CP_NEG_INF;
  double d = CP_PERR(R,b,CP_OTHERS,CP_DROP), g = CP_PERR(R,b,CP_OTHERS,CP_GAIN);
  double peob = d > g ? d : g;
  d = CP_PERR(R,e,CP_OTHERS,CP_DROP); g = CP_PERR(R,e,CP_OTHERS,CP_GAIN);
  double peoe = d > g ? d : g;
  out->peo_b = (peob != CP_NEG_INF) ? cp_log(peob) : CP_NEG_INF;
  out->peo_e = (peoe != CP_NEG_INF) ? cp_log(peoe) : CP_NEG_INF;
}

// Sum over i in [lo,hi) of the upward (sgn = +1) or downward (sgn = -1) steps prof[i+1]-prof[i] of a count
// profile, eight counts per load where that stays inside the read's plen counts (the addresses are only
// 2-byte aligned; gfx9 global loads take unaligned addresses).  Same terms as the loops of wall.c:972-1001.
// hi may equal plen (wall.c:976-978, `last = I.b+lmax` with a low-complexity run that reaches the end of the read) and
// even exceed it by up to 126 (a homopolymer of more than 127 bases: rctx holds a reversed copy of capped values there,
// context.c:24-25, so lmax can be 127 with fewer bases left): the reference then reads profile[plen ..], cells no read
// owns -- fresh heap for the first read of a thread.  Here every count at or beyond plen reads 0 whatever follows the
// read in memory (the tail loop below never dereferences them), so a read's result is a function of the read alone.
template <class PROF>
CP_HD int cp_sum_steps(const PROF &prof, int lo, int hi, int plen, int sgn)
{ int acc = 0, i = lo;
  if (lo >= hi) return 0;
  int prev = prof[lo];
  // The two long sums of an interval that passed the tests are K-1 and K-2 steps: with K <= 41 and forty counts after `lo`
  // inside the read, their five wide loads are issued together -- as a loop each load waited for the sum of the one before
  // (the trip count is the lane's own), five round trips in a row, twice per interval.
  if (hi-lo > 8 && hi-lo <= 40 && lo+40 < plen)
    { cp_u16x8 x[5];
#ifdef __HIPCC__
#pragma unroll
#endif
      for (int g = 0; g < 5; g++) x[g] = cp_load_u16x8(prof,lo+1+8*g);
      const int n = hi-lo;
#ifdef __HIPCC__
#pragma unroll
#endif
      for (int g = 0; g < 5; g++)
        {
#ifdef __HIPCC__
#pragma unroll
#endif
          for (int q = 0; q < 8; q++)
            { const int cur = x[g].v[q], d = sgn*(cur-prev);
              if (8*g+q < n && d > 0) acc += d;
              prev = cur;
            }
        }
      return acc;
    }
  // (a last group of fewer than eight steps takes one wide load as well, of which it uses what it needs, while the eight
  //  counts lie inside the read: K-1 = 39 steps were four wide loads and SEVEN single ones, each a round trip of its own)
  while (i < hi && i+8 < plen)
    { const cp_u16x8 x = cp_load_u16x8(prof,i+1);
      const int n = hi-i;                                 // steps still wanted (>= 8: all of the group)
#ifdef __HIPCC__
#pragma unroll
#endif
      for (int q = 0; q < 8; q++)
        { const int cur = x.v[q], d = sgn*(cur-prev);
          if (q < n && d > 0) acc += d;
          prev = cur;
        }
      i += 8;
    }
  for (; i < hi; i++)
    { const int cur = (i+1 < plen) ? prof[i+1] : 0, d = sgn*(cur-prev);   // prof[plen] is DEFINED as 0 (hazard 8)
      if (d > 0) acc += d;
      prev = cur;
    }
  return acc;
}

// correct_wall_cnt + the filters of find_rel_intvl, wall.c:960-1051, for interval `idx`.
// The reference's position-indexed loops at wall.c:999-1006 only ever touch the interval itself when
// its index equals its start position (SURVEY.md hazard 2); that case is applied explicitly.
// seq_b / seq_e: the read's bases as seen by the context scan after the interval's begin / before its end (the same
// pointer on the host; on the device two short LDS windows backed by the pointer, kernels.hip).
// The view of the read's bases that the scans around position `pos` use (dir = +1: they go right of it, -1: left): the
// view itself for plain pointers and LDS windows; kernels.hip's cp_seq_rsrc answers with sixteen bases in registers.
template <class S> CP_HD const S &cp_seq_window(const S &s, int, int, int) { return s; }

// The three tests of wall.c:1016-1019 that need nothing but the record as cp_make_interval left it.
CP_HD bool cp_rel_prefilter(const cp_dev_params *P, const cp_intvl *I)
{ if (I->e-I->b < P->K)
    return false;
  if ((I->cb > I->ce ? I->cb : I->ce) >= P->cov[CP_REPEAT])
    return false;
  if (I->pe >= P->log_pe_final)                          // cp_log(PE_THRES[FINAL][SELF]), wall.c:1018
    return false;
  return true;
}

// ... and the rest for an interval that passed them: a function of (b, e, cb, ce, idx) and the read, so that any lane
// can do it for any interval (k_find_wall packs the intervals that got here into as few 64-lane steps as they need).
template <class PROF, class SEQB, class SEQE>
CP_HD bool cp_rel_counts(const cp_dev_params *P, const PROF &prof, const SEQB &seq_b, const SEQE &seq_e, int rlen,
                         int Ib, int Ie, int Icb, int Ice, int idx, int *ccb_out, int *cce_out)
{ const int K = P->K;
  CP_ET0();
  const auto wseq_b = cp_seq_window(seq_b,Ib+K-1,rlen,+1);     // rctx scans start here and go right (and a little to the left)
  const auto wseq_e = cp_seq_window(seq_e,Ie-1,rlen,-1);       // lctx scans start here and go left

  int first, last, n_gain = 0, n_drop = 0, lmax;
  const int plen = rlen-(K-1);
  last = (Ib+K-1 < Ie-1) ? Ib+K-1 : Ie-1;
  n_gain += cp_sum_steps(prof,Ib,last,plen,+1);
  CP_ET(5);
  if (Ib+K-1 < Ie)
    { lmax = 0;
      int l3[3];
      cp_rctx3(wseq_b,rlen,Ib+K-1,l3);
      for (int t = 0; t < 3; t++)
        { int l = l3[t]*(t+1);
          if (lmax < l) lmax = l;
        }
      last = Ib+lmax;
      n_gain -= cp_sum_steps(prof,Ib,last,plen,-1);
    }
  first = (Ie-K+1 > Ib) ? Ie-K+1 : Ib;
  n_drop += cp_sum_steps(prof,first,Ie-1,plen,-1);
  if (Ib < Ie-K+1)
    { lmax = 0;
      int l3[3];
      cp_lctx3(wseq_e,rlen,(Ie-K+1)+K-2,l3);                 // ctx[DROP][e-K+1]
      for (int t = 0; t < 3; t++)
        { int l = l3[t]*(t+1);
          if (lmax < l) lmax = l;
        }
      first = Ie-lmax;
      n_drop -= cp_sum_steps(prof,first,Ie-1,plen,+1);
    }
  CP_ET(6);
  int ccb = Icb+(n_gain > 0 ? n_gain : 0);
  int cce = Ice+(n_drop > 0 ? n_drop : 0);
  if (ccb > CP_MAX_KMER_CNT) ccb = CP_MAX_KMER_CNT;
  if (cce > CP_MAX_KMER_CNT) cce = CP_MAX_KMER_CNT;
  if (idx == Ib && Ie-2*K <= Ib && cce < prof[Ib])       // wall.c:1003-1006 with intvl index == position
    cce = prof[Ib];
  *ccb_out = ccb;
  *cce_out = cce;

  const double lpt_ = cp_logp_trans(P,Ib,Ie,ccb,cce,(ccb+cce)/2);
  const bool rel_ = !(lpt_ < CP_THRES_DIFF_REL) && !((ccb > cce ? ccb : cce) == CP_MAX_KMER_CNT);
  CP_ET(7);
  return rel_;
}

template <class PROF, class SEQB, class SEQE>
CP_HD bool cp_rel_interval(const cp_dev_params *P, const PROF &prof, const SEQB &seq_b, const SEQE &seq_e, int rlen,
                           cp_intvl *I, int idx)
{ if (!cp_rel_prefilter(P,I))
    return false;
  int ccb, cce;
  const bool rel = cp_rel_counts(P,prof,seq_b,seq_e,rlen,I->b,I->e,I->cb,I->ce,idx,&ccb,&cce);
  I->ccb = (uint16_t)ccb;
  I->cce = (uint16_t)cce;
  return rel;
}
