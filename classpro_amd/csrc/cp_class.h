// cp_class.h -- interval classification (reference src/class_rel.c, src/class_unrel.c) as device
// functions.
//
// classify_rel is a 4-state (E,R,H,D) Viterbi-like pass over the M reliable intervals of a read, run
// forward and backward.  The reference keeps, for every (interval, state), a full copy of the best
// path (`bt`, O(M^2) bytes and copies per read) only to answer three questions about that path:
//   - does it already contain a D (an H)?                                   class_rel.c:418-431,461-474
//   - which are the nearest alternating anchors H/D/H (D/H/D)?  calc_dh_ratio, class_rel.c:113-156
//   - the final traceback                                                   class_rel.c:606-613
// Here each DP cell carries four indices instead (last H, last D, last H before the last D, last D
// before the last H) which answer the first two in O(1), plus a one-byte back-pointer per cell for
// the traceback.  Only the previous interval's cells are live, so the DP state stays in registers /
// LDS and HBM scratch is 5 bytes per (interval, direction).
#pragma once
#include "cp_math.h"

#define CP_NONE (-1)

#ifndef CP_HDM
#ifdef __HIPCC__
#define CP_HDM __host__ __device__ __forceinline__
#else
#define CP_HDM inline
#endif
#endif

struct cp_cell                       // one DP cell = (interval i, state s); ClassPro.h:210-219 per cell
  { double dp;                       // normalised log score, -inf = unreachable
    double dhr;                      // D/H ratio on the best path (dh_ratio), -inf = none yet
    int    pos[4];                   // st[.][R|H|D].pos   (index by state code; [E] unused)
    int    cnt[4];                   // st[.][R|H|D].cnt   (cnt_t: kept masked to 16 bits)
    int    lastH, lastD;             // most recent path index with state H / D (inclusive of this cell)
    int    lastHbD, lastDbH;         // last H before lastD / last D before lastH
  };

// Hot fields of a reliable interval (the DP never needs more).
struct cp_riv { int b, e, ccb, cce; double pe; };
CP_HD cp_riv cp_riv_of(const cp_intvl &I) { cp_riv r; r.b = I.b; r.e = I.e; r.ccb = I.ccb; r.cce = I.cce; r.pe = I.pe; return r; }

// class_rel.c:45-58 accessors.  fw: positions grow with the path; bw: the mirror.
CP_HD int cp_pred(int x, int F)   { return F ? x-1 : x+1; }
CP_HD int cp_offs(int x, int F)   { return F ? x-CP_OFFSET : x+CP_OFFSET; }
CP_HD int cp_beg_pos(const cp_riv &I, int F) { return F ? I.b : I.e-1; }
CP_HD int cp_beg_cnt(const cp_riv &I, int F) { return F ? I.ccb : I.cce; }
CP_HD int cp_end_pos(const cp_riv &I, int F) { return F ? I.e-1 : I.b; }
CP_HD int cp_end_cnt(const cp_riv &I, int F) { return F ? I.cce : I.ccb; }

// class_rel.c:158-170
CP_HD double cp_logp_e(const cp_dev_params *P, const cp_riv &I, const int *COV)
{ double logp_po = cp_logp_poisson(P,I.ccb,COV[CP_ERROR])+cp_logp_poisson(P,I.cce,COV[CP_ERROR])+CP_E_PO_BASE;
  return (logp_po > I.pe) ? logp_po : I.pe;
}

// class_rel.c:172-211
CP_HD double cp_logp_r(const cp_dev_params *P, const cp_riv &I, int pred_r_cnt, int F, const int *COV)
{ int beg_cnt = cp_beg_cnt(I,F);
  double logp = (beg_cnt < pred_r_cnt) ? cp_logp_binom_pre(P->logfact,beg_cnt,pred_r_cnt,P->r_lp,P->r_l1mp) : -INFINITY;
  if (logp > CP_R_LOGP)
    return logp;
  int max_cc = I.ccb > I.cce ? I.ccb : I.cce;
  if (max_cc >= COV[CP_REPEAT]) return CP_R_LOGP;
  if (max_cc >= pred_r_cnt)     return CP_R_LOGP;
  return logp;
}

// class_rel.c:213-240.  With a D/H ratio on the path only the D-anchored transition survives.
CP_HD double cp_logp_h(const cp_dev_params *P, const cp_riv &I, const cp_cell &pr, int F)
{ int beg_pos = cp_beg_pos(I,F), beg_cnt = cp_beg_cnt(I,F);
  if (pr.dhr != -INFINITY)
    return cp_logp_trans(P,cp_pred(pr.pos[CP_DIPLO],F),beg_pos,pr.cnt[CP_DIPLO],(int)(pr.dhr*beg_cnt),pr.cnt[CP_DIPLO]);
  return cp_logp_trans(P,cp_pred(pr.pos[CP_HAPLO],F),beg_pos,pr.cnt[CP_HAPLO],beg_cnt,pr.cnt[CP_HAPLO]);
}

// class_rel.c:242-270.  Line 264 overwrites the ratio branch: always the D-anchored transition.
CP_HD double cp_logp_d(const cp_dev_params *P, const cp_riv &I, const cp_cell &pr, int F)
{ return cp_logp_trans(P,cp_pred(pr.pos[CP_DIPLO],F),cp_beg_pos(I,F),pr.cnt[CP_DIPLO],cp_beg_cnt(I,F),pr.cnt[CP_DIPLO]); }

// class_rel.c:272-277: transition s@pred -> t@i
CP_HD double cp_calc_logp(const cp_dev_params *P, int t, const cp_riv &I, const cp_cell &pr, int F, const int *COV)
{ if (t == CP_ERROR)      return cp_logp_e(P,I,COV);
  else if (t == CP_HAPLO) return cp_logp_h(P,I,pr,F);
  else if (t == CP_DIPLO) return cp_logp_d(P,I,pr,F);
  else                    return cp_logp_r(P,I,pr.cnt[CP_REPEAT],F,COV);
}

// calc_dh_ratio (class_rel.c:113-156) from the anchor indices: i1 = this interval (state init_s),
// i2 = nearest earlier interval of the other class, i3 = nearest interval of init_s before i2.
CP_HD double cp_dh_ratio(int init_s, const cp_riv &I1, const cp_riv &I2, const cp_riv &I3, int F)
{ int s1p = cp_beg_pos(I1,F), s1c = cp_beg_cnt(I1,F);
  int tp  = cp_end_pos(I2,F), tc  = cp_end_cnt(I2,F);
  int s2p = cp_end_pos(I3,F), s2c = cp_end_cnt(I3,F);
  if (!F)
    { int a = s1p; s1p = s2p; s2p = a;
      a = s1c; s1c = s2c; s2c = a;
    }
  double est = cp_linear_interpolation(tp,s2p,s2c,s1p,s1c);
  return (init_s == CP_DIPLO) ? est/tc : tc/est;
}

// class_rel.c:80-96 with s fixed (best target of source s) or t fixed (best source of target t).
// dp[4] = scores of the previous interval's cells, tr = normalised log transition matrix [s*4+t].
CP_HD int cp_argmax_tr(const double *dp, const double *tr, int s, int t, double *best)
{ double max_logp = -INFINITY;
  int max_x = CP_N_STATE;
  for (int x = 0; x < 4; x++)
    { int _s = (s < CP_N_STATE) ? s : x;
      int _t = (t < CP_N_STATE) ? t : x;
      double logp = dp[_s]+tr[_s*4+_t];
      if (max_logp < logp)
        { max_logp = logp;
          max_x = x;
        }
    }
  *best = max_logp;
  return max_x;
}

// First cell of a pass for state s (class_rel.c:544-580), before the normalisation of :582-586.
CP_HD void cp_rel_init_cell(const cp_dev_params *P, int s, const cp_riv &I, int i, int plen, int F,
                            const int *COV, cp_cell *c)
{ const int POS_INIT = cp_offs(F ? 0 : plen,F);
  const int ep = cp_end_pos(I,F), ec = cp_end_cnt(I,F), bc = cp_beg_cnt(I,F);
  for (int t = 0; t < 4; t++)
    { c->pos[t] = POS_INIT;
      c->cnt[t] = COV[t];
    }
  c->dhr = -INFINITY;
  c->lastH = c->lastD = c->lastHbD = c->lastDbH = CP_NONE;
  if (s == CP_ERROR)
    c->dp = cp_logp_e(P,I,COV);
  else if (s == CP_REPEAT)
    { c->dp = cp_logp_r(P,I,c->cnt[CP_REPEAT],F,COV);
      c->pos[CP_REPEAT] = ep;
      c->cnt[CP_REPEAT] = ec < COV[CP_REPEAT] ? ec : COV[CP_REPEAT];
    }
  else if (s == CP_HAPLO)
    { c->dp = cp_logp_poisson(P,bc,COV[CP_HAPLO]);
      c->pos[CP_HAPLO] = ep;
      c->cnt[CP_HAPLO] = ec;
      c->pos[CP_DIPLO] = cp_offs(ep,F);
      c->cnt[CP_DIPLO] = (ec+COV[CP_HAPLO]) & 0xffff;
      c->lastH = i;
    }
  else
    { c->dp = cp_logp_poisson(P,bc,COV[CP_DIPLO]);
      c->pos[CP_HAPLO] = cp_offs(ep,F);
      c->cnt[CP_HAPLO] = ((ec/2 > ec-COV[CP_HAPLO]) ? ec/2 : ec-COV[CP_HAPLO]) & 0xffff;
      c->pos[CP_DIPLO] = ep;
      c->cnt[CP_DIPLO] = ec;
      c->lastD = i;
    }
}

// "Only R reachable" step (class_rel.c:349-380) for state s: the cell is carried over and the path
// gains one more interval of state s (whose data stands in for the predecessor's).
CP_HD void cp_rel_only_r_cell(int s, int i, cp_cell *c)
{ if (c->dp == -INFINITY)
    return;
  if (s == CP_HAPLO)      { c->lastDbH = c->lastD; c->lastH = i; }
  else if (s == CP_DIPLO) { c->lastHbD = c->lastH; c->lastD = i; }
  c->dhr = -INFINITY;                                    // dh_ratio[i][s] is left at its reset value (:520-522)
}

// New cell for target state t at interval i given its best predecessor (class_rel.c:390-499).
// View gives the hot fields of the interval standing in for path index k: view(k) -> cp_riv.
// `prev` = the four cells of the previous interval (cp_cell, or a padded record derived from it: Cell).
template <class View, class Cell>
CP_HD void cp_rel_target_cell(const cp_dev_params *P, int t, int i, const cp_riv &I, int F, const int *COV,
                              int max_s, double max_logp, const Cell *prev, const View &view, cp_cell *out)
{ cp_cell c;
  c.dp = max_logp;
  c.dhr = -INFINITY;
  if (max_s == CP_N_STATE)
    { for (int k = 0; k < 4; k++) { c.pos[k] = 0; c.cnt[k] = 0; }
      c.lastH = c.lastD = c.lastHbD = c.lastDbH = CP_NONE;
      *out = c;
      return;
    }
  const cp_cell pr = prev[max_s];
  const int end_pos = cp_end_pos(I,F), end_cnt = cp_end_cnt(I,F);
  c.lastH = pr.lastH; c.lastD = pr.lastD; c.lastHbD = pr.lastHbD; c.lastDbH = pr.lastDbH;
  for (int k = 0; k < 4; k++) { c.pos[k] = pr.pos[k]; c.cnt[k] = pr.cnt[k]; }

  if (t == CP_ERROR)
    { /* anchors R,H,D carried over (:409-412) */ }
  else if (t == CP_REPEAT)
    { c.pos[CP_HAPLO] = c.pos[CP_DIPLO] = cp_offs(end_pos,F);          // counts carried over (:415-418)
      int r_cnt = end_cnt < COV[CP_REPEAT] ? end_cnt : COV[CP_REPEAT];
      if (!(pr.cnt[CP_REPEAT] < r_cnt))
        { c.pos[CP_REPEAT] = cp_offs(end_pos,F);
          c.cnt[CP_REPEAT] = r_cnt;
        }
    }
  else
    { const int other = (t == CP_HAPLO) ? CP_DIPLO : CP_HAPLO;
      const int i2 = (t == CP_HAPLO) ? pr.lastD : pr.lastH;               // nearest other-class anchor
      const int i3 = (t == CP_HAPLO) ? pr.lastHbD : pr.lastDbH;           // nearest same-class anchor before it
      int curr_t = end_cnt, curr_o;
      if (i2 == CP_NONE || i3 == CP_NONE)                                 // r == -inf (:423-438 / :466-481)
        { if (i2 != CP_NONE)
            curr_o = pr.cnt[other];
          else if (t == CP_HAPLO)
            curr_o = curr_t+COV[CP_HAPLO];
          else
            curr_o = (curr_t/2 > curr_t-COV[CP_HAPLO]) ? curr_t/2 : curr_t-COV[CP_HAPLO];
        }
      else
        { double r = cp_dh_ratio(t,I,view(i2),view(i3),F);
          curr_o = (t == CP_HAPLO) ? (int)(r*curr_t) : (int)((double)curr_t/r);
          c.dhr = r;
        }
      int curr_d = (t == CP_HAPLO) ? curr_o : curr_t;
      int curr_h = (t == CP_HAPLO) ? curr_t : curr_o;
      int curr_r = (int)(P->dr_ratio*curr_d);
      c.pos[CP_HAPLO] = c.pos[CP_DIPLO] = c.pos[CP_REPEAT] = cp_offs(end_pos,F);
      c.cnt[CP_HAPLO]  = curr_h & 0xffff;
      c.cnt[CP_DIPLO]  = curr_d & 0xffff;
      c.cnt[CP_REPEAT] = curr_r & 0xffff;
      if (t == CP_HAPLO) { c.lastDbH = pr.lastD; c.lastH = i; }
      else               { c.lastHbD = pr.lastH; c.lastD = i; }
    }
  if (!((c.cnt[CP_HAPLO] < c.cnt[CP_DIPLO]) && (c.cnt[CP_DIPLO] < c.cnt[CP_REPEAT])))   // :496-498
    c.dp = -INFINITY;
  *out = c;
}

struct cp_aos_view                   // path index -> hot fields, through the `eff` indirection
  { const cp_intvl *rintvl; const int *eff;
    CP_HDM cp_riv operator()(int k) const { return cp_riv_of(rintvl[eff[k]]); }
  };

// One direction of _classify_rel (class_rel.c:515-614) for a read, sequential form.
//   rintvl[M]   reliable intervals (read-only)
//   parent[M*4] back-pointers, eff[M] index of the interval whose data stands in for i (the
//               reference overwrites arg->intvl[i] with its predecessor at "only R" steps, :351)
//   rpos[M]     "absolutely repeat" flags (:350), asgn[M] out
CP_HD void cp_rel_direction(const cp_dev_params *P, const cp_intvl *rintvl, int M, int plen, int F,
                            const int *COV, int8_t *parent, int *eff, uint8_t *rpos, int8_t *asgn)
{ cp_cell prev[4], cur[4];
  int i = F ? 0 : M-1;
  cp_aos_view view; view.rintvl = rintvl; view.eff = eff;

  { const cp_riv I = cp_riv_of(rintvl[i]);                // init, class_rel.c:544-586
    for (int s = 0; s < 4; s++)
      { cp_rel_init_cell(P,s,I,i,plen,F,COV,&prev[s]);
        parent[i*4+s] = (int8_t)s;
      }
    double psum = 0.;
    for (int s = 0; s < 4; s++)
      psum += cp_exp(prev[s].dp);
    for (int s = 0; s < 4; s++)
      prev[s].dp = cp_log(cp_exp(prev[s].dp)/psum);
    rpos[i] = 0;
    eff[i] = i;
  }

  while (true)                                           // class_rel.c:599-605 -> _update (:279-513)
    { const int i_pred = i;
      i = F ? i+1 : i-1;
      if ((F && i >= M) || (!F && i < 0))
        break;
      const cp_riv I = cp_riv_of(rintvl[i]);
      rpos[i] = 0;
      eff[i] = i;

      double tr[16], dp[4];                              // :300-336
      for (int s = 0; s < 4; s++)
        { dp[s] = prev[s].dp;
          if (prev[s].dp == -INFINITY)
            { for (int t = 0; t < 4; t++) tr[s*4+t] = 0.;
              continue;
            }
          for (int t = 0; t < 4; t++)
            tr[s*4+t] = cp_exp(cp_calc_logp(P,t,I,prev[s],F,COV));
        }
      double psum = 0.;
      for (int k = 0; k < 16; k++)
        psum += tr[k];
      if (psum == 0.)                                    // :324-333 (DEBUG build keeps going with E)
        { for (int s = 0; s < 4; s++)
            tr[s*4+CP_ERROR] = 1.;
          psum = 4.;
        }
      for (int k = 0; k < 16; k++)
        tr[k] = cp_log(tr[k]/psum);

      bool only_r = true;                                // :348-380
      for (int s = 0; s < 4; s++)
        { double dummy;
          int maxt = cp_argmax_tr(dp,tr,s,CP_N_STATE,&dummy);
          if (maxt != CP_N_STATE && maxt != CP_REPEAT)
            { only_r = false;
              break;
            }
        }
      if (only_r)
        { rpos[i] = 1;
          eff[i] = eff[i_pred];
          for (int s = 0; s < 4; s++)
            { parent[i*4+s] = (int8_t)s;
              cp_rel_only_r_cell(s,i,&prev[s]);
            }
          continue;
        }

      { double dummy;                                    // :382-386
        int maxs_h = cp_argmax_tr(dp,tr,CP_N_STATE,CP_HAPLO,&dummy);
        int maxs_d = cp_argmax_tr(dp,tr,CP_N_STATE,CP_DIPLO,&dummy);
        if (maxs_h == CP_HAPLO && maxs_d == CP_DIPLO)
          { double mn = tr[CP_HAPLO*4+CP_HAPLO] < tr[CP_DIPLO*4+CP_DIPLO] ? tr[CP_HAPLO*4+CP_HAPLO] : tr[CP_DIPLO*4+CP_DIPLO];
            tr[CP_HAPLO*4+CP_HAPLO] = tr[CP_DIPLO*4+CP_DIPLO] = mn;
          }
      }

      for (int t = 0; t < 4; t++)                        // :390-499
        { double max_logp;
          int max_s = cp_argmax_tr(dp,tr,CP_N_STATE,t,&max_logp);
          parent[i*4+t] = (int8_t)(max_s == CP_N_STATE ? t : max_s);
          cp_rel_target_cell(P,t,i,I,F,COV,max_s,max_logp,prev,view,&cur[t]);
        }
      for (int t = 0; t < 4; t++)
        prev[t] = cur[t];
    }

  // traceback, class_rel.c:606-613
  int last = F ? M-1 : 0;
  double max_logp = -INFINITY;
  int s = CP_ERROR;
  for (int x = 0; x < 4; x++)
    if (max_logp < prev[x].dp)
      { max_logp = prev[x].dp;
        s = x;
      }
  for (int k = last; F ? (k >= 0) : (k < M); k += F ? -1 : 1)
    { asgn[k] = rpos[k] ? (int8_t)CP_REPEAT : (int8_t)s;
      s = parent[k*4+s];
    }
}

// Coverage heuristics after a pass (class_rel.c:629-735 / :743-845), split in two so that the
// lane-parallel kernel can re-run the DP in between.
//   part 1: "no H" check; returns true when the pass must be repeated with COV[H], COV[D] adjusted.
template <class Rv>   // Rv(i) -> cp_riv of reliable interval i
CP_HD bool cp_rel_post1(const cp_dev_params *P, const Rv &rv, int M, int F, const int8_t *asgn, int *COV)
{ const int *G = P->cov;
  for (int i = 0; i < M; i++)
    if (asgn[i] == CP_HAPLO) return false;
  int l, lsum = 0, csum = 0, seed = -1;
  for (int i = 0; i < M; i++)
    if (asgn[i] == CP_DIPLO)
      { cp_riv I = rv(i);
        l = I.e-I.b;
        lsum += l;
        csum += (I.ccb+I.cce)*l/2;
        if (F) { if (seed == -1) seed = i; }             // first D (:641-642)
        else   seed = i;                                 // last D  (:757)
      }
  if (seed < 0) return false;
  double mean_dcov = (double)csum/lsum;
  if (!(mean_dcov < G[CP_DIPLO])) return false;
  cp_riv S = rv(seed);
  COV[CP_HAPLO] = F ? S.ccb : S.cce;
  COV[CP_DIPLO] = (COV[CP_HAPLO]+G[CP_HAPLO]) & 0xffff;
  return true;
}

//   part 2: after an optional re-run (`rerun`), the remaining relabelling rules; returns hdrr.
template <class Rv>
CP_HD double cp_rel_post2(const cp_dev_params *P, const Rv &rv, int M, int F, int8_t *asgn, bool rerun)
{ const int *G = P->cov;
  (void)F;
  if (rerun)                                                 // :651-669
    { bool no_h = true;
      for (int i = 0; i < M; i++)
        if (asgn[i] == CP_HAPLO) no_h = false;
      if (no_h)
        { int l, lsum = 0, csum = 0;
          for (int i = 0; i < M; i++)
            if (asgn[i] == CP_DIPLO)
              { cp_riv I = rv(i);
                l = I.e-I.b;
                lsum += l;
                csum += (I.ccb+I.cce)*l/2;
              }
          double mean_dcov = (double)csum/lsum;
          if (fabs(mean_dcov-G[CP_HAPLO]) <= fabs(mean_dcov-G[CP_DIPLO]))
            for (int i = 0; i < M; i++)
              if (asgn[i] == CP_DIPLO)
                asgn[i] = CP_HAPLO;
        }
    }

  { bool all_h = true;                                       // :674-689
    for (int i = 0; i < M; i++)
      if (asgn[i] != CP_HAPLO) all_h = false;
    if (all_h)
      { int l, lsum = 0, csum = 0;
        for (int i = 0; i < M; i++)
          { cp_riv I = rv(i);
            l = I.e-I.b;
            lsum += l;
            csum += (I.ccb+I.cce)*l/2;
          }
        double mean_hcov = (double)csum/lsum;
        if (fabs(mean_hcov-G[CP_HAPLO]) >= fabs(mean_hcov-G[CP_DIPLO]))
          for (int i = 0; i < M; i++)
            asgn[i] = CP_DIPLO;
      }
  }

  { int n = 0;                                               // :691-712
    for (int i = 0; i < M; i++)
      if (asgn[i] == CP_HAPLO) n++;
    if (n >= M * 0.7)
      { int l, lsum = 0, csum = 0;
        for (int i = 0; i < M; i++)
          if (asgn[i] == CP_HAPLO)
            { cp_riv I = rv(i);
              l = I.e-I.b;
              lsum += l;
              csum += (I.ccb+I.cce)*l/2;
            }
        double mean_hcov = (double)csum/lsum;
        if (fabs(mean_hcov-G[CP_HAPLO]) >= fabs(mean_hcov-G[CP_DIPLO]))
          for (int i = 0; i < M; i++)
            { if (asgn[i] == CP_HAPLO)      asgn[i] = CP_DIPLO;
              else if (asgn[i] == CP_DIPLO) asgn[i] = CP_REPEAT;
            }
      }
  }

  int first_d = -1, last_d = -1, first_h = -1, last_h = -1;  // :714-731
  for (int i = 0; i < M; i++)
    { if (asgn[i] == CP_DIPLO)
        { if (first_d == -1) first_d = i;
          last_d = i;
        }
      else if (asgn[i] == CP_HAPLO)
        { if (first_h == -1) first_h = i;
          last_h = i;
        }
    }
  if (!(first_d >= 0 && first_h >= 0))
    return 1.;
  return ((double)rv(first_d).ccb/rv(first_h).ccb)/((double)rv(last_d).cce/rv(last_h).cce);
}

struct cp_aos_rv
  { const cp_intvl *rintvl;
    CP_HDM cp_riv operator()(int i) const { return cp_riv_of(rintvl[i]); }
  };

// classify_rel_fw / classify_rel_bw (class_rel.c:623-845): one direction plus the coverage
// heuristics, sequential form; returns hdrr.
CP_HD double cp_rel_dir_full(const cp_dev_params *P, const cp_intvl *rintvl, int M, int plen, int F,
                             int8_t *parent, int *eff, uint8_t *rpos, int8_t *asgn)
{ const int *G = P->cov;
  int COV[4] = { G[0], G[1], G[2], G[3] };
  cp_aos_rv rv; rv.rintvl = rintvl;
  cp_rel_direction(P,rintvl,M,plen,F,COV,parent,eff,rpos,asgn);
  bool rerun = cp_rel_post1(P,rv,M,F,asgn,COV);
  if (rerun)
    cp_rel_direction(P,rintvl,M,plen,F,COV,parent,eff,rpos,asgn);
  return cp_rel_post2(P,rv,M,F,asgn,rerun);
}

// Reconcile forward and backward (class_rel.c:904-938).  fw/bw are the two assignments; the choice
// is written to out[M].  `!= true` at :848,:860 compares the state code with 1 (REPEAT) and the
// scans treat any non-ERROR code as set.
CP_HD void cp_rel_reconcile(const int8_t *fw, const int8_t *bw, int M, double hdrr_f, double hdrr_b, int8_t *out)
{ bool eq = true;
  for (int i = 0; i < M; i++)
    if (fw[i] != bw[i]) { eq = false; break; }
  bool take_bw = false;
  if (!eq)
    { bool pre = (fw[0] == 1);
      if (pre)
        { int i = 0;
          while (i < M && fw[i]) i++;
          while (i < M) { if (fw[i]) { pre = false; break; } i++; }
        }
      if (!pre)
        { bool suf = (fw[M-1] == 1);
          if (suf)
            { int i = M-2;
              while (i >= 0 && fw[i]) i--;
              while (i >= 0) { if (fw[i]) { suf = false; break; } i--; }
            }
          if (suf) take_bw = true;
          else if (!(fabs(hdrr_f-1.) <= fabs(hdrr_b-1.))) take_bw = true;
        }
    }
  for (int i = 0; i < M; i++)
    out[i] = take_bw ? bw[i] : fw[i];
}

// ---------------------------------------------------------------------------------------------
//  class_unrel.c
// ---------------------------------------------------------------------------------------------
// class_unrel.c:11-25
CP_HD void cp_find_nn_u(int idx, int s, const cp_intvl *intvl, int N, int *l_out, int *r_out)
{ int l = idx-1;
  while (l >= 0 && !(intvl[l].asgn == (int8_t)s && intvl[l].is_rel)) l--;
  *l_out = (l < 0) ? -1 : l;
  int r = idx+1;
  while (r < N && !(intvl[r].asgn == (int8_t)s && intvl[r].is_rel)) r++;
  *r_out = (r >= N) ? -1 : r;
}

// class_unrel.c:27-51 (the recursion unrolled: from_est == true is the inner call)
CP_HD int cp_est_cov_1(int x, int l, int r, const cp_intvl *intvl)
{ if (l != -1 && r != -1)
    return (int)(uint16_t)cp_linear_interpolation(x,intvl[l].e-1,intvl[l].cce,intvl[r].b,intvl[r].ccb);
  else if (l != -1) return intvl[l].cce;
  else if (r != -1) return intvl[r].ccb;
  return -1;
}

CP_HD int cp_est_cov(const cp_dev_params *P, int x, int idx, const cp_intvl *intvl, int N, int s, int l, int r)
{ int c = cp_est_cov_1(x,l,r,intvl);
  if (c >= 0) return c;
  int ol, orr;
  cp_find_nn_u(idx,(s == CP_HAPLO) ? CP_DIPLO : CP_HAPLO,intvl,N,&ol,&orr);
  int cov = cp_est_cov_1(x,ol,orr,intvl);
  if (cov < 0) cov = 0;
  if (cov > 0)
    return ((s == CP_HAPLO) ? cov/2 : cov*2) & 0xffff;
  return P->cov[s];
}

// class_unrel.c:53-65
CP_HD double cp_logp_e_u(const cp_dev_params *P, const cp_intvl &I)
{ double logp_po = cp_logp_poisson(P,I.cb,P->cov[CP_ERROR])+cp_logp_poisson(P,I.ce,P->cov[CP_ERROR])+CP_E_PO_BASE;
  return (I.pe > logp_po) ? I.pe : logp_po;
}

// class_unrel.c:67-113
CP_HD double cp_logp_r_u(const cp_dev_params *P, int idx, const cp_intvl *intvl, int N)
{ const cp_intvl &I = intvl[idx];
  if ((I.cb > I.ce ? I.cb : I.ce) >= P->cov[CP_REPEAT])
    return 0.;
  int l, r;
  cp_find_nn_u(idx,CP_DIPLO,intvl,N,&l,&r);
  int dcov_l, dcov_r;
  if (l == -1 && r == -1) dcov_l = dcov_r = P->cov[CP_DIPLO];
  else if (l == -1)       dcov_l = dcov_r = intvl[r].cb;
  else if (r == -1)       dcov_l = dcov_r = intvl[l].ce;
  else                    { dcov_l = intvl[l].ce; dcov_r = intvl[r].cb; }
  int rcov_l = (uint16_t)(P->dr_ratio*dcov_l);
  int rcov_r = (uint16_t)(P->dr_ratio*dcov_r);
  if (I.cb >= rcov_l || I.ce >= rcov_r)
    return CP_R_LOGP;
  return cp_logp_binom_pre(P->logfact,I.cb,rcov_l,P->r_lp,P->r_l1mp)+cp_logp_binom_pre(P->logfact,I.ce,rcov_r,P->r_lp,P->r_l1mp);
}

// class_unrel.c:115-175
CP_HD double cp_logp_hd_u(const cp_dev_params *P, int s, int idx, const cp_intvl *intvl, int N)
{ const cp_intvl &I = intvl[idx];
  int l_rel, r_rel;
  cp_find_nn_u(idx,s,intvl,N,&l_rel,&r_rel);
  double logp_l, logp_r;
  { double er = -INFINITY, sf = -INFINITY, sf_er = -INFINITY;
    if (idx-1 >= 0 && intvl[idx-1].asgn == (int8_t)s)
      er = I.peo_b;
    if (l_rel != -1)
      sf = cp_logp_trans(P,intvl[l_rel].e-1,I.b,intvl[l_rel].cce,I.cb,intvl[l_rel].cce);
    int est = cp_est_cov(P,I.b,idx,intvl,N,s,l_rel,r_rel);
    if (est >= I.cb)
      sf_er = cp_logp_uerr(P,est,I.cb);
    double m = (er > sf) ? er : sf;
    logp_l = (m > sf_er) ? m : sf_er;
  }
  { double er = -INFINITY, sf = -INFINITY, sf_er = -INFINITY;
    if (idx+1 < N && intvl[idx+1].asgn == (int8_t)s)
      er = I.peo_e;
    if (r_rel != -1)
      sf = cp_logp_trans(P,I.e-1,intvl[r_rel].b,I.ce,intvl[r_rel].ccb,intvl[r_rel].ccb);
    int est = cp_est_cov(P,I.e-1,idx,intvl,N,s,l_rel,r_rel);
    if (est >= I.ce)
      sf_er = cp_logp_uerr(P,est,I.ce);
    double m = (er > sf) ? er : sf;
    logp_r = (m > sf_er) ? m : sf_er;
  }
  if (logp_l == -INFINITY && logp_r == -INFINITY)
    { logp_l = cp_logp_poisson(P,I.cb,P->cov[s]);
      logp_r = cp_logp_poisson(P,I.ce,P->cov[s]);
    }
  else if (logp_l == -INFINITY) logp_l = logp_r;
  else if (logp_r == -INFINITY) logp_r = logp_l;
  return logp_l+logp_r;
}

// class_unrel.c:192-236: argmax in enum order E,R,H,D with strict '<' (E wins ties)
CP_HD void cp_update_state(const cp_dev_params *P, int idx, cp_intvl *intvl, int N)
{ const cp_intvl &I = intvl[idx];
  if ((I.cb > I.ce ? I.cb : I.ce) >= P->cov[CP_REPEAT])
    { intvl[idx].asgn = CP_REPEAT;
      return;
    }
  double logpmax = -INFINITY;
  int smax = -1;
  for (int s = 0; s < 4; s++)
    { double logp;
      if (s == CP_ERROR)       logp = cp_logp_e_u(P,I);
      else if (s == CP_REPEAT) logp = cp_logp_r_u(P,idx,intvl,N);
      else                     logp = cp_logp_hd_u(P,s,idx,intvl,N);
      if (logpmax < logp)
        { logpmax = logp;
          smax = s;
        }
    }
  if (smax >= 0)
    intvl[idx].asgn = (int8_t)smax;
}
