// cp_math.h -- scalar numeric primitives of the hot path as device functions.
//   bessel.c:390-521 (bessi0/bessi1/bessi), prob.c:33-112, util.c:9-55 of the reference.
// All arithmetic is IEEE double in the reference's evaluation order; the translation unit is built
// with -ffp-contract=off so no FMA is formed.  log() of constants comes from host-built tables
// (cp_dev_params); the device evaluates exp/log/sqrt only on data-dependent values.
//
// The functions are `__host__ __device__` only so that tests/ can unit-test them against the oracle
// in a GPU-less container (tests/host_harness.cpp); the product never runs them on the CPU.
#pragma once
#include <math.h>
#include "cp_types.h"
#include "cp_libm.h"

#ifndef __HIPCC__
#undef  CP_HD
#define CP_HD static inline
#endif

// ---- bessel.c:390-411 ----------------------------------------------------------------------
CP_HD double cp_bessi0(double x)
{ double ax = fabs(x), y, ans;
  if (ax < 3.75)
    { y = x/3.75; y = y*y;
      ans = 1.0+y*(3.5156229+y*(3.0899424+y*(1.2067492
            +y*(0.2659732+y*(0.360768e-1+y*0.45813e-2)))));
    }
  else
    { y = 3.75/ax;
      ans = (cp_exp(ax)/sqrt(ax))*(0.39894228+y*(0.1328592e-1
            +y*(0.225319e-2+y*(-0.157565e-2+y*(0.916281e-2
            +y*(-0.2057706e-1+y*(0.2635537e-1+y*(-0.1647633e-1
            +y*0.392377e-2))))))));
    }
  return ans;
}

// ---- bessel.c:416-438 ----------------------------------------------------------------------
CP_HD double cp_bessi1(double x)
{ double ax = fabs(x), y, ans;
  if (ax < 3.75)
    { y = x/3.75; y = y*y;
      ans = ax*(0.5+y*(0.87890594+y*(0.51498869+y*(0.15084934
            +y*(0.2658733e-1+y*(0.301532e-2+y*0.32411e-3))))));
    }
  else
    { y = 3.75/ax;
      ans = 0.2282967e-1+y*(-0.2895312e-1+y*(0.1787654e-1
            -y*0.420059e-2));
      ans = 0.39894228+y*(-0.3988024e-1+y*(-0.362018e-2
            +y*(0.163801e-2+y*(-0.1031555e-1+y*ans))));
      ans *= (cp_exp(ax)/sqrt(ax));
    }
  return x < 0.0 ? -ans : ans;
}

// ---- bessel.c:478-521: downward recurrence, 2*(n+(int)sqrt(40 n)) steps, 1e10 rescale ------
// The rescale happens a handful of times per call; on the device it is kept out of the hot loop by a
// wave-level vote (otherwise the compiler if-converts it into three extra multiplies and six selects
// per step).  `dj` mirrors the integer counter exactly, so dj*tox*bi == j*tox*bi bit for bit.
#if defined(__HIP_DEVICE_COMPILE__)
#define CP_ANY_LANE(c) __any(c)
#define CP_NO_IFCVT() asm volatile("" ::: "memory")     /* keeps the rare block a real branch */
#else
#define CP_ANY_LANE(c) (c)
#define CP_NO_IFCVT() ((void)0)
#endif
// One step of the recurrence for counter value dj: (older,newer) = (a,b) -> b becomes the older value
// and the returned sum the newer one.  `ans` takes part in the rescale only once it has been captured.
#define CP_BESSI_STEP(t,a,b,WITH_ANS)                                   \
  t = a+dj*tox*b;                                                       \
  dj -= 1.0;                                                            \
  if (CP_ANY_LANE(fabs(t) > 1.0e10))                                    \
    { CP_NO_IFCVT();                                                    \
      if (fabs(t) > 1.0e10)                                             \
        { if (WITH_ANS) ans *= 1.0e-10;                                 \
          t *= 1.0e-10; b *= 1.0e-10;                                   \
        }                                                               \
    }

// One step with the rescale applied as a multiplication by 1.0 or 1e-10 (multiplying by 1.0 is exact, so the
// values are those of the reference's `if`).  scripts/microbench/bessi_bench.hip, ms for the same work with
// 1 / 3 / 8 waves per SIMD:  branch per step, unrolled by 2: 1.67 / 2.11 / 4.44;  unrolled by 4: 1.32 /
// 2.01 / 4.13;  one test per 2 or 4 steps with an exact replay of a tripped group: slower than either;
// no test at all (inexact bound): 0.73 / 1.17 / 2.67;  this form, unrolled by 4: 0.84 / 1.93 / 4.66.  The
// classify kernels run at 1-3 waves per SIMD and their long reads finish alone, so this form is used.
#define CP_BESSI_STEPF(t,a,b,WITH_ANS)                                  \
  { t = a+dj*tox*b;                                                     \
    dj -= 1.0;                                                          \
    const double s_ = fabs(t) > 1.0e10 ? 1.0e-10 : 1.0;                 \
    t *= s_; b *= s_;                                                   \
    if (WITH_ANS) ans *= s_;                                            \
  }
#define CP_BESSI_LOOP(WITH_ANS)                                         \
  for (; c >= 4; c -= 4)                                                \
    { CP_BESSI_STEPF(t1,a,b,WITH_ANS)                                   \
      CP_BESSI_STEPF(t2,b,t1,WITH_ANS)                                  \
      CP_BESSI_STEPF(t3,t1,t2,WITH_ANS)                                 \
      CP_BESSI_STEPF(t4,t2,t3,WITH_ANS)                                 \
      a = t3; b = t4;                                                   \
    }                                                                   \
  if (c >= 2)                                                           \
    { CP_BESSI_STEPF(t1,a,b,WITH_ANS)                                   \
      CP_BESSI_STEPF(t2,b,t1,WITH_ANS)                                  \
      a = t1; b = t2; c -= 2;                                           \
    }                                                                   \
  if (c == 1)                                                           \
    { CP_BESSI_STEPF(t1,a,b,WITH_ANS)                                   \
      a = b; b = t1;                                                    \
    }

CP_HD double cp_bessi(int n, double x)
{ if (n == 0) return cp_bessi0(x);
  if (n == 1) return cp_bessi1(x);
  if (x == 0.0) return 0.0;
  const double tox = 2.0/fabs(x);
  const int jmax = 2*(n+(int)sqrt(40.0*n));
  double dj = (double)jmax;
  double a = 0.0, b = 1.0, t1, t2, t3, t4, ans = 0.0;    // a = bip (older), b = bi (newer)
  // j = jmax .. n: ans is still 0, so the reference's `ans *= BIGNI` is a no-op here (bessel.c:500-509)
  int c = jmax-n+1;
  CP_BESSI_LOOP(false)
  ans = a;                                               // `if (j == n) ans = bip` (bessel.c:510)
  // j = n-1 .. 1
  c = n-1;
  CP_BESSI_LOOP(true)
  ans *= cp_bessi0(x)/b;
  return (x < 0.0 && (n & 1)) ? -ans : ans;
}

// ---- prob.c:22-30: counts are cnt_t (uint16) clamped to MAX_KMER_CNT (DEBUG build) ---------
CP_HD int cp_check_cnt(int n)
{ n &= 0xffff;
  return n > CP_MAX_KMER_CNT ? CP_MAX_KMER_CNT : n;
}

// ---- prob.c:33-39 ---------------------------------------------------------------------------
CP_HD double cp_logp_poisson(const cp_dev_params *P, int k, int lambda)
{ k = cp_check_cnt(k);
  double ll = (lambda >= 0 && lambda <= CP_MAX_KMER_CNT) ? P->logint[lambda] : cp_log((double)lambda);
  return k * ll - lambda - P->logfact[k];
}

// ---- prob.c:41-44 ---------------------------------------------------------------------------
CP_HD double cp_logp_skellam(int k, double lambda)
{ return -2. * lambda + cp_log(cp_bessi(k < 0 ? -k : k,2.*lambda)); }

// ---- prob.c:59-73 with log(p), log(1-p) supplied ----------------------------------------------
// `lf` is anything indexable like the log-factorial table: P->logfact, or an LDS-backed accessor.
template <class LF>
CP_HD double cp_logp_binom_pre(const LF &lf, int k, int n, double lpe, double l1mpe)
{ k = cp_check_cnt(k);
  n = cp_check_cnt(n);
  return lf[n] - lf[k] - lf[n-k] + k * lpe + (n-k) * l1mpe;
}

// ---- prob.c:76-112, exact == false ------------------------------------------------------------
template <class LF>
CP_HD double cp_binom_test_g(const LF &P, int k, int n, double pe, double lpe, double l1mpe)
{ k = cp_check_cnt(k);
  n = cp_check_cnt(n);
  const double mean = n * pe;
  double s, p_first, p_curr;
  if ((double)k >= mean)
    { s = p_first = cp_exp(cp_logp_binom_pre(P,k,n,lpe,l1mpe));
      for (int x = k+1; x <= n; x++)
        { s += p_curr = cp_exp(cp_logp_binom_pre(P,x,n,lpe,l1mpe));
          if (10 * p_curr < p_first)
            break;
        }
    }
  else
    { s = p_first = (k == 0) ? 0. : cp_exp(cp_logp_binom_pre(P,k-1,n,lpe,l1mpe));
      for (int x = k-2; x >= 0; x--)
        { s += p_curr = cp_exp(cp_logp_binom_pre(P,x,n,lpe,l1mpe));
          if (10 * p_curr < p_first)
            break;
        }
      s = 1-s;
    }
  return s;
}

// ---- util.c:46-55 -----------------------------------------------------------------------------
template <class LF>
CP_HD double cp_p_errorin(const LF &lf, int e, double erate, double lpe, double l1mpe, int cout, int cin)
{ return cp_binom_test_g(lf,(e == CP_SELF) ? cin : cout-cin,cout,erate,lpe,l1mpe); }

// ---- util.c:35-44 -----------------------------------------------------------------------------
// (double)cov*d is an exact integer, so the value is a function of (|ce-cb|, cov*d): the table of cp_dev_params holds
// what the line below gives for the pairs it covers (k_skellam_table runs this very function).
// Out of line on the device: with the table in place this is the rare path, and inlined into the classify kernels its
// Bessel recurrence would set their register count (and so their occupancy) for nothing.
#if defined(__HIPCC__)
__host__ __device__ __attribute__((noinline))
#else
static inline
#endif
double cp_logp_trans_calc(const cp_dev_params *P, int k, double cd)
{ return cp_logp_skellam(k,cd/P->read_len); }

CP_HD double cp_logp_trans(const cp_dev_params *P, int b, int e, int cb, int ce, int cov)
{ cov &= 0xffff;
  int d = e-b;
  if (d < 0) d = -d;
#if defined(__HIP_DEVICE_COMPILE__)
  const int k = ce-cb < 0 ? cb-ce : ce-cb;
  const long long cd = (long long)cov*d;
  if (P->skel && k <= P->skel_kmax && cd <= P->skel_cdmax)
    return P->skel[cd*(P->skel_kmax+1)+k];
#endif
  return cp_logp_trans_calc(P,ce-cb,(double)cov*d);
}

// exp(logp_trans(...)): what the DP step of classify_rel needs of a transition (class_rel.c:300-319 exponentiates every
// one of them).  Inside the table the value was computed by cp_exp_t(cp_logp_trans_calc(...)) -- this very expression;
// outside it runs here.  `xt`: the caller's exp table (cp_libm.h).
template <class TAB>
CP_HD double cp_exp_logp_trans(const cp_dev_params *P, int b, int e, int cb, int ce, int cov, TAB xt)
{ cov &= 0xffff;
  int d = e-b;
  if (d < 0) d = -d;
#if defined(__HIP_DEVICE_COMPILE__)
  const int k = ce-cb < 0 ? cb-ce : ce-cb;
  const long long cd = (long long)cov*d;
  if (k <= P->skel_kmax && cd <= P->skel_cdmax)
    { if (P->eskel) return P->eskel[cd*(P->skel_kmax+1)+k];
      if (P->skel)  return cp_exp_t(P->skel[cd*(P->skel_kmax+1)+k],xt);
    }
#endif
  return cp_exp_t(cp_logp_trans_calc(P,ce-cb,(double)cov*d),xt);
}

// ---- p_errorin with one of the model's own error rates pe[t][l] (every call of the candidate walk) ----
#if defined(__HIPCC__)
__host__ __device__ __attribute__((noinline))
#else
static inline
#endif
double cp_p_errorin_calc(const cp_dev_params *P, int e, int t, int l, int cout, int cin)
{ return cp_p_errorin(P->logfact,e,P->pe[t][l],P->lpe[t][l],P->l1mpe[t][l],cout,cin); }

CP_HD double cp_p_errorin_tl(const cp_dev_params *P, int e, int t, int l, int cout, int cin)
{
#if defined(__HIP_DEVICE_COMPILE__)
  if (P->petab && cin >= 0 && cin <= cout && cout <= P->pe_cmax)
    return P->petab[((long long)(e*63+t*21+l)*(P->pe_cmax+1)+cout)*(P->pe_cmax+1)+cin];
#endif
  return cp_p_errorin_calc(P,e,t,l,cout,cin);
}

// ---- class_unrel.c:137-147: log P(count c | estimated count est, max_erate 0.1), est >= c ------
#if defined(__HIPCC__)
__host__ __device__ __attribute__((noinline))
#else
static inline
#endif
double cp_logp_uerr_calc(const cp_dev_params *P, int est, int c)
{ return cp_log(cp_p_errorin(P->logfact,CP_OTHERS,0.1,P->u_lpe,P->u_l1mpe,est,c)); }

CP_HD double cp_logp_uerr(const cp_dev_params *P, int est, int c)
{
#if defined(__HIP_DEVICE_COMPILE__)
  if (P->uerr && est <= P->uerr_max && c >= 0 && c <= est)
    return P->uerr[(long long)est*(P->uerr_max+1)+c];
#endif
  return cp_logp_uerr_calc(P,est,c);
}

// ---- util.c:24-33 (positions are strictly ordered on every call path) -------------------------
CP_HD double cp_linear_interpolation(int x, int pos1, int cnt1, int pos2, int cnt2)
{ return (double)cnt1+((double)cnt2-cnt1)*(x-pos1)/(pos2-pos1); }
