// cp_libm.h -- exp() and log() for double, bit for bit as glibc 2.35's libm computes them on an x86-64 host with FMA.
//
// Why: the reference's numbers are whatever its libm returns (prob.c, class_rel.c, class_unrel.c, wall.c call exp/log),
// and the decision path compares such values: classify_unrel's argmax (class_unrel.c:192-246) meets log(px*py) against
// log(px)+log(py) -- equal in mathematics, apart by the last bit or two in floating point -- whenever an interval's
// SELF and OTHERS error probabilities coincide, and which way that falls is a property of the log routine.  ROCm's
// ocml exp/log are good to about an ulp but are different routines: one such read in 3 000 adversarial ones got a
// different label (scripts/fuzz_parity.py, seed 305).  With the host's own routine on the device there is no tolerance
// to state: every double of the stage API is bit-identical to the oracle's.
//
// What: Szabolcs Nagy's exp and log (ARM optimized-routines, MIT licence; glibc >= 2.28 sysdeps/ieee754/dbl-64/e_exp.c,
// e_log.c), in the operation order of the variants glibc's ifunc selects on a CPU with FMA (__ieee754_exp_fma,
// __ieee754_log_fma of libm-2.35: compiled with -mfma, where the compiler fused the multiply-adds below; the order was
// read off that object code).  Every a*b+c that is ONE rounding there is an explicit fma here, everything else is
// left unfused by -ffp-contract=off.  Tables: cp_libm_tables.h (generated from the same libm).  Rounding mode: nearest
// only; errno and exception flags: none (the reference checks neither).
// Checked: tests/test_libm.py -- host build against the host's libm on 4e8 arguments (whole range, the decision path's
// ranges, every branch), device build against the host's libm on the GPU (`-m gpu`).  On a host WITHOUT FMA glibc picks
// the other variant, whose results differ in the last bit now and then; the tests would say so.
#pragma once
#include <stdint.h>
#include "cp_types.h"
#include "cp_libm_tables.h"

#ifndef __HIPCC__
#undef  CP_HD
#define CP_HD static inline
#endif

namespace cp_libm {
constexpr uint64_t exp_tab[256] = CP_EXP_TAB_INIT;
constexpr double   log_tab[256] = CP_LOG_TAB_INIT;
}

CP_HD uint64_t cp_asuint64(double x) { return __builtin_bit_cast(uint64_t,x); }
CP_HD double   cp_asdouble(uint64_t u) { return __builtin_bit_cast(double,u); }

#ifdef CP_LIBM_OCML                        // diagnostic builds only (scripts/build_diag.sh): ROCm's ocml routines, to time the difference
#include <math.h>
CP_HD double cp_exp(double x) { return exp(x); }
CP_HD double cp_log(double x) { return log(x); }
template <class TAB> CP_HD double cp_exp_t(double x, TAB) { return exp(x); }
template <class TAB> CP_HD double cp_log_t(double x, TAB) { return log(x); }
#else
// e_exp.c: exp(x) = 2^(k/128) * exp(r), x = k ln2/128 + r, |r| <= ln2/256; 2^(k/128) ~= scale*(1+tail) from the table
// (Tried: the main paths computed unconditionally and the special values -- exp(-inf), log(0), both common on the DP's
//  impossible transitions -- selected afterwards, log's two evaluations both computed: same bits, fewer exec-mask
//  instructions, but k_classify_rel_grp 1.41 -> 1.52 ms: the second polynomial costs more than the branches.)
template <class TAB>
CP_HD double cp_exp_t(double x, TAB exp_tab)
{ const uint64_t ix = cp_asuint64(x);
  uint32_t abstop = (uint32_t)(ix >> 52) & 0x7ff;
  if (abstop-0x3c9u >= 0x3fu)                                   // |x| < 2^-54 or |x| >= 512 or not finite
    { if (abstop-0x3c9u >= 0x80000000u)
        return 1.0+x;                                           // tiny
      if (abstop >= 0x409u)                                     // |x| >= 1024, inf, nan
        { if (ix == 0xfff0000000000000ull) return 0.0;
          if (abstop >= 0x7ffu) return 1.0+x;
          return (ix >> 63) ? 0.0 : cp_asdouble(0x7ff0000000000000ull);
        }
      abstop = 0;                                               // 512 <= |x| < 1024: the scale may leave the normal range
    }
  const double kd0 = __builtin_fma(x,CP_EXP_INVLN2N,CP_EXP_SHIFT);
  const uint64_t ki = cp_asuint64(kd0);
  const double kd = kd0-CP_EXP_SHIFT;
  double r = __builtin_fma(kd,CP_EXP_NEGLN2HIN,x);
  r = __builtin_fma(kd,CP_EXP_NEGLN2LON,r);
  const uint64_t idx = 2*(ki & 127);
  const uint64_t top = ki << 45;
  const double tail = cp_asdouble(exp_tab[idx]);
  uint64_t sbits = exp_tab[idx+1]+top;
  const double r2 = r*r;
  const double p23 = __builtin_fma(r,CP_EXP_C3,CP_EXP_C2);
  const double p45 = __builtin_fma(r,CP_EXP_C5,CP_EXP_C4);
  double tmp = __builtin_fma(p23,r2,tail+r);
  tmp = __builtin_fma(r2*r2,p45,tmp);
  if (abstop == 0)
    { if ((ki & 0x80000000ull) == 0)                            // k > 0: the exponent of scale may have overflowed
        { sbits -= 1009ull << 52;
          const double scale = cp_asdouble(sbits);
          return 0x1p1009*__builtin_fma(scale,tmp,scale);
        }
      sbits += 1022ull << 52;                                   // k < 0: care in the subnormal range
      const double scale = cp_asdouble(sbits);
      const double st = scale*tmp;
      double y = scale+st;
      if (y < 1.0)
        { double lo = scale-y;
          lo = lo+st;
          const double hi = 1.0+y;
          double t = 1.0-hi;
          t = t+y;
          t = t+lo;
          y = (t+hi)-1.0;
          if (y == 0.0) y = 0.0;
        }
      return 0x1p-1022*y;
    }
  const double scale = cp_asdouble(sbits);
  return __builtin_fma(scale,tmp,scale);
}

// e_log.c: x = 2^k z, z in [0x1.6p-1, 0x1.6p0); log(x) = log1p(z/c-1) + log(c) + k ln2 with c near the centre of z's
// subinterval (128 of them); arguments near 1 take a degree-12 polynomial with a double-double head instead
// (the table through a pointer of the caller's choosing: a kernel whose DP step has exp -> log on its critical path keeps
//  a copy of both tables in LDS -- from global memory each look-up is a dependent cache miss in the middle of the chain)
CP_HD double cp_exp(double x) { return cp_exp_t(x,cp_libm::exp_tab); }

template <class TAB>
CP_HD double cp_log_t(double x, TAB log_tab)
{ uint64_t ix = cp_asuint64(x);
  const uint32_t top = (uint32_t)(ix >> 48);
  if (ix-0x3fee000000000000ull < 0x3090000000000ull)            // 1-2^-4 <= x < 1+0x1.09p-4
    { if (ix == 0x3ff0000000000000ull) return 0.0;
      const double r = x-1.0;
      const double r2 = r*r;
      const double r3 = r*r2;
      double a = __builtin_fma(r,CP_LOG_B2,CP_LOG_B1);
      double b = __builtin_fma(r,CP_LOG_B5,CP_LOG_B4);
      double c = __builtin_fma(r,CP_LOG_B8,CP_LOG_B7);
      a = __builtin_fma(r2,CP_LOG_B3,a);
      b = __builtin_fma(r2,CP_LOG_B6,b);
      c = __builtin_fma(r2,CP_LOG_B9,c);
      c = __builtin_fma(r3,CP_LOG_B10,c);
      double p = __builtin_fma(c,r3,b);
      p = __builtin_fma(p,r3,a);
      const double rw = __builtin_fma(r,0x1p27,r);
      const double rhi = __builtin_fma(-0x1p27,r,rw);
      const double rlo = r-rhi;
      const double q = rhi*rhi;
      const double hi = __builtin_fma(q,CP_LOG_B0,r);
      double lo = r-hi;
      lo = __builtin_fma(q,CP_LOG_B0,lo);
      lo = __builtin_fma(CP_LOG_B0*rlo,r+rhi,lo);
      const double y = __builtin_fma(p,r3,lo);
      return y+hi;
    }
  if (top-0x0010u >= 0x7fe0u)                                   // x < 2^-1022, inf, nan
    { if (ix*2 == 0) return -cp_asdouble(0x7ff0000000000000ull);
      if (ix == 0x7ff0000000000000ull) return x;
      if ((top & 0x8000u) || (top & 0x7ff0u) == 0x7ff0u)
        return (x-x)/(x-x);                                     // negative or nan: nan
      ix = cp_asuint64(x*0x1p52);                               // subnormal: normalise
      ix -= 52ull << 52;
    }
  const uint64_t tmp = ix-0x3fe6000000000000ull;
  const int i = (int)((tmp >> 45) & 127);
  const int k = (int)((int64_t)tmp >> 52);
  const uint64_t iz = ix-(tmp & 0xfff0000000000000ull);
  const double invc = log_tab[2*i], logc = log_tab[2*i+1];
  const double z = cp_asdouble(iz);
  const double r = __builtin_fma(z,invc,-1.0);
  const double kd = (double)k;
  const double w = __builtin_fma(kd,CP_LOG_LN2HI,logc);
  const double p12 = __builtin_fma(r,CP_LOG_A2,CP_LOG_A1);
  const double hi = r+w;
  const double r2 = r*r;
  double lo = w-hi;
  lo = lo+r;
  lo = __builtin_fma(kd,CP_LOG_LN2LO,lo);
  const double rr2 = r*r2;
  const double p34 = __builtin_fma(r,CP_LOG_A4,CP_LOG_A3);
  lo = __builtin_fma(r2,CP_LOG_A0,lo);
  const double p = __builtin_fma(p34,r2,p12);
  const double y = __builtin_fma(rr2,p,lo);
  return y+hi;
}
CP_HD double cp_log(double x) { return cp_log_t(x,cp_libm::log_tab); }
#endif
