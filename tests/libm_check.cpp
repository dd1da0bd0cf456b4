// tests/libm_check.cpp -- cp_libm.h (the product's exp/log, compiled for the host) against the host's libm, bit for bit.
//   libm_check <millions of arguments per function and range> <threads>      prints "ok <n>" or the first mismatches
// Also exports cp_libm_eval() so the GPU test can get the host libm's answers for the arguments it sends to the device.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <vector>
#include <atomic>
#include "../classpro_amd/csrc/cp_libm.h"

static inline uint64_t splitmix(uint64_t &s)
{ uint64_t z = (s += 0x9e3779b97f4a7c15ull);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
static inline double u01(uint64_t &s) { return (double)(splitmix(s) >> 11) * 0x1p-53; }

// argument families: 0 = any bit pattern; 1 = exp on the decision path's range (log-probabilities, -800..+40);
// 2 = log of probabilities and ratios (1e-320 .. 1e6, dense near 1); 3 = around the routines' branch points
static double arg_exp(int fam, uint64_t &s)
{ switch (fam)
    { case 0: return cp_asdouble(splitmix(s));
      case 1: return -800.0+840.0*u01(s);
      case 2: return (u01(s)-0.5)*ldexp(1.0,(int)(splitmix(s) % 70)-60);
      default:
        { static const double c[] = { 0x1p-54, 512.0, 1024.0, -512.0, -1024.0, -708.3964185322641, -745.1332191019412, 709.782712893384, 0.0 };
          const double b = c[splitmix(s) % 9];
          return b+(u01(s)-0.5)*ldexp(1.0,(int)(splitmix(s) % 12)-10);
        }
    }
}
static double arg_log(int fam, uint64_t &s)
{ switch (fam)
    { case 0: return cp_asdouble(splitmix(s));
      case 1: return exp(-750.0+760.0*u01(s));
      case 2: return 1.0+(u01(s)-0.5)*ldexp(1.0,(int)(splitmix(s) % 54)-52);
      default:
        { static const double c[] = { 0.9375, 1.064697265625, 0x1.6p-1, 0x1.6p0, 0x1p-1022, 1.0, 2.0, 0.5 };
          const double b = c[splitmix(s) % 8];
          return b*(1.0+(u01(s)-0.5)*ldexp(1.0,(int)(splitmix(s) % 50)-50));
        }
    }
}
static inline bool same(double a, double b)
{ return cp_asuint64(a) == cp_asuint64(b) || (a != a && b != b); }

extern "C" void cp_libm_eval(int fn, const double *x, double *y, long n)     // the HOST libm's answers
{ for (long i = 0; i < n; i++) y[i] = fn == 0 ? exp(x[i]) : log(x[i]); }
extern "C" void cp_libm_eval_product(int fn, const double *x, double *y, long n)   // cp_libm.h compiled for the host
{ for (long i = 0; i < n; i++) y[i] = fn == 0 ? cp_exp(x[i]) : cp_log(x[i]); }
extern "C" void cp_libm_args(int fn, int fam, uint64_t seed, double *x, long n)
{ uint64_t s = seed; for (long i = 0; i < n; i++) x[i] = fn == 0 ? arg_exp(fam,s) : arg_log(fam,s); }

#ifndef LIBM_CHECK_NO_MAIN
int main(int argc, char **argv)
{ const long per = (argc > 1 ? atol(argv[1]) : 10)*1000000L;
  const int nt = argc > 2 ? atoi(argv[2]) : 8;
  std::atomic<long> bad(0), done(0);
  std::vector<std::thread> th;
  for (int t = 0; t < nt; t++)
    th.emplace_back([&,t]()
      { for (int fn = 0; fn < 2; fn++)
          for (int fam = 0; fam < 4; fam++)
            { uint64_t s = 0x1234567ull*(t+1)+fn*977+fam*131;
              for (long i = t; i < per; i += nt)
                { const double x = fn == 0 ? arg_exp(fam,s) : arg_log(fam,s);
                  const double a = fn == 0 ? cp_exp(x) : cp_log(x), b = fn == 0 ? exp(x) : log(x);
                  if (!same(a,b))
                    { if (bad.fetch_add(1) < 10)
                        fprintf(stderr,"%s(%a) = %a, libm %a\n",fn == 0 ? "exp" : "log",x,a,b);
                    }
                  done.fetch_add(1,std::memory_order_relaxed);
                }
            }
      });
  for (auto &x : th) x.join();
  if (bad.load()) { printf("MISMATCH %ld of %ld\n",bad.load(),done.load()); return 1; }
  printf("ok %ld\n",done.load());
  return 0;
}
#endif
