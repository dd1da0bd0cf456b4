// Diagnostic micro-benchmark (not part of the product): cycles per step of the modified-Bessel downward
// recurrence (bessel.c:478-521) for several ways of placing the |bi| > 1e10 rescale test, with one wave
// per SIMD and with several.  Every variant must return the same bits as V0.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scripts/microbench/bessi_bench.hip -o build/bessi_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

#define STEP(t,a,b,WITH_ANS)                                   \
  t = a+dj*tox*b;                                              \
  dj -= 1.0;                                                   \
  if (__any(fabs(t) > 1.0e10))                                 \
    { asm volatile("" ::: "memory");                           \
      if (fabs(t) > 1.0e10)                                    \
        { if (WITH_ANS) ans *= 1.0e-10;                        \
          t *= 1.0e-10; b *= 1.0e-10;                          \
        }                                                      \
    }

// per-step test on the high word first (conservative), exact test only inside
#define STEPI(t,a,b,WITH_ANS)                                  \
  t = a+dj*tox*b;                                              \
  dj -= 1.0;                                                   \
  if (__any(((unsigned)__double2hiint(t) & 0x7fffffffu) >= 0x4202A05Fu)) \
    { asm volatile("" ::: "memory");                           \
      if (fabs(t) > 1.0e10)                                    \
        { if (WITH_ANS) ans *= 1.0e-10;                        \
          t *= 1.0e-10; b *= 1.0e-10;                          \
        }                                                      \
    }
#define STEPF(t,a,b,WITH_ANS)                                  \
  { t = a+dj*tox*b; dj -= 1.0;                                 \
    double s_ = fabs(t) > 1.0e10 ? 1.0e-10 : 1.0;              \
    t *= s_; b *= s_; if (WITH_ANS) ans *= s_; }

// exact replay of NS steps with the per-step test, out of line so that the fast loop stays small
template <int NS, bool WA> __device__ __noinline__ void replay_steps(double &a, double &b, double &dj, double tox, double &ans)
{ double t;
  for (int q = 0; q < NS; q++)
    { t = a+dj*tox*b; dj -= 1.0;
      if (fabs(t) > 1.0e10) { if (WA) ans *= 1.0e-10; t *= 1.0e-10; b *= 1.0e-10; }
      a = b; b = t;
    }
}

template <int V> __device__ __forceinline__ double rec(int n, double x, double b0)
{ const double tox = 2.0/fabs(x);
  const int jmax = 2*(n+(int)sqrt(40.0*n));
  double dj = (double)jmax;
  double a = 0.0, b = 1.0, t1, t2, t3, t4, ans = 0.0;
  int c = jmax-n+1;
  for (int phase = 0; phase < 2; phase++)
    { const bool wa = phase == 1;
      if (V == 0)
        { for (; c >= 2; c -= 2)
            { if (wa) { STEP(t1,a,b,true) STEP(t2,b,t1,true) } else { STEP(t1,a,b,false) STEP(t2,b,t1,false) }
              a = t1; b = t2;
            }
        }
      else if (V == 1)                                         // per-step test, unrolled by four
        { for (; c >= 4; c -= 4)
            { if (wa) { STEP(t1,a,b,true) STEP(t2,b,t1,true) STEP(t3,t1,t2,true) STEP(t4,t2,t3,true) }
              else    { STEP(t1,a,b,false) STEP(t2,b,t1,false) STEP(t3,t1,t2,false) STEP(t4,t2,t3,false) }
              a = t3; b = t4;
            }
          if (c >= 2)
            { if (wa) { STEP(t1,a,b,true) STEP(t2,b,t1,true) } else { STEP(t1,a,b,false) STEP(t2,b,t1,false) }
              a = t1; b = t2; c -= 2;
            }
        }
      else if (V == 6)                                         // branch-free, unrolled by four
        { for (; c >= 4; c -= 4)
            { if (wa) { STEPF(t1,a,b,true) STEPF(t2,b,t1,true) STEPF(t3,t1,t2,true) STEPF(t4,t2,t3,true) }
              else    { STEPF(t1,a,b,false) STEPF(t2,b,t1,false) STEPF(t3,t1,t2,false) STEPF(t4,t2,t3,false) }
              a = t3; b = t4;
            }
          if (c >= 2)
            { if (wa) { STEPF(t1,a,b,true) STEPF(t2,b,t1,true) } else { STEPF(t1,a,b,false) STEPF(t2,b,t1,false) }
              a = t1; b = t2; c -= 2;
            }
        }
      else if (V == 7 || V == 8)                               // high-word test first, unrolled by 4 / 8
        { if (V == 8)
            for (; c >= 8; c -= 8)
              { double t5, t6, t7, t8;
                if (wa) { STEPI(t1,a,b,true) STEPI(t2,b,t1,true) STEPI(t3,t1,t2,true) STEPI(t4,t2,t3,true) STEPI(t5,t3,t4,true) STEPI(t6,t4,t5,true) STEPI(t7,t5,t6,true) STEPI(t8,t6,t7,true) }
                else    { STEPI(t1,a,b,false) STEPI(t2,b,t1,false) STEPI(t3,t1,t2,false) STEPI(t4,t2,t3,false) STEPI(t5,t3,t4,false) STEPI(t6,t4,t5,false) STEPI(t7,t5,t6,false) STEPI(t8,t6,t7,false) }
                a = t7; b = t8;
              }
          for (; c >= 4; c -= 4)
            { if (wa) { STEPI(t1,a,b,true) STEPI(t2,b,t1,true) STEPI(t3,t1,t2,true) STEPI(t4,t2,t3,true) }
              else    { STEPI(t1,a,b,false) STEPI(t2,b,t1,false) STEPI(t3,t1,t2,false) STEPI(t4,t2,t3,false) }
              a = t3; b = t4;
            }
          if (c >= 2)
            { if (wa) { STEPI(t1,a,b,true) STEPI(t2,b,t1,true) } else { STEPI(t1,a,b,false) STEPI(t2,b,t1,false) }
              a = t1; b = t2; c -= 2;
            }
        }
      else if (V == 9)                                         // per-step test, unrolled by eight
        { for (; c >= 8; c -= 8)
            { double t5, t6, t7, t8;
              if (wa) { STEP(t1,a,b,true) STEP(t2,b,t1,true) STEP(t3,t1,t2,true) STEP(t4,t2,t3,true) STEP(t5,t3,t4,true) STEP(t6,t4,t5,true) STEP(t7,t5,t6,true) STEP(t8,t6,t7,true) }
              else    { STEP(t1,a,b,false) STEP(t2,b,t1,false) STEP(t3,t1,t2,false) STEP(t4,t2,t3,false) STEP(t5,t3,t4,false) STEP(t6,t4,t5,false) STEP(t7,t5,t6,false) STEP(t8,t6,t7,false) }
              a = t7; b = t8;
            }
          for (; c >= 2; c -= 2)
            { if (wa) { STEP(t1,a,b,true) STEP(t2,b,t1,true) } else { STEP(t1,a,b,false) STEP(t2,b,t1,false) }
              a = t1; b = t2;
            }
        }
      else if (V == 10 || V == 11)                             // one test per 8 / 4 steps, replay out of line
        { constexpr int NS = (V == 10) ? 8 : 4;
          for (; c >= NS; c -= NS)
            { double p = a, q = b, m = 0.;
#pragma unroll
              for (int u = 0; u < NS; u++)
                { const double t = p+(dj-(double)u)*tox*q;
                  m = fmax(m,fabs(t));
                  p = q; q = t;
                }
              if (__any(m > 1.0e10))
                { if (m > 1.0e10)
                    { if (wa) replay_steps<NS,true>(a,b,dj,tox,ans); else replay_steps<NS,false>(a,b,dj,tox,ans); }
                  else { a = p; b = q; dj -= (double)NS; }
                }
              else { a = p; b = q; dj -= (double)NS; }
            }
          for (; c >= 2; c -= 2)
            { if (wa) { STEPF(t1,a,b,true) STEPF(t2,b,t1,true) } else { STEPF(t1,a,b,false) STEPF(t2,b,t1,false) }
              a = t1; b = t2;
            }
        }
      else if (V == 2 || V == 3)                               // grouped test over 2 steps (fmax / high words)
        { for (; c >= 2; c -= 2)
            { t1 = a+dj*tox*b;
              t2 = b+(dj-1.0)*tox*t1;
              bool big;
              if (V == 2) big = fmax(fabs(t1),fabs(t2)) > 1.0e10;
              else
                { unsigned h1 = (unsigned)__double2hiint(t1) & 0x7fffffffu, h2 = (unsigned)__double2hiint(t2) & 0x7fffffffu;
                  big = (h1 > h2 ? h1 : h2) >= 0x4202A05Fu;
                }
              if (__any(big))
                { asm volatile("" ::: "memory");
                  if (big)
                    { double sa = a, sb = b, keep = dj;
                      if (wa) { STEP(t1,sa,sb,true) STEP(t2,sb,t1,true) } else { STEP(t1,sa,sb,false) STEP(t2,sb,t1,false) }
                      dj = keep;
                    }
                }
              a = t1; b = t2; dj -= 2.0;
            }
        }
      else if (V == 4)                                         // no test at all: lower bound, NOT exact
        { for (; c >= 2; c -= 2)
            { t1 = a+dj*tox*b; t2 = b+(dj-1.0)*tox*t1; a = t1; b = t2; dj -= 2.0; }
        }
      else if (V == 5)                                         // branch-free: multiply by 1.0 or 1e-10
        { for (; c >= 2; c -= 2)
            { t1 = a+dj*tox*b; dj -= 1.0;
              double s1 = fabs(t1) > 1.0e10 ? 1.0e-10 : 1.0;
              t1 *= s1; b *= s1; if (wa) ans *= s1;
              t2 = b+dj*tox*t1; dj -= 1.0;
              double s2 = fabs(t2) > 1.0e10 ? 1.0e-10 : 1.0;
              t2 *= s2; t1 *= s2; if (wa) ans *= s2;
              a = t1; b = t2;
            }
        }
      if (c == 1)
        { if (wa) { STEP(t1,a,b,true) } else { STEP(t1,a,b,false) }
          a = b; b = t1;
        }
      if (phase == 0) { ans = a; c = n-1; }
    }
  return ans*b0/b;
}

template <int V> __global__ void __launch_bounds__(64)
k(const int *ns, const double *xs, int reps, double *out, long long *cyc)
{ const int lane = threadIdx.x;
  const int n = ns[lane]; const double x = xs[lane];
  double acc = 0.;
  long long t0 = clock64();
  for (int r = 0; r < reps; r++)
    acc += rec<V>(n+(r & 1),x,1.0);
  long long t1 = clock64();
  out[blockIdx.x*64+lane] = acc;
  if (lane == 0) cyc[blockIdx.x] = t1-t0;
}

template <int V> void run(const char *name, const int *dn, const double *dx, int reps, int blocks, double steps_per_call, double *dout, long long *dcyc, std::vector<double> &ref)
{ hipLaunchKernelGGL(k<V>,dim3(blocks),dim3(64),0,0,dn,dx,reps,dout,dcyc);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0,0);
  hipLaunchKernelGGL(k<V>,dim3(blocks),dim3(64),0,0,dn,dx,reps,dout,dcyc);
  hipEventRecord(e1,0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms,e0,e1);
  std::vector<double> o(64); std::vector<long long> c(blocks);
  hipMemcpy(o.data(),dout,64*8,hipMemcpyDeviceToHost);
  hipMemcpy(c.data(),dcyc,blocks*8,hipMemcpyDeviceToHost);
  bool same = true;
  if (ref.empty()) ref = o; else for (int i = 0; i < 64; i++) if (o[i] != ref[i]) same = false;
  double cy = 0; for (auto v : c) cy += v; cy /= blocks;
  printf("  %-34s blocks %5d: %8.3f ms, %7.1f clk64-cycles/step (wave view)%s\n",name,blocks,ms,cy/(reps*steps_per_call),same ? "" : "   [differs from V0]");
}

int main(int argc, char **argv)
{ int nmode = argc > 1 ? atoi(argv[1]) : 0;
  std::vector<int> n(64); std::vector<double> x(64);
  srand(1);
  for (int i = 0; i < 64; i++)
    { n[i] = nmode == 0 ? 20 : 5+rand()%36;                 // uniform order, or orders 5..40 across lanes
      x[i] = 0.05+0.01*(rand()%400);
    }
  int nmax = 0; for (int v : n) if (v > nmax) nmax = v;
  const double steps = 2*(nmax+(int)sqrt(40.0*nmax))+0.5;     // trip count of the slowest lane (r&1 adds ~0.5)
  int *dn; double *dx, *dout; long long *dcyc;
  hipMalloc(&dn,256); hipMalloc(&dx,512); hipMalloc(&dout,8*64*16384); hipMalloc(&dcyc,8*16384);
  hipMemcpy(dn,n.data(),256,hipMemcpyHostToDevice); hipMemcpy(dx,x.data(),512,hipMemcpyHostToDevice);
  const int reps = 200;
  for (int blocks : { 1024, 3072, 8192 })                    // 1, 3, 8 waves per SIMD on 256 CUs x 4 SIMDs
    { std::vector<double> ref;
      printf("orders %s, %d waves per SIMD\n",nmode == 0 ? "20 on every lane" : "5..40 across lanes",blocks/1024);
      run<0>("V0 per-step test, unroll 2",dn,dx,reps,blocks,steps,dout,dcyc,ref);
      run<1>("V1 per-step test, unroll 4",dn,dx,reps,blocks,steps,dout,dcyc,ref);
      run<9>("V9 per-step test, unroll 8",dn,dx,reps,blocks,steps,dout,dcyc,ref);
      run<7>("V7 high-word test, unroll 4",dn,dx,reps,blocks,steps,dout,dcyc,ref);
      run<8>("V8 high-word test, unroll 8",dn,dx,reps,blocks,steps,dout,dcyc,ref);
      run<5>("V5 branch-free, unroll 2",dn,dx,reps,blocks,steps,dout,dcyc,ref);
      run<6>("V6 branch-free, unroll 4",dn,dx,reps,blocks,steps,dout,dcyc,ref);
      run<11>("V11 test per 4, replay out of line",dn,dx,reps,blocks,steps,dout,dcyc,ref);
      run<10>("V10 test per 8, replay out of line",dn,dx,reps,blocks,steps,dout,dcyc,ref);
      run<4>("V4 no test (bound, inexact)",dn,dx,reps,blocks,steps,dout,dcyc,ref);
    }
  return 0;
}
