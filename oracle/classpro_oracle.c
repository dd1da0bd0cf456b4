/*
 * classpro_oracle.c -- TEST INFRASTRUCTURE ONLY (see classpro_oracle.h for the pinning status).
 *
 * Sequential CPU restatement of ClassPro's per-read classification path.  Every function cites
 * the reference file:line it follows (paths relative to /root/reference/src).  Arithmetic is kept
 * in the reference's evaluation order (double, no FMA contraction: build with -ffp-contract=off)
 * so that results are bit-identical to the reference's gcc -O3 x86-64 build.
 *
 * Defined behaviour where the reference reads stale per-thread memory:
 *   - wall[plen] / perror[plen] are reset at read start (reference resets only i < plen, wall.c:581);
 *   - rctx cells that calc_seq_context never writes (only possible for runs longer than the
 *     127 cap) read as 0;
 *   - Intvl.ccb/cce slots that correct_wall_cnt touches by *position* (wall.c:999-1006) start at 0.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <pthread.h>
#include <malloc.h>
#include "classpro_oracle.h"

#define MINI(a,b) ((a) < (b) ? (a) : (b))
#define MAXI(a,b) ((a) > (b) ? (a) : (b))

/* const.c:46-73 */
static const int    N_SIGMA_RCOV   = 5;
static const int    MAX_N_LC       = 20;
static const int    MAX_N_HC       = 5;
static const int    MIN_CNT_CHANGE = 3;
static const int    MAX_CNT_CHANGE = 5;
static const double PE_THRES[2][2] = { {0.001, 0.05}, {1e-5, 1e-5} };
static const double THRES_DIFF_EO  = -23.025851;
static const double THRES_DIFF_REL = -9.210340;
static const int    OFFSET         = 1000;
static const int    N_SIGMA_R      = 2;
static const double R_LOGP         = -10.;
static const double E_PO_BASE      = -10.;
static const double PE_MEAN        = 0.01;
static const char   STOC[4]        = { 'E', 'R', 'H', 'D' };   /* const.c:19 */

/* ------------------------------------------------------------------------------------------
 *  bessel.c:390-521 (modified Bessel I_n; polynomial coefficients are the published
 *  Abramowitz & Stegun 9.8.1-9.8.4 values used by the reference)
 * ------------------------------------------------------------------------------------------ */
static double bessi0(double x)                      /* bessel.c:390-411 */
{ double ax = fabs(x), y, ans;
  if (ax < 3.75)
    { y = x/3.75; y = y*y;
      ans = 1.0+y*(3.5156229+y*(3.0899424+y*(1.2067492
            +y*(0.2659732+y*(0.360768e-1+y*0.45813e-2)))));
    }
  else
    { y = 3.75/ax;
      ans = (exp(ax)/sqrt(ax))*(0.39894228+y*(0.1328592e-1
            +y*(0.225319e-2+y*(-0.157565e-2+y*(0.916281e-2
            +y*(-0.2057706e-1+y*(0.2635537e-1+y*(-0.1647633e-1
            +y*0.392377e-2))))))));
    }
  return ans;
}

static double bessi1(double x)                      /* bessel.c:416-438 */
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
      ans *= (exp(ax)/sqrt(ax));
    }
  return x < 0.0 ? -ans : ans;
}

double cpo_bessi(int n, double x)                   /* bessel.c:478-521 */
{ if (n < 0)
    { fprintf(stderr,"n<0 @ bessi\n"); exit(1); }
  if (n == 0) return bessi0(x);
  if (n == 1) return bessi1(x);
  if (x == 0.0) return 0.0;
  double tox = 2.0/fabs(x), bip = 0.0, ans = 0.0, bi = 1.0, bim;
  for (int j = 2*(n+(int)sqrt(40.0*n)); j > 0; j--)   /* ACC = 40.0, bessel.c:78 */
    { bim = bip+j*tox*bi;
      bip = bi;
      bi  = bim;
      if (fabs(bi) > 1.0e10)                        /* BIGNO / BIGNI, bessel.c:79-80 */
        { ans *= 1.0e-10; bi *= 1.0e-10; bip *= 1.0e-10; }
      if (j == n) ans = bip;
    }
  ans *= bessi0(x)/bi;
  return (x < 0.0 && n%2 == 1) ? -ans : ans;
}

/* ------------------------------------------------------------------------------------------
 *  prob.c / util.c
 * ------------------------------------------------------------------------------------------ */
static inline int check_cnt(int n)                  /* prob.c:22-30 (DEBUG is ON, ClassPro.h:17) */
{ n &= 0xffff;                                      /* arguments are cnt_t (uint16) */
  if (n > CPO_MAX_KMER_CNT)
    { fprintf(stderr,"K-mer count (%d) > MAX_KMER_CNT (%d) (due to D/R ratio?)\n",n,CPO_MAX_KMER_CNT);
      return CPO_MAX_KMER_CNT;
    }
  return n;
}

static inline double logp_poisson(const cpo_params *p, int k, int lambda)   /* prob.c:33-39 */
{ k = check_cnt(k);
  return k * log((double)lambda) - lambda - p->logfact[k];
}

static inline double logp_skellam(int k, double lambda)                     /* prob.c:41-44 */
{ return -2. * lambda + log(cpo_bessi(abs(k),2.*lambda)); }

static inline void check_cnt_binom(int *k, int *n)                          /* prob.c:47-57 */
{ *k = check_cnt(*k);
  *n = check_cnt(*n);
  if (*k > *n)
    { fprintf(stderr,"k (%d) > n (%d) in Binom\n",*k,*n); exit(1); }
}

static inline double logp_binom(const cpo_params *p, int k, int n, double pr)   /* prob.c:59-65 */
{ check_cnt_binom(&k,&n);
  const double *lf = p->logfact;
  return lf[n] - lf[k] - lf[n-k] + k * log(pr) + (n-k) * log(1-pr);
}

static inline double logp_binom_pre(const cpo_params *p, int k, int n, double lpe, double l1mpe) /* prob.c:67-73 */
{ check_cnt_binom(&k,&n);
  const double *lf = p->logfact;
  return lf[n] - lf[k] - lf[n-k] + k * lpe + (n-k) * l1mpe;
}

static double binom_test_g(const cpo_params *p, int k, int n, double pe, int exact)   /* prob.c:76-112 */
{ check_cnt_binom(&k,&n);
  const double lpe   = log(pe);
  const double l1mpe = log(1-pe);
  const double mean  = n * pe;
  const int decrease = ((double)k >= mean);
  double s, p_first, p_curr;
  if (decrease)
    { s = p_first = exp(logp_binom_pre(p,k,n,lpe,l1mpe));
      for (int x = k+1; x <= n; x++)
        { s += p_curr = exp(logp_binom_pre(p,x,n,lpe,l1mpe));
          if (!exact && 10 * p_curr < p_first)
            break;
        }
    }
  else
    { s = p_first = (k == 0) ? 0. : exp(logp_binom_pre(p,k-1,n,lpe,l1mpe));
      for (int x = k-2; x >= 0; x--)
        { s += p_curr = exp(logp_binom_pre(p,x,n,lpe,l1mpe));
          if (!exact && 10 * p_curr < p_first)
            break;
        }
      s = 1-s;
    }
  return s;
}

static inline int plus_sigma(int cnt, int n_sigma)                          /* util.c:9-11 */
{ return (cnt + (int)(uint16_t)(sqrt(cnt) * n_sigma)) & 0xffff; }

typedef struct { int pos; int cnt; } pos_cnt;                               /* ClassPro.h:201-204 */

static inline double linear_interpolation(int x, pos_cnt pc1, pos_cnt pc2)  /* util.c:24-33 */
{ if (!(pc1.pos < x && x < pc2.pos))
    { fprintf(stderr,"Invalid points for interpolation: x1=%d, x=%d, x2=%d\n",pc1.pos,x,pc2.pos);
      exit(1);
    }
  return (double)pc1.cnt+((double)pc2.cnt-pc1.cnt)*(x-pc1.pos)/(pc2.pos-pc1.pos);
}

static inline double logp_trans(const cpo_params *p, int b, int e, int cb, int ce, int cov)   /* util.c:35-44 */
{ cov &= 0xffff;                                    /* cnt_t parameter */
  return logp_skellam(ce-cb,(double)cov*abs(e-b)/p->read_len);
}

static inline double p_errorin(const cpo_params *p, int e, double erate, int cout, int cin)   /* util.c:46-55 */
{ if (!(cin <= cout))
    { fprintf(stderr,"Violate cin (%d) <= cout (%d)\n",cin,cout); exit(1); }
  return binom_test_g(p,(e == CPO_SELF) ? cin : cout-cin,cout,erate,0);
}

double cpo_logp_poisson(const cpo_params *p, int k, int lambda) { return logp_poisson(p,k,lambda); }
double cpo_logp_skellam(int k, double lambda) { return logp_skellam(k,lambda); }
double cpo_logp_binom(const cpo_params *p, int k, int n, double pr) { return logp_binom(p,k,n,pr); }
double cpo_binom_test_g(const cpo_params *p, int k, int n, double pe, int exact) { return binom_test_g(p,k,n,pe,exact); }
double cpo_logp_trans(const cpo_params *p, int b, int e, int cb, int ce, int cov) { return logp_trans(p,b,e,cb,ce,cov); }

/* ------------------------------------------------------------------------------------------
 *  Global setup: ClassPro.c:536-554, prob.c:14-19, wall.c:117-244
 * ------------------------------------------------------------------------------------------ */
/* load_himodel (wall.c:55-115): HIsim error model -> pe[t][l] by a quadratic fit of the mean error rate of
 * 2..5 unit copies (plus the fixed point (1, 0.002)).  The reference fits with GSL's gsl_multifit_linear
 * (wall.c:11-41), which is not in this image: the fit below solves the same least-squares problem with a
 * Householder QR in double -- PARITY UNPINNED for this function (no GSL, no model file in the reference
 * tree); the two agree to rounding.  Returns 0, or -1 when the file cannot be read. */
typedef struct { float all, ins, op[9]; } hi_erates;     /* wall.c:43-47 */
typedef struct { float all, op[6]; } hi_mrates;          /* wall.c:49-52 */

static void quad_fit5(const double *x, const double *y, double *coef)
{ double A[5][3], b[5];
  for (int i = 0; i < 5; i++)
    { A[i][0] = 1.; A[i][1] = x[i]; A[i][2] = x[i]*x[i]; b[i] = y[i]; }
  for (int k = 0; k < 3; k++)                            /* Householder reflections, column by column */
    { double nrm = 0.;
      for (int i = k; i < 5; i++) nrm += A[i][k]*A[i][k];
      nrm = sqrt(nrm);
      double alpha = (A[k][k] > 0) ? -nrm : nrm;
      double v[5] = {0,0,0,0,0};
      for (int i = k; i < 5; i++) v[i] = A[i][k];
      v[k] -= alpha;
      double vv = 0.;
      for (int i = k; i < 5; i++) vv += v[i]*v[i];
      if (vv == 0.) continue;
      for (int j = k; j < 3; j++)
        { double d = 0.;
          for (int i = k; i < 5; i++) d += v[i]*A[i][j];
          d = 2*d/vv;
          for (int i = k; i < 5; i++) A[i][j] -= d*v[i];
        }
      double d = 0.;
      for (int i = k; i < 5; i++) d += v[i]*b[i];
      d = 2*d/vv;
      for (int i = k; i < 5; i++) b[i] -= d*v[i];
    }
  for (int k = 2; k >= 0; k--)
    { double t = b[k];
      for (int j = k+1; j < 3; j++) t -= A[k][j]*coef[j];
      coef[k] = t/A[k][k];
    }
}

int cpo_load_himodel(const char *path, double *pe63)
{ FILE *f = fopen(path,"rb");
  if (f == NULL) return -1;
  int kmer;
  if (fread(&kmer,sizeof(int),1,f) != 1) { fclose(f); return -1; }
  const int krange = kmer/2-6;
  if (krange < 1 || krange > 1000 || fseek(f,(long)sizeof(hi_erates)*0x4000,SEEK_CUR) != 0) { fclose(f); return -1; }
  double x[5] = {1,2,3,4,5}, y[5], coef[3];
  y[0] = 0.002;
  for (int t = 0; t < 3; t++)
    { int ulen = t+1, N = 1 << (2*ulen);
      hi_mrates *m = malloc(sizeof(hi_mrates)*N*krange);
      if (fread(m,sizeof(hi_mrates),(size_t)N*krange,f) != (size_t)N*krange) { free(m); fclose(f); return -1; }
      for (int j = 2; j <= 5; j++)                       /* wall.c:87-99; mics[t] is biased by -2*ulen */
        { double sum = 0.; int n = 0;
          for (int i = 0; i < N; i++)
            { double p = m[krange*i+(j-2)*ulen].all;
              if (p > 0.) { sum += p; n++; }
            }
          y[j-1] = sum/n;
        }
      free(m);
      quad_fit5(x,y,coef);
      int lmax = MAX_N_LC/(t+1);
      pe63[t*21] = 0.;
      for (int l = 1; l <= lmax; l++)
        pe63[t*21+l] = coef[0]+coef[1]*l+coef[2]*l*l;
    }
  fclose(f);
  return 0;
}

static const double *g_pe_override = NULL;               /* set only inside cpo_params_new_model */

cpo_params *cpo_params_new_model(int K, int read_len, int hcov, int dcov, const double *pe63)
{ g_pe_override = pe63;
  cpo_params *p = cpo_params_new(K,read_len,hcov,dcov);
  g_pe_override = NULL;
  return p;
}

cpo_params *cpo_params_new(int K, int read_len, int hcov, int dcov)
{ cpo_params *p = calloc(1,sizeof(cpo_params));
  p->K = K;
  p->read_len = read_len;

  p->logfact[0] = 0.;                               /* prob.c:14-19 */
  for (int n = 1; n <= CPO_MAX_KMER_CNT; n++)
    p->logfact[n] = p->logfact[n-1]+log(n);

  p->cov[CPO_HAPLO]  = hcov & 0xffff;               /* ClassPro.c:544-548 */
  p->cov[CPO_DIPLO]  = dcov & 0xffff;
  p->cov[CPO_ERROR]  = 1;
  p->cov[CPO_REPEAT] = plus_sigma(p->cov[CPO_DIPLO],N_SIGMA_RCOV);
  p->dr_ratio = 1.+(double)N_SIGMA_R*(1./sqrt(p->cov[CPO_DIPLO]));

  if (p->cov[CPO_REPEAT] > 255)                     /* wall.c:174-177 */
    { fprintf(stderr,"Too high REPEAT coverage (%d) > 255\n",p->cov[CPO_REPEAT]);
      free(p);
      return NULL;
    }
  p->cmax = p->cov[CPO_REPEAT];                     /* wall.c:178 */

  for (int t = 0; t < 3; t++)                       /* wall.c:119-143 (default model) */
    { p->lmax[t] = (uint8_t)(MAX_N_LC/(t+1));
      p->pe[t][0] = 0.;
      for (int l = 1; l <= p->lmax[t]; l++)
        p->pe[t][l] = g_pe_override ? g_pe_override[t*21+l] : 0.002 * l * l + 0.002;
    }
  p->hc_erate = p->pe[CPO_HP][1];                   /* wall.c:180 */

  for (int t = 0; t < 3; t++)                       /* wall.c:190-224 */
    for (int l = 1; l <= p->lmax[t]; l++)
      { double pe = p->pe[t][l];
        double lpe = log(pe);
        double l1mpe = log(1-pe);
        for (int cout = 1; cout < p->cmax; cout++)
          { uint8_t ct[2];
            int found[2][2];
            ct[CPO_SELF] = (uint8_t)cout;
            ct[CPO_OTHERS] = 0;
            for (int s = 0; s < 2; s++)
              for (int e = 0; e < 2; e++)
                { p->cthres[t][l][cout][s][e] = ct[e];
                  found[s][e] = 0;
                }
            double psum = 1.;
            for (int cin = 0; cin <= cout; cin++)
              { if (found[0][0] && found[1][0] && found[0][1] && found[1][1])
                  break;
                ct[CPO_SELF] = (uint8_t)cin;
                ct[CPO_OTHERS] = (uint8_t)(cout-cin);
                psum -= exp(logp_binom_pre(p,cin,cout,lpe,l1mpe));
                for (int s = 0; s < 2; s++)
                  for (int e = 0; e < 2; e++)
                    if (!found[s][e] && psum < PE_THRES[s][e])
                      { p->cthres[t][l][cout][s][e] = ct[e];
                        found[s][e] = 1;
                      }
              }
          }
      }
  return p;
}

void cpo_params_free(cpo_params *p) { free(p); }
const uint8_t *cpo_params_cthres(const cpo_params *p) { return &p->cthres[0][0][0][0][0]; }
const double  *cpo_params_logfact(const cpo_params *p) { return p->logfact; }
const double  *cpo_params_pe(const cpo_params *p) { return &p->pe[0][0]; }
const int     *cpo_params_lmax(const cpo_params *p) { return &p->lmax[0]; }
void cpo_params_scalars(const cpo_params *p, int *cov4, double *dr_ratio, int *cmax, double *hc_erate)
{ for (int i = 0; i < 4; i++) cov4[i] = p->cov[i];
  *dr_ratio = p->dr_ratio; *cmax = p->cmax; *hc_erate = p->hc_erate;
}

/* ------------------------------------------------------------------------------------------
 *  hist.c:28-105 on top of libfastk.c:22-47 (toggle), 51-102 (load), 106-147 (modify)
 * ------------------------------------------------------------------------------------------ */
int cpo_hist_covs(const int64_t *disk, int low, int high, int64_t ilowcnt, int64_t ihighcnt,
                  int coverage_opt, int *hcov, int *dcov)
{ if (coverage_opt > 0)                             /* hist.c:44-50 */
    { *dcov = coverage_opt;
      *hcov = coverage_opt >> 1;
      return 0;
    }
  if (low > 1)
    return 2;                                       /* reference would index hist[low-1]: undefined */

  /* Load_Histogram: hist[low..high] from disk, hist[high+1]=ilowcnt, hist[high+2]=ihighcnt.
     Modify_Histogram(H,low,high,0) keeps the range and toggles to instance counts:
     interior cells *= i, boundary cells swapped with the hidden ones. */
  int64_t *buf  = malloc(sizeof(int64_t)*((high-low)+3));
  int64_t *hist = buf-low;
  memcpy(buf,disk,sizeof(int64_t)*((high-low)+1));
  hist[high+1] = ilowcnt;
  hist[high+2] = ihighcnt;
  for (int i = low+1; i < high; i++)                /* libfastk.c:30-33 */
    hist[i] *= i;
  { int64_t x = hist[high+1]; hist[high+1] = hist[low];  hist[low]  = x;   /* libfastk.c:41-47 */
            x = hist[high+2]; hist[high+2] = hist[high]; hist[high] = x;
  }

  int     maxcnt = 0;                               /* hist.c:58-73 */
  int64_t maxpk  = 0;
  for (int i = MAXI(2,low); i < MINI(1000,high); i++)
    if (hist[i-1] < hist[i] && hist[i] > hist[i+1] && maxpk < hist[i])
      { maxcnt = i;
        maxpk  = hist[i];
      }
  if (maxcnt < 10)
    { free(buf);
      return 1;
    }

  int     lmaxcnt = 0, rmaxcnt = 0, is_lpeak = 0, is_rpeak = 0;
  int64_t lmaxpk = 0, rmaxpk = 0;
  double  m, s;

  m = (double)maxcnt/2;                             /* hist.c:75-85 */
  s = sqrt(m);
  for (int i = (int)round(m-s); i <= (int)round(m+s); i++)
    if (lmaxpk < hist[i])
      { lmaxcnt = i;
        lmaxpk = hist[i];
        is_lpeak = (hist[i-1] < hist[i] && hist[i] > hist[i+1]) ? 1 : 0;
      }

  m = (double)maxcnt*2;                             /* hist.c:87-97 */
  s = sqrt(m);
  for (int i = (int)round(m-s); i <= (int)round(m+s); i++)
    if (rmaxpk < hist[i])
      { rmaxcnt = i;
        rmaxpk = hist[i];
        is_rpeak = (hist[i-1] < hist[i] && hist[i] > hist[i+1]) ? 1 : 0;
      }

  if (lmaxpk > rmaxpk)                              /* hist.c:99-107 */
    { *dcov = maxcnt;
      *hcov = is_lpeak ? lmaxcnt : (maxcnt >> 1);
    }
  else
    { *hcov = maxcnt;
      *dcov = is_rpeak ? rmaxcnt : (maxcnt << 1);
    }
  free(buf);
  return 0;
}

/* ------------------------------------------------------------------------------------------
 *  libfastk.c:1467-1534 (Fetch_Profile's decoder, buffer refills removed)
 * ------------------------------------------------------------------------------------------ */
int cpo_decode_profile(const uint8_t *code, int64_t len, uint16_t *profile, int cap)
{ if (len == 0)
    return 0;
  const uint8_t *c = code, *q = code+len;
  uint16_t x, d;
  int n;

  x = *c++;
  if ((x & 0x80) != 0)
    d = ((x & 0x7f) << 8) | *c++;
  else
    d = x;
  n = 1;
  if (cap > 0)
    { profile[0] = d;
      while (c < q)
        { x = *c++;
          if ((x & 0xc0) == 0)                      /* 00rrrrrr: run of current count */
            { if (n+x > cap)
                { n += x;
                  break;
                }
              for (int i = 0; i < x; i++)
                profile[n++] = d;
            }
          else
            { if ((x & 0x80) != 0)                  /* 1sxxxxxx yyyyyyyy: 15-bit delta */
                { if ((x & 0x40) != 0)
                    x <<= 8;
                  else
                    x = (x << 8) & 0x7fff;
                  x |= *c++;
                  d = (d+x) & 0x7fff;
                }
              else                                  /* 01sxxxxx: 6-bit signed delta */
                { if ((x & 0x20) != 0)
                    d += (x & 0x1fu) | 0xffe0u;
                  else
                    d += (x & 0x1fu);
                }
              if (n >= cap)
                { n += 1;
                  break;
                }
              profile[n++] = d;
            }
        }
    }
  while (c < q)                                     /* count what did not fit */
    { x = *c++;
      if ((x & 0xc0) == 0)
        n += x;
      else
        { if ((x & 0x80) != 0)
            c += 1;
          n += 1;
        }
    }
  return n;
}

/* ------------------------------------------------------------------------------------------
 *  context.c:8-108.  L(i,t)/R(i,t): [pos][ctype] uint8; presets from ClassPro.c:139-140.
 * ------------------------------------------------------------------------------------------ */
void cpo_seq_context(const char *seq, int rlen, uint8_t *lctx, uint8_t *rctx)
{
#define L(i,t) lctx[(size_t)(i)*3+(t)]
#define R(i,t) rctx[(size_t)(i)*3+(t)]
  int in_hp, in_ds, in_ts;
  const int rlenm1 = rlen-1;

  memset(lctx,0,(size_t)rlen*3);
  memset(rctx,0,(size_t)rlen*3);
  L(0,CPO_HP) = 1;                                  /* ClassPro.c:139 */
  L(0,CPO_DS) = L(0,CPO_TS) = 0;                    /* ClassPro.c:140 */
  if (rlen > 1) L(1,CPO_TS) = 0;

  in_ds = in_ts = 0;
  for (int i = 1; i < rlen; i++)
    { in_hp = (seq[i-1] == seq[i]) ? 1 : 0;
      in_ds = in_ts = 0;

      if (in_hp)                                    /* context.c:17-20 */
        { L(i,CPO_HP) = MINI(L(i-1,CPO_HP)+1,127);
          L(i,CPO_DS) = R(i-1,CPO_DS) = 0;
        }
      else                                          /* context.c:21-31 */
        { L(i,CPO_HP) = 1;
          L(i,CPO_DS) = R(i-1,CPO_DS) = 1;
          for (int j = i-L(i-1,CPO_HP), n = 0; j < i; j++, n++)
            R(j,CPO_HP) = L(i-1-n,CPO_HP);
          if (i >= 3 && seq[i-3] == seq[i-1] && seq[i-2] == seq[i])
            { L(i,CPO_DS) = MINI(L(i-2,CPO_DS)+1,127);
              in_ds = 1;
            }
        }

      if (!in_ds)                                   /* context.c:33-40 */
        { int l = i-1;
          while (L(l,CPO_DS) > 1)
            l--;
          if (l < i-1)
            for (int j = l-1, n = 0; j < i; j++, n++)
              R(j-1,CPO_DS) = L(i-1-n,CPO_DS);
        }

      if (i >= 2)                                   /* context.c:42-60 */
        { if (in_hp && seq[i-2] == seq[i-1])
            L(i,CPO_TS) = R(i-2,CPO_TS) = 0;
          else if (i >= 5 && seq[i-5] == seq[i-2] && seq[i-4] == seq[i-1] && seq[i-3] == seq[i])
            { L(i,CPO_TS) = MINI(L(i-3,CPO_TS)+1,127);
              in_ts = 1;
            }
          else
            L(i,CPO_TS) = R(i-1,CPO_TS) = R(i-2,CPO_TS) = 1;

          if (!in_ts)
            { int l = i-1;
              while (L(l,CPO_TS) > 1)
                l--;
              if (l < i-1)
                for (int j = l-2, n = 0; j < i; j++, n++)
                  R(j-2,CPO_TS) = L(i-1-n,CPO_TS);
            }
        }
    }

  for (int j = rlen-L(rlenm1,CPO_HP), n = 0; j < rlen; j++, n++)   /* context.c:63-64 */
    R(j,CPO_HP) = L(rlenm1-n,CPO_HP);

  if (in_ds)                                        /* context.c:66-73 */
    { int l = rlenm1;
      while (L(l,CPO_DS) > 1)
        l--;
      if (l < rlenm1)
        for (int j = l-1, n = 0; j < rlen; j++, n++)
          R(j-1,CPO_DS) = L(rlenm1-n,CPO_DS);
    }

  if (in_ts)                                        /* context.c:75-82 */
    { int l = rlenm1;
      while (L(l,CPO_TS) > 1)
        l--;
      if (l < rlenm1)
        for (int j = l-2, n = 0; j < rlen; j++, n++)
          R(j-2,CPO_TS) = L(rlenm1-n,CPO_TS);
    }

  R(rlenm1,CPO_DS) = R(rlenm1,CPO_TS) = 0;          /* context.c:84 */
  if (rlen >= 2) R(rlen-2,CPO_TS) = 0;
#undef L
#undef R
}

/* ------------------------------------------------------------------------------------------
 *  wall.c:264-568 helpers
 * ------------------------------------------------------------------------------------------ */
typedef struct { int b, e; double pe; } eintvl_t;    /* ClassPro.h:153-157 */

static const uint8_t MASK_WALL_BY[2]   = { 0x01, 0x10 };   /* wall.c:264-269 */
static const uint8_t MASK_PAIRED_BY[2] = { 0x02, 0x20 };
static const uint8_t MASK_PAIRED_MULT  = 0x40;
static const uint8_t MASK_ERROR        = 0x80;

typedef struct
  { const cpo_params *p;
    const uint16_t   *profile;
    int               plen;
    const uint8_t    *ctx[2];      /* ctx[DROP] = _lctx + (K-2), ctx[GAIN] = rctx (ClassPro.c:138-142) */
    uint8_t          *wall;        /* [plen+1] */
    double           *perror;      /* [plen+1][2][2] */
  } wall_ctx;

#define CTX(W,w,i,t)  ((W)->ctx[w][(size_t)(i)*3+(t)])
#define PERR(W,i,e,w) ((W)->perror[(size_t)(i)*4+(e)*2+(w)])

static inline void update_perror(wall_ctx *W, int i, int e, int w, int cout, int cin, double erate)  /* wall.c:310-315 */
{ if (PERR(W,i,e,w) == -INFINITY)
    PERR(W,i,e,w) = p_errorin(W->p,e,erate,cout,cin);
}

static inline double logp_diff_pair(const wall_ctx *W, int i, int j)        /* wall.c:317-322 */
{ const uint16_t *pr = W->profile;
  int n_drop = (int)pr[i-1]-pr[i];
  int n_gain = (int)pr[j]-pr[j-1];
  int cov    = MAXI(pr[i-1],pr[j]);
  return logp_trans(W->p,i,j,n_drop,n_gain,cov);
}

static inline int cthres_ng(int e, int cin, int ct)                         /* wall.c:324-329 */
{ cin &= 0xff;                                      /* uint8 parameter */
  return (e == CPO_SELF) ? (cin >= ct) : (cin < ct);
}

/* wall.c:331-416: partner GAIN for a DROP at i */
static int find_gain(wall_ctx *W, int i, int cout, int cin, int e, int t, int l, double erate, eintvl_t *out)
{ const cpo_params *p = W->p;
  const uint16_t *pr = W->profile;
  const int plen = W->plen, K = p->K, CMAX = p->cmax;
  const int ipk = i+K-1, ulen = t+1;
  int m, n, j, max_j = -1, cout_j, cin_j;
  double pe, max_pe = -INFINITY;

  m = ulen*l;                                       /* low-complexity error */
  n = 0;
  while (1)
    { int idx = i+ulen*(n+1);
      if (idx >= plen || CTX(W,CPO_DROP,idx,t) != m+n+1)
        break;
      n++;
    }
  j = ipk+n-m;
  if (j <= i)
    return 0;
  if (j >= plen)
    { j = plen;
      pe = PERR(W,i,e,CPO_DROP) * PERR(W,i,e,CPO_DROP);
    }
  else
    { cin_j  = pr[j-1];
      cout_j = pr[j];
      pe = -INFINITY;
      if (cin_j <= cout_j
          && !(cout_j < CMAX && cthres_ng(e,cin_j,p->cthres[t][l][cout_j][CPO_FINAL][e]))
          && (e == CPO_SELF || logp_diff_pair(W,i,j) >= THRES_DIFF_EO))
        { update_perror(W,j,e,CPO_GAIN,cout_j,cin_j,erate);
          pe = PERR(W,i,e,CPO_DROP)*PERR(W,j,e,CPO_GAIN);
        }
    }
  if (max_pe < pe)
    { max_j  = j;
      max_pe = pe;
    }

  m = 0;                                            /* high-complexity errors */
  for (n = 0; n <= MAX_N_HC; n++)
    { j = ipk+n-m;
      if (j >= plen)
        break;
      cin_j  = pr[j-1];
      cout_j = pr[j];
      if (!(cin_j <= cout_j))
        continue;
      if ((cout < CMAX && cthres_ng(e,cin,p->cthres[CPO_HP][1][cout][CPO_FINAL][e]))
          || (cout_j < CMAX && cthres_ng(e,cin_j,p->cthres[CPO_HP][1][cout_j][CPO_FINAL][e])))
        continue;
      if (e == CPO_OTHERS && logp_diff_pair(W,i,j) < THRES_DIFF_EO)
        continue;
      double pe_i = p_errorin(p,e,p->hc_erate,cout,cin);
      double pe_j = p_errorin(p,e,p->hc_erate,cout_j,cin_j);
      pe = pe_i * pe_j;
      if (max_pe < pe)
        { max_j  = j;
          max_pe = pe;
        }
    }

  if (max_j == -1)
    return 0;
  out->b  = i;
  out->e  = max_j;
  out->pe = max_pe;
  return 1;
}

/* wall.c:418-507: partner DROP for a GAIN at i */
static int find_drop(wall_ctx *W, int i, int cout, int cin, int e, int t, int l, double erate, eintvl_t *out)
{ const cpo_params *p = W->p;
  const uint16_t *pr = W->profile;
  const int K = p->K, CMAX = p->cmax;
  const int imk = i-K+1, ulen = t+1;
  int m, n, j, max_j = -1, cout_j, cin_j;
  double pe, max_pe = -INFINITY;

  m = ulen*l;
  n = 0;
  while (1)
    { int idx = i-ulen*(n+1);
      if (idx <= 0)
        break;
      if (CTX(W,CPO_GAIN,idx,t) != m+n+1)
        break;
      n++;
    }
  j = imk-n+m;
  if (j >= i)
    return 0;
  if (j <= 0)
    { j = 0;
      pe = PERR(W,i,e,CPO_GAIN) * PERR(W,i,e,CPO_GAIN);
    }
  else
    { cout_j = pr[j-1];
      cin_j  = pr[j];
      pe = -INFINITY;
      if (cin_j <= cout_j
          && !(cout_j < CMAX && cthres_ng(e,cin_j,p->cthres[t][l][cout_j][CPO_FINAL][e]))
          && (e == CPO_SELF || logp_diff_pair(W,j,i) >= THRES_DIFF_EO))
        { update_perror(W,j,e,CPO_DROP,cout_j,cin_j,erate);
          pe = PERR(W,j,e,CPO_DROP)*PERR(W,i,e,CPO_GAIN);
        }
    }
  if (max_pe < pe)
    { max_j  = j;
      max_pe = pe;
    }

  m = 0;
  for (n = 0; n <= MAX_N_HC; n++)
    { j = imk-n+m;
      if (j <= 0)
        break;
      cout_j = pr[j-1];
      cin_j  = pr[j];
      if (!(cin_j <= cout_j))
        continue;
      if ((cout < CMAX && cthres_ng(e,cin,p->cthres[CPO_HP][1][cout][CPO_FINAL][e]))
          || (cout_j < CMAX && cthres_ng(e,cin_j,p->cthres[CPO_HP][1][cout_j][CPO_FINAL][e])))
        continue;
      if (e == CPO_OTHERS && logp_diff_pair(W,j,i) < THRES_DIFF_EO)
        continue;
      double pe_i = p_errorin(p,e,p->hc_erate,cout,cin);
      double pe_j = p_errorin(p,e,p->hc_erate,cout_j,cin_j);
      pe = pe_i * pe_j;
      if (max_pe < pe)
        { max_j  = j;
          max_pe = pe;
        }
    }

  if (max_j == -1)
    return 0;
  out->b  = max_j;
  out->e  = i;
  out->pe = max_pe;
  return 1;
}

/* wall.c:519-528 compares (b, e, then (int)(pe_b - pe_a) which is 0 for probabilities); glibc's
 * qsort is a stable merge sort at these sizes, so: stable sort by (b,e). */
static inline int eintvl_less(const eintvl_t *a, const eintvl_t *b)
{ if (a->b != b->b) return a->b < b->b;
  if (a->e != b->e) return a->e < b->e;
  return ((int)(b->pe - a->pe)) < 0;
}

static void sort_eintvl(eintvl_t *v, int n, eintvl_t *tmp)
{ if (n < 2) return;
  for (int w = 1; w < n; w *= 2)                    /* bottom-up stable merge sort */
    { for (int lo = 0; lo < n; lo += 2*w)
        { int mid = MINI(lo+w,n), hi = MINI(lo+2*w,n);
          int a = lo, b = mid, k = lo;
          while (a < mid && b < hi)
            { if (eintvl_less(&v[b],&v[a])) tmp[k++] = v[b++];
              else                          tmp[k++] = v[a++];
            }
          while (a < mid) tmp[k++] = v[a++];
          while (b < hi)  tmp[k++] = v[b++];
        }
      memcpy(v,tmp,sizeof(eintvl_t)*n);
    }
}

static int bs_eintvl(const eintvl_t *v, int l, int r, int b, int e)         /* wall.c:530-546 */
{ while (l <= r)
    { int m = (l+r)/2;
      if (v[m].b == b)
        { if (v[m].e == e) return m;
          else if (e > v[m].e) l = m+1;
          else r = m-1;
        }
      else if (b > v[m].b) l = m+1;
      else r = m-1;
    }
  return -1;
}

static int remove_duplicates(eintvl_t *v, int N, eintvl_t *tmp)             /* wall.c:548-568 */
{ sort_eintvl(v,N,tmp);
  if (N >= 2)
    { int i = 1;
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
      N = i;
    }
  return N;
}

/* ------------------------------------------------------------------------------------------
 *  find_wall, wall.c:570-958
 * ------------------------------------------------------------------------------------------ */
int cpo_find_wall(const cpo_params *p, const uint16_t *profile, int plen,
                  const uint8_t *lctx, const uint8_t *rctx, cpo_intvl *intvl, int cap)
{ const int K = p->K, CMAX = p->cmax;
  const int REP = p->cov[CPO_REPEAT], HAP = p->cov[CPO_HAPLO];
  wall_ctx Wc, *W = &Wc;
  int ecap = plen+2;
  uint8_t  *wall   = calloc(plen+1,1);
  double   *perror = malloc(sizeof(double)*4*(plen+1));
  eintvl_t *eintvl = malloc(sizeof(eintvl_t)*ecap);
  eintvl_t *ointvl = malloc(sizeof(eintvl_t)*ecap);
  eintvl_t *tmp    = malloc(sizeof(eintvl_t)*ecap);
  int ret = -1;

  W->p = p; W->profile = profile; W->plen = plen;
  W->ctx[CPO_DROP] = lctx+(size_t)(K-2)*3;
  W->ctx[CPO_GAIN] = rctx;
  W->wall = wall; W->perror = perror;
  for (int i = 0; i <= plen; i++)                   /* wall.c:581-586 (+ index plen, see header) */
    for (int x = 0; x < 4; x++)
      perror[(size_t)i*4+x] = -INFINITY;

  int ct[2] = {0,0};
  int eidx = 0, oidx = 0;
  for (int i = 1; i < plen; i++)                    /* wall.c:590-707 */
    { int cim1 = profile[i-1], ci = profile[i];
      if (MINI(cim1,ci) >= REP)
        continue;
      int cng = abs(cim1-ci);
      if (cng < MIN_CNT_CHANGE)
        continue;

      int wtype, cin, cout;
      if (cim1 > ci) { wtype = CPO_DROP; cin = ci;   cout = cim1; }
      else           { wtype = CPO_GAIN; cin = cim1; cout = ci;   }

      int maxt = -1, maxl = -1;                     /* wall.c:624-634 */
      double maxpe = -INFINITY;
      for (int t = 0; t < 3; t++)
        { int l = MINI(CTX(W,wtype,i,t),p->lmax[t]);
          double pe = p->pe[t][l];
          if (maxpe < pe)
            { maxpe = pe; maxt = t; maxl = l; }
        }

      for (int e = CPO_SELF; e <= CPO_OTHERS; e++)  /* wall.c:638-691 */
        { if (wall[i] & MASK_PAIRED_BY[e])
            continue;
          if (cout < CMAX)
            { for (int s = 0; s < 2; s++)
                ct[s] = p->cthres[maxt][maxl][cout][s][e];
              if (!(cng > MAX_CNT_CHANGE || cin < MAXI(ct[CPO_INIT],3)))
                continue;
            }
          if (e == CPO_SELF)
            { if (cout < CMAX && cin >= ct[CPO_FINAL])
                continue;
              update_perror(W,i,e,wtype,cout,cin,maxpe);
              if (PERR(W,i,e,wtype) < PE_THRES[CPO_FINAL][e])
                continue;
              eintvl_t I;
              int found = (wtype == CPO_DROP) ? find_gain(W,i,cout,cin,e,maxt,maxl,maxpe,&I)
                                              : find_drop(W,i,cout,cin,e,maxt,maxl,maxpe,&I);
              if (found && I.pe >= PE_THRES[CPO_FINAL][e])
                { wall[I.b] |= MASK_WALL_BY[e];
                  wall[I.e] |= MASK_WALL_BY[e];
                  wall[I.b] |= MASK_PAIRED_BY[e];
                  wall[I.e] |= MASK_PAIRED_BY[e];
                  if (eidx >= ecap-1) goto done;
                  eintvl[eidx++] = I;
                }
            }
          else
            { if (cng >= HAP || (cout < CMAX && cin < ct[CPO_FINAL]))
                { wall[i] |= MASK_WALL_BY[CPO_OTHERS];
                  continue;
                }
              update_perror(W,i,e,wtype,cout,cin,maxpe);
              if (PERR(W,i,e,wtype) < PE_THRES[CPO_FINAL][e])
                { wall[i] |= MASK_WALL_BY[CPO_OTHERS];
                  continue;
                }
              eintvl_t I;
              int found = (wtype == CPO_DROP) ? find_gain(W,i,cout,cin,e,maxt,maxl,maxpe,&I)
                                              : find_drop(W,i,cout,cin,e,maxt,maxl,maxpe,&I);
              if (found && I.pe >= PE_THRES[CPO_FINAL][e])
                { wall[I.b] |= MASK_PAIRED_BY[e];
                  wall[I.e] |= MASK_PAIRED_BY[e];
                  if (oidx >= ecap-1) goto done;
                  ointvl[oidx++] = I;
                  continue;
                }
              wall[i] |= MASK_WALL_BY[e];
            }
        }
    }

  int NS = eidx, NO = oidx;

  for (int i = 0; i < NO; i++)                      /* wall.c:722-731 */
    { wall[ointvl[i].b] &= ~MASK_WALL_BY[CPO_OTHERS];
      wall[ointvl[i].e] &= ~MASK_WALL_BY[CPO_OTHERS];
    }
  for (int i = 0; i < NS; i++)
    for (int j = eintvl[i].b+1; j < eintvl[i].e; j++)
      wall[j] &= ~MASK_WALL_BY[CPO_OTHERS];

  NS = remove_duplicates(eintvl,eidx,tmp);          /* wall.c:734-735 */
  NO = remove_duplicates(ointvl,oidx,tmp);
  (void)NO;

  int midx = NS;                                    /* wall.c:760-861 */
  double pe, pe_i, pe_j;
  for (int i = 1; i < plen; i++)
    { if (!((wall[i] & MASK_WALL_BY[CPO_OTHERS]) && !(wall[i] & MASK_WALL_BY[CPO_SELF])))
        continue;
      if (wall[i] & MASK_PAIRED_MULT)
        continue;
      for (int w = CPO_DROP; w <= CPO_GAIN; w++)
        { if ((pe_i = PERR(W,i,CPO_SELF,w)) < PE_THRES[CPO_FINAL][CPO_SELF])
            continue;
          if (w == CPO_DROP)
            { for (int j = i+1; j < MINI(i+200,plen+1); j++)
                { if (j == plen)
                    { if ((pe = pe_i * pe_i) < PE_THRES[CPO_FINAL][CPO_SELF])
                        continue;
                      eintvl[midx].b = i; eintvl[midx].e = plen; eintvl[midx].pe = pe;
                      wall[i] |= MASK_PAIRED_MULT;
                      midx++;
                      if (midx >= plen) goto done;  /* reference exits ("# E-intvls >= plen") */
                    }
                  if (!(wall[j] & MASK_WALL_BY[CPO_SELF]) && !(wall[j] & MASK_WALL_BY[CPO_OTHERS]))
                    continue;
                  if (bs_eintvl(eintvl,0,NS-1,i,j) == -1)
                    { pe_j = PERR(W,j,CPO_SELF,CPO_GAIN);
                      if ((pe = pe_i * pe_j) >= PE_THRES[CPO_FINAL][CPO_SELF])
                        { eintvl[midx].b = i; eintvl[midx].e = j; eintvl[midx].pe = pe;
                          wall[i] |= MASK_PAIRED_MULT;
                          wall[j] |= MASK_PAIRED_MULT;
                          midx++;
                          if (midx >= plen) goto done;
                        }
                    }
                  if (wall[j] & MASK_WALL_BY[CPO_OTHERS])
                    break;
                }
            }
          else
            { for (int j = i-1; j >= MAXI(i-200,0); j--)
                { if (j == 0)
                    { if ((pe = pe_i * pe_i) < PE_THRES[CPO_FINAL][CPO_SELF])
                        continue;
                      eintvl[midx].b = 0; eintvl[midx].e = i; eintvl[midx].pe = pe;
                      wall[i] |= MASK_PAIRED_MULT;
                      midx++;
                      if (midx >= plen) goto done;
                    }
                  if (!(wall[j] & MASK_WALL_BY[CPO_SELF]) && !(wall[j] & MASK_WALL_BY[CPO_OTHERS]))
                    continue;
                  if (bs_eintvl(eintvl,0,NS-1,j,i) == -1)
                    { pe_j = PERR(W,j,CPO_SELF,CPO_DROP);
                      if ((pe = pe_i * pe_j) >= PE_THRES[CPO_FINAL][CPO_SELF])
                        { eintvl[midx].b = j; eintvl[midx].e = i; eintvl[midx].pe = pe;
                          wall[i] |= MASK_PAIRED_MULT;
                          wall[j] |= MASK_PAIRED_MULT;
                          midx++;
                          if (midx >= plen) goto done;
                        }
                    }
                  if (wall[j] & MASK_WALL_BY[CPO_OTHERS])
                    break;
                }
            }
        }
    }

  for (int i = NS; i < midx; i++)                   /* wall.c:868-872 */
    for (int j = eintvl[i].b+1; j < eintvl[i].e; j++)
      wall[j] &= ~MASK_WALL_BY[CPO_OTHERS];
  if (NS < midx)                                    /* wall.c:873-876 */
    { NS = midx;
      sort_eintvl(eintvl,NS,tmp);
    }

  { int i = 0, j;                                   /* wall.c:879-909: merge overlapping E-intvls */
    while (i < NS-1)
      { int    max_e  = eintvl[i].e;
        double max_pe = eintvl[i].pe;
        j = i;
        while (j < NS-1)
          { if (eintvl[j+1].b <= eintvl[j].e)
              { max_e  = MAXI(max_e,eintvl[j+1].e);
                max_pe = (max_pe > eintvl[j+1].pe) ? max_pe : eintvl[j+1].pe;
                j++;
              }
            else
              break;
          }
        if (i < j)
          { eintvl[NS].b = eintvl[i].b; eintvl[NS].e = max_e; eintvl[NS].pe = max_pe;
            NS++;
            if (NS >= plen) goto done;
          }
        i = j+1;
      }
  }
  sort_eintvl(eintvl,NS,tmp);                       /* wall.c:910 */

  for (int i = 0; i < NS; i++)                      /* wall.c:917-919 */
    for (int j = eintvl[i].b; j < eintvl[i].e; j++)
      wall[j] |= MASK_ERROR;

  { int N = 0, b = 0;                               /* wall.c:922-948 */
    for (int i = 1; i <= plen; i++)
      if (i == plen
          || ((wall[i-1] & MASK_ERROR) != 0) != ((wall[i] & MASK_ERROR) != 0)
          || (!(wall[i] & MASK_ERROR) && (wall[i] & MASK_WALL_BY[CPO_OTHERS])))
        { int e = i;
          int idx = bs_eintvl(eintvl,0,NS-1,b,e);
          if (N >= cap) goto done;
          memset(&intvl[N],0,sizeof(cpo_intvl));
          intvl[N].b = b;
          intvl[N].e = e;
          intvl[N].cb = profile[b];
          intvl[N].ce = profile[e-1];
          intvl[N].is_rel = 0;
          intvl[N].pe = (idx != -1) ? log(eintvl[idx].pe) : -INFINITY;
          double peob = (PERR(W,b,CPO_OTHERS,CPO_DROP) > PERR(W,b,CPO_OTHERS,CPO_GAIN)) ? PERR(W,b,CPO_OTHERS,CPO_DROP) : PERR(W,b,CPO_OTHERS,CPO_GAIN);
          double peoe = (PERR(W,e,CPO_OTHERS,CPO_DROP) > PERR(W,e,CPO_OTHERS,CPO_GAIN)) ? PERR(W,e,CPO_OTHERS,CPO_DROP) : PERR(W,e,CPO_OTHERS,CPO_GAIN);
          intvl[N].peo_b = (peob != -INFINITY) ? log(peob) : -INFINITY;
          intvl[N].peo_e = (peoe != -INFINITY) ? log(peoe) : -INFINITY;
          intvl[N].asgn = CPO_N_STATE;
          N++;
          b = e;
        }
    ret = N;
  }

done:
  free(wall); free(perror); free(eintvl); free(ointvl); free(tmp);
  return ret;
}

/* ------------------------------------------------------------------------------------------
 *  correct_wall_cnt + find_rel_intvl, wall.c:960-1051
 * ------------------------------------------------------------------------------------------ */
int cpo_find_rel_intvl(const cpo_params *p, cpo_intvl *intvl, int N, cpo_intvl *rintvl,
                       const uint16_t *profile, int plen, const uint8_t *lctx, const uint8_t *rctx)
{ const int K = p->K;
  const uint8_t *ctxD = lctx+(size_t)(K-2)*3, *ctxG = rctx;
  /* intvl[] is indexed by *position* at wall.c:999-1006; shadow the slots beyond N. */
  uint16_t *sh_ccb = calloc(plen+1,sizeof(uint16_t));
  uint16_t *sh_cce = calloc(plen+1,sizeof(uint16_t));
#define CCB(x) (*((x) < N ? &intvl[x].ccb : &sh_ccb[x]))
#define CCE(x) (*((x) < N ? &intvl[x].cce : &sh_cce[x]))
  int M = 0;
  double logpthres = log(PE_THRES[CPO_FINAL][CPO_SELF]);

  for (int idx = 0; idx < N; idx++)
    { cpo_intvl I = intvl[idx];
      if (I.e-I.b < K)                              /* wall.c:1021-1022 */
        continue;
      if (MAXI(I.cb,I.ce) >= p->cov[CPO_REPEAT])    /* wall.c:1023-1024 */
        continue;
      if (I.pe >= logpthres)                        /* wall.c:1025-1026 */
        continue;

      { int first, last, n_gain = 0, n_drop = 0, lmax;   /* correct_wall_cnt, wall.c:960-1014 */
        last = MINI(I.b+K-1,I.e-1);
        for (int i = I.b; i < last; i++)
          n_gain += MAXI((int)profile[i+1]-profile[i],0);
        if (I.b+K-1 < I.e)
          { lmax = 0;
            for (int t = 0; t < 3; t++)
              { int l = ctxG[(size_t)(I.b+K-1)*3+t]*(t+1);
                if (lmax < l) lmax = l;
              }
            last = I.b+lmax;
            /* wall.c:976-978 reads profile[plen] when the low-complexity run that starts at read base b+K-1
               reaches the end of the read (last == plen) -- and cells beyond it when that run is a homopolymer of
               more than 127 bases: context.c:24-25 fills rctx with a reversed copy of CAPPED values there, so lmax can
               be 127 with fewer than 127 bases left (last > plen, by up to 126).  In the reference those cells are
               whatever the thread's profile buffer holds: 0 on fresh heap (the first read of a thread), else a
               longer earlier read's counts.  DEFINED here (hazard 8): profile[x] == 0 for x >= plen. */
            for (int i = I.b; i < last; i++)
              n_gain -= MAXI((i < plen ? (int)profile[i] : 0)-(i+1 < plen ? (int)profile[i+1] : 0),0);
          }
        first = MAXI(I.e-K+1,I.b);
        for (int i = first; i < I.e-1; i++)
          n_drop += MAXI((int)profile[i]-profile[i+1],0);
        if (I.b < I.e-K+1)
          { lmax = 0;
            for (int t = 0; t < 3; t++)
              { int l = ctxD[(size_t)(I.e-K+1)*3+t]*(t+1);
                if (lmax < l) lmax = l;
              }
            first = I.e-lmax;
            for (int i = first; i < I.e-1; i++)
              n_drop -= MAXI((int)profile[i+1]-profile[i],0);
          }
        intvl[idx].ccb = (uint16_t)MINI(I.cb+MAXI(n_gain,0),CPO_MAX_KMER_CNT);
        intvl[idx].cce = (uint16_t)MINI(I.ce+MAXI(n_drop,0),CPO_MAX_KMER_CNT);

        last = MINI(I.b+2*K,I.e);                   /* wall.c:999-1006 (index = position, literal) */
        for (int i = I.b; i < last; i++)
          if (CCB(i) < profile[i])
            CCB(i) = profile[i];
        first = MAXI(I.e-2*K,I.b);
        for (int i = first; i < I.e; i++)
          if (CCE(i) < profile[i])
            CCE(i) = profile[i];
      }

      if (logp_trans(p,intvl[idx].b,intvl[idx].e,intvl[idx].ccb,intvl[idx].cce,
                     (intvl[idx].ccb+intvl[idx].cce)/2) < THRES_DIFF_REL)      /* wall.c:1028-1030 */
        continue;
      if (MAXI(intvl[idx].ccb,intvl[idx].cce) == CPO_MAX_KMER_CNT)             /* wall.c:1032-1033 */
        continue;
      intvl[idx].is_rel = 1;
      rintvl[M] = intvl[idx];
      M++;
    }
#undef CCB
#undef CCE
  free(sh_ccb); free(sh_cce);
  return M;
}

/* ------------------------------------------------------------------------------------------
 *  class_rel.c
 * ------------------------------------------------------------------------------------------ */
typedef pos_cnt covs_t[4];                          /* ClassPro.h:206 */

typedef struct                                      /* ClassPro.h:210-219 */
  { const cpo_params *p;
    int        FORWARD;
    int        COV[4];
    double    *dp;         /* [M*4] */
    covs_t    *st;         /* [M*4] */
    int8_t    *bt;         /* [M*4][M] */
    double    *dh_ratio;   /* [M*4] */
    uint8_t   *rpos;       /* [M] */
    cpo_intvl *intvl;      /* [M] */
    int        M;
  } rel_arg;

#define REL_IDX(i,s) ((i)*4+(s))
#define BT(A,idx)    ((A)->bt+(size_t)(idx)*(A)->M)

static inline int _pred(int x, int F)   { return F ? x-1 : x+1; }             /* class_rel.c:39-40 */
static inline int _succ(int x, int F)   { return F ? x+1 : x-1; }             /* class_rel.c:42-43 */
static inline int _offset(int x, int F) { return F ? x-OFFSET : x+OFFSET; }   /* class_rel.c:45-46 */
static inline int _beg_pos(const cpo_intvl *I, int F) { return F ? I->b : I->e-1; }     /* :48-49 */
static inline int _beg_cnt(const cpo_intvl *I, int F) { return F ? I->ccb : I->cce; }   /* :51-52 */
static inline int _end_pos(const cpo_intvl *I, int F) { return F ? I->e-1 : I->b; }     /* :54-55 */
static inline int _end_cnt(const cpo_intvl *I, int F) { return F ? I->cce : I->ccb; }   /* :57-58 */

static inline int find_max_dp(const double *dp, int i)                      /* class_rel.c:62-73 */
{ double max_logp = -INFINITY;
  int max_s = CPO_N_STATE;
  for (int s = 0; s < 4; s++)
    if (max_logp < dp[REL_IDX(i,s)])
      { max_logp = dp[REL_IDX(i,s)];
        max_s = s;
      }
  return max_s;
}

typedef struct { int max_x; double max_logp; } max_cell;

static inline max_cell find_max_dp_tr(const double *dp, double logp_tr[4][4], int i, int s, int t, int F)  /* :80-96 */
{ int i_pred = _pred(i,F);
  max_cell r = { CPO_N_STATE, -INFINITY };
  for (int x = 0; x < 4; x++)
    { int _s = (s < CPO_N_STATE) ? s : x;
      int _t = (t < CPO_N_STATE) ? t : x;
      double logp = dp[REL_IDX(i_pred,_s)]+logp_tr[_s][_t];
      if (r.max_logp < logp)
        { r.max_logp = logp;
          r.max_x = x;
        }
    }
  return r;
}

static inline int find_nn(int forward, int i, int s, const int8_t *asgn, int L)   /* class_rel.c:98-107 */
{ int idx = i;
  if (forward)
    while (idx < L && asgn[idx] != (int8_t)s) idx++;
  else
    while (idx >= 0 && asgn[idx] != (int8_t)s) idx--;
  return idx;
}

static double calc_dh_ratio(int init_s, const int8_t *asgn, const cpo_intvl *intvl, int L, int F)  /* class_rel.c:113-156 */
{ int idx[4];
  idx[0] = F ? L : -1;
  int s = init_s;
  for (int i = 0; i < 3; i++)
    { idx[i+1] = find_nn(!F,_pred(idx[i],F),s,asgn,L);
      if ((F && idx[i+1] < 0) || (!F && idx[i+1] >= L))
        return -INFINITY;
      s = (s == CPO_HAPLO) ? CPO_DIPLO : CPO_HAPLO;
    }
  pos_cnt s1 = { _beg_pos(&intvl[idx[1]],F), _beg_cnt(&intvl[idx[1]],F) };
  pos_cnt t  = { _end_pos(&intvl[idx[2]],F), _end_cnt(&intvl[idx[2]],F) };
  pos_cnt s2 = { _end_pos(&intvl[idx[3]],F), _end_cnt(&intvl[idx[3]],F) };
  if (!F)
    { pos_cnt tmp = s1; s1 = s2; s2 = tmp; }
  double est_s_cnt = linear_interpolation(t.pos,s2,s1);
  return (init_s == CPO_DIPLO) ? est_s_cnt/t.cnt : t.cnt/est_s_cnt;
}

static double logp_e(const rel_arg *A, int idx)                             /* class_rel.c:158-170 */
{ const cpo_intvl *I = &A->intvl[idx];
  double logp_er = I->pe;
  double logp_po = logp_poisson(A->p,I->ccb,A->COV[CPO_ERROR])+logp_poisson(A->p,I->cce,A->COV[CPO_ERROR])+E_PO_BASE;
  return (logp_po > logp_er) ? logp_po : logp_er;
}

static double logp_r(const rel_arg *A, int idx, pos_cnt st_pred_r)          /* class_rel.c:172-211 */
{ const cpo_intvl *I = &A->intvl[idx];
  int beg_cnt = _beg_cnt(I,A->FORWARD);
  double logp_sf = -INFINITY;
  double logp_er = (beg_cnt < st_pred_r.cnt) ? logp_binom(A->p,beg_cnt,st_pred_r.cnt,1-PE_MEAN) : -INFINITY;
  double logp = (logp_sf > logp_er) ? logp_sf : logp_er;
  if (logp > R_LOGP)
    return logp;
  int max_cc = MAXI(I->ccb,I->cce);
  if (max_cc >= A->COV[CPO_REPEAT])
    return R_LOGP;
  if (max_cc >= st_pred_r.cnt)
    return R_LOGP;
  return logp;
}

static double logp_h(const rel_arg *A, int idx, int s, const pos_cnt *st_pred)   /* class_rel.c:213-240 */
{ const int F = A->FORWARD;
  const cpo_intvl *I = &A->intvl[idx];
  int beg_pos = _beg_pos(I,F), beg_cnt = _beg_cnt(I,F);
  pos_cnt st = st_pred[CPO_HAPLO];
  double logp_sf_h = logp_trans(A->p,_pred(st.pos,F),beg_pos,st.cnt,beg_cnt,st.cnt);
  double logp_sf_d = 0.;
  double r = A->dh_ratio[REL_IDX(_pred(idx,F),s)];
  if (r != -INFINITY)
    { st = st_pred[CPO_DIPLO];
      logp_sf_h = logp_trans(A->p,_pred(st.pos,F),beg_pos,st.cnt,(int)(r*beg_cnt),st.cnt);
    }
  return logp_sf_h+logp_sf_d;
}

static double logp_d(const rel_arg *A, int idx, int s, const pos_cnt *st_pred)   /* class_rel.c:242-270 */
{ const int F = A->FORWARD;
  const cpo_intvl *I = &A->intvl[idx];
  int beg_pos = _beg_pos(I,F), beg_cnt = _beg_cnt(I,F);
  /* The D/H-ratio branch at :253-259 computes a value that :264 overwrites; only the
     D-anchored transition survives (SURVEY hazard 3). */
  (void)s;
  pos_cnt st = st_pred[CPO_DIPLO];
  double logp_sf_h = logp_trans(A->p,_pred(st.pos,F),beg_pos,st.cnt,beg_cnt,st.cnt);
  double logp_sf_d = 0.;
  return logp_sf_h+logp_sf_d;
}

static double calc_logp(const rel_arg *A, int s, int t, int idx, const pos_cnt *st_pred)   /* class_rel.c:272-277 */
{ if (t == CPO_ERROR)      return logp_e(A,idx);
  else if (t == CPO_HAPLO) return logp_h(A,idx,s,st_pred);
  else if (t == CPO_DIPLO) return logp_d(A,idx,s,st_pred);
  else                     return logp_r(A,idx,st_pred[CPO_REPEAT]);
}

static void rel_update(rel_arg *A, int i)                                   /* class_rel.c:279-513 */
{ const int F = A->FORWARD, M = A->M;
  const int *COV = A->COV;
  double *dp = A->dp;
  covs_t *st = A->st;
  double *dh_ratio = A->dh_ratio;
  cpo_intvl *intvl = A->intvl;

  cpo_intvl I = intvl[i];
  int end_pos = _end_pos(&I,F), end_cnt = _end_cnt(&I,F);
  int i_pred = _pred(i,F);

  double logp_tr[4][4];
  for (int s = 0; s < 4; s++)
    for (int t = 0; t < 4; t++)
      logp_tr[s][t] = -INFINITY;
  for (int s = 0; s < 4; s++)
    { int idx = REL_IDX(i_pred,s);
      if (dp[idx] == -INFINITY)
        { for (int t = 0; t < 4; t++)
            logp_tr[s][t] = 0.;
          continue;
        }
      for (int t = 0; t < 4; t++)
        logp_tr[s][t] = exp(calc_logp(A,s,t,i,st[idx]));
    }
  double psum = 0.;
  for (int s = 0; s < 4; s++)
    for (int t = 0; t < 4; t++)
      psum += logp_tr[s][t];
  if (psum == 0.)                                   /* class_rel.c:324-333 (DEBUG on) */
    { fprintf(stderr,"No possible state @ %d\n",i);
      for (int s = 0; s < 4; s++)
        logp_tr[s][CPO_ERROR] = 1.;
      psum = 4.;
    }
  for (int s = 0; s < 4; s++)
    for (int t = 0; t < 4; t++)
      logp_tr[s][t] = log(logp_tr[s][t]/psum);

  int only_r = 1;                                   /* class_rel.c:348-380 */
  for (int s = 0; s < 4; s++)
    { int maxt = find_max_dp_tr(dp,logp_tr,i,s,CPO_N_STATE,F).max_x;
      if (maxt != CPO_N_STATE && maxt != CPO_REPEAT)
        { only_r = 0;
          break;
        }
    }
  if (only_r)
    { A->rpos[i] = 1;
      intvl[i] = intvl[i_pred];
      for (int s = 0; s < 4; s++)
        { int idx = REL_IDX(i,s), idx_pred = REL_IDX(i_pred,s);
          dp[idx] = dp[idx_pred];
          if (dp[idx] == -INFINITY)
            continue;
          if (F) { for (int ii = 0; ii < i; ii++)   BT(A,idx)[ii] = BT(A,idx_pred)[ii]; }
          else   { for (int ii = i+1; ii < M; ii++) BT(A,idx)[ii] = BT(A,idx_pred)[ii]; }
          BT(A,idx)[i] = (int8_t)s;
          for (int t = 0; t < 4; t++)
            st[idx][t] = st[idx_pred][t];
        }
      return;
    }

  int maxs_h = find_max_dp_tr(dp,logp_tr,i,CPO_N_STATE,CPO_HAPLO,F).max_x;   /* class_rel.c:382-386 */
  int maxs_d = find_max_dp_tr(dp,logp_tr,i,CPO_N_STATE,CPO_DIPLO,F).max_x;
  if (maxs_h == CPO_HAPLO && maxs_d == CPO_DIPLO)
    { double mn = (logp_tr[CPO_HAPLO][CPO_HAPLO] < logp_tr[CPO_DIPLO][CPO_DIPLO]) ? logp_tr[CPO_HAPLO][CPO_HAPLO] : logp_tr[CPO_DIPLO][CPO_DIPLO];
      logp_tr[CPO_HAPLO][CPO_HAPLO] = logp_tr[CPO_DIPLO][CPO_DIPLO] = mn;
    }

  int curr_h, curr_d, curr_r;
  double r;
  for (int t = 0; t < 4; t++)                       /* class_rel.c:390-499 */
    { max_cell mc = find_max_dp_tr(dp,logp_tr,i,CPO_N_STATE,t,F);
      int max_s = mc.max_x;
      int idx = REL_IDX(i,t);
      int idx_pred = REL_IDX(i_pred,max_s);
      dp[idx] = mc.max_logp;
      if (max_s == CPO_N_STATE)
        continue;

      if (F) { for (int ii = 0; ii < i; ii++)   BT(A,idx)[ii] = BT(A,idx_pred)[ii]; }
      else   { for (int ii = i+1; ii < M; ii++) BT(A,idx)[ii] = BT(A,idx_pred)[ii]; }
      BT(A,idx)[i] = (int8_t)t;

      if (t == CPO_ERROR)
        { for (int s = CPO_REPEAT; s <= CPO_DIPLO; s++)
            st[idx][s] = st[idx_pred][s];
        }
      else if (t == CPO_REPEAT)
        { for (int s = CPO_HAPLO; s <= CPO_DIPLO; s++)
            { st[idx][s].pos = _offset(end_pos,F);
              st[idx][s].cnt = st[idx_pred][s].cnt;
            }
          int r_cnt = MINI(end_cnt,COV[CPO_REPEAT]);
          if (st[idx_pred][CPO_REPEAT].cnt < r_cnt)
            st[idx][CPO_REPEAT] = st[idx_pred][CPO_REPEAT];
          else
            { st[idx][CPO_REPEAT].pos = _offset(end_pos,F);
              st[idx][CPO_REPEAT].cnt = r_cnt;
            }
        }
      else if (t == CPO_HAPLO)
        { curr_h = end_cnt;
          r = calc_dh_ratio(CPO_HAPLO,F ? BT(A,idx) : BT(A,idx)+i,F ? intvl : intvl+i,F ? i+1 : M-i,F);
          if (r == -INFINITY)
            { int has_d = 0;
              if (F) { for (int ii = 0; ii < i; ii++)   if (BT(A,idx)[ii] == CPO_DIPLO) has_d = 1; }
              else   { for (int ii = i+1; ii < M; ii++) if (BT(A,idx)[ii] == CPO_DIPLO) has_d = 1; }
              if (has_d)
                curr_d = st[idx_pred][CPO_DIPLO].cnt;
              else
                curr_d = curr_h+COV[CPO_HAPLO];
            }
          else
            { curr_d = (int)(r*curr_h);
              dh_ratio[idx] = r;
            }
          curr_r = (int)(A->p->dr_ratio*curr_d);
          st[idx][CPO_HAPLO].pos  = _offset(end_pos,F);
          st[idx][CPO_HAPLO].cnt  = curr_h & 0xffff;
          st[idx][CPO_DIPLO].pos  = _offset(end_pos,F);
          st[idx][CPO_DIPLO].cnt  = curr_d & 0xffff;
          st[idx][CPO_REPEAT].pos = _offset(end_pos,F);
          st[idx][CPO_REPEAT].cnt = curr_r & 0xffff;
        }
      else
        { curr_d = end_cnt;
          r = calc_dh_ratio(CPO_DIPLO,F ? BT(A,idx) : BT(A,idx)+i,F ? intvl : intvl+i,F ? i+1 : M-i,F);
          if (r == -INFINITY)
            { int has_h = 0;
              if (F) { for (int ii = 0; ii < i; ii++)   if (BT(A,idx)[ii] == CPO_HAPLO) has_h = 1; }
              else   { for (int ii = i+1; ii < M; ii++) if (BT(A,idx)[ii] == CPO_HAPLO) has_h = 1; }
              if (has_h)
                curr_h = st[idx_pred][CPO_HAPLO].cnt;
              else
                curr_h = MAXI(curr_d/2,curr_d-COV[CPO_HAPLO]);
            }
          else
            { curr_h = (int)((double)curr_d/r);
              dh_ratio[idx] = r;
            }
          curr_r = (int)(A->p->dr_ratio*curr_d);
          st[idx][CPO_HAPLO].pos  = _offset(end_pos,F);
          st[idx][CPO_HAPLO].cnt  = curr_h & 0xffff;
          st[idx][CPO_DIPLO].pos  = _offset(end_pos,F);
          st[idx][CPO_DIPLO].cnt  = curr_d & 0xffff;
          st[idx][CPO_REPEAT].pos = _offset(end_pos,F);
          st[idx][CPO_REPEAT].cnt = curr_r & 0xffff;
        }

      if (!((st[idx][CPO_HAPLO].cnt < st[idx][CPO_DIPLO].cnt)                /* class_rel.c:496-498 */
            && (st[idx][CPO_DIPLO].cnt < st[idx][CPO_REPEAT].cnt)))
        dp[idx] = -INFINITY;
    }
}

static int8_t *rel_classify(rel_arg *A, const cpo_intvl *rintvl, int plen)  /* _classify_rel, class_rel.c:515-614 */
{ const int F = A->FORWARD, M = A->M;
  const int *COV = A->COV;
  double *dp = A->dp;
  covs_t *st = A->st;
  int idx;

  for (int i = 0; i < M; i++)
    { for (int s = 0; s < 4; s++)
        { idx = REL_IDX(i,s);
          dp[idx] = -INFINITY;
          A->dh_ratio[idx] = -INFINITY;
        }
      A->rpos[i] = 0;
      A->intvl[i] = rintvl[i];
    }

  const int POS_INIT = _offset(F ? 0 : plen,F);
  int i = F ? 0 : M-1;
  cpo_intvl I = A->intvl[i];

  for (int s = 0; s < 4; s++)                       /* class_rel.c:551-558 */
    { idx = REL_IDX(i,s);
      for (int t = CPO_REPEAT; t <= CPO_DIPLO; t++)
        { st[idx][t].pos = POS_INIT;
          st[idx][t].cnt = COV[t];
        }
      BT(A,idx)[i] = (int8_t)s;
    }

  idx = REL_IDX(i,CPO_ERROR);                       /* class_rel.c:560-580 */
  dp[idx] = logp_e(A,i);

  idx = REL_IDX(i,CPO_REPEAT);
  dp[idx] = logp_r(A,i,st[idx][CPO_REPEAT]);
  st[idx][CPO_REPEAT].pos = _end_pos(&I,F);
  st[idx][CPO_REPEAT].cnt = MINI(_end_cnt(&I,F),COV[CPO_REPEAT]);

  idx = REL_IDX(i,CPO_HAPLO);
  dp[idx] = logp_poisson(A->p,_beg_cnt(&I,F),COV[CPO_HAPLO]);
  st[idx][CPO_HAPLO].pos = _end_pos(&I,F);
  st[idx][CPO_HAPLO].cnt = _end_cnt(&I,F);
  st[idx][CPO_DIPLO].pos = _offset(_end_pos(&I,F),F);
  st[idx][CPO_DIPLO].cnt = (_end_cnt(&I,F)+COV[CPO_HAPLO]) & 0xffff;

  idx = REL_IDX(i,CPO_DIPLO);
  dp[idx] = logp_poisson(A->p,_beg_cnt(&I,F),COV[CPO_DIPLO]);
  st[idx][CPO_HAPLO].pos = _offset(_end_pos(&I,F),F);
  st[idx][CPO_HAPLO].cnt = MAXI(_end_cnt(&I,F)/2,_end_cnt(&I,F)-COV[CPO_HAPLO]) & 0xffff;
  st[idx][CPO_DIPLO].pos = _end_pos(&I,F);
  st[idx][CPO_DIPLO].cnt = _end_cnt(&I,F);

  double psum = 0.;                                 /* class_rel.c:582-586 */
  for (int s = 0; s < 4; s++)
    psum += exp(dp[REL_IDX(i,s)]);
  for (int s = 0; s < 4; s++)
    dp[REL_IDX(i,s)] = log(exp(dp[REL_IDX(i,s)])/psum);

  while (1)                                         /* class_rel.c:599-605 */
    { i = _succ(i,F);
      if ((F && i >= M) || (!F && i < 0))
        break;
      rel_update(A,i);
    }

  i = F ? M-1 : 0;                                  /* class_rel.c:607-613 */
  int max_s = find_max_dp(dp,i);
  if (max_s == CPO_N_STATE)
    max_s = CPO_ERROR;   /* reference would index bt[(i+1)*4]: undefined; never seen (psum==0 fallback keeps E alive) */
  idx = REL_IDX(i,max_s);
  for (int j = 0; j < M; j++)
    if (A->rpos[j])
      BT(A,idx)[j] = CPO_REPEAT;
  return BT(A,idx);
}

typedef struct { int8_t *asgn; int d_diff, h_diff; double hdrr; } iter_rel;   /* class_rel.c:616-621 */

/* classify_rel_fw (class_rel.c:623-735) and classify_rel_bw (:737-845) differ only in the
 * direction flag and in which D interval seeds the adjusted coverage (:648 vs :761). */
static iter_rel classify_rel_dir(rel_arg *A, const cpo_intvl *rintvl, int plen, int forward)
{ const int M = A->M;
  const int *G = A->p->cov;
  A->FORWARD = forward;
  for (int s = 0; s < 4; s++)
    A->COV[s] = G[s];
  int8_t *asgn = rel_classify(A,rintvl,plen);

  int no_h = 1;
  for (int i = 0; i < M; i++)
    if (asgn[i] == CPO_HAPLO) no_h = 0;
  if (no_h)
    { int l, lsum = 0, csum = 0;
      int seed_d_idx = -1;
      for (int i = 0; i < M; i++)
        if (asgn[i] == CPO_DIPLO)
          { l = rintvl[i].e-rintvl[i].b;
            lsum += l;
            csum += (rintvl[i].ccb+rintvl[i].cce)*l/2;
            if (forward) { if (seed_d_idx == -1) seed_d_idx = i; }   /* first D, :641-642 */
            else         seed_d_idx = i;                            /* last D,  :757 */
          }
      if (seed_d_idx >= 0)
        { double mean_dcov = (double)csum/lsum;
          if (mean_dcov < G[CPO_DIPLO])
            { A->COV[CPO_HAPLO] = forward ? rintvl[seed_d_idx].ccb : rintvl[seed_d_idx].cce;
              A->COV[CPO_DIPLO] = (A->COV[CPO_HAPLO]+G[CPO_HAPLO]) & 0xffff;
              asgn = rel_classify(A,rintvl,plen);
              no_h = 1;
              for (int i = 0; i < M; i++)
                if (asgn[i] == CPO_HAPLO) no_h = 0;
              if (no_h)
                { lsum = 0; csum = 0;
                  for (int i = 0; i < M; i++)
                    if (asgn[i] == CPO_DIPLO)
                      { l = rintvl[i].e-rintvl[i].b;
                        lsum += l;
                        csum += (rintvl[i].ccb+rintvl[i].cce)*l/2;
                      }
                  mean_dcov = (double)csum/lsum;
                  if (fabs(mean_dcov-G[CPO_HAPLO]) <= fabs(mean_dcov-G[CPO_DIPLO]))
                    for (int i = 0; i < M; i++)
                      if (asgn[i] == CPO_DIPLO)
                        asgn[i] = CPO_HAPLO;
                }
            }
        }
    }

  { int all_h = 1;                                  /* class_rel.c:674-689 / :787-800 */
    for (int i = 0; i < M; i++)
      if (asgn[i] != CPO_HAPLO) all_h = 0;
    if (all_h)
      { int l, lsum = 0, csum = 0;
        for (int i = 0; i < M; i++)
          { l = rintvl[i].e-rintvl[i].b;
            lsum += l;
            csum += (rintvl[i].ccb+rintvl[i].cce)*l/2;
          }
        double mean_hcov = (double)csum/lsum;
        if (fabs(mean_hcov-G[CPO_HAPLO]) >= fabs(mean_hcov-G[CPO_DIPLO]))
          for (int i = 0; i < M; i++)
            asgn[i] = CPO_DIPLO;
      }
  }

  { int n = 0;                                      /* class_rel.c:691-712 / :802-822 */
    for (int i = 0; i < M; i++)
      if (asgn[i] == CPO_HAPLO) n++;
    if (n >= M * 0.7)
      { int l, lsum = 0, csum = 0;
        for (int i = 0; i < M; i++)
          if (asgn[i] == CPO_HAPLO)
            { l = rintvl[i].e-rintvl[i].b;
              lsum += l;
              csum += (rintvl[i].ccb+rintvl[i].cce)*l/2;
            }
        double mean_hcov = (double)csum/lsum;
        if (fabs(mean_hcov-G[CPO_HAPLO]) >= fabs(mean_hcov-G[CPO_DIPLO]))
          for (int i = 0; i < M; i++)
            { if (asgn[i] == CPO_HAPLO)      asgn[i] = CPO_DIPLO;
              else if (asgn[i] == CPO_DIPLO) asgn[i] = CPO_REPEAT;
            }
      }
  }

  int first_d = -1, last_d = -1, first_h = -1, last_h = -1;   /* class_rel.c:714-731 */
  for (int i = 0; i < M; i++)
    { if (asgn[i] == CPO_DIPLO)
        { if (first_d == -1) first_d = i;
          last_d = i;
        }
      else if (asgn[i] == CPO_HAPLO)
        { if (first_h == -1) first_h = i;
          last_h = i;
        }
    }
  iter_rel ret;
  ret.asgn   = asgn;
  ret.d_diff = (first_d >= 0) ? abs(rintvl[first_d].ccb-rintvl[last_d].cce) : 0;
  ret.h_diff = (first_h >= 0) ? abs(rintvl[first_h].ccb-rintvl[last_h].cce) : 0;
  ret.hdrr   = (first_d >= 0 && first_h >= 0)
               ? ((double)rintvl[first_d].ccb/rintvl[first_h].ccb)/((double)rintvl[last_d].cce/rintvl[last_h].cce) : 1.;
  return ret;
}

/* class_rel.c:847-869: `.asgn != true` compares the state code with 1 (== REPEAT), and the
 * scans treat any non-ERROR code as "true". */
static int is_eq_prefix(const cpo_intvl *r, int M)
{ if (r[0].asgn != 1) return 0;
  int i = 0;
  while (i < M && r[i].asgn) i++;
  while (i < M)
    { if (r[i].asgn) return 0;
      i++;
    }
  return 1;
}

static int is_eq_suffix(const cpo_intvl *r, int M)
{ if (r[M-1].asgn != 1) return 0;
  int i = M-2;
  while (i >= 0 && r[i].asgn) i--;
  while (i >= 0)
    { if (r[i].asgn) return 0;
      i--;
    }
  return 1;
}

void cpo_classify_rel(const cpo_params *p, cpo_intvl *rintvl, int M, cpo_intvl *intvl, int N, int plen,
                      int8_t *fw_out, int8_t *bw_out)                       /* class_rel.c:871-963 */
{ if (M == 0)
    return;
  rel_arg A;
  A.p = p; A.M = M;
  A.dp       = malloc(sizeof(double)*4*M);
  A.st       = malloc(sizeof(covs_t)*4*M);
  A.bt       = calloc((size_t)4*M*M,1);
  A.dh_ratio = malloc(sizeof(double)*4*M);
  A.rpos     = malloc(M);
  A.intvl    = malloc(sizeof(cpo_intvl)*M);
  memset(A.st,0,sizeof(covs_t)*4*M);

  iter_rel cr_f = classify_rel_dir(&A,rintvl,plen,1);
  for (int i = 0; i < M; i++)
    rintvl[i].asgn = cr_f.asgn[i];
  if (fw_out) memcpy(fw_out,cr_f.asgn,M);

  iter_rel cr_b = classify_rel_dir(&A,rintvl,plen,0);
  if (bw_out) memcpy(bw_out,cr_b.asgn,M);

  int eq = 1;
  for (int i = 0; i < M; i++)
    if (rintvl[i].asgn != cr_b.asgn[i])
      { eq = 0;
        break;
      }
  if (!eq)
    { if (is_eq_prefix(rintvl,M))
        { /* keep forward */ }
      else if (is_eq_suffix(rintvl,M))
        { for (int i = 0; i < M; i++)
            rintvl[i].asgn = cr_b.asgn[i];
        }
      else if (!(fabs(cr_f.hdrr-1.) <= fabs(cr_b.hdrr-1.)))
        { for (int i = 0; i < M; i++)
            rintvl[i].asgn = cr_b.asgn[i];
        }
    }

  for (int ridx = 0, iidx = 0; ridx < M; ridx++, iidx++)   /* class_rel.c:949-960 */
    { while (iidx < N && !intvl[iidx].is_rel)
        iidx++;
      if (iidx >= N || rintvl[ridx].b != intvl[iidx].b || rintvl[ridx].e != intvl[iidx].e)
        { fprintf(stderr,"Inconsistent reliable interval (%d,%d)\n",rintvl[ridx].b,rintvl[ridx].e);
          exit(1);
        }
      intvl[iidx].asgn = rintvl[ridx].asgn;
    }

  free(A.dp); free(A.st); free(A.bt); free(A.dh_ratio); free(A.rpos); free(A.intvl);
}

/* ------------------------------------------------------------------------------------------
 *  class_unrel.c
 * ------------------------------------------------------------------------------------------ */
static void find_nn_u(int idx, int s, const cpo_intvl *intvl, int N, int ret[2])   /* class_unrel.c:11-25 */
{ int l = idx-1;
  while (l >= 0 && !(intvl[l].asgn == (int8_t)s && intvl[l].is_rel))
    l--;
  if (l < 0) l = -1;
  ret[0] = l;
  int r = idx+1;
  while (r < N && !(intvl[r].asgn == (int8_t)s && intvl[r].is_rel))
    r++;
  if (r >= N) r = -1;
  ret[1] = r;
}

static int est_cov(const cpo_params *p, int x, int idx, const cpo_intvl *intvl, int N, int s, int from_est)  /* class_unrel.c:27-51 */
{ int nn[2];
  find_nn_u(idx,s,intvl,N,nn);
  int l = nn[0], r = nn[1];
  if (l != -1 && r != -1)
    { pos_cnt pc1 = { intvl[l].e-1, intvl[l].cce }, pc2 = { intvl[r].b, intvl[r].ccb };
      return (int)(uint16_t)linear_interpolation(x,pc1,pc2);
    }
  else if (l != -1)
    return intvl[l].cce;
  else if (r != -1)
    return intvl[r].ccb;
  if (from_est)
    return 0;
  int cov = est_cov(p,x,idx,intvl,N,(s == CPO_HAPLO) ? CPO_DIPLO : CPO_HAPLO,1);
  if (cov > 0)
    return ((s == CPO_HAPLO) ? cov/2 : cov*2) & 0xffff;
  else
    return p->cov[s];
}

static double logp_e_u(const cpo_params *p, int idx, const cpo_intvl *intvl)   /* class_unrel.c:53-65 */
{ const cpo_intvl *I = &intvl[idx];
  double logp_er = I->pe;
  double logp_po = logp_poisson(p,I->cb,p->cov[CPO_ERROR])+logp_poisson(p,I->ce,p->cov[CPO_ERROR])+E_PO_BASE;
  return (logp_er > logp_po) ? logp_er : logp_po;
}

static double logp_r_u(const cpo_params *p, int idx, const cpo_intvl *intvl, int N)   /* class_unrel.c:67-113 */
{ const cpo_intvl *I = &intvl[idx];
  if (MAXI(I->cb,I->ce) >= p->cov[CPO_REPEAT])
    return 0.;
  int nn[2];
  find_nn_u(idx,CPO_DIPLO,intvl,N,nn);
  int l = nn[0], r = nn[1];
  int dcov_l, dcov_r;
  if (l == -1 && r == -1)  dcov_l = dcov_r = p->cov[CPO_DIPLO];
  else if (l == -1)        dcov_l = dcov_r = intvl[r].cb;
  else if (r == -1)        dcov_l = dcov_r = intvl[l].ce;
  else                     { dcov_l = intvl[l].ce; dcov_r = intvl[r].cb; }
  int rcov_l = (uint16_t)(p->dr_ratio*dcov_l);
  int rcov_r = (uint16_t)(p->dr_ratio*dcov_r);
  if (I->cb >= rcov_l || I->ce >= rcov_r)
    return R_LOGP;
  double logp_l = logp_binom(p,I->cb,rcov_l,1-PE_MEAN);
  double logp_r = logp_binom(p,I->ce,rcov_r,1-PE_MEAN);
  return logp_l+logp_r;
}

static double logp_hd_u(const cpo_params *p, int s, int idx, const cpo_intvl *intvl, int N)   /* class_unrel.c:115-175 */
{ const cpo_intvl *I = &intvl[idx];
  int nn[2];
  find_nn_u(idx,s,intvl,N,nn);
  int l_rel = nn[0], r_rel = nn[1];
  double logp_l, logp_r;

  { double er = -INFINITY, sf = -INFINITY, sf_er = -INFINITY;
    int l = idx-1;
    if (l >= 0 && intvl[l].asgn == (int8_t)s)
      er = I->peo_b;
    if (l_rel != -1)
      { const cpo_intvl *L = &intvl[l_rel];
        sf = logp_trans(p,L->e-1,I->b,L->cce,I->cb,L->cce);
      }
    int est_cnt = est_cov(p,I->b,idx,intvl,N,s,0);
    if (est_cnt >= I->cb)
      sf_er = log(p_errorin(p,CPO_OTHERS,0.1,est_cnt,I->cb));
    double m = (er > sf) ? er : sf;
    logp_l = (m > sf_er) ? m : sf_er;
  }
  { double er = -INFINITY, sf = -INFINITY, sf_er = -INFINITY;
    int r = idx+1;
    if (r < N && intvl[r].asgn == (int8_t)s)
      er = I->peo_e;
    if (r_rel != -1)
      { const cpo_intvl *Rr = &intvl[r_rel];
        sf = logp_trans(p,I->e-1,Rr->b,I->ce,Rr->ccb,Rr->ccb);
      }
    int est_cnt = est_cov(p,I->e-1,idx,intvl,N,s,0);
    if (est_cnt >= I->ce)
      sf_er = log(p_errorin(p,CPO_OTHERS,0.1,est_cnt,I->ce));
    double m = (er > sf) ? er : sf;
    logp_r = (m > sf_er) ? m : sf_er;
  }

  if (logp_l == -INFINITY && logp_r == -INFINITY)
    { logp_l = logp_poisson(p,I->cb,p->cov[s]);
      logp_r = logp_poisson(p,I->ce,p->cov[s]);
    }
  else if (logp_l == -INFINITY)
    logp_l = logp_r;
  else if (logp_r == -INFINITY)
    logp_r = logp_l;
  return logp_l+logp_r;
}

static void update_state(const cpo_params *p, int idx, cpo_intvl *intvl, int N)   /* class_unrel.c:192-236 */
{ const cpo_intvl *I = &intvl[idx];
  if (MAXI(I->cb,I->ce) >= p->cov[CPO_REPEAT])
    { intvl[idx].asgn = CPO_REPEAT;
      return;
    }
  double logpmax = -INFINITY;
  int smax = -1;
  for (int s = 0; s < 4; s++)
    { double logp;
      if (s == CPO_ERROR)      logp = logp_e_u(p,idx,intvl);
      else if (s == CPO_HAPLO) logp = logp_hd_u(p,CPO_HAPLO,idx,intvl,N);
      else if (s == CPO_DIPLO) logp = logp_hd_u(p,CPO_DIPLO,idx,intvl,N);
      else                     logp = logp_r_u(p,idx,intvl,N);
      if (logpmax < logp)
        { logpmax = logp;
          smax = s;
        }
    }
  if (smax == -1)
    { fprintf(stderr,"No valid probability for interval %d\n",idx);
      exit(1);
    }
  intvl[idx].asgn = (int8_t)smax;
}

void cpo_classify_unrel(const cpo_params *p, cpo_intvl *intvl, int N)      /* class_unrel.c:248-300 */
{ if (N <= 0) return;
  uint8_t *is_fixed = malloc(N);
  int *ord = malloc(sizeof(int)*N), *key = malloc(sizeof(int)*N), *tmp = malloc(sizeof(int)*N);
  for (int i = 0; i < N; i++)
    { is_fixed[i] = (intvl[i].is_rel && (intvl[i].asgn == CPO_HAPLO || intvl[i].asgn == CPO_DIPLO));
      ord[i] = i;
      key[i] = MINI(intvl[i].cb,intvl[i].ce);
    }
  /* qsort(compare_iic) on {idx,cnt}: glibc merge sort => stable ascending by cnt */
  for (int w = 1; w < N; w *= 2)
    { for (int lo = 0; lo < N; lo += 2*w)
        { int mid = MINI(lo+w,N), hi = MINI(lo+2*w,N);
          int a = lo, b = mid, k = lo;
          while (a < mid && b < hi)
            { if (key[ord[b]] < key[ord[a]]) tmp[k++] = ord[b++];
              else                           tmp[k++] = ord[a++];
            }
          while (a < mid) tmp[k++] = ord[a++];
          while (b < hi)  tmp[k++] = ord[b++];
        }
      memcpy(ord,tmp,sizeof(int)*N);
    }
  for (int i = N-1; i >= 0; i--)
    if (!is_fixed[ord[i]])
      update_state(p,ord[i],intvl,N);
  for (int i = 0; i < N; i++)
    if (!is_fixed[ord[i]])
      update_state(p,ord[i],intvl,N);
  free(is_fixed); free(ord); free(key); free(tmp);
}

/* ------------------------------------------------------------------------------------------
 *  whole read: ClassPro.c:229-271
 * ------------------------------------------------------------------------------------------ */
int cpo_classify_read(const cpo_params *p, const char *seq, int rlen, const uint16_t *profile,
                      char *labels, cpo_intvl *intvl_out, int cap, int *M_out)
{ const int K = p->K, plen = rlen-(K-1);
  for (int i = 0; i < K-1 && i < rlen; i++)
    labels[i] = 'N';
  if (rlen <= K-1)
    return 0;
  uint8_t *lctx = malloc((size_t)rlen*3), *rctx = malloc((size_t)rlen*3);
  int icap = plen+2;
  cpo_intvl *intvl  = malloc(sizeof(cpo_intvl)*icap);
  cpo_intvl *rintvl = malloc(sizeof(cpo_intvl)*icap);

  cpo_seq_context(seq,rlen,lctx,rctx);
  int N = cpo_find_wall(p,profile,plen,lctx,rctx,intvl,icap);
  if (N < 0)                                        /* "# E-intvls >= plen": the reference exits (wall.c:783-788) */
    { for (int j = 0; j < plen; j++)
        labels[K-1+j] = '!';
      free(lctx); free(rctx); free(intvl); free(rintvl);
      if (M_out) *M_out = 0;
      return -1;
    }
  int M = cpo_find_rel_intvl(p,intvl,N,rintvl,profile,plen,lctx,rctx);
  cpo_classify_rel(p,rintvl,M,intvl,N,plen,NULL,NULL);
  cpo_classify_unrel(p,intvl,N);
  char *pasgn = labels+(K-1);
  for (int i = 0; i < N; i++)
    { char c = STOC[(int)intvl[i].asgn];
      for (int j = intvl[i].b; j < intvl[i].e; j++)
        pasgn[j] = c;
    }
  if (intvl_out)
    memcpy(intvl_out,intvl,sizeof(cpo_intvl)*MINI(N,cap));
  if (M_out) *M_out = M;
  free(lctx); free(rctx); free(intvl); free(rintvl);
  return N;
}

typedef struct
  { const cpo_params *p;
    const char *seq; const int64_t *seq_off;
    const uint16_t *prof; const int64_t *prof_off;
    char *labels;
    int beg, end;
  } batch_arg;

static void *batch_thread(void *x)
{ batch_arg *a = x;
  for (int r = a->beg; r < a->end; r++)
    { int rlen = (int)(a->seq_off[r+1]-a->seq_off[r]);
      cpo_classify_read(a->p,a->seq+a->seq_off[r],rlen,a->prof+a->prof_off[r],
                        a->labels+a->seq_off[r],NULL,0,NULL);
    }
  return NULL;
}

/* Contiguous read ranges per thread, like ClassPro.c:530 / io.c:353-354. */
void cpo_classify_batch(const cpo_params *p, const char *seq, const int64_t *seq_off,
                        const uint16_t *prof, const int64_t *prof_off, int nreads,
                        char *labels, int nthreads)
{ if (nthreads < 1) nthreads = 1;
  /* keep the per-read scratch of every thread in its malloc arena (no mmap/munmap per read), the
     moral equivalent of the reference allocating its scratch once per thread (ClassPro.c:114-143) */
  mallopt(M_MMAP_THRESHOLD,1<<30);
  mallopt(M_TRIM_THRESHOLD,1<<30);
  int nparts = nreads/nthreads + (nreads%nthreads == 0 ? 0 : 1);
  pthread_t *th = malloc(sizeof(pthread_t)*nthreads);
  batch_arg *arg = malloc(sizeof(batch_arg)*nthreads);
  for (int t = 0; t < nthreads; t++)
    { batch_arg a = { p, seq, seq_off, prof, prof_off, labels, MINI(t*nparts,nreads), MINI((t+1)*nparts,nreads) };
      arg[t] = a;
      if (t > 0) pthread_create(&th[t],NULL,batch_thread,&arg[t]);
    }
  batch_thread(&arg[0]);
  for (int t = 1; t < nthreads; t++)
    pthread_join(th[t],NULL);
  free(th); free(arg);
}
