// cp_host_setup.h -- one-time global setup on the host (the reference does this once in main(),
// src/ClassPro.c:536-554): log-factorial table, global coverages, default error model and the
// count-threshold table.  Pure host code (glibc log/exp/sqrt, like the reference); the result is
// uploaded to HBM once and read by every kernel.
#pragma once
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "cp_types.h"

// hist.c:28-105 over the histogram as Load_Histogram + Modify_Histogram(H,low,high,0) leave it
// (libfastk.c:22-147: interior cells become instance counts, boundary cells swap with the two
// hidden cells).
static inline int cp_host_hist_covs(const int64_t *disk, int low, int high, int64_t ilowcnt, int64_t ihighcnt,
                                    int coverage_opt, int *hcov, int *dcov)
{ if (coverage_opt > 0)                                   // hist.c:44-50
    { *dcov = coverage_opt;
      *hcov = coverage_opt >> 1;
      return CP_OK;
    }
  if (low > 1 || high < 4 || high > (1 << 20))
    return CP_EINVAL;
  // The partner-peak windows below reach up to 2*maxcnt + sqrt(2*maxcnt) + 1 with maxcnt < min(1000,high): past the
  // histogram when `high` is small (FastK always writes 1..32767, where this cannot happen).  Cells past the two hidden
  // ones read as zero here; the reference reads whatever follows its array.
  const size_t reach = (size_t)(2*(high < 1000 ? high : 1000)+64+8);
  std::vector<int64_t> buf(std::max((size_t)(high-low)+3+4,reach-(size_t)low+8),0);
  int64_t *hist = buf.data()-low;
  memcpy(buf.data(),disk,sizeof(int64_t)*((size_t)(high-low)+1));
  for (int i = low+1; i < high; i++)                      // toggle to instance counts
    hist[i] *= i;
  hist[high+1] = hist[low];   hist[low]  = ilowcnt;
  hist[high+2] = hist[high];  hist[high] = ihighcnt;

  int maxcnt = 0;
  int64_t maxpk = 0;
  const int lo = low > 2 ? low : 2, hi = high < 1000 ? high : 1000;
  for (int i = lo; i < hi; i++)                           // tallest strict local maximum, hist.c:58-64
    if (hist[i-1] < hist[i] && hist[i] > hist[i+1] && maxpk < hist[i])
      { maxcnt = i;
        maxpk = hist[i];
      }
  if (maxcnt < 10)
    return CP_ENOPEAK;

  int lmaxcnt = 0, rmaxcnt = 0, is_lpeak = 0, is_rpeak = 0;
  int64_t lmaxpk = 0, rmaxpk = 0;
  double m = (double)maxcnt/2, s = sqrt(m);               // hist.c:75-85
  for (int i = (int)round(m-s); i <= (int)round(m+s); i++)
    if (lmaxpk < hist[i])
      { lmaxcnt = i; lmaxpk = hist[i];
        is_lpeak = (hist[i-1] < hist[i] && hist[i] > hist[i+1]);
      }
  m = (double)maxcnt*2; s = sqrt(m);                      // hist.c:87-97
  for (int i = (int)round(m-s); i <= (int)round(m+s); i++)
    if (rmaxpk < hist[i])
      { rmaxcnt = i; rmaxpk = hist[i];
        is_rpeak = (hist[i-1] < hist[i] && hist[i] > hist[i+1]);
      }
  if (lmaxpk > rmaxpk)                                    // hist.c:99-107
    { *dcov = maxcnt;
      *hcov = is_lpeak ? lmaxcnt : (maxcnt >> 1);
    }
  else
    { *hcov = maxcnt;
      *dcov = is_rpeak ? rmaxcnt : (maxcnt << 1);
    }
  return CP_OK;
}

// -M<model_path>: HIsim error model -> pe[t][l] (load_himodel, wall.c:55-115).  File layout: int kmer;
// 0x4000 heptamer records of 11 floats (skipped, as in the reference); then for unit length 1,2,3 a table of
// 4^ulen x (kmer/2-6) micro-satellite records of 7 floats, first float = overall error rate, row i =
// unit sequence, column c = run length 2*ulen+c.  For j = 2..5 unit copies y[j] is the mean positive
// rate over unit sequences; y[1] is fixed at 0.002; pe[t][l] = c0 + c1 l + c2 l^2 is the least-squares
// quadratic through (1..5, y).  The reference calls GSL's gsl_multifit_linear for the fit (wall.c:11-41);
// x = 1..5 is fixed, so the normal equations have the constant matrix below and are solved exactly in
// long double arithmetic here -- the same minimiser, no GSL.  (Parity of the fit itself is unpinned: the
// reference's wall.c cannot be built in this image and its tree holds no model file.)
static inline int cp_host_load_himodel(const char *path, double pe[3][21], char *msg, size_t msglen)
{ FILE *f = fopen(path,"rb");
  if (!f) { snprintf(msg,msglen,"Cannot open error model %s",path); return CP_EINVAL; }
  int kmer = 0;
  if (fread(&kmer,4,1,f) != 1) { fclose(f); snprintf(msg,msglen,"Error model %s is truncated",path); return CP_EINVAL; }
  const int krange = kmer/2-6;
  if (krange < 1 || krange > 4096 || fseek(f,44L*0x4000,SEEK_CUR) != 0)
    { fclose(f); snprintf(msg,msglen,"Error model %s: bad k-mer length %d",path,kmer); return CP_EINVAL; }
  for (int t = 0; t < 3; t++)
    { const int ulen = t+1, N = 1 << (2*ulen);
      std::vector<float> tab((size_t)7*N*krange);
      if (fread(tab.data(),28,(size_t)N*krange,f) != (size_t)N*krange)
        { fclose(f); snprintf(msg,msglen,"Error model %s is truncated",path); return CP_EINVAL; }
      double y[6];
      y[1] = 0.002;
      for (int j = 2; j <= 5; j++)
        { if ((j-2)*ulen >= krange)
            { fclose(f); snprintf(msg,msglen,"Error model %s: k-mer length %d too short for the fit",path,kmer); return CP_EINVAL; }
          double sum = 0.; int n = 0;
          for (int i = 0; i < N; i++)
            { double p = tab[(size_t)7*(krange*i+(j-2)*ulen)];
              if (p > 0.) { sum += p; n++; }
            }
          y[j] = sum/n;
        }
      // normal equations  (X^T X) c = X^T y  with X = [1 x x^2], x = 1..5
      long double M[3][4] = { { 5, 15, 55, 0 }, { 15, 55, 225, 0 }, { 55, 225, 979, 0 } };
      for (int j = 1; j <= 5; j++)
        { M[0][3] += (long double)y[j]; M[1][3] += (long double)y[j]*j; M[2][3] += (long double)y[j]*j*j; }
      for (int k = 0; k < 3; k++)
        for (int r = 0; r < 3; r++)
          if (r != k)
            { long double q = M[r][k]/M[k][k];
              for (int c = k; c < 4; c++) M[r][c] -= q*M[k][c];
            }
      const double c0 = (double)(M[0][3]/M[0][0]), c1 = (double)(M[1][3]/M[1][1]), c2 = (double)(M[2][3]/M[2][2]);
      pe[t][0] = 0.;
      for (int l = 1; l <= CP_MAX_N_LC/(t+1); l++)
        { pe[t][l] = c0+c1*l+c2*l*l;                          // wall.c:102-103
          if (!(pe[t][l] > 0. && pe[t][l] < 1.))
            { fclose(f); snprintf(msg,msglen,"Error model %s gives an error rate outside (0,1) (%g for unit %d x %d)",path,pe[t][l],ulen,l); return CP_EINVAL; }
        }
    }
  fclose(f);
  return CP_OK;
}

static inline int cp_host_fill_params(cp_dev_params *P, int K, int read_len, int hcov, int dcov,
                                      const double (*model_pe)[21] = nullptr)
{ memset(P,0,sizeof(*P));
  if (K < 2 || read_len < 1 || hcov < 1 || dcov < 1 || dcov > 65535)
    return CP_EINVAL;
  P->K = K;
  P->read_len = read_len;

  P->logfact[0] = 0.;                                     // prob.c:14-19: running sum, not lgamma
  P->logint[0] = log(0.0);
  for (int n = 1; n <= CP_MAX_KMER_CNT; n++)
    { P->logint[n]  = log((double)n);
      P->logfact[n] = P->logfact[n-1]+log(n);
    }

  P->cov[CP_HAPLO]  = hcov & 0xffff;                      // ClassPro.c:544-548, util.c:9-11
  P->cov[CP_DIPLO]  = dcov & 0xffff;
  P->cov[CP_ERROR]  = 1;
  P->cov[CP_REPEAT] = (P->cov[CP_DIPLO] + (int)(uint16_t)(sqrt((double)P->cov[CP_DIPLO]) * CP_N_SIGMA_RCOV)) & 0xffff;
  P->dr_ratio = 1.+(double)CP_N_SIGMA_R*(1./sqrt((double)P->cov[CP_DIPLO]));
  if (P->cov[CP_REPEAT] > 255)                            // wall.c:174-177
    return CP_ERCOV;
  P->cmax = P->cov[CP_REPEAT];

  for (int t = 0; t < 3; t++)                             // default error model, wall.c:119-143
    { P->lmax[t] = CP_MAX_N_LC/(t+1);
      P->pe[t][0] = 0.;
      P->lpe[t][0] = log(0.0);
      P->l1mpe[t][0] = log(1.0);
      for (int l = 1; l <= P->lmax[t]; l++)
        { double pe = model_pe ? model_pe[t][l] : 0.002 * l * l + 0.002;
          P->pe[t][l]    = pe;
          P->lpe[t][l]   = log(pe);
          P->l1mpe[t][l] = log(1-pe);
        }
    }
  P->hc_erate = P->pe[CP_HP][1];                          // wall.c:180
  P->hc_lpe   = log(P->hc_erate);
  P->hc_l1mpe = log(1-P->hc_erate);
  { double u = 0.1;                                       // class_unrel.c:137,155
    P->u_lpe = log(u);
    P->u_l1mpe = log(1-u);
    double pr = 1-CP_PE_MEAN;                             // class_rel.c:185, class_unrel.c:104-105
    P->r_lp = log(pr);
    P->r_l1mp = log(1-pr);
    P->log_pe_final = log(CP_PE_THRES_FINAL);             // wall.c:1018
  }

  const double thres[2][2] = { {CP_PE_THRES_INIT_S, CP_PE_THRES_INIT_O}, {CP_PE_THRES_FINAL, CP_PE_THRES_FINAL} };
  for (int t = 0; t < 3; t++)                             // wall.c:190-224
    for (int l = 1; l <= P->lmax[t]; l++)
      { const double lpe = P->lpe[t][l], l1mpe = P->l1mpe[t][l];
        for (int cout = 1; cout < P->cmax; cout++)
          { bool found[2][2] = { {false,false}, {false,false} };
            for (int s = 0; s < 2; s++)
              { P->cthres[t][l][cout][s][CP_SELF]   = (uint8_t)cout;
                P->cthres[t][l][cout][s][CP_OTHERS] = 0;
              }
            double psum = 1.;
            for (int cin = 0; cin <= cout; cin++)
              { if (found[0][0] && found[1][0] && found[0][1] && found[1][1])
                  break;
                const double *lf = P->logfact;
                psum -= exp(lf[cout] - lf[cin] - lf[cout-cin] + cin * lpe + (cout-cin) * l1mpe);
                for (int s = 0; s < 2; s++)
                  for (int e = 0; e < 2; e++)
                    if (!found[s][e] && psum < thres[s][e])
                      { P->cthres[t][l][cout][s][e] = (uint8_t)(e == CP_SELF ? cin : cout-cin);
                        found[s][e] = true;
                      }
              }
          }
      }
  return CP_OK;
}

// libfastk.c:1467-1534 without the file-buffer refills.
static inline int cp_host_decode_profile(const uint8_t *code, int64_t len, uint16_t *profile, int cap)
{ if (len <= 0)
    return 0;
  const uint8_t *c = code, *q = code+len;
  uint16_t x = *c++, d;
  if (x & 0x80) d = (uint16_t)(((x & 0x7f) << 8) | *c++);
  else          d = x;
  int n = 1;
  bool writing = cap > 0;                                 // the reference stops storing at the first item that
  if (writing)                                            // does not fit and only counts from there on
    profile[0] = d;
  while (c < q)
    { x = *c++;
      if ((x & 0xc0) == 0)                                // 00rrrrrr: run of the current count
        { if (writing && n+x > cap)
            writing = false;
          if (writing)
            for (int i = 0; i < x; i++)
              profile[n+i] = d;
          n += x;
        }
      else
        { if (x & 0x80)                                   // 1sxxxxxx yyyyyyyy: 15-bit delta
            { x = (x & 0x40) ? (uint16_t)(x << 8) : (uint16_t)((x << 8) & 0x7fff);
              x |= *c++;
              d = (uint16_t)((d+x) & 0x7fff);
            }
          else if (x & 0x20)                              // 011xxxxx: negative 6-bit delta
            d = (uint16_t)(d + ((x & 0x1fu) | 0xffe0u));
          else                                            // 010xxxxx
            d = (uint16_t)(d + (x & 0x1fu));
          if (writing && n >= cap)
            writing = false;
          if (writing)
            profile[n] = d;
          n++;
        }
    }
  return n;
}

// Inverse of the decoder above (what FastK writes per read; tooling for tests, benchmarks and the
// synthetic-data path -- ClassPro itself only ever decodes).  Returns the number of code bytes;
// `code` must hold at least 2*n+2 bytes.
static inline int64_t cp_host_encode_profile(const uint16_t *cnt, int n, uint8_t *code)
{ if (n <= 0) return 0;
  uint8_t *o = code;
  int d = cnt[0] & 0x7fff;
  if (d < 128) *o++ = (uint8_t)d;
  else { *o++ = (uint8_t)(0x80 | (d >> 8)); *o++ = (uint8_t)(d & 0xff); }
  int i = 1;
  while (i < n)
    { int c = cnt[i] & 0x7fff;
      if (c == d)
        { int run = 1;
          while (i+run < n && (cnt[i+run] & 0x7fff) == d && run < 63) run++;
          *o++ = (uint8_t)run;
          i += run;
        }
      else
        { int dl = c-d;
          if (dl >= 1 && dl <= 31)        *o++ = (uint8_t)(0x40 | dl);
          else if (dl >= -32 && dl <= -1) *o++ = (uint8_t)(0x60 | (dl+32));
          else
            { int x = dl & 0x7fff;
              *o++ = (uint8_t)(0x80 | (x >> 8));
              *o++ = (uint8_t)(x & 0xff);
            }
          d = c;
          i++;
        }
    }
  return (int64_t)(o-code);
}
