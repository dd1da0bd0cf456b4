// capi.hip -- host side of the C ABI (include/classpro_amd.h): parameter upload, device scratch
// management and kernel launches.  No torch, no CPU compute path: every stage is a HIP kernel and
// every entry point fails with CP_EHIP when the device call fails.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <mutex>
#include <vector>
#include <chrono>
#include <thread>
#include <time.h>
#include "cp_host_setup.h"
#include "kernels.hip"

static thread_local std::string g_err;
static int set_err(int code, const std::string &msg) { g_err = msg; return code; }

#define HIPCHK(call)                                                                       \
  do { hipError_t e_ = (call);                                                             \
       if (e_ != hipSuccess)                                                               \
         return set_err(CP_EHIP,std::string(#call)+": "+hipGetErrorString(e_));            \
     } while (0)

extern "C" const char *cp_last_error(void) { return g_err.c_str(); }
extern "C" const char *cp_version(void) { return "classpro_amd 0.1 (gfx950)"; }

// ---------------------------------------------------------------------------------------------
struct cp_params
  { cp_dev_params  host;
    cp_dev_params *dev;
    int            device;                 // HIP device the tables live on
    double        *skel;                   // device table of logp_trans values (cp_types.h), or NULL
    double        *eskel;                  // exp() of them, or NULL
    double        *uerr;                   // device table of classify_unrel's binomial-test logs, or NULL
    double        *petab;                  // device table of the walk's P(error in) values, or NULL
    size_t         skel_bytes, uerr_bytes, petab_bytes;
  };

// ---------------------------------------------------------------------------------------------
//  The three read-only tables are pure functions of a few setup values (logp_trans: READ_LEN; the unrel binomial
//  term: nothing; P(error in): the error model), so every cp_params of a process that agrees on those shares ONE copy
//  per device (the command line creates one cp_params per device shard, CLASSPRO_DEVICES=0,0,0,0 four on one card:
//  1 GB instead of 4).  Reference-counted; filled on a private non-blocking stream, so that creating a cp_params does
//  not wait for -- or stall -- kernels other threads have in flight on the device's blocking streams.
// ---------------------------------------------------------------------------------------------
struct cp_tab_entry { int dev; std::string key; double *p; size_t bytes; int refs; };
static std::mutex g_tab_mu;
static std::vector<cp_tab_entry> g_tabs;

template <class Fill>
static double *tab_acquire(int dev, const std::string &key, size_t bytes, Fill fill)
{ std::lock_guard<std::mutex> lk(g_tab_mu);
  for (cp_tab_entry &t : g_tabs)
    if (t.dev == dev && t.key == key && t.bytes == bytes) { t.refs++; return t.p; }
  double *p = NULL;
  if (hipMalloc((void **)&p,bytes) != hipSuccess) { (void)hipGetLastError(); return NULL; }   // no room: computed on the spot
  hipStream_t st = NULL;
  hipError_t e = hipStreamCreateWithFlags(&st,hipStreamNonBlocking);
  if (e == hipSuccess)
    { (void)hipGetLastError();                    // (when this is the process's first HIP work, the runtime's own start-up
                                                  //  probing can leave an error code behind: it is not this launch's)
      fill(p,st);
      e = hipGetLastError();
      if (e == hipSuccess) e = hipStreamSynchronize(st);
      (void)hipStreamDestroy(st);
    }
  if (e != hipSuccess) { (void)hipFree(p); (void)hipGetLastError(); return NULL; }
  g_tabs.push_back(cp_tab_entry{dev,key,p,bytes,1});
  return p;
}

static void tab_release(double *p)
{ if (!p) return;
  std::lock_guard<std::mutex> lk(g_tab_mu);
  for (size_t i = 0; i < g_tabs.size(); i++)
    if (g_tabs[i].p == p)
      { if (--g_tabs[i].refs == 0)
          { (void)hipFree(p);
            g_tabs.erase(g_tabs.begin()+(long)i);
          }
        return;
      }
}

extern "C" int cp_hist_covs(const int64_t *hist, int low, int high, int64_t ilowcnt, int64_t ihighcnt,
                            int coverage_opt, int *hcov, int *dcov)
{ if (!hcov || !dcov || (coverage_opt <= 0 && !hist))
    return set_err(CP_EINVAL,"cp_hist_covs: null argument");
  int rc = cp_host_hist_covs(hist,low,high,ilowcnt,ihighcnt,coverage_opt,hcov,dcov);
  if (rc == CP_ENOPEAK)
    return set_err(rc,"[ERROR] Could not find any peak count >= 10 in the histogram. Revise data and use the `-c` option.");
  if (rc != CP_OK)
    return set_err(rc,"cp_hist_covs: unsupported histogram range");
  return CP_OK;
}

extern "C" int cp_params_create(int K, int read_len, int hcov, int dcov, cp_params **out)
{ return cp_params_create_model(K,read_len,hcov,dcov,NULL,out); }

extern "C" int cp_load_error_model(const char *model_path, double *pe63)
{ if (!model_path || !pe63) return set_err(CP_EINVAL,"cp_load_error_model: null argument");
  char msg[512];
  double pe[3][21];
  memset(pe,0,sizeof(pe));
  int rc = cp_host_load_himodel(model_path,pe,msg,sizeof(msg));
  if (rc != CP_OK) return set_err(rc,msg);
  memcpy(pe63,pe,sizeof(pe));
  return CP_OK;
}

static int params_create(int K, int read_len, int hcov, int dcov, const double (*pe)[21], cp_params **out);

extern "C" int cp_params_create_model(int K, int read_len, int hcov, int dcov, const char *model_path, cp_params **out)
{ if (!out) return set_err(CP_EINVAL,"cp_params_create: null out");
  double pe[3][21];
  if (model_path)
    { char msg[512];
      memset(pe,0,sizeof(pe));
      int rc = cp_host_load_himodel(model_path,pe,msg,sizeof(msg));
      if (rc != CP_OK) return set_err(rc,msg);
    }
  return params_create(K,read_len,hcov,dcov,model_path ? pe : nullptr,out);
}

// The error model as a table: pe63 = double[3][21], rows HP / DS / TS, pe[t][l] for l = 1 .. MAX_N_LC/(t+1) (entry 0 and
// the entries beyond are ignored).  For a caller that has the rates already -- the reference's own load_himodel with its GSL
// fit, or cp_load_error_model -- and for tests that hand the reference's find_wall and this library the same table.
extern "C" int cp_params_create_pe(int K, int read_len, int hcov, int dcov, const double *pe63, cp_params **out)
{ if (!out || !pe63) return set_err(CP_EINVAL,"cp_params_create_pe: null argument");
  double pe[3][21];
  memcpy(pe,pe63,sizeof(pe));
  for (int t = 0; t < 3; t++)
    for (int l = 1; l <= 20/(t+1); l++)
      if (!(pe[t][l] > 0.) || !(pe[t][l] < 1.))
        return set_err(CP_EINVAL,"cp_params_create_pe: error rates must lie in (0,1)");
  return params_create(K,read_len,hcov,dcov,pe,out);
}

static int params_create(int K, int read_len, int hcov, int dcov, const double (*pe)[21], cp_params **out)
{ cp_params *p = (cp_params *)malloc(sizeof(cp_params));
  if (!p) return set_err(CP_ENOMEM,"cp_params_create: out of memory");
  int rc = cp_host_fill_params(&p->host,K,read_len,hcov,dcov,pe);
  if (rc != CP_OK)
    { char buf[128];
      if (rc == CP_ERCOV) snprintf(buf,sizeof(buf),"Too high REPEAT coverage (%d) > 255",p->host.cov[CP_REPEAT]);
      else                snprintf(buf,sizeof(buf),"cp_params_create: invalid argument");
      free(p);
      return set_err(rc,buf);
    }
  p->dev = NULL; p->skel = p->eskel = p->uerr = p->petab = NULL; p->skel_bytes = p->uerr_bytes = p->petab_bytes = 0;
  p->device = 0;
  hipStream_t st = NULL;
  hipError_t e = hipGetDevice(&p->device);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&st,hipStreamNonBlocking);
  if (e == hipSuccess) e = hipMalloc((void **)&p->dev,sizeof(cp_dev_params));
  if (e == hipSuccess) e = hipMemcpyAsync(p->dev,&p->host,sizeof(cp_dev_params),hipMemcpyHostToDevice,st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess)
    { if (p->dev) (void)hipFree(p->dev);
      if (st) (void)hipStreamDestroy(st);
      free(p);
      return set_err(CP_EHIP,std::string("cp_params_create: ")+hipGetErrorString(e));
    }
  // The tables (cp_types.h).  CLASSPRO_TABLES=0: none of them; CLASSPRO_SKELLAM_TABLE_MB=<n>: size of the logp_trans
  // table (default 1024 MB = 2^19 products cov*|e-b|; 0 = no such table, the two small ones stay).  A table that
  // finds no memory is left out -- every value is then computed on the spot, same bits -- and cp_params_tables() says so.
  bool tables = true;
  if (const char *t = getenv("CLASSPRO_TABLES")) tables = atol(t) != 0;
  if (tables)
    { const cp_dev_params *dP = p->dev;
      // |ce-cb| <= 255 covers every pair of counts of reliable intervals (both below the REPEAT coverage <= 255)
      long mb = 1024;
      if (const char *m = getenv("CLASSPRO_SKELLAM_TABLE_MB")) mb = atol(m);
      const int kmax = 255;
      const long long cdmax = (long long)mb*1024*1024/8/(kmax+1)-1;
      if (cdmax >= 0)
        { const size_t bytes = (size_t)(cdmax+1)*(kmax+1)*8;
          p->skel = tab_acquire(p->device,"skel r"+std::to_string(read_len),bytes,[&](double *t, hipStream_t s)
            { hipLaunchKernelGGL(k_skellam_table,dim3(4096),dim3(256),0,s,dP,t,kmax,cdmax); });
          if (p->skel) { p->host.skel = p->skel; p->host.skel_kmax = kmax; p->host.skel_cdmax = cdmax; p->skel_bytes = bytes; }
          // exp() of the same entries for the DP step of classify_rel (CLASSPRO_EXP_TABLE=0: none; cp_exp_logp_trans, cp_math.h)
          const char *xe = getenv("CLASSPRO_EXP_TABLE");
          if (p->skel && (!xe || atol(xe) != 0))
            { const double *src = p->skel;
              const long long n = (cdmax+1)*(long long)(kmax+1);
              p->eskel = tab_acquire(p->device,"eskel r"+std::to_string(read_len)+" "+std::to_string(bytes),bytes,[&](double *t, hipStream_t s)
                { hipLaunchKernelGGL(k_eskel_table,dim3(4096),dim3(256),0,s,src,t,n); });
              if (p->eskel) p->host.eskel = p->eskel;
            }
        }
      const int emax = 1023;                                   // 1024 x 1024 doubles
      { const size_t bytes = (size_t)(emax+1)*(emax+1)*8;
        p->uerr = tab_acquire(p->device,"uerr",bytes,[&](double *t, hipStream_t s)
          { hipLaunchKernelGGL(k_uerr_table,dim3(1024),dim3(256),0,s,dP,t,emax); });
        if (p->uerr) { p->host.uerr = p->uerr; p->host.uerr_max = emax; p->uerr_bytes = bytes; }
      }
      const int cmax = 255;                                    // 2 error types x 63 error rates x counts <= 255: 66 MB
      { const size_t bytes = (size_t)2*63*(cmax+1)*(cmax+1)*8;
        std::string key("petab ");
        key.append((const char *)p->host.pe,sizeof(p->host.pe));      // the error model decides the values
        p->petab = tab_acquire(p->device,key,bytes,[&](double *t, hipStream_t s)
          { hipLaunchKernelGGL(k_pe_table,dim3(4096),dim3(256),0,s,dP,t,cmax); });
        if (p->petab) { p->host.petab = p->petab; p->host.pe_cmax = cmax; p->petab_bytes = bytes; }
      }
      e = hipMemcpyAsync(p->dev,&p->host,sizeof(cp_dev_params),hipMemcpyHostToDevice,st);
      if (e == hipSuccess) e = hipStreamSynchronize(st);
      if (e != hipSuccess)
        { tab_release(p->skel); tab_release(p->eskel); tab_release(p->uerr); tab_release(p->petab);
          (void)hipFree(p->dev); (void)hipStreamDestroy(st); free(p);
          return set_err(CP_EHIP,std::string("cp_params_create: ")+hipGetErrorString(e));
        }
    }
  (void)hipStreamDestroy(st);
  *out = p;
  return CP_OK;
}

extern "C" void cp_params_destroy(cp_params *p)
{ if (!p) return;
  tab_release(p->skel); tab_release(p->eskel); tab_release(p->uerr); tab_release(p->petab);
  if (p->dev) (void)hipFree(p->dev);
  free(p);
}

extern "C" int cp_params_tables(const cp_params *p, size_t *skel_bytes, size_t *uerr_bytes, size_t *petab_bytes)
{ if (!p) return set_err(CP_EINVAL,"cp_params_tables: null params");
  if (skel_bytes) *skel_bytes = p->skel_bytes+(p->eskel ? p->skel_bytes : 0);   // logp_trans and, same shape, exp() of it
  if (uerr_bytes) *uerr_bytes = p->uerr_bytes;
  if (petab_bytes) *petab_bytes = p->petab_bytes;
  return CP_OK;
}

__global__ void k_math_eval(int fn, const double *__restrict__ x, const double *__restrict__ x2, double *__restrict__ y, int64_t n)
{ for (int64_t i = blockIdx.x*(int64_t)blockDim.x+threadIdx.x; i < n; i += (int64_t)gridDim.x*blockDim.x)
    { const double a = x[i];
      double r;
      switch (fn)
        { case 0:  r = cp_exp(a); break;
          case 1:  r = cp_log(a); break;
          case 2:  r = sqrt(a); break;
          case 3:  r = cp_bessi((int)x2[i],a); break;
          default: r = cp_logp_skellam((int)x2[i],a); break;
        }
      y[i] = r;
    }
}

extern "C" int cp_math_eval(int fn, const double *d_x, const double *d_x2, double *d_y, int64_t n, void *stream)
{ if (fn < 0 || fn > 4 || n < 0 || (n > 0 && (!d_x || !d_y)) || (fn >= 3 && n > 0 && !d_x2))
    return set_err(CP_EINVAL,"cp_math_eval: bad argument");
  if (n == 0) return CP_OK;
  const int64_t blocks = (n+255)/256;
  hipLaunchKernelGGL(k_math_eval,dim3((unsigned)(blocks < 4096 ? blocks : 4096)),dim3(256),0,(hipStream_t)stream,fn,d_x,d_x2,d_y,n);
  HIPCHK(hipGetLastError());
  return CP_OK;
}

extern "C" int cp_params_export(const cp_params *p, int *cov4, double *dr_ratio, int *cmax, double *hc_erate,
                                uint8_t *cthres, double *pe, double *logfact)
{ if (!p) return set_err(CP_EINVAL,"cp_params_export: null params");
  if (cov4) memcpy(cov4,p->host.cov,sizeof(int)*4);
  if (dr_ratio) *dr_ratio = p->host.dr_ratio;
  if (cmax) *cmax = p->host.cmax;
  if (hc_erate) *hc_erate = p->host.hc_erate;
  if (cthres) memcpy(cthres,p->host.cthres,sizeof(p->host.cthres));
  if (pe) memcpy(pe,p->host.pe,sizeof(p->host.pe));
  if (logfact) memcpy(logfact,p->host.logfact,sizeof(p->host.logfact));
  return CP_OK;
}

extern "C" int cp_decode_profile(const uint8_t *code, int64_t len, uint16_t *profile, int cap)
{ if (len < 0 || (len > 0 && !code) || (cap > 0 && !profile))
    return set_err(CP_EINVAL,"cp_decode_profile: bad argument");
  return cp_host_decode_profile(code,len,profile,cap);
}

// ---------------------------------------------------------------------------------------------
//  Workspace: device scratch grown on demand (never shrunk), reused across batches.
// ---------------------------------------------------------------------------------------------
struct dbuf { void *p; size_t cap; };

struct cp_workspace
  { dbuf bitmap, ncand, nintvl, nrel, ioff, eoff, hoff, wall, wall_s, hkeys, hvals, eintvl, ointvl, intvl, rintvl,
         rrec, pcls, ivA, ivB, ivC, relmap, parent, eff, rpos, asgn, ord, err, perm, wlist, err2, tres, fwc, dtot,
         s_cap, s_rcap, s_dummy, s_key, s_seg, s_aux, s_mi, s_rep, s_repcnt, scan_state, order_tmp;
    int      scan_epoch;      // tag of the next k_prefix_caps_mb launch (its state array is never cleared)
    int64_t *h_totals;        // pinned: [totalI, totalE, totalH]
    unsigned long long *h_tot64, *d_tot64;   // pinned, and its device address: the totals as {value << 24 | launch tag}, written by the kernel itself
    int32_t *h_err;           // pinned
    // shape of the last run
    int      nreads;
    int64_t  total_kmers, total_bases, totalI, totalE, totalH, nwords;
    int      last_stage;
    int      last_compact;    // the last run was a whole-path call with compact records: classes in `pcls`, no `rintvl` (cp_run_stages)
    int      decode_pending;  // a cp_decode_profiles result has not been checked yet
    int      seed_nreads; int64_t seed_totalR;   // shape of the last cp_find_seeds_batch
    std::vector<void *> *retired;   // buffers that were outgrown while a stream could still be using them (ensure())
    hipStream_t stream;
    hipStream_t aux;          // size classes of one stage run side by side: the rare long reads are latency-bound
    hipEvent_t  ev_fork, ev_join;
  };

// Grow-only.  A buffer that is outgrown may still be read by kernels of the previous call on the workspace's stream (the
// entry points are asynchronous), and hipFree synchronises the whole device: the old buffer is parked on the
// workspace's retired list and freed where the stream is waited for anyway (cp_workspace_check, cp_workspace_destroy).
// A quarter of head-room, so that a sequence of sub-batches of about one size allocates once.
static int ensure(cp_workspace *ws, dbuf &b, size_t need)
{ if (need <= b.cap) return CP_OK;
  size_t want = need+need/4+256;
  void *np = NULL;
  hipError_t e = hipMalloc(&np,want);
  if (e != hipSuccess && ws->retired && !ws->retired->empty())
    { (void)hipGetLastError();                     // out of memory with parked buffers: give them back and try again
      if (ws->stream) (void)hipStreamSynchronize(ws->stream);
      for (void *q : *ws->retired) (void)hipFree(q);
      ws->retired->clear();
      e = hipMalloc(&np,want);
    }
  if (e != hipSuccess)
    { char m[160];
      snprintf(m,sizeof(m),"hipMalloc(%zu bytes): %s",want,hipGetErrorString(e));
      (void)hipGetLastError();
      return set_err(CP_ENOMEM,m);
    }
  if (b.p) ws->retired->push_back(b.p);
  b.p = np; b.cap = want;
  return CP_OK;
}
#define ENSURE(b,n) do { int rc_ = ensure(ws,b,(size_t)(n)); if (rc_ != CP_OK) return rc_; } while (0)

extern "C" int cp_workspace_create(cp_workspace **out)
{ if (!out) return set_err(CP_EINVAL,"cp_workspace_create: null out");
  cp_workspace *ws = (cp_workspace *)calloc(1,sizeof(cp_workspace));
  if (!ws) return set_err(CP_ENOMEM,"cp_workspace_create: out of memory");
  ws->retired = new std::vector<void *>();
  hipError_t e = hipHostMalloc((void **)&ws->h_totals,4*sizeof(int64_t),hipHostMallocDefault);
  if (e == hipSuccess) e = hipHostMalloc((void **)&ws->h_tot64,4*sizeof(unsigned long long),hipHostMallocDefault);
  if (e == hipSuccess) { memset(ws->h_tot64,0,4*sizeof(unsigned long long)); e = hipHostGetDevicePointer((void **)&ws->d_tot64,ws->h_tot64,0); }
  if (e == hipSuccess) e = hipHostMalloc((void **)&ws->h_err,4*sizeof(int32_t),hipHostMallocDefault);
  // the error words (device): [0] find_wall / classification, [1] find_seeds, [2] profile decode.  Sticky: kernels OR
  // into them, only cp_workspace_check reads and clears them, so a flag raised by any call since the last check is seen
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&ws->aux,hipStreamNonBlocking);
  if (e == hipSuccess) e = hipMalloc(&ws->err.p,16);
  if (e == hipSuccess) { ws->err.cap = 16; e = hipMemsetAsync(ws->err.p,0,16,ws->aux); }
  if (e == hipSuccess) e = hipStreamSynchronize(ws->aux);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&ws->ev_fork,hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&ws->ev_join,hipEventDisableTiming);
  if (e != hipSuccess)
    { if (ws->err.p) (void)hipFree(ws->err.p);
      delete ws->retired;
      free(ws);
      return set_err(CP_EHIP,std::string("cp_workspace_create: ")+hipGetErrorString(e));
    }
  *out = ws;
  return CP_OK;
}

extern "C" void cp_workspace_destroy(cp_workspace *ws)
{ if (!ws) return;
  dbuf *all[] = { &ws->bitmap,&ws->ncand,&ws->nintvl,&ws->nrel,&ws->ioff,&ws->eoff,&ws->hoff,&ws->wall,&ws->wall_s,&ws->hkeys,&ws->hvals,
                  &ws->eintvl,&ws->ointvl,&ws->intvl,&ws->rintvl,&ws->rrec,&ws->pcls,&ws->ivA,&ws->ivB,&ws->ivC,&ws->relmap,&ws->parent,&ws->eff,&ws->rpos,
                  &ws->asgn,&ws->ord,&ws->err,&ws->perm,&ws->wlist,&ws->err2,&ws->tres,&ws->fwc,&ws->dtot,
                  &ws->s_cap,&ws->s_rcap,&ws->s_dummy,&ws->s_key,&ws->s_seg,&ws->s_aux,&ws->s_mi,&ws->s_rep,&ws->s_repcnt,&ws->scan_state,&ws->order_tmp };
  for (dbuf *b : all) if (b->p) (void)hipFree(b->p);
  for (void *q : *ws->retired) (void)hipFree(q);
  delete ws->retired;
  if (ws->aux) (void)hipStreamDestroy(ws->aux);
  if (ws->ev_fork) (void)hipEventDestroy(ws->ev_fork);
  if (ws->ev_join) (void)hipEventDestroy(ws->ev_join);
  if (ws->h_totals) (void)hipHostFree(ws->h_totals);
  if (ws->h_tot64) (void)hipHostFree(ws->h_tot64);
  if (ws->h_err) (void)hipHostFree(ws->h_err);
  free(ws);
}

extern "C" size_t cp_workspace_bytes(const cp_workspace *ws)
{ if (!ws) return 0;
  const dbuf *all[] = { &ws->bitmap,&ws->ncand,&ws->nintvl,&ws->nrel,&ws->ioff,&ws->eoff,&ws->hoff,&ws->wall,&ws->wall_s,&ws->hkeys,&ws->hvals,
                        &ws->eintvl,&ws->ointvl,&ws->intvl,&ws->rintvl,&ws->rrec,&ws->pcls,&ws->ivA,&ws->ivB,&ws->ivC,&ws->relmap,&ws->parent,&ws->eff,&ws->rpos,
                        &ws->asgn,&ws->ord,&ws->err,&ws->perm,&ws->wlist,&ws->err2,&ws->tres,&ws->fwc,&ws->dtot,
                  &ws->s_cap,&ws->s_rcap,&ws->s_dummy,&ws->s_key,&ws->s_seg,&ws->s_aux,&ws->s_mi,&ws->s_rep,&ws->s_repcnt,&ws->scan_state,&ws->order_tmp };
  size_t s = 0;
  for (const dbuf *b : all) s += b->cap;
  return s;
}

// ---------------------------------------------------------------------------------------------
static int scan_grid(int64_t total)
{ int64_t groups = total >> 3;
  int64_t blocks = (groups+256*SCAN_UNROLL-1)/(256*SCAN_UNROLL);
  if (blocks < 1) blocks = 1;
  if (blocks > 256*SCAN_BLOCKS_PER_CU) blocks = 256*SCAN_BLOCKS_PER_CU;   // grid-stride beyond that
  return (int)blocks;
}

static int launch_scan(const cp_dev_params *hostP, const uint16_t *d_prof, int64_t total, uint64_t *d_bitmap,
                       int64_t nwords, hipStream_t st)
{ if (((uintptr_t)d_prof) & 15)
    return set_err(CP_EINVAL,"profile buffer must be 16-byte aligned");
  // A batch is scanned in launches of at most 2^31 positions (4 GB of counts): 8-GB launches run 3-5 % slower per byte
  // (1581-1630 us against 2 x 768-770 us, A/B in one call, round 5).  CLASSPRO_SCAN_CHUNK_KMERS: another size (tests).
  int64_t chunk = (int64_t)1 << 31;
  if (const char *e = getenv("CLASSPRO_SCAN_CHUNK_KMERS")) { const int64_t v = atoll(e); if (v >= 4096) chunk = v & ~(int64_t)4095; }
  uint8_t *bm = (uint8_t *)d_bitmap;
  for (int64_t p0 = 0; p0 < total || p0 == 0; p0 += chunk)
    { const bool last = p0+chunk >= total;
      const int64_t n = last ? total-p0 : chunk;            // (a whole number of bitmap bytes and of 16-byte loads unless last)
      hipLaunchKernelGGL(k_scan_candidates,dim3(scan_grid(n)),dim3(256),0,st,
                         d_prof+p0,n,hostP->cov[CP_REPEAT],bm+(p0 >> 3),last ? nwords*8-(p0 >> 3) : (n >> 3),p0 > 0 ? 1 : 0);
      if (last) break;
    }
  HIPCHK(hipGetLastError());
  return CP_OK;
}

extern "C" int cp_scan_candidates(const cp_params *p, const uint16_t *d_prof, int64_t total_kmers,
                                  uint64_t *d_bitmap, void *stream)
{ if (!p || !d_prof || !d_bitmap || total_kmers < 0)
    return set_err(CP_EINVAL,"cp_scan_candidates: bad argument");
  return launch_scan(&p->host,d_prof,total_kmers,d_bitmap,total_kmers/64+1,(hipStream_t)stream);
}

// exclusive prefix sums of the three capacity arrays (+ totals), one block per 1024 reads
static int launch_prefix_caps(cp_workspace *ws, int64_t *a, int64_t *b, int64_t *c, int n, int64_t *totals, hipStream_t st,
                              bool to_host = false)
{ const int tiles = (n+SCAN_TILE-1)/SCAN_TILE;
  const size_t c0 = ws->scan_state.cap;
  ENSURE(ws->scan_state,(size_t)tiles*sizeof(cp_scan_state));
  // The launch tag only ever grows (1 .. 2^24-1): a freshly zeroed array (tag 0) is valid for any launch, so growing it
  // does NOT start the count over -- the pinned words the host polls still hold the tags of earlier launches, and a
  // second "launch 1" would match them at once and hand the previous batch's totals to this one.
  if (ws->scan_state.cap != c0)
    HIPCHK(hipMemsetAsync(ws->scan_state.p,0,ws->scan_state.cap,st));
  if (++ws->scan_epoch > (int)SCAN_TAG_MASK)              // the tags start over: from a cleared array, and with the polled
    { HIPCHK(hipMemsetAsync(ws->scan_state.p,0,ws->scan_state.cap,st)); ws->scan_epoch = 1; }   // words invalidated below
  if (to_host)                                            // no earlier launch's totals can pass for this one's
    for (int q = 0; q < 3; q++) __atomic_store_n(&ws->h_tot64[q],0ull,__ATOMIC_RELEASE);
  hipLaunchKernelGGL(k_prefix_caps_mb,dim3(tiles),dim3(WAVE),0,st,a,b,c,n,totals,(cp_scan_state *)ws->scan_state.p,ws->scan_epoch,
                     to_host ? ws->d_tot64 : (unsigned long long *)NULL);
  HIPCHK(hipGetLastError());
  return CP_OK;
}

// perm[] = read ids by decreasing key >> shift (kernels.hip: k_order_hist / k_order_scatter)
static int launch_order_by_work(cp_workspace *ws, const int32_t *key, int n, int shift, int32_t *perm, hipStream_t st,
                                const int64_t *long_off = nullptr)
{ const size_t c0 = ws->order_tmp.cap;
  ENSURE(ws->order_tmp,(size_t)(2*ORDER_BINS+1)*4);
  if (ws->order_tmp.cap != c0) HIPCHK(hipMemsetAsync(ws->order_tmp.p,0,ws->order_tmp.cap,st));   // zero once: the scatter kernel's last block leaves it zero
  const int blocks = (n+ORDER_TILE-1)/ORDER_TILE;
  int32_t *ghist = (int32_t *)ws->order_tmp.p, *gcur = ghist+ORDER_BINS;
  hipLaunchKernelGGL(k_order_hist,dim3(blocks),dim3(WAVE),0,st,key,n,shift,ghist,long_off);
  hipLaunchKernelGGL(k_order_scatter,dim3(blocks),dim3(WAVE),0,st,key,n,shift,ghist,gcur,perm,long_off);
  HIPCHK(hipGetLastError());
  return CP_OK;
}

extern "C" int cp_run_stages(const cp_params *p, cp_workspace *ws,
                             const char *d_seq, const int64_t *d_seq_off,
                             const uint16_t *d_prof, const int64_t *d_prof_off,
                             int nreads, int64_t total_bases, int64_t total_kmers,
                             char *d_labels, int last_stage, void *stream)
{ if (!p || !ws || nreads < 0 || total_bases < 0 || total_kmers < 0)
    return set_err(CP_EINVAL,"cp_run_stages: bad argument");
  if (nreads > 0 && (!d_seq || !d_seq_off || !d_prof || !d_prof_off))
    return set_err(CP_EINVAL,"cp_run_stages: null device pointer");
  if (last_stage < CP_STAGE_SCAN || last_stage > CP_STAGE_LABELS)
    return set_err(CP_EINVAL,"cp_run_stages: bad stage");
  if (last_stage == CP_STAGE_LABELS && nreads > 0 && !d_labels)
    return set_err(CP_EINVAL,"cp_run_stages: labels buffer required");
  // capacities per read are at most 16 per k-mer position + 64 (k_count_caps); their prefix sums travel in 40 bits
  // (k_prefix_caps_mb): a batch that could overflow them is refused here, not found out by a wrong offset
  if (total_kmers > CP_MAX_BATCH_KMERS || 16*total_kmers+64*(int64_t)nreads >= ((int64_t)1 << 40))
    return set_err(CP_EINVAL,"cp_run_stages: more than CP_MAX_BATCH_KMERS k-mer positions in one batch: split it");
  hipStream_t st = (hipStream_t)stream;
  ws->nreads = nreads; ws->total_kmers = total_kmers; ws->total_bases = total_bases;
  ws->totalI = ws->totalE = 0; ws->last_stage = last_stage; ws->stream = st; ws->last_compact = 0;
  ws->nwords = total_kmers/64+2;
  if (nreads == 0)
    return CP_OK;

  // ---- stage 1: candidate scan + capacities ------------------------------------------------
  ENSURE(ws->bitmap,ws->nwords*8);
  ENSURE(ws->ncand,(size_t)nreads*4);
  ENSURE(ws->nintvl,(size_t)nreads*4);
  ENSURE(ws->nrel,(size_t)nreads*4);
  ENSURE(ws->ioff,((size_t)nreads+1)*8);
  ENSURE(ws->eoff,((size_t)nreads+1)*8);
  ENSURE(ws->hoff,((size_t)nreads+1)*8);
  // The head of a call: scan, candidate counts, prefix sums, then the one host round trip (scratch sizes depend on the
  // data).  (Tried: the head on a high-priority stream of its own, forked from the caller's -- 143.9 against 146.9
  // Gbases/s without, 187 against 192 with the kernels of the end of round 3: the other stream's wide kernels are what fills the machine, and pre-empting them costs more than
  // the head's latency gains.)
  hipStream_t hs = st;
  if (last_stage < CP_STAGE_LABELS)                      // (the whole path writes both counts for every read: the zeros are for the
    { HIPCHK(hipMemsetAsync(ws->nintvl.p,0,(size_t)nreads*4,hs));      //  stage API's read-back after an early stop)
      HIPCHK(hipMemsetAsync(ws->nrel.p,0,(size_t)nreads*4,hs));
    }
  int rc = launch_scan(&p->host,d_prof,total_kmers,(uint64_t *)ws->bitmap.p,ws->nwords,hs);
  if (rc != CP_OK) return rc;
  hipLaunchKernelGGL(k_count_caps,dim3(nreads),dim3(WAVE),0,hs,
                     (const uint64_t *)ws->bitmap.p,d_prof_off,nreads,
                     (int32_t *)ws->ncand.p,(int64_t *)ws->ioff.p,(int64_t *)ws->eoff.p,(int64_t *)ws->hoff.p);
  ENSURE(ws->dtot,32);
  rc = launch_prefix_caps(ws,(int64_t *)ws->ioff.p,(int64_t *)ws->eoff.p,(int64_t *)ws->hoff.p,nreads,(int64_t *)ws->dtot.p,hs,true);
  if (rc != CP_OK) return rc;
  if (last_stage == CP_STAGE_SCAN)
    return CP_OK;

  // the only host round trip of the pipeline: the last tile of the prefix sums stores the three totals, tagged with the
  // launch's number, into pinned host memory and the host polls for them (every so often it asks the stream whether
  // it died instead)
  { const unsigned tag = (unsigned)ws->scan_epoch;         // 1 .. 2^24-1 (launch_prefix_caps)
    volatile unsigned long long *ht = ws->h_tot64;
    // Polling costs a host core while it lasts (about the head's 1 ms per call and caller: INTEGRATION.md).  Each turn
    // is a cpu-relax; the clock is read every 64 turns; the stream is asked whether it died every 100 us; and a wait
    // that has lasted 2 ms -- the head queued behind other streams' work -- goes on in 20-us sleeps instead of spinning.
    const auto t0 = std::chrono::steady_clock::now();
    auto next_query = t0+std::chrono::microseconds(100);
    bool sleepy = false;
    for (unsigned spins = 1; ; spins++)
      { const unsigned long long g0 = __atomic_load_n(&ht[0],__ATOMIC_ACQUIRE), g1 = __atomic_load_n(&ht[1],__ATOMIC_ACQUIRE),
                                 g2 = __atomic_load_n(&ht[2],__ATOMIC_ACQUIRE);
        if ((g0 & 0xffffffu) == tag && (g1 & 0xffffffu) == tag && (g2 & 0xffffffu) == tag)
          { ws->h_totals[0] = (int64_t)(g0 >> 24); ws->h_totals[1] = (int64_t)(g1 >> 24); ws->h_totals[2] = (int64_t)(g2 >> 24);
            break;
          }
        if (sleepy) { struct timespec ts = { 0, 20000 }; nanosleep(&ts,NULL); }
        else __builtin_ia32_pause();
        if (sleepy || (spins & 63) == 0)
          { const auto now = std::chrono::steady_clock::now();
            if (now-t0 > std::chrono::milliseconds(2)) sleepy = true;
            if (now < next_query) continue;
            next_query = now+std::chrono::microseconds(sleepy ? 1000 : 100);
            const hipError_t q = hipStreamQuery(hs);
            if (q != hipErrorNotReady)                   // the stream is idle (or failed) and the totals never came
              { if (q != hipSuccess) return set_err(CP_EHIP,std::string("cp_run_stages: ")+hipGetErrorString(q));
                HIPCHK(hipMemcpy(&ws->h_totals[0],ws->dtot.p,24,hipMemcpyDeviceToHost));
                break;
              }
            (void)hipGetLastError();
          }
      }
  }
  const int64_t totalI = ws->h_totals[0], totalE = ws->h_totals[1], totalH = ws->h_totals[2];
  ws->totalI = totalI; ws->totalE = totalE; ws->totalH = totalH;

  // ---- stage 2: find_wall ----------------------------------------------------------------------
  const int64_t ncell = total_kmers+nreads;
  // The two flag arrays are all zero between batches: k_find_wall clears, before it ends, the few cells a read's walk
  // touched (its candidates and the ends of its E-/O-intervals), so only a newly allocated array is filled here --
  // not 2 x (one byte per k-mer) per batch.
  { const size_t c0 = ws->wall.cap, c1 = ws->wall_s.cap;
    ENSURE(ws->wall,ncell);
    ENSURE(ws->wall_s,ncell);
    if (ws->wall.cap != c0)   HIPCHK(hipMemsetAsync(ws->wall.p,0,ws->wall.cap,st));
    if (ws->wall_s.cap != c1) HIPCHK(hipMemsetAsync(ws->wall_s.p,0,ws->wall_s.cap,st));
  }
  ENSURE(ws->hkeys,(size_t)totalH*4);
  ENSURE(ws->hvals,(size_t)totalH*4*8);
  ENSURE(ws->eintvl,(size_t)totalE*sizeof(cp_eintvl));
  ENSURE(ws->ointvl,(size_t)totalE*sizeof(cp_eintvl));
  ENSURE(ws->intvl,(size_t)totalI*sizeof(cp_intvl));
  ENSURE(ws->perm,(size_t)nreads*4);
  ENSURE(ws->wlist,(size_t)totalI*4*4);
  // longest reads first: key = wall candidates / 4 (bins of 4 up to 4096 candidates)
  rc = launch_order_by_work(ws,(const int32_t *)ws->ncand.p,nreads,2,(int32_t *)ws->perm.p,st);
  if (rc != CP_OK) return rc;
  ENSURE(ws->tres,(size_t)totalI*sizeof(task_res));
  ENSURE(ws->fwc,(size_t)nreads*16);
  // the walk's read-only part (candidate list, filters, the pure part of every live (candidate, error type) pair), then
  // the replay and the list phases: two kernels because the first wants registers and the second waves (kernels.hip)
  hipLaunchKernelGGL(k_wall_tasks,dim3(nreads),dim3(WAVE),0,st,
                     p->dev,d_seq,d_seq_off,d_prof,d_prof_off,nreads,(const uint64_t *)ws->bitmap.p,(uint8_t *)ws->wall.p,
                     (const int64_t *)ws->ioff.p,(const int32_t *)ws->perm.p,(int32_t *)ws->wlist.p,(task_res *)ws->tres.p,
                     (int32_t *)ws->fwc.p);
  // find_rel_intvl inside k_find_wall (CLASSPRO_FUSE_REL=0: as a kernel of its own, always so when the call stops at
  // the wall stage: the stage API shows find_wall's records as the reference's find_wall leaves them)
  const char *fuse_e = getenv("CLASSPRO_FUSE_REL");       // (read per call: the parity test runs both forms in one process)
  const bool fuse_rel = (!fuse_e || atol(fuse_e) != 0) && last_stage >= CP_STAGE_REL;
  ENSURE(ws->rintvl,(size_t)totalI*sizeof(cp_intvl));
  ENSURE(ws->relmap,(size_t)totalI*4);
  // whole-path calls: the reliable intervals go to classify_rel as 24-byte records (kernels.hip: cp_rrec) instead of
  // 48-byte copies + an index, and only intvl[].asgn comes back (CLASSPRO_COMPACT_REL=0: the copies, as the stage API has them)
  const char *crel_e = getenv("CLASSPRO_COMPACT_REL");
  const bool compact_rel = fuse_rel && last_stage == CP_STAGE_LABELS && (!crel_e || atol(crel_e) != 0);
  // ... and the interval records as three arrays (cp_soa; CLASSPRO_COMPACT_REL=1: the 24-byte and 4-byte records only)
  const bool soa_on = compact_rel && (!crel_e || atol(crel_e) != 1);
  if (compact_rel) { ENSURE(ws->rrec,(size_t)totalI*sizeof(cp_rrec)); ENSURE(ws->pcls,(size_t)totalI*4); }
  if (soa_on) { ENSURE(ws->ivA,(size_t)totalI*sizeof(cp_ivA)); ENSURE(ws->ivB,(size_t)totalI*sizeof(cp_ivB)); ENSURE(ws->ivC,(size_t)totalI*sizeof(cp_ivC)); }
  cp_soa soa; soa.a = soa_on ? (cp_ivA *)ws->ivA.p : nullptr; soa.b = soa_on ? (cp_ivB *)ws->ivB.p : nullptr; soa.c = soa_on ? (cp_ivC *)ws->ivC.p : nullptr;
  ws->last_compact = compact_rel ? (soa_on ? 2 : 1) : 0;
  hipLaunchKernelGGL(k_find_wall,dim3(nreads),dim3(WAVE),0,st,
                     p->dev,d_seq,d_seq_off,d_prof,d_prof_off,nreads,
                     (uint8_t *)ws->wall.p,(uint8_t *)ws->wall_s.p,(int32_t *)ws->hkeys.p,(double *)ws->hvals.p,(const int64_t *)ws->hoff.p,
                     (cp_eintvl *)ws->eintvl.p,(cp_eintvl *)ws->ointvl.p,
                     (const int64_t *)ws->eoff.p,(cp_intvl *)ws->intvl.p,(const int64_t *)ws->ioff.p,
                     (int32_t *)ws->nintvl.p,(int32_t *)ws->err.p,(const int32_t *)ws->perm.p,(int32_t *)ws->wlist.p,
                     (const task_res *)ws->tres.p,(const int32_t *)ws->fwc.p,
                     (cp_intvl *)ws->rintvl.p,(int32_t *)ws->relmap.p,(int32_t *)ws->nrel.p,compact_rel ? 2 : fuse_rel ? 1 : 0,
                     (cp_rrec *)ws->rrec.p,soa);
  HIPCHK(hipGetLastError());
  if (last_stage == CP_STAGE_WALL)
    return CP_OK;

  // ---- stage 3: find_rel_intvl (on the whole-path call: done by k_find_wall on the records it emits) ----------
  if (!fuse_rel)
    { hipLaunchKernelGGL(k_find_rel,dim3(nreads),dim3(WAVE),0,st,
                         p->dev,d_seq,d_seq_off,d_prof,d_prof_off,nreads,(cp_intvl *)ws->intvl.p,(cp_intvl *)ws->rintvl.p,
                         (int32_t *)ws->relmap.p,(const int64_t *)ws->ioff.p,(const int32_t *)ws->nintvl.p,(int32_t *)ws->nrel.p);
      HIPCHK(hipGetLastError());
    }
  if (last_stage == CP_STAGE_REL)
    return CP_OK;

  // ---- stage 4: classify_rel -------------------------------------------------------------------
  ENSURE(ws->parent,(size_t)totalI*2*4);
  ENSURE(ws->eff,(size_t)totalI*2*4);
  ENSURE(ws->rpos,(size_t)totalI*2);
  ENSURE(ws->asgn,(size_t)totalI*2);
  // (the fw / bw assignments are only read back through the stage API: the whole-path call leaves the array as it is --
  //  a 37-MB fill that sat 0.7 ms in the serial path of every 1-Gbase sub-batch)
  if (last_stage < CP_STAGE_LABELS) HIPCHK(hipMemsetAsync(ws->asgn.p,0xff,(size_t)totalI*2,st));
  // size classes (kernels.hip: REL_SMALL_*): M <= 128 four reads per wave, up to 1024 one read per wave, larger (or a
  // read beyond 65535 k-mers): the sequential kernel
  rc = launch_order_by_work(ws,(const int32_t *)ws->nrel.p,nreads,0,(int32_t *)ws->perm.p,st,d_prof_off);
  if (rc != CP_OK) return rc;
  // (the classes touch disjoint reads; the rare classes are a handful of latency-bound waves -- on most batches none at
  //  all, and the sequential kernel's 1024 scratch-using waves still cost 0.9 ms to start -- so they run beside the main
  //  class on the auxiliary stream)
  HIPCHK(hipEventRecord(ws->ev_fork,st));
  HIPCHK(hipStreamWaitEvent(ws->aux,ws->ev_fork,0));
  hipLaunchKernelGGL(k_classify_rel,dim3(nreads < 1024 ? nreads : 1024),dim3(WAVE),0,ws->aux,
                     p->dev,d_prof_off,nreads,(cp_intvl *)ws->intvl.p,(cp_intvl *)ws->rintvl.p,(const int32_t *)ws->relmap.p,
                     (const int64_t *)ws->ioff.p,(const int32_t *)ws->nrel.p,(int8_t *)ws->parent.p,(int32_t *)ws->eff.p,
                     (uint8_t *)ws->rpos.p,(int8_t *)ws->asgn.p,totalI,(const int32_t *)ws->perm.p,soa);
  hipLaunchKernelGGL((k_classify_rel_grp<REL_SMALL_MAXM,1024,1,1,0>),dim3(nreads < 2048 ? nreads : 2048),dim3(WAVE),0,ws->aux,
                     p->dev,d_prof_off,nreads,(cp_intvl *)ws->intvl.p,(cp_intvl *)ws->rintvl.p,(const int32_t *)ws->relmap.p,
                     (const int64_t *)ws->ioff.p,(const int32_t *)ws->nrel.p,(int8_t *)ws->asgn.p,totalI,(const int32_t *)ws->perm.p,
                     (const cp_rrec *)nullptr,soa);
  HIPCHK(hipEventRecord(ws->ev_join,ws->aux));
  if (compact_rel)
    hipLaunchKernelGGL((k_classify_rel_grp<0,REL_SMALL_MAXM,REL_SMALL_G,REL_SMALL_WPB,1>),
                       dim3((nreads+REL_SMALL_G*REL_SMALL_WPB-1)/(REL_SMALL_G*REL_SMALL_WPB)),dim3(WAVE*REL_SMALL_WPB),0,st,
                       p->dev,d_prof_off,nreads,(cp_intvl *)ws->intvl.p,(cp_intvl *)ws->rintvl.p,(const int32_t *)ws->relmap.p,
                       (const int64_t *)ws->ioff.p,(const int32_t *)ws->nrel.p,(int8_t *)ws->asgn.p,totalI,(const int32_t *)ws->perm.p,
                       (const cp_rrec *)ws->rrec.p,soa);
  else
    hipLaunchKernelGGL((k_classify_rel_grp<0,REL_SMALL_MAXM,REL_SMALL_G,REL_SMALL_WPB,0>),
                       dim3((nreads+REL_SMALL_G*REL_SMALL_WPB-1)/(REL_SMALL_G*REL_SMALL_WPB)),dim3(WAVE*REL_SMALL_WPB),0,st,
                       p->dev,d_prof_off,nreads,(cp_intvl *)ws->intvl.p,(cp_intvl *)ws->rintvl.p,(const int32_t *)ws->relmap.p,
                       (const int64_t *)ws->ioff.p,(const int32_t *)ws->nrel.p,(int8_t *)ws->asgn.p,totalI,(const int32_t *)ws->perm.p,
                       (const cp_rrec *)nullptr,soa);
  HIPCHK(hipStreamWaitEvent(st,ws->ev_join,0));
  HIPCHK(hipGetLastError());
  if (last_stage == CP_STAGE_CLASS_REL)
    return CP_OK;

  // ---- stage 5: classify_unrel -------------------------------------------------------------------
  ENSURE(ws->ord,(size_t)totalI*4);
  // size classes (kernels.hip: UNREL_SMALL_*): N <= 256 and up to 1024, two reads per wave each (four speculative update
  // slots per read), larger: the sequential kernel; the rare classes on the auxiliary stream again
  rc = launch_order_by_work(ws,(const int32_t *)ws->nintvl.p,nreads,0,(int32_t *)ws->perm.p,st,d_prof_off);
  if (rc != CP_OK) return rc;
  // the second sweep re-evaluates only the intervals whose inputs changed since the first (kernels.hip);
  // CLASSPRO_UNREL_SWEEP2=full: every one of them, as the reference does (same classes: tests, A/B)
  const char *sw2_e = getenv("CLASSPRO_UNREL_SWEEP2");
  const int full_sweep2 = (sw2_e && !strcmp(sw2_e,"full")) ? 1 : 0;
  HIPCHK(hipEventRecord(ws->ev_fork,st));
  HIPCHK(hipStreamWaitEvent(ws->aux,ws->ev_fork,0));
  hipLaunchKernelGGL(k_classify_unrel,dim3(nreads < 1024 ? nreads : 1024),dim3(WAVE),0,ws->aux,
                     p->dev,nreads,(cp_intvl *)ws->intvl.p,(const int64_t *)ws->ioff.p,(const int32_t *)ws->nintvl.p,
                     (int32_t *)ws->ord.p,(const int32_t *)ws->perm.p,d_prof_off,compact_rel ? (uint32_t *)ws->pcls.p : (uint32_t *)nullptr,soa);
  hipLaunchKernelGGL((k_classify_unrel_grp<UNREL_SMALL_MAXN,1024,UNREL_BIG_G,0>),dim3((nreads+UNREL_BIG_G-1)/UNREL_BIG_G < 2048 ? (nreads+UNREL_BIG_G-1)/UNREL_BIG_G : 2048),dim3(WAVE),0,ws->aux,
                     p->dev,nreads,(cp_intvl *)ws->intvl.p,(const int64_t *)ws->ioff.p,(const int32_t *)ws->nintvl.p,
                     (const int32_t *)ws->perm.p,d_prof_off,compact_rel ? (uint32_t *)ws->pcls.p : (uint32_t *)nullptr,soa,full_sweep2);
  HIPCHK(hipEventRecord(ws->ev_join,ws->aux));
  if (soa_on)
    hipLaunchKernelGGL((k_classify_unrel_grp<0,UNREL_SMALL_MAXN,UNREL_SMALL_G,1>),dim3((nreads+UNREL_SMALL_G-1)/UNREL_SMALL_G),dim3(WAVE),0,st,
                       p->dev,nreads,(cp_intvl *)ws->intvl.p,(const int64_t *)ws->ioff.p,(const int32_t *)ws->nintvl.p,
                       (const int32_t *)ws->perm.p,d_prof_off,(uint32_t *)ws->pcls.p,soa,full_sweep2);
  else
    hipLaunchKernelGGL((k_classify_unrel_grp<0,UNREL_SMALL_MAXN,UNREL_SMALL_G,0>),dim3((nreads+UNREL_SMALL_G-1)/UNREL_SMALL_G),dim3(WAVE),0,st,
                       p->dev,nreads,(cp_intvl *)ws->intvl.p,(const int64_t *)ws->ioff.p,(const int32_t *)ws->nintvl.p,
                       (const int32_t *)ws->perm.p,d_prof_off,compact_rel ? (uint32_t *)ws->pcls.p : (uint32_t *)nullptr,soa,full_sweep2);
  HIPCHK(hipStreamWaitEvent(st,ws->ev_join,0));
  HIPCHK(hipGetLastError());
  if (last_stage == CP_STAGE_CLASS_ALL)
    return CP_OK;

  // ---- stage 6: labels ---------------------------------------------------------------------------
  hipLaunchKernelGGL(k_paint_labels,dim3(nreads),dim3(WAVE),0,st,
                     p->dev,d_seq_off,nreads,(const cp_intvl *)ws->intvl.p,(const int64_t *)ws->ioff.p,
                     (const int32_t *)ws->nintvl.p,d_labels,compact_rel ? (const uint32_t *)ws->pcls.p : (const uint32_t *)nullptr);
  HIPCHK(hipGetLastError());
  return CP_OK;
}

extern "C" int cp_classify_batch(const cp_params *p, cp_workspace *ws,
                                 const char *d_seq, const int64_t *d_seq_off,
                                 const uint16_t *d_prof, const int64_t *d_prof_off,
                                 int nreads, int64_t total_bases, int64_t total_kmers,
                                 char *d_labels, void *stream)
{ return cp_run_stages(p,ws,d_seq,d_seq_off,d_prof,d_prof_off,nreads,total_bases,total_kmers,d_labels,
                       CP_STAGE_LABELS,stream);
}

// Waits for the last run on `ws` and reports scratch overflow (E-interval list or interval array).
// The reference aborts in the same situation ("# E-intvls >= plen", wall.c:783-788).
extern "C" int cp_workspace_check(cp_workspace *ws)
{ if (!ws) return set_err(CP_EINVAL,"cp_workspace_check: null workspace");
  // one read-back of the three sticky error words, then they are cleared: every flag raised by a call on this
  // workspace since the previous check is reported once
  HIPCHK(hipMemcpyAsync(ws->h_err,ws->err.p,16,hipMemcpyDeviceToHost,ws->stream));
  HIPCHK(hipMemsetAsync(ws->err.p,0,16,ws->stream));
  HIPCHK(hipStreamSynchronize(ws->stream));
  for (void *q : *ws->retired) (void)hipFree(q);          // nothing on the stream uses an outgrown buffer any more
  ws->retired->clear();
  ws->decode_pending = 0;
  const int ew = ws->h_err[0], es = ws->h_err[1], ed = ws->h_err[2];
  if (ed)
    return set_err(CP_EINVAL,"cp_decode_profiles: a code string does not expand to its read's profile length (rlen != plen+K-1)");
  if (ew)
    { char m[160];
      if (ew & 8) snprintf(m,sizeof(m),"# E-intvls >= plen: the reference exits on a read of this batch (wall.c:783-788) (flags=%d)",ew);
      else snprintf(m,sizeof(m),"scratch overflow in find_wall (flags=%d): too many E-intervals for a read",ew);
      return set_err(CP_EOVERFLOW,m);
    }
  if (es)
    { char m[160];
      snprintf(m,sizeof(m),"scratch overflow in find_seeds (flags=%d)",es);
      return set_err(CP_EOVERFLOW,m);
    }
  return CP_OK;
}

extern "C" int cp_get_counts(cp_workspace *ws, int32_t *n_cand, int32_t *n_intvl, int32_t *n_rel, int64_t *cap_off)
{ if (!ws) return set_err(CP_EINVAL,"cp_get_counts: null workspace");
  int rc = cp_workspace_check(ws);
  if (rc != CP_OK) return rc;
  size_t n = (size_t)ws->nreads;
  if (n == 0) return CP_OK;
  if (n_cand)  HIPCHK(hipMemcpy(n_cand,ws->ncand.p,n*4,hipMemcpyDeviceToHost));
  if (n_intvl) HIPCHK(hipMemcpy(n_intvl,ws->nintvl.p,n*4,hipMemcpyDeviceToHost));
  if (n_rel)   HIPCHK(hipMemcpy(n_rel,ws->nrel.p,n*4,hipMemcpyDeviceToHost));
  if (cap_off) HIPCHK(hipMemcpy(cap_off,ws->ioff.p,(n+1)*8,hipMemcpyDeviceToHost));
  return CP_OK;
}

extern "C" int cp_get_intervals(cp_workspace *ws, cp_intvl *intvl, cp_intvl *rintvl, int64_t capacity)
{ if (!ws) return set_err(CP_EINVAL,"cp_get_intervals: null workspace");
  if (capacity < ws->totalI) return set_err(CP_EINVAL,"cp_get_intervals: capacity too small");
  int rc = cp_workspace_check(ws);
  if (rc != CP_OK) return rc;
  size_t bytes = (size_t)ws->totalI*sizeof(cp_intvl);
  if (bytes == 0) return CP_OK;
  if (intvl && ws->last_stage >= CP_STAGE_WALL) HIPCHK(hipMemcpy(intvl,ws->intvl.p,bytes,hipMemcpyDeviceToHost));
  if (ws->last_compact)
    { // after a whole-path call (cp_classify_batch) the final classes live in the (end, class) words the label paint reads,
      // not in the records, and the reliable-interval copies were never made: the classes are put into the records here,
      // `rintvl` comes back zeroed (run the stages up to CP_STAGE_CLASS_ALL for it)
      if (rintvl) memset(rintvl,0,bytes);
      if (intvl)
        try
        { std::vector<uint32_t> pc((size_t)ws->totalI);
          std::vector<int64_t> off((size_t)ws->nreads+1);
          std::vector<int32_t> ni((size_t)ws->nreads);
          HIPCHK(hipMemcpy(pc.data(),ws->pcls.p,(size_t)ws->totalI*4,hipMemcpyDeviceToHost));
          HIPCHK(hipMemcpy(off.data(),ws->ioff.p,((size_t)ws->nreads+1)*8,hipMemcpyDeviceToHost));
          HIPCHK(hipMemcpy(ni.data(),ws->nintvl.p,(size_t)ws->nreads*4,hipMemcpyDeviceToHost));
          std::vector<cp_ivA> A; std::vector<cp_ivB> B; std::vector<cp_ivC> Cc;
          if (ws->last_compact == 2)                      // the records went to the device as three arrays: put them together
            { A.resize((size_t)ws->totalI); B.resize((size_t)ws->totalI); Cc.resize((size_t)ws->totalI);
              HIPCHK(hipMemcpy(A.data(),ws->ivA.p,(size_t)ws->totalI*sizeof(cp_ivA),hipMemcpyDeviceToHost));
              HIPCHK(hipMemcpy(B.data(),ws->ivB.p,(size_t)ws->totalI*sizeof(cp_ivB),hipMemcpyDeviceToHost));
              HIPCHK(hipMemcpy(Cc.data(),ws->ivC.p,(size_t)ws->totalI*sizeof(cp_ivC),hipMemcpyDeviceToHost));
            }
          for (int r = 0; r < ws->nreads; r++)
            for (int k = 0; k < ni[(size_t)r]; k++)
              { const size_t at = (size_t)(off[(size_t)r]+k);
                cp_intvl &I = intvl[at];
                if (ws->last_compact == 2)
                  { memset(&I,0,sizeof(I));
                    I.b = A[at].b; I.e = A[at].e; I.cb = A[at].cb; I.ce = A[at].ce; I.ccb = A[at].ccb; I.cce = A[at].cce;
                    I.is_rel = Cc[at].is_rel; I.pe = B[at].pe; I.peo_b = B[at].peo_b; I.peo_e = B[at].peo_e;
                  }
                I.asgn = (int8_t)CP_PCLS_CLS(pc[at]);
              }
        }
        catch (const std::bad_alloc &) { return set_err(CP_ENOMEM,"cp_get_intervals: out of host memory"); }   // (never through the C ABI)
      return CP_OK;
    }
  if (rintvl && ws->last_stage >= CP_STAGE_REL) HIPCHK(hipMemcpy(rintvl,ws->rintvl.p,bytes,hipMemcpyDeviceToHost));
  return CP_OK;
}

extern "C" int cp_get_rel_asgn(cp_workspace *ws, int8_t *fw, int8_t *bw, int64_t capacity)
{ if (!ws) return set_err(CP_EINVAL,"cp_get_rel_asgn: null workspace");
  if (capacity < ws->totalI) return set_err(CP_EINVAL,"cp_get_rel_asgn: capacity too small");
  if (ws->last_stage < CP_STAGE_CLASS_REL) return set_err(CP_EINVAL,"cp_get_rel_asgn: stage not run");
  if (ws->last_stage == CP_STAGE_LABELS) return set_err(CP_EINVAL,"cp_get_rel_asgn: run cp_run_stages up to CP_STAGE_CLASS_REL or CP_STAGE_CLASS_ALL (the whole-path call does not keep the two assignments of reads without reliable intervals defined)");
  int rc = cp_workspace_check(ws);
  if (rc != CP_OK) return rc;
  size_t n = (size_t)ws->totalI;
  if (n == 0) return CP_OK;
  if (fw) HIPCHK(hipMemcpy(fw,ws->asgn.p,n,hipMemcpyDeviceToHost));
  if (bw) HIPCHK(hipMemcpy(bw,(int8_t *)ws->asgn.p+n,n,hipMemcpyDeviceToHost));
  return CP_OK;
}

extern "C" int cp_get_bitmap(cp_workspace *ws, uint64_t *words, int64_t nwords)
{ if (!ws || !words) return set_err(CP_EINVAL,"cp_get_bitmap: null argument");
  if (nwords > ws->nwords) nwords = ws->nwords;
  HIPCHK(hipStreamSynchronize(ws->stream));
  if (nwords > 0) HIPCHK(hipMemcpy(words,ws->bitmap.p,(size_t)nwords*8,hipMemcpyDeviceToHost));
  return CP_OK;
}

extern "C" int64_t cp_encode_profile(const uint16_t *profile, int n, uint8_t *code, int64_t cap)
{ if (n < 0 || (n > 0 && (!profile || !code)) || cap < 2*(int64_t)n+2)
    return set_err(CP_EINVAL,"cp_encode_profile: bad argument (code buffer needs 2*n+2 bytes)");
  return cp_host_encode_profile(profile,n,code);
}

extern "C" int cp_decode_profiles(cp_workspace *ws, const uint8_t *d_codes, const int64_t *d_code_off,
                                  const int64_t *d_prof_off, int nreads, uint16_t *d_prof, void *stream)
{ if (!ws || nreads < 0 || (nreads > 0 && (!d_codes || !d_code_off || !d_prof_off || !d_prof)))
    return set_err(CP_EINVAL,"cp_decode_profiles: bad argument");
  if (nreads == 0) return CP_OK;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_decode_profiles,dim3(nreads),dim3(WAVE),0,st,
                     d_codes,d_code_off,d_prof_off,nreads,d_prof,(int32_t *)ws->err.p+2);
  HIPCHK(hipGetLastError());
  ws->decode_pending = 1;
  ws->stream = st;
  return CP_OK;
}

// -s: find_seeds (seed.c:966-1032) for every read of a classified batch
extern "C" int cp_find_seeds_batch(const cp_params *p, cp_workspace *ws, const char *d_seq, const int64_t *d_seq_off,
                                   const uint16_t *d_prof, const int64_t *d_prof_off, const char *d_labels,
                                   int nreads, int64_t total_bases, int64_t total_kmers, char *d_seeds, void *stream)
{ if (!p || !ws || nreads < 0 || total_bases < 0 || total_kmers < 0)
    return set_err(CP_EINVAL,"cp_find_seeds_batch: bad argument");
  if (nreads > 0 && (!d_seq || !d_seq_off || !d_prof || !d_prof_off || !d_labels || !d_seeds))
    return set_err(CP_EINVAL,"cp_find_seeds_batch: null device pointer");
  if (total_kmers > CP_MAX_BATCH_KMERS)                   // same limit as cp_run_stages (the seed capacities are ~2 per position)
    return set_err(CP_EINVAL,"cp_find_seeds_batch: more than CP_MAX_BATCH_KMERS k-mer positions in one batch: split it");
  hipStream_t st = (hipStream_t)stream;
  ws->stream = st; ws->seed_nreads = nreads; ws->seed_totalR = 0;
  if (nreads == 0) return CP_OK;
  const int K = p->host.K;
  ENSURE(ws->s_cap,((size_t)nreads+1)*8);
  ENSURE(ws->s_rcap,((size_t)nreads+1)*8);
  ENSURE(ws->s_dummy,((size_t)nreads+1)*8);
  ENSURE(ws->s_key,(size_t)nreads*4);
  ENSURE(ws->perm,(size_t)nreads*4);
  ENSURE(ws->s_repcnt,(size_t)nreads*4);
  HIPCHK(hipMemsetAsync(ws->s_dummy.p,0,((size_t)nreads+1)*8,st));
  hipLaunchKernelGGL(k_seed_caps,dim3(nreads),dim3(WAVE),0,st,d_prof,d_prof_off,d_labels,d_seq_off,K,nreads,
                     (int64_t *)ws->s_cap.p,(int64_t *)ws->s_rcap.p,(int32_t *)ws->s_key.p);
  ENSURE(ws->dtot,32);
  { int rc = launch_prefix_caps(ws,(int64_t *)ws->s_cap.p,(int64_t *)ws->s_rcap.p,(int64_t *)ws->s_dummy.p,nreads,(int64_t *)ws->dtot.p,st);
    if (rc != CP_OK) return rc;
  }
  HIPCHK(hipMemcpyAsync(&ws->h_totals[0],ws->dtot.p,16,hipMemcpyDeviceToHost,st));
  HIPCHK(hipStreamSynchronize(st));
  const int64_t totalS = ws->h_totals[0], totalR = ws->h_totals[1];
  ws->seed_totalR = totalR;
  ENSURE(ws->s_seg,(size_t)totalS*2*16);
  ENSURE(ws->s_aux,(size_t)totalS*4);
  ENSURE(ws->s_mi,((size_t)totalS+3*(size_t)nreads)*2*4);
  ENSURE(ws->s_rep,(size_t)totalR*2*4+16);
  HIPCHK(hipMemsetAsync(d_seeds,'E',(size_t)total_bases,st));
  hipLaunchKernelGGL(k_seed_prefix,dim3(nreads),dim3(WAVE),0,st,d_seq_off,K,nreads,d_seeds);
  // longest reads first: key = plen / 64
  { int rc = launch_order_by_work(ws,(const int32_t *)ws->s_key.p,nreads,6,(int32_t *)ws->perm.p,st);
    if (rc != CP_OK) return rc;
  }
  hipLaunchKernelGGL(k_find_seeds,dim3(nreads),dim3(WAVE),0,st,d_seq,d_seq_off,d_prof,d_prof_off,d_labels,K,nreads,
                     (const int64_t *)ws->s_cap.p,(const int64_t *)ws->s_rcap.p,(const int32_t *)ws->perm.p,
                     (int32_t *)ws->s_seg.p,(int32_t *)ws->s_aux.p,(int32_t *)ws->s_mi.p,
                     (int32_t *)ws->s_rep.p,(int32_t *)ws->s_repcnt.p,d_seeds,(int32_t *)ws->err.p+1,totalS);
  HIPCHK(hipGetLastError());
  return CP_OK;
}

extern "C" int cp_get_rep_masks(cp_workspace *ws, int32_t *count, int64_t *cap_off, int32_t *pairs, int64_t capacity)
{ if (!ws) return set_err(CP_EINVAL,"cp_get_rep_masks: null workspace");
  if (capacity < ws->seed_totalR) return set_err(CP_EINVAL,"cp_get_rep_masks: capacity too small");
  int rc = cp_workspace_check(ws);
  if (rc != CP_OK) return rc;
  const size_t n = (size_t)ws->seed_nreads;
  if (n == 0) return CP_OK;
  if (count)   HIPCHK(hipMemcpy(count,ws->s_repcnt.p,n*4,hipMemcpyDeviceToHost));
  if (cap_off) HIPCHK(hipMemcpy(cap_off,ws->s_rcap.p,(n+1)*8,hipMemcpyDeviceToHost));
  if (pairs && ws->seed_totalR > 0) HIPCHK(hipMemcpy(pairs,ws->s_rep.p,(size_t)ws->seed_totalR*8,hipMemcpyDeviceToHost));
  return CP_OK;
}
extern "C" int64_t cp_rep_masks_capacity(const cp_workspace *ws) { return ws ? ws->seed_totalR : 0; }

extern "C" int cp_unpack_bases(const uint8_t *d_packed, const int64_t *d_pack_off, const int64_t *d_seq_off,
                               int nreads, char *d_seq, void *stream)
{ if (nreads < 0 || (nreads > 0 && (!d_packed || !d_pack_off || !d_seq_off || !d_seq)))
    return set_err(CP_EINVAL,"cp_unpack_bases: bad argument");
  if (nreads == 0) return CP_OK;
  hipLaunchKernelGGL(k_unpack_bases,dim3(nreads),dim3(256),0,(hipStream_t)stream,d_packed,d_pack_off,d_seq_off,nreads,d_seq);
  HIPCHK(hipGetLastError());
  return CP_OK;
}

extern "C" int cp_pack_labels(const char *d_labels, const int64_t *d_seq_off, const int64_t *d_pack_off, int nreads,
                              uint8_t *d_packed, void *stream)
{ if (nreads < 0 || (nreads > 0 && (!d_labels || !d_seq_off || !d_pack_off || !d_packed)))
    return set_err(CP_EINVAL,"cp_pack_labels: bad argument");
  if (nreads == 0) return CP_OK;
  hipLaunchKernelGGL(k_pack_labels,dim3(nreads),dim3(256),0,(hipStream_t)stream,d_labels,d_seq_off,d_pack_off,nreads,d_packed);
  HIPCHK(hipGetLastError());
  return CP_OK;
}

// host: the inverse of cp_pack_labels for one read (stoc of const.c:19 on the codes; the first K-1 labels are 'N')
extern "C" int cp_unpack_labels(const uint8_t *packed, int rlen, int K, char *labels)
{ if (rlen < 0 || K < 1 || (rlen > 0 && (!packed || !labels)))
    return set_err(CP_EINVAL,"cp_unpack_labels: bad argument");
  static const char stoc[4] = { 'E', 'R', 'H', 'D' };
  const int np = K-1 < rlen ? K-1 : rlen;
  for (int i = 0; i < np; i++) labels[i] = 'N';
  for (int i = np; i < rlen; i++)
    labels[i] = stoc[(packed[i >> 2] >> (6-2*(i & 3))) & 3];
  return CP_OK;
}

// host: a read's bases as 2-bit codes in the layout cp_unpack_bases expands (Compress_Read, gene_core.c:235-254), if it
// consists of upper-case A, C, G, T only.  Returns 1 (packed: (rlen+3)/4 bytes written) or 0 (another letter: the read
// -- and the batch it is in -- goes to the device as characters; the context scans compare raw characters,
// context.c:8-108, so a lower-case or ambiguous base must arrive as it is).
// 32 bases per step with AVX2 + BMI2 where the host has them (every x86-64 server a MI355X sits in does): the feeder packs
// a batch's bases while the previous batch is on the wire, and a byte-at-a-time table walk (0.5 GB/s per thread) would
// be the slowest stage of the PCIe pipeline.  (b >> 1) & 3 maps A C G T to 0 1 3 2; x ^ (x >> 1) turns that into the
// Dazzler codes 0 1 2 3; a letter is valid iff "ACTG"[(b >> 1) & 3] == b.  Returns the number of bases packed (a
// multiple of 32; the caller finishes the tail), or -1 at the first group holding another letter.
#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("avx2,bmi2"))) static int64_t pack_bases_avx2(const char *seq, int64_t n, uint8_t *packed)
{ const __m256i tab = _mm256_setr_epi8('A','C','T','G',0,0,0,0,0,0,0,0,0,0,0,0, 'A','C','T','G',0,0,0,0,0,0,0,0,0,0,0,0);
  const __m256i three = _mm256_set1_epi8(3);
  int64_t i = 0;
  for (; i+32 <= n; i += 32)
    { const __m256i b = _mm256_loadu_si256((const __m256i *)(seq+i));
      const __m256i idx = _mm256_and_si256(_mm256_srli_epi16(b,1),three);
      if (_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_shuffle_epi8(tab,idx),b)) != -1) return -1;
      const __m256i code = _mm256_xor_si256(idx,_mm256_and_si256(_mm256_srli_epi16(idx,1),_mm256_set1_epi8(1)));
      uint64_t q[4];
      _mm256_storeu_si256((__m256i *)q,code);
      for (int k = 0; k < 4; k++)                             // 8 codes (one per byte, little end first) -> 16 bits, first base on top
        { const uint64_t v = __builtin_bswap64(q[k]);        // first base in the top byte
          const uint32_t w = (uint32_t)_pext_u64(v,0x0303030303030303ull);   // 16 bits, first base in bits 15:14
          packed[(i >> 2)+2*k]   = (uint8_t)(w >> 8);
          packed[(i >> 2)+2*k+1] = (uint8_t)w;
        }
    }
  return i;
}
#endif

extern "C" int cp_pack_bases(const char *seq, int rlen, uint8_t *packed)
{ if (rlen < 0 || (rlen > 0 && (!seq || !packed)))
    return set_err(CP_EINVAL,"cp_pack_bases: bad argument");
#if defined(__x86_64__)
  static const bool fast = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2");
  if (fast && rlen >= 64)
    { const int64_t done = pack_bases_avx2(seq,rlen,packed);
      if (done < 0) return 0;
      if (done == rlen) return 1;
      const int rc = cp_pack_bases(seq+done,(int)(rlen-done),packed+(done >> 2));   // the tail (< 32 bases) by the table
      return rc;
    }
#endif
  static const signed char code[256] = {
#define X -1
    X,X,X,X,X,X,X,X,X,X,X,X,X,X,X,X, X,X,X,X,X,X,X,X,X,X,X,X,X,X,X,X, X,X,X,X,X,X,X,X,X,X,X,X,X,X,X,X, X,X,X,X,X,X,X,X,X,X,X,X,X,X,X,X,
    X,0,X,1,X,X,X,2,X,X,X,X,X,X,X,X, X,X,X,X,3,X,X,X,X,X,X,X,X,X,X,X, X,X,X,X,X,X,X,X,X,X,X,X,X,X,X,X, X,X,X,X,X,X,X,X,X,X,X,X,X,X,X,X,
    X,X,X,X,X,X,X,X,X,X,X,X,X,X,X,X, X,X,X,X,X,X,X,X,X,X,X,X,X,X,X,X, X,X,X,X,X,X,X,X,X,X,X,X,X,X,X,X, X,X,X,X,X,X,X,X,X,X,X,X,X,X,X,X,
    X,X,X,X,X,X,X,X,X,X,X,X,X,X,X,X, X,X,X,X,X,X,X,X,X,X,X,X,X,X,X,X, X,X,X,X,X,X,X,X,X,X,X,X,X,X,X,X, X,X,X,X,X,X,X,X,X,X,X,X,X,X,X,X
#undef X
  };
  int bad = 0;
  int i = 0;
  for (; i+4 <= rlen; i += 4)
    { const int a = code[(unsigned char)seq[i]], b = code[(unsigned char)seq[i+1]], c = code[(unsigned char)seq[i+2]], d = code[(unsigned char)seq[i+3]];
      bad |= a | b | c | d;
      packed[i >> 2] = (uint8_t)((a << 6) | (b << 4) | (c << 2) | d);
    }
  if (i < rlen)
    { int v = 0;
      for (int q = 0; q < 4; q++)
        { const int a = (i+q < rlen) ? code[(unsigned char)seq[i+q]] : 0;
          bad |= a;
          v |= (a & 3) << (6-2*q);
        }
      packed[i >> 2] = (uint8_t)v;
    }
  return bad < 0 ? 0 : 1;
}

// cp_pack_bases for every read of a batch on `nthreads` host threads (contiguous read ranges, balanced by bases): the
// host side of the 2-bit payload, called by a feeder on the pinned staging buffer it is about to send.
extern "C" int cp_pack_bases_batch(const char *seq, const int64_t *seq_off, int nreads, uint8_t *packed, const int64_t *pack_off,
                                   int nthreads)
{ if (nreads < 0 || (nreads > 0 && (!seq || !seq_off || !packed || !pack_off)))
    return set_err(CP_EINVAL,"cp_pack_bases_batch: bad argument");
  if (nreads == 0) return 1;
  if (nthreads < 1) nthreads = 1;
  if (nthreads > nreads) nthreads = nreads;
  std::vector<int> ok((size_t)nthreads,1);
  auto work = [&](int t)
    { const int64_t b0 = seq_off[0], tot = seq_off[nreads]-b0;
      // first read whose start is at or beyond t/nthreads of the bases
      auto cut = [&](int q) -> int
        { const int64_t want = b0+tot*q/nthreads;
          int lo = 0, hi = nreads;
          while (lo < hi) { const int m = (lo+hi) >> 1; if (seq_off[m] < want) lo = m+1; else hi = m; }
          return lo;
        };
      const int r0 = t == 0 ? 0 : cut(t), r1 = t == nthreads-1 ? nreads : cut(t+1);
      int good = 1;
      for (int r = r0; r < r1; r++)
        good &= cp_pack_bases(seq+seq_off[r],(int)(seq_off[r+1]-seq_off[r]),packed+pack_off[r]) == 1;
      ok[(size_t)t] = good;
    };
  if (nthreads == 1) work(0);
  else
    { // A thread that cannot be started (std::system_error under a process / thread limit) must not unwind through the
      // C ABI: the ranges that got no thread are done by the caller, the threads that did start are joined either way.
      std::vector<std::thread> th;
      int started = 0;
      try
        { th.reserve((size_t)nthreads);
          for (; started < nthreads-1; started++) th.emplace_back(work,started);
        }
      catch (...) { }
      for (int t = started; t < nthreads; t++) work(t);
      for (auto &x : th) x.join();
    }
  for (int v : ok) if (!v) return 0;
  return 1;
}

// Labels as runs (kernels.hip: k_label_runs) of the batch last classified on `ws` (cp_run_stages up to CP_STAGE_CLASS_ALL or
// cp_classify_batch): d_ends / d_cls take cp_label_runs_capacity(ws) entries, d_nruns nreads, d_cap_off nreads+1 (a copy of the
// capacity offsets: read r's runs start at index d_cap_off[r]).
extern "C" int64_t cp_label_runs_capacity(const cp_workspace *ws) { return ws ? ws->totalI : 0; }

extern "C" int cp_label_runs(const cp_params *p, cp_workspace *ws, int32_t *d_ends, uint8_t *d_cls, int32_t *d_nruns, int64_t *d_cap_off,
                             void *stream)
{ if (!p || !ws || (ws->nreads > 0 && (!d_ends || !d_cls || !d_nruns || !d_cap_off)))
    return set_err(CP_EINVAL,"cp_label_runs: bad argument");
  if (ws->nreads == 0) return CP_OK;
  if (ws->last_stage < CP_STAGE_CLASS_ALL)
    return set_err(CP_EINVAL,"cp_label_runs: the batch on this workspace was not classified (run at least CP_STAGE_CLASS_ALL)");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_label_runs,dim3(ws->nreads),dim3(WAVE),0,st,p->dev,ws->nreads,(const cp_intvl *)ws->intvl.p,(const int64_t *)ws->ioff.p,
                     (const int32_t *)ws->nintvl.p,d_ends,d_cls,d_nruns,ws->last_compact ? (const uint32_t *)ws->pcls.p : (const uint32_t *)nullptr);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(d_cap_off,ws->ioff.p,((size_t)ws->nreads+1)*8,hipMemcpyDeviceToDevice,st));
  return CP_OK;
}

// host: the label string of one read from its runs (K-1 'N', then run j = labels [ends[j-1], ends[j]) of class cls[j])
extern "C" int cp_expand_label_runs(const int32_t *ends, const uint8_t *cls, int nruns, int rlen, int K, char *labels)
{ if (rlen < 0 || nruns < 0 || K < 1 || (rlen > 0 && !labels) || (nruns > 0 && (!ends || !cls)))
    return set_err(CP_EINVAL,"cp_expand_label_runs: bad argument");
  int pos = K-1 < rlen ? K-1 : rlen;
  memset(labels,'N',(size_t)pos);
  for (int j = 0; j < nruns; j++)
    { const int e = ends[j];
      if (e < pos || e > rlen) return set_err(CP_EINVAL,"cp_expand_label_runs: runs out of order or beyond the read");
      memset(labels+pos,(int)cls[j],(size_t)(e-pos));
      pos = e;
    }
  if (pos != rlen) return set_err(CP_EINVAL,"cp_expand_label_runs: the runs do not cover the read");   // (rlen < K: pos == rlen already)
  return CP_OK;
}

extern "C" int cp_seq_context(const char *d_seq, const int64_t *d_seq_off, int nreads, int64_t total_bases,
                              uint8_t *d_lctx, uint8_t *d_rctx, void *stream)
{ if (!d_seq || !d_seq_off || !d_lctx || !d_rctx || nreads < 0)
    return set_err(CP_EINVAL,"cp_seq_context: bad argument");
  (void)total_bases;
  if (nreads == 0) return CP_OK;
  hipLaunchKernelGGL(k_seq_context,dim3(nreads),dim3(256),0,(hipStream_t)stream,d_seq,d_seq_off,nreads,d_lctx,d_rctx);
  HIPCHK(hipGetLastError());
  return CP_OK;
}

#ifdef CP_BOUNDS
// -DCP_BOUNDS diagnostic builds only (cp_bounds.h): accesses outside a read counted since the last call, and the first
// one (kind 1 / 9: a count / a run of counts, 2 / 10: a base / a run of bases, 3 / 4: the seed kernel's counts / bases).
extern "C" int cp_debug_bounds(unsigned long long *out4)
{ HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpyFromSymbol(out4,HIP_SYMBOL(g_bounds),sizeof(unsigned long long)*4));
  unsigned long long z[4] = { 0, 0, 0, 0 };
  HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_bounds),z,sizeof(z)));
  return CP_OK;
}
#endif

#if defined(CP_SEED_DEBUG) || defined(CP_SEED_DEBUG_TAKES)
extern "C" int cp_debug_seed_set(int read)
{ HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_seed_dbg_read),&read,sizeof(int))); return CP_OK; }
extern "C" int cp_debug_seed_get(int *out4, int *recs)
{ HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpyFromSymbol(out4,HIP_SYMBOL(g_seed_dbg_n),16));
  HIPCHK(hipMemcpyFromSymbol(recs,HIP_SYMBOL(g_seed_dbg),8192*16));
  return CP_OK;
}
#endif
#ifdef CP_SEED_PROF
// diagnostic builds only: wall-clock ticks (100 MHz) of lane 0 of every wave of k_find_seeds_fast per phase, then reset
extern "C" int cp_debug_seed_prof(unsigned long long *out8)
{ HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpyFromSymbol(out8,HIP_SYMBOL(g_seed_prof),8*sizeof(unsigned long long)));
  unsigned long long z[8] = {0};
  HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_seed_prof),z,sizeof(z)));
  return CP_OK;
}
#endif
#ifdef CP_PROF_WALK
// diagnostic builds only: per read (candidates, tasks, SELF tasks, overflow) as k_wall_tasks left them (scripts/count_dist.py)
extern "C" int cp_debug_task_counts(cp_workspace *ws, int32_t *out, int64_t nreads)
{ if (!ws || nreads > ws->nreads) return set_err(CP_EINVAL,"cp_debug_task_counts: bad arguments");
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(out,ws->fwc.p,(size_t)nreads*4*sizeof(int32_t),hipMemcpyDeviceToHost));
  return CP_OK;
}
// diagnostic builds only: per-phase wave times of k_find_wall (max / sum / arg-max over reads), then reset
extern "C" int cp_debug_live_prof(unsigned long long *out8)
{ HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpyFromSymbol(out8,HIP_SYMBOL(g_live_prof),8*sizeof(unsigned long long)));
  unsigned long long z[8] = {0};
  HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_live_prof),z,sizeof(z)));
  return CP_OK;
}
extern "C" int cp_debug_emit_prof(unsigned long long *out8)
{ HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpyFromSymbol(out8,HIP_SYMBOL(g_emit_prof),8*sizeof(unsigned long long)));
  unsigned long long z[8] = {0};
  HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_emit_prof),z,sizeof(z)));
  return CP_OK;
}
extern "C" int cp_debug_phase_prof(unsigned long long *out36)
{ HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpyFromSymbol(out36,HIP_SYMBOL(g_phase_max),12*sizeof(unsigned long long)));
  HIPCHK(hipMemcpyFromSymbol(out36+12,HIP_SYMBOL(g_phase_sum),12*sizeof(unsigned long long)));
  HIPCHK(hipMemcpyFromSymbol(out36+24,HIP_SYMBOL(g_phase_arg),12*sizeof(unsigned long long)));
  unsigned long long z[12] = {0};
  HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_phase_max),z,sizeof(z)));
  HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_phase_sum),z,sizeof(z)));
  return CP_OK;
}
#endif
