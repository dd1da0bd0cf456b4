/*
 * ref_driver.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Thin export layer over the reference's OWN source files, compiled where they lie under
 * /root/reference/src (nothing is copied into this repo).  ClassPro is a unity build whose hot-path
 * functions are `static inline`, so the only way to call them is to #include the .c files into one
 * translation unit, exactly as ClassPro.c:16-25 does.  This driver includes every hot-path file that
 * compiles from the image's own headers:
 *     const.c  prob.c(+bessel.c)  util.c  hist.c  context.c  class_rel.c  class_unrel.c  seed.c(+nthash.h, kdq.h)
 * and is linked with libfastk.c (which includes gene_core.c), DB.c and QV.c (DB.c:38 defines Prog_Name),
 * the same objects the reference links into ClassPro (src/Makefile:23-24).
 *
 * NOT included: wall.c (find_wall / find_rel_intvl / calc_init_thres).  It has
 * `#include <gsl/gsl_multifit.h>` (wall.c:9); GSL is not installed and src/gsl-2.7.tar.gz is a
 * missing blob, so wall.c is unbuildable in this image.  No stand-in header is provided.
 *
 * The globals below are the ones ClassPro.c:27-32 defines for the unity build.
 * Output: oracle/_ref/libclasspro_ref.so (git-ignored; travels to the GPU box with gpurun).
 */
#include <stdio.h>
#include <stdlib.h>
#include <stdbool.h>
#include <string.h>
#include <math.h>
#include <float.h>
#include <limits.h>

#include "ClassPro.h"

bool  VERBOSE;
int   READ_LEN;
bool  IS_DB;
bool  IS_DAM;
cnt_t GLOBAL_COV[N_STATE];

#include "const.c"
#include "prob.c"
#include "util.c"
#include "hist.c"
#include "context.c"
#include "class_rel.c"
#include "class_unrel.c"
#include "seed.c"

/* ---- setup (what ClassPro.c:536-548 does, minus calc_init_thres which lives in wall.c) ---- */
void ref_setup(int read_len, int hcov, int dcov)
{ static int done = 0;
  VERBOSE = false; IS_DB = false; IS_DAM = false;
  READ_LEN = read_len;
  if (!done) { precompute_logfact(); done = 1; }
  GLOBAL_COV[HAPLO]  = hcov;
  GLOBAL_COV[DIPLO]  = dcov;
  GLOBAL_COV[ERROR]  = 1;
  GLOBAL_COV[REPEAT] = plus_sigma(GLOBAL_COV[DIPLO],N_SIGMA_RCOV);
  DR_RATIO = 1.+(double)N_SIGMA_R*(1./sqrt(GLOBAL_COV[DIPLO]));
}
void ref_globals(int *cov4, double *dr_ratio)
{ for (int i = 0; i < 4; i++) cov4[i] = GLOBAL_COV[i];
  *dr_ratio = DR_RATIO;
}
const double *ref_logfact(void) { return logfact; }

/* ---- primitives ---- */
double ref_bessi(int n, double x)                         { return bessi(n,x); }
double ref_logp_poisson(int k, int lambda)                { return logp_poisson((cnt_t)k,lambda); }
double ref_logp_skellam(int k, double lambda)             { return logp_skellam(k,lambda); }
double ref_logp_binom(int k, int n, double p)             { return logp_binom((cnt_t)k,(cnt_t)n,p); }
double ref_binom_test_g(int k, int n, double pe, int ex)  { return binom_test_g((cnt_t)k,(cnt_t)n,pe,ex); }
double ref_logp_trans(int b, int e, int cb, int ce, int cov) { return logp_trans(b,e,cb,ce,(cnt_t)cov); }
double ref_p_errorin(int e, double erate, int cout, int cin) { return p_errorin(e,erate,(cnt_t)cout,(cnt_t)cin); }
int    ref_plus_sigma(int cnt, int n)                     { return plus_sigma((cnt_t)cnt,n); }

/* ---- histogram: process_global_hist on <root>.hist (hist.c:28) ---- */
void ref_hist_covs(char *fk_root, int coverage, int *hcov, int *dcov)
{ process_global_hist(fk_root,coverage);
  *hcov = lambda_prior[0];
  *dcov = lambda_prior[1];
}

/* ---- FASTK profiles: Open_Profiles / Fetch_Profile (libfastk.c:1267,1414) ---- */
void *ref_open_profiles(char *root)                       { return Open_Profiles(root); }
void  ref_free_profiles(void *P)                          { Free_Profiles((Profile_Index *)P); }
int   ref_profiles_nreads(void *P)                        { return ((Profile_Index *)P)->nreads; }
int   ref_profiles_kmer(void *P)                          { return ((Profile_Index *)P)->kmer; }
int   ref_fetch_profile(void *P, long long id, int cap, unsigned short *out)
{ return Fetch_Profile((Profile_Index *)P,(int64)id,cap,out); }

/* ---- context.c; buffers laid out as the caller at ClassPro.c:136-142 does ---- */
void ref_seq_context(char *seq, int rlen, unsigned char *lctx_out, unsigned char *rctx_out)
{ Seq_Ctx *_lctx = Malloc((rlen+1)*sizeof(Seq_Ctx),"l");
  Seq_Ctx *rctx  = Malloc((rlen+1)*sizeof(Seq_Ctx),"r");
  memset(_lctx,0,(rlen+1)*sizeof(Seq_Ctx));
  memset(rctx,0,(rlen+1)*sizeof(Seq_Ctx));
  _lctx[0][HP] = 1;
  _lctx[0][DS] = _lctx[0][TS] = _lctx[1][TS] = 0;
  calc_seq_context(_lctx,rctx,seq,rlen);
  memcpy(lctx_out,_lctx,(size_t)rlen*3);
  memcpy(rctx_out,rctx,(size_t)rlen*3);
  free(_lctx); free(rctx);
}

/* ---- interval exchange record (same field set as oracle's cpo_intvl, 48 B) ---- */
typedef struct
  { int b, e;
    unsigned short cb, ce, ccb, cce;
    unsigned char is_rel;
    signed char asgn;
    unsigned char _pad[6];
    double pe, peo_b, peo_e;
  } xintvl;

static void x2i(const xintvl *x, Intvl *I)
{ memset(I,0,sizeof(Intvl));
  I->b = x->b; I->e = x->e; I->cb = x->cb; I->ce = x->ce; I->ccb = x->ccb; I->cce = x->cce;
  I->is_rel = x->is_rel; I->pe = x->pe; I->pe_o.b = x->peo_b; I->pe_o.e = x->peo_e; I->asgn = x->asgn;
}
static void i2x(const Intvl *I, xintvl *x)
{ x->b = I->b; x->e = I->e; x->cb = I->cb; x->ce = I->ce; x->ccb = I->ccb; x->cce = I->cce;
  x->is_rel = I->is_rel; x->pe = I->pe; x->peo_b = I->pe_o.b; x->peo_e = I->pe_o.e; x->asgn = I->asgn;
}

/* classify_rel (class_rel.c:871) + classify_unrel (class_unrel.c:248) on caller-supplied intervals.
 * stage: 1 = classify_rel only, 2 = classify_rel then classify_unrel. */
void ref_classify(xintvl *rx, int M, xintvl *ix, int N, int plen, int stage)
{ int cap = (M > N ? M : N)+8;
  Rel_Arg *arg   = alloc_rel_arg(cap);
  Intvl *rintvl  = Malloc(sizeof(Intvl)*cap,"r");
  Intvl *intvl   = Malloc(sizeof(Intvl)*cap,"i");
  for (int i = 0; i < M; i++) x2i(&rx[i],&rintvl[i]);
  for (int i = 0; i < N; i++) x2i(&ix[i],&intvl[i]);
  classify_rel(arg,rintvl,M,intvl,N,plen);
  if (stage >= 2)
    classify_unrel(intvl,N);
  for (int i = 0; i < M; i++) i2x(&rintvl[i],&rx[i]);
  for (int i = 0; i < N; i++) i2x(&intvl[i],&ix[i]);
  free_rel_arg(arg,cap);
  free(rintvl); free(intvl);
}

/* directional runs, to expose the fw / bw assignments separately (class_rel.c:623,737) */
void ref_classify_rel_dir(xintvl *rx, int M, int plen, int forward, signed char *asgn_out, double *hdrr)
{ int cap = M+8;
  Rel_Arg *arg  = alloc_rel_arg(cap);
  Intvl *rintvl = Malloc(sizeof(Intvl)*cap,"r");
  for (int i = 0; i < M; i++) x2i(&rx[i],&rintvl[i]);
  Iter_Rel r = forward ? classify_rel_fw(arg,rintvl,M,plen) : classify_rel_bw(arg,rintvl,M,plen);
  for (int i = 0; i < M; i++) asgn_out[i] = r.asgn[i];
  *hdrr = r.hdrr;
  free_rel_arg(arg,cap);
  free(rintvl);
}

/* ---- DAZZ_DB access through the reference's own DB.c: pins the test-side database writer
 *      (classpro_amd/dazz.py) and the track files the product writes ------------------------------ */
static DAZZ_DB  g_db;
static int      g_db_open = 0;
static char    *g_db_buf = NULL;

int ref_db_open(const char *path)
{ if (g_db_open) { Close_DB(&g_db); g_db_open = 0; }
  int st = Open_DB((char *)path,&g_db);
  if (st < 0) return -1;
  g_db_open = 1;
  g_db_buf = New_Read_Buffer(&g_db);
  return st;                                  /* 0 = .db, 1 = .dam */
}
int ref_db_nreads(void) { return g_db.nreads; }
int ref_db_maxlen(void) { return g_db.maxlen; }
int ref_db_read(int i, char *out, int *origin, int *fpulse, long long *coff)   /* Load_Read(db,i,buf,2) */
{ if (!g_db_open || i < 0 || i >= g_db.nreads) return -1;
  DAZZ_READ *r = g_db.reads+i;
  Load_Read(&g_db,i,g_db_buf,2);
  memcpy(out,g_db_buf,r->rlen);
  *origin = r->origin; *fpulse = r->fpulse; *coff = r->coff;
  return r->rlen;
}
/* Open_Track + Load_All_Track_Data: lengths per read into alen[nreads], concatenated data into data */
long long ref_db_track(const char *name, int *alen, unsigned char *data, long long cap)
{ if (!g_db_open) return -1;
  DAZZ_TRACK *t = Open_Track(&g_db,(char *)name);
  if (t == NULL) return -1;
  if (t->data == NULL)                        /* a mask-less header-only track opens with data == NULL */
    { Close_Track(&g_db,t); return 0; }
  Load_All_Track_Data(t);
  long long tot = 0;
  int64 *a = (int64 *)t->anno;
  for (int i = 0; i < t->nreads; i++)
    { alen[i] = t->alen[i];
      if (tot+t->alen[i] <= cap)
        memcpy(data+tot,(char *)t->data+a[i],t->alen[i]);
      tot += t->alen[i];
    }
  Close_Track(&g_db,t);
  return tot;
}
void ref_db_close(void) { if (g_db_open) { Close_DB(&g_db); g_db_open = 0; } }

/* ---- seed.c: find_seeds (seed.c:966-1032) for one read, with the buffers of ClassPro.c:120-134.
 *      `mintvl` is zeroed before the call: the reference searches and sorts one slot past the live part of this
 *      array (seed.c:141,161-166), i.e. it reads whatever an earlier read of the same thread left there; the
 *      defined behaviour of this build is "the array is all zeros at the start of every read" (within a read
 *      the three passes see each other's leftovers exactly as the reference code does).
 *      Outputs: sasgn[plen] ('E','H','D','R' per k-mer, seed.c:1007-1015), the repeat-mask intervals written to
 *      the .rep data track (int pairs, read coordinates, seed.c:531-566) and the canonical hashes. ---- */
int ref_find_seeds(const char *seq, const char *pasgn, const unsigned short *profile, int plen, int K,
                   int *sasgn, int *hash_out, int *rep_pairs, int rep_cap)
{ kdq_t(hmer_t) *Q = kdq_init(hmer_t);
  seg_t   *cprofile = Malloc((plen+1)*sizeof(seg_t),"c");
  int     *hash     = Malloc((plen+1)*sizeof(int),"h");
  intvl_t *mintvl   = Malloc((plen+2)*sizeof(intvl_t),"m");
  memset(mintvl,0,(plen+2)*sizeof(intvl_t));
  char *abuf = NULL, *dbuf = NULL; size_t alen = 0, dlen = 0;
  FILE *ranno = open_memstream(&abuf,&alen), *rdata = open_memstream(&dbuf,&dlen);
  int64 ridx = 0;
  find_seeds(Q,seq,pasgn,profile,cprofile,hash,sasgn,mintvl,plen,K,ranno,rdata,&ridx);
  fclose(ranno); fclose(rdata);
  int npair = (int)(dlen/(2*sizeof(int)));
  if (rep_pairs)
    memcpy(rep_pairs,dbuf,sizeof(int)*2*(npair < rep_cap ? npair : rep_cap));
  if (hash_out) memcpy(hash_out,hash,sizeof(int)*plen);
  free(abuf); free(dbuf);
  kdq_destroy(hmer_t,Q);
  free(cprofile); free(hash); free(mintvl);
  return npair;
}
