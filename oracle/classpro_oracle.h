/*
 * classpro_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, sequential) of the per-read k-mer classification path of
 * yoshihikosuzuki/ClassPro.  It exists to CHECK the HIP product in classpro_amd/; it is
 * never linked, imported or called by the product.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may use it.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - bessel / prob / util / context / class_rel / class_unrel / hist peak search /
 *     FASTK profile decoding are PINNED against the reference's own code compiled from
 *     /root/reference/src (oracle/_ref, built by oracle/Makefile) and the golden vectors
 *     under tests/golden/ generated from that build (oracle/gen_golden.py).
 *   - find_wall / find_rel_intvl (reference src/wall.c:245-1051) are PINNED since round 5 against the reference's
 *     own text: the GSL-free line ranges of wall.c are compiled into oracle/_ref (oracle/Makefile `ref`,
 *     oracle/ref_driver.c), golden vectors tests/golden/wall.npz and labels.npz, live legs in
 *     tests/test_oracle_wall.py.
 *   - calc_init_thres (wall.c:167-243) and the -M fit (wall.c:11-115) are **parity unpinned**: they reach
 *     load_himodel -> GSL (wall.c:9-115), GSL is absent from this image and its tarball is a missing blob, and no
 *     stand-in header was written.  The threshold table is pinned by exact integer arithmetic instead
 *     (tests/test_first_principles.py).
 */
#ifndef CLASSPRO_ORACLE_H
#define CLASSPRO_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CPO_MAX_KMER_CNT 32767          /* const.c:38 */
#define CPO_MAX_READ_LEN 60000          /* const.c:55 */

enum { CPO_ERROR = 0, CPO_REPEAT = 1, CPO_HAPLO = 2, CPO_DIPLO = 3, CPO_N_STATE = 4 }; /* ClassPro.h:57 */
enum { CPO_HP = 0, CPO_DS = 1, CPO_TS = 2 };                                           /* ClassPro.h:58 */
enum { CPO_SELF = 0, CPO_OTHERS = 1 };                                                 /* ClassPro.h:59 */
enum { CPO_DROP = 0, CPO_GAIN = 1 };                                                   /* ClassPro.h:60 */
enum { CPO_INIT = 0, CPO_FINAL = 1 };                                                  /* ClassPro.h:122 */

/* Interval record: same fields as the reference's Intvl (ClassPro.h:159-170), own layout (48 B). */
typedef struct
  { int32_t  b, e;
    uint16_t cb, ce, ccb, cce;
    uint8_t  is_rel;
    int8_t   asgn;
    uint8_t  _pad[6];
    double   pe;
    double   peo_b, peo_e;
  } cpo_intvl;

/* Global read-only parameters (ClassPro.c:27-32,536-554; wall.c:117-244; prob.c:12-19). */
typedef struct
  { int      K;
    int      read_len;                 /* READ_LEN (-r) */
    int      cov[4];                   /* GLOBAL_COV[E,R,H,D] */
    double   dr_ratio;                 /* DR_RATIO */
    int      cmax;                     /* CMAX */
    double   hc_erate;                 /* HC_ERATE */
    int      lmax[3];
    double   pe[3][21];
    uint8_t  cthres[3][21][256][2][2]; /* [ctype][l][cout][thresT][etype] */
    double   logfact[CPO_MAX_KMER_CNT+1];
  } cpo_params;

cpo_params *cpo_params_new(int K, int read_len, int hcov, int dcov);
/* -M<model>: wall.c:55-115 (fit restated without GSL: parity unpinned for the fit itself) */
int cpo_load_himodel(const char *path, double *pe63);
cpo_params *cpo_params_new_model(int K, int read_len, int hcov, int dcov, const double *pe63);
void        cpo_params_free(cpo_params *p);
const uint8_t *cpo_params_cthres(const cpo_params *p);     /* flat [3][21][256][2][2] */
const double  *cpo_params_logfact(const cpo_params *p);
const double  *cpo_params_pe(const cpo_params *p);         /* flat [3][21] */
const int     *cpo_params_lmax(const cpo_params *p);       /* [3] */
void        cpo_params_scalars(const cpo_params *p, int *cov4, double *dr_ratio, int *cmax, double *hc_erate);

/* numeric primitives (prob.c, bessel.c, util.c) */
double cpo_bessi(int n, double x);
double cpo_logp_poisson(const cpo_params *p, int k, int lambda);
double cpo_logp_skellam(int k, double lambda);
double cpo_logp_binom(const cpo_params *p, int k, int n, double pr);
double cpo_binom_test_g(const cpo_params *p, int k, int n, double pe, int exact);
double cpo_logp_trans(const cpo_params *p, int b, int e, int cb, int ce, int cov);

/* histogram -> (H,D) coverage (hist.c:28-105 + libfastk.c:22-147).  `hist` is the on-disk
 * array hist[0..high-low] of a FASTK .hist file (unique counts).  Returns 0, or 1 when no
 * peak >= 10 exists (the reference exits). */
int cpo_hist_covs(const int64_t *hist, int low, int high, int64_t ilowcnt, int64_t ihighcnt,
                  int coverage_opt, int *hcov, int *dcov);

/* FASTK profile decode (libfastk.c:1467-1534) from an in-memory code string. */
int cpo_decode_profile(const uint8_t *code, int64_t len, uint16_t *profile, int cap);

/* sequence context (context.c:8-108).  lctx/rctx are [rlen][3] uint8, index = read position. */
void cpo_seq_context(const char *seq, int rlen, uint8_t *lctx, uint8_t *rctx);

/* wall.c:570-958.  lctx/rctx as produced by cpo_seq_context for the read (rlen = plen+K-1). */
int cpo_find_wall(const cpo_params *p, const uint16_t *profile, int plen,
                  const uint8_t *lctx, const uint8_t *rctx, cpo_intvl *intvl, int cap);
/* wall.c:960-1051 */
int cpo_find_rel_intvl(const cpo_params *p, cpo_intvl *intvl, int N, cpo_intvl *rintvl,
                       const uint16_t *profile, int plen, const uint8_t *lctx, const uint8_t *rctx);
/* class_rel.c:871-963; optional fw_out/bw_out[M] receive the two directional assignments */
void cpo_classify_rel(const cpo_params *p, cpo_intvl *rintvl, int M, cpo_intvl *intvl, int N, int plen,
                      int8_t *fw_out, int8_t *bw_out);
/* class_unrel.c:248-300 */
void cpo_classify_unrel(const cpo_params *p, cpo_intvl *intvl, int N);

/* whole read (ClassPro.c:229-271): labels[rlen] = 'N'*(K-1) + E/H/D/R per k-mer.
 * Returns N (number of intervals); optional outputs may be NULL. */
int cpo_classify_read(const cpo_params *p, const char *seq, int rlen, const uint16_t *profile,
                      char *labels, cpo_intvl *intvl_out, int cap, int *M_out);

/* batch: reads r has bases seq[seq_off[r]..seq_off[r+1]) and counts prof[prof_off[r]..prof_off[r+1]);
 * labels uses the seq offsets.  nthreads pthreads over contiguous read ranges. */
void cpo_classify_batch(const cpo_params *p, const char *seq, const int64_t *seq_off,
                        const uint16_t *prof, const int64_t *prof_off, int nreads,
                        char *labels, int nthreads);

/* -s seed path (src/seed.c), oracle/classpro_oracle_seed.c.  cls[plen] = the read's k-mer labels (E/H/D/R);
 * sasgn[plen] receives 'E' (not a seed) or the seed's class 'H'/'D'/'R' (seed.c:1007-1015); returns the number of
 * repeat-mask intervals, rep_pairs = (b,e) in read coordinates (seed.c:531-566).  mintvl starts as zeros. */
int  cpo_find_seeds(const char *seq, const char *cls, const uint16_t *profile, int plen, int K,
                    int *sasgn, int *rep_pairs, int rep_cap);
void cpo_kmer_hash(const char *seq, int plen, int K, int *hash);      /* seed.c:28-55 */

#ifdef __cplusplus
}
#endif
#endif
