/*
 * classpro_amd.h -- C ABI of the MI355X-native per-read k-mer classifier (libclasspro_amd.so).
 *
 * Drop-in boundary for ClassPro's per-read hot path.  The reference has no FFI; what a maintainer
 * would bind is (i) the one-time global setup and (ii) the six per-read calls of the thread loop
 * (src/ClassPro.c:229-271).  Each entry point below names the reference interface it replaces.
 * The per-read calls are replaced by *batched* calls over many reads laid out flat in HBM:
 *
 *     seq      char   [seq_off[nreads]]   read bases, ASCII, concatenated, no terminators
 *     seq_off  int64  [nreads+1]          read r = seq[seq_off[r] .. seq_off[r+1])
 *     prof     uint16 [prof_off[nreads]]  k-mer counts (what Fetch_Profile yields), concatenated;
 *                                         base address 16-byte aligned
 *     prof_off int64  [nreads+1]          plen_r = prof_off[r+1]-prof_off[r] = rlen_r-(K-1)
 *     labels   char   [seq_off[nreads]]   out: 'N'*(K-1) then E/H/D/R per k-mer (ClassPro.c:116-119,265-271)
 *
 * All reads of a batch must have rlen >= K (the caller prints shorter reads itself, as
 * ClassPro.c:209-226 does).  CP_MAX_READ_LEN is the reference's limit for FASTX inputs (ClassPro.c:184-187: the
 * command line enforces it there); the library itself takes longer reads (a Dazzler database is sized by its
 * longest read, ClassPro.c:87,110): reads beyond 65535 k-mers go through the sequential classify kernels.
 *
 * Pointers named d_* are DEVICE pointers (HBM); everything else is host memory.  `stream` is a
 * hipStream_t passed as void* (NULL = default stream).  No torch types appear in this ABI.
 * Every function returns CP_OK (0) or a negative CP_E* code; cp_last_error() gives the message.
 * The reference's behaviour on these errors is fprintf(stderr)+exit(1); the CLI mirrors that.
 */
#ifndef CLASSPRO_AMD_H
#define CLASSPRO_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CP_MAX_KMER_CNT 32767          /* src/const.c:38  MAX_KMER_CNT */
#define CP_MAX_READ_LEN 60000          /* src/const.c:55  MAX_READ_LEN (FASTX inputs) */

enum { CP_OK = 0, CP_EINVAL = -1, CP_ENOPEAK = -2, CP_ERCOV = -3, CP_EHIP = -4, CP_EOVERFLOW = -5, CP_ENOMEM = -6 };

/* State codes (src/ClassPro.h:57) and their characters (src/const.c:19). */
enum { CP_ERROR = 0, CP_REPEAT = 1, CP_HAPLO = 2, CP_DIPLO = 3, CP_N_STATE = 4 };

/* Interval record exchanged by the stage entry points: the fields of the reference's `Intvl`
 * (src/ClassPro.h:159-170) in a fixed 48-byte layout. */
typedef struct
  { int32_t  b, e;                 /* [b,e) in profile coordinates */
    uint16_t cb, ce;               /* counts at b and e-1 */
    uint16_t ccb, cce;             /* error-corrected boundary counts (find_rel_intvl) */
    uint8_t  is_rel;
    int8_t   asgn;                 /* CP_ERROR..CP_DIPLO, CP_N_STATE = unclassified */
    uint8_t  _pad[6];
    double   pe;                   /* log P(interval is a sequencing error in this read) or -inf */
    double   peo_b, peo_e;         /* log P(boundary explained by errors in other reads) or -inf */
  } cp_intvl;

const char *cp_last_error(void);
const char *cp_version(void);

/* ------------------------------------------------------------------------------------------
 * One-time global setup (host).
 * ------------------------------------------------------------------------------------------ */

/* Replaces process_global_hist (src/hist.c:28-105) on a loaded FASTK histogram
 * (Load_Histogram/Modify_Histogram, src/libfastk.c:51-147).  `hist` is the on-disk array
 * hist[0..high-low] of <root>.hist (unique counts); coverage_opt is `-c` (0 = estimate).
 * CP_ENOPEAK when no peak count >= 10 exists (the reference exits, hist.c:66-69). */
int cp_hist_covs(const int64_t *hist, int low, int high, int64_t ilowcnt, int64_t ihighcnt,
                 int coverage_opt, int *hcov, int *dcov);

/* Replaces precompute_logfact (src/prob.c:14-19), the GLOBAL_COV / DR_RATIO block
 * (src/ClassPro.c:544-548) and calc_init_thres(NULL) (src/wall.c:167-244): builds the read-only
 * tables on the host and uploads them to the current HIP device.  It also has the device fill a table of
 * logp_trans values (util.c:35-44: a function of |ce-cb| and the integer cov*|e-b| alone; 1 GB of device memory by
 * default, environment CLASSPRO_SKELLAM_TABLE_MB, 0 = no such table; and a second table of the same size with their
 * exponentials, which is what classify_rel's DP step uses of a transition -- CLASSPRO_EXP_TABLE=0: without it) and two small ones (classify_unrel's
 * binomial-test logs, the walk's P(error in): 74 MB) with the code the kernels otherwise run on the spot, so results
 * are the same bits with or without them (CLASSPRO_TABLES=0: none of the three).  The tables depend on READ_LEN / the
 * error model only, so all cp_params of a process that agree on those share one reference-counted copy per device.
 * A table that finds no device memory is left out (slower, same results); cp_params_tables tells which are in use.
 * CP_ERCOV when the repeat threshold exceeds 255 (wall.c:174-177). */
typedef struct cp_params cp_params;
int  cp_params_create(int K, int read_len, int hcov, int dcov, cp_params **out);

/* -M<model_path> (ClassPro.c:451-453, load_himodel wall.c:55-115): as cp_params_create, with the
 * low-complexity error rates pe[t][l] fitted from a HIsim error-model file instead of the default
 * 0.002 l^2 + 0.002.  model_path == NULL is the default model.  The quadratic fit the reference does
 * with GSL is solved in closed form here (no GSL).  cp_load_error_model returns just the fitted table
 * (pe63 = double[3][21], rows HP/DS/TS). */
int  cp_params_create_model(int K, int read_len, int hcov, int dcov, const char *model_path, cp_params **out);
int  cp_load_error_model(const char *model_path, double *pe63);
/* The error model handed in as a table (pe63 = double[3][21], rows HP / DS / TS; entries l = 1 .. 20/(t+1) are used and
 * must lie in (0,1)): for a caller that has the rates already -- e.g. the reference's own load_himodel (wall.c:55-115) with
 * its GSL fit, when bit-identity with THAT fit is wanted -- or from cp_load_error_model.  find_wall (wall.c:570) takes the
 * Error_Model as a parameter in the reference too. */
int  cp_params_create_pe(int K, int read_len, int hcov, int dcov, const double *pe63, cp_params **out);
void cp_params_destroy(cp_params *p);
/* Device bytes of the look-up tables this cp_params uses (0 = that table is not in use: its values are
 * computed on the spot).  skel_bytes counts the logp_trans table AND, when present, the table of its exponentials
 * (same shape: CLASSPRO_SKELLAM_TABLE_MB costs twice its value unless CLASSPRO_EXP_TABLE=0). */
int  cp_params_tables(const cp_params *p, size_t *skel_bytes, size_t *uerr_bytes, size_t *petab_bytes);
/* The device's scalar numerics, for inspection/tests: y[i] = f(x[i]) computed ON THE DEVICE by the functions the kernels
 * call -- fn 0: exp, 1: log (csrc/cp_libm.h: glibc 2.35's routines, the libm the reference's prob.c / class_rel.c /
 * class_unrel.c / wall.c results come from; bit-identical to the host's exp()/log() on an x86-64 host with FMA),
 * 2: sqrt, 3: bessi(n = (int)x2[i], x[i]) (bessel.c:478-521), 4: logp_skellam(k = (int)x2[i], lambda = x[i])
 * (prob.c:41-44).  d_x2 may be null for fn 0-2. */
int  cp_math_eval(int fn, const double *d_x, const double *d_x2, double *d_y, int64_t n, void *stream);
/* Host copies of the tables, for inspection/tests: cov[4]=GLOBAL_COV[E,R,H,D]; cthres is
 * [3][21][256][2][2] = [ctype][l][cout][INIT|FINAL][SELF|OTHERS]; pe is [3][21]; logfact[32768]. */
int  cp_params_export(const cp_params *p, int *cov4, double *dr_ratio, int *cmax, double *hc_erate,
                      uint8_t *cthres, double *pe, double *logfact);

/* Replaces the decoder of Fetch_Profile (src/libfastk.c:1467-1534) for one read's code string
 * (host; the north star keeps FASTK decoding on the host).  Returns the profile length (may
 * exceed `cap`, in which case only `cap` counts were written), or a negative CP_E* code. */
int cp_decode_profile(const uint8_t *code, int64_t len, uint16_t *profile, int cap);

/* Inverse of cp_decode_profile: the FASTK code string of one read's counts (what FastK writes;
 * tooling for tests / synthetic inputs).  `code` needs 2*n+2 bytes; returns the code length. */
int64_t cp_encode_profile(const uint16_t *profile, int n, uint8_t *code, int64_t cap);

/* ------------------------------------------------------------------------------------------
 * Batched device path.
 * ------------------------------------------------------------------------------------------ */
typedef struct cp_workspace cp_workspace;          /* device scratch, grown on demand, reusable */
int  cp_workspace_create(cp_workspace **out);
void cp_workspace_destroy(cp_workspace *ws);
size_t cp_workspace_bytes(const cp_workspace *ws); /* device bytes currently held */

/* Whole hot path for a batch: replaces the body of the read loop, ClassPro.c:229-271
 * (calc_seq_context, find_wall, find_rel_intvl, classify_rel, classify_unrel, label paint).
 * Asynchronous on `stream` except for one small D2H size read-back after the scan stage.
 * A read's labels are a function of that read alone (its bases, its counts, the parameters): they do not depend on
 * the batch it travels in, its place in it, or what lies before or after its counts in memory.  (The reference reads
 * profile[plen] in correct_wall_cnt, wall.c:976-978, when a low-complexity run reaches the end of the read; that
 * cell is defined as 0 here.)
 * Limit: a batch holds at most CP_MAX_BATCH_KMERS k-mer positions (its scratch capacities, up to 16 per position,
 * are summed in 40 bits); a larger one is refused with CP_EINVAL -- split it. */
#define CP_MAX_BATCH_KMERS ((int64_t)1 << 35)
int cp_classify_batch(const cp_params *p, cp_workspace *ws,
                      const char *d_seq, const int64_t *d_seq_off,
                      const uint16_t *d_prof, const int64_t *d_prof_off,
                      int nreads, int64_t total_bases, int64_t total_kmers,
                      char *d_labels, void *stream);

/* Fetch_Profile's decoder on the device: d_codes holds the concatenated code
 * strings of the batch (read r = d_codes[d_code_off[r] .. d_code_off[r+1])), d_prof receives the
 * counts at d_prof_off (plen_r = rlen_r-(K-1), known from the read lengths).  Bit-exact with
 * cp_decode_profile for every code whose counts stay in [0,32767] (FastK caps counts there).  A code
 * that does not expand to exactly plen_r counts (the reference's "rlen != plen+Km1" abort,
 * ClassPro.c:234-237), ends inside a 2-byte token or steps out of [0,32767] is reported as CP_EINVAL by the
 * next cp_workspace_check; d_prof is then unspecified. */
int cp_decode_profiles(cp_workspace *ws, const uint8_t *d_codes, const int64_t *d_code_off,
                       const int64_t *d_prof_off, int nreads, uint16_t *d_prof, void *stream);

/* Dazzler 2-bit bases on the device (Load_Read(db,i,buf,2), DB.c:1232-1298 with Uncompress_Read / Upper_Read,
 * DB.c:342-381): read r's COMPRESSED_LEN(rlen_r) = (rlen_r+3)/4 bytes start at d_packed[d_pack_off[r]] (the bytes
 * of the .bps file at DAZZ_READ.boff), 4 bases per byte, first base in the top two bits; d_seq receives
 * 'A' 'C' 'G' 'T' at d_seq_off[r].  A database input then crosses PCIe at 0.25 B/base. */
int  cp_unpack_bases(const uint8_t *d_packed, const int64_t *d_pack_off, const int64_t *d_seq_off,
                     int nreads, char *d_seq, void *stream);

/* 2-bit payloads across PCIe for FASTX inputs too.  Bases: cp_pack_bases (host) packs one read in
 * the layout cp_unpack_bases expands, when it holds upper-case A, C, G, T only (returns 1; 0 = another letter: send the
 * batch as characters -- calc_seq_context compares raw characters, context.c:8-108).  Labels: cp_pack_labels (device)
 * turns the label string of every read into 2-bit codes, four per byte, first label in the top bits: ctos (const.c:21-36:
 * N, E -> 0, R -> 1, H -> 2, D -> 3) + Compress_Read (gene_core.c:235-254), i.e. exactly the read's payload of the
 * .class.data track (ClassPro.c:291-300); read r's (rlen_r+3)/4 bytes start at d_packed[d_pack_off[r]] (the offsets of
 * cp_unpack_bases).  cp_unpack_labels (host) is the inverse for one read: K-1 'N', then E/R/H/D (stoc, const.c:19).
 * A batch then crosses PCIe at about 0.52 B/base in (bases + FASTK codes) and 0.25 B/base out instead of 1.27 and 1. */
int cp_pack_bases(const char *seq, int rlen, uint8_t *packed);
/* cp_pack_bases for every read of a host batch on `nthreads` host threads: read r's (rlen_r+3)/4 bytes go to
 * packed[pack_off[r]]; 1 = every read packed, 0 = a read holds another letter (send the batch as characters). */
int cp_pack_bases_batch(const char *seq, const int64_t *seq_off, int nreads, uint8_t *packed, const int64_t *pack_off,
                        int nthreads);
int cp_pack_labels(const char *d_labels, const int64_t *d_seq_off, const int64_t *d_pack_off, int nreads,
                   uint8_t *d_packed, void *stream);
int cp_unpack_labels(const uint8_t *packed, int rlen, int K, char *labels);

/* Labels as RUNS, the smallest form in which a batch's result crosses PCIe (~0.05 B/base): the label string of a read
 * (ClassPro.c:265-271: K-1 'N', then the class of every interval over its k-mers) is K-1 'N' followed by runs; run j
 * covers label positions [ends[j-1], ends[j]) (ends[-1] = K-1) with the character cls[j] (stoc, const.c:19).  After
 * cp_classify_batch -- or cp_run_stages(..., CP_STAGE_CLASS_ALL, ...), which skips painting the 1 B/base label string
 * altogether -- cp_label_runs writes the runs of read r at index d_cap_off[r] .. d_cap_off[r]+d_nruns[r] of d_ends /
 * d_cls; the arrays take cp_label_runs_capacity(ws) entries (a loose bound: the interval capacity of the batch, about
 * 0.01 per base), d_nruns nreads, d_cap_off nreads+1 entries.  cp_expand_label_runs (host) rebuilds one read's string. */
int64_t cp_label_runs_capacity(const cp_workspace *ws);
int cp_label_runs(const cp_params *p, cp_workspace *ws, int32_t *d_ends, uint8_t *d_cls, int32_t *d_nruns, int64_t *d_cap_off,
                  void *stream);
int cp_expand_label_runs(const int32_t *ends, const uint8_t *cls, int nruns, int rlen, int K, char *labels);

/* -s: replaces find_seeds (src/seed.c:966-1032; call site ClassPro.c:281-282) for every read of a batch that
 * cp_classify_batch has labelled.  d_labels is that call's output; d_seeds[total_bases] receives, in the same layout,
 * 'N' for the first K-1 bases of a read and per k-mer 'E' (no seed) or the class of the seed, 'H' / 'D' / 'R' (a seed
 * inside a repetitive stretch) -- the values the .class.data track carries under -s (ClassPro.c:293).  The repeat-mask
 * intervals of anno_repeat (seed.c:482-566, the .rep.data track) stay in `ws`: cp_get_rep_masks copies, per read, their
 * number and the (b,e) pairs in read coordinates (read r's pairs start at pairs[2*cap_off[r]]; cp_rep_masks_capacity
 * gives the pairs' total capacity).  Defined behaviour: the reference's masked-interval array starts as zeros at
 * every read (it is read one slot past its live part, seed.c:141,161-166). */
int cp_find_seeds_batch(const cp_params *p, cp_workspace *ws,
                        const char *d_seq, const int64_t *d_seq_off,
                        const uint16_t *d_prof, const int64_t *d_prof_off, const char *d_labels,
                        int nreads, int64_t total_bases, int64_t total_kmers,
                        char *d_seeds, void *stream);
int cp_get_rep_masks(cp_workspace *ws, int32_t *count, int64_t *cap_off, int32_t *pairs, int64_t capacity);
int64_t cp_rep_masks_capacity(const cp_workspace *ws);

/* Waits for the work queued on `ws` and returns CP_EOVERFLOW if, in ANY call on `ws` since the previous check, a read
 * needed more E-interval / interval / seed scratch than its capacity (the reference aborts likewise: "# E-intvls >=
 * plen", src/wall.c:783-788), CP_EINVAL for a bad code string in cp_decode_profiles.  The device-side flags are sticky
 * and are cleared by this call only.  Call after cp_classify_batch before trusting the labels. */
int cp_workspace_check(cp_workspace *ws);

/* Stage entry points (the reference's per-read internal contract, batched).  They run the
 * pipeline up to and including the named stage and keep the results in `ws`:
 *   CP_STAGE_SCAN      candidate scan of find_wall (wall.c:590-607): bitmap of wall candidates
 *   CP_STAGE_WALL      find_wall       (src/wall.c:570-958)
 *   CP_STAGE_REL       find_rel_intvl  (src/wall.c:1016-1051)
 *   CP_STAGE_CLASS_REL classify_rel    (src/class_rel.c:871-963)
 *   CP_STAGE_CLASS_ALL classify_unrel  (src/class_unrel.c:248-300)
 *   CP_STAGE_LABELS    label paint     (src/ClassPro.c:265-271)   == cp_classify_batch */
enum { CP_STAGE_SCAN = 1, CP_STAGE_WALL = 2, CP_STAGE_REL = 3, CP_STAGE_CLASS_REL = 4,
       CP_STAGE_CLASS_ALL = 5, CP_STAGE_LABELS = 6 };
int cp_run_stages(const cp_params *p, cp_workspace *ws,
                  const char *d_seq, const int64_t *d_seq_off,
                  const uint16_t *d_prof, const int64_t *d_prof_off,
                  int nreads, int64_t total_bases, int64_t total_kmers,
                  char *d_labels, int last_stage, void *stream);

/* Read-back of stage results of the last cp_run_stages/cp_classify_batch on `ws` (host buffers;
 * synchronises the stream).  counts: n_intvl[nreads], n_rel[nreads]; offsets into the flat arrays
 * are the exclusive prefix sums of the per-read capacities returned in cap_off[nreads+1].
 * After a whole-path call (cp_classify_batch = CP_STAGE_LABELS) the interval records come back complete (the final classes
 * included), but the copies of the reliable intervals and the forward / backward assignments do not exist -- that path
 * hands classify_rel 24-byte records and the label paint 4-byte (end, class) words instead: `rintvl`
 * is returned zeroed.  Stop at CP_STAGE_REL .. CP_STAGE_CLASS_ALL for them.  (CLASSPRO_COMPACT_REL=0: the full records
 * on every call.) */
int cp_get_counts(cp_workspace *ws, int32_t *n_cand, int32_t *n_intvl, int32_t *n_rel, int64_t *cap_off);
int cp_get_intervals(cp_workspace *ws, cp_intvl *intvl, cp_intvl *rintvl, int64_t capacity);
int cp_get_rel_asgn(cp_workspace *ws, int8_t *fw, int8_t *bw, int64_t capacity);   /* after a run that stopped at CP_STAGE_CLASS_REL or CP_STAGE_CLASS_ALL */
int cp_get_bitmap(cp_workspace *ws, uint64_t *words, int64_t nwords);

/* calc_seq_context (src/context.c:8-108), batched and dense: d_lctx/d_rctx are [total_bases][3]
 * uint8 indexed by read position (the `_lctx`/`rctx` buffers of ClassPro.c:136-142).  The hot path
 * evaluates contexts on demand and never materialises these arrays; this entry point exists for
 * the reference's own call and for parity tests. */
int cp_seq_context(const char *d_seq, const int64_t *d_seq_off, int nreads, int64_t total_bases,
                   uint8_t *d_lctx, uint8_t *d_rctx, void *stream);

/* Profile-scan kernel alone (the HBM-roofline kernel): d_bitmap holds ceil(total_kmers/64)+1 words. */
int cp_scan_candidates(const cp_params *p, const uint16_t *d_prof, int64_t total_kmers,
                       uint64_t *d_bitmap, void *stream);

#ifdef __cplusplus
}
#endif
#endif
