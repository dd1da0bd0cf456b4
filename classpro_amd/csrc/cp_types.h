// cp_types.h -- constants and the read-only parameter block shared by host setup and HIP kernels.
// Constants follow the reference's src/const.c:46-73; nothing here is executed on the CPU by the
// product path (the host only fills the tables once, as ClassPro.c:536-554 does).
#pragma once
#include <stdint.h>
#include "../../include/classpro_amd.h"

#define CP_HD __host__ __device__ __forceinline__

enum { CP_HP = 0, CP_DS = 1, CP_TS = 2 };          // ClassPro.h:58  Ctype
enum { CP_SELF = 0, CP_OTHERS = 1 };               // ClassPro.h:59  Etype
enum { CP_DROP = 0, CP_GAIN = 1 };                 // ClassPro.h:60  Wtype
enum { CP_INIT = 0, CP_FINAL = 1 };                // ClassPro.h:122 ThresT

// const.c:55-73
#define CP_N_SIGMA_RCOV    5
#define CP_MAX_N_LC        20
#define CP_MAX_N_HC        5
#define CP_MIN_CNT_CHANGE  3
#define CP_MAX_CNT_CHANGE  5
#define CP_PE_THRES_INIT_S   0.001
#define CP_PE_THRES_INIT_O   0.05
#define CP_PE_THRES_FINAL    1e-5          // PE_THRES[FINAL][SELF] == PE_THRES[FINAL][OTHERS]
#define CP_THRES_DIFF_EO   (-23.025851)
#define CP_THRES_DIFF_REL  (-9.210340)
#define CP_OFFSET          1000
#define CP_N_SIGMA_R       2
#define CP_R_LOGP          (-10.)
#define CP_E_PO_BASE       (-10.)
#define CP_PE_MEAN         0.01
#define CP_MULT_WINDOW     200             // wall.c:773,817

// wall bit flags, wall.c:264-269
#define CP_W_WALL_S    0x01
#define CP_W_WALL_O    0x10
#define CP_W_PAIRED_S  0x02
#define CP_W_PAIRED_O  0x20
#define CP_W_PAIRED_M  0x40
#define CP_W_ERROR     0x80

// Read-only tables, one copy in HBM (kernels take a pointer).  Every log() of a *constant* the
// reference evaluates per call (log(pe), log(1-pe), log(lambda)) is tabulated here by the host, so
// the device only runs exp/log on data-dependent values.
struct cp_dev_params
  { int     K;
    int     read_len;                      // READ_LEN (-r)
    int     cov[4];                        // GLOBAL_COV[E,R,H,D]
    int     cmax;                          // CMAX = GLOBAL_COV[REPEAT], wall.c:178
    int     lmax[3];                       // emodel[t].lmax
    double  dr_ratio;                      // DR_RATIO, ClassPro.c:548
    double  hc_erate;                      // HC_ERATE = pe[HP][1], wall.c:180
    double  hc_lpe, hc_l1mpe;              // log(HC_ERATE), log(1-HC_ERATE)
    double  u_lpe, u_l1mpe;                // log(0.1), log(0.9): class_unrel.c:137 max_erate
    double  r_lp, r_l1mp;                  // log(1-PE_MEAN), log(1-(1-PE_MEAN)): class_rel.c:185
    double  log_pe_final;                  // log(PE_THRES[FINAL][SELF]): wall.c:1018
    double  pe[3][21];                     // emodel[t].pe[l]
    double  lpe[3][21], l1mpe[3][21];      // log(pe), log(1-pe)
    uint8_t cthres[3][21][256][2][2];      // [ctype][l][cout][INIT|FINAL][SELF|OTHERS], wall.c:190-224
    double  logfact[CP_MAX_KMER_CNT+1];    // prob.c:12-19
    double  logint[CP_MAX_KMER_CNT+1];     // log((double)n), n >= 1 (logp_poisson's log(lambda))
    // logp_trans (util.c:35-44) depends on its five arguments only through |ce-cb| and the integer cov*|e-b|: values for
    // |ce-cb| <= skel_kmax and cov*|e-b| <= skel_cdmax, computed once on the device by the same code
    // (skel[cd*(skel_kmax+1)+k]); NULL = always computed on the spot
    const double *skel;
    const double *eskel;                   // exp() of the same entries (same sizes), for the DP step of k_classify_rel_grp; NULL = none
    int     skel_kmax;
    long long skel_cdmax;
    // classify_unrel's log(binom_test(count | estimated count, 0.1)) (class_unrel.c:137-147) is a function of two
    // integers: uerr[est*(uerr_max+1)+c] for c <= est <= uerr_max, filled on the device by the same code; NULL = none
    const double *uerr;
    int     uerr_max;
    // the walk's P(error in) (util.c:46-55 -> prob.c:76-112) for one of the 63 error rates pe[t][l], error type e and
    // counts cin <= cout <= pe_cmax: petab[((e*63+t*21+l)*(pe_cmax+1)+cout)*(pe_cmax+1)+cin], same code; NULL = none
    const double *petab;
    int     pe_cmax;
  };

// E-/O-interval, ClassPro.h:153-157
struct cp_eintvl { int b, e; double pe; };
