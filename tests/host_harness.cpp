// host_harness.cpp -- TEST-ONLY.  Compiles the product's scalar device functions (cp_math.h, cp_ctx.h,
// cp_wall.h, cp_class.h) and host setup (cp_host_setup.h) for the host, with a plain sequential
// orchestration, so their logic can be unit-tested against the oracle in a GPU-less container.
// This library is never loaded by the product (classpro_amd/), only by tests/.
#include <cstring>
#include <cmath>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "../classpro_amd/csrc/cp_host_setup.h"
#include "../classpro_amd/csrc/cp_wall.h"
#include "../classpro_amd/csrc/cp_class.h"
#include "../classpro_amd/csrc/cp_seed.h"

extern "C" {

void hh_seq_context(const char *seq, int rlen, unsigned char *lctx, unsigned char *rctx)
{ for (int i = 0; i < rlen; i++)
    for (int t = 0; t < 3; t++)
      { lctx[i*3+t] = (unsigned char)cp_lctx(seq,rlen,i,t);
        rctx[i*3+t] = (unsigned char)cp_rctx(seq,rlen,i,t);
      }
}

// the same from the three-at-once forms (cp_lctx3 / cp_rctx3: one 8-base word for the three contexts of a position) and
// from the per-base scans alone; returns how many (position, direction) pairs the words decided without a scan
long long hh_seq_context3(const char *seq, int rlen, unsigned char *lctx, unsigned char *rctx, unsigned char *lscan, unsigned char *rscan)
{ long long hits = 0;
  for (int i = 0; i < rlen; i++)
    { int l3[3], r3[3];
      cp_lctx3(seq,rlen,i,l3);
      cp_rctx3(seq,rlen,i,r3);
      for (int t = 0; t < 3; t++)
        { lctx[i*3+t] = (unsigned char)l3[t];
          rctx[i*3+t] = (unsigned char)r3[t];
          lscan[i*3+t] = (unsigned char)cp_lctx_scan(seq,rlen,i,t);
          rscan[i*3+t] = (unsigned char)cp_rctx_scan(seq,rlen,i,t);
        }
      uint64_t D = 0;
      int nv = cp_seq_dirword(seq,rlen,i,-1,&D);
      if (nv && cp_word_hp(D,nv) >= 0 && cp_word_ds(D,nv) >= 0 && cp_word_ts(D,nv) >= 0) hits++;
      nv = cp_seq_dirword(seq,rlen,i,+1,&D);
      if (nv && cp_word_hp(D,nv) >= 0 && cp_word_ds(D,nv) >= 0 && cp_word_ts(D,nv) >= 0) hits++;
    }
  return hits;
}

double hh_bessi(int n, double x) { return cp_bessi(n,x); }

void *hh_params_new(int K, int read_len, int hcov, int dcov)
{ cp_dev_params *P = (cp_dev_params *)malloc(sizeof(cp_dev_params));
  if (cp_host_fill_params(P,K,read_len,hcov,dcov) != CP_OK) { free(P); return NULL; }
  return P;
}
// -M<model_path>: the product's own loader + fit (cp_host_load_himodel), then the same table builder
void *hh_params_new_model(int K, int read_len, int hcov, int dcov, const char *model_path)
{ double pe[3][21];
  char msg[256];
  if (cp_host_load_himodel(model_path,pe,msg,sizeof(msg)) != CP_OK) return NULL;
  cp_dev_params *P = (cp_dev_params *)malloc(sizeof(cp_dev_params));
  if (cp_host_fill_params(P,K,read_len,hcov,dcov,pe) != CP_OK) { free(P); return NULL; }
  return P;
}
void hh_params_free(void *P) { free(P); }
const unsigned char *hh_params_cthres(void *P) { return &((cp_dev_params *)P)->cthres[0][0][0][0][0]; }
const double *hh_params_logfact(void *P) { return ((cp_dev_params *)P)->logfact; }
int hh_hist_covs(const int64_t *h, int low, int high, int64_t il, int64_t ih, int c, int *hc, int *dc)
{ return cp_host_hist_covs(h,low,high,il,ih,c,hc,dc); }
int hh_decode_profile(const uint8_t *code, int64_t len, uint16_t *out, int cap)
{ return cp_host_decode_profile(code,len,out,cap); }

static void sort_e(cp_eintvl *v, int n)
{ std::stable_sort(v,v+n,[](const cp_eintvl &a, const cp_eintvl &b)
    { if (a.b != b.b) return a.b < b.b; return a.e < b.e; });
}

static thread_local std::vector<cp_intvl> g_pre_unrel;

// Sequential orchestration of the scalar device functions for one read (mirrors what the kernels do
// with lanes).  Returns N or -1 on list overflow.
int hh_classify_read(void *Pv, const char *seq, int rlen, const uint16_t *prof, char *labels,
                     cp_intvl *intvl, int cap, int *M_out, cp_intvl *rintvl, int8_t *fw, int8_t *bw)
{ const cp_dev_params *P = (const cp_dev_params *)Pv;
  const int K = P->K, plen = rlen-(K-1);
  for (int i = 0; i < K-1 && i < rlen; i++) labels[i] = 'N';
  if (plen <= 0) return 0;
  std::vector<uint8_t> wall(plen+1,0);
  std::vector<double> perror((size_t)(plen+1)*4,-INFINITY);
  int ecap = 16*plen+64;
  std::vector<cp_eintvl> ev(ecap), ov(ecap);
  cp_read_t<cp_perr_dense> R;
  R.P = P; R.prof = prof; R.seq = seq; R.lf = P->logfact; R.plen = plen; R.rlen = rlen;
  R.wall = wall.data(); R.wall_s = wall.data(); R.perror.a = perror.data();
  R.eintvl = ev.data(); R.ointvl = ov.data();
  R.ecap = ecap; R.eidx = R.oidx = 0; R.overflow = 0;

  for (int i = 1; i < plen; i++)
    { int a = prof[i-1], b = prof[i];
      if ((a < b ? a : b) >= P->cov[CP_REPEAT]) continue;
      if (abs(a-b) < CP_MIN_CNT_CHANGE) continue;
      cp_wall_candidate(&R,i);
    }
  int NS = R.eidx, NO = R.oidx;
  for (int i = 0; i < NO; i++) { wall[ov[i].b] &= ~CP_W_WALL_O; wall[ov[i].e] &= ~CP_W_WALL_O; }
  for (int i = 0; i < NS; i++) for (int j = ev[i].b+1; j < ev[i].e; j++) wall[j] &= ~CP_W_WALL_O;
  sort_e(ev.data(),NS); NS = cp_dedupe_sorted(ev.data(),NS);
  int midx = NS;
  for (int i = 1; i < plen; i++)
    { if (!((wall[i] & CP_W_WALL_O) && !(wall[i] & CP_W_WALL_S))) continue;
      if (wall[i] & CP_W_PAIRED_M) continue;
      cp_wall_mult(&R,i,NS,&midx);
    }
  for (int i = NS; i < midx; i++) for (int j = ev[i].b+1; j < ev[i].e; j++) wall[j] &= ~CP_W_WALL_O;
  if (NS < midx) { NS = midx; sort_e(ev.data(),NS); }
  NS = cp_merge_eintvl(&R,NS);
  sort_e(ev.data(),NS);
  if (R.overflow) return -1;
  for (int i = 0; i < NS; i++) for (int j = ev[i].b; j < ev[i].e; j++) wall[j] |= CP_W_ERROR;
  int N = 0, b = 0;
  for (int i = 1; i <= plen; i++)
    if (i == plen || ((wall[i-1] & CP_W_ERROR) != 0) != ((wall[i] & CP_W_ERROR) != 0)
        || (!(wall[i] & CP_W_ERROR) && (wall[i] & CP_W_WALL_O)))
      { if (N >= cap) return -1;
        cp_make_interval(&R,NS,b,i,&intvl[N]);
        N++; b = i;
      }
  int M = 0;
  std::vector<int> relmap;
  for (int idx = 0; idx < N; idx++)
    if (cp_rel_interval(P,prof,seq,seq,rlen,&intvl[idx],idx))
      { intvl[idx].is_rel = 1; rintvl[M++] = intvl[idx]; relmap.push_back(idx); }
  *M_out = M;
  if (M > 0)
    { std::vector<int8_t> parent((size_t)M*4), asg(M);
      std::vector<int> eff(M);
      std::vector<uint8_t> rpos(M);
      double hf = cp_rel_dir_full(P,rintvl,M,plen,1,parent.data(),eff.data(),rpos.data(),fw);
      double hb = cp_rel_dir_full(P,rintvl,M,plen,0,parent.data(),eff.data(),rpos.data(),bw);
      cp_rel_reconcile(fw,bw,M,hf,hb,asg.data());
      for (int r = 0; r < M; r++) { rintvl[r].asgn = asg[r]; intvl[relmap[r]].asgn = asg[r]; }
    }
  g_pre_unrel.assign(intvl,intvl+N);                      // (for hh_unrel_memo: the records classify_unrel starts from)
  { std::vector<int> ord(N);
    std::vector<uint8_t> fixed(N);
    for (int i = 0; i < N; i++)
      { ord[i] = i;
        fixed[i] = intvl[i].is_rel && (intvl[i].asgn == CP_HAPLO || intvl[i].asgn == CP_DIPLO);
      }
    std::stable_sort(ord.begin(),ord.end(),[&](int x, int y)
      { int kx = intvl[x].cb < intvl[x].ce ? intvl[x].cb : intvl[x].ce;
        int ky = intvl[y].cb < intvl[y].ce ? intvl[y].cb : intvl[y].ce;
        return kx < ky; });
    for (int i = N-1; i >= 0; i--) if (!fixed[ord[i]]) cp_update_state(P,ord[i],intvl,N);
    for (int i = 0; i < N; i++)    if (!fixed[ord[i]]) cp_update_state(P,ord[i],intvl,N);
  }
  static const char stoc[5] = { 'E','R','H','D','?' };
  for (int i = 0; i < N; i++)
    for (int j = intvl[i].b; j < intvl[i].e; j++)
      labels[K-1+j] = stoc[(int)intvl[i].asgn];
  return N;
}

// The records the last hh_classify_read on this thread handed to classify_unrel (classes of classify_rel in place).
int hh_last_pre_unrel(cp_intvl *out, int cap)
{ const int N = (int)g_pre_unrel.size();
  if (N > cap) return -1;
  if (N) memcpy(out,g_pre_unrel.data(),(size_t)N*sizeof(cp_intvl));
  return N;
}

// The second sweep of classify_unrel with the RE-EVALUATION RULE of k_classify_unrel_grp (kernels.hip), sequentially:
// update_state(idx) (class_unrel.c:192-236) is a pure function of the constant fields of interval idx, of
// "asgn == H" / "asgn == D" of its two neighbours (class_unrel.c:126,145) and of the nearest reliable-H / reliable-D
// interval on either side (find_nn_u, class_unrel.c:11-25) -- never of its own class.  So an interval whose inputs did
// not change since its evaluation in the first sweep keeps its class in the second, and the evaluation can be skipped:
//   need[k] = 1 at the start; an evaluation of k clears need[k]; a class change of j from `old` to `new`
//     * sets need[j-1], need[j+1] if old or new is H or D (the neighbours' "asgn == s" tests),
//     * if j is reliable, for s in {H,D} with old == s or new == s (j leaves / joins the set of find_nn_u): sets need[k]
//       for every k from the nearest member of the set below j to the nearest member above j, both included (the
//       intervals whose nearest member on one side is, or was, j).
// Runs the plain two sweeps and the two sweeps with the rule on copies; intvl gets the classes of the form with the
// rule; returns the number of intervals whose class differs between the two (0 if the rule is exact);
// stats[0] = evaluations of the second sweep, stats[1] = those the rule skips, stats[2] = class changes in sweep 1,
// stats[3] = class changes in sweep 2.
int hh_unrel_memo(void *Pv, cp_intvl *intvl, int N, int64_t *stats)
{ const cp_dev_params *P = (const cp_dev_params *)Pv;
  std::vector<cp_intvl> A(intvl,intvl+N), B(intvl,intvl+N);
  std::vector<int> ord(N);
  std::vector<uint8_t> fixed(N), need(N,1);
  for (int i = 0; i < N; i++)
    { ord[i] = i;
      fixed[i] = intvl[i].is_rel && (intvl[i].asgn == CP_HAPLO || intvl[i].asgn == CP_DIPLO);
    }
  std::stable_sort(ord.begin(),ord.end(),[&](int x, int y)
    { int kx = intvl[x].cb < intvl[x].ce ? intvl[x].cb : intvl[x].ce;
      int ky = intvl[y].cb < intvl[y].ce ? intvl[y].cb : intvl[y].ce;
      return kx < ky; });
  for (int i = N-1; i >= 0; i--) if (!fixed[ord[i]]) cp_update_state(P,ord[i],A.data(),N);
  for (int i = 0; i < N; i++)    if (!fixed[ord[i]]) cp_update_state(P,ord[i],A.data(),N);
  stats[0] = stats[1] = stats[2] = stats[3] = 0;
  auto changed = [&](int j, int old, int nw)
    { if (old == CP_HAPLO || old == CP_DIPLO || nw == CP_HAPLO || nw == CP_DIPLO)
        { if (j > 0) need[j-1] = 1;
          if (j+1 < N) need[j+1] = 1;
        }
      if (B[j].is_rel)
        for (int s = CP_HAPLO; s <= CP_DIPLO; s++)
          if (old == s || nw == s)
            { int lo = j-1, hi = j+1;
              while (lo >= 0 && !(B[lo].is_rel && B[lo].asgn == s)) lo--;
              while (hi < N && !(B[hi].is_rel && B[hi].asgn == s)) hi++;
              for (int k = (lo < 0 ? 0 : lo); k <= (hi >= N ? N-1 : hi); k++) need[k] = 1;
            }
    };
  for (int pass = 0; pass < 2; pass++)
    for (int t = 0; t < N; t++)
      { const int idx = ord[pass == 0 ? N-1-t : t];
        if (fixed[idx]) continue;
        if (pass == 1)
          { stats[0]++;
            if (!need[idx]) { stats[1]++; continue; }
          }
        const int old = B[idx].asgn;
        cp_update_state(P,idx,B.data(),N);
        need[idx] = 0;
        if (B[idx].asgn != old) { stats[2+pass]++; changed(idx,old,B[idx].asgn); }
      }
  int bad = 0;
  for (int i = 0; i < N; i++) { bad += A[i].asgn != B[i].asgn; intvl[i].asgn = B[i].asgn; }
  return bad;
}

int hh_kmer_hash(const char *seq, int j, int K) { return cp_kmer_hash(seq,j,K); }
}
