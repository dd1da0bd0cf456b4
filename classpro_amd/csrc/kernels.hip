// kernels.hip -- HIP kernels of the read-batched classification pipeline (gfx950 / CDNA4, wave64).
//
// Layout in HBM (see include/classpro_amd.h): all reads of a batch are concatenated; read r owns
// prof[prof_off[r] .. prof_off[r+1]) and seq[seq_off[r] .. seq_off[r+1]).  Per-read scratch is
// addressed with the same offsets:
//   bitmap   1 bit per profile position over the WHOLE concatenated profile (bit g = candidate at g)
//   wall     1 byte per position, read r at wall + prof_off[r] + r        (plen+1 cells per read)
//   perror   4 doubles per position, same indexing
//   eintvl / ointvl   E-/O-interval lists, read r at eoff[r], capacity eoff[r+1]-eoff[r]
//   intvl / rintvl / relmap / DP scratch   read r at ioff[r], capacity ioff[r+1]-ioff[r]
//
// Kernels:
//   k_scan_candidates   streaming pass over the profile (the HBM-roofline kernel): wall.c:590-607
//   k_count_caps        per-read candidate count -> scratch capacities
//   k_prefix_caps       exclusive prefix sums of the capacities (single block)
//   k_fill_f64          perror := -inf
//   k_find_wall         one wave per read: wall.c:570-958
//   k_find_rel          one wave per read, one lane per interval: wall.c:960-1051
//   k_classify_rel      one wave per read: class_rel.c:871-963
//   k_classify_unrel    one wave per read: class_unrel.c:248-300
//   k_paint_labels      one wave per read: ClassPro.c:116-119,265-271
//   k_seq_context       dense context arrays (stage API / parity tests only): context.c:8-108
//
// No MFMA anywhere: the path has no dense contraction.  Built with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "cp_wall.h"
#include "cp_class.h"

#define WAVE 64

// Make one lane's global stores visible to the other lanes of the same wave (blocks are one wave).
__device__ __forceinline__ void wave_sync() { __syncthreads(); }

__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE-1); }

// ---------------------------------------------------------------------------------------------
//  k_scan_candidates: bit g of the bitmap <=> g >= 1, min(c[g-1],c[g]) < R and |c[g-1]-c[g]| >= 3
//  (wall.c:592-608).  Each lane loads 8 consecutive counts with one 16-byte load, gets the count
//  before its first one from the neighbouring lane, and stores one byte of flags.  Bits at the
//  first position of a read compare across a read boundary; consumers skip position 0.
//  Algorithmic traffic: 2 B read + 1/8 B written per position.
// ---------------------------------------------------------------------------------------------
#define SCAN_UNROLL 4

__global__ void __launch_bounds__(256)
k_scan_candidates(const uint16_t *__restrict__ prof, int64_t total, int rep, uint8_t *__restrict__ bitmap)
{ const int64_t ngroups = total >> 3;                   // full groups of 8 positions
  const int64_t nthreads = (int64_t)gridDim.x*blockDim.x;
  const int lane = lane_id();
  const uint4 *vp = reinterpret_cast<const uint4 *>(prof);

  for (int64_t base = (int64_t)blockIdx.x*blockDim.x*SCAN_UNROLL; base < ngroups; base += nthreads*SCAN_UNROLL)
    { uint4 v[SCAN_UNROLL];
      int64_t g[SCAN_UNROLL];
#pragma unroll
      for (int u = 0; u < SCAN_UNROLL; u++)
        { g[u] = base+(int64_t)u*blockDim.x+threadIdx.x;
          if (g[u] < ngroups)
            v[u] = vp[g[u]];
          else
            v[u] = make_uint4(0,0,0,0);
        }
#pragma unroll
      for (int u = 0; u < SCAN_UNROLL; u++)
        { const bool live = g[u] < ngroups;
          unsigned last = v[u].w >> 16;                 // count at 8g+7
          unsigned prev = __shfl_up(last,1);            // all lanes take part
          if (lane == 0 && live)
            prev = (g[u] > 0) ? prof[g[u]*8-1] : (v[u].x & 0xffff);
          unsigned c[8] = { v[u].x & 0xffff, v[u].x >> 16, v[u].y & 0xffff, v[u].y >> 16,
                            v[u].z & 0xffff, v[u].z >> 16, v[u].w & 0xffff, v[u].w >> 16 };
          unsigned bits = 0;
#pragma unroll
          for (int k = 0; k < 8; k++)
            { unsigned a = prev, b = c[k];
              unsigned mn = a < b ? a : b, df = a < b ? b-a : a-b;
              bits |= ((mn < (unsigned)rep) && (df >= CP_MIN_CNT_CHANGE)) ? (1u << k) : 0u;
              prev = b;
            }
          if (live)
            bitmap[g[u]] = (uint8_t)bits;
        }
    }

  if (blockIdx.x == 0 && threadIdx.x == 0)              // ragged tail (< 8 positions) + zero pad to a word
    { int64_t p0 = ngroups << 3;
      if (p0 < total)
        { unsigned bits = 0;
          for (int64_t p = p0; p < total; p++)
            if (p > 0)
              { unsigned a = prof[p-1], b = prof[p];
                unsigned mn = a < b ? a : b, df = a < b ? b-a : a-b;
                if (mn < (unsigned)rep && df >= CP_MIN_CNT_CHANGE) bits |= 1u << (p-p0);
              }
          bitmap[ngroups] = (uint8_t)bits;
        }
    }
}

// ---------------------------------------------------------------------------------------------
//  Candidate count per read and scratch capacities.
//    N (intervals)  <= 2*ncand+3   (boundaries are O-walls = candidates, or E-interval endpoints)
//    E-list entries <= 16*ncand+64 (checked at run time; overflow is reported, never silent)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t bitmap_word(const uint64_t *bm, int64_t w, int64_t lo, int64_t hi)
{ // bits of word w restricted to global positions [lo,hi)
  uint64_t x = bm[w];
  int64_t w0 = w << 6;
  if (lo > w0)      x &= ~0ull << (lo-w0);
  if (hi < w0+64)   x &= (hi <= w0) ? 0ull : (~0ull >> (w0+64-hi));
  return x;
}

__global__ void __launch_bounds__(WAVE)
k_count_caps(const uint64_t *__restrict__ bm, const int64_t *__restrict__ prof_off, int nreads,
             int32_t *__restrict__ ncand, int64_t *__restrict__ icap, int64_t *__restrict__ ecap)
{ const int r = blockIdx.x;
  if (r >= nreads) return;
  const int lane = lane_id();
  const int64_t lo = prof_off[r]+1, hi = prof_off[r+1];
  int cnt = 0;
  if (hi > lo)
    for (int64_t w = (lo >> 6)+lane; w <= ((hi-1) >> 6); w += WAVE)
      cnt += __popcll(bitmap_word(bm,w,lo,hi));
  for (int o = 32; o > 0; o >>= 1)
    cnt += __shfl_xor(cnt,o);
  if (lane == 0)
    { ncand[r] = cnt;
      icap[r]  = 2*(int64_t)cnt+4;
      ecap[r]  = 16*(int64_t)cnt+64;
    }
}

// exclusive prefix sums of two int64 arrays, in place, totals appended at [n]; single block.
__global__ void __launch_bounds__(1024)
k_prefix_caps(int64_t *__restrict__ a, int64_t *__restrict__ b, int n)
{ __shared__ int64_t sa[1024], sb[1024];
  const int t = threadIdx.x, T = blockDim.x;
  const int per = (n+T-1)/T;
  const int lo = t*per, hi = (lo+per < n) ? lo+per : n;
  int64_t xa = 0, xb = 0;
  for (int i = lo; i < hi; i++) { xa += a[i]; xb += b[i]; }
  sa[t] = xa; sb[t] = xb;
  __syncthreads();
  if (t == 0)
    { int64_t ra = 0, rb = 0;
      for (int i = 0; i < T; i++)
        { int64_t ya = sa[i], yb = sb[i];
          sa[i] = ra; sb[i] = rb;
          ra += ya; rb += yb;
        }
      a[n] = ra; b[n] = rb;
    }
  __syncthreads();
  xa = sa[t]; xb = sb[t];
  for (int i = lo; i < hi; i++)
    { int64_t ya = a[i], yb = b[i];
      a[i] = xa; b[i] = xb;
      xa += ya; xb += yb;
    }
}

__global__ void __launch_bounds__(256)
k_fill_f64(double *__restrict__ p, int64_t n, double v)
{ const int64_t stride = (int64_t)gridDim.x*blockDim.x;
  int64_t n2 = n >> 1;
  double2 *p2 = reinterpret_cast<double2 *>(p);
  const double2 vv = make_double2(v,v);
  for (int64_t i = (int64_t)blockIdx.x*blockDim.x+threadIdx.x; i < n2; i += stride)
    p2[i] = vv;
  if (blockIdx.x == 0 && threadIdx.x == 0 && (n & 1))
    p[n-1] = v;
}

// ---------------------------------------------------------------------------------------------
//  Stable rank sort of E-intervals by (b,e) across the lanes of a wave (wall.c:519-528,734,875,910).
// ---------------------------------------------------------------------------------------------
__device__ void wave_sort_eintvl(cp_eintvl *v, int n, cp_eintvl *tmp)
{ if (n < 2) return;
  const int lane = lane_id();
  for (int k = lane; k < n; k += WAVE)
    { const cp_eintvl x = v[k];
      int rank = 0;
      for (int m = 0; m < n; m++)
        rank += cp_eintvl_before(v[m],m,x,k) ? 1 : 0;
      tmp[rank] = x;
    }
  wave_sync();
  for (int k = lane; k < n; k += WAVE)
    v[k] = tmp[k];
  wave_sync();
}

// clear / set a flag over the positions [b,e) of the wall array with the lanes of the wave
__device__ __forceinline__ void wave_wall_and(uint8_t *wall, int b, int e, uint8_t mask)
{ for (int j = b+lane_id(); j < e; j += WAVE) wall[j] &= mask; }
__device__ __forceinline__ void wave_wall_or(uint8_t *wall, int b, int e, uint8_t mask)
{ for (int j = b+lane_id(); j < e; j += WAVE) wall[j] |= mask; }

// ---------------------------------------------------------------------------------------------
//  k_find_wall: wall.c:570-958, one wave per read.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(WAVE)
k_find_wall(const cp_dev_params *__restrict__ P, const char *__restrict__ seq, const int64_t *__restrict__ seq_off,
            const uint16_t *__restrict__ prof, const int64_t *__restrict__ prof_off, int nreads,
            const uint64_t *__restrict__ bm, uint8_t *__restrict__ wall_all, double *__restrict__ perror_all,
            cp_eintvl *__restrict__ eintvl_all, cp_eintvl *__restrict__ ointvl_all, const int64_t *__restrict__ eoff,
            cp_intvl *__restrict__ intvl_all, const int64_t *__restrict__ ioff,
            int32_t *__restrict__ nintvl, int32_t *__restrict__ err)
{ const int r = blockIdx.x;
  if (r >= nreads) return;
  const int lane = lane_id();
  const int64_t po = prof_off[r];
  const int plen = (int)(prof_off[r+1]-po);
  const int rlen = (int)(seq_off[r+1]-seq_off[r]);

  cp_read R;
  R.P = P; R.prof = prof+po; R.seq = seq+seq_off[r]; R.plen = plen; R.rlen = rlen;
  R.wall = wall_all+po+r;
  R.perror = perror_all+(po+r)*4;
  R.eintvl = eintvl_all+eoff[r];
  R.ointvl = ointvl_all+eoff[r];
  R.ecap = (int)(eoff[r+1]-eoff[r]);
  R.eidx = R.oidx = 0; R.overflow = 0;
  cp_intvl *intvl = intvl_all+ioff[r];
  const int icap = (int)(ioff[r+1]-ioff[r]);
  uint8_t *wall = R.wall;

  // ---- candidate walk (wall.c:590-707): order-dependent, one lane -------------------------
  if (lane == 0 && plen > 1)
    { const int64_t lo = po+1, hi = po+plen;
      for (int64_t w = lo >> 6; w <= ((hi-1) >> 6); w++)
        { uint64_t bits = bitmap_word(bm,w,lo,hi);
          while (bits)
            { int k = __ffsll((long long)bits)-1;
              bits &= bits-1;
              cp_wall_candidate(&R,(int)((w << 6)+k-po));
            }
        }
    }
  int NS = __shfl(R.eidx,0), NO = __shfl(R.oidx,0);
  int overflow = __shfl(R.overflow,0);
  wave_sync();

  // ---- un-wall positions explained by O-pairs / inside E-intervals (wall.c:722-731) --------
  if (lane == 0)
    for (int k = 0; k < NO; k++)
      { wall[R.ointvl[k].b] &= ~CP_W_WALL_O;
        wall[R.ointvl[k].e] &= ~CP_W_WALL_O;
      }
  wave_sync();
  for (int k = 0; k < NS; k++)
    wave_wall_and(wall,R.eintvl[k].b+1,R.eintvl[k].e,(uint8_t)~CP_W_WALL_O);
  wave_sync();

  // ---- sort + dedupe E-intervals (wall.c:734); the O list is not used again ------------------
  wave_sort_eintvl(R.eintvl,NS,R.ointvl);
  if (lane == 0)
    NS = cp_dedupe_sorted(R.eintvl,NS);
  NS = __shfl(NS,0);
  wave_sync();

  // ---- multi-error / boundary E-intervals (wall.c:760-861) -----------------------------------
  int midx = NS;
  for (int base = 1; base < plen; base += WAVE)
    { int i = base+lane;
      bool cand = false;
      if (i < plen)
        { uint8_t wv = wall[i];
          cand = (wv & CP_W_WALL_O) && !(wv & CP_W_WALL_S);
        }
      uint64_t mask = __ballot(cand);
      if (mask && lane == 0)
        { while (mask)
            { int k = __ffsll((long long)mask)-1;
              mask &= mask-1;
              int ii = base+k;
              if (wall[ii] & CP_W_PAIRED_M)            // may have been set by an earlier i
                continue;
              cp_wall_mult(&R,ii,NS,&midx);
            }
        }
      wave_sync();
    }
  midx = __shfl(midx,0);
  overflow |= __shfl(R.overflow,0);
  for (int k = NS; k < midx; k++)                      // wall.c:868-872
    wave_wall_and(wall,R.eintvl[k].b+1,R.eintvl[k].e,(uint8_t)~CP_W_WALL_O);
  wave_sync();
  if (NS < midx)                                       // wall.c:873-876
    { NS = midx;
      wave_sort_eintvl(R.eintvl,NS,R.ointvl);
    }
  if (lane == 0)                                       // wall.c:878-909
    NS = cp_merge_eintvl(&R,NS);
  NS = __shfl(NS,0);
  overflow |= __shfl(R.overflow,0);
  wave_sync();
  wave_sort_eintvl(R.eintvl,NS,R.ointvl);              // wall.c:910

  for (int k = 0; k < NS; k++)                         // wall.c:917-919
    wave_wall_or(wall,R.eintvl[k].b,R.eintvl[k].e,CP_W_ERROR);
  wave_sync();

  // ---- emit intervals at error transitions and O-walls (wall.c:922-948) ----------------------
  int N = 0, prev_b = 0;
  for (int base = 1; base <= plen; base += WAVE)
    { int i = base+lane;
      bool bd = false;
      if (i <= plen)
        { if (i == plen) bd = true;
          else
            { uint8_t w0 = wall[i-1], w1 = wall[i];
              bd = (((w0 ^ w1) & CP_W_ERROR) != 0) || (!(w1 & CP_W_ERROR) && (w1 & CP_W_WALL_O));
            }
        }
      uint64_t mask = __ballot(bd);
      if (bd)
        { uint64_t below = mask & ((1ull << lane)-1);
          int rank = __popcll(below);
          int b = below ? base+(63-__clzll((long long)below)) : prev_b;
          if (N+rank < icap)
            cp_make_interval(&R,NS,b,i,&intvl[N+rank]);
        }
      if (mask)
        { N += __popcll(mask);
          prev_b = base+(63-__clzll((long long)mask));
        }
    }
  if (N > icap) overflow |= 2;
  if (lane == 0)
    { nintvl[r] = (N > icap) ? icap : N;
      if (overflow) atomicOr(err,overflow);
    }
}

// ---------------------------------------------------------------------------------------------
//  k_find_rel: wall.c:1016-1051.  Every interval is independent: one lane each, then an ordered
//  compaction of the reliable ones (ballot + popcount) into rintvl / relmap.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(WAVE)
k_find_rel(const cp_dev_params *__restrict__ P, const char *__restrict__ seq, const int64_t *__restrict__ seq_off,
           const uint16_t *__restrict__ prof, const int64_t *__restrict__ prof_off, int nreads,
           cp_intvl *__restrict__ intvl_all, cp_intvl *__restrict__ rintvl_all, int32_t *__restrict__ relmap_all,
           const int64_t *__restrict__ ioff, const int32_t *__restrict__ nintvl, int32_t *__restrict__ nrel)
{ const int r = blockIdx.x;
  if (r >= nreads) return;
  const int lane = lane_id();
  const int N = nintvl[r];
  const int rlen = (int)(seq_off[r+1]-seq_off[r]);
  const uint16_t *pr = prof+prof_off[r];
  const char *sq = seq+seq_off[r];
  cp_intvl *intvl = intvl_all+ioff[r], *rintvl = rintvl_all+ioff[r];
  int32_t *relmap = relmap_all+ioff[r];
  int M = 0;
  for (int base = 0; base < N; base += WAVE)
    { int idx = base+lane;
      bool ok = false;
      cp_intvl I;
      if (idx < N)
        { I = intvl[idx];
          ok = cp_rel_interval(P,pr,sq,rlen,&I,idx);
          I.is_rel = ok ? 1 : 0;
          intvl[idx] = I;
        }
      uint64_t mask = __ballot(ok);
      if (ok)
        { int rank = __popcll(mask & ((1ull << lane)-1));
          rintvl[M+rank] = I;
          relmap[M+rank] = idx;
        }
      M += __popcll(mask);
    }
  if (lane == 0)
    nrel[r] = M;
}

// ---------------------------------------------------------------------------------------------
//  k_classify_rel: class_rel.c:871-963.  Forward pass on lane 0, backward pass on lane 1 (they are
//  independent until the reconciliation), DP cells in registers/private memory, 5 B of HBM scratch
//  per (interval, direction).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(WAVE)
k_classify_rel(const cp_dev_params *__restrict__ P, const int64_t *__restrict__ prof_off, int nreads,
               cp_intvl *__restrict__ intvl_all, cp_intvl *__restrict__ rintvl_all, const int32_t *__restrict__ relmap_all,
               const int64_t *__restrict__ ioff, const int32_t *__restrict__ nrel,
               int8_t *__restrict__ parent_all, int32_t *__restrict__ eff_all, uint8_t *__restrict__ rpos_all,
               int8_t *__restrict__ asgn_all, int64_t totalI)
{ const int r = blockIdx.x;
  if (r >= nreads) return;
  const int lane = lane_id();
  const int M = nrel[r];
  if (M == 0) return;
  const int plen = (int)(prof_off[r+1]-prof_off[r]);
  const int64_t o = ioff[r];
  cp_intvl *rintvl = rintvl_all+o;
  int8_t *fw = asgn_all+o, *bw = asgn_all+totalI+o;
  double hdrr = 1.;
  if (lane < 2)
    { const int F = (lane == 0);
      const int64_t d = F ? 0 : totalI;
      hdrr = cp_rel_dir_full(P,rintvl,M,plen,F,parent_all+(d+o)*4,eff_all+d+o,rpos_all+d+o,F ? fw : bw);
    }
  double hf = __shfl(hdrr,0), hb = __shfl(hdrr,1);
  wave_sync();
  // reconcile (class_rel.c:904-938) and copy to the interval arrays (:949-960)
  bool eq = true;
  for (int base = 0; base < M; base += WAVE)
    { int i = base+lane;
      bool ne = (i < M) && (fw[i] != bw[i]);
      if (__ballot(ne)) eq = false;
    }
  int take_bw = 0;
  if (!eq && lane == 0)
    { bool pre = (fw[0] == 1);
      if (pre)
        { int i = 0;
          while (i < M && fw[i]) i++;
          while (i < M) { if (fw[i]) { pre = false; break; } i++; }
        }
      if (!pre)
        { bool suf = (fw[M-1] == 1);
          if (suf)
            { int i = M-2;
              while (i >= 0 && fw[i]) i--;
              while (i >= 0) { if (fw[i]) { suf = false; break; } i--; }
            }
          if (suf) take_bw = 1;
          else if (!(fabs(hf-1.) <= fabs(hb-1.))) take_bw = 1;
        }
    }
  take_bw = __shfl(take_bw,0);
  cp_intvl *intvl = intvl_all+o;
  const int32_t *relmap = relmap_all+o;
  for (int i = lane; i < M; i += WAVE)
    { int8_t a = take_bw ? bw[i] : fw[i];
      rintvl[i].asgn = a;
      intvl[relmap[i]].asgn = a;
    }
}

// ---------------------------------------------------------------------------------------------
//  k_classify_unrel: class_unrel.c:248-300.  Stable rank sort by min(cb,ce) across lanes; the two
//  sweeps are order-dependent (each update reads its neighbours' current classes) and run on lane 0.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(WAVE)
k_classify_unrel(const cp_dev_params *__restrict__ P, int nreads, cp_intvl *__restrict__ intvl_all,
                 const int64_t *__restrict__ ioff, const int32_t *__restrict__ nintvl,
                 int32_t *__restrict__ ord_all)
{ const int r = blockIdx.x;
  if (r >= nreads) return;
  const int lane = lane_id();
  const int N = nintvl[r];
  cp_intvl *intvl = intvl_all+ioff[r];
  int32_t *ord = ord_all+ioff[r];
  for (int k = lane; k < N; k += WAVE)                   // ord[rank] = index | fixed<<31
    { const cp_intvl I = intvl[k];
      const int key = I.cb < I.ce ? I.cb : I.ce;
      int rank = 0;
      for (int m = 0; m < N; m++)
        { int km = intvl[m].cb < intvl[m].ce ? intvl[m].cb : intvl[m].ce;
          rank += (km < key || (km == key && m < k)) ? 1 : 0;
        }
      int fixed = (I.is_rel && (I.asgn == CP_HAPLO || I.asgn == CP_DIPLO)) ? 1 : 0;
      ord[rank] = k | (fixed << 30);
    }
  wave_sync();
  if (lane == 0)
    { for (int i = N-1; i >= 0; i--)
        if (!(ord[i] >> 30))
          cp_update_state(P,ord[i],intvl,N);
      for (int i = 0; i < N; i++)
        if (!(ord[i] >> 30))
          cp_update_state(P,ord[i] & 0x3fffffff,intvl,N);
    }
}

// ---------------------------------------------------------------------------------------------
//  k_paint_labels: ClassPro.c:116-119 ('N' x (K-1)) and :265-271 (interval class per k-mer).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(WAVE)
k_paint_labels(const cp_dev_params *__restrict__ P, const int64_t *__restrict__ seq_off, int nreads,
               const cp_intvl *__restrict__ intvl_all, const int64_t *__restrict__ ioff,
               const int32_t *__restrict__ nintvl, char *__restrict__ labels)
{ const int r = blockIdx.x;
  if (r >= nreads) return;
  const int lane = lane_id();
  const int K = P->K;
  char *lab = labels+seq_off[r];
  for (int i = lane; i < K-1; i += WAVE)
    lab[i] = 'N';
  char *pasgn = lab+(K-1);
  const cp_intvl *intvl = intvl_all+ioff[r];
  const int N = nintvl[r];
  for (int k = 0; k < N; k++)
    { const int b = intvl[k].b, e = intvl[k].e;
      const int a = intvl[k].asgn;
      const char c = (a == CP_ERROR) ? 'E' : (a == CP_REPEAT) ? 'R' : (a == CP_HAPLO) ? 'H' : (a == CP_DIPLO) ? 'D' : '?';
      for (int j = b+lane; j < e; j += WAVE)
        pasgn[j] = c;
    }
}

// ---------------------------------------------------------------------------------------------
//  k_seq_context: dense lctx/rctx ([base][3] uint8) for the stage API and parity tests.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_seq_context(const char *__restrict__ seq, const int64_t *__restrict__ seq_off, int nreads,
              uint8_t *__restrict__ lctx, uint8_t *__restrict__ rctx)
{ const int r = blockIdx.x;
  if (r >= nreads) return;
  const int64_t so = seq_off[r];
  const int rlen = (int)(seq_off[r+1]-so);
  const char *s = seq+so;
  for (int i = threadIdx.x; i < rlen; i += blockDim.x)
    for (int t = 0; t < 3; t++)
      { lctx[(so+i)*3+t] = (uint8_t)cp_lctx(s,rlen,i,t);
        rctx[(so+i)*3+t] = (uint8_t)cp_rctx(s,rlen,i,t);
      }
}
