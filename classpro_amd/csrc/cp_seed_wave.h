// cp_seed_wave.h -- the `-s` seed path of the reference (src/seed.c:966-1032, find_seeds) on the device, ONE WAVE PER
// READ (device only; included by kernels.hip).
//
// What the reference computes per read, from the label string, the count profile and the bases:
//   * anno_repeat (seed.c:482-592): a k-mer position is "unique" when it lies in a maximal stretch of non-R labels
//     that holds at least 2.5 K H/D positions, otherwise "repetitive"; the repetitive stretches, in read coordinates,
//     are the intervals of the .rep mask track;
//   * three seed selections that share one masked-interval list: count MAXimizers among the H-labelled and among the
//     D-labelled k-mers (_find_seeds, seed.c:190-476, window 1000), count MINimizers among the non-E k-mers of the
//     repetitive stretches (_find_seeds_rep, seed.c:667-951, window 200).  A selection compresses the profile into
//     segments of equal count (compress_profile, seed.c:61-110 / :599-665), gives every segment the number of windows
//     in which it is the extreme one (a monotone deque), visits the segments by decreasing window count and marks the
//     sequence minimizers (canonical ntHash, src/nthash.h) of the segments it takes as seeds;
//   * the result per k-mer: 'E' (no seed) or the class of the seed ('H', 'D', or 'R' for a repeat seed),
//     seed.c:1007-1015 -- what the .class.data track carries under -s (ClassPro.c:293).
//
// How the wave does it (every phase is wave-parallel; nothing runs on one lane except the short label-run bookkeeping of
// anno_repeat):
//   * segments: 256 k-mer positions are loaded coalesced per step; run boundaries and the selection's valid flags are two
//     ballots per 64 positions and the SEGMENT STARTS follow from them in closed form by scalar bit operations (a position
//     starts a segment iff it is the first valid k-mer of its run of equal counts, or it begins a run whose predecessor
//     run held a valid k-mer, or it is position 0); the starts go to an LDS list and get their records 64 at a time;
//   * window counts: the reference's monotone deque restated per segment (the comment of sw_windows) -- two searches over
//     a window of segments, from an LDS ring of segments with a range-maximum table beside their keys, and two prefix
//     operations over the segment order (ballots with carried scalars);
//   * the order of the walk: counters per window count in LDS and a stable scatter (ranks from ballots);
//   * the walk: 64 sorted segments at once tested against the masked-interval list (a binary search per lane); a take
//     updates the list wave-uniformly; the canonical-ntHash minimizers of all taken segments are marked at the end of the
//     selection, a lane per k-mer over tables of pre-rotated seeds in LDS;
//   * the masked-interval list is restated slot for slot: the reference searches and sorts one slot past the live part
//     of that array (seed.c:141,161-166), so its leftovers matter; the defined behaviour is that the array
//     is all zeros when a read starts.
// A selection runs as four calls (sw_segments, sw_windows, sw_sort, sw_walk) whose state travels through LDS: each phase
// has its registers to itself.
// Nothing is bounded by the on-chip sizes: a window that reaches beyond the ring, a list longer than SW_MI, more than
// SW_REP repetitive stretches, a group with more than SW_PEND members to take, a selection of more than 65535 segments
// move to (or are flagged in) the read's HBM scratch and the same code goes on there.
#pragma once
#include "cp_seed.h"

#ifndef SW_RING
#define SW_RING  256                             // segments of the window-count pass kept on chip (10 bytes each + the range maxima; beyond: read from HBM)
#define SW_BACK  96                              // ... of which this many lie behind the tile being worked on
#endif
#ifndef SW_LEV
#define SW_LEV   7                               // levels of the range-maximum table: blocks of 1 .. 64 segments
#endif
#define SW_TOPB  (SW_RING >> (SW_LEV-1))          // blocks of the largest size that cover the ring
#define SW_SORTW 512                             // window counts (keys) put in order by one round of the sort
#ifndef SW_SORTN
#define SW_SORTN (1 << 30)                       // ... or ends at the key that brings it to this many segments (see sw_sort_range: not worth it)
#endif
#ifndef SW_MI
#define SW_MI    128
#endif
#define SW_PEND  64
#ifndef SW_REP
#define SW_REP   64
#endif
#define SW_KMAX  64                              // k-mer lengths served by the rotated-seed table (longer: byte-wise fold)
#ifndef SW_STEP
#define SW_STEP  2                               // chunks of 64 positions per load step
#endif

struct cp_seedw_read
  { const char *seq, *cls; const uint16_t *prof; char *state;
    int plen, K, cap, rep_cap;
    int4 *rec;                                                          // cap segments in position order: (begin, end, window count, -)
    int4 *orec;                                                         // cap segments by window count: (begin, end, window count, take flag)
    int32_t *tmp;                                                       // cap slots
    int32_t *gmi_b, *gmi_e;                                             // cap+3 slots each
    int32_t *rep_pairs;
    int32_t *err;                                                       // bit 4: a read needed more segments than its scratch holds
    int dbg_read;                                                       // (diagnostic builds) this is the read to dump
  };

struct cp_seedw_lds
  { int2     rbp[SW_RING];                       // (begin, predecessor's begin) of the valid segments around the tile being worked on
    int16_t  rkey[SW_RING];                      // ... their keys = level 0 of the range-maximum table: level k, slot j mod SW_RING = the largest key of [j, j+2^k)
    union                                        // the table's higher levels live during the window counts only: they share their
      { int16_t rkeyk[SW_LEV-1][SW_RING];        // block with the buffers of the phases around it (3 KB of the wave's 6.8: a sixth wave per SIMD)
        struct
          { int32_t  cval[SW_STEP*WAVE];         // label classes (anno_repeat), the take buffer of the walk, base classes for the hash
            int32_t  bins[32];
            int32_t  pend_b[SW_PEND], pend_e[SW_PEND];
          };
        int32_t sbins[SW_SORTW];                 // the sort's counters, one per window count of the range being put in order
      };
    int32_t  mi_b[SW_MI], mi_e[SW_MI];           // masked-interval list while it fits
    int32_t  rep[2*SW_REP];                      // repetitive stretches in k-mer coordinates while they fit
    cp_seedw_read R;                             // the read (the selections are calls: what they share travels through here)
    int32_t  lm_big;                             // the masked-interval list has moved to the read's HBM scratch
    int32_t  sel[12];                            // what the phases of a selection hand to each other (SEL_*)
  };
static_assert(offsetof(cp_seedw_lds,rkeyk) == offsetof(cp_seedw_lds,rkey)+SW_RING*sizeof(int16_t),"the levels of the range-maximum table are one array");

__shared__ cp_seedw_lds sw_S;                    // the wave's LDS block (one wave per workgroup)

#ifdef CP_SEED_PROF
#define SW_STAMP(k) do { if (lane == 0) { unsigned long long t_ = wall_clock64(); sw_t[k] += t_-sw_last; sw_last = t_; } } while (0)
#define SW_PROF_ARGS , unsigned long long *sw_t, unsigned long long &sw_last
#define SW_PROF_PASS , sw_t, sw_last
#else
#define SW_STAMP(k) ((void)0)
#define SW_PROF_ARGS
#define SW_PROF_PASS
#endif
#ifdef CP_SEED_STOP_AT                            // diagnostic builds: every selection ends after phase CP_SEED_STOP_AT (instruction counts per phase)
#define SW_STOP(k) do { if ((k) == CP_SEED_STOP_AT) return; } while (0)
#else
#define SW_STOP(k) ((void)0)
#endif

__device__ __forceinline__ int sw_first(int v) { return __builtin_amdgcn_readfirstlane(v); }
// the value of the lane below (lane 0: `first`), as a DPP move (wave_shr:1) instead of a round trip through the LDS crossbar
__device__ __forceinline__ int sw_from_below(int v, int first) { return __builtin_amdgcn_update_dpp(first,v,0x138,0xf,0xf,false); }
__device__ __forceinline__ int sw_of_last(int v) { return __builtin_amdgcn_readlane(v,WAVE-1); }
// (the pointer is rebuilt as a GLOBAL one: out of an integer it would be a flat pointer, whose loads and stores count on
//  both memory counters and wait for each other)
template <class T> __device__ __forceinline__ T *sw_first_ptr(T *p)
{ const uint64_t v = (uint64_t)p;
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
  typedef T __attribute__((address_space(1))) *gptr;
  return (T *)(gptr)(((uint64_t)hi << 32) | lo);
}

// k applications of cp_nt_srol (nthash.h:181-207: rol1, then bits 0 and 33 swapped) = the low 33 bits and the high 31
// bits each rotated left by k within themselves
constexpr uint64_t sw_srol_k(uint64_t v, int k)
{ const uint64_t lo = v & 0x1ffffffffull, hi = v >> 33;
  const int a = k % 33, b = k % 31;
  const uint64_t l2 = a ? (((lo << a) | (lo >> (33-a))) & 0x1ffffffffull) : lo;
  const uint64_t h2 = b ? (((hi << b) | (hi >> (31-b))) & 0x7fffffffull) : hi;
  return l2 | (h2 << 33);
}
// rot[c*SW_KMAX+k] = seed of base class c (A C G T, 4 = none: zero) rotated k times: msTab of nthash.h, made at compile
// time; read-only device memory (its 2.5 KB stay in the caches; in LDS they cost every wave a fifth of its block)
struct sw_rot_tab { uint64_t v[5*SW_KMAX]; };
constexpr sw_rot_tab sw_make_rot()
{ sw_rot_tab t{};
  for (int c = 0; c < 5; c++)
    for (int k = 0; k < SW_KMAX; k++)
      t.v[c*SW_KMAX+k] = sw_srol_k(c == 0 ? 0x3c8bfbb395c60474ull : c == 1 ? 0x3193c18562a02b4cull : c == 2 ? 0x20323ed082572324ull
                                   : c == 3 ? 0x295549f54be24456ull : 0ull,k);
  return t;
}
__device__ const sw_rot_tab sw_ROT = sw_make_rot();

// ---- the masked-interval list (seed.c:120-188), in LDS or (big) in the read's HBM scratch ------------------------
struct sw_list
  { int32_t *gb, *ge; bool big;
    __device__ __forceinline__ int  b(int i) const { return big ? gb[i] : sw_S.mi_b[i]; }
    __device__ __forceinline__ int  e(int i) const { return big ? ge[i] : sw_S.mi_e[i]; }
    __device__ __forceinline__ void set(int i, int vb, int ve) { if (big) { gb[i] = vb; ge[i] = ve; } else { sw_S.mi_b[i] = vb; sw_S.mi_e[i] = ve; } }
    __device__ __forceinline__ void set_b(int i, int vb) { if (big) gb[i] = vb; else sw_S.mi_b[i] = vb; }
    __device__ __forceinline__ void set_e(int i, int ve) { if (big) ge[i] = ve; else sw_S.mi_e[i] = ve; }
  };
// slots 0..M, M included (the reference searches one slot past the live part)
__device__ __forceinline__ int sw_mi_find(const sw_list &Lm, int M, int b, int e)
{ int l = 0, r = M;
  while (l <= r)
    { const int m = (l+r) >> 1;
      const int mb = Lm.b(m), me = Lm.e(m);
      if (cp_seed_ovlp(mb,me,b,e)) return m;
      if (mb < b) l = m+1; else r = m-1;
    }
  return -1;
}
// the list outgrows LDS: all SW_MI slots move to HBM, the slots above them are zero (wave-uniform call)
__device__ __forceinline__ void sw_mi_grow(sw_list &Lm, int cap3, int lane)
{ __syncthreads();
  for (int q = lane; q < cap3; q += WAVE)
    { Lm.gb[q] = q < SW_MI ? sw_S.mi_b[q] : 0; Lm.ge[q] = q < SW_MI ? sw_S.mi_e[q] : 0; }
  Lm.big = true;
  __syncthreads();
}
// wave-uniform: every lane walks the same search (broadcast reads); slots are shifted 64 at a time
__device__ __forceinline__ int sw_mi_add(sw_list &Lm, int M, int b, int e, int lane)
{ const int idx = sw_mi_find(Lm,M,b,e);
  if (idx < 0)
    { M++;
      __syncthreads();
      if (lane == 0) Lm.set(M,b,e);                            // waits one slot past the sorted part
      __syncthreads();
      if (M >= 2)                                              // the reference's sort of slots [0,M) moves one element: slot M-1
        { const int xb = Lm.b(M-1), xe = Lm.e(M-1);
          int lo = 0, hi = M-1;                                // first slot whose begin is > xb (stable)
          while (lo < hi) { const int m = (lo+hi) >> 1; if (Lm.b(m) > xb) hi = m; else lo = m+1; }
          for (int top = M-1; top > lo; top -= WAVE)           // slots (lo,M-1] <- their left neighbours, from the top down
            { const int i = top-lane;
              int tb = 0, te = 0;
              if (i > lo) { tb = Lm.b(i-1); te = Lm.e(i-1); }
              __syncthreads();
              if (i > lo) Lm.set(i,tb,te);
              __syncthreads();
            }
          if (lane == 0) Lm.set(lo,xb,xe);
          __syncthreads();
        }
      return M;
    }
  int l = idx-1;
  while (l >= 0 && cp_seed_ovlp(Lm.b(l),Lm.e(l),b,e)) l--;
  l++;
  int r = idx+1;
  while (r < M && cp_seed_ovlp(Lm.b(r),Lm.e(r),b,e)) r++;
  r--;
  const int lb = Lm.b(l), re = Lm.e(r);
  __syncthreads();
  if (lane == 0) { if (b < lb) Lm.set_b(l,b); Lm.set_e(l,re > e ? re : e); }
  __syncthreads();
  if (l == r) return M;
  const int d = r-l;
  M -= d;
  for (int base = l+1; base < M; base += WAVE)                 // close the gap, from the bottom up
    { const int i = base+lane;
      int tb = 0, te = 0;
      if (i < M) { tb = Lm.b(i+d); te = Lm.e(i+d); }
      __syncthreads();
      if (i < M) Lm.set(i,tb,te);
      __syncthreads();
    }
  return M;
}

// ---- minimum canonical hash over the k-mers [b,e) and a mark on every k-mer that attains it (seed.c:392-397) -------
// 64 k-mers at a time, a lane each: the bases' seed classes go to LDS once, then K steps of two table look-ups
// (the forward strand's seed rotated K-1-t times, the reverse strand's rotated t times: nthash.h:215-235 unrolled)
__device__ __attribute__((noinline)) void sw_mark(const char *seq, const char *cls, char *state, int rlen, int K,
                                                  int b, int e, bool rep, int lane)
{ seq = sw_first_ptr(seq); cls = sw_first_ptr(cls); state = sw_first_ptr(state);       // (wave-uniform; global, not flat: sw_first_ptr)
  rlen = sw_first(rlen); K = sw_first(K); b = sw_first(b); e = sw_first(e); rep = sw_first(rep ? 1 : 0) != 0;
  if (e-b == 1) { if (lane == 0) state[b] = rep ? 'R' : cls[b]; return; }
  const bool table = K <= SW_KMAX;
  int mh = CP_SEED_MOD, h = CP_SEED_MOD;
  for (int pass = 0; pass < 2; pass++)                        // minimum first, marks second
    { for (int base = b; base < e; base += WAVE)
        { const int j = base+lane;
          h = CP_SEED_MOD;
          if (table)
            { __syncthreads();
              for (int q = lane; q < WAVE+K-1; q += WAVE)      // seed classes of the bases base .. base+63+K-1
                { int code = 4 | (4 << 3);
                  if (base+q < rlen)
                    { const unsigned c = (unsigned char)seq[base+q];
                      int f = 4, r = 4;
                      switch (c)                               // seedTab, nthash.h:26-59
                        { case 'A': case 'a': case 4: case 5: f = 0; break;
                          case 'C': case 'c': case 7: f = 1; break;
                          case 'G': case 'g': case 3: f = 2; break;
                          case 'T': case 't': case 'U': case 'u': case 1: f = 3; break;
                        }
                      switch (c & 7u)                          // seedTab[c & cpOff], nthash.h:17,226-235: the complement's seed
                        { case 1: r = 3; break;
                          case 3: r = 2; break;
                          case 4: case 5: r = 0; break;
                          case 7: r = 1; break;
                        }
                      code = f | (r << 3);
                    }
                  sw_S.cval[q] = code;
                }
              __syncthreads();
              if (j < e)
                { uint64_t fh = 0, rh = 0;
                  for (int t = 0; t < K; t++)
                    { const int code = sw_S.cval[lane+t];
                      fh ^= sw_ROT.v[(code & 7)*SW_KMAX+(K-1-t)];
                      rh ^= sw_ROT.v[(code >> 3)*SW_KMAX+t];
                    }
                  h = (int)((rh < fh ? rh : fh) % CP_SEED_MOD);
                }
            }
          else if (j < e) h = cp_kmer_hash(seq,j,K);
          if (pass == 0) mh = h < mh ? h : mh;
          else if (j < e && h == mh) state[j] = rep ? 'R' : cls[j];
        }
      if (pass == 0)
        { for (int o = 32; o > 0; o >>= 1) { const int x = __shfl_xor(mh,o); mh = x < mh ? x : mh; }
          if (e-b <= WAVE)                                     // one chunk: its hashes are still in hand
            { if (b+lane < e && h == mh) state[b+lane] = rep ? 'R' : cls[b+lane];
              break;
            }
        }
    }
}

// ---- the marks of ALL segments a selection took, at its end ---------------------------------------------------------
// A take has two halves: the masked-interval list, which later takes of the same selection depend on, and the marks --
// the segment's k-mers with the smallest canonical hash -- which nothing reads before the next selection starts.  Marked
// one take at a time, a segment of a dozen k-mers costs the wave a cold load of its bases and a 40-step hash loop with
// a fifth of the lanes working.  Here the takes only append (begin, end) to a list; at the end of the selection the
// k-mers of all taken segments are laid side by side, a lane each whatever segment they belong to: 64 segments per
// round, their lengths prefix-summed in LDS, a lane finds its segment by binary search, hashes its k-mer from the bases
// directly (a 256-entry letter table, then the rotated-seed table) and joins its segment's minimum with an LDS atomic;
// a second sweep marks the k-mers that attain it.
struct sw_code_tab { uint8_t v[256]; };
constexpr sw_code_tab sw_make_codes()
{ sw_code_tab t{};
  for (int c = 0; c < 256; c++)
    { int f = 4, r = 4;                                      // seedTab / seedTab[c & cpOff], nthash.h:17,26-59,226-235
      if (c == 'A' || c == 'a' || c == 4 || c == 5) f = 0;
      else if (c == 'C' || c == 'c' || c == 7) f = 1;
      else if (c == 'G' || c == 'g' || c == 3) f = 2;
      else if (c == 'T' || c == 't' || c == 'U' || c == 'u' || c == 1) f = 3;
      const int l = c & 7;
      if (l == 1) r = 3; else if (l == 3) r = 2; else if (l == 4 || l == 5) r = 0; else if (l == 7) r = 1;
      t.v[c] = (uint8_t)(f | (r << 3));
    }
  return t;
}
__device__ const sw_code_tab sw_CODE = sw_make_codes();

// the k-mer's bases come in with four 16-byte loads issued together (a byte load per step made the 40 steps 40 memory
// round trips); `rlen` bounds them: a k-mer within 64 bases of the read's end is read byte by byte.  `rot` / `code`: the
// two tables (sw_ROT, sw_CODE) -- sw_mark_all passes its on-chip copies.
template <int STRIDE, class ROT, class CODE>
__device__ __forceinline__ int sw_hash_at(const char *seq, int j, int K, int rlen, const ROT rot, const CODE code_of)
{ if (K > SW_KMAX) return cp_kmer_hash(seq,j,K);
  uint64_t fh = 0, rh = 0;
  CP_BCHK(4,j,rlen-K+1);                                 // a k-mer of the read
  if (j+SW_KMAX <= rlen)
    { uint32_t w[SW_KMAX/4];
      CP_BCHK(4,j+SW_KMAX-1,rlen);
#pragma unroll
      for (int q = 0; q < SW_KMAX/16; q++)
        { const cp_u8x16 x = *reinterpret_cast<const cp_u8x16 *>(seq+j+16*q);
          w[4*q] = x.v[0]; w[4*q+1] = x.v[1]; w[4*q+2] = x.v[2]; w[4*q+3] = x.v[3];
        }
      // Eight bases per group: their letter classes are looked up together, one group ahead, and the group's sixteen
      // rotated seeds together -- a group costs one table round trip.  (Step by step under `if (t < K)` the compiler
      // made groups of four with two dependent round trips each: twenty LDS latencies per k-mer at K = 40 and almost
      // nothing else in flight -- the marks took 88 us per selection for ~540 k-mers.)  Steps beyond K read the zero row.
      constexpr int GB = 8;
      int cd[GB], cdn[GB];
#pragma unroll
      for (int u = 0; u < GB; u++) { cd[u] = code_of[(w[u >> 2] >> (8*(u & 3))) & 0xff]; cdn[u] = 0; }
#pragma unroll
      for (int t0 = 0; t0 < SW_KMAX; t0 += GB)
        if (t0 < K)
          { if (t0+GB < SW_KMAX && t0+GB < K)
              {
#pragma unroll
                for (int u = 0; u < GB; u++) { const int t = t0+GB+u; cdn[u] = code_of[(w[t >> 2] >> (8*(t & 3))) & 0xff]; }
              }
#pragma unroll
            for (int u = 0; u < GB; u++)
              { const int t = t0+u;
                const bool on = t < K;
                const int code = on ? cd[u] : (4 | (4 << 3));
                fh ^= rot[(code & 7)*STRIDE+(on ? K-1-t : 0)];
                rh ^= rot[(code >> 3)*STRIDE+(on ? t : 0)];
              }
#pragma unroll
            for (int u = 0; u < GB; u++) cd[u] = cdn[u];
          }
    }
  else
    for (int t = 0; t < K; t++)
      { const int code = code_of[(unsigned char)seq[j+t]];
        fh ^= rot[(code & 7)*STRIDE+(K-1-t)];
        rh ^= rot[(code >> 3)*STRIDE+t];
      }
  return (int)((rh < fh ? rh : fh) % CP_SEED_MOD);
}

// (defined below sw_mark_all's comment) one copy of the unrolled hash for all call sites of sw_mark_all
__device__ __attribute__((noinline)) int sw_hash_lds(const char *seq, int j, int K, int rlen);

// The hash loop reads a letter's class and two rotated seeds per base, each lane at its own address: as global loads
// (the tables sit in the caches) these 120 scattered loads per k-mer kept the address unit busy for a third of the
// kernel's time (0.37 G load instructions per launch of 48 000 reads).  For the duration of sw_mark_all the two tables
// live in LDS, in blocks that are idle then: the window-count ring (2560 bytes = the rotated seeds) and the take buffer
// (the 256 letter classes).  A k-mer is hashed ONCE: the hashes of up to four sweeps (256 k-mers, all but a few rounds)
// wait in registers for the mark sweep.
__device__ __attribute__((noinline)) void sw_mark_all(const char *seq, const char *cls, char *state, int K, int rlen, const int32_t *takes,
                                                      int ntake, bool rep, int lane)
{ seq = sw_first_ptr(seq); cls = sw_first_ptr(cls); state = sw_first_ptr(state); takes = sw_first_ptr(takes);
  K = sw_first(K); rlen = sw_first(rlen); ntake = sw_first(ntake); rep = sw_first(rep ? 1 : 0) != 0;
  if (ntake == 0) return;
  int32_t *s_pre = sw_S.pend_b, *s_min = sw_S.pend_e;        // (the pending-take buffers are idle here; SW_PEND >= 64)
  static_assert(SW_PEND >= WAVE,"one slot per segment of a round");
  // (a table row per base class, 65 entries apart: 64 apart the five rows start in the same LDS bank and the lanes' reads
  //  of one step -- same column, one of five rows -- went through it one after the other)
  static_assert(offsetof(cp_seedw_lds,rkey) == sizeof(sw_S.rbp) && offsetof(cp_seedw_lds,cval) == offsetof(cp_seedw_lds,rkey)+sizeof(sw_S.rkey)
                && sizeof(sw_S.rbp)+sizeof(sw_S.rkey)+sizeof(sw_S.cval)/2 >= 5*(SW_KMAX+1)*sizeof(uint64_t),
                "the rotated seeds fit the ring and the first half of the take buffer");
  static_assert(sizeof(sw_S.cval)/2 >= sizeof(sw_code_tab),"the letter classes fit the second half of the take buffer");
  uint64_t *lrot = reinterpret_cast<uint64_t *>(&sw_S.rbp[0]);
  uint8_t *lcode = reinterpret_cast<uint8_t *>(&sw_S.cval[SW_STEP*WAVE/2]);
  __syncthreads();
  for (int q = lane; q < 5*SW_KMAX; q += WAVE) lrot[(q/SW_KMAX)*(SW_KMAX+1)+q%SW_KMAX] = sw_ROT.v[q];
  for (int q = lane; q < 256/4; q += WAVE) reinterpret_cast<uint32_t *>(lcode)[q] = reinterpret_cast<const uint32_t *>(sw_CODE.v)[q];
  __syncthreads();
  constexpr int KEEP = 4;
  int ns = 0;
  for (int base = 0; base < ntake; base += ns)
    { // A round takes as many of the next segments as hold KEEP*WAVE k-mers together (one at least): their hashes then
      // wait in registers for the mark sweep.  (64 segments per round before: with ~16 k-mers per taken segment nearly
      // every round was over that size and hashed every k-mer twice, once for the minimum and once for the marks.)
      const int nmax = ntake-base < WAVE ? ntake-base : WAVE;
      int b = 0, len = 0;
      if (lane < nmax) { b = takes[2*(base+lane)]; len = takes[2*(base+lane)+1]-b; }
      int incl = len;
      for (int o = 1; o < WAVE; o <<= 1) { const int x = __shfl_up(incl,o); if (lane >= o) incl += x; }
      ns = __popcll(__ballot(lane < nmax && incl <= KEEP*WAVE));      // (the sums grow: a prefix of the lanes)
      if (ns == 0) ns = 1;
      if (lane >= ns) { b = 0; len = 0; }
      const int total = __builtin_amdgcn_readlane(incl,ns-1);
      __syncthreads();
      s_pre[lane] = incl-len;                                // first k-mer slot of segment `lane`
      s_min[lane] = CP_SEED_MOD;
      __syncthreads();
      auto locate = [&](int q, int &lo, int &j)              // k-mer slot q -> its segment (of this round) and its position
        { lo = 0;
#pragma unroll
          for (int step = WAVE/2; step > 0; step >>= 1)
            if (lo+step < ns && s_pre[lo+step] <= q) lo += step;
          j = __shfl(b,lo)+(q-s_pre[lo]);
        };
      if (total <= KEEP*WAVE)                                // the usual round: hash once, keep, mark
        { int hk[KEEP], jk[KEEP], lk[KEEP];
#pragma unroll
          for (int u = 0; u < KEEP; u++)
            { const int q = u*WAVE+lane;
              hk[u] = -1; jk[u] = 0; lk[u] = 0;
              if (u*WAVE < total)
                { const bool on = q < total;
                  int lo = 0, j = 0;
                  locate(on ? q : 0,lo,j);
                  if (on)
                    { hk[u] = sw_hash_lds(seq,j,K,rlen); jk[u] = j; lk[u] = lo;
                      atomicMin(&s_min[lo],hk[u]);
                    }
                }
            }
          __syncthreads();
#pragma unroll
          for (int u = 0; u < KEEP; u++)
            if (hk[u] >= 0 && hk[u] == s_min[lk[u]]) state[jk[u]] = rep ? 'R' : cls[jk[u]];
          __syncthreads();
          continue;
        }
      // a round with more k-mers: two sweeps, the segments' minima, then the marks
      for (int pass = 0; pass < 2; pass++)
        { for (int q0 = 0; q0 < total; q0 += WAVE)
            { const int q = q0+lane;
              const bool on = q < total;
              int lo = 0, j = 0;
              locate(on ? q : 0,lo,j);
              if (on)
                { const int h = sw_hash_lds(seq,j,K,rlen);
                  if (pass == 0) atomicMin(&s_min[lo],h);
                  else if (h == s_min[lo]) state[j] = rep ? 'R' : cls[j];
                }
            }
          __syncthreads();
        }
    }
}

// The hash of sw_mark_all as a call: inlined at its five call sites the unrolled 64-step body made sw_mark_all 10 000
// lines of code (the tables are where sw_mark_all put them: rotated seeds over the ring, letter classes in the take buffer).
__device__ __attribute__((noinline)) int sw_hash_lds(const char *seq, int j, int K, int rlen)
{ typedef const char __attribute__((address_space(1))) *gcp;
  const uint64_t *lrot = reinterpret_cast<const uint64_t *>(&sw_S.rbp[0]);
  const uint8_t *lcode = reinterpret_cast<const uint8_t *>(&sw_S.cval[SW_STEP*WAVE/2]);
  return sw_hash_at<SW_KMAX+1>((const char *)(gcp)(uintptr_t)seq,j,K,rlen,lrot,lcode);
}

// ---- one selection (seed.c:190-476 with C = 'H'/'D'; seed.c:667-951 with C = 0) --------------------------------------
// The window counts WITHOUT the deque.  The reference feeds the segments to a monotone deque one by one (seed.c:218-324 /
// :694-810) and a segment's "number of windows" is decided when it leaves the deque.  When and how it leaves follows from
// the segments around it alone (scripts/proto/seed_windows.py checks the restatement below against
// the literal loop on random inputs).  With b(i) the begin of segment i, c(i) its count, W the window, "beats" = larger
// count (H/D) or smaller count (repeats), x(i) the first later segment that begins at or beyond b(i)+W:
//   * segment i is BEATEN if a valid segment g in (i, x(i)] beats it (the first such one counts).  Its value is the
//     reference's stand-in (count, or W_REP - count), unless no segment that beats i and begins beyond b(g-1)-W precedes
//     it -- then i shared the front's count when the deque was wiped and gets min(b(g)-b(i), W);
//   * otherwise i EXPIRES from the front at step x(i) (or at the end of the read) with W, or with
//     min(b(i) - last_oor_pos + 1, W) when "last_oor" holds: the previous expiring segment exists and no WIPE (a valid
//     segment k that beats everything still in the deque: some valid segment begins beyond b(k-1)-W before k, and none
//     of those is as good as k) happened after that segment's expiry step and up to i.  last_oor_pos is the end of the
//     last expiring segment before i that was followed in (q, x(q)] by a valid segment and by none of its own count.
// So every segment needs one forward and one backward search over a window of W positions, and the expiring ones two
// prefix operations ("last such segment before me") over the segment order.  A wave takes 64 segments at a time, a lane
// each: short searches run per lane (most segments have a better neighbour a few segments away), the few long ones are
// done one by one by the whole wave, 64 window segments per step (ballots); the prefix operations are ballots with three
// carried scalars.  The segments' (begin, count) sit in an LDS ring around the tile; a window that reaches beyond it is
// read from the records in HBM.
// A selection runs as four CALLS (not inlined) -- segments, window counts, sort, walk -- so that each phase has its
// registers allotted on its own: inlined three times into the kernel, the wave-uniform state of everything (a dozen
// pointers, the list, the counters) was live across all of it and 196 scalar and 45 vector registers were spilled,
// reloaded from scratch inside the per-position loop.  What a phase needs of the read it takes from the LDS block
// (sw_S.R; made wave-uniform by readfirstlane, since a callee's arguments arrive in vector registers); what the phases
// hand to each other goes through sw_S.sel[] and sw_S.lm_big.
__device__ __forceinline__ cp_seedw_read sw_the_read()
{ cp_seedw_read R;
  R.seq = sw_first_ptr(sw_S.R.seq); R.cls = sw_first_ptr(sw_S.R.cls); R.prof = sw_first_ptr(sw_S.R.prof); R.state = sw_first_ptr(sw_S.R.state);
  R.plen = sw_first(sw_S.R.plen); R.K = sw_first(sw_S.R.K); R.cap = sw_first(sw_S.R.cap); R.rep_cap = sw_first(sw_S.R.rep_cap);
  R.rec = sw_first_ptr(sw_S.R.rec); R.orec = sw_first_ptr(sw_S.R.orec); R.tmp = sw_first_ptr(sw_S.R.tmp);
  R.gmi_b = sw_first_ptr(sw_S.R.gmi_b); R.gmi_e = sw_first_ptr(sw_S.R.gmi_e); R.rep_pairs = sw_first_ptr(sw_S.R.rep_pairs);
  R.err = sw_first_ptr(sw_S.R.err); R.dbg_read = sw_first(sw_S.R.dbg_read);
  return R;
}
enum { SEL_N = 0, SEL_M, SEL_NBIG, SEL_BLAST, SEL_STOP,      // sw_S.sel[]: valid segments, skipped stretches (list slots), counts > 1000, begin of the last segment, selection over
       SEL_NSORTED, SEL_NEXTKEY, SEL_POS, SEL_NTAKE, SEL_MORE }; // segments in order so far, first key not yet in order; the walk's place, its takes, "put more in order"
__device__ __forceinline__ sw_list sw_the_list(const cp_seedw_read &R)
{ sw_list Lm; Lm.gb = R.gmi_b; Lm.ge = R.gmi_e; Lm.big = sw_first(sw_S.lm_big) != 0; return Lm; }
__device__ __forceinline__ int sw_sel(int k) { return sw_first(sw_S.sel[k]); }
// (wave-uniform values; the barriers make them visible to the next phase)
__device__ __forceinline__ void sw_sel_put(int lane, int n, int M, int nbig, int b_last, int stop, bool big)
{ __syncthreads();
  if (lane == 0)
    { sw_S.sel[SEL_N] = n; sw_S.sel[SEL_M] = M; sw_S.sel[SEL_NBIG] = nbig; sw_S.sel[SEL_BLAST] = b_last; sw_S.sel[SEL_STOP] = stop;
      sw_S.lm_big = big ? 1 : 0;
    }
  __syncthreads();
}

template <bool rep>
__device__ __attribute__((noinline)) void sw_segments(int C_, int nrep_, int rep_big_, int lane SW_PROF_ARGS)
{ const cp_seedw_read R = sw_the_read();
  sw_list Lm = sw_the_list(R);
  const int C = sw_first(C_), nrep = sw_first(nrep_);
  const bool rep_big = sw_first(rep_big_) != 0;
  const int plen = R.plen, Km1 = R.K-1;
  const uint64_t lt = (1ull << lane)-1;
  int n = 0, M = 0, b_last = 0;
  // ---- segments (seed.c:61-110 / :599-665 in closed form) ----
  // valid(i): the k-mer takes part in this selection.  A valid segment is a run of equal counts from its first valid
  // k-mer on (it may run on over k-mers of other classes); the k-mers skipped between segments form invalid ones.  A
  // segment that would start at the last k-mer is never made (the reference's loop ends first).
  // Per 64 k-mers the lanes only produce two masks -- run boundaries, valid k-mers -- and the segment starts come out of
  // them with scalar bit operations, the same for every lane: a segmented OR-scan ("this run already holds a valid
  // k-mer") by one carry-propagating addition.  The VALID segments get records of their own, (begin, end, -, key+1) at
  // R.rec[0..n), the skipped stretches go to a side list (R.orec) and from there to the masked-interval list: the window
  // counts, the sort and the walk then only see valid segments.  Two skipped stretches are never adjacent, so a valid
  // segment's predecessor of either kind is known from its valid predecessor's begin and end.
  { bool carry = false;                                      // the run entering the chunk already holds a valid k-mer
    int nv = 0, ni = 0;                                      // valid / skipped segments so far
    int ri = 0;                                              // (repeats) first repetitive stretch that ends beyond the chunk start
    int prevc = -1;                                          // count of the k-mer before the chunk
    int nbuf = 0;                                            // segment starts waiting in sw_S.cval (< 2*WAVE)
    int prev_e = -1, prev_idx = 0;                           // the last start written so far ((position << 1) | valid, its index), -1: none
    auto emit = [&](int m)                                   // records of the first m <= WAVE waiting starts (wave-uniform)
      { __syncthreads();
        const bool a = lane < m;
        const int e = a ? sw_S.cval[lane] : 0;
        const int p = e >> 1;
        const bool mv = (e & 1) != 0;
        const uint64_t vb = __ballot(a && mv), ib = __ballot(a && !mv);
        const int idx = mv ? nv+__popcll(vb & lt) : ni+__popcll(ib & lt);
        const int ep = sw_from_below(e,prev_e), pidx = sw_from_below(idx,prev_idx);    // the segment before mine ends where mine begins
        int cnt = 0;
        if (a && mv) { CP_BCHK(3,p,plen); cnt = (int)R.prof[p]; }
        if (a)
          { if (ep >= 0 && pidx < R.cap) { if (ep & 1) R.rec[pidx].y = p; else R.orec[pidx].y = p; }
            if (idx < R.cap)                                 // (z: the begin of my predecessor of either kind, until the window count replaces it)
              { if (mv) { R.rec[idx].x = p; R.rec[idx].z = ep >= 0 ? (ep >> 1) : -(1 << 30); R.rec[idx].w = (rep ? 32767-cnt : cnt)+1; }
                else R.orec[idx].x = p;
              }
          }
        prev_e = __builtin_amdgcn_readlane(e,m-1); prev_idx = __builtin_amdgcn_readlane(idx,m-1);
        nv += __popcll(vb); ni += __popcll(ib);
        const int rest = nbuf-m;                             // (< WAVE) move the waiting rest to the front
        int t = 0;
        if (lane < rest) t = sw_S.cval[m+lane];
        __syncthreads();
        if (lane < rest) sw_S.cval[lane] = t;
        nbuf = rest;
        __syncthreads();
      };
    int ncnt[SW_STEP]; char ncl[SW_STEP], nst[SW_STEP];      // a step's loads are issued one step ahead
    auto load_step = [&](int e0)
      {
#pragma unroll
        for (int u = 0; u < SW_STEP; u++)
          { const int p = e0+u*WAVE+lane;
            const bool in = p < plen;
            if (in) CP_BCHK(3,p,plen);
            ncnt[u] = in ? (int)R.prof[p] : 0;
            ncl[u] = in ? R.cls[p] : (char)0;
            nst[u] = (in && rep) ? R.state[p] : (char)'E';
          }
      };
    load_step(0);
    for (int e0 = 0; e0 < plen; e0 += SW_STEP*WAVE)
      { int cnt[SW_STEP]; char cl[SW_STEP], st[SW_STEP];
#pragma unroll
        for (int u = 0; u < SW_STEP; u++) { cnt[u] = ncnt[u]; cl[u] = ncl[u]; st[u] = nst[u]; }
        if (e0+SW_STEP*WAVE < plen) load_step(e0+SW_STEP*WAVE);
#pragma unroll
        for (int u = 0; u < SW_STEP; u++)
          { const int c0 = e0+u*WAVE, p = c0+lane;
            if (c0 >= plen) break;
            const bool in = p < plen;
            const int prevc0 = prevc;
            prevc = sw_of_last(cnt[u]);
            uint64_t rmask = ~0ull;
            if (rep)                                         // the chunk's k-mers inside repetitive stretches (scalar: the stretches are sorted)
              { rmask = 0;
                while (ri < nrep)
                  { const int rb = rep_big ? R.rep_pairs[2*ri]-Km1 : sw_S.rep[2*ri], re = rep_big ? R.rep_pairs[2*ri+1]-Km1 : sw_S.rep[2*ri+1];
                    if (rb >= c0+WAVE) break;
                    const int lo = rb > c0 ? rb-c0 : 0, hi = re < c0+WAVE ? re-c0 : WAVE;      // [lo,hi) of the chunk
                    if (hi > lo) rmask |= (hi == WAVE ? ~0ull : ((1ull << hi)-1)) & ~((1ull << lo)-1);
                    if (re <= c0+WAVE) ri++; else break;
                  }
              }
            const bool v = in && (rep ? (cl[u] != 'E' && st[u] == 'E' && ((rmask >> lane) & 1)) : cl[u] == (char)C);
            const uint64_t vm = __ballot(v);
            // 64 k-mers without a valid one, entered by a run that holds none: no segment starts here and nothing carries
            // over (hv = 0 below) -- most chunks of the H and of the repeat selection
            if (vm == 0 && !carry && c0 != 0) continue;
            const int cpv = sw_from_below(cnt[u],prevc0);
            const bool bnd = in && cnt[u] != cpv;
            const uint64_t inm = plen-c0 >= WAVE ? ~0ull : ((1ull << (plen-c0))-1);
            const uint64_t bm = __ballot(bnd);
            // hv: the run of a k-mer holds a valid k-mer at or before it -- a segmented OR-scan of vm over the runs that
            // start at bm, by ONE addition: X marks the k-mers whose successor belongs to the same run; adding a valid
            // k-mer's bit to X sends a carry up through the rest of its run, which flips those bits of X and sets the run's
            // last one (a second valid k-mer on the way is not flipped, but it is in vm).  (Six shift-and-mask steps
            // before: 33 of the 81 scalar instructions per 64 k-mers of a loop whose bound is the scalar unit.)
            const uint64_t v0 = vm | ((carry && !(bm & 1)) ? 1ull : 0ull);       // (the run entering the chunk goes on at k-mer 0)
            const uint64_t X = ~(bm >> 1) & 0x7fffffffffffffffull;
            const uint64_t hv = v0 | ((X+(v0 & X)) ^ X);
            const uint64_t hvs = (hv << 1) | (carry ? 1ull : 0ull);       // the same one k-mer earlier
            // a k-mer starts a segment iff it is the first valid one of its run, or it begins a run whose predecessor run
            // held a valid k-mer, or it is the read's first
            const uint64_t sm = inm & ((vm & ~(hvs & ~bm)) | (bm & hvs) | (c0 == 0 ? 1ull : 0ull));
            carry = (hv >> 63) != 0;
            // The starts only go to a list in LDS here ((position, valid?) in position order); their records are written
            // 64 starts at a time by sw_emit below, every lane with a start of its own -- written on the spot, the record
            // code ran for every chunk with a handful of its lanes, and the last start's place had to be kept by scalar
            // bookkeeping in a loop that the scalar unit bounds.
            if (sm)
              { if ((sm >> lane) & 1) sw_S.cval[nbuf+__popcll(sm & lt)] = (p << 1) | (int)((vm >> lane) & 1);
                nbuf += __popcll(sm);
                if (nbuf >= WAVE) emit(WAVE);
              }
          }
      }
    if (nbuf > 0) emit(nbuf);
    const int lpos = prev_e >= 0 ? (prev_e >> 1) : -1, ltype = prev_e & 1, lidx = prev_idx;
    if (lpos >= 0 && lpos >= plen-1)                         // a segment at the last k-mer is never made: it only ends its predecessor
      { if (ltype) nv--; else ni--;
      }
    else if (lpos >= 0 && lane == 0 && lidx < R.cap)         // the last segment runs to the end of the read
      { if (ltype) R.rec[lidx].y = plen; else R.orec[lidx].y = plen; }
    n = nv; M = ni;
    b_last = lpos >= plen-1 ? -1 : lpos;                     // begin of the last segment of either kind (-1: see below)
    __syncthreads();                                         // the records are visible to the wave
    if (nv > R.cap || ni > R.cap) { if (lane == 0) atomicOr(R.err,4); sw_sel_put(lane,0,0,0,0,1,Lm.big); return; }   // (k_seed_caps sizes the scratch; reported, never written past)
    if (lpos >= plen-1)                                      // the dropped start's predecessor is the last segment
      { int bl = -1;
        if (nv > 0) bl = R.rec[nv-1].x;
        if (ni > 0) { const int x = R.orec[ni-1].x; bl = x > bl ? x : bl; }
        b_last = bl;
      }
    // the skipped stretches are masked from the start: slots 0..M-1 of the list (on chip while M+3 < SW_MI)
    if (!Lm.big && M+3 >= SW_MI) sw_mi_grow(Lm,R.cap+3,lane);
    for (int q = lane; q < M; q += WAVE) { const int4 t = R.orec[q]; Lm.set(q,t.x,t.y); }
    __syncthreads();
  }
  SW_STAMP(1);
  sw_sel_put(lane,n,M,0,b_last,0,Lm.big);
}

template <bool rep>
__device__ __attribute__((noinline)) void sw_windows(int lane SW_PROF_ARGS)
{ const cp_seedw_read R = sw_the_read();
  const int W = rep ? CP_SEED_W_REP : CP_SEED_W;
  const uint64_t lt = (1ull << lane)-1;
  const int n = sw_sel(SEL_N), b_last = sw_sel(SEL_BLAST);
  int nbig = 0;
  // ---- window counts, 64 valid segments at a time ----
  // key = count (H/D) or 32767 - count (repeats): both selections look for the larger key.  pb(j): the begin of j's
  // predecessor of either kind (record field z until the window count replaces it); j is still within reach of i
  // (j <= x(i)) iff pb(j) < b(i)+W.
  { constexpr int NONE = -(1 << 30);
    struct sg { int b, pb, key; };
    int ring_hi = 0;                                         // segments [ring_hi-SW_RING, ring_hi) are in the ring
    auto seg = [&](int j) -> sg                              // (a segment whose window count is written already: pb is gone, nobody asks)
      { sg r;
        if (j < ring_hi && j >= ring_hi-SW_RING) { const int2 t = sw_S.rbp[j & (SW_RING-1)]; r.b = t.x; r.pb = t.y; r.key = sw_S.rkey[j & (SW_RING-1)]; }
        else { const int4 t = R.rec[j]; r.b = t.x; r.pb = t.z; r.key = t.w-1; }
        return r;
      };
    // the tile and its successor: always in the ring (it covers [t0-SW_BACK, t0+SW_RING-SW_BACK) of [0,n)), no test
    static_assert(WAVE+1 <= SW_RING-SW_BACK && SW_TOPB >= 2,"a tile and its successor are in the ring");
    auto segr = [&](int j) -> sg
      { sg r; const int2 t = sw_S.rbp[j & (SW_RING-1)]; r.b = t.x; r.pb = t.y; r.key = sw_S.rkey[j & (SW_RING-1)]; return r; };
    bool c_have = false, c_wipe = false; int c_expb = 0, c_wpb = 0, c_pos = 0;   // carried: begin of the last expiring segment, pb of the last wipe, last_oor_pos
    for (int t0 = 0; t0 < n; t0 += WAVE)
      { { int upto = t0+SW_RING-SW_BACK;                     // the ring covers [t0-SW_BACK, t0+SW_RING-SW_BACK)
          if (upto > n) upto = n;
          if (ring_hi < upto)
            { __syncthreads();                               // (the slots being replaced are no longer read)
              const int old_hi = ring_hi;
              for (int j = old_hi+lane; j < upto; j += WAVE)
                { const int4 t = R.rec[j];
                  sw_S.rbp[j & (SW_RING-1)] = make_int2(t.x,t.z); sw_S.rkey[j & (SW_RING-1)] = (int16_t)(t.w-1);
                }
              // level k: the blocks [j, j+2^k) that have just become complete (j+2^k in (old_hi, upto]), from the two halves
              // one level down
#pragma unroll
              for (int k = 1; k < SW_LEV; k++)
                { __syncthreads();
                  const int h = 1 << (k-1);
                  int j0 = old_hi-2*h+1;
                  if (j0 < 0) j0 = 0;
                  for (int j = j0+lane; j+2*h <= upto; j += WAVE)
                    { const int16_t a = sw_S.rkey[(k-1)*SW_RING+(j & (SW_RING-1))], c = sw_S.rkey[(k-1)*SW_RING+((j+h) & (SW_RING-1))];
                      sw_S.rkey[k*SW_RING+(j & (SW_RING-1))] = a > c ? a : c;
                    }
                }
              ring_hi = upto;
              __syncthreads();
            }
        }
        const int i = t0+lane;
        const bool act = i < n;
        sg me; me.b = 0; me.pb = NONE; me.key = -1;
        if (act) me = segr(i);
        const int bi = me.b, ki = me.key, pbi = me.pb;
        int ei = 0;                                          // my end = the begin of my successor of either kind
        if (act)
          { if (i+1 < n) { const sg nx = segr(i+1); ei = nx.pb == bi ? nx.b : nx.pb; }
            else ei = R.rec[i].y;
          }
        // -- forward: the first segment within reach that beats me; else what the reach holds --
        // The reach (begins grow with the index) is found by bisection.  What it holds comes from the range-maximum table
        // beside the ring keys (rkey[k][j] = the largest key of [j, j+2^k)): the largest key of the reach from four
        // overlapping blocks -- larger than mine: I am beaten; equal: a segment of my own count follows me -- and the FIRST
        // segment that beats me by a descent over the block sizes (a block that holds nothing better is stepped over
        // whole): nine steps of one LDS read, the same for every lane.  (Before: per-lane walks over aligned blocks of
        // 16 / 4 / 1 segments in a loop that ran until the wave's slowest lane was done, ~17 rounds of 25 instructions
        // for a segment that is beaten by nothing in a reach of ~70 segments.)
        const int hiR = ring_hi-1, loR = ring_hi-SW_RING > 0 ? ring_hi-SW_RING : 0;      // the ring holds segments [loR, hiR]
        auto lev = [](int k, int j) -> int { return (int)sw_S.rkey[k*SW_RING+(j & (SW_RING-1))]; };
        auto range_max = [&](int s, int e) -> int            // largest key of [s,e], 1 <= e-s+1 <= 255, both inside the ring
          { const int len = e-s+1;
            int k = 31-__clz(len);
            k = k > SW_LEV-1 ? SW_LEV-1 : k;
            const int last = e-(1 << k)+1, st = 1 << k;
            const int16_t *row = sw_S.rkey+k*SW_RING;
            int m = row[last & (SW_RING-1)];
#pragma unroll
            for (int q = 0; q < SW_TOPB-1; q++)              // (below the top level two blocks cover the range: the others repeat the last)
              m = max(m,(int)row[min(s+q*st,last) & (SW_RING-1)]);
            return m;
          };
        int g = -1, bg = 0, pbg = 0;
        bool eq = false, nonempty = false, fopen = false;
#ifndef SW_SKIP_FWD
        { const int limv = bi+W;
          int x = i;                                         // x(i) within the ring: the last segment with pb < b(i)+W
#pragma unroll
          for (int st = SW_RING/2; st > 0; st >>= 1)
            { const int c = x+st;
              const int pbc = sw_S.rbp[c & (SW_RING-1)].y;
              x = (act && c <= hiR && pbc < limv) ? c : x;
            }
          nonempty = x > i;
          const int mx = nonempty ? range_max(i+1,x) : -1;
          eq = nonempty && mx == ki;
          if (mx > ki)                                       // (never for a lane that is not active: its key is -1 and its reach empty)
            { int pos = i+1;
#pragma unroll
              for (int q = 0; q < SW_LEV+SW_TOPB-2; q++)     // the largest blocks SW_TOPB-1 times (a reach holds < 256 segments), then the smaller ones
                { const int k = q < SW_TOPB-1 ? SW_LEV-1 : SW_LEV+SW_TOPB-3-q;
                  const int m = lev(k,pos);
                  pos = (pos+(1 << k)-1 <= x && m <= ki) ? pos+(1 << k) : pos;
                }
              g = pos;
              const int2 tg = sw_S.rbp[g & (SW_RING-1)];
              bg = tg.x; pbg = tg.y;
            }
          fopen = act && g < 0 && x == hiR && hiR+1 < n;     // the reach may go on beyond the ring
        }
#endif
        // searches that leave the ring, one at a time, 64 segments per step
#ifdef SW_SKIP_COOP
        fopen = false;
#endif
        // (block by block, every open search looking at the block while it is in registers: searched one after the other,
        //  each search loaded its blocks from HBM again -- a round trip per open lane and block, one behind the other)
        { uint64_t um = __ballot(fopen);
          for (int j0 = ring_hi; um; j0 += WAVE)
            { const int jj = j0+lane;
              sg sj; sj.b = 0; sj.pb = 0; sj.key = -1;
              if (jj < n) sj = seg(jj);
              uint64_t still = 0;
              for (uint64_t t = um; t; t &= t-1)
                { const int src = __ffsll((long long)t)-1;
                  const int sb = __builtin_amdgcn_readlane(bi,src), sk = __builtin_amdgcn_readlane(ki,src);
                  const bool reach = jj < n && sj.pb < sb+W;   // (monotone: begins grow)
                  const uint64_t mR = __ballot(reach), mB = __ballot(reach && sj.key > sk), mE = __ballot(reach && sj.key == sk);
                  if (mB)
                    { const int fb = __ffsll((long long)mB)-1;
                      const uint64_t blt = (1ull << fb)-1;
                      const int rbg = __builtin_amdgcn_readlane(sj.b,fb), rpbg = __builtin_amdgcn_readlane(sj.pb,fb);
                      if (lane == src) { g = j0+fb; bg = rbg; pbg = rpbg; nonempty = nonempty || (mR & blt) != 0; eq = eq || (mE & blt) != 0; }
                    }
                  else
                    { if (lane == src) { nonempty = nonempty || mR != 0; eq = eq || mE != 0; }
                      if (mR == ~0ull) still |= 1ull << src;     // (else the reach ended inside these 64)
                    }
                }
              um = still;
            }
        }
        // -- backward: what the deque holds when I arrive (segments beginning beyond pb(me)-W): does any of them match
        //    or beat me (then I wipe nothing), and where does the nearest one that beats me begin (a beaten segment
        //    needs that; an expiring one only needs to know that it wipes nothing) --
        bool blocked = false, bnon = false, havep = false, bopen = false;
        int pbeg = 0;
        const int limw = pbi-W;
#ifndef SW_SKIP_BWD
        { const bool actb = act && pbi != NONE;
          int y = i;                                         // y(i) within the ring: the first segment with b > pb(me)-W (i: none)
#pragma unroll
          for (int st = SW_RING/2; st > 0; st >>= 1)
            { const int c = y-st;
              const int bc = sw_S.rbp[c & (SW_RING-1)].x;
              y = (actb && c >= loR && bc > limw) ? c : y;
            }
          bnon = y < i;
          const bool need = g >= 0;
          const int mx = bnon ? range_max(y,i-1) : -1;
          blocked = bnon && mx >= ki;
          if (need && mx > ki)                               // the nearest one before me that beats me
            { int pos = i-1;
#pragma unroll
              for (int q = 0; q < SW_LEV+SW_TOPB-2; q++)
                { const int k = q < SW_TOPB-1 ? SW_LEV-1 : SW_LEV+SW_TOPB-3-q;
                  const int m = lev(k,pos-(1 << k)+1);
                  pos = (pos-(1 << k)+1 >= y && m <= ki) ? pos-(1 << k) : pos;
                }
              havep = true;
              pbeg = sw_S.rbp[pos & (SW_RING-1)].x;
            }
          const bool fin = (blocked && !need) || havep;
          bopen = actb && !fin && y == loR && loR > 0;       // the window may go on below the ring
        }
#endif
#ifdef SW_SKIP_COOP
        bopen = false;
#endif
        { uint64_t um = __ballot(bopen);
          for (int j0 = loR-1; um; j0 -= WAVE)
            { const int jj = j0-lane;
              sg sj; sj.b = 0; sj.pb = 0; sj.key = -1;
              if (jj >= 0) sj = seg(jj);
              uint64_t still = 0;
              for (uint64_t t = um; t; t &= t-1)
                { const int src = __ffsll((long long)t)-1;
                  const int sk = __builtin_amdgcn_readlane(ki,src), slim = __builtin_amdgcn_readlane(limw,src), sgi = __builtin_amdgcn_readlane(g,src);
                  const bool sbl = __builtin_amdgcn_readlane(blocked ? 1 : 0,src) != 0;
                  const bool inw = jj >= 0 && sj.b > slim;     // (monotone)
                  const uint64_t mI = __ballot(inw), mS = __ballot(inw && sj.key > sk), mN = __ballot(inw && sj.key >= sk);
                  if (mS)
                    { const int fs = __ffsll((long long)mS)-1;
                      const int spb = __builtin_amdgcn_readlane(sj.b,fs);
                      if (lane == src) { havep = true; pbeg = spb; blocked = true; bnon = true; }
                    }
                  else
                    { const bool nbl = sbl || mN != 0;
                      if (lane == src) { bnon = bnon || mI != 0; blocked = nbl; }
                      if (mI == ~0ull && !(nbl && sgi < 0)) still |= 1ull << src;
                    }
                }
              um = still;
            }
        }
        // -- values --
        const bool wipe = act && bnon && !blocked;
        const bool isexp = act && g < 0;
        const bool flag = isexp && nonempty && !eq && !(rep && b_last < bi+W);
        int nw = 0;
        bool big = false;
        if (act && g >= 0)
          { if (havep && pbeg > pbg-W)
              { const int ci = rep ? 32767-ki : ki;
                nw = rep ? (CP_SEED_W_REP-ci > 0 ? CP_SEED_W_REP-ci : 0) : ci; big = nw > 1000;
              }
            else { nw = bg-bi; if (nw > W) nw = W; }
          }
        const uint64_t EM = __ballot(isexp), WM = __ballot(wipe), FM = __ballot(flag);
        { const uint64_t pe = EM & lt, pw = WM & (lt | (1ull << lane)), pf = FM & lt;
          const int se = pe ? 63-__clzll((long long)pe) : 0, sw = pw ? 63-__clzll((long long)pw) : 0, sf = pf ? 63-__clzll((long long)pf) : 0;
          const int bprev = __shfl(bi,se), wpb = __shfl(pbi,sw), eprev = __shfl(ei,sf);
          if (isexp)
            { const bool have = pe ? true : c_have;
              const int xb = pe ? bprev : c_expb;            // begin of the previous expiring segment
              const bool hw = pw ? true : c_wipe;
              const int lwpb = pw ? wpb : c_wpb;             // pb of the last wipe up to me
              const int pos = pf ? eprev : c_pos;
              const bool oor = have && !(hw && lwpb >= xb+W);    // no wipe after that segment's expiry step
              nw = W;
              if (oor) { nw = bi-pos+1; if (nw > W) nw = W; }
            }
          if (EM) { c_have = true; c_expb = __shfl(bi,63-__clzll((long long)EM)); }
          if (WM) { c_wipe = true; c_wpb = __shfl(pbi,63-__clzll((long long)WM)); }
          if (FM) c_pos = __shfl(ei,63-__clzll((long long)FM));
        }
        nbig += __popcll(__ballot(big));
        if (act) R.rec[i].z = nw;
      }
  }
  __syncthreads();
  if (lane == 0) sw_S.sel[SEL_NBIG] = nbig;
  __syncthreads();
}

// ---- the order of the walk: decreasing window count, segments of equal count in position order (the reference's qsort is
//      glibc's stable merge sort).  Key = 1000 - count (counts above 1000 share key 0, the skipped stretches, -10, come
//      last).  A round puts the segments of the next SW_SORTW keys in order behind those ordered so far: counters per key
//      in LDS, their prefix sums, then a stable scatter of the range's segments -- ranks among the 64 of a step from
//      ballots over the key's bits; the walk goes on from there and asks for the next round when it runs out (two rounds
//      cover all keys).  (Two 5-bit radix passes over all segments with an index array in between before: 3.4 ms of the
//      kernel's 23.)  A round can also end early, at the key that brings it to SW_SORTN segments (a group of equal counts
//      is never cut) -- in the hope that the walk stops long before the last segment.  It does not: a read is seldom
//      covered before its last groups, every round costs a pass over all segments, and 128 / 256 / 512 / 1024 / all
//      segments per round take 25.4 / 23.7 / 22.9 / 22.3 / 22.1 ms on the 60x set -- so a round takes all its keys. ----
__device__ __forceinline__ int sw_sort_key(int nw) { return nw > 1000 ? 0 : 1000-nw; }
// PACKED: two 16-bit counters per LDS word -- twice the keys per round, so ONE round covers all keys (0 .. 1010); possible
// while a read has fewer than 65536 segments (its counts and offsets then fit 16 bits).
template <bool PACKED>
__device__ __forceinline__ void sw_sort_range_t(const cp_seedw_read &R, int n, int nbig, int lane)
{ const uint64_t lt = (1ull << lane)-1;
  const int lo = sw_sel(SEL_NEXTKEY), base = sw_sel(SEL_NSORTED);
  constexpr int NKEY = PACKED ? 2*SW_SORTW : SW_SORTW;      // keys of a round
  constexpr int PER = NKEY/WAVE;                             // counters per lane
  uint16_t *sb16 = reinterpret_cast<uint16_t *>(sw_S.sbins);
  __syncthreads();
  for (int q = lane; q < SW_SORTW; q += WAVE) sw_S.sbins[q] = 0;
  __syncthreads();
  for (int q = lane; q < n; q += WAVE)
    { const int d = sw_sort_key(R.rec[q].z)-lo;
      if (d >= 0 && d < NKEY)
        { if (PACKED) atomicAdd(&sw_S.sbins[d >> 1],(d & 1) ? 0x10000 : 1);
          else atomicAdd(&sw_S.sbins[d],1);
        }
    }
  __syncthreads();
  // the range ends at the first key whose running total reaches SW_SORTN (else with the round's keys); exclusive prefix sums
  int c[PER], tot = 0;
#pragma unroll
  for (int u = 0; u < PER; u++)
    { c[u] = PACKED ? (int)sb16[PER*lane+u] : sw_S.sbins[PER*lane+u];
      tot += c[u];
    }
  int incl = tot;
  for (int o = 1; o < WAVE; o <<= 1) { const int x = __shfl_up(incl,o); if (lane >= o) incl += x; }
  int run = incl-tot, dend = -1, cend = 0;                   // this lane's first key that reaches the target, the total up to it
#pragma unroll
  for (int u = 0; u < PER; u++)
    { const int ex = run;
      run += c[u];
      if (dend < 0 && run >= SW_SORTN) { dend = PER*lane+u; cend = run; }
      c[u] = ex;
    }
  const uint64_t hit = __ballot(dend >= 0);
  int hi = NKEY-1, cnt = __shfl(incl,WAVE-1);
  if (hit) { const int src = __ffsll((long long)hit)-1; hi = __shfl(dend,src); cnt = __shfl(cend,src); }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < PER; u++)
    { if (PACKED) sb16[PER*lane+u] = (uint16_t)c[u];
      else sw_S.sbins[PER*lane+u] = c[u];
    }
  __syncthreads();
  const int nbits = hi > 0 ? 32-__clz(hi) : 0;
  for (int q0 = 0; q0 < n && cnt > 0; q0 += WAVE)
    { const int q = q0+lane;
      int4 t = make_int4(0,0,0,0);
      int d = -1;
      if (q < n) { t = R.rec[q]; d = sw_sort_key(t.z)-lo; }
      const bool part = d >= 0 && d <= hi;
      uint64_t peers = __ballot(part);
      if (peers == 0) continue;
      for (int bit = 0; bit < nbits; bit++)
        { const uint64_t bmk = __ballot(part && ((d >> bit) & 1));
          peers &= ((d >> bit) & 1) ? bmk : ~bmk;
        }
      const int rank = __popcll(peers & lt);
      int dst = 0;
      if (part) dst = base+(PACKED ? (int)sb16[d] : sw_S.sbins[d])+rank;
      __syncthreads();
      if (part && rank == 0)
        { if (PACKED) sb16[d] = (uint16_t)(sb16[d]+__popcll(peers));
          else sw_S.sbins[d] += __popcll(peers);
        }
      __syncthreads();
      if (part) R.orec[dst] = t;
    }
  __syncthreads();
  if (nbig > 0 && lo == 0)                                   // counts above 1000 share the first key with 1000: that head is put
    { if (lane == 0)                                         // in order by insertion (rare, short)
        { int m = 0;
          while (m < cnt && R.orec[m].z >= 1000) m++;
          for (int a = 1; a < m; a++)
            { const int4 x = R.orec[a];
              int j = a-1;
              while (j >= 0 && R.orec[j].z < x.z) { R.orec[j+1] = R.orec[j]; j--; }
              R.orec[j+1] = x;
            }
        }
      __syncthreads();
    }
  if (lane == 0) { sw_S.sel[SEL_NSORTED] = base+cnt; sw_S.sel[SEL_NEXTKEY] = lo+hi+1; }
  __syncthreads();
}
__device__ __forceinline__ void sw_sort_range(const cp_seedw_read &R, int n, int nbig, int lane)
{ if (n < 65536) sw_sort_range_t<true>(R,n,nbig,lane);
  else sw_sort_range_t<false>(R,n,nbig,lane);
}

__device__ __attribute__((noinline)) void sw_sort(int rep_, int lane SW_PROF_ARGS)
{ const cp_seedw_read R = sw_the_read();
  const sw_list Lm = sw_the_list(R);
  const int plen = R.plen;
  const uint64_t lt = (1ull << lane)-1;
  int n = sw_sel(SEL_N);
  const int M = sw_sel(SEL_M), nbig = sw_sel(SEL_NBIG);
  // The skipped stretches join the records after all, behind the valid segments, with the reference's -10: the sort
  // puts them last (in their own order: it is stable), and the walk does reach them when the read is still uncovered --
  // the reference then tests them against its list like any segment, and its search of that list, one slot past the
  // live part, can miss the very interval they were masked with (tests/golden/seeds.npz holds such reads).
  if (n+M > R.cap) { if (lane == 0) atomicOr(R.err,4); sw_sel_put(lane,0,0,0,0,1,Lm.big); return; }
  for (int q = lane; q < M; q += WAVE) { const int4 t = R.orec[q]; R.rec[n+q] = make_int4(t.x,t.y,-10,0); }
  n += M;
  __syncthreads();                                           // the window counts are visible to the wave
#ifdef CP_SEED_DEBUG
  if (R.dbg_read && sw_first(rep_))                                     // diagnostic builds: the repeat selection's records of one read
    { for (int q = lane; q < n && q < 8192; q += WAVE) g_seed_dbg[q] = R.rec[q];
      if (lane == 0) { g_seed_dbg_n[0] = n; g_seed_dbg_n[1] = M; g_seed_dbg_n[2] = sw_sel(SEL_BLAST); g_seed_dbg_n[3] = 0; }
    }
#endif
  SW_STAMP(5);
#ifdef CP_SEED_PROF
  if (lane == 0) { sw_t[7] += n; }
#endif
#ifdef CP_SEED_STOP_AT
  if (CP_SEED_STOP_AT == 2) { sw_sel_put(lane,n,M,nbig,0,1,Lm.big); return; }
#endif
  if (M > 0 && Lm.b(0) == 0 && Lm.e(0) == plen) { sw_sel_put(lane,n,M,nbig,0,1,Lm.big); return; }
  __syncthreads();
  if (lane == 0) { sw_S.sel[SEL_NSORTED] = 0; sw_S.sel[SEL_NEXTKEY] = 0; sw_S.sel[SEL_POS] = 0; sw_S.sel[SEL_NTAKE] = 0; sw_S.sel[SEL_MORE] = 0; }
  sw_sel_put(lane,n,M,nbig,0,0,Lm.big);
  sw_sort_range(R,n,nbig,lane);
  SW_STAMP(2);
}
// (the walk ran out of ordered segments with the read still uncovered)
__device__ __attribute__((noinline)) void sw_sort_more(int lane SW_PROF_ARGS)
{ const cp_seedw_read R = sw_the_read();
  sw_sort_range(R,sw_sel(SEL_N),0,lane);
  SW_STAMP(2);
}

template <bool rep>
__device__ __attribute__((noinline)) void sw_walk(int lane SW_PROF_ARGS)
{ const cp_seedw_read R = sw_the_read();
  sw_list Lm = sw_the_list(R);
  const int W = rep ? CP_SEED_W_REP : CP_SEED_W;
  const int plen = R.plen, Km1 = R.K-1;
  const uint64_t lt = (1ull << lane)-1;
  const int ntot = sw_sel(SEL_N), n = sw_sel(SEL_NSORTED);   // segments in all, segments put in order so far (sw_sort_range)
  int M = sw_sel(SEL_M);
  const bool was_big = Lm.big;
  // ---- selection (wave-uniform control flow) ----
  // taken segments are listed in R.tmp (idle after the sort) and marked at the end; the list is written 64 takes at a
  // time from an LDS buffer (cval), so that the list update of a take -- a chain of wave barriers -- never waits for a
  // global store of the take before it
  int ntake = sw_sel(SEL_NTAKE), nbuf = 0;
  auto flush_takes = [&]()
    { __syncthreads();
      for (int q = lane; q < 2*nbuf; q += WAVE) R.tmp[2*(ntake-nbuf)+q] = sw_S.cval[q];
      nbuf = 0;
      __syncthreads();
    };
  auto take = [&](int b, int e)                              // mask the segment with a margin of W, mark its hash minimizers
    { if (!Lm.big && M+4 >= SW_MI) sw_mi_grow(Lm,R.cap+3,lane);
      M = sw_mi_add(Lm,M,b-W > 0 ? b-W : 0,e+W < plen ? e+W : plen,lane);
      if (2*ntake+2 <= R.cap)
        { if (nbuf == WAVE) flush_takes();
          if (lane == 0) { sw_S.cval[2*nbuf] = b; sw_S.cval[2*nbuf+1] = e; }
          nbuf++; ntake++;
        }
      else                                                   // (no room in the list: marked on the spot)
        { flush_takes();
          sw_mark(R.seq,R.cls,R.state,plen+Km1,R.K,b,e,rep,lane);
        }
    };
  int pos = sw_sel(SEL_POS);
  for (; pos < n; pos++)                                     // every segment that is extreme over a whole window
    { const int4 t = R.orec[pos];
      if (t.z < W) break;
      take(t.x,t.y);
    }
  SW_STAMP(3);
  SW_STOP(4);
  // Then groups of equal window count, while uncovered.  The members of a group are tested against the list as it was
  // before the group, and the list changes only when a group with members outside it ends: so 64 records are tested
  // at once (a lane each) against the current list; everything up to the first record found outside is settled, that
  // record's group is tested to its end, its outside members are taken in order, and the walk resumes behind the group.
  // (The reference looks for full coverage after every group; coverage changes only with a take, so it is looked for
  // after the whole-window takes and after every group that took something.)
  int npend = 0, g = 0, ovpos = -1;                          // ovpos: from this record on the group's outside members are flagged in HBM
  bool ingroup = false, covered = M > 0 && Lm.b(0) == 0 && Lm.e(0) == plen;
  auto flush = [&](int gend)
    { for (int q = 0; q < npend; q++) take(sw_S.pend_b[q],sw_S.pend_e[q]);
      if (ovpos >= 0)
        { __syncthreads();
          for (int q = ovpos; q < gend; q++) { const int4 t = R.orec[q]; if (t.w < 0) take(t.x,t.y); }
        }
      npend = 0; ovpos = -1; ingroup = false;
      covered = M > 0 && Lm.b(0) == 0 && Lm.e(0) == plen;
    };
  while (pos < n && !covered)
    { const int q = pos+lane;
      const bool act = q < n;
      int4 rc = make_int4(0,0,-0x7fffffff,0);
      if (act) rc = R.orec[q];
      const int sb = rc.x, se = rc.y, nw = rc.z;
      bool outside = false;
      if (act)
        { const int idx = sw_mi_find(Lm,M,sb,se);
          outside = !(idx >= 0 && Lm.b(idx) <= sb && se <= Lm.e(idx));
        }
      const int len = n-pos < WAVE ? n-pos : WAVE;
      int k0 = 0;
      if (!ingroup)
        { const uint64_t om = __ballot(outside);
          if (om == 0) { pos += len; continue; }
          k0 = __ffsll((long long)om)-1;
          g = __shfl(nw,k0);
        }
      // members of group g from lane k0 on: a contiguous run (the records are sorted)
      const uint64_t notg = __ballot(!(act && nw == g)) & ~((1ull << k0)-1);
      const int endl = notg ? __ffsll((long long)notg)-1 : WAVE;
      const bool mine = outside && lane >= k0 && lane < endl;
      const uint64_t pm = __ballot(mine);
      if (ovpos < 0 && npend+__popcll(pm) > SW_PEND) ovpos = pos;
      __syncthreads();
      if (ovpos >= 0) { if (act) R.orec[q].w = mine ? -1 : 0; }
      else if (mine) { const int s = npend+__popcll(pm & lt); sw_S.pend_b[s] = sb; sw_S.pend_e[s] = se; }
      __syncthreads();
      if (ovpos < 0) npend += __popcll(pm);
      if (endl == WAVE && pos+WAVE < n) { ingroup = true; pos += WAVE; continue; }      // the group runs on into the next chunk
      pos += endl < len ? endl : len;
      flush(pos);
    }
  SW_STAMP(4);
  SW_STOP(5);
  flush_takes();                                             // the list of taken segments is complete and visible to the wave
  if (!covered && n < ntot)                                  // out of ordered segments: the next range of counts, then on from here
    { if (lane == 0)
        { sw_S.sel[SEL_M] = M; sw_S.sel[SEL_POS] = pos; sw_S.sel[SEL_NTAKE] = ntake; sw_S.sel[SEL_MORE] = 1;
          sw_S.lm_big = Lm.big ? 1 : 0;
        }
      __syncthreads();
      return;
    }
  if (lane == 0) sw_S.sel[SEL_MORE] = 0;
#ifdef CP_SEED_DEBUG_TAKES
  if (R.dbg_read && rep)
    { int *d = (int *)g_seed_dbg;
      for (int q = lane; q < 2*ntake && q < 4*8192; q += WAVE) d[q] = R.tmp[q];
      if (lane == 0) { g_seed_dbg_n[0] = ntake; g_seed_dbg_n[1] = M; g_seed_dbg_n[2] = n; g_seed_dbg_n[3] = 0; }
    }
#endif
  sw_mark_all(R.seq,R.cls,R.state,R.K,plen+Km1,R.tmp,ntake,rep,lane);
#ifdef CP_SEED_PROF
  { int tk = 0; for (int q = lane; q < ntake; q += WAVE) tk += R.tmp[2*q+1]-R.tmp[2*q];
    for (int o = 32; o > 0; o >>= 1) tk += __shfl_xor(tk,o);
    if (lane == 0) { sw_t[6] += tk; }
  }
#endif
  SW_STAMP(3);
  __syncthreads();                                           // the marks are visible to the next selection
  if (Lm.big != was_big)
    { if (lane == 0) sw_S.lm_big = 1;
      __syncthreads();
    }
}

// one selection (seed.c:190-476 with C = 'H'/'D'; seed.c:667-951 with C = 0)
template <bool rep>
__device__ __forceinline__ void sw_select(int C, int nrep, int rep_big, int lane SW_PROF_ARGS)
{ sw_segments<rep>(C,nrep,rep_big,lane SW_PROF_PASS);
  SW_STOP(1);
  if (sw_sel(SEL_STOP)) return;
  sw_windows<rep>(lane SW_PROF_PASS);
  sw_sort(rep ? 1 : 0,lane SW_PROF_PASS);
  SW_STOP(3);
  if (sw_sel(SEL_STOP)) return;
  for (;;)
    { sw_walk<rep>(lane SW_PROF_PASS);
      if (!sw_sel(SEL_MORE)) break;
      sw_sort_more(lane SW_PROF_PASS);
    }
}

// ---- the whole path for one read.  R.state holds 'E' at every k-mer on entry.  Returns the number of .rep intervals
//      (wave-uniform); the first rep_cap of them are in R.rep_pairs. ----
__device__ __forceinline__ int sw_find_seeds(const cp_seedw_read &R, int lane SW_PROF_ARGS)
{ const int plen = R.plen, K = R.K, Km1 = K-1;
  if (plen <= 0) return 0;
  for (int q = lane; q < SW_MI; q += WAVE) { sw_S.mi_b[q] = 0; sw_S.mi_e[q] = 0; }     // defined start state of the list
  if (lane == 0) { sw_S.R = R; sw_S.lm_big = 0; }            // (visible to the selections: barriers below)
  // ---- unique / repetitive stretches -> .rep intervals (seed.c:482-566): label runs from a ballot, lane 0 over the runs ----
  const int min_uniq = (int)(K*2.5);
  int nrep = 0, rs = -1, i0 = 0, normal = 0, lrun_s = 0; bool inR = false; char lrun_c = 0;
  auto emit = [&](int b, int e)                              // a repetitive stretch, k-mer coordinates
    { if (nrep < R.rep_cap) { R.rep_pairs[2*nrep] = b+Km1; R.rep_pairs[2*nrep+1] = e+Km1; }      // read coordinates: the first K-1 bases belong to the first k-mer
      if (nrep < SW_REP) { sw_S.rep[2*nrep] = b; sw_S.rep[2*nrep+1] = e; }
      nrep++;
    };
  auto close_stretch = [&](int e)                            // the non-R stretch [i0,e) ends: unique if it holds >= 2.5 K H/D positions
    { if (normal >= min_uniq) { if (rs >= 0 && rs < i0) emit(rs,i0); rs = e; }
      else if (rs < 0) rs = i0;
    };
  auto label_run = [&](int s, char c, int len, bool first)   // lane 0: the label run [s,s+len) of class c
    { if (c == 'R') { if (first) rs = 0; else if (!inR) close_stretch(s); inR = true; }
      else
        { if (inR || first) { i0 = s; normal = 0; inR = false; }
          if (c == 'H' || c == 'D') normal += len;
        }
    };
  // (four chunks of 64 labels per step, their loads issued together: chunk by chunk the loop was one HBM round trip per
  //  64 positions with nothing to do in between -- 12 % of the kernel's time for 3 % of its instructions)
  constexpr int AN = 4;
  char cprev = 0;                                            // the label before the chunk (carried from chunk to chunk)
  for (int e0 = 0; e0 < plen; e0 += AN*WAVE)
    { char cc[AN];
#pragma unroll
      for (int u = 0; u < AN; u++)
        { const int p = e0+u*WAVE+lane;
          cc[u] = p < plen ? R.cls[p] : (char)0;
        }
#pragma unroll
      for (int u = 0; u < AN; u++)
        { const int c0 = e0+u*WAVE, p = c0+lane;
          if (c0 >= plen) break;
          const bool in = p < plen;
          const char c = cc[u];
          const int cpi = sw_from_below((int)c,(int)cprev);
          cprev = (char)sw_of_last((int)c);                  // (the next chunk's lane 0 uses it)
          const uint64_t sm = __ballot(in && (p == 0 || (int)c != cpi));
          if (sm == 0) continue;
          __syncthreads();
          sw_S.cval[lane] = c;
          __syncthreads();
          if (lane == 0)
            { uint64_t m = sm;
              while (m)
                { const int j = __ffsll((long long)m)-1;
                  m &= m-1;
                  const int s = c0+j;
                  if (s > 0) label_run(lrun_s,lrun_c,s-lrun_s,lrun_s == 0);
                  lrun_s = s; lrun_c = (char)sw_S.cval[j];
                }
            }
        }
    }
  if (lane == 0)
    { label_run(lrun_s,lrun_c,plen-lrun_s,lrun_s == 0);
      if (!inR) close_stretch(plen);
      if (rs >= 0 && rs < plen) emit(rs,plen);
    }
  __syncthreads();
  SW_STAMP(0);
#ifdef CP_SEED_STOP_AT
  if (CP_SEED_STOP_AT == 0) return 0;
#endif
  nrep = sw_first(nrep);
  const bool rep_big = nrep > SW_REP;                        // then the HBM list is searched (complete: rep_cap bounds the label runs)
  const int nr = nrep > R.rep_cap ? R.rep_cap : nrep;
  for (int sel = 0; sel < 2; sel++)                          // H, then D: one copy of the code
    sw_select<false>(sel == 0 ? 'H' : 'D',nr,rep_big ? 1 : 0,lane SW_PROF_PASS);
  sw_select<true>(0,nr,rep_big ? 1 : 0,lane SW_PROF_PASS);
  return nrep;
}
