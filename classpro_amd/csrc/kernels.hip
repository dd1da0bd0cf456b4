// kernels.hip -- HIP kernels of the read-batched classification pipeline (gfx950 / CDNA4, wave64).
//
// Layout in HBM (see include/classpro_amd.h): all reads of a batch are concatenated; read r owns
// prof[prof_off[r] .. prof_off[r+1]) and seq[seq_off[r] .. seq_off[r+1]).  Per-read scratch is
// addressed with the same offsets:
//   bitmap   1 bit per profile position over the WHOLE concatenated profile (bit g = candidate at g)
//   wall     1 byte per position, read r at wall + prof_off[r] + r        (plen+1 cells per read; only reads whose flags do not fit on chip use it)
//   hkeys/hvals  per-read open-addressing table position -> 4 memoised probabilities (perror), read r at hoff[r]
//   eintvl / ointvl   E-/O-interval lists, read r at eoff[r], capacity eoff[r+1]-eoff[r]
//   intvl / rintvl / relmap / DP scratch   read r at ioff[r], capacity ioff[r+1]-ioff[r]
//
// Kernels:
//   k_scan_candidates   streaming pass over the profile (the HBM-roofline kernel): wall.c:590-607
//   k_count_caps        per-read candidate count -> scratch capacities
//   k_prefix_caps_mb    exclusive prefix sums of the capacities (one block per 1024 reads, single pass with look-back)
//   k_wall_tasks        one wave per read: the read-only part of the candidate walk, wall.c:590-707 (lists + task results to HBM)
//   k_find_wall         one wave per read: the walk's replay (dependency rounds, flags in LDS) and everything after it, wall.c:639-958
//   k_find_rel          one wave per read, one lane per interval: wall.c:960-1051
//   k_classify_rel_grp   4 reads per wave (1 for M > 128), 8 lanes per direction: class_rel.c:871-963
//   k_classify_unrel_grp 2 reads per wave, speculative update slots committed in order: class_unrel.c:248-300
//   k_classify_rel / k_classify_unrel   sequential forms for reads with > 1024 intervals or > 65535 k-mers
//   k_skellam_table     the table of logp_trans values (cp_types.h), filled once per cp_params
//   k_seed_caps / k_find_seeds   the -s seed path, one wave per read (cp_seed_wave.h): seed.c:966-1032
//   k_paint_labels      one wave per read: ClassPro.c:116-119,265-271
//   k_seq_context       dense context arrays (stage API / parity tests only): context.c:8-108
//   k_decode_profiles   FASTK code strings -> counts on the device: libfastk.c:1467-1534
//
// No MFMA anywhere: the path has no dense contraction.  Built with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "cp_wall.h"
#include "cp_class.h"
#include "cp_seed.h"

#ifdef CP_BOUNDS
__device__ unsigned long long g_bounds[4];               // cp_bounds.h
#endif
#define WAVE 64
#define REL_MAXM 1024           // reads with more reliable intervals use the sequential kernel
#define UNREL_MAXN 1024         // reads with more intervals use the sequential kernel
#ifndef REL_SMALL_MAXM
#define REL_SMALL_MAXM 112      // the main size class of k_classify_rel_grp (four reads per wave): M <= this
#endif
#ifndef UNREL_SMALL_MAXN
#define UNREL_SMALL_MAXN 256    // the main size class of k_classify_unrel_grp (two reads per wave): N <= this
#endif
#define GRP_MAX_PLEN 65535      // the lane-parallel classify kernels keep interval ends as 16-bit LDS fields; a longer read
                                // (a Dazzler database may hold them: ClassPro.c:87,110 sizes by db->maxlen) takes the sequential kernels

// Make one lane's global stores visible to the other lanes of the same wave (blocks are one wave).
__device__ __forceinline__ void wave_sync() { __syncthreads(); }

__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE-1); }
// the value of the lane below (lane 0: `first`) as a DPP move (wave_shr:1), the last lane's value as a readlane: neither
// goes through the LDS crossbar as __shfl_up / __shfl do
__device__ __forceinline__ int wave_from_below(int v, int first) { return __builtin_amdgcn_update_dpp(first,v,0x138,0xf,0xf,false); }
__device__ __forceinline__ int wave_of_last(int v) { return __builtin_amdgcn_readlane(v,WAVE-1); }

// ---------------------------------------------------------------------------------------------
//  k_scan_candidates: bit g of the bitmap <=> g >= 1, min(c[g-1],c[g]) < R and |c[g-1]-c[g]| >= 3
//  (wall.c:592-608).  Each lane loads 8 consecutive counts with one 16-byte load, gets the count
//  before its first one from the neighbouring lane, and stores one byte of flags.  Bits at the
//  first position of a read compare across a read boundary; consumers skip position 0.
//  Algorithmic traffic: 2 B read + 1/8 B written per position.  The bitmap buffer has `nbytes` bytes (whole
//  64-bit words plus one spare word); the bytes after the last group are cleared here.
//  Round 5, what separates this kernel (5.0-5.4 TB/s) from a read-only kernel of the same access pattern (6.7-7.0 TB/s on
//  4 GB: scripts/microbench/read_bw.hip, scan_variants.hip; profiles/r05_scan_variants.txt): the flag arithmetic costs
//  8 % (6.1-6.2 TB/s without the stores) and the bitmap stores -- 1/16 of the bytes -- 16 %: writes interleaved with the
//  read stream cost HBM efficiency out of proportion to their size, whatever their width (a byte per lane, dwords gathered
//  by DPP, dwordx4 through LDS) or cache policy; storing one row in 4 / 16 / 64 takes back 50 / 70 / 90 % of the loss.  The
//  full-width form gained 4 % in the micro-benchmark and nothing on the bench's profiles (779-791 us against 784-788 per
//  4.0-GB launch, A/B in one call), so the simple form stays.  The bitmap IS the output: a sparser one (candidate lists,
//  1/8 of the bytes) would buy back about half of the 16 % of a kernel that is 8 % of a step.
// ---------------------------------------------------------------------------------------------
typedef unsigned cp_u4v __attribute__((ext_vector_type(4)));
// The profile is read once and the bitmap written once per batch: both go past the caches as streaming accesses
// (non-temporal loads AND stores together: 5.1-5.4 TB/s against 5.0-5.2 with plain ones; either alone changes nothing).
#ifndef SCAN_PLAIN
__device__ __forceinline__ uint4 scan_load(const uint4 *p)
{ cp_u4v x = __builtin_nontemporal_load(reinterpret_cast<const cp_u4v *>(p)); return make_uint4(x.x,x.y,x.z,x.w); }
#define SCAN_LOAD(p) scan_load(p)
#define SCAN_STORE(p,v) __builtin_nontemporal_store((uint8_t)(v),(p))
#else
#define SCAN_LOAD(p) (*(p))
#define SCAN_STORE(p,v) (*(p) = (uint8_t)(v))
#endif
#ifndef SCAN_UNROLL
#define SCAN_UNROLL 4
#endif
#ifndef SCAN_BLOCKS_PER_CU
#define SCAN_BLOCKS_PER_CU 8
#endif

// per 16-bit half of (prevcur, cur): 1 if min < R and |difference| >= 3, else 0 (wall.c:592-608)
typedef unsigned short cp_us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned scan_pair_flags(unsigned prevcur, unsigned cur, int rep)
{ const cp_us2 a = __builtin_bit_cast(cp_us2,prevcur), b = __builtin_bit_cast(cp_us2,cur);
  const cp_us2 mn = __builtin_elementwise_min(a,b), mx = __builtin_elementwise_max(a,b);
  const cp_us2 r2 = { (unsigned short)rep, (unsigned short)rep }, two = { 2, 2 }, one = { 1, 1 };
  const cp_us2 s1 = __builtin_elementwise_sub_sat(r2,mn);              // != 0 <=> min < R
  const cp_us2 s2 = __builtin_elementwise_sub_sat((cp_us2)(mx-mn),two);  // != 0 <=> difference >= 3
  const cp_us2 m = __builtin_elementwise_min(__builtin_elementwise_min(s1,s2),one);
  return __builtin_bit_cast(unsigned,m);
}

__global__ void __launch_bounds__(256)
k_scan_candidates(const uint16_t *__restrict__ prof, int64_t total, int rep, uint8_t *__restrict__ bitmap, int64_t nbytes, int has_prev)
{ // has_prev: this launch continues a profile (capi.hip: launch_scan cuts a batch into launches of at most 2^31
  // positions): the count before its first position is prof[-1], as it would be inside a single launch
  const int64_t ngroups = total >> 3;                   // full groups of 8 positions
  const int64_t nthreads = (int64_t)gridDim.x*blockDim.x;
  const int lane = lane_id();
  const uint4 *vp = reinterpret_cast<const uint4 *>(prof);

#ifndef SCAN_INTERLEAVED
  // every wave owns SCAN_UNROLL consecutive 1-KB rows, so that lane 0 takes the count before its group from
  // lane 63 of the previous row instead of a second (2-byte) load per row
  const int64_t wavesz = (int64_t)WAVE*SCAN_UNROLL;
  const int64_t nwaves = nthreads/WAVE;
  const int64_t wid = ((int64_t)blockIdx.x*blockDim.x+threadIdx.x)/WAVE;
  for (int64_t base = wid*wavesz; base < ngroups; base += nwaves*wavesz)
    { uint4 v[SCAN_UNROLL];
      int64_t g[SCAN_UNROLL];
#pragma unroll
      for (int u = 0; u < SCAN_UNROLL; u++)
        { g[u] = base+(int64_t)u*WAVE+lane;
          v[u] = (g[u] < ngroups) ? SCAN_LOAD(&vp[g[u]]) : make_uint4(0,0,0,0);
        }
      unsigned carry = 0;
      if (lane == 0)
        carry = (base > 0 || has_prev) ? prof[base*8-1] : (v[0].x & 0xffff);
#pragma unroll
      for (int u = 0; u < SCAN_UNROLL; u++)
        { const bool live = g[u] < ngroups;
          // packed 16-bit arithmetic, two counts per instruction: pair k of a dword holds (count, its
          // predecessor); min < R and |difference| >= 3 become "both saturating differences non-zero"
          // (DPP move / readlane; as __shfl_up / __shfl through the LDS crossbar the kernel ran at the same rate: A/B 0.64 both)
          const unsigned before = (unsigned)wave_from_below((int)v[u].w,(int)(carry << 16));   // the dword that ends with this lane's predecessor
          carry = (unsigned)wave_of_last((int)v[u].w) >> 16;          // lane 0's predecessor in the next row
          const unsigned w0 = scan_pair_flags(__builtin_amdgcn_alignbit(v[u].x,before,16),v[u].x,rep);
          const unsigned w1 = scan_pair_flags(__builtin_amdgcn_alignbit(v[u].y,v[u].x,16),v[u].y,rep);
          const unsigned w2 = scan_pair_flags(__builtin_amdgcn_alignbit(v[u].z,v[u].y,16),v[u].z,rep);
          const unsigned w3 = scan_pair_flags(__builtin_amdgcn_alignbit(v[u].w,v[u].z,16),v[u].w,rep);
          const unsigned t = w0 | (w1 << 2) | (w2 << 4) | (w3 << 6);  // flags of the even counts in bits 0,2,4,6, odd ones 16 higher
          const unsigned bits = (t & 0x55u) | ((t >> 15) & 0xaau);
          if (live)
            SCAN_STORE(&bitmap[g[u]],bits);
        }
    }
#else
  for (int64_t base = (int64_t)blockIdx.x*blockDim.x*SCAN_UNROLL; base < ngroups; base += nthreads*SCAN_UNROLL)
    { uint4 v[SCAN_UNROLL];
      int64_t g[SCAN_UNROLL];
#pragma unroll
      for (int u = 0; u < SCAN_UNROLL; u++)
        { g[u] = base+(int64_t)u*blockDim.x+threadIdx.x;
          if (g[u] < ngroups)
            v[u] = SCAN_LOAD(&vp[g[u]]);
          else
            v[u] = make_uint4(0,0,0,0);
        }
#pragma unroll
      for (int u = 0; u < SCAN_UNROLL; u++)
        { const bool live = g[u] < ngroups;
          unsigned last = v[u].w >> 16;                 // count at 8g+7
          unsigned prev = __shfl_up(last,1);            // all lanes take part
          if (lane == 0 && live)
            prev = (g[u] > 0 || has_prev) ? prof[g[u]*8-1] : (v[u].x & 0xffff);
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
            SCAN_STORE(&bitmap[g[u]],bits);
        }
    }

#endif

  if (blockIdx.x == 0 && threadIdx.x == 0)              // ragged tail (< 8 positions) + zero pad to a word
    { int64_t p0 = ngroups << 3;
      if (p0 < total)
        { unsigned bits = 0;
          for (int64_t p = p0; p < total; p++)
            if (p > 0 || has_prev)
              { unsigned a = prof[p-1], b = prof[p];
                unsigned mn = a < b ? a : b, df = a < b ? b-a : a-b;
                if (mn < (unsigned)rep && df >= CP_MIN_CNT_CHANGE) bits |= 1u << (p-p0);
              }
          bitmap[ngroups] = (uint8_t)bits;
          p0 += 8;
        }
      for (int64_t q = p0 >> 3; q < nbytes; q++)          // zero pad to the end of the bitmap's last words
        bitmap[q] = 0;
    }
}

// ---------------------------------------------------------------------------------------------
//  Candidate count per read and scratch capacities.
//    N (intervals)  <= 2*ncand+3   (boundaries are O-walls = candidates, or E-interval endpoints)
//    E-list entries <= 16*ncand+64 (checked at run time; overflow is reported, never silent)
//    perror slots   <= 3*ncand+2   (a candidate's own position + one low-complexity partner per error type)
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
             int32_t *__restrict__ ncand, int64_t *__restrict__ icap, int64_t *__restrict__ ecap,
             int64_t *__restrict__ hcap)
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
      int64_t h = 32;                                   // perror tables (one per error type): power of two >= 4*ncand+16
      while (h < 4*(int64_t)cnt+16) h <<= 1;
      hcap[r] = 2*h;
    }
}

// Exclusive prefix sum of one value per thread over a 256-thread block (4 waves): wave scans by shuffle, the 4 wave
// totals combined by every thread; *total gets the block sum.  `tmp` holds 4 values.
template <class T>
__device__ __forceinline__ T block_scan_excl_256(T x, T *tmp, T *total)
{ const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  T inc = x;
  for (int o = 1; o < WAVE; o <<= 1)
    { T y = __shfl_up(inc,o); if (lane >= o) inc += y; }
  __syncthreads();                                        // (tmp may still be read from an earlier call)
  if (lane == WAVE-1) tmp[wv] = inc;
  __syncthreads();
  T base = 0, all = 0;
#pragma unroll
  for (int w = 0; w < 4; w++) { const T v = tmp[w]; if (w < wv) base += v; all += v; }
  *total = all;
  return base+inc-x;
}

// Exclusive prefix sums of three int64 arrays, in place, totals appended at [n]; one 256-thread block per 1024 values,
// four consecutive values per thread (a single 1024-thread block doing all of it sat, for 0.2-2.4 ms, in front of the one
// host round trip of every sub-batch, starved by the other stream's 50 000-block kernels -- and so did 1024-thread
// blocks of a multi-block form: 16 free wave slots on one CU are rare beside those kernels; profiles/r03_timeline_*.txt).
// Single pass: a block scans its tile, publishes its three tile sums in `state` (tagged with this launch's epoch, so the
// array is never cleared) and then looks back: lane j of its first wave waits for and reads tile j's sums, 64 tiles
// per round, and a wave reduction gives the block its offset.  A tile only waits for tiles with smaller block ids, which
// the dispatcher started before it.  The last tile writes the totals.
#define SCAN_TILE 1024
// A tile's three sums are published as three 8-byte granules {sum (40 bits), launch tag (24 bits)}: one agent-scope store
// each carries data and tag together, so neither side needs a fence.  (With a separate flag behind a __threadfence()
// every tile wrote back its XCD's whole L2 -- dirtied all the while by the other stream's kernels -- and the kernel took
// 0.8 ms in the pipeline for 10 us of work.)  A tile's sums stay below 2^40 because a whole batch's do: the host refuses
// a batch whose capacities could add up to more (capi.hip: cp_run_stages, CP_MAX_BATCH_KMERS).  The tag runs from 1 to
// 2^24-1; the host clears the array before it starts over (launch_prefix_caps), and a freshly cleared array (tag 0)
// is valid for any launch.
#define SCAN_TAG_BITS 24
#define SCAN_TAG_MASK 0xffffffull
struct cp_scan_state { unsigned long long g[3]; unsigned long long pad_; };
// One WAVE per tile of 1024 values, 16 consecutive values per lane: beside the other stream's 50 000-block kernels a
// single free wave slot turns up at once, four on one CU (a 256-thread block) only now and then -- the 256-thread form
// of this kernel still took 1 ms in the pipeline (profiles/r03_timeline_*.txt), all of it waiting to be placed.
__global__ void __launch_bounds__(WAVE)
k_prefix_caps_mb(int64_t *__restrict__ a, int64_t *__restrict__ b, int64_t *__restrict__ c, int n, int64_t *__restrict__ totals,
                 cp_scan_state *__restrict__ state, int epoch, unsigned long long *__restrict__ host_tot)
{ const int t = threadIdx.x, tile = blockIdx.x, i0 = tile*SCAN_TILE+16*t;
  int64_t *arr[3] = { a, b, c };
  int64_t off[3], sum[3], pre[3];
#pragma unroll
  for (int q = 0; q < 3; q++)
    { int64_t s16 = 0;
      for (int k = 0; k < 16; k++) s16 += (i0+k < n) ? arr[q][i0+k] : 0;
      int64_t inc = s16;
      for (int o = 1; o < WAVE; o <<= 1) { const int64_t y = __shfl_up(inc,o); if (t >= o) inc += y; }
      off[q] = inc-s16;
      sum[q] = __shfl(inc,WAVE-1);
    }
  if (t < 3)
    { const int64_t mine = t == 0 ? sum[0] : t == 1 ? sum[1] : sum[2];
      __hip_atomic_store(&state[tile].g[t],((unsigned long long)mine << SCAN_TAG_BITS) | (unsigned)epoch,__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_AGENT);
    }
  { int64_t p0 = 0, p1 = 0, p2 = 0;                       // look back, 64 tiles per round
    for (int base = 0; base < tile; base += WAVE)
      { const int j = base+t;
        int64_t v0 = 0, v1 = 0, v2 = 0;
        if (j < tile)
          { unsigned long long g0, g1, g2;
            while (((g0 = __hip_atomic_load(&state[j].g[0],__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_AGENT)) & SCAN_TAG_MASK) != (unsigned)epoch)
              __builtin_amdgcn_s_sleep(1);
            while (((g1 = __hip_atomic_load(&state[j].g[1],__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_AGENT)) & SCAN_TAG_MASK) != (unsigned)epoch)
              __builtin_amdgcn_s_sleep(1);
            while (((g2 = __hip_atomic_load(&state[j].g[2],__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_AGENT)) & SCAN_TAG_MASK) != (unsigned)epoch)
              __builtin_amdgcn_s_sleep(1);
            v0 = (int64_t)(g0 >> SCAN_TAG_BITS); v1 = (int64_t)(g1 >> SCAN_TAG_BITS); v2 = (int64_t)(g2 >> SCAN_TAG_BITS);
          }
        for (int o = 32; o > 0; o >>= 1)
          { v0 += __shfl_xor(v0,o); v1 += __shfl_xor(v1,o); v2 += __shfl_xor(v2,o); }
        p0 += v0; p1 += v1; p2 += v2;
      }
    pre[0] = p0; pre[1] = p1; pre[2] = p2;
  }
#pragma unroll
  for (int q = 0; q < 3; q++)
    { int64_t run = pre[q]+off[q];
      for (int k = 0; k < 16; k++)
        if (i0+k < n) { const int64_t x = arr[q][i0+k]; arr[q][i0+k] = run; run += x; }
      if (tile == (int)gridDim.x-1 && t == 0)
        { arr[q][n] = pre[q]+sum[q];
          if (totals) totals[q] = pre[q]+sum[q];
          // the totals straight into pinned host memory, value and launch tag in one 8-byte store: the host polls for
          // them (capi.hip) instead of paying a copy engine's start-up and an interrupt's wake-up in every sub-batch
          if (host_tot)
            __hip_atomic_store(&host_tot[q],((unsigned long long)(pre[q]+sum[q]) << SCAN_TAG_BITS) | (unsigned)epoch,
                               __ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ---------------------------------------------------------------------------------------------
//  k_order_by_work: perm[] = read ids sorted by a work key (candidates / intervals / reliable
//  intervals), largest first.  The per-read kernels are latency-bound chains whose length grows with
//  the key: launching long reads first removes the tail, and reads that share a wave (rel: 2,
//  unrel: 8) get similar trip counts.  Counting sort: histogram, suffix sums, scatter (order inside a
//  bin is arbitrary; reads are independent, so results do not depend on it).
// ---------------------------------------------------------------------------------------------
#define ORDER_BINS 1024
#define ORDER_TILE 1024
// One wave per 1024 reads (a single 1024-thread block took 0.2-0.9 ms in front of a 50 000-block kernel, three times
// per sub-batch): k_order_hist adds every block's LDS histogram to the global one; k_order_scatter turns the global
// histogram into bin starts (every block for itself: 1024 values), reserves, per block and non-empty bin, a range of
// the bin with ONE global atomic, and places its reads there.  `ghist` and `gcur` (ORDER_BINS ints each, adjacent, a
// counter of finished blocks behind them) are zero on entry: the last block of the previous sort cleared them.
// `long_off` (the reads' profile offsets, or NULL): a read of more than GRP_MAX_PLEN k-mers goes to the top bin whatever
// its key -- the sequential classify kernels, which take such reads and the reads with key > 1024, then find their whole
// class at the front of `perm` (order_top_bin) and stop at the first read of a lower bin.
__device__ __forceinline__ int order_bin(const int32_t *__restrict__ key, int i, int shift, const int64_t *__restrict__ long_off)
{ int b = key[i] >> shift;
  if (b >= ORDER_BINS) b = ORDER_BINS-1;
  if (long_off && long_off[i+1]-long_off[i] > GRP_MAX_PLEN && key[i] > 0) b = ORDER_BINS-1;
  return b;
}

__global__ void __launch_bounds__(WAVE)
k_order_hist(const int32_t *__restrict__ key, int n, int shift, int32_t *__restrict__ ghist, const int64_t *__restrict__ long_off)
{ __shared__ int hist[ORDER_BINS];
  const int t = threadIdx.x, i0 = blockIdx.x*ORDER_TILE;
  for (int k = t; k < ORDER_BINS; k += WAVE) hist[k] = 0;
  __syncthreads();
  for (int i = i0+t; i < i0+ORDER_TILE && i < n; i += WAVE)
    atomicAdd(&hist[order_bin(key,i,shift,long_off)],1);

  __syncthreads();
  for (int k = t; k < ORDER_BINS; k += WAVE)
    if (hist[k]) atomicAdd(&ghist[k],hist[k]);
}

__global__ void __launch_bounds__(WAVE)
k_order_scatter(const int32_t *__restrict__ key, int n, int shift, int32_t *__restrict__ ghist, int32_t *__restrict__ gcur,
                int32_t *__restrict__ perm, const int64_t *__restrict__ long_off)
{ __shared__ int hist[ORDER_BINS];                        // this block's count per bin, then its cursor inside the bin
  __shared__ int start[ORDER_BINS];
  const int t = threadIdx.x, i0 = blockIdx.x*ORDER_TILE;
  for (int k = t; k < ORDER_BINS; k += WAVE) hist[k] = 0;
  { // descending keys: bin b starts after all larger bins.  Lane t owns the sixteen bins 1023-16t .. 1008-16t
    int g16[16], s16 = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) { g16[k] = ghist[ORDER_BINS-1-(16*t+k)]; s16 += g16[k]; }
    int inc = s16;
    for (int o = 1; o < WAVE; o <<= 1) { const int y = __shfl_up(inc,o); if (t >= o) inc += y; }
    int run = inc-s16;
#pragma unroll
    for (int k = 0; k < 16; k++) { start[ORDER_BINS-1-(16*t+k)] = run; run += g16[k]; }
  }
  __syncthreads();
  int b[16], r[16];
#pragma unroll
  for (int k = 0; k < 16; k++)
    { const int i = i0+t+WAVE*k;
      b[k] = -1; r[k] = 0;
      if (i < n)
        { const int q = order_bin(key,i,shift,long_off);
          b[k] = q;
          r[k] = atomicAdd(&hist[q],1);                   // rank inside this block's share of the bin
        }
    }
  __syncthreads();
  for (int k = t; k < ORDER_BINS; k += WAVE)
    if (hist[k]) start[k] += atomicAdd(&gcur[k],hist[k]); // this block's range of bin k
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 16; k++)
    if (b[k] >= 0) perm[start[b[k]]+r[k]] = i0+t+WAVE*k;
  // the last block to get here leaves the histogram and the cursors zero for the next sort (a counter behind them
  // counts the blocks that are done with both): no clearing launch in front of every sort
  __shared__ int s_last;
  __syncthreads();
  if (t == 0) s_last = atomicAdd(&gcur[ORDER_BINS],1) == (int)gridDim.x-1;
  __syncthreads();
  if (s_last)
    { for (int k = t; k < 2*ORDER_BINS; k += WAVE) ghist[k] = 0;      // (gcur follows ghist)
      if (t == 0) gcur[ORDER_BINS] = 0;
    }
}

// ---------------------------------------------------------------------------------------------
//  The E-interval list of k_find_wall.  Every phase after the candidate walk -- sort + dedupe, the multi-error search,
//  the merge, the interval records -- is a chain of dependent reads of this list (binary searches, rank sorts, a
//  sequential merge): in HBM each of them is a round trip of a microsecond or more, and a wave waiting on it holds its
//  registers and LDS all the while (what the pipeline as a whole runs out of: profiles/r03_*).  A read has a few dozen
//  E-intervals, so the list lives in the wave's LDS (FW_EVL slots) and nothing of it goes to HBM at all; a list that
//  outgrows the slots moves to the read's HBM array (capacity ecap) and the same code goes on there.
// ---------------------------------------------------------------------------------------------
#ifndef FW_EVL
#ifndef FW_EVL
#define FW_EVL 128
#endif
#endif
struct fw_evl
  { cp_eintvl *g;                                       // the read's array in HBM
    CP_LDS_PTR(cp_eintvl) l;                            // the wave's FW_EVL on-chip slots (null: HBM only)
    int big;                                            // the list is in g[]
    __device__ __forceinline__ fw_evl(cp_eintvl *q = nullptr) : g(q), l(nullptr), big(1) {}
    __device__ __forceinline__ int       b(int i) const  { return big ? g[i].b : l[i].b; }
    __device__ __forceinline__ int       e(int i) const  { return big ? g[i].e : l[i].e; }
    __device__ __forceinline__ double    pe(int i) const { return big ? g[i].pe : l[i].pe; }
    __device__ __forceinline__ cp_eintvl get(int i) const { return big ? g[i] : l[i]; }
    // one lane's append (the walk's replay, the merge): the list moves to HBM when slot i is beyond the on-chip ones
    __device__ __forceinline__ void put(int i, const cp_eintvl &v)
    { if (!big && i >= FW_EVL)
        { for (int k = 0; k < FW_EVL; k++) g[k] = l[k];
          big = 1;
        }
      if (big) g[i] = v; else l[i] = v;
    }
    // all lanes together, before slot i is written by one of them
    __device__ __forceinline__ void wave_grow(int i)
    { if (!big && i >= FW_EVL)
        { __syncthreads();
          for (int k = lane_id(); k < FW_EVL; k += WAVE) g[k] = l[k];
          big = 1;
          __syncthreads();
        }
    }
  };

// Stable rank sort by (b,e) across the lanes of a wave (wall.c:519-528,734,875,910), optionally keeping only the first
// of each run of equal (b,e) (wall.c:548-568).  On-chip list (n <= FW_EVL <= 128): a lane holds up to two elements in
// registers, ranks them against the whole list (broadcast reads) and writes them back in place.  Returns the new length.
__device__ int wave_sort_ev_lds(CP_LDS_PTR(cp_eintvl) v, int n, bool dedupe)
{ static_assert(FW_EVL <= 2*WAVE,"two elements per lane");
  if (n < 2) return n;
  const int lane = lane_id();
  const uint64_t lt = (1ull << lane)-1;
  cp_eintvl x0 = { 0, 0, 0. }, x1 = { 0, 0, 0. };
  if (lane < n) x0 = v[lane];
  if (lane+WAVE < n) x1 = v[lane+WAVE];
  int r0 = 0, r1 = 0;
  for (int m = 0; m < n; m++)
    { const cp_eintvl y = v[m];
      r0 += cp_eintvl_before(y,m,x0,lane) ? 1 : 0;
      r1 += cp_eintvl_before(y,m,x1,lane+WAVE) ? 1 : 0;
    }
  wave_sync();
  if (lane < n) v[r0] = x0;
  if (lane+WAVE < n) v[r1] = x1;
  wave_sync();
  if (!dedupe) return n;
  bool k0 = false, k1 = false;
  if (lane < n)
    { x0 = v[lane];
      k0 = (lane == 0);
      if (lane > 0) { const cp_eintvl p = v[lane-1]; k0 = !(p.b == x0.b && p.e == x0.e); }
    }
  if (lane+WAVE < n)
    { x1 = v[lane+WAVE];
      const cp_eintvl p = v[lane+WAVE-1];
      k1 = !(p.b == x1.b && p.e == x1.e);
    }
  const uint64_t m0 = __ballot(k0), m1 = __ballot(k1);
  wave_sync();
  if (k0) v[__popcll(m0 & lt)] = x0;
  if (k1) v[__popcll(m0)+__popcll(m1 & lt)] = x1;
  wave_sync();
  return __popcll(m0)+__popcll(m1);
}

// The same order when the first ns elements are in order already and only a few were appended behind them (the sorts
// after the multi-error search and after the merge, wall.c:875,910): an element of the sorted part moves up by the
// number of appended elements that sort before it; an appended element goes behind the elements of the sorted part
// that are not after it (a binary search; ties keep insertion order, and the sorted part was inserted first) and behind
// the appended elements before it.  n - ns steps per lane instead of n.
__device__ int wave_sort_ev_lds_tail(CP_LDS_PTR(cp_eintvl) v, int n, int ns)
{ const int lane = lane_id();
  cp_eintvl x0 = { 0, 0, 0. }, x1 = { 0, 0, 0. };
  if (lane < n) x0 = v[lane];
  if (lane+WAVE < n) x1 = v[lane+WAVE];
  auto rank_of = [&](const cp_eintvl &x, int idx) -> int
    { int r;
      if (idx < ns)
        { r = idx;
          for (int a = ns; a < n; a++)
            { const cp_eintvl y = v[a];
              r += (y.b != x.b ? y.b < x.b : y.e < x.e) ? 1 : 0;       // strictly before: on a tie the sorted part's element stays in front
            }
        }
      else
        { int lo = 0, hi = ns;                             // elements of the sorted part that are not after x
          while (lo < hi)
            { const int m = (lo+hi) >> 1;
              const cp_eintvl y = v[m];
              if (y.b != x.b ? y.b < x.b : y.e <= x.e) lo = m+1; else hi = m;
            }
          r = lo;
          for (int a = ns; a < n; a++)
            { const cp_eintvl y = v[a];
              r += cp_eintvl_before(y,a,x,idx) ? 1 : 0;
            }
        }
      return r;
    };
  const int r0 = (lane < n) ? rank_of(x0,lane) : 0, r1 = (lane+WAVE < n) ? rank_of(x1,lane+WAVE) : 0;
  wave_sync();
  if (lane < n) v[r0] = x0;
  if (lane+WAVE < n) v[r1] = x1;
  wave_sync();
  return n;
}

// the same on an HBM list: rank sort into tmp, then a copy / an ordered compaction back into v
__device__ int wave_sort_ev_hbm(cp_eintvl *v, int n, cp_eintvl *tmp, bool dedupe)
{ if (n < 2) return n;
  const int lane = lane_id();
  for (int k = lane; k < n; k += WAVE)
    { const cp_eintvl x = v[k];
      int rank = 0;
      for (int m = 0; m < n; m++)
        rank += cp_eintvl_before(v[m],m,x,k) ? 1 : 0;
      tmp[rank] = x;
    }
  wave_sync();
  int out = 0;
  for (int base = 0; base < n; base += WAVE)
    { const int k = base+lane;
      bool keep = false;
      cp_eintvl x = { 0, 0, 0. };
      if (k < n)
        { x = tmp[k];
          keep = !dedupe || (k == 0) || !(tmp[k-1].b == x.b && tmp[k-1].e == x.e);
        }
      const uint64_t m = __ballot(keep);
      if (keep) v[out+__popcll(m & ((1ull << lane)-1))] = x;
      out += __popcll(m);
    }
  wave_sync();
  return out;
}

// ns: the first ns elements are known to be in order (0: nothing is known)
__device__ __forceinline__ int wave_sort_ev(fw_evl &ev, int n, cp_eintvl *tmp, bool dedupe, int ns = 0)
{ if (ev.big) return wave_sort_ev_hbm(ev.g,n,tmp,dedupe);
  if (!dedupe && ns > 0 && n >= 2 && n-ns <= 16)
    return n == ns ? n : wave_sort_ev_lds_tail(ev.l,n,ns);
  return wave_sort_ev_lds(ev.l,n,dedupe);
}

// wall.c:722-731 / 868-872: clear WALL_O at every position strictly inside one of the E-intervals ev[lo..hi).
// OTHERS walls only exist at candidate positions, so the lanes test the candidates instead of sweeping
// the flag array once per interval (an HBM list: the interval ends are staged in sbuf, 2*scap ints, when they fit).
__device__ __forceinline__ void wave_unwall_inside(uint8_t *wall, const int32_t *clist, int n_c, const fw_evl &ev, int lo, int hi,
                                   int *sbuf, int scap)
{ const int lane = lane_id();
  const int n = hi-lo;
  if (n <= 0) return;
  const bool staged = ev.big && n <= scap;
  if (staged)
    { for (int k = lane; k < n; k += WAVE)
        { sbuf[2*k] = ev.g[lo+k].b; sbuf[2*k+1] = ev.g[lo+k].e; }
      wave_sync();
    }
  for (int q = lane; q < n_c; q += WAVE)
    { const int i = clist[q];
      bool in = false;
      if (staged)
        for (int k = 0; k < n && !in; k++)
          in = (sbuf[2*k] < i && i < sbuf[2*k+1]);
      else
        for (int k = lo; k < hi && !in; k++)
          in = (ev.b(k) < i && i < ev.e(k));
      if (in)
        { const uint8_t f = wall[i];
          if (f & CP_W_WALL_O) wall[i] = f & (uint8_t)~CP_W_WALL_O;
        }
    }
  wave_sync();
}

// cp_wall_mult (wall.c:763-860) with the whole wave: the reference walks up to 200 positions away from an
// O-only wall i looking for walls j that would close an error interval, one flag load per step.  Here 64
// positions are loaded at once; a ballot picks the ones the reference would act on (a wall of either
// kind, or the read boundary) and the first OTHERS wall, where the walk ends; the picked positions are
// then handled in order by all lanes together (same values everywhere, stores by lane 0).
template <class RD>
__device__ __forceinline__ void wave_wall_mult(RD *R, int i, int NS, int *midx)
{ const int lane = lane_id();
  uint8_t *wall = R->wall;
  const uint8_t *wall_s = R->wall_s;
  const int plen = R->plen;
  fw_evl &ev = R->eintvl;
  for (int w = CP_DROP; w <= CP_GAIN; w++)
    { const double pe_i = CP_PERR(R,i,CP_SELF,w);
      if (pe_i < CP_PE_THRES_FINAL)
        continue;
      const bool right = (w == CP_DROP);
      const int jend = right ? ((i+CP_MULT_WINDOW < plen+1) ? i+CP_MULT_WINDOW : plen+1)
                             : ((i-CP_MULT_WINDOW > 0) ? i-CP_MULT_WINDOW : 0);
      const int bound = right ? plen : 0;
      bool done = false;
      for (int c0 = 0; c0 < CP_MULT_WINDOW && !done; c0 += WAVE)
        { const int j = right ? i+1+c0+lane : i-1-c0-lane;
          const bool valid = right ? (j < jend) : (j >= jend);
          const int fo = valid ? wall[j] : 0, fs = valid ? wall_s[j] : 0;
          const bool isw = ((fo & CP_W_WALL_O) | (fs & CP_W_WALL_S)) != 0;
          const uint64_t mi = __ballot(valid && (isw || j == bound));
          const uint64_t mw = __ballot(valid && isw);
          const uint64_t ms = __ballot(valid && (fo & CP_W_WALL_O));
          if (__ballot(valid) == 0) break;
          for (uint64_t t = mi; t; t &= t-1)
            { const int b = __ffsll((long long)t)-1;
              const int jj = right ? i+1+c0+b : i-1-c0-b;
              if (jj == bound)                           // wall.c:772-789 / 816-833
                { const double pe = pe_i * pe_i;
                  if (pe < CP_PE_THRES_FINAL)
                    continue;
                  if (*midx >= R->ecap) { R->overflow = 1; return; }
                  ev.wave_grow(*midx);
                  if (lane == 0)
                    { cp_eintvl x; x.b = right ? i : 0; x.e = right ? plen : i; x.pe = pe;
                      ev.put(*midx,x);
                      wall[i] |= CP_W_PAIRED_M;
                    }
                  (*midx)++;
                  if (*midx >= plen) { R->overflow = 8; return; }      // the reference exits here: "# E-intvls >= plen" (wall.c:783-788)
                }
              if (!((mw >> b) & 1))
                continue;
              if (cp_bs_eintvl(ev,0,NS-1,right ? i : jj,right ? jj : i) == -1)
                { const double pe_j = CP_PERR(R,jj,CP_SELF,right ? CP_GAIN : CP_DROP);
                  const double pe = pe_i * pe_j;
                  if (pe >= CP_PE_THRES_FINAL)
                    { if (*midx >= R->ecap) { R->overflow = 1; return; }
                      ev.wave_grow(*midx);
                      if (lane == 0)
                        { cp_eintvl x; x.b = right ? i : jj; x.e = right ? jj : i; x.pe = pe;
                          ev.put(*midx,x);
                          wall[i] |= CP_W_PAIRED_M;
                          wall[jj] |= CP_W_PAIRED_M;
                        }
                      (*midx)++;
                      if (*midx >= plen) { R->overflow = 8; return; }
                    }
                }
              if ((ms >> b) & 1) { done = true; break; }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
//  Candidate flags on chip.  After the walk a flag byte is non-zero only at wall candidates and at the ends of E-intervals,
//  and in all but a few reads those ends are candidates too (the partner of an error's DROP is the GAIN K-1 positions on;
//  the exceptions: a partner at the read's boundary, or one whose count change is below the scan's threshold).  The
//  phases after the walk then never need the flag ARRAYS: the wave loads the candidates' two flag bytes once and works
//  on (position, flags) of the candidates in LDS -- the multi-error search becomes a walk over the neighbouring
//  candidates instead of 64-byte loads of both arrays around every O-only wall, each a round trip to HBM.
//  c_pos[q] = position of candidate q (16 bits), c_fo[q] / c_fs[q] = its byte of the OTHERS / SELF array.
// ---------------------------------------------------------------------------------------------
struct fw_cflags { uint16_t *pos; uint8_t *fo, *fs; int n; };

__device__ __forceinline__ int cf_find(const fw_cflags &F, int x)        // index of the candidate at position x, or -1
{ int lo = 0, hi = F.n-1;
  while (lo <= hi)
    { const int m = (lo+hi) >> 1, v = F.pos[m];
      if (v == x) return m;
      if (v < x) lo = m+1; else hi = m-1;
    }
  return -1;
}

// wall.c:722-731 / 868-872 on the on-chip flags
__device__ __forceinline__ void cf_unwall_inside(const fw_cflags &F, const fw_evl &ev, int lo, int hi)
{ if (hi <= lo) return;
  for (int q = lane_id(); q < F.n; q += WAVE)
    if (F.fo[q] & CP_W_WALL_O)
      { const int i = F.pos[q];
        bool in = false;
        for (int k = lo; k < hi && !in; k++)
          in = (ev.b(k) < i && i < ev.e(k));
        if (in) F.fo[q] &= (uint8_t)~CP_W_WALL_O;
      }
  wave_sync();
}

// cp_wall_mult (wall.c:763-860) for the O-only wall at candidate q, all lanes together: the walls the reference meets
// within 200 positions are the flagged candidates after (DROP) / before (GAIN) q, 64 of them per step, in order; the read's
// boundary (plen / 0), when it lies within reach, comes last.  Same values on every lane, stores by lane 0.
// xs[0..nx): the SELF walls that are no candidates (partners below the scan's threshold, in increasing order; usually
// none): the reference meets them on its way like any other wall; they are never OTHERS walls, so they end no search.
template <class RD>
__device__ __forceinline__ void cf_wall_mult(RD *R, const fw_cflags &F, int q, int NS, int *midx, const uint16_t *xs, int nx)
{ const int lane = lane_id();
  const int plen = R->plen, i = F.pos[q];
  fw_evl &ev = R->eintvl;
  for (int w = CP_DROP; w <= CP_GAIN; w++)
    { const double pe_i = CP_PERR(R,i,CP_SELF,w);
      if (pe_i < CP_PE_THRES_FINAL)
        continue;
      const bool right = (w == CP_DROP);
      const int jend = right ? ((i+CP_MULT_WINDOW < plen+1) ? i+CP_MULT_WINDOW : plen+1)
                             : ((i-CP_MULT_WINDOW > 0) ? i-CP_MULT_WINDOW : 0);
      bool done = false;
      int xi = right ? 0 : nx-1;                           // the next off-list wall in the direction of the search
      if (right) { while (xi < nx && (int)xs[xi] <= i) xi++; } else { while (xi >= 0 && (int)xs[xi] >= i) xi--; }
      // an off-list wall at jj (wall.c:793-806 / 837-850 for a wall that is not an OTHERS wall); false: a capacity ran out
      auto visit_x = [&](int jj) -> bool
        { if (cp_bs_eintvl(ev,0,NS-1,right ? i : jj,right ? jj : i) == -1)
            { const double pe_j = CP_PERR(R,jj,CP_SELF,right ? CP_GAIN : CP_DROP);
              const double pe = pe_i * pe_j;
              if (pe >= CP_PE_THRES_FINAL)
                { if (*midx >= R->ecap) { R->overflow = 1; return false; }
                  ev.wave_grow(*midx);
                  if (lane == 0)
                    { cp_eintvl x; x.b = right ? i : jj; x.e = right ? jj : i; x.pe = pe;
                      ev.put(*midx,x);
                      F.fo[q] |= CP_W_PAIRED_M;            // (the flag of jj itself is read by nobody: jj is the origin of no search)
                    }
                  (*midx)++;
                  if (*midx >= plen) { R->overflow = 8; return false; }
                }
            }
          return true;
        };
      // the off-list walls before position `lim` in the direction of the search (and inside the window)
      auto flush_x = [&](int lim) -> bool
        { while (right ? (xi < nx && (int)xs[xi] < lim && (int)xs[xi] < jend) : (xi >= 0 && (int)xs[xi] > lim && (int)xs[xi] >= jend))
            { if (!visit_x((int)xs[xi])) return false;
              xi += right ? 1 : -1;
            }
          return true;
        };
      for (int c0 = 0; !done; c0 += WAVE)
        { const int qq = right ? q+1+c0+lane : q-1-c0-lane;
          int pj = 0, fo = 0, fs = 0;
          bool valid = qq >= 0 && qq < F.n;
          if (valid) { pj = F.pos[qq]; valid = right ? (pj < jend) : (pj >= jend); }
          if (valid) { fo = F.fo[qq]; fs = F.fs[qq]; }
          const uint64_t mv = __ballot(valid);
          const uint64_t mw = __ballot(valid && (((fo & CP_W_WALL_O) | (fs & CP_W_WALL_S)) != 0));
          const uint64_t ms = __ballot(valid && (fo & CP_W_WALL_O));
          for (uint64_t t = mw; t; t &= t-1)
            { const int b = __ffsll((long long)t)-1;
              const int jj = __shfl(pj,b), qj = right ? q+1+c0+b : q-1-c0-b;
              if (nx > 0 && !flush_x(jj)) return;
              if (cp_bs_eintvl(ev,0,NS-1,right ? i : jj,right ? jj : i) == -1)
                { const double pe_j = CP_PERR(R,jj,CP_SELF,right ? CP_GAIN : CP_DROP);
                  const double pe = pe_i * pe_j;
                  if (pe >= CP_PE_THRES_FINAL)
                    { if (*midx >= R->ecap) { R->overflow = 1; return; }
                      ev.wave_grow(*midx);
                      if (lane == 0)
                        { cp_eintvl x; x.b = right ? i : jj; x.e = right ? jj : i; x.pe = pe;
                          ev.put(*midx,x);
                          F.fo[q] |= CP_W_PAIRED_M;
                          F.fo[qj] |= CP_W_PAIRED_M;
                        }
                      (*midx)++;
                      if (*midx >= plen) { R->overflow = 8; return; }
                    }
                }
              if ((ms >> b) & 1) { done = true; break; }
            }
          if (mv != ~0ull) break;                          // the candidates within reach end inside these 64
        }
      if (nx > 0 && !done && !flush_x(right ? 0x7fffffff : -1)) return;      // off-list walls beyond the last candidate wall
      if (!done && (right ? plen < jend : jend == 0))     // the boundary is within reach: wall.c:772-789 / 816-833
        { const double pe = pe_i * pe_i;
          if (pe >= CP_PE_THRES_FINAL)
            { if (*midx >= R->ecap) { R->overflow = 1; return; }
              ev.wave_grow(*midx);
              if (lane == 0)
                { cp_eintvl x; x.b = right ? i : 0; x.e = right ? plen : i; x.pe = pe;
                  ev.put(*midx,x);
                  F.fo[q] |= CP_W_PAIRED_M;
                }
              (*midx)++;
              if (*midx >= plen) { R->overflow = 8; return; }      // the reference exits here: "# E-intvls >= plen" (wall.c:783-788)
            }
        }
      wave_sync();                                         // lane 0's flag updates before the other direction reads them
    }
}

// Whole-path calls: the end and the final class of an interval as one word, written by the classify_unrel kernels and read
// by k_paint_labels / k_label_runs instead of the 48-byte records (an end is below 2^28; a class is 0..3, 7 = none).
#define CP_PCLS(e,a)   (((uint32_t)(e) << 3) | ((uint32_t)(a) & 7u))
#define CP_PCLS_END(x) ((int)((x) >> 3))
#define CP_PCLS_CLS(x) (((x) & 7u) == 7u ? -1 : (int)((x) & 7u))

// What classify_rel needs of a reliable interval, 24 bytes instead of a 48-byte copy of the record plus a 4-byte index:
// written by k_find_wall on whole-path calls (do_rel == 2) and read by the main size class of k_classify_rel_grp; the
// stage API and the rare size classes keep the `rintvl` copies (round 5: 0.9 GB less HBM traffic per 2 Gbases).
struct cp_rrec { int32_t b, e; uint16_t ccb, cce; int32_t idx; double pe; };
static_assert(sizeof(cp_rrec) == 24,"compact reliable-interval record");
// ... and the interval records themselves as three arrays (whole-path calls): what every classify_unrel update reads
// (16 bytes, loaded coalesced), the three log-probabilities that only the updates of non-fixed intervals come back for (24
// bytes), and the two bytes classify_rel and classify_unrel exchange -- instead of 48-byte records of which a wave's loads
// use a third of every line they fetch.  The 48-byte records are still written for the reads of the rare classify_unrel
// classes (N > UNREL_SMALL_MAXN, or beyond GRP_MAX_PLEN k-mers), whose kernels work on them, and cp_get_intervals puts them
// together again on read-back.
struct cp_ivA { int32_t b, e; uint16_t cb, ce, ccb, cce; };
struct cp_ivB { double pe, peo_b, peo_e; };
struct cp_ivC { uint8_t is_rel; int8_t asgn; };
static_assert(sizeof(cp_ivA) == 16 && sizeof(cp_ivB) == 24 && sizeof(cp_ivC) == 2,"interval records as arrays");
struct cp_soa { cp_ivA *a; cp_ivB *b; cp_ivC *c; };       // all NULL: the 48-byte records only
__device__ __forceinline__ void cp_soa_put(const cp_soa &S, int64_t at, const cp_intvl &I)
{ cp_ivA a; a.b = I.b; a.e = I.e; a.cb = I.cb; a.ce = I.ce; a.ccb = I.ccb; a.cce = I.cce;
  cp_ivB b; b.pe = I.pe; b.peo_b = I.peo_b; b.peo_e = I.peo_e;
  cp_ivC c; c.is_rel = I.is_rel; c.asgn = I.asgn;
  S.a[at] = a; S.b[at] = b; S.c[at] = c;
}
__device__ __forceinline__ cp_intvl cp_soa_get(const cp_soa &S, int64_t at)
{ const cp_ivA a = S.a[at]; const cp_ivB b = S.b[at]; const cp_ivC c = S.c[at];
  cp_intvl I;
  memset(&I,0,sizeof(I));
  I.b = a.b; I.e = a.e; I.cb = a.cb; I.ce = a.ce; I.ccb = a.ccb; I.cce = a.cce;
  I.is_rel = c.is_rel; I.asgn = c.asgn; I.pe = b.pe; I.peo_b = b.peo_b; I.peo_e = b.peo_e;
  return I;
}

// ---------------------------------------------------------------------------------------------
//  k_find_wall: wall.c:570-958, one wave per read.
// ---------------------------------------------------------------------------------------------
#ifdef CP_PROF_WALK
__device__ unsigned long long g_phase_max[12], g_phase_sum[12], g_phase_arg[12];
__device__ unsigned long long g_live_prof[8];
__device__ unsigned long long g_emit_prof[8];           // wave time inside the emission loop: [0] boundaries, [1] make_interval, [2] find_rel, [3] record stores, [4] compaction
#ifdef CP_PROF_INNER
#define EM_LT0() unsigned long long em_t = wall_clock64()
#define EM_LT(k) do { unsigned long long t_ = wall_clock64(); if (lane == 0) atomicAdd(&g_emit_prof[k],t_-em_t); em_t = wall_clock64(); } while (0)
#endif
#define PH_STAMP(k) do { if (lane == 0) { unsigned long long t_ = wall_clock64(); unsigned long long d_ = t_-ph_t; ph_t = t_; \
      atomicAdd(&g_phase_sum[k],d_); unsigned long long o_ = atomicMax(&g_phase_max[k],d_); if (d_ > o_) g_phase_arg[k] = ((unsigned long long)r << 32) | (unsigned)n_c; } } while (0)
#elif defined(CP_STOP_AT)
// diagnostic builds: the kernel ends after phase CP_STOP_AT (instruction counts per phase from counter passes of the
// truncated kernels, scripts/phase_insts.sh); k_wall_tasks then reports no candidates
#define PH_STAMP(k) do { if ((k) == CP_STOP_AT) { PH_STOP_EXTRA; return; } } while (0)
#else
#define PH_STAMP(k) ((void)0)
#endif
#ifndef EM_LT0
#define EM_LT0() ((void)0)
#define EM_LT(k) ((void)0)
#endif
#define PH_STOP_EXTRA

// The walk of wall.c:590-707 is two kernels.  Every load of the reference's walk is a cold miss here (the profile,
// bases and flag arrays of a sub-batch do not stay in L2), a candidate's loads depend on each other, and its binomial-tail
// evaluations are long serial chains, so the walk is split (cp_wall.h) into
//   0.  the wave lists the read's candidate positions from the bitmap (64 words per step);
//   1a. 64 candidates at a time, lane k evaluates candidate k's shared prelude and the threshold filters of both error
//       types (cp_wall_candidate_pre / _filter) and the (candidate, error type) pairs that stay alive -- ~40 % for
//       SELF, ~3 % for OTHERS -- are appended to a task list;
//   1b. 64 tasks at a time, lane k evaluates everything else about its pair that is a function of the read alone
//       (cp_wall_candidate_live: own P(error), low-complexity partner, best high-complexity partner).  Dense lanes
//       matter: this is where the instructions are -- and the registers: 96 VGPRs and 192 bytes of spills with this
//       part inside, 70 and none without it;
//   2.  the tasks are replayed (paired flags, perror memo, flag and interval-list updates): a lane per task in rounds of
//       tasks with no open dependency on an earlier one, on flag bytes kept in LDS (k_find_wall below); a read whose
//       memo does not fit on chip takes the one-lane form (cp_wall_candidate_replay: the SELF pass on lane 0 and the
//       OTHERS pass on lane 1 -- disjoint state: own flag array, own memo table, own interval list).
// k_wall_tasks does 0-1b (4 waves per SIMD) and leaves the lists and the task results in HBM; k_find_wall does 2 and
// the list phases after the walk at 5.5 waves per SIMD: they are chains of dependent loads and want waves, not registers.
// Scratch: the four int lists of `wl` hold per candidate [0] maxt,maxl and both filter results, [1] the position,
// [2] the count pair, and [3] the task list; `tres` the task results; `fwc` per read n_c, n_t, SELF tasks, overflow.
struct task_res { double own_pe, lc_v, hc_pe; int lc_j, hc;  };   // hc: lc_kind | (hc_j >= 0) << 2 | (hc_j - i) << 16

// The sequence contexts of the walk (cp_ctx.h) look at the bases next to a position.  Phase 1a needs the three contexts
// of the candidate itself: eight bases before i+K-2 (a DROP) or on either side of i (a GAIN) -- ONE 16-byte load into
// registers (cp_seq_rwin below).  Phase 1b asks for the context of the low-complexity partners, a few unit lengths to the
// right (DROP) or left (GAIN) of the candidate, in a loop: a lane copies the 32 bases that cover the first five or more
// partners into its row of an LDS block (two independent 16-byte loads) and the loop reads the row; a position outside
// the row is read from HBM.  (Until the contexts came from words, round 5, both phases filled a 96-base row: six loads
// and 24 LDS stores per lane and phase.)
#define FW_SEQ_WIN    32
#define FW_SEQ_STRIDE 36             // bytes per lane: 9 dwords, an odd number of banks apart
struct cp_seq_lwin
  { CP_SEQ_T g; CP_LDS_PTR(const char) w; int lo, len;
    __device__ __forceinline__ char operator[](int p) const
    { const unsigned d = (unsigned)(p-lo); return d < (unsigned)len ? w[d] : g[p]; }
  };
// eight bases of an LDS window as one word (cp_ctx.h: the contexts without loops): three aligned dwords and two funnel
// shifts (the rows are four-byte aligned, the position is not); a position whose eight bases are not all in the window
// is left to the loops, which read past the window from the read itself
__device__ __forceinline__ int cp_seq_dirword(const cp_seq_lwin &sq, int rlen, int pos, int dir, uint64_t *D)
{ (void)rlen;
  const int d = (dir > 0 ? pos : pos-7)-sq.lo;
  if (d < 0 || d+8 > sq.len) return 0;
  CP_LDS_PTR(const uint32_t) rw = (CP_LDS_PTR(const uint32_t))(sq.w+(d & ~3));
  const uint32_t x0 = rw[0], x1 = rw[1], x2 = rw[2];
  const uint32_t sh = (uint32_t)(d & 3)*8;
  const uint32_t lo = __builtin_amdgcn_alignbit(x1,x0,sh), hi = __builtin_amdgcn_alignbit(x2,x1,sh);
  const uint64_t x = (uint64_t)lo | ((uint64_t)hi << 32);
  *D = dir > 0 ? x : __builtin_bswap64(x);
  return 8;
}
struct __attribute__((packed, aligned(1))) cp_u8x16 { uint32_t v[4]; };
struct __attribute__((packed, aligned(1))) cp_u8x4 { uint32_t v; };
// Sixteen bases around a position IN REGISTERS (round 5): the context scans of correct_wall_cnt inside k_find_wall's
// emission loop (cp_rel_interval) read a handful of bases right of b+K-1 and left of e-1; from the global pointer every
// one of them was a load of its own in a loop that waits for it (k_find_wall has no LDS left for windows like k_find_rel's).
// One 16-byte load per side instead; a base outside the sixteen is read from the read as before.  dir = +1: the
// window starts seven bases before the position, -1: it ends four bases after it.
struct cp_seq_rwin
  { CP_SEQ_T g;
    uint64_t wl, wh;                                       // bases lo .. lo+7, lo+8 .. lo+15
    int lo, len;
    __device__ __forceinline__ char operator[](int p) const
    { const unsigned d = (unsigned)(p-lo);
      if (d < (unsigned)len)                               // (one select and a 64-bit shift: a chain of four 32-bit selects the
        { const uint64_t x = (d < 8) ? wl : wh;            //  compiler turns into an indexed load from a copy in scratch)
          return (char)(x >> ((d & 7)*8));
        }
      return g[p];
    }
  };
__device__ __forceinline__ int cp_seq_dirword(const cp_seq_rwin &sq, int rlen, int pos, int dir, uint64_t *D)
{ (void)rlen;
  const int d = (dir > 0 ? pos : pos-7)-sq.lo;
  if (sq.len != 16 || d < 0 || d > 8) return 0;
  const uint64_t x = d == 0 ? sq.wl : d == 8 ? sq.wh : (sq.wl >> (8*d)) | (sq.wh << (64-8*d));
  *D = dir > 0 ? x : __builtin_bswap64(x);
  return 8;
}
struct cp_seq_rsrc { CP_SEQ_T g; };                        // "make me a register window": what k_find_wall hands to cp_rel_interval
__device__ __forceinline__ cp_seq_rwin cp_seq_window(const cp_seq_rsrc &src, int pos, int rlen, int dir)
{ cp_seq_rwin sq;
  int lo = dir > 0 ? pos-7 : pos-11;                       // (the eight bases on either side of a position scanned to the right, cp_rctx3)
  if (lo > rlen-16) lo = rlen-16;
  if (lo < 0) lo = 0;
  sq.g = src.g; sq.lo = lo; sq.len = 0; sq.wl = sq.wh = 0;
  if (rlen-lo >= 16)
    { const cp_u8x16 x = *reinterpret_cast<const cp_u8x16 *>(CP_SPAN(src.g,lo,16));
      sq.wl = (uint64_t)x.v[0] | ((uint64_t)x.v[1] << 32);
      sq.wh = (uint64_t)x.v[2] | ((uint64_t)x.v[3] << 32);
      sq.len = 16;
    }
  return sq;
}

#define FR_STRIDE 68                 // k_find_rel: two 32-base windows per lane, 17 dwords apart
__device__ __forceinline__ void fr_seq_win_load(cp_seq_lwin &sq, char *row, int lo, int rlen)
{ if (lo > rlen-32) lo = rlen-32;
  if (lo < 0) lo = 0;
  const int len = rlen-lo < 32 ? rlen-lo : 32;
  uint32_t *rw = reinterpret_cast<uint32_t *>(row);
  if (len == 32)
    {
#pragma unroll
      for (int k = 0; k < 2; k++)
        { const cp_u8x16 x = *reinterpret_cast<const cp_u8x16 *>(CP_SPAN(sq.g,lo+16*k,16));
          rw[4*k] = x.v[0]; rw[4*k+1] = x.v[1]; rw[4*k+2] = x.v[2]; rw[4*k+3] = x.v[3];
        }
    }
  else
    for (int k = 0; k < len; k++) row[k] = sq.g[lo+k];
  sq.lo = lo; sq.len = len;
}

#ifndef FW_WAVES_PER_EU
#define FW_WAVES_PER_EU 4
#endif
#undef PH_STOP_EXTRA
#define PH_STOP_EXTRA do { if (lane == 0) { fwc[4*(int64_t)r] = 0; fwc[4*(int64_t)r+1] = 0; fwc[4*(int64_t)r+2] = 0; fwc[4*(int64_t)r+3] = 0; } } while (0)
__global__ void __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(FW_WAVES_PER_EU)))
k_wall_tasks(const cp_dev_params *__restrict__ P, const char *__restrict__ seq, const int64_t *__restrict__ seq_off,
             const uint16_t *__restrict__ prof, const int64_t *__restrict__ prof_off, int nreads,
             const uint64_t *__restrict__ bm, uint8_t *__restrict__ wall_all, const int64_t *__restrict__ ioff,
             const int32_t *__restrict__ perm, int32_t *__restrict__ wlist, task_res *__restrict__ tres_all,
             int32_t *__restrict__ fwc)
{ if ((int)blockIdx.x >= nreads) return;
  const int r = perm[blockIdx.x];
  const int lane = lane_id();
  const int64_t po = prof_off[r];
  const int plen = (int)(prof_off[r+1]-po);
  const int rlen = (int)(seq_off[r+1]-seq_off[r]);
  __shared__ __attribute__((aligned(4))) char s_win[WAVE*FW_SEQ_STRIDE];
  cp_read_t<cp_perr_hybrid,cp_seq_lwin> R;
  R.P = P; R.prof = CP_PROF_VIEW(prof+po,plen); R.lf = P->logfact; R.plen = plen; R.rlen = rlen;
  R.seq.g = CP_SEQ_VIEW(seq+seq_off[r],rlen); R.seq.w = (CP_LDS_PTR(const char))(s_win+lane*FW_SEQ_STRIDE); R.seq.lo = 0; R.seq.len = 0;
  R.wall = wall_all+po+r; R.wall_s = R.wall;
  R.eintvl = R.ointvl = nullptr; R.ecap = 0; R.eidx = R.oidx = 0; R.overflow = 0;
  cp_read_t<cp_perr_hybrid,cp_seq_rwin> R1;             // phase 1a's view of the read: the bases as a register window
  R1.P = P; R1.prof = R.prof; R1.lf = P->logfact; R1.plen = plen; R1.rlen = rlen;
  R1.wall = R.wall; R1.wall_s = R.wall; R1.eintvl = R1.ointvl = nullptr; R1.ecap = 0; R1.eidx = R1.oidx = 0; R1.overflow = 0;
  const int icap = (int)(ioff[r+1]-ioff[r]);
  int32_t *wl = wlist+ioff[r]*4;                        // four int32 lists of capacity icap
  task_res *tres = tres_all+ioff[r];
  uint8_t *wall = R.wall;
  int32_t *cinfo = wl, *clist = wl+icap, *ccnt = wl+2*(int64_t)icap, *tlist = wl+3*(int64_t)icap;
  int n_c = 0;
#ifdef CP_PROF_WALK
  unsigned long long ph_t = wall_clock64();
#endif
  if (plen > 1)
    { const int64_t lo = po+1, hi = po+plen;
      const int64_t w_lo = lo >> 6, w_hi = (hi-1) >> 6;
      for (int64_t wb = w_lo; wb <= w_hi; wb += WAVE)
        { const int64_t w = wb+lane;
          const uint64_t bits = (w <= w_hi) ? bitmap_word(bm,w,lo,hi) : 0ull;
          int c = __popcll(bits), off = c;
          for (int o = 1; o < WAVE; o <<= 1)
            { int x = __shfl_up(off,o); if (lane >= o) off += x; }
          const int tot = __shfl(off,WAVE-1);
          off = n_c+off-c;
          for (uint64_t t = bits; t; t &= t-1)
            { const int k = __ffsll((long long)t)-1;
              if (off < icap) clist[off] = (int)((w << 6)+k-po);
              off++;
            }
          n_c += tot;
        }
    }
  if (n_c > icap) n_c = icap;                           // cannot happen: icap = 2*ncand+4
  wave_sync();
  PH_STAMP(0);
  int n_t = 0, n_live0 = 0;                             // tasks so far, SELF tasks among them
  for (int base = 0; base < n_c; base += WAVE)          // 1a
    { int f0 = 0, f1 = 0;
      if (base+lane < n_c)
        { const int i = clist[base+lane];
          const bool drop = R.prof[i-1] > R.prof[i];       // (cp_wall_candidate_pre's rule: the contexts of a DROP end at i+K-2, those of a GAIN begin at i)
          cp_seq_rsrc rs; rs.g = R.seq.g;
          R1.seq = cp_seq_window(rs,drop ? i+P->K-2 : i,rlen,drop ? -1 : +1);
          cp_wall_pre pre;
          cp_wall_candidate_pre(&R1,i,&pre);
          f0 = cp_wall_candidate_filter(P,CP_SELF,pre);
          f1 = cp_wall_candidate_filter(P,CP_OTHERS,pre);
          cinfo[base+lane] = pre.maxt*32+pre.maxl+(f0 << 8)+(f1 << 12);
          ccnt[base+lane] = (int)R.prof[i-1] | ((int)R.prof[i] << 16);
          // (a candidate that is an OTHERS wall whatever the replay finds -- CP_CF_WALLNOW in f1 -- gets its flag from
          //  k_find_wall, which knows whether the read's flags live on chip or in the flag array)
        }
      const int l0 = (f0 & CP_CF_LIVE) ? 1 : 0, l1 = (f1 & CP_CF_LIVE) ? 1 : 0;
      int c = l0+l1, off = c;
      for (int o = 1; o < WAVE; o <<= 1)
        { int x = __shfl_up(off,o); if (lane >= o) off += x; }
      const int tot = __shfl(off,WAVE-1);
      off = n_t+off-c;
      if (l0) { if (off < icap) tlist[off] = (base+lane)*2;   off++; }
      if (l1) { if (off < icap) tlist[off] = (base+lane)*2+1; }
      n_t += tot;
      n_live0 += __popcll(__ballot(l0));
    }
  int ovf = 0;
  if (n_t > icap) { ovf = 1; n_t = 0; n_c = 0; }        // cannot happen: 2*n_c <= icap
  wave_sync();
  PH_STAMP(7);
  for (int tb = 0; tb < n_t; tb += WAVE)                // 1b: lane k evaluates task k
    { const int nb = (n_t-tb < WAVE) ? n_t-tb : WAVE;
      if (lane < nb)
        { const int code = tlist[tb+lane], k = code >> 1, e = code & 1;
          const int i = clist[k], tl = cinfo[k] & 255, cc = ccnt[k];
          cp_wall_pre pre;
          const int cim1 = cc & 0xffff, ci = (cc >> 16) & 0xffff;
          // the partners of a DROP lie to the right and their contexts end K-2 bases further on, those of a GAIN to the left
          fr_seq_win_load(R.seq,s_win+lane*FW_SEQ_STRIDE,cim1 > ci ? i+P->K-16 : i-24,rlen);
          pre.cng = cim1 > ci ? cim1-ci : ci-cim1;
          if (cim1 > ci) { pre.wtype = CP_DROP; pre.cin = ci;   pre.cout = cim1; }
          else           { pre.wtype = CP_GAIN; pre.cin = cim1; pre.cout = ci;   }
          pre.maxt = tl >> 5; pre.maxl = tl & 31;
          pre.maxpe = P->pe[pre.maxt][pre.maxl];
          pre.lpe   = P->lpe[pre.maxt][pre.maxl];
          pre.l1mpe = P->l1mpe[pre.maxt][pre.maxl];
          cp_cand_pure c;
          cp_wall_candidate_live(&R,i,e,pre,&c);
          task_res q;
          q.own_pe = c.own_pe; q.lc_v = c.lc_v; q.hc_pe = c.hc_pe;
          q.lc_j = c.lc_j;
          q.hc = c.lc_kind | (c.hc_j >= 0 ? 4 : 0) | ((c.hc_j >= 0 ? c.hc_j-i : 0) << 16);
          tres[tb+lane] = q;
        }
    }
  PH_STAMP(6);
  if (lane == 0) { fwc[4*(int64_t)r] = n_c; fwc[4*(int64_t)r+1] = n_t; fwc[4*(int64_t)r+2] = n_live0; fwc[4*(int64_t)r+3] = ovf; }
}

#undef PH_STOP_EXTRA
#define PH_STOP_EXTRA
#ifndef FW2_WAVES_PER_EU
#define FW2_WAVES_PER_EU 5        // (LDS, 7.4 KB per wave with the on-chip interval list, allows 5.4; at 6 the find_rel part spills a VGPR to scratch)
#endif
__global__ void __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(FW2_WAVES_PER_EU)))
k_find_wall(const cp_dev_params *__restrict__ P, const char *__restrict__ seq, const int64_t *__restrict__ seq_off,
            const uint16_t *__restrict__ prof, const int64_t *__restrict__ prof_off, int nreads,
            uint8_t *__restrict__ wall_all, uint8_t *__restrict__ walls_all,
            int32_t *__restrict__ hkeys, double *__restrict__ hvals, const int64_t *__restrict__ hoff,
            cp_eintvl *__restrict__ eintvl_all, cp_eintvl *__restrict__ ointvl_all, const int64_t *__restrict__ eoff,
            cp_intvl *__restrict__ intvl_all, const int64_t *__restrict__ ioff,
            int32_t *__restrict__ nintvl, int32_t *__restrict__ err, const int32_t *__restrict__ perm,
            int32_t *__restrict__ wlist, const task_res *__restrict__ tres_all, const int32_t *__restrict__ fwc,
            cp_intvl *__restrict__ rintvl_all, int32_t *__restrict__ relmap_all, int32_t *__restrict__ nrel, int do_rel,
            cp_rrec *__restrict__ rrec_all, cp_soa soa)
{ if ((int)blockIdx.x >= nreads) return;
  const int r = perm[blockIdx.x];
  const int lane = lane_id();
  const int64_t po = prof_off[r];
  const int plen = (int)(prof_off[r+1]-po);
  const int rlen = (int)(seq_off[r+1]-seq_off[r]);

#ifndef FW_XCAP
#define FW_XCAP 8                     // off-list SELF walls kept on chip per read (a diagnostic build with 1 drives the full-list path: scripts/r5_xcap.sh)
#endif
#ifndef FW_LCAP1
#define FW_LCAP1 64
#endif
  if (plen <= 0)                                         // a read shorter than K has no k-mer: no interval (ClassPro.c:209-226 prints
    { if (lane == 0) { nintvl[r] = 0; if (do_rel) nrel[r] = 0; }      // its N's and goes on).  Found by the -DCP_BOUNDS build: the
      return;                                            // emission below made one interval [0,0) of it and read prof[0], prof[-1]
    }                                                    // -- another read's counts, or the cells around the batch's buffer.
  constexpr int LCAP0 = 256, LCAP1 = FW_LCAP1;           // on-chip memo slots of the SELF / OTHERS pass
  __shared__ int32_t s_mkey[LCAP0+LCAP1];
  __shared__ double  s_mval[LCAP0+LCAP1];
  __shared__ uint16_t s_x[FW_XCAP];                      // SELF walls off the candidates (see the replay), sorted after it
  __shared__ int s_nx;
  __shared__ cp_eintvl s_ev[FW_EVL];                     // the E-interval list while it is short (fw_evl)
  cp_read_t<cp_perr_hybrid,CP_SEQ_T,CP_PROF_T,const double *,fw_evl> R;
  R.P = P; R.prof = CP_PROF_VIEW(prof+po,plen); R.seq = CP_SEQ_VIEW(seq+seq_off[r],rlen); R.lf = P->logfact; R.plen = plen; R.rlen = rlen;
  R.wall = wall_all+po+r;
  R.wall_s = walls_all+po+r;
  { const int64_t ho = hoff[r], hc = (hoff[r+1]-ho) >> 1;     // two tables, one per error type
    R.perror.g.keys = hkeys+ho;
    R.perror.g.vals = hvals+ho*4;
    R.perror.g.mask = (uint32_t)hc-1;
    R.perror.lkeys = (CP_LDS_PTR(int32_t))s_mkey;
    R.perror.lvals = (CP_LDS_PTR(double))s_mval;
    R.perror.lcap0 = LCAP0; R.perror.lcap1 = LCAP1;
    R.perror.use_lds = 0;
  }
  if (lane == 0) s_nx = 0;
  for (int k = lane; k < LCAP0+LCAP1; k += WAVE)
    s_mkey[k] = -1;
  R.eintvl.g = eintvl_all+eoff[r];
  R.eintvl.l = (CP_LDS_PTR(cp_eintvl))s_ev;
  R.ointvl.g = ointvl_all+eoff[r];                      // the O-pairs (a handful; only their ends are read back) stay in HBM
  R.ecap = (int)(eoff[r+1]-eoff[r]);
  R.eidx = R.oidx = 0; R.overflow = 0;
  cp_intvl *intvl = intvl_all+ioff[r];
  const int icap = (int)(ioff[r+1]-ioff[r]);
  int32_t *wl = wlist+ioff[r]*4;                        // four int32 lists of capacity icap
  const task_res *tres = tres_all+ioff[r];
  uint8_t *wall = R.wall;
  const uint8_t *wall_s = R.wall_s;
  __shared__ task_res s_res[WAVE];
  int32_t *clist = wl+icap, *ccnt = wl+2*(int64_t)icap, *tlist = wl+3*(int64_t)icap;
  const int n_c = fwc[4*(int64_t)r], n_t = fwc[4*(int64_t)r+1], n_live0 = fwc[4*(int64_t)r+2];
  if (fwc[4*(int64_t)r+3]) R.overflow = 1;
#ifdef CP_PROF_WALK
  unsigned long long ph_t = wall_clock64();
#endif
  // a pass memoises at most two entries per live candidate (its own and its low-complexity partner's): the table never
  // fills while twice the tasks leave eight slots free.  (Until the end of round 5 the rule was a load of 3/4 -- 96 SELF
  // tasks -- and 9 % of the bench's reads, the ones with 97-120, took the one-lane replay on the tables in HBM, several
  // times the time of the lane-parallel one: a long probe sequence in LDS costs less than that.)
  R.perror.use_lds = ((2*n_live0 <= LCAP0-8) ? 1 : 0) | ((2*(n_t-n_live0) <= LCAP1-8) ? 2 : 0);
  // the SELF pass appends at most one E-interval per live task: with that many slots on chip the replay never moves the list
  R.eintvl.big = (n_live0 > FW_EVL || R.ecap < FW_EVL) ? 1 : 0;
  // a pass whose memo does not fit on chip uses its table in HBM: empty it here (the few reads that need one), instead
  // of a memset over every read's tables per batch
  { const int hc = (int)R.perror.g.mask+1;
    for (int e = 0; e < 2; e++)
      if (!((R.perror.use_lds >> e) & 1))
        for (int k = lane; k < hc; k += WAVE) R.perror.g.keys[(size_t)e*hc+k] = -1;
  }
  wave_sync();
  // 2. The replay (wall.c:639-690 with the values of step 1), 64 tasks at a time, A LANE PER TASK.  What one task reads
  //    of another's: the "paired" flag of its own position, set by an earlier task of the same pass that took the position
  //    as its partner, and the perror memo of its position and of its low-complexity partner's (the first request's value
  //    stands).  So a task only has to wait for the EARLIER tasks of its pass that touch one of its three positions (its
  //    own, the low-complexity partner, the best high-complexity partner).  Per chunk every lane enters its bit in an LDS
  //    table of 64-bit masks indexed by a hash of (position, pass), ORs the masks of its positions -- its dependencies, a
  //    superset when two positions share a slot -- and the chunk is done in rounds: the tasks none of whose dependencies
  //    is still open run together.  An error's DROP and GAIN make chains of two, so a chunk takes two or three rounds
  //    where the one-lane replay took 64 steps.  E-/O-intervals are appended after the rounds, in task order (ballots).
  //    The memo tables take concurrent inserts of DIFFERENT keys (compare-and-swap on the key slot); tasks of one round
  //    never share a position.  A read whose memo does not fit on chip takes the one-lane replay below.
  // The candidates' flag bytes ON CHIP from the start (fw_cflags: positions, OTHERS flags, SELF flags in the front of the
  // staging block): every cell the walk writes is a candidate's or a partner's, and a partner is nearly always a candidate
  // itself.  The read's flags then never exist in HBM: no cold read-modify-write per flag of the replay, none of the
  // clean-up stores that keep the flag arrays zero between batches (together 2/3 of this kernel's write traffic), and
  // k_wall_tasks writes no flag either.  A read whose walk pairs a position that is no candidate (a boundary, a partner
  // below the scan's threshold) starts over with its flags in the arrays.
  int *sbuf = reinterpret_cast<int *>(s_res);
  constexpr int SBUF = (int)(sizeof(task_res)*WAVE/sizeof(int));      // 512 ints
  constexpr int SCOMP = 64;                             // components of the error regions kept on chip
  constexpr int U16 = 2*(SBUF-2*SCOMP);                 // 16-bit list slots in front of the component arrays
  uint16_t *s16 = reinterpret_cast<uint16_t *>(sbuf);
  fw_cflags F;
  F.pos = s16; F.fo = reinterpret_cast<uint8_t *>(s16+n_c); F.fs = F.fo+n_c; F.n = n_c;
  constexpr int NDEP = 128;                             // slots of the dependency table (behind the flags: the block's second half)
  static_assert(sizeof(task_res)*WAVE >= 4*256+NDEP*sizeof(unsigned long long),"flags of 255 candidates and the dependency masks fit the staging block");
  bool onchip = R.perror.use_lds == 3 && plen <= 65535 && n_c <= 255 && 3*n_c+2*SCOMP+1 <= U16 && !R.eintvl.big;
  const int32_t *cinfo = wl;
  auto flags_to_hbm = [&]()                             // the flags k_wall_tasks knew: OTHERS walls whatever the replay finds (cp_read_t::spec_wallnow)
    { for (int q = lane; q < n_c; q += WAVE)
        if ((cinfo[q] >> 12) & CP_CF_WALLNOW) wall[clist[q]] = CP_W_WALL_O;
      wave_sync();
    };
  if (R.perror.use_lds == 3)
    { unsigned long long *s_dep = reinterpret_cast<unsigned long long *>(s_res)+(sizeof(task_res)*WAVE/sizeof(unsigned long long)-NDEP);
      const uint64_t ltm = (1ull << lane)-1;
      const int ecap = R.ecap;
      const bool ebig = R.eintvl.big != 0;
      int eidx = 0, oidx = 0, ovf = 0;
      auto memo_cell = [&](int pos, int e, int w) -> int      // slot of (pos,e,w) in the on-chip memo, claimed empty (-inf) if new
        { const int key = pos*2+w, off = e ? LCAP0 : 0;
          const uint32_t m = (uint32_t)(e ? LCAP1 : LCAP0)-1;
          uint32_t h = ((uint32_t)key*0x9E3779B1u >> 9) & m;
          while (true)
            { const int32_t k = s_mkey[off+h];
              if (k == key) break;
              if (k < 0)
                { const int32_t o = atomicCAS(&s_mkey[off+h],-1,key);
                  if (o == -1) { s_mval[off+h] = -INFINITY; break; }
                  if (o == key) break;
                }
              h = (h+1) & m;
            }
          return off+(int)h;
        };
      for (;;)
        { if (onchip)
            { for (int q = lane; q < n_c; q += WAVE)
                { F.pos[q] = (uint16_t)clist[q]; F.fo[q] = ((cinfo[q] >> 12) & CP_CF_WALLNOW) ? CP_W_WALL_O : 0; F.fs[q] = 0; }
              wave_sync();
            }
          else flags_to_hbm();
          bool bail = false;
          for (int tb = 0; tb < n_t; tb += WAVE)
            { const int nb = (n_t-tb < WAVE) ? n_t-tb : WAVE;
              bool open = lane < nb;
              int pos = 1, kc = 0, e = 0, w = 0, lc_j = -1, lc_kind = CP_LC_NONE, hc_j = -1;
              double own_pe = 0., lc_v = 0., hc_pe = 0.;
              if (open)
                { const int code = tlist[tb+lane], k = code >> 1, cc = ccnt[k];
                  const task_res q = tres[tb+lane];
                  pos = clist[k]; kc = k; e = code & 1;
                  w = (cc & 0xffff) > ((cc >> 16) & 0xffff) ? CP_DROP : CP_GAIN;
                  own_pe = q.own_pe; lc_v = q.lc_v; hc_pe = q.hc_pe;
                  lc_j = q.lc_j; lc_kind = q.hc & 3;
                  hc_j = (q.hc & 4) ? pos+(q.hc >> 16) : -1;
                }
              const int p1 = (lc_kind == CP_LC_PAIR || lc_kind == CP_LC_BOUNDARY) ? lc_j : -1, p2 = hc_j;
              for (int q = lane; q < NDEP; q += WAVE) s_dep[q] = 0ull;
              wave_sync();
              const int s0 = (int)((uint32_t)(pos*2+e)*0x9E3779B1u >> 25), s1 = (int)((uint32_t)(p1*2+e)*0x9E3779B1u >> 25),
                        s2 = (int)((uint32_t)(p2*2+e)*0x9E3779B1u >> 25);
              if (open)
                { atomicOr(&s_dep[s0],1ull << lane);
                  if (p1 >= 0) atomicOr(&s_dep[s1],1ull << lane);
                  if (p2 >= 0) atomicOr(&s_dep[s2],1ull << lane);
                }
              wave_sync();
              uint64_t dep = 0;
              if (open)
                { dep = s_dep[s0];
                  if (p1 >= 0) dep |= s_dep[s1];
                  if (p2 >= 0) dep |= s_dep[s2];
                  dep &= ltm;
                }
              // The table is hashed (the keys of a chunk outnumber its 128 slots): a lane's mask holds every earlier task
              // whose keys fall into its slots, and 5.2 dependency rounds per chunk came of it where the tasks that really
              // share a position make 2.0 (diagnostic build, scripts/walk_profile.py).  Every bit is checked against the
              // keys of the lane it names (three ds_bpermute per step, as many steps as the fullest mask has bits).
              { const int k0 = open ? pos*2+e : -2, k1 = (open && p1 >= 0) ? p1*2+e : -1, k2 = (open && p2 >= 0) ? p2*2+e : -1;
                uint64_t rem = dep, keep = 0;
                while (__ballot(rem != 0))
                  { const int u = rem ? __ffsll((long long)rem)-1 : lane;
                    const int a0 = __shfl(k0,u), a1 = __shfl(k1,u), a2 = __shfl(k2,u);
                    const bool hit = a0 == k0 || a0 == k1 || a0 == k2
                                     || (a1 >= 0 && (a1 == k0 || a1 == k1 || a1 == k2))
                                     || (a2 >= 0 && (a2 == k0 || a2 == k1 || a2 == k2));
                    if (rem) { if (hit) keep |= 1ull << u; rem &= rem-1; }
                  }
                dep = keep;
              }
              bool hasE = false, hasO = false, off_list = false;
              cp_eintvl I; I.b = I.e = 0; I.pe = 0.;
#ifdef CP_PROF_WALK
              { // rounds this chunk takes with the hashed dependency table, and with exact key comparisons
                uint64_t depx = 0;
                const int k0 = pos*2+e, k1 = p1 >= 0 ? p1*2+e : -1, k2 = p2 >= 0 ? p2*2+e : -1;
                for (int u = 0; u < WAVE; u++)
                  { const int a0 = __shfl(k0,u), a1 = __shfl(k1,u), a2 = __shfl(k2,u);
                    const bool uo = __shfl((int)open,u) != 0;
                    if (uo && u < lane && (a0 == k0 || (k1 >= 0 && a0 == k1) || (k2 >= 0 && a0 == k2)
                                          || (a1 >= 0 && (a1 == k0 || a1 == k1 || a1 == k2)) || (a2 >= 0 && (a2 == k0 || a2 == k1 || a2 == k2))))
                      depx |= 1ull << u;
                  }
                int rh = 0, rx = 0;
                bool oh = open, ox = open;
                for (uint64_t um = __ballot(oh); um; um = __ballot(oh)) { if (oh && (dep & um) == 0) oh = false; rh++; }
                for (uint64_t um = __ballot(ox); um; um = __ballot(ox)) { if (ox && (depx & um) == 0) ox = false; rx++; }
                if (lane == 0) { atomicAdd(&g_live_prof[0],(unsigned long long)rh); atomicAdd(&g_live_prof[1],(unsigned long long)rx); atomicAdd(&g_live_prof[2],1ull); atomicAdd(&g_live_prof[3],(unsigned long long)nb); }
              }
#endif
              for (uint64_t um = __ballot(open); um; um = __ballot(open))
                { if (open && (dep & um) == 0)
                    { open = false;
                      const uint8_t fl = onchip ? ((e == CP_SELF) ? F.fs[kc] : F.fo[kc]) : ((e == CP_SELF) ? R.wall_s[pos] : wall[pos]);
                      if (!(fl & (e == CP_SELF ? CP_W_PAIRED_S : CP_W_PAIRED_O)))          // wall.c:639
                        { const int ci = memo_cell(pos,e,w);
                          double pe_i = s_mval[ci];
                          if (pe_i == CP_NEG_INF) { pe_i = own_pe; s_mval[ci] = pe_i; }     // update_perror(i), wall.c:310-315
                          bool accept = false;
                          if (pe_i >= CP_PE_THRES_FINAL && lc_kind != CP_LC_NONE)           // find_gain / find_drop, see cp_find_pair_replay
                            { const bool right = (w == CP_DROP);
                              int max_j = -1;
                              double pe = CP_NEG_INF, max_pe = CP_NEG_INF;
                              if (lc_kind == CP_LC_BOUNDARY) pe = pe_i*pe_i;
                              else if (lc_kind == CP_LC_PAIR)
                                { const int cj = memo_cell(lc_j,e,1-w);
                                  double pe_j = s_mval[cj];
                                  if (pe_j == CP_NEG_INF) { pe_j = lc_v; s_mval[cj] = pe_j; }
                                  pe = right ? pe_i*pe_j : pe_j*pe_i;
                                }
                              if (max_pe < pe) { max_j = lc_j; max_pe = pe; }
                              if (hc_j >= 0 && max_pe < hc_pe) { max_j = hc_j; max_pe = hc_pe; }
                              if (max_j != -1 && max_pe >= CP_PE_THRES_FINAL)
                                { accept = true;
                                  I.b = right ? pos : max_j; I.e = right ? max_j : pos; I.pe = max_pe;
                                  const int kj = onchip ? cf_find(F,max_j) : 0;
                                  // A partner that is no candidate: its SELF flag would be a wall the later phases must see --
                                  // the read starts over; its OTHERS flag nobody ever reads (a task reads the flag of its own
                                  // position, a candidate, and the ends of O-pairs are un-walled after the walk): not kept.
                                  if (kj < 0 && e == CP_SELF)
                                    { // No task ever reads this flag (a task reads the flag of its own position, a candidate);
                                      // the multi-error search after the walk does, position by position: the partner goes to a
                                      // short list of off-list SELF walls (cf_wall_mult), the candidate's own flag stays here.  A
                                      // list that is full sends the flag -- and after the replay all the others -- to the arrays.
                                      F.fs[kc] = (CP_W_WALL_S|CP_W_PAIRED_S);
                                      bool known = false;
                                      const int nx0 = s_nx;
                                      for (int k = 0; k < nx0 && k < FW_XCAP; k++) known = known || (int)s_x[k] == max_j;
                                      if (!known)
                                        { const int sl = atomicAdd(&s_nx,1);
                                          if (sl < FW_XCAP) s_x[sl] = (uint16_t)max_j;
                                          else { R.wall_s[max_j] = (CP_W_WALL_S|CP_W_PAIRED_S); off_list = true; }
                                        }
                                      hasE = true;
                                    }
                                  else if (e == CP_SELF)                                     // wall.c:655-668 (the SELF array only ever holds these two bits)
                                    { if (onchip) { F.fs[kc] = (CP_W_WALL_S|CP_W_PAIRED_S); F.fs[kj] = (CP_W_WALL_S|CP_W_PAIRED_S); }
                                      else { R.wall_s[I.b] = (CP_W_WALL_S|CP_W_PAIRED_S); R.wall_s[I.e] = (CP_W_WALL_S|CP_W_PAIRED_S); }
                                      hasE = true;
                                    }
                                  else                                                       // wall.c:678-686
                                    { if (onchip)
                                        { F.fo[kc] = fl | CP_W_PAIRED_O;
                                          if (kj < 0) { }
                                          else if (max_j > pos) F.fo[kj] = CP_W_PAIRED_O;    // not reached yet: the reference has 0 there (cp_read_t::spec_wallnow)
                                          else                  F.fo[kj] |= CP_W_PAIRED_O;
                                        }
                                      else
                                        { wall[pos] = fl | CP_W_PAIRED_O;
                                          if (max_j > pos) wall[max_j] = CP_W_PAIRED_O;
                                          else             wall[max_j] |= CP_W_PAIRED_O;
                                        }
                                      hasO = true;
                                    }
                                }
                            }
                          if (!accept && e == CP_OTHERS)                                     // wall.c:673-676,688
                            { if (onchip) F.fo[kc] = fl | CP_W_WALL_O; else wall[pos] = fl | CP_W_WALL_O; }
                        }
                    }
                  wave_sync();                                     // the round's flags and memo entries are visible to the next
                }
              if (__ballot(off_list)) bail = true;            // (not a restart any more: see below)
              const uint64_t mE = __ballot(hasE), mO = __ballot(hasO);
              if (hasE)
                { const int sl = eidx+__popcll(mE & ltm);
                  if (sl < ecap) { if (ebig) R.eintvl.g[sl] = I; else R.eintvl.l[sl] = I; }
                }
              if (hasO)
                { const int sl = oidx+__popcll(mO & ltm);
                  if (sl < ecap) R.ointvl.g[sl] = I;
                }
              eidx += __popcll(mE); oidx += __popcll(mO);
              if (eidx > ecap) { eidx = ecap; ovf = 1; }
              if (oidx > ecap) { oidx = ecap; ovf = 1; }
            }
          wave_sync();
          if (bail)                                        // more SELF pairs end off the candidates than the list holds: the phases
            {                                              // after the walk work on the flag ARRAYS
#ifdef CP_PROF_WALK
              if (lane == 0) atomicAdd(&g_live_prof[6],1ull);
#endif
              for (int q = lane; q < n_c; q += WAVE)
                { const int i = F.pos[q];
                  if (F.fo[q]) wall[i] = F.fo[q];
                  if (F.fs[q]) R.wall_s[i] = F.fs[q];
                }
              if (lane < FW_XCAP) R.wall_s[s_x[lane]] = (CP_W_WALL_S|CP_W_PAIRED_S);     // (the list is full)
              onchip = false;                              // (until the end of round 5 such a read started the replay over on the arrays:
              wave_sync();                                 //  11 % of the bench's reads, twice through the most expensive phase)
            }
          break;
        }
      if (lane < 2) { R.eidx = lane == 0 ? eidx : 0; R.oidx = lane == 1 ? oidx : 0; R.overflow |= ovf; }
    }
  else
  {
  flags_to_hbm();
  // one-lane replay, 64 tasks at a time: the chunk's task results come from HBM into LDS, then the chunk is replayed in order,
  //    lane 0 taking the SELF tasks and lane 1 the OTHERS tasks (one lane active per task; position and count pair
  //    come from the loading lane's registers by readlane).
  if (lane < 2)
    { R.use_win = (P->K+24+CP_MAX_N_HC < 128) ? 1 : 0;
      R.win_base = 0; R.win_lo = R.win_hi = 0;
      R.spec_wallnow = 1;
    }
  for (int tb = 0; tb < n_t; tb += WAVE)
    { const int nb = (n_t-tb < WAVE) ? n_t-tb : WAVE;
      int code_l = 0, cc_l = 0, pos_l = 1;
      if (lane < nb)
        { const int code = tlist[tb+lane], k = code >> 1;
          code_l = code; cc_l = ccnt[k]; pos_l = clist[k];
          s_res[lane] = tres[tb+lane];
        }
      wave_sync();
      for (int k = 0; k < nb; k++)
        { const int code = __builtin_amdgcn_readlane(code_l,k), cc = __builtin_amdgcn_readlane(cc_l,k);
          const int pos = __builtin_amdgcn_readlane(pos_l,k);
          if (lane == (code & 1))
            { cp_cand_pure c;
              const task_res &q = s_res[k];
              c.flags = CP_CF_LIVE;
              c.own_pe = q.own_pe; c.lc_v = q.lc_v; c.hc_pe = q.hc_pe;
              c.lc_j = q.lc_j; c.lc_kind = q.hc & 3;
              c.hc_j = (q.hc & 4) ? pos+(q.hc >> 16) : -1;
              cp_wall_candidate_replay(&R,pos,lane,(cc & 0xffff) > ((cc >> 16) & 0xffff) ? CP_DROP : CP_GAIN,c);
            }
        }
      wave_sync();
    }
  }
  R.use_win = 0; R.spec_wallnow = 0;
  int n_x = 0;                                          // off-list SELF walls of a read whose flags stayed on chip, in increasing order
  if (onchip)
    { n_x = __builtin_amdgcn_readfirstlane(s_nx);
      if (n_x > FW_XCAP) n_x = FW_XCAP;
      if (lane == 0)
        for (int a = 1; a < n_x; a++)
          { const uint16_t v = s_x[a];
            int b = a-1;
            while (b >= 0 && s_x[b] > v) { s_x[b+1] = s_x[b]; b--; }
            s_x[b+1] = v;
          }
      wave_sync();
    }
  PH_STAMP(1);
  int NS = __shfl(R.eidx,0), NO = __shfl(R.oidx,1);
  int overflow = __shfl(R.overflow,0) | __shfl(R.overflow,1);
  R.eidx = NS; R.oidx = NO;
  wave_sync();

  // ---- un-wall positions explained by O-pairs / inside E-intervals (wall.c:722-731) --------
  // From here on the lists of the walk are dead except the candidate positions (clist, second quarter of
  // wl); s_res is reused for the short lists of the phases below.
  int *s_cb = sbuf+SBUF-2*SCOMP, *s_ce = sbuf+SBUF-SCOMP;
  // positions fit 16 bits and the lists of this read fit the slots: O-only walls (<= n_c) during the multi-error search,
  // then the walls outside error regions (<= n_c) followed by the boundaries (<= n_c + 2*SCOMP + 1)
  const bool on16 = plen <= 65535 && 2*n_c+2*SCOMP+1 <= U16;
  R.eintvl.big = __shfl(R.eintvl.big,0);                // (the replay's lanes cannot have moved it: see above)
  // (the "paired by OTHERS" bit goes too: nothing reads it after the walk, and an O-pair partner that is no candidate
  //  must be left clean for the next batch -- the O-interval list is reused as sort scratch below)
  if (onchip)
    { for (int k = lane; k < NO; k += WAVE)           // (pairs that share an end clear the same bits of it)
        { const cp_eintvl o = R.ointvl.g[k];
          const int kb = cf_find(F,o.b), ke = cf_find(F,o.e);
          if (kb >= 0) F.fo[kb] &= (uint8_t)~(CP_W_WALL_O|CP_W_PAIRED_O);
          if (ke >= 0) F.fo[ke] &= (uint8_t)~(CP_W_WALL_O|CP_W_PAIRED_O);
        }
    }
  else if (lane == 0)
    for (int k = 0; k < NO; k++)
      { const cp_eintvl o = R.ointvl.g[k];
        wall[o.b] &= ~(CP_W_WALL_O|CP_W_PAIRED_O);
        wall[o.e] &= ~(CP_W_WALL_O|CP_W_PAIRED_O);
      }
  wave_sync();
  // ---- the candidates' flags on chip for the phases after the walk as well (positions in s16[0..n_c), the two flag bytes
  //      behind them, the boundaries from s16[2*n_c) on): they are there already when the replay ran on them; a read that
  //      replayed on the arrays copies them in when its lists fit and every flagged position is a candidate ----
  bool cf = onchip;
  if (!cf && plen <= 65535 && n_c <= 255 && 3*n_c+2*SCOMP+1 <= U16 && !R.eintvl.big)
    { for (int q = lane; q < n_c; q += WAVE)
        { const int i = clist[q];
          F.pos[q] = (uint16_t)i; F.fo[q] = wall[i]; F.fs[q] = wall_s[i];
        }
      wave_sync();
      bool miss = false;                               // an E-interval that ends off the candidates (a boundary pair, a partner below the scan's threshold)
      for (int k = lane; k < NS; k += WAVE)
        miss = miss || cf_find(F,R.eintvl.b(k)) < 0 || cf_find(F,R.eintvl.e(k)) < 0;
      cf = __ballot(miss) == 0;
    }
  // With the flags on chip and a list of at most 64 intervals the sort (wall.c:734) comes first -- neither step reads what
  // the other writes, and the duplicates the sort drops cover nothing their twins do not -- and "strictly inside one of
  // the intervals" is then a binary search: with the intervals in order of their begins, position i lies inside one iff
  // the largest end among the intervals that begin before i exceeds i (a prefix maximum over the lanes, one interval
  // each).  Every candidate used to test every interval (45 of them, two LDS reads each, and most candidates lie in none).
  bool unwalled = false;
#ifdef CP_PROF_WALK
  if (lane == 0) { atomicAdd(&g_emit_prof[0],NS <= WAVE ? 1ull : 0ull); atomicAdd(&g_emit_prof[1],(unsigned long long)NS); }
#endif
  if (cf && !R.eintvl.big && NS <= WAVE)
    { NS = wave_sort_ev(R.eintvl,NS,R.ointvl.g,true);
      int b_k = 0x7fffffff, pm = -1;
      if (lane < NS) { b_k = R.eintvl.l[lane].b; pm = R.eintvl.l[lane].e; }
      for (int o = 1; o < WAVE; o <<= 1) { const int y = __shfl_up(pm,o); if (lane >= o && y > pm) pm = y; }
      for (int base = 0; base < n_c; base += WAVE)         // (every lane takes part in the shuffles)
        { const int q = base+lane;
          int below = 0, i = 0;
          const bool test = q < n_c && (F.fo[q] & CP_W_WALL_O);
          if (test)
            { i = F.pos[q];
              int lo = 0, hi = NS;                         // intervals that begin before i
              while (lo < hi) { const int m = (lo+hi) >> 1; if (R.eintvl.l[m].b < i) lo = m+1; else hi = m; }
              below = lo;
            }
          const int pmax = __shfl(pm,below > 0 ? below-1 : 0);
          if (test && below > 0 && pmax > i) F.fo[q] &= (uint8_t)~CP_W_WALL_O;
        }
      wave_sync();
      unwalled = true;
    }
  else if (cf) cf_unwall_inside(F,R.eintvl,0,NS);
  else    wave_unwall_inside(wall,clist,n_c,R.eintvl,0,NS,sbuf,SBUF/2);

  // ---- sort + dedupe E-intervals (wall.c:734); the O list is not used again ------------------
  if (!unwalled) NS = wave_sort_ev(R.eintvl,NS,R.ointvl.g,true);

  // ---- multi-error / boundary E-intervals (wall.c:760-861) -----------------------------------
  // The reference scans every position for O-only walls; OTHERS walls are only ever set at wall
  // candidates, so the lanes test the candidates and compact the qualifying positions, in order, into a
  // short list that the wave then processes one by one (processing a wall can mark later ones as
  // paired, wall.c:766-767).
  int32_t *olist = wl, *compB = wl+2*(int64_t)icap, *compE = compB+(icap >> 1), *bnd = wl+3*(int64_t)icap;
  const int ccap = icap >> 1;                           // components <= candidates+1 <= icap/2
  int midx = NS;
  // Most O-only walls are haplotype boundaries, not errors: cp_wall_mult (wall.c:763-860) does nothing at all for a wall
  // whose SELF-pass P(error) is below the threshold for both the DROP and the GAIN reading (wall.c:765, 808) -- and nothing
  // writes the memo in this phase -- so the lanes look the two entries up for all walls at once and only the walls that
  // pass are taken one by one (the whole wave went through every wall, two dependent memo probes and a barrier each).
  auto mult_live = [&](int i) -> bool
    {
#ifdef CP_NO_MULT_PREFILTER                              // (A/B build: every O-only wall is taken, as before)
      (void)i; return true;
#else
      return !(CP_PERR(&R,i,CP_SELF,CP_DROP) < CP_PE_THRES_FINAL) || !(CP_PERR(&R,i,CP_SELF,CP_GAIN) < CP_PE_THRES_FINAL);
#endif
    };
  if (cf)                                              // the O-only walls are met candidate by candidate, in order
    { PH_STAMP(2);
      for (int base = 0; base < n_c; base += WAVE)
        { const int q = base+lane;
          const uint64_t m = __ballot(q < n_c && (F.fo[q] & CP_W_WALL_O) && !(F.fs[q] & CP_W_WALL_S) && mult_live(F.pos[q]));
          for (uint64_t t = m; t; t &= t-1)
            { const int qq = base+__ffsll((long long)t)-1;
              wave_sync();                             // lane 0's flag updates of the previous wall
              if (F.fo[qq] & CP_W_PAIRED_M)            // may have been set by an earlier wall
                continue;
              cf_wall_mult(&R,F,qq,NS,&midx,s_x,n_x);
            }
        }
    }
  else
    { int n_o = 0;
      for (int base = 0; base < n_c; base += WAVE)
        { const int q = base+lane;
          int i = 0; bool keep = false;
          if (q < n_c)
            { i = clist[q];
              keep = (wall[i] & CP_W_WALL_O) && !(wall_s[i] & CP_W_WALL_S) && mult_live(i);
            }
          const uint64_t m = __ballot(keep);
          if (keep)
            { const int o = n_o+__popcll(m & ((1ull << lane)-1));       // n_o < n_c <= icap
              if (on16) s16[o] = (uint16_t)i; else olist[o] = i;
            }
          n_o += __popcll(m);
        }
      wave_sync();
      PH_STAMP(2);
      for (int q = 0; q < n_o; q++)                    // all lanes together, see wave_wall_mult
        { const int ii = on16 ? (int)s16[q] : olist[q];
          wave_sync();                                 // lane 0's flag stores of the previous wall
          if (wall[ii] & CP_W_PAIRED_M)                // may have been set by an earlier i
            continue;
          wave_wall_mult(&R,ii,NS,&midx);
        }
    }
  overflow |= R.overflow;
  wave_sync();
  PH_STAMP(3);
  if (cf) cf_unwall_inside(F,R.eintvl,NS,midx);        // wall.c:868-872
  else    wave_unwall_inside(wall,clist,n_c,R.eintvl,NS,midx,sbuf,SBUF/2);
  if (NS < midx)                                       // wall.c:873-876
    { const int ns_sorted = NS;
      NS = midx;
      wave_sort_ev(R.eintvl,NS,R.ointvl.g,false,ns_sorted);
    }
  const int ns_before_merge = NS;
  // wall.c:878-909: runs of intervals each of which begins inside its predecessor are united, the unions appended.  The
  // reference's scan is bounded by the GROWING length, so the run that holds the last interval goes on into the unions
  // appended before it (cp_merge_eintvl's header); every run before that one is what it looks like in the sorted list.
  // Those runs are found by all lanes at once (a lane per interval: "my successor begins inside me", one ballot) and
  // united by the lane of their first interval, the unions appended in order; one lane then scans on from the first
  // interval of the last run, as the reference does from there.  (One lane used to walk the whole list, four dependent
  // LDS reads per interval, while 63 waited.)  Lists beyond 64 intervals or close to a capacity take the one-lane scan.
  int merge_from = 0;
#ifdef CP_PROF_WALK
  if (lane == 0) { atomicAdd(&g_emit_prof[2],NS <= WAVE ? 1ull : 0ull); atomicAdd(&g_emit_prof[3],(unsigned long long)NS); }
#endif
  if (!R.eintvl.big && NS >= 2 && NS <= WAVE)
    { cp_eintvl x = { 0, 0, 0. };
      if (lane < NS) x = R.eintvl.l[lane];
      const int b_next = __shfl_down(x.b,1);
      const bool link = lane < NS-1 && b_next <= x.e;      // interval lane+1 continues the run of interval lane
      const uint64_t Lm = __ballot(link);
      const bool head = lane < NS && (lane == 0 || !((Lm >> (lane-1)) & 1));
      const uint64_t Hm = __ballot(head);
      const int hl = 63-__clzll((long long)Hm);            // first interval of the run that holds the last interval
      const uint64_t Mm = __ballot(head && link && lane != hl);
      const int m1 = __popcll(Mm);
      int lim = R.ecap < plen ? R.ecap : plen;
      if (lim > FW_EVL) lim = FW_EVL;
      if (NS+2*(m1+1) < lim)                               // (every append of the scan stays below the capacities: no overflow path, the list stays on chip)
        { if ((Mm >> lane) & 1)
            { const int last = lane+__ffsll((long long)~(Lm >> lane))-1;     // the run is lane .. last
              int max_e = x.e;
              double max_pe = x.pe;
              for (int k = lane+1; k <= last; k++)
                { const cp_eintvl y = R.eintvl.l[k];
                  if (max_e < y.e) max_e = y.e;
                  if (!(max_pe > y.pe)) max_pe = y.pe;
                }
              cp_eintvl u; u.b = x.b; u.e = max_e; u.pe = max_pe;
              R.eintvl.l[NS+__popcll(Mm & ((1ull << lane)-1))] = u;
            }
          NS += m1;
          merge_from = hl;
          wave_sync();
        }
    }
  if (lane == 0)
    NS = cp_merge_eintvl(&R,NS,merge_from);
  NS = __shfl(NS,0);
  overflow |= __shfl(R.overflow,0);
  R.eintvl.big = __shfl(R.eintvl.big,0);               // lane 0's appends may have moved the list to HBM
  wave_sync();
  wave_sort_ev(R.eintvl,NS,R.ointvl.g,false,ns_before_merge);   // wall.c:910 (the merge only appended)
  PH_STAMP(4);

  // ---- interval boundaries (wall.c:917-948) without per-position passes -----------------------
  // The reference marks every position of every E-interval as "error" and then scans all positions
  // for error transitions and OTHERS walls outside error regions.  Equivalent, on lists:
  //   error regions  = connected components of the union of the sorted E-intervals,
  //   boundaries     = component starts (>= 1) and ends, OTHERS walls (candidate positions) outside
  //                    every component, and plen; merged in increasing order.
  // Short lists (the usual case) stay in LDS, so that the one-lane loops below step through on-chip memory instead of
  // paying a global-memory round trip per element: the walls in s16[0..n_w), the boundaries in s16[n_c..), the
  // component starts / ends in s_cb / s_ce.
  EM_LT0();
  int C = 0;
#ifdef CP_PROF_WALK
  if (lane == 0) { atomicAdd(&g_emit_prof[4],NS <= WAVE ? 1ull : 0ull); atomicAdd(&g_emit_prof[5],(unsigned long long)NS); }
#endif
  if (NS <= WAVE && ccap >= WAVE)                       // a lane per E-interval: a component starts where b exceeds every end before it
    { int b_k = 0, e_k = -1;
      if (lane < NS) { b_k = R.eintvl.b(lane); e_k = R.eintvl.e(lane); }
      int inc = e_k;                                    // inclusive prefix maximum of the ends
      for (int o = 1; o < WAVE; o <<= 1) { const int y = __shfl_up(inc,o); if (lane >= o && y > inc) inc = y; }
      const int exc = __shfl_up(inc,1);
      const bool st = lane < NS && (lane == 0 || b_k > exc);
      const uint64_t sm = __ballot(st);
      const bool last = lane < NS && (lane == NS-1 || ((sm >> (lane+1)) & 1));
      const int cid = __popcll(sm & ((2ull << lane)-1))-1;
      if (st)   { compB[cid] = b_k; s_cb[cid] = b_k; }
      if (last) { compE[cid] = inc; s_ce[cid] = inc; }
      C = __popcll(sm);
    }
  else if (lane == 0)
    { int k = 0;
      while (k < NS)
        { int cb = R.eintvl.b(k), ce = R.eintvl.e(k);
          k++;
          while (k < NS && R.eintvl.b(k) <= ce)
            { const int e2 = R.eintvl.e(k);
              if (e2 > ce) ce = e2;
              k++;
            }
          if (C < ccap) { compB[C] = cb; compE[C] = ce; }
          if (C < SCOMP) { s_cb[C] = cb; s_ce[C] = ce; }
          C++;
        }
    }
  if (!(NS <= WAVE && ccap >= WAVE)) C = __shfl(C,0);
  if (C > ccap) { overflow |= 2; C = ccap; }
  const bool smallc = on16 && C <= SCOMP;               // components, walls and boundaries all on chip
  // (with the flags on chip the walls are kept as candidate numbers, a byte each, where the SELF flags were -- those
  //  are dead after the multi-error search -- and the boundaries follow the flags; else: walls from s16[0), boundaries behind)
  uint16_t *s_bnd = cf ? s16+2*n_c : s16+n_c;
  uint8_t *s_wq = F.fs;
  wave_sync();
  int N = 0;
  if (cf && smallc)
    { // The boundaries = the union of the component starts / ends inside (0,plen) and the OTHERS walls outside every
      // component, in order, then plen.  Every member's place in that order is a count: for a wall, the transitions
      // below it (two per component that ends at or before it) plus the walls below it; for a transition, its number
      // among the transitions plus the walls below it (a running count kept per candidate).  No merge loop.
      bool vb = false, ve = false;
      int cbk = 0, cek = 0;
      if (lane < C) { cbk = s_cb[lane]; cek = s_ce[lane]; vb = cbk >= 1 && cbk < plen; ve = cek >= 1 && cek < plen; }
      const uint64_t tb = __ballot(vb), te = __ballot(ve), ltm = (1ull << lane)-1;
      int nwl = 0;                                      // walls so far that are no transition
      for (int base = 0; base < n_c; base += WAVE)
        { const int q = base+lane;
          bool w = false; int i = 0, below = 0;
          if (q < n_c && (F.fo[q] & CP_W_WALL_O))
            { i = F.pos[q];
              int a2 = 0, z = C-1, in = 0;              // a2: components that end at or before i
              while (a2 <= z)
                { const int m = (a2+z) >> 1;
                  if (i < s_cb[m]) z = m-1;
                  else if (i >= s_ce[m]) a2 = m+1;
                  else { in = 1; break; }
                }
              if (!in)
                { const uint64_t am = a2 >= WAVE ? ~0ull : ((1ull << a2)-1);
                  const bool int_ = a2 > 0 && s_ce[a2-1] == i && ((te >> (a2-1)) & 1);   // the wall sits on a component's end: that transition stands for it
                  below = __popcll(tb & am)+__popcll(te & am)-(int_ ? 1 : 0);
                  w = !int_;
                }
            }
          const uint64_t wm = __ballot(w);
          if (w) s_bnd[below+nwl+__popcll(wm & ltm)] = (uint16_t)i;
          if (q < n_c) s_wq[q] = (uint8_t)(nwl+__popcll(wm & (ltm | (1ull << lane))));   // walls up to and including candidate q
          nwl += __popcll(wm);
        }
      wave_sync();
      if (vb || ve)
        { // candidates below the transition -> walls below it
          auto walls_below = [&](int t) -> int
            { int lo = 0, hi = n_c;                     // first candidate at or beyond t
              while (lo < hi) { const int m = (lo+hi) >> 1; if ((int)F.pos[m] < t) lo = m+1; else hi = m; }
              return lo > 0 ? (int)s_wq[lo-1] : 0;
            };
          const int tbefore = __popcll(tb & ltm)+__popcll(te & ltm);
          if (vb) s_bnd[tbefore+walls_below(cbk)] = (uint16_t)cbk;
          if (ve) s_bnd[tbefore+(vb ? 1 : 0)+walls_below(cek)] = (uint16_t)cek;
        }
      N = __popcll(tb)+__popcll(te)+nwl+1;
      if (lane == 0) s_bnd[N-1] = (uint16_t)plen;
    }
  else
    {
  int n_w = 0;                                         // OTHERS walls outside error regions, in order
  for (int base = 0; base < n_c; base += WAVE)
    { const int q = base+lane;
      int i = 0; bool keep = false;
      if (q < n_c)
        { i = cf ? (int)F.pos[q] : clist[q];
          if ((cf ? F.fo[q] : wall[i]) & CP_W_WALL_O)
            { int a = 0, z = C-1, in = 0;              // inside a component?  compB sorted, disjoint
              while (a <= z)
                { int m = (a+z) >> 1;
                  const int mb = smallc ? s_cb[m] : compB[m], me = smallc ? s_ce[m] : compE[m];
                  if (i < mb) z = m-1;
                  else if (i >= me) a = m+1;
                  else { in = 1; break; }
                }
              keep = !in;
            }
        }
      const uint64_t m = __ballot(keep);
      if (keep)
        { const int o = n_w+__popcll(m & ((1ull << lane)-1));
          if (cf && smallc) s_wq[o] = (uint8_t)q;
          else if (smallc) s16[o] = (uint16_t)i;
          else olist[o] = i;                           // n_w < n_c <= icap
        }
      n_w += __popcll(m);
    }
  wave_sync();
  if (lane == 0)                                       // merge: transitions, walls, plen
    { int ci = 0, phase = 0, wi = 0, last = 0;         // phase 0: next transition is compB[ci], 1: compE[ci]
      while (true)
        { int tpos = plen;
          while (ci < C)
            { int v = smallc ? (phase ? s_ce[ci] : s_cb[ci]) : (phase ? compE[ci] : compB[ci]);
              if (v >= 1 && v < plen && v > last) { tpos = v; break; }
              if (phase) { ci++; phase = 0; } else phase = 1;
            }
          int wpos = (wi < n_w) ? (smallc ? (cf ? (int)F.pos[s_wq[wi]] : (int)s16[wi]) : olist[wi]) : plen;
          int nb = tpos < wpos ? tpos : wpos;
          if (nb >= plen) break;
          if (N < icap) { if (smallc) s_bnd[N] = (uint16_t)nb; else bnd[N] = nb; }
          N++;
          last = nb;
          if (wpos == nb) wi++;
          if (tpos == nb) { if (phase) { ci++; phase = 0; } else phase = 1; }
        }
      if (N < icap) { if (smallc) s_bnd[N] = (uint16_t)plen; else bnd[N] = plen; }
      N++;
    }
    }
  if (!(cf && smallc)) N = __shfl(N,0);
  if (N > icap) overflow |= 2;
  wave_sync();
  PH_STAMP(8);
  // wall.c:928-946, one lane per interval -- and, on the whole-path call (do_rel), find_rel_intvl / correct_wall_cnt
  // (wall.c:960-1051) on the record while it is still in the lane's registers: every interval is independent, the
  // reliable ones are compacted in order (ballot + popcount) into rintvl / relmap.  As a kernel of its own (k_find_rel,
  // now only with CLASSPRO_FUSE_REL=0) this pass re-read every record, wrote it a second time and sat 0.9 ms per 1-Gbase sub-batch
  // between the walk and the classification (fused: 2.13 ms for both against 1.53 + 0.95 alone on the machine).
  // (Tried on top: the four step sums of correct_wall_cnt from two 64-count register windows fetched by sixteen
  //  independent loads, instead of cp_sum_steps' eight counts per dependent load -- as an inlined body 205 registers
  //  spilled, as a call 19 and 2.18 ms: the sums' loads are not what this pass waits for.)
  { const int Ncl = N < icap ? N : icap;
    cp_intvl *rintvl = rintvl_all+ioff[r];
    int32_t *relmap = relmap_all+ioff[r];
    // whole-path calls (do_rel == 2): the records as arrays (cp_soa); the 48-byte form only for a read of a rare
    // classify_unrel class, whose kernels work on it
    const bool use_soa = do_rel == 2 && soa.a != nullptr;
    const bool aos = !use_soa || Ncl > UNREL_SMALL_MAXN || plen > GRP_MAX_PLEN;
    int M = 0;
    EM_LT(0);
    // Two steps of 64 intervals at a time.  About half of a read's intervals fail the three record-only tests of
    // wall.c:1016-1019 (the E-intervals, mostly), and correct_wall_cnt -- four step sums over up to K-1 counts and six
    // context scans: most of this loop's time -- would run with half of each step's lanes idle.  The intervals that pass
    // are packed into one step instead: their (b, e, cb | ce, index) travel to the lanes 0 .. n-1 by ds_permute (a lane
    // pushes to the rank of its interval among the passing ones; the lanes that hold none push to the slots left over, so
    // that every push is a permutation of the wave), any lane does the counts for what it got (cp_rel_counts is a
    // function of those five numbers and the read), and the owner pulls ccb | cce | reliable back by ds_bpermute.  Two
    // steps with more than 64 passing intervals between them take two rounds.
    // (What find_rel does not touch of a record -- the three log-probabilities -- is stored straight after cp_make_interval,
    //  so that two steps' worth of it does not sit in registers across the counts: b, e, cb, ce and pe stay.)
    for (int base = 0; base < Ncl; base += 2*WAVE)
      { const int k0 = base+lane, k1 = base+WAVE+lane;
        int b0 = 0, e0 = 0, c0 = 0, b1 = 0, e1 = 0, c1 = 0;          // c: cb | ce << 16, later ccb | cce << 16 | reliable << 15
        double pe0 = 0., pe1 = 0.;
        bool p0 = false, p1 = false;
        auto make = [&](int k, int &b, int &e, int &c, double &pe, bool &p) __attribute__((always_inline))
          { cp_intvl I;
            b = k ? (smallc ? (int)s_bnd[k-1] : bnd[k-1]) : 0; e = smallc ? (int)s_bnd[k] : bnd[k];
            cp_make_interval(&R,NS,b,e,&I);
            p = do_rel && cp_rel_prefilter(P,&I);
            c = (int)I.cb | ((int)I.ce << 16); pe = I.pe;
            if (aos) intvl[k] = I;
            if (use_soa) { cp_ivB q; q.pe = I.pe; q.peo_b = I.peo_b; q.peo_e = I.peo_e; soa.b[ioff[r]+k] = q; }
          };
        if (k0 < Ncl) make(k0,b0,e0,c0,pe0,p0);
        if (k1 < Ncl) make(k1,b1,e1,c1,pe1,p1);
        EM_LT(1);
        int res0 = 0, res1 = 0;                                       // ccb | cce << 16 | reliable << 15
        if (do_rel)
          { const uint64_t m0 = __ballot(p0), m1 = __ballot(p1);
            const int n0 = __popcll(m0), n1 = __popcll(m1);
            const int lt0 = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32),__builtin_amdgcn_mbcnt_lo((uint32_t)m0,0u));
            const int lt1 = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32),__builtin_amdgcn_mbcnt_lo((uint32_t)m1,0u));
            const bool both = n0+n1 <= WAVE;
            const int nrounds = (both || n1 == 0) ? 1 : 2;
            const int tA = p0 ? lt0 : n0+(lane-lt0);                  // a permutation of 0 .. 63: the passing ones first, in order
            for (int round = 0; round < nrounds; round++)
              { const bool useA = both || round == 0, useB = n1 > 0 && (both || round == 1);
                const int offB = both ? n0 : 0;
                const int tB = (p1 ? offB+lt1 : offB+n1+(lane-lt1)) & (WAVE-1);
                int xb = 0, xe = 0, xc = 0, xk = 0;
                bool act = false;
                if (useA)
                  { xb = __builtin_amdgcn_ds_permute(tA*4,b0);
                    xe = __builtin_amdgcn_ds_permute(tA*4,e0);
                    xc = __builtin_amdgcn_ds_permute(tA*4,c0);
                    xk = base+__builtin_amdgcn_ds_permute(tA*4,lane);
                    act = lane < n0;
                  }
                if (useB)
                  { const int yb = __builtin_amdgcn_ds_permute(tB*4,b1);
                    const int ye = __builtin_amdgcn_ds_permute(tB*4,e1);
                    const int yc = __builtin_amdgcn_ds_permute(tB*4,c1);
                    const int yk = base+WAVE+__builtin_amdgcn_ds_permute(tB*4,lane);
                    if (lane >= offB && lane < offB+n1) { xb = yb; xe = ye; xc = yc; xk = yk; act = true; }
                  }
                int res = 0;
                if (act)
                  { cp_seq_rsrc rs; rs.g = R.seq;
                    int ccb, cce;
                    const bool rel = cp_rel_counts(P,R.prof,rs,rs,rlen,xb,xe,xc & 0xffff,(xc >> 16) & 0xffff,xk,&ccb,&cce);
                    res = ccb | (cce << 16) | (rel ? 0x8000 : 0);      // counts are at most CP_MAX_KMER_CNT = 0x7fff
                  }
                if (useA) { const int g = __builtin_amdgcn_ds_bpermute(tA*4,res); if (p0) res0 = g; }
                if (useB) { const int g = __builtin_amdgcn_ds_bpermute(tB*4,res); if (p1) res1 = g; }
              }
          }
        EM_LT(2);
        auto put = [&](int k, int b, int e, int c, double pe, int res) __attribute__((always_inline))
          { const bool ok = (res & 0x8000) != 0;
            const uint16_t ccb = (uint16_t)(res & 0x7fff), cce = (uint16_t)((res >> 16) & 0x7fff);
            if (k < Ncl)
              { if (aos && res) { intvl[k].ccb = ccb; intvl[k].cce = cce; intvl[k].is_rel = ok ? 1 : 0; }
                if (use_soa)
                  { cp_ivA qa; qa.b = b; qa.e = e; qa.cb = (uint16_t)(c & 0xffff); qa.ce = (uint16_t)((c >> 16) & 0xffff); qa.ccb = ccb; qa.cce = cce;
                    cp_ivC qc; qc.is_rel = ok ? 1 : 0; qc.asgn = CP_N_STATE;
                    soa.a[ioff[r]+k] = qa; soa.c[ioff[r]+k] = qc;
                  }
              }
            if (do_rel)
              { const uint64_t mask = __ballot(ok);
                if (ok)
                  { const int rank = __popcll(mask & ((1ull << lane)-1));
                    if (do_rel == 2)                           // whole-path call: the compact record only
                      { cp_rrec q; q.b = b; q.e = e; q.ccb = ccb; q.cce = cce; q.idx = k; q.pe = pe;
                        rrec_all[ioff[r]+M+rank] = q;
                      }
                    else                                       // (the stage API: a copy of the record this lane has just completed)
                      { rintvl[M+rank] = intvl[k];
                        relmap[M+rank] = k;
                      }
                  }
                M += __popcll(mask);
              }
          };
        put(k0,b0,e0,c0,pe0,res0);
        if (base+WAVE < Ncl) put(k1,b1,e1,c1,pe1,res1);
        EM_LT(4);
      }
    if (do_rel && lane == 0) nrel[r] = M;
    // the rare size classes (M beyond the main class, or a read beyond GRP_MAX_PLEN k-mers: the one-read-per-wave and the
    // sequential kernels) work on full records: a read of theirs gets its copies here, from the records just written
    if (do_rel == 2 && (M > REL_SMALL_MAXM || plen > GRP_MAX_PLEN))
      { wave_sync();
        const cp_rrec *rr = rrec_all+ioff[r];
        for (int i = lane; i < M; i += WAVE)
          { const int k = rr[i].idx;
            rintvl[i] = aos ? intvl[k] : cp_soa_get(soa,ioff[r]+k);
            relmap[i] = k;
          }
      }
  }
  PH_STAMP(5);
  // Leave the flag arrays all zero (capi.hip fills them only when they are allocated): every cell the walk and the
  // multi-error phase wrote is a candidate position or an end of an E-interval of the final list (the ends of O-pairs
  // were cleaned above).  A read whose lists overflowed may have unrecorded cells: its whole range is cleared.
  wave_sync();
#ifdef CP_PROF_WALK
  if (lane == 0) { atomicAdd(&g_live_prof[4],R.perror.use_lds == 3 ? 1ull : 0ull); atomicAdd(&g_live_prof[5],onchip ? 1ull : 0ull); atomicAdd(&g_live_prof[7],cf ? 1ull : 0ull); }
#endif
  if (onchip) { }                                      // (this read's flags never left the chip)
  else if (overflow)
    { for (int j = lane; j <= plen; j += WAVE) { wall[j] = 0; R.wall_s[j] = 0; } }
  else
    { for (int q = lane; q < n_c; q += WAVE)
        { const int i = clist[q]; wall[i] = 0; R.wall_s[i] = 0; }
      for (int k = lane; k < NS; k += WAVE)
        { const int b = R.eintvl.b(k), e = R.eintvl.e(k);
          wall[b] = 0; wall[e] = 0; R.wall_s[b] = 0; R.wall_s[e] = 0;
        }
    }
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
  const CP_PROF_T pr = CP_PROF_VIEW(prof+prof_off[r],rlen-(P->K-1));
  const CP_SEQ_T sq = CP_SEQ_VIEW(seq+seq_off[r],rlen);
  cp_intvl *intvl = intvl_all+ioff[r], *rintvl = rintvl_all+ioff[r];
  int32_t *relmap = relmap_all+ioff[r];
  __shared__ __attribute__((aligned(4))) char s_ctx[WAVE*FR_STRIDE];
  int M = 0;
  for (int base = 0; base < N; base += WAVE)
    { int idx = base+lane;
      bool ok = false;
      cp_intvl I;
      if (idx < N)
        { I = intvl[idx];
          // the two context scans of correct_wall_cnt read forward from b+K-1 and backward from e-1: 32 bases each in LDS
          cp_seq_lwin sb, se;
          sb.g = se.g = sq;
          sb.w = (CP_LDS_PTR(const char))(s_ctx+lane*FR_STRIDE); se.w = (CP_LDS_PTR(const char))(s_ctx+lane*FR_STRIDE+32);
          sb.lo = se.lo = 0; sb.len = se.len = 0;
          if (I.e-I.b >= P->K)
            { fr_seq_win_load(sb,s_ctx+lane*FR_STRIDE,I.b+P->K-1,rlen);
              fr_seq_win_load(se,s_ctx+lane*FR_STRIDE+32,I.e-1-31,rlen);
            }
          ok = cp_rel_interval(P,pr,sb,se,rlen,&I,idx);
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
// (128 registers, the rest in scratch: this is the slow path whatever it holds, and a wave of 203 registers -- what the
//  compiler takes when left alone -- only starts where TWO of the main class's 128-register waves have left a SIMD: beside
//  the main class its 1024 waves, nearly all of which leave after two loads, took 1.8-2.7 ms to get on and off the machine;
//  now 0.15.  The stage is no shorter for it -- the next kernel on the auxiliary stream waits for its holes instead,
//  profiles/r05_rare_classes.txt -- but 1024 fat waves no longer queue for the register file.)
__global__ void __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(4)))
k_classify_rel(const cp_dev_params *__restrict__ P, const int64_t *__restrict__ prof_off, int nreads,
               cp_intvl *__restrict__ intvl_all, cp_intvl *__restrict__ rintvl_all, const int32_t *__restrict__ relmap_all,
               const int64_t *__restrict__ ioff, const int32_t *__restrict__ nrel,
               int8_t *__restrict__ parent_all, int32_t *__restrict__ eff_all, uint8_t *__restrict__ rpos_all,
               int8_t *__restrict__ asgn_all, int64_t totalI, const int32_t *__restrict__ perm, cp_soa soa)
{ const int lane = lane_id();
  // Few reads need this kernel, on most batches none.  `perm` lists the reads by decreasing M with the reads beyond
  // GRP_MAX_PLEN k-mers in the top bin as well (order_bin): the class is inside the top bin, at the front of the list,
  // and a wave leaves at its first read of a lower bin -- after two loads when the class is empty.  (Until round 5
  // every wave strode over the whole batch, three dependent cold loads per read: 0.3-0.5 ms of 1024 waves holding half
  // of every SIMD's registers beside the main class.)
  for (int i = blockIdx.x; i < nreads; i += gridDim.x)
  {
  const int r = perm[i];
  const int M = nrel[r];
  const int plen = (int)(prof_off[r+1]-prof_off[r]);
  if (M < ORDER_BINS-1 && plen <= GRP_MAX_PLEN) break;               // below the top bin: nothing of this class from here on
  if (M == 0 || (M <= REL_MAXM && plen <= GRP_MAX_PLEN)) continue;   // other reads: k_classify_rel_grp
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
      if (soa.c) soa.c[o+relmap[i]].asgn = a;              // whole-path calls: classify_unrel reads the class there
    }
  wave_sync();
  }
}

// ---------------------------------------------------------------------------------------------
//  k_classify_unrel: class_unrel.c:248-300.  Stable rank sort by min(cb,ce) across lanes; the two
//  sweeps are order-dependent (each update reads its neighbours' current classes) and run on lane 0.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(WAVE)
k_classify_unrel(const cp_dev_params *__restrict__ P, int nreads, cp_intvl *__restrict__ intvl_all,
                 const int64_t *__restrict__ ioff, const int32_t *__restrict__ nintvl,
                 int32_t *__restrict__ ord_all, const int32_t *__restrict__ perm, const int64_t *__restrict__ prof_off,
                 uint32_t *__restrict__ pcls_all, cp_soa soa)
{ const int lane = lane_id();
  for (int i = blockIdx.x; i < nreads; i += gridDim.x)     // the class is at the front of `perm`: see k_classify_rel
  {
  const int r = perm[i];
  const int N = nintvl[r];
  const int plen = (int)(prof_off[r+1]-prof_off[r]);
  if (N < ORDER_BINS-1 && plen <= GRP_MAX_PLEN) break;
  cp_intvl *intvl = intvl_all+ioff[r];
  if (N == 0 || (N <= UNREL_MAXN && plen <= GRP_MAX_PLEN)) continue;   // other reads: k_classify_unrel_grp (the last interval ends at plen)
  int32_t *ord = ord_all+ioff[r];
  if (soa.c)                                             // whole-path calls: classify_rel left its classes in the two-byte array
    { for (int k = lane; k < N; k += WAVE) intvl[k].asgn = soa.c[ioff[r]+k].asgn;
      wave_sync();
    }
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
  wave_sync();
  if (pcls_all)                                            // whole-path calls: (end, class) per interval for k_paint_labels (CP_PCLS)
    { for (int k = lane; k < N; k += WAVE) pcls_all[ioff[r]+k] = CP_PCLS(intvl[k].e,intvl[k].asgn);
      wave_sync();
    }
  }
}

// ---------------------------------------------------------------------------------------------
//  k_classify_rel_grp<MAXM,G>: class_rel.c:871-963, lane-parallel, G reads per wave (64/G >= 32 lanes
//  per read) for reads with MINM < M <= MAXM.
//
//  Per read, the forward pass runs on lanes 0-15 of its lane group and the backward pass on lanes
//  16-31 at the same time; inside a pass lane (s,t) evaluates the transition s@pred -> t@i.  The 8
//  H/D transitions of every read and direction meet at ONE convergent Bessel call (class_rel.c:213-270
//  are all `logp_trans` with different arguments), so a DP step of the whole wave costs one
//  recurrence.  Interval fields (16-bit), DP cells, back-pointers and both assignments live in LDS;
//  HBM is touched for pe (once per step) and for the results.
// ---------------------------------------------------------------------------------------------
// LDS record of a wave (G reads).  Sized for waves: with MAXM = 112, G = 4 and the compact cells it is 9.2 KB, and four
// four-wave blocks (with their 4 KB of libm tables each) fit a CU:
//   * `eff` (index of the interval whose data stands in for path index k) and the "absolutely repeat" flag rpos share
//     one element: the index in the low bits, the flag in the top bit (one byte while MAXM <= 128);
//   * the traceback overwrites a back-pointer byte with the assignment of the same interval once it has read it, so
//     `asgn` is the `parent` array seen after the traceback.
#ifndef REL_CELL_PAD
#define REL_CELL_PAD 2
#define REL_ROW_PAD  2
#define REL_TR_PAD   2
#endif
template <int MAXM, int G>
struct rel_grp_lds
  { typedef typename std::conditional<(MAXM <= 128),uint8_t,uint16_t>::type eff_t;
    static constexpr int RPOS = (MAXM <= 128) ? 0x80 : 0x8000;
    // Bank layout (64 banks of 4 B; a 32-lane half holds 4 (read, direction) groups): rows and records are padded so
    // that the groups' copies of one field fall on different banks -- unpadded, the 64-byte cells of the 16 (group,
    // state) pairs of a half shared 2-4 banks and 58 % of the kernel's LDS cycles were conflict cycles
    // (profiles/r03_sq_counters.txt: SQ_LDS_BANK_CONFLICT 3.3e8 of SQ_LDS_IDX_ACTIVE 5.7e8 per sub-batch).
    // a DP cell as the wave keeps it: the E entries of pos / cnt are never read, counts are 16 bits, the four anchor
    // indices are interval numbers (< MAXM): 40 bytes instead of cp_cell's 64 -- with the size class at 96 intervals the
    // wave's record is 8.9 KB and four four-wave blocks (with their libm tables) fit a CU: 4 waves per SIMD
    typedef typename std::conditional<(MAXM <= 128),int8_t,int16_t>::type idx_t;
    struct alignas(8) cell_t
      { double dp, dhr; int32_t pos_[3]; uint16_t cnt_[3]; idx_t last_[4];
        __device__ __forceinline__ operator cp_cell() const
        { cp_cell c; c.dp = dp; c.dhr = dhr;
          c.pos[0] = 0; c.cnt[0] = 0;
          for (int k = 0; k < 3; k++) { c.pos[k+1] = pos_[k]; c.cnt[k+1] = cnt_[k]; }
          c.lastH = last_[0]; c.lastD = last_[1]; c.lastHbD = last_[2]; c.lastDbH = last_[3];
          return c;
        }
        __device__ __forceinline__ void set(const cp_cell &c)
        { dp = c.dp; dhr = c.dhr;
          for (int k = 0; k < 3; k++) { pos_[k] = c.pos[k+1]; cnt_[k] = (uint16_t)c.cnt[k+1]; }
          last_[0] = (idx_t)c.lastH; last_[1] = (idx_t)c.lastD; last_[2] = (idx_t)c.lastHbD; last_[3] = (idx_t)c.lastDbH;
        }
      };
    uint16_t b[G][MAXM+REL_ROW_PAD], e[G][MAXM+REL_ROW_PAD], ccb[G][MAXM+REL_ROW_PAD], cce[G][MAXM+REL_ROW_PAD];
    uint8_t  parent[G][2][MAXM];         // back-pointers of the 4 cells of an interval, 2 bits each; then the assignment
    eff_t    eff[G][2][MAXM];            // eff | RPOS flag
    cell_t   cell[G][2][2][4];           // [read][direction][buffer][state]
    double   tr[G][2][16+REL_TR_PAD];    // [read][direction][s*4+t]
    __device__ __forceinline__ int8_t *asgn(int g, int d) { return reinterpret_cast<int8_t *>(parent[g][d]); }
  };

template <int MAXM, int G>
struct rel_grp_view                      // path index -> hot fields through `eff`
  { const rel_grp_lds<MAXM,G> *S; int g, d;
    __device__ __forceinline__ cp_riv operator()(int k) const
    { int j = S->eff[g][d][k] & (rel_grp_lds<MAXM,G>::RPOS-1);
      cp_riv r; r.b = S->b[g][j]; r.e = S->e[g][j]; r.ccb = S->ccb[g][j]; r.cce = S->cce[g][j]; r.pe = 0.;
      return r;
    }
  };

template <int MAXM, int G>
struct rel_grp_rv
  { const rel_grp_lds<MAXM,G> *S; int g;
    __device__ __forceinline__ cp_riv operator()(int i) const
    { cp_riv r; r.b = S->b[g][i]; r.e = S->e[g][i]; r.ccb = S->ccb[g][i]; r.cce = S->cce[g][i]; r.pe = 0.;
      return r;
    }
  };

// One DP pass (_classify_rel, class_rel.c:515-614) for every (read, direction) whose lanes have
// active == true.  M differs per read group; the wave iterates to the largest.
// libm tables of a block in LDS (cp_libm.h): the DP step's chain is exp -> sum -> log, and from global memory each of the
// two look-ups was a cache miss in the middle of it (k_classify_rel_grp 2.2 -> 3.2 ms per sub-batch when ocml's table-free
// routines gave way to glibc's; with the tables on chip the look-up is an LDS read).
struct rel_libm_lds { uint64_t exp_tab[256]; double log_tab[256]; };
// A block of this kernel is WPB waves that share nothing but those tables: "sync" is the order of one wave's own LDS
// operations (the hardware keeps it; the compiler is told by the fence), never a barrier across the block's waves.
template <int WPB>
__device__ __forceinline__ void grp_sync()
{ if (WPB == 1) __syncthreads();
  else { __builtin_amdgcn_fence(__ATOMIC_RELEASE,"wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE,"wavefront"); }
}

// pe of reliable interval i: a field of the 48-byte record copies (stage API, rare classes) or of the compact records
#define REL_PE(i) (*reinterpret_cast<const double *>(pe0+(size_t)(i)*pe_stride))
template <int MAXM, int G, int WPB>
__device__ void rel_grp_pass(const cp_dev_params *P, rel_grp_lds<MAXM,G> &S, const rel_libm_lds &T, const char *pe0, int pe_stride, int M, int plen,
                             bool active, const int *COV)
{ CP_LDS_PTR(const uint64_t) xt = (CP_LDS_PTR(const uint64_t))T.exp_tab;
  CP_LDS_PTR(const double)   lt = (CP_LDS_PTR(const double))T.log_tab; // Lanes of a read: LD per direction (forward first).  With LD = 16 (G <= 2) lane (s,t) owns transition
  // s -> t; with LD = 8 (G = 4) lane (s,h) owns two, s -> H|D (a Skellam term, the expensive kind) and
  // s -> E|R (table look-ups), so that every lane of the wave has a Bessel evaluation to do.
  constexpr int L = WAVE/G, LD = (L >= 32) ? 16 : 8;
  const int lane = lane_id();
  const int g = lane/L, ql = lane%L;
  const int d = (ql / LD) & 1, F = (d == 0);
  const int ld = ql % LD;
  const int s = (LD == 16) ? (ld >> 2) : (ld >> 1);
  const int t1 = (LD == 16) ? (ld & 3) : -1;               // the single transition of a 16-lane direction
  const int t_sk  = (LD == 16) ? ((t1 == CP_HAPLO || t1 == CP_DIPLO) ? t1 : -1) : ((ld & 1) ? CP_DIPLO : CP_HAPLO);
  const int t_tab = (LD == 16) ? ((t1 == CP_ERROR || t1 == CP_REPEAT) ? t1 : -1) : ((ld & 1) ? CP_REPEAT : CP_ERROR);
  const bool in_grp = active && ql < 2*LD && M > 0;
  rel_grp_view<MAXM,G> view; view.S = &S; view.g = g; view.d = d;
  rel_grp_rv<MAXM,G> rv; rv.S = &S; rv.g = g;
  int maxM = in_grp ? M : 0;
  for (int o = 32; o > 0; o >>= 1)
    { int x = __shfl_xor(maxM,o); maxM = x > maxM ? x : maxM; }

  int i = F ? 0 : M-1;
  if (in_grp && ld < 4)                                    // init, class_rel.c:544-580
    { cp_riv I = rv(i);
      I.pe = REL_PE(i);
      cp_cell c;
      cp_rel_init_cell(P,ld,I,i,plen,F,COV,&c);
      S.cell[g][d][0][ld].set(c);
      S.tr[g][d][ld] = cp_exp_t(c.dp,xt);
      if (ld == 0)
        { S.parent[g][d][i] = 0xe4;                        // each state its own parent: 3,2,1,0
          S.eff[g][d][i] = (typename rel_grp_lds<MAXM,G>::eff_t)i;            // (rpos flag clear)
        }
    }
  grp_sync<WPB>();
  if (in_grp && ld < 4)                                    // :582-586
    { double psum = 0.;
      for (int x = 0; x < 4; x++)
        psum += S.tr[g][d][x];
      S.cell[g][d][0][ld].dp = cp_log_t(S.tr[g][d][ld]/psum,lt);
    }
  grp_sync<WPB>();

  int cur = 0;
  double pe_next = 0.;                                     // pe of the next interval (E lanes), one step ahead
  if (in_grp && t_tab == CP_ERROR && M > 1)
    pe_next = REL_PE(F ? 1 : M-2);
  for (int k = 1; k < maxM; k++)                           // _update, class_rel.c:279-513
    { const bool on = in_grp && k < M;
      const int i_pred = i;
      if (on) i = F ? k : M-1-k;
      cp_riv I; I.b = I.e = I.ccb = I.cce = 0; I.pe = 0.;
      double v_sk = 0., v_tab = 0.;
      // ---- :300-319: the 16 transitions of a (read, direction).  Every lane first works out which table
      //      cells (log-factorials, log lambda) and which Skellam arguments its transition(s) need; the
      //      table loads of all lanes are then issued together and the Skellam/Bessel evaluation is one
      //      call for all lanes that have one.
      bool live = false, is_sk = false, r_tab = false;
      int tb = 0, te = 0, tcb = 0, tce = 0, tcov = 0;
      int f0 = 0, f1 = 0, f2 = 0, li = 1, rk = 0, rn = 0;
      double lp_tab = -INFINITY;
      const double pe_now = pe_next;                       // fetched one step ahead
      if (on)
        { I = rv(i);
          if (t_tab == CP_ERROR && k+1 < M)
            pe_next = REL_PE(F ? k+1 : M-2-k);
          const auto &pr = S.cell[g][d][cur][s];
          live = pr.dp != -INFINITY;
          if (live)
            { if (t_tab == CP_ERROR)                       // logp_e, class_rel.c:158-170
                { I.pe = pe_now;
                  f0 = cp_check_cnt(I.ccb); f1 = cp_check_cnt(I.cce); li = COV[CP_ERROR];
                }
              else if (t_tab == CP_REPEAT)                 // logp_r, class_rel.c:172-211
                { const int beg_cnt = cp_beg_cnt(I,F), prc = pr.cnt_[CP_REPEAT-1];
                  if (beg_cnt < prc)
                    { r_tab = true;
                      rk = cp_check_cnt(beg_cnt); rn = cp_check_cnt(prc);
                      f0 = rn; f1 = rk; f2 = rn-rk;
                    }
                }
              if (t_sk >= 0)
                { is_sk = true;                            // logp_h / logp_d, class_rel.c:213-270
                  const int beg_pos = cp_beg_pos(I,F), beg_cnt = cp_beg_cnt(I,F);
                  if (t_sk == CP_HAPLO && pr.dhr == -INFINITY)
                    { tb = cp_pred(pr.pos_[CP_HAPLO-1],F); te = beg_pos; tcb = pr.cnt_[CP_HAPLO-1]; tce = beg_cnt; tcov = pr.cnt_[CP_HAPLO-1]; }
                  else
                    { tb = cp_pred(pr.pos_[CP_DIPLO-1],F); te = beg_pos; tcb = pr.cnt_[CP_DIPLO-1];
                      tce = (t_sk == CP_HAPLO) ? (int)(pr.dhr*beg_cnt) : beg_cnt;
                      tcov = pr.cnt_[CP_DIPLO-1];
                    }
                }
            }
        }
      { const double A = P->logfact[f0], B = P->logfact[f1], C = P->logfact[f2];
        const double D = (li >= 0 && li <= CP_MAX_KMER_CNT) ? P->logint[li] : cp_log((double)li);
        if (on && live)
          { if (t_tab == CP_ERROR)
              { double po = (f0 * D - li - A)+(f1 * D - li - B)+CP_E_PO_BASE;       // prob.c:33-39 twice
                lp_tab = (po > I.pe) ? po : I.pe;
              }
            else if (t_tab == CP_REPEAT)
              { const auto &pr = S.cell[g][d][cur][s];
                double l = r_tab ? (A - B - C + rk * P->r_lp + (rn-rk) * P->r_l1mp) : -INFINITY;   // prob.c:67-73
                if (!(l > CP_R_LOGP))
                  { int max_cc = I.ccb > I.cce ? I.ccb : I.cce;
                    if (max_cc >= COV[CP_REPEAT] || max_cc >= pr.cnt_[CP_REPEAT-1]) l = CP_R_LOGP;
                  }
                lp_tab = l;
              }
          }
      }
      if (is_sk)                                           // exp(logp_trans) straight from its table (cp_math.h): one exp per lane and step less
        v_sk = cp_exp_logp_trans(P,tb,te,tcb,tce,tcov,xt);
      if (on)
        { if (live) v_tab = cp_exp_t(lp_tab,xt);           // (is_sk implies live; a lane without a Skellam term keeps v_sk = 0 = exp(-inf))
          if (t_sk >= 0)  S.tr[g][d][s*4+t_sk]  = v_sk;
          if (t_tab >= 0) S.tr[g][d][s*4+t_tab] = v_tab;
        }
      grp_sync<WPB>();
      double nv_sk = 0., nv_tab = 0.;
      if (on)                                              // :320-336
        { double psum = 0.;
          for (int x = 0; x < 16; x++)
            psum += S.tr[g][d][x];
          if (psum == 0.)
            { if (t_tab == CP_ERROR) v_tab = 1.;
              psum = 4.;
            }
          if (t_sk >= 0)  nv_sk  = cp_log_t(v_sk/psum,lt);
          if (t_tab >= 0) nv_tab = cp_log_t(v_tab/psum,lt);
        }
      grp_sync<WPB>();
      if (on)
        { if (t_sk >= 0)  S.tr[g][d][s*4+t_sk]  = nv_sk;
          if (t_tab >= 0) S.tr[g][d][s*4+t_tab] = nv_tab;
        }
      grp_sync<WPB>();
      // :348-499: one lane per state.  The four state lanes of a (read, direction) are an aligned quad; what the reference
      // computes once per step and all four need -- "only R reachable" (four row maxima, :348-380) and the H->H / D->D
      // equalisation (two column maxima, :382-386) -- is split over them, a row and a column each, and joined through a
      // ballot, instead of every lane doing all six maxima (a quarter of this kernel's vector instructions).
      const bool st_lane = on && ld < 4;
      const int l16 = ld & 3;
      const double *tr = S.tr[g][d];
      double dp[4] = { 0., 0., 0., 0. };
      bool row_r = false;
      int  col_s = CP_N_STATE;
      double col_logp = -INFINITY;
      if (st_lane)
        { for (int x = 0; x < 4; x++) dp[x] = S.cell[g][d][cur][x].dp;
          { double best = -INFINITY;                       // best target of source l16 (cp_argmax_tr with s fixed)
            int maxt = CP_N_STATE;
            for (int x = 0; x < 4; x++)
              { const double logp = dp[l16]+tr[l16*4+x];
                if (best < logp) { best = logp; maxt = x; }
              }
            row_r = (maxt == CP_N_STATE || maxt == CP_REPEAT);
          }
          for (int x = 0; x < 4; x++)                      // best source of target l16 (t fixed), before the equalisation
            { const double logp = dp[x]+tr[x*4+l16];
              if (col_logp < logp) { col_logp = logp; col_s = x; }
            }
        }
      const int qsh = lane & ~3;                           // the quad's bits of a ballot
      const bool only_r = ((__ballot(st_lane && row_r) >> qsh) & 0xf) == 0xf;
      const bool equal  = ((__ballot(st_lane && ((l16 == CP_HAPLO || l16 == CP_DIPLO) && col_s == l16)) >> qsh) & 0xf)
                          == ((1u << CP_HAPLO) | (1u << CP_DIPLO));
      if (st_lane)
        { cp_cell c;
          int pv = l16;
          if (only_r)
            { c = (cp_cell)S.cell[g][d][cur][l16];
              cp_rel_only_r_cell(l16,i,&c);
              if (l16 == 0)
                S.eff[g][d][i] = (typename rel_grp_lds<MAXM,G>::eff_t)(S.eff[g][d][i_pred] | rel_grp_lds<MAXM,G>::RPOS);
            }
          else
            { double max_logp = col_logp;
              int max_s = col_s;
              if (equal && (l16 == CP_HAPLO || l16 == CP_DIPLO))                      // :382-386, then :391 again for this target
                { const double a = tr[CP_HAPLO*4+CP_HAPLO], bb = tr[CP_DIPLO*4+CP_DIPLO];
                  const double mn = a < bb ? a : bb;
                  max_logp = -INFINITY; max_s = CP_N_STATE;
                  for (int x = 0; x < 4; x++)
                    { const double w = (x == l16) ? mn : tr[x*4+l16];
                      const double logp = dp[x]+w;
                      if (max_logp < logp) { max_logp = logp; max_s = x; }
                    }
                }
              pv = (max_s == CP_N_STATE) ? l16 : max_s;
              if (l16 == 0)
                S.eff[g][d][i] = (typename rel_grp_lds<MAXM,G>::eff_t)i;
              cp_rel_target_cell(P,l16,i,I,F,COV,max_s,max_logp,&S.cell[g][d][cur][0],view,&c);
            }
          S.cell[g][d][cur^1][l16].set(c);
          const int l0 = lane-l16;                         // pack the four back-pointers into one byte
          const int pk = pv | (__shfl(pv,l0+1) << 2) | (__shfl(pv,l0+2) << 4) | (__shfl(pv,l0+3) << 6);
          if (l16 == 0)
            S.parent[g][d][i] = (uint8_t)pk;
        }
      grp_sync<WPB>();
      cur ^= 1;
    }
  // the buffer holding the last interval's cells: M-1 swaps happened for this read
  const int fin = (M > 0) ? ((M-1) & 1) : 0;
  if (in_grp && ld == 0)                                   // traceback, class_rel.c:606-613
    { double max_logp = -INFINITY;
      int st = CP_ERROR;
      for (int x = 0; x < 4; x++)
        if (max_logp < S.cell[g][d][fin][x].dp)
          { max_logp = S.cell[g][d][fin][x].dp;
            st = x;
          }
      // (the assignment of interval k replaces its back-pointer byte, which has just been read)
      int8_t *as = S.asgn(g,d);
      if (F)
        for (int k = M-1; k >= 0; k--)
          { const int pk = S.parent[g][d][k];
            as[k] = (S.eff[g][d][k] & rel_grp_lds<MAXM,G>::RPOS) ? (int8_t)CP_REPEAT : (int8_t)st;
            st = (pk >> (2*st)) & 3;
          }
      else
        for (int k = 0; k < M; k++)
          { const int pk = S.parent[g][d][k];
            as[k] = (S.eff[g][d][k] & rel_grp_lds<MAXM,G>::RPOS) ? (int8_t)CP_REPEAT : (int8_t)st;
            st = (pk >> (2*st)) & 3;
          }
    }
  grp_sync<WPB>();
}

// size classes of the grouped classify kernels (reads per wave / largest interval count)
// Chosen on the bench batch (per read: M <= 132, N <= 230; a read that falls into the next class runs alone
// there, and a lone wave was the tail of its stage).  With two reads per wave in the rel kernel and
// (MAXM; G,MAXN) = (128; 8,192) 6.68 ms per step, (192; 8,192) 6.48, (128; 4,256) 6.51, (192; 4,256) 6.31,
// (256; 4,256) 6.29.  Later, with four reads per wave in the rel kernel (8 lanes per direction, every lane
// with a Bessel term): 5.00 ms at MAXM 256, 4.77 at 192, 5.39 at 128.
// With the table of logp_trans values (cp_types.h) the Bessel term is a load and the balance moved again: whole step
// 70.7 ms at MAXM 192, 68.7 at 128 or 96 (less LDS per wave: 3 waves per SIMD instead of 2); G = 2 74.8, G = 8 89.7.
#ifndef REL_SMALL_G
#define REL_SMALL_G 4
#endif
// (REL_SMALL_MAXM: defined at the top of the file, k_find_wall needs it)
// (unrel, later, with K = 64/G/8 speculative update slots per read: G = 4 4.28 ms per step, G = 2 4.07, G = 1 4.04)
#ifndef UNREL_SMALL_G
#define UNREL_SMALL_G 2
#endif
// (UNREL_SMALL_MAXN: defined at the top of the file, k_find_wall needs it)
// the class above it holds a handful of reads per sub-batch, each a chain of several hundred updates: one read per wave
// (eight speculative slots) -- with two, the stage's main kernel waited 0.5 ms for this one at its end
#ifndef UNREL_BIG_G
#define UNREL_BIG_G 1
#endif
#ifndef REL_EXTRA_ATTR
#define REL_EXTRA_ATTR
#endif
// waves per block of the main size class (they share the block's copy of the libm tables, 4 KB, and nothing else)
#ifndef REL_SMALL_WPB
#define REL_SMALL_WPB 4
#endif
// (round 3: 4 waves per SIMD = 128 VGPRs with 32 of them spilled to scratch, against 167 and none at 3: 192 against 186
//  Gbases/s.  Round 4: four-wave blocks with the libm tables in LDS; with 64-byte cells and a size class of 128 a block was
//  52 KB, three to a CU = 3 waves per SIMD (149 VGPRs, nothing spilled): 194.4 Gbases/s; with the 40-byte cell_t and a
//  size class of 112 a block is 40.7 KB, four to a CU = 4 waves per SIMD at 128 VGPRs, 14 spilled: 198.6-199.7)
#ifndef REL_WAVES_PER_EU
#define REL_WAVES_PER_EU 4
#endif

template <int MINM, int MAXM, int G, int WPB, int COMPACT = 0>
__global__ void __launch_bounds__(WAVE*WPB) __attribute__((amdgpu_waves_per_eu(REL_WAVES_PER_EU))) REL_EXTRA_ATTR
k_classify_rel_grp(const cp_dev_params *__restrict__ P, const int64_t *__restrict__ prof_off, int nreads,
                   cp_intvl *__restrict__ intvl_all, cp_intvl *__restrict__ rintvl_all, const int32_t *__restrict__ relmap_all,
                   const int64_t *__restrict__ ioff, const int32_t *__restrict__ nrel,
                   int8_t *__restrict__ asgn_all, int64_t totalI, const int32_t *__restrict__ perm,
                   const cp_rrec *__restrict__ rrec_all, cp_soa soa)
{ // COMPACT (whole-path calls, main size class only): the reliable intervals come as compact records (rrec_all) and only
  // intvl[].asgn is written (the fw / bw assignments and rintvl[].asgn are what the stage API reads back)
  __shared__ rel_grp_lds<MAXM,G> Sw[WPB];
  __shared__ rel_libm_lds T;
  for (int k = threadIdx.x; k < 256; k += WAVE*WPB)
    { T.exp_tab[k] = cp_libm::exp_tab[k]; T.log_tab[k] = cp_libm::log_tab[k]; }
  __syncthreads();                                         // the only barrier across the block's waves
  const int wv = threadIdx.x/WAVE;
  rel_grp_lds<MAXM,G> &S = Sw[wv];
  constexpr int L = WAVE/G;
  const int lane = lane_id();
  const int g = lane/L, ql = lane%L;
  // A block takes the read groups blockIdx.x, blockIdx.x+gridDim.x, ...  `perm` lists the reads by decreasing M, so
  // the reads of a class with MINM > 0 are a prefix of it: such a class is launched with a small grid and a block
  // stops at its first group below the class (an empty class costs a few hundred blocks, not one block with this
  // kernel's LDS per read of the batch).
  for (int blk = blockIdx.x*WPB+wv; blk*G < nreads; blk += gridDim.x*WPB)
  {
  const int slot = blk*G+g;
  const int r = (slot < nreads) ? perm[slot] : nreads;
  int M = (r < nreads) ? nrel[r] : 0;
  const int rr = (r < nreads) ? r : 0;
  const int plen = (int)(prof_off[rr+1]-prof_off[rr]);
  // (the reads beyond GRP_MAX_PLEN k-mers sit in the top bin whatever their M -- order_bin -- and are no reason to stop)
  if (MINM > 0 && __ballot(M > MINM || (r < nreads && plen > GRP_MAX_PLEN)) == 0) break;
  if (M <= MINM || M > MAXM || plen > GRP_MAX_PLEN) M = 0; // other size classes / sequential kernel
  if (__ballot(M > 0) == 0) continue;
  const int64_t o = ioff[rr];
  cp_intvl *rintvl = rintvl_all+o;
  const cp_rrec *rrec = COMPACT ? rrec_all+o : nullptr;
  const char *pe0 = COMPACT ? reinterpret_cast<const char *>(&rrec[0].pe) : reinterpret_cast<const char *>(&rintvl[0].pe);
  constexpr int pe_stride = COMPACT ? (int)sizeof(cp_rrec) : (int)sizeof(cp_intvl);
  if (COMPACT)
    for (int k = ql; k < M; k += L)
      { const cp_rrec q = rrec[k];
        S.b[g][k] = (uint16_t)q.b; S.e[g][k] = (uint16_t)q.e; S.ccb[g][k] = q.ccb; S.cce[g][k] = q.cce;
      }
  else
    for (int k = ql; k < M; k += L)
      { S.b[g][k] = (uint16_t)rintvl[k].b; S.e[g][k] = (uint16_t)rintvl[k].e;
        S.ccb[g][k] = rintvl[k].ccb; S.cce[g][k] = rintvl[k].cce;
      }
  grp_sync<WPB>();

  constexpr int LD = (L >= 32) ? 16 : 8;                   // lanes per direction, see rel_grp_pass
  const int d = (ql / LD) & 1, F = (d == 0);
  const int leadlane = g*L+d*LD;
  const bool lead = (M > 0) && (ql < 2*LD) && ((ql % LD) == 0);
  int COV[4] = { P->cov[0], P->cov[1], P->cov[2], P->cov[3] };
  rel_grp_rv<MAXM,G> rv; rv.S = &S; rv.g = g;
  rel_grp_pass<MAXM,G,WPB>(P,S,T,pe0,pe_stride,M,plen,M > 0,COV);

  int rerun = 0;                                           // class_rel.c:629-650
  if (lead)
    rerun = cp_rel_post1(P,rv,M,F,S.asgn(g,d),COV) ? 1 : 0;
  rerun = __shfl(rerun,leadlane);
  COV[CP_HAPLO] = __shfl(COV[CP_HAPLO],leadlane);
  COV[CP_DIPLO] = __shfl(COV[CP_DIPLO],leadlane);
  if (M == 0 || ql >= 2*LD) rerun = 0;
#ifdef CP_PROF_WALK
  { const uint64_t rm = __ballot(lead && rerun != 0), lm_ = __ballot(lead);      // (read, direction) pairs that repeat the pass / all; waves that do
    if (lane == 0) { atomicAdd(&g_emit_prof[6],(unsigned long long)__popcll(rm) | ((unsigned long long)__popcll(lm_) << 32)); atomicAdd(&g_emit_prof[7],(rm ? 1ull : 0ull) | (1ull << 32)); }
  }
#endif
  if (__ballot(rerun != 0))
    rel_grp_pass<MAXM,G,WPB>(P,S,T,pe0,pe_stride,M,plen,rerun != 0,COV);
  double hdrr = 1.;
  if (lead)
    hdrr = cp_rel_post2(P,rv,M,F,S.asgn(g,d),rerun != 0);
  const double hf = __shfl(hdrr,g*L), hb = __shfl(hdrr,g*L+LD);
  grp_sync<WPB>();

  int take_bw = 0;                                         // class_rel.c:904-938
  if (M > 0 && ql == 0)
    { const int8_t *fw = S.asgn(g,0), *bw = S.asgn(g,1);
      bool eq = true;
      for (int i = 0; i < M; i++)
        if (fw[i] != bw[i]) { eq = false; break; }
      if (!eq)
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
    }
  take_bw = __shfl(take_bw,g*L);
  cp_intvl *intvl = intvl_all+o;
  const int32_t *relmap = relmap_all+o;
  int8_t *gfw = asgn_all+o, *gbw = asgn_all+totalI+o;
  if (COMPACT)
    for (int i = ql; i < M; i += L)                        // class_rel.c:949-960, the class only: into the two-byte array
      { const int8_t a = take_bw ? S.asgn(g,1)[i] : S.asgn(g,0)[i];            // (or the record, without the arrays)
        if (soa.c) soa.c[o+rrec[i].idx].asgn = a; else intvl[rrec[i].idx].asgn = a;
      }
  else
  for (int i = ql; i < M; i += L)                          // class_rel.c:949-960
    { int8_t f = S.asgn(g,0)[i], w = S.asgn(g,1)[i];
      int8_t a = take_bw ? w : f;
      gfw[i] = f; gbw[i] = w;
      rintvl[i].asgn = a;
      intvl[relmap[i]].asgn = a;
      if (soa.c) soa.c[o+relmap[i]].asgn = a;
    }
  grp_sync<WPB>();                                             // the LDS record is reused by the block's next group
  }
}

// ---------------------------------------------------------------------------------------------
//  k_classify_unrel_grp<MINN,MAXN,G>: class_unrel.c:248-300, G reads per wave (64/G >= 8 lanes per
//  read) for reads with MINN < N <= MAXN.
//
//  Interval fields (16-bit) sit in LDS; "nearest reliable interval of class s" (find_nn_u, a linear
//  scan in the reference) is a bit scan over two LDS bitsets kept up to date as classes change.  The
//  two sweeps stay sequential per read (every update reads its neighbours' current classes), but one
//  wave advances G reads at once and inside an update the eight expensive terms -- {H,D} x
//  {left,right} x {Skellam transition, binomial tail} (class_unrel.c:115-175) -- sit on eight lanes,
//  the Skellam ones of all reads meeting at one convergent Bessel call.
// ---------------------------------------------------------------------------------------------
template <int MAXN, int G>
struct unrel_grp_lds
  { static_assert(MAXN % 64 == 0,"the reliable-interval bitsets are whole 64-bit words"); uint16_t b[G][MAXN], e[G][MAXN], cb[G][MAXN], ce[G][MAXN], ccb[G][MAXN], cce[G][MAXN];
    int8_t   asgn[G][MAXN];
    uint8_t  isrel[G][MAXN];
    typedef typename std::conditional<(MAXN > 256),uint32_t,uint16_t>::type ord_t;
    ord_t    ord[G][MAXN];               // the non-fixed intervals in index order, their rank in the high bits (scratch of the sort)
    uint16_t key[G][MAXN+4];             // min(cb,ce) of those, padded with 0xffff to a multiple of four (the sort reads four per load); then: the order
    uint64_t rel[G][2][MAXN/64];         // [0] reliable & H, [1] reliable & D
    uint64_t need[G][MAXN/64];           // second sweep: intervals whose inputs changed since the first sweep evaluated them
    int16_t  mail_idx[G][8];             // interval, new class and order position of each speculative slot of a round
    int16_t  mail_pos[G][8];
    int8_t   mail_s[G][8];
  };

__device__ __forceinline__ int bits_left(const uint64_t *bits, int idx)       // nearest set bit < idx
{ int w = idx >> 6;
  uint64_t m = bits[w] & ((1ull << (idx & 63))-1);
  while (true)
    { if (m) return (w << 6)+63-__clzll((long long)m);
      if (--w < 0) return -1;
      m = bits[w];
    }
}
__device__ __forceinline__ int bits_right(const uint64_t *bits, int idx, int nwords)   // nearest set bit > idx
{ int w = idx >> 6;
  uint64_t m = bits[w] & ~((2ull << (idx & 63))-1);
  while (true)
    { if (m) return (w << 6)+__ffsll((long long)m)-1;
      if (++w >= nwords) return -1;
      m = bits[w];
    }
}

#ifndef UNREL_WAVES_PER_EU
#define UNREL_WAVES_PER_EU 4
#endif
template <int MINN, int MAXN, int G, int COMPACT = 0>
__global__ void __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(UNREL_WAVES_PER_EU)))
k_classify_unrel_grp(const cp_dev_params *__restrict__ P, int nreads, cp_intvl *__restrict__ intvl_all,
                     const int64_t *__restrict__ ioff, const int32_t *__restrict__ nintvl,
                     const int32_t *__restrict__ perm, const int64_t *__restrict__ prof_off, uint32_t *__restrict__ pcls_all,
                     cp_soa soa, int full_sweep2)
{ // COMPACT (whole-path calls, main size class): the intervals come as arrays (cp_soa) -- 16 bytes per interval in the load
  // loop, the three log-probabilities from their own array when an update asks for them.  Without COMPACT but with soa.c
  // (the rare class on whole-path calls): the 48-byte records, the classes of classify_rel from the two-byte array.
  __shared__ unrel_grp_lds<MAXN,G> S;
  constexpr int L = WAVE/G;
  static_assert(L >= 8 && (L % 8) == 0, "8 role lanes per read");
  const int lane = lane_id();
  const int g = lane/L, ql = lane%L, gbase = g*L;
  for (int blk = blockIdx.x; blk*G < nreads; blk += gridDim.x)     // groups of a block, as in k_classify_rel_grp
  {
  const int slot = blk*G+g;
  const int r = (slot < nreads) ? perm[slot] : nreads;
  int N = (r < nreads) ? nintvl[r] : 0;
  const int rr = (r < nreads) ? r : 0;
  const bool too_long = prof_off[rr+1]-prof_off[rr] > GRP_MAX_PLEN;  // (the last interval ends at plen)
  if (MINN > 0 && __ballot(N > MINN || (r < nreads && too_long)) == 0) break;     // (long reads: in the top bin whatever their N)
  const int64_t io = ioff[rr];
  cp_intvl *intvl = intvl_all+io;
  if (N <= MINN || N > MAXN || too_long) N = 0;            // other size classes / sequential kernel
  if (__ballot(N > 0) == 0) continue;
  const int nwords = (N+63) >> 6;
  const int REP = P->cov[CP_REPEAT];
  int maxN = N;
  for (int o = 32; o > 0; o >>= 1)
    { int x = __shfl_xor(maxN,o); maxN = x > maxN ? x : maxN; }

  for (int k = ql; k < MAXN/64; k += L)
    { S.rel[g][0][k] = 0; S.rel[g][1][k] = 0; S.need[g][k] = ~0ull; }
  wave_sync();
  for (int base = 0; base < maxN; base += L)               // load + bitsets, L intervals of every read per step
    { const int k = base+ql;
      bool h = false, dd = false;
      if (k < N && COMPACT)
        { const cp_ivA A = soa.a[io+k]; const cp_ivC Cc = soa.c[io+k];
          S.b[g][k] = (uint16_t)A.b; S.e[g][k] = (uint16_t)A.e;
          S.cb[g][k] = A.cb; S.ce[g][k] = A.ce; S.ccb[g][k] = A.ccb; S.cce[g][k] = A.cce;
          S.asgn[g][k] = Cc.asgn; S.isrel[g][k] = Cc.is_rel;
          h  = Cc.is_rel && Cc.asgn == CP_HAPLO;
          dd = Cc.is_rel && Cc.asgn == CP_DIPLO;
        }
      else if (k < N)
        { cp_intvl I = intvl[k];
          if (soa.c) I.asgn = soa.c[io+k].asgn;
          S.b[g][k] = (uint16_t)I.b; S.e[g][k] = (uint16_t)I.e;
          S.cb[g][k] = I.cb; S.ce[g][k] = I.ce; S.ccb[g][k] = I.ccb; S.cce[g][k] = I.cce;
          S.asgn[g][k] = I.asgn; S.isrel[g][k] = I.is_rel;
          h  = I.is_rel && I.asgn == CP_HAPLO;
          dd = I.is_rel && I.asgn == CP_DIPLO;
        }
      const uint64_t mh = __ballot(h) >> gbase, md = __ballot(dd) >> gbase;
      if (ql == 0 && base < N)                             // this read's L flags start at bit `base` (L | 64)
        { const uint64_t lm = (L == 64) ? ~0ull : ((1ull << L)-1);
          S.rel[g][0][base >> 6] |= (mh & lm) << (base & 63);
          S.rel[g][1][base >> 6] |= (md & lm) << (base & 63);
        }
    }
  wave_sync();
  // The order of the updates (class_unrel.c:246-258): a stable sort of ALL intervals by min(cb,ce), of which the loops then
  // skip the fixed ones.  The same order from less work (round 5): the non-fixed intervals first, in index order (ballot +
  // popcount, L at a time), then a stable rank sort of those alone -- rank = keys below mine + equal keys before me, the
  // keys in LDS padded to a multiple of four with 0xffff (above every count), four per 8-byte load -- nnf^2 comparisons
  // instead of N^2 with nnf about 0.4 N.  The rank goes into the high bits of the entry, then the entries go to their
  // places in the key array's storage, which holds the order from here on (S_ORD).
  constexpr int RSH = (MAXN > 256) ? 16 : 8;
  int nnf = 0;
  for (int base = 0; base < maxN; base += L)
    { const int i = base+ql;
      const bool keep = i < N && !(S.isrel[g][i] && (S.asgn[g][i] == CP_HAPLO || S.asgn[g][i] == CP_DIPLO));
      const uint64_t lm = (L == 64) ? ~0ull : ((1ull << L)-1);
      const uint64_t mk = (__ballot(keep) >> gbase) & lm;
      if (keep) S.ord[g][nnf+__popcll(mk & ((1ull << ql)-1))] = (typename unrel_grp_lds<MAXN,G>::ord_t)i;
      nnf += __popcll(mk);
    }
  wave_sync();
  for (int j = ql; j < ((nnf+3) & ~3); j += L)
    { uint16_t key = 0xffff;
      if (j < nnf) { const int k = S.ord[g][j]; key = S.cb[g][k] < S.ce[g][k] ? S.cb[g][k] : S.ce[g][k]; }
      S.key[g][j] = key;
    }
  wave_sync();
  for (int j = ql; j < nnf; j += L)
    { const unsigned key = S.key[g][j];
      int rank = 0;
      for (int m = 0; m < nnf; m += 4)
        { const uint2 q = *reinterpret_cast<const uint2 *>(&S.key[g][m]);
          const unsigned k0 = q.x & 0xffffu, k1 = q.x >> 16, k2 = q.y & 0xffffu, k3 = q.y >> 16;
          rank += (k0 < key || (k0 == key && m   < j)) ? 1 : 0;
          rank += (k1 < key || (k1 == key && m+1 < j)) ? 1 : 0;
          rank += (k2 < key || (k2 == key && m+2 < j)) ? 1 : 0;
          rank += (k3 < key || (k3 == key && m+3 < j)) ? 1 : 0;
        }
      S.ord[g][j] = (typename unrel_grp_lds<MAXN,G>::ord_t)((unsigned)S.ord[g][j] | ((unsigned)rank << RSH));
    }
  wave_sync();                                             // every rank is computed: the key array is free
  for (int j = ql; j < nnf; j += L)
    { const unsigned v = S.ord[g][j];
      S.key[g][v >> RSH] = (uint16_t)(v & ((1u << RSH)-1));
    }
#define S_ORD(p) ((int)S.key[g][p])
  nnf = __shfl(nnf,gbase);
  int maxNF = nnf;
  for (int o = 32; o > 0; o >>= 1)
    { int x = __shfl_xor(maxNF,o); maxNF = x > maxNF ? x : maxNF; }
  wave_sync();

  // role of this lane inside its slot: class (H,D) x side (L,R) x kind (0: max(er,sf), 1: sf_er).
  // The reference updates the non-fixed intervals one after the other (class_unrel.c:260-274), and an
  // update costs one Bessel recurrence of latency.  An update reads the neighbouring intervals' classes
  // and the reliable-H / reliable-D sets, which most updates leave untouched, so the K = L/8 groups of 8
  // lanes of a read take K consecutive updates at once (slots), and one lane then commits them in order:
  // a slot is committed only if no earlier slot of the round changed the class of one of its neighbours
  // or either set; otherwise the round ends there and the next one starts from that update.
  //
  // THE SECOND SWEEP ONLY RE-EVALUATES WHAT CAN HAVE CHANGED (round 5).  update_state(idx) is a function of the constant fields
  // of interval idx, of "asgn == H" / "asgn == D" of its two neighbours (class_unrel.c:126,145) and of the nearest
  // reliable-H / reliable-D interval on either side (find_nn_u, class_unrel.c:11-25) -- never of the interval's own
  // class.  S.need holds a bit per interval: set at the start, cleared when the interval is evaluated, set again by a
  // class change of j (old -> new) for j-1 and j+1 if old or new is H or D, and, if j is reliable and joins or leaves
  // the H or the D set, for every interval between the nearest members of that set on either side of j (both
  // included: the intervals whose nearest member on one side is or was j).  An interval whose bit is clear in the second
  // sweep would get the class it has: it is skipped -- on reads like the bench's 95 % of that sweep
  // (tests/test_unrel_memo.py runs this rule against the two plain sweeps on the host).  A round of the second sweep
  // looks at the next L order positions, gives its K slots the first K whose bit is set, and ends behind a slot that
  // changes a class (what lies behind may have been skipped on the strength of a bit that slot has just set).
  constexpr int K = L/8;
  const int sub = ql >> 3, role = ql & 7, sbase = gbase+sub*8;
  const int s2 = (role >> 2) & 1, side = (role >> 1) & 1, kind = role & 1;
  const int s = s2 ? CP_DIPLO : CP_HAPLO;
  const uint64_t glm = (L == 64) ? ~0ull : ((1ull << L)-1);
  int pass = 0, it = 0;                                    // class_unrel.c:260-274, position of this read
  bool done = (N == 0) || (nnf == 0);
#ifdef CP_PROF_WALK
  int prof_rounds[2] = { 0, 0 };
#endif
  while (__ballot(!done) != 0)
      { int mypos = it+sub;                                // order position of this slot's update
        int cover = (it+K < nnf) ? it+K : nnf;             // the round settles the positions [it, cover) if every slot commits
        if (pass == 1)
          { const int p = it+ql;
            bool dirty = false;
            if (!done && p < nnf)
              { const int k = S_ORD(p);
                dirty = full_sweep2 || ((S.need[g][k >> 6] >> (k & 63)) & 1);     // (full_sweep2: the A/B and test knob CLASSPRO_UNREL_SWEEP2=full)
              }
            const uint64_t m = (__ballot(dirty) >> gbase) & glm;
            uint64_t t = m;
            for (int j = 0; j < sub; j++) t &= t-1;
            mypos = t ? it+__ffsll((long long)t)-1 : nnf;
            uint64_t t2 = m;
#pragma unroll
            for (int j = 0; j < K; j++) t2 &= t2-1;
            cover = t2 ? it+__ffsll((long long)t2)-1 : ((it+L < nnf) ? it+L : nnf);
          }
        const bool act = !done && mypos < nnf;             // this slot has an update to do
        bool on = act;
        int idx = 0;
        if (act)
          idx = S_ORD(pass == 0 ? nnf-1-mypos : mypos);
        int snew = -1;
        bool do_sf = false, do_bin = false;
        int tb = 0, te = 0, tcb = 0, tce = 0, tcov = 0, est = 0, c = 0;
        int Icb = 0, Ice = 0, lD = -1, rD = -1;
        double er = -INFINITY;
        if (on)
          { const int Ib = S.b[g][idx], Ie = S.e[g][idx];
            Icb = S.cb[g][idx]; Ice = S.ce[g][idx];
            if ((Icb > Ice ? Icb : Ice) >= REP)            // update_state, class_unrel.c:195-199
              { snew = CP_REPEAT;
                on = false;
              }
            else
              { const int lH = bits_left(S.rel[g][0],idx), rH = bits_right(S.rel[g][0],idx,nwords);
                lD = bits_left(S.rel[g][1],idx); rD = bits_right(S.rel[g][1],idx,nwords);
                const int l_rel = s2 ? lD : lH, r_rel = s2 ? rD : rH;
                { if (kind == 0)                           // class_unrel.c:123-163, arguments only
                      { if (side == 0)
                          { if (idx-1 >= 0 && S.asgn[g][idx-1] == (int8_t)s) er = COMPACT ? soa.b[io+idx].peo_b : intvl[idx].peo_b;
                            if (l_rel != -1)
                              { do_sf = true; tb = S.e[g][l_rel]-1; te = Ib; tcb = S.cce[g][l_rel]; tce = Icb; tcov = tcb; }
                          }
                        else
                          { if (idx+1 < N && S.asgn[g][idx+1] == (int8_t)s) er = COMPACT ? soa.b[io+idx].peo_e : intvl[idx].peo_e;
                            if (r_rel != -1)
                              { do_sf = true; tb = Ie-1; te = S.b[g][r_rel]; tcb = Ice; tce = S.ccb[g][r_rel]; tcov = tce; }
                          }
                      }
                    else
                      { const int x = side ? Ie-1 : Ib;
                        c = side ? Ice : Icb;
                        // est_cov, class_unrel.c:27-51: this class's neighbours, else the other class's
                        int ll = l_rel, rr = r_rel, e1 = -1;
                        for (int rep = 0; rep < 2 && e1 < 0; rep++)
                          { if (ll != -1 && rr != -1)
                              e1 = (int)(uint16_t)cp_linear_interpolation(x,S.e[g][ll]-1,S.cce[g][ll],S.b[g][rr],S.ccb[g][rr]);
                            else if (ll != -1) e1 = S.cce[g][ll];
                            else if (rr != -1) e1 = S.ccb[g][rr];
                            if (e1 < 0)
                              { if (rep == 0) { ll = s2 ? lH : lD; rr = s2 ? rH : rD; }
                                else e1 = -2;
                              }
                            else if (rep == 1)
                              e1 = (e1 > 0) ? (((s == CP_HAPLO) ? e1/2 : e1*2) & 0xffff) : -2;
                          }
                        est = (e1 >= 0) ? e1 : P->cov[s];
                        do_bin = est >= c;
                      }
                  }
              }
          }
        // The Skellam term and the binomial term are table look-ups (cp_types.h: skel, uerr; the recurrence / the
        // tail sum only outside the tables), so the second sweep simply evaluates them again: the per-interval memo
        // in HBM that used to carry them from the first sweep cost more than that.
        double val = -INFINITY;
        if (do_sf)
          val = cp_logp_trans(P,tb,te,tcb,tce,tcov);
        if (do_bin)
          val = cp_logp_uerr(P,est,c);
        if (on)
          { if (kind == 0)
              val = (er > val) ? er : val;
            double v1 = __shfl_down(val,1);
            double sidev = (val > v1) ? val : v1;              // MAX(MAX(er,sf),sf_er) on lanes with kind 0
            double vr = __shfl_down(sidev,2);
            double logp_l = sidev, logp_r = vr;                // meaningful on role lanes 0 (H) and 4 (D)
            if (logp_l == -INFINITY && logp_r == -INFINITY)    // class_unrel.c:165-173
              { logp_l = cp_logp_poisson(P,Icb,P->cov[s]);
                logp_r = cp_logp_poisson(P,Ice,P->cov[s]);
              }
            else if (logp_l == -INFINITY) logp_l = logp_r;
            else if (logp_r == -INFINITY) logp_r = logp_l;
            const double logp_s = logp_l+logp_r;
            const double vH = __shfl(logp_s,sbase), vD = __shfl(logp_s,sbase+4);
            // E and R are table look-ups (class_unrel.c:53-113)
            const double pe = COMPACT ? soa.b[io+idx].pe : intvl[idx].pe;
            const double po = cp_logp_poisson(P,Icb,P->cov[CP_ERROR])+cp_logp_poisson(P,Ice,P->cov[CP_ERROR])+CP_E_PO_BASE;
            const double vE = (pe > po) ? pe : po;
            double vR;
            { int dcov_l, dcov_r;
              if (lD == -1 && rD == -1) dcov_l = dcov_r = P->cov[CP_DIPLO];
              else if (lD == -1)        dcov_l = dcov_r = S.cb[g][rD];
              else if (rD == -1)        dcov_l = dcov_r = S.ce[g][lD];
              else                      { dcov_l = S.ce[g][lD]; dcov_r = S.cb[g][rD]; }
              int rcov_l = (uint16_t)(P->dr_ratio*dcov_l);
              int rcov_r = (uint16_t)(P->dr_ratio*dcov_r);
              if (Icb >= rcov_l || Ice >= rcov_r)
                vR = CP_R_LOGP;
              else
                vR = cp_logp_binom_pre(P->logfact,Icb,rcov_l,P->r_lp,P->r_l1mp)+cp_logp_binom_pre(P->logfact,Ice,rcov_r,P->r_lp,P->r_l1mp);
            }
            double logpmax = -INFINITY;                        // class_unrel.c:208-218: E,R,H,D with strict <
            if (logpmax < vE) { logpmax = vE; snew = CP_ERROR; }
            if (logpmax < vR) { logpmax = vR; snew = CP_REPEAT; }
            if (logpmax < vH) { logpmax = vH; snew = CP_HAPLO; }
            if (logpmax < vD) { logpmax = vD; snew = CP_DIPLO; }
          }
        if (role == 0)
          { S.mail_idx[g][sub] = (int16_t)idx;
            S.mail_pos[g][sub] = (int16_t)mypos;
            S.mail_s[g][sub] = (int8_t)(act ? snew : -1);
          }
        wave_sync();                                           // every slot has read the old state
        int newit = cover;
        if (ql == 0 && !done)                                  // commit the round's slots in order
          { bool sets_changed = false;
            int changed[K], nch = 0;
            for (int j = 0; j < K; j++)
              { const int sj = S.mail_s[g][j], ij = S.mail_idx[g][j];
                if (sj < 0) break;                             // no update in this slot
                bool clash = sets_changed;
                for (int m = 0; m < nch; m++)
                  clash = clash || changed[m] == ij-1 || changed[m] == ij+1;
                if (clash) { newit = S.mail_pos[g][j]; break; }        // computed from a state an earlier slot changed
                const int old = S.asgn[g][ij];
                S.need[g][ij >> 6] &= ~(1ull << (ij & 63));    // evaluated on the state as it is now
                if (old != sj)
                  { const bool ohd = old == CP_HAPLO || old == CP_DIPLO, nhd = sj == CP_HAPLO || sj == CP_DIPLO;
                    if (ohd || nhd)                            // the neighbours' "asgn == s" tests
                      { if (ij > 0)   S.need[g][(ij-1) >> 6] |= 1ull << ((ij-1) & 63);
                        if (ij+1 < N) S.need[g][(ij+1) >> 6] |= 1ull << ((ij+1) & 63);
                      }
                    if (S.isrel[g][ij] && (ohd || nhd))
                      { const uint64_t bit = 1ull << (ij & 63);
                        for (int q = 0; q < 2; q++)            // ij leaves / joins the reliable-H (q = 0) or the reliable-D set
                          { const int sq = q ? CP_DIPLO : CP_HAPLO;
                            if (old != sq && sj != sq) continue;
                            int lo = bits_left(S.rel[g][q],ij), hi = bits_right(S.rel[g][q],ij,nwords);   // (ij's own bit plays no part)
                            if (lo < 0) lo = 0;
                            if (hi < 0) hi = N-1;
                            for (int w = lo >> 6; w <= (hi >> 6); w++)
                              { const uint64_t ml = (w == (lo >> 6)) ? (~0ull << (lo & 63)) : ~0ull;
                                const uint64_t mh = (w == (hi >> 6)) ? (~0ull >> (63-(hi & 63))) : ~0ull;
                                S.need[g][w] |= ml & mh;
                              }
                            if (old == sq) S.rel[g][q][ij >> 6] &= ~bit; else S.rel[g][q][ij >> 6] |= bit;
                            sets_changed = true;
                          }
                      }
                    S.asgn[g][ij] = (int8_t)sj;
                    if (ohd || nhd) changed[nch++] = ij;       // (a change among E, R and "no class yet" is no input of anybody's update)
                    if (pass == 1 && (ohd || nhd)) { newit = S.mail_pos[g][j]+1; break; }   // positions behind it were skipped on bits this change may have set
                  }
              }
          }
        newit = __shfl(newit,gbase);
#ifdef CP_PROF_WALK
        if (!done) prof_rounds[pass]++;
#endif
        if (!done)
          { it = newit;
            if (it >= nnf) { it = 0; pass++; if (pass == 2) done = true; }
          }
        wave_sync();
      }
#ifdef CP_PROF_WALK
  if (ql == 0 && N > 0 && MINN == 0)                       // rounds of the first / second sweep, non-fixed intervals, reads (main class)
    { atomicAdd(&g_phase_sum[9],(unsigned long long)prof_rounds[0]); atomicAdd(&g_phase_max[9],(unsigned long long)prof_rounds[1]);
      atomicAdd(&g_phase_sum[10],(unsigned long long)nnf); atomicAdd(&g_phase_sum[11],1ull); }
#endif
  if (pcls_all)                                            // whole-path calls: 4 bytes per interval for k_paint_labels instead of a
    for (int k = ql; k < N; k += L)                        // one-byte store into every 48-byte record (CP_PCLS)
      pcls_all[io+k] = CP_PCLS((int)S.e[g][k],S.asgn[g][k]);
  else
    for (int k = ql; k < N; k += L)
      intvl[k].asgn = S.asgn[g][k];
  wave_sync();                                             // the LDS record is reused by the block's next group
  }
}

// ---------------------------------------------------------------------------------------------
//  k_paint_labels: ClassPro.c:116-119 ('N' x (K-1)) and :265-271 (interval class per k-mer).
//  One wave per read.  The read's label string is cut into 16-byte pieces of the output (aligned to the buffer), a lane
//  each, 1 KB per step: the interval ends (in label coordinates, the 'N' prefix as interval 0) sit in LDS, a lane finds
//  the interval of its piece's first label by binary search, splats that class over the piece and patches what follows
//  an interval end inside it (one piece in ten holds one).  The form before -- interval by interval, every interval
//  painted by the whole wave -- issued 57 instructions per interval, 0.4 G of the pipeline's 4.5 G per sub-batch
//  (profiles/r03_sq_insts.txt), mostly scalar address arithmetic; this one about a quarter of that.
// ---------------------------------------------------------------------------------------------
#define PAINT_MAX 512
__device__ __forceinline__ unsigned cp_label_char(int a)
{ return (a == CP_ERROR) ? 'E' : (a == CP_REPEAT) ? 'R' : (a == CP_HAPLO) ? 'H' : (a == CP_DIPLO) ? 'D' : '?'; }

__global__ void __launch_bounds__(WAVE)
k_paint_labels(const cp_dev_params *__restrict__ P, const int64_t *__restrict__ seq_off, int nreads,
               const cp_intvl *__restrict__ intvl_all, const int64_t *__restrict__ ioff,
               const int32_t *__restrict__ nintvl, char *__restrict__ labels, const uint32_t *__restrict__ pcls_all)
{ const int r = blockIdx.x;
  if (r >= nreads) return;
  const int lane = lane_id();
  const int K = P->K;
  const uint32_t *pcls = pcls_all ? pcls_all+ioff[r] : nullptr;      // whole-path calls: (end, class) words (CP_PCLS)
  char *lab = labels+seq_off[r];
  const int rlen = (int)(seq_off[r+1]-seq_off[r]);
  const cp_intvl *intvl = intvl_all+ioff[r];
  const int N = nintvl[r];
  __shared__ int s_end[PAINT_MAX];                       // end of interval k (k = 0: the 'N' prefix), label coordinates
  __shared__ uint8_t s_chr[PAINT_MAX];
  if (N+1 <= PAINT_MAX)
    { if (lane == 0) { s_end[0] = K-1 < rlen ? K-1 : rlen; s_chr[0] = 'N'; }
      if (pcls)
        for (int k = lane; k < N; k += WAVE)
          { const uint32_t x = pcls[k]; s_end[k+1] = CP_PCLS_END(x)+(K-1); s_chr[k+1] = (uint8_t)cp_label_char(CP_PCLS_CLS(x)); }
      else
      for (int k = lane; k < N; k += WAVE)
        { s_end[k+1] = intvl[k].e+(K-1); s_chr[k+1] = (uint8_t)cp_label_char(intvl[k].asgn); }
      wave_sync();
      const int M = N+1;
      auto find = [&](int q) -> int                        // the interval of label q: the first one that ends beyond it
        { int lo = 0, hi = M-1;
          while (lo < hi) { const int m = (lo+hi) >> 1; if (s_end[m] > q) hi = m; else lo = m+1; }
          return lo;
        };
      const int head = (int)((16-((uintptr_t)lab & 15)) & 15);        // bytes before the first aligned piece
      const int h = head < rlen ? head : rlen;
      if (lane < h) lab[lane] = (char)s_chr[find(lane)];
      const int npiece = (rlen-h) >> 4;
      uint4 *dst = reinterpret_cast<uint4 *>(lab+h);
      for (int pc = lane; pc < npiece; pc += WAVE)
        { const int q0 = h+16*pc;
          int k = find(q0);
          unsigned c = s_chr[k]*0x01010101u;
          uint32_t w[4] = { c, c, c, c };
          int e = s_end[k];
          while (e < q0+16 && k+1 < M)                   // an interval ends inside the piece: the rest of it is the next one's
            { k++;
              const unsigned c2 = s_chr[k]*0x01010101u;
              const int o = e-q0;                        // first byte of the piece that belongs to interval k
#pragma unroll
              for (int d = 0; d < 4; d++)
                { const int lo = o-4*d;                  // first byte of dword d to replace
                  if (lo <= 0) w[d] = c2;
                  else if (lo < 4) { const uint32_t m = 0xffffffffu << (8*lo); w[d] = (w[d] & ~m) | (c2 & m); }
                }
              e = s_end[k];
            }
#ifdef PAINT_NT                                          // (A/B knob: non-temporal label stores, measured and dropped -- this kernel 1.18 -> 1.51 ms
                                                         //  per 4 Gbases in the pipeline, the step 0.4-0.9 % slower: profiles/r05_paint_nt_fw5_ab.txt)
          { const cp_u4v v = { w[0], w[1], w[2], w[3] };
            __builtin_nontemporal_store(v,reinterpret_cast<cp_u4v *>(dst+pc));
          }
#else
          dst[pc] = make_uint4(w[0],w[1],w[2],w[3]);
#endif
        }
      const int t0 = h+16*npiece;                        // the bytes behind the last aligned piece
      if (t0+lane < rlen) lab[t0+lane] = (char)s_chr[find(t0+lane)];
      return;
    }
  // a read with more intervals than the table holds: interval by interval, every interval painted by the whole wave
  for (int i = lane; i < K-1 && i < rlen; i += WAVE)
    lab[i] = 'N';
  char *pasgn = lab+(K-1);
  for (int base = 0; base < N; base += WAVE)             // 64 interval records per round of loads
    { int b_l = 0, e_l = 0, a_l = -1;
      if (base+lane < N)
        { if (pcls)
            { const uint32_t x = pcls[base+lane];
              b_l = base+lane ? CP_PCLS_END(pcls[base+lane-1]) : 0; e_l = CP_PCLS_END(x); a_l = CP_PCLS_CLS(x);
            }
          else { b_l = intvl[base+lane].b; e_l = intvl[base+lane].e; a_l = intvl[base+lane].asgn; }
        }
      const int nb = (N-base < WAVE) ? N-base : WAVE;
      for (int k = 0; k < nb; k++)
        { const int b = __builtin_amdgcn_readlane(b_l,k), e = __builtin_amdgcn_readlane(e_l,k);
          const int a = __builtin_amdgcn_readlane(a_l,k);
          const unsigned c = cp_label_char(a);
          char *p0 = pasgn+b, *p1 = pasgn+e;               // [p0,p1): bytes up to the first aligned word, words, rest
          char *w0 = (char *)(((uintptr_t)p0+3) & ~(uintptr_t)3), *w1 = (char *)((uintptr_t)p1 & ~(uintptr_t)3);
          if (w0 >= w1)
            { for (char *q = p0+lane; q < p1; q += WAVE) *q = (char)c; }
          else
            { if (p0+lane < w0) p0[lane] = (char)c;
              if (w1+lane < p1) w1[lane] = (char)c;
              uint32_t *ww = (uint32_t *)w0;
              const int nw = (int)((w1-w0) >> 2);
              for (int j = lane; j < nw; j += WAVE)
                ww[j] = c*0x01010101u;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
//  k_decode_profiles: Fetch_Profile's decoder (libfastk.c:1467-1534) on the device, one wave per read
//  (ship the ~0.27 B/base FASTK code strings over PCIe instead of
//  2 B/base of counts).  The reference walks the code byte by byte; here a wave takes 64 code bytes at
//  a time:
//    1. token starts.  A byte with bit 7 set is the head of a 2-byte token *if* it is a token start,
//       so start(i) = !(start(i-1) && bit7(i-1)): inside a run of bit-7 bytes starts alternate, and a
//       byte after a bit-7-clear byte is always a start.  One ballot + a count-leading-zeros gives
//       every lane the beginning of its bit-7 run, hence its parity; (bit7, start) of lane 63 carry
//       into the next 64 bytes.
//    2. each start lane decodes its token to (number of counts it emits, signed delta); two wave
//       scans turn these into (first output index, count value after the token).
//    3. expansion: output j is owned by the last lane whose first output index is <= j (binary search
//       in LDS), so runs are written as coalesced 128-byte stores whatever their lengths.
//  The reference adds 2-byte deltas modulo 2^15 and 6-bit deltas modulo 2^16; as long as no 6-bit
//  delta takes a count out of [0,32767] (FastK caps counts at 32767) every count is the prefix sum
//  of the deltas modulo 2^15, which is what the scan computes.  A 6-bit delta that leaves the range,
//  a code that ends inside a 2-byte token, or one that does not expand to exactly plen counts sets
//  err bit 4 (cp_workspace_check reports it) instead of producing counts.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(WAVE)
k_decode_profiles(const uint8_t *__restrict__ codes, const int64_t *__restrict__ code_off,
                  const int64_t *__restrict__ prof_off, int nreads, uint16_t *__restrict__ prof,
                  int32_t *__restrict__ err)
{ const int r = blockIdx.x, lane = threadIdx.x;
  __shared__ int s_o[WAVE], s_v[WAVE];
  const uint8_t *c = codes+code_off[r];
  const int clen = (int)(code_off[r+1]-code_off[r]);
  uint16_t *out = prof+prof_off[r];
  const int cap = (int)(prof_off[r+1]-prof_off[r]);
  int npos = 0, dcur = 0, bad = 0;
  bool prev_h = false, prev_s = false;
  for (int base = 0; base < clen; base += WAVE)
    { const int g = base+lane;
      const bool valid = g < clen;
      const int x  = valid ? c[g] : 0;
      const int nx = (g+1 < clen) ? c[g+1] : 0;
      const bool h = valid && (x & 0x80);
      const uint64_t hm = __ballot(h);
      const uint64_t zeros = ~hm & (((uint64_t)1 << lane)-1);      // bit-7-clear bytes below this lane
      bool st;
      if (zeros)
        st = ((lane-(64-__clzll((long long)zeros))) & 1) == 0;     // run of bit-7 bytes starts right after the nearest one
      else
        { const bool s0 = (base == 0 || !prev_h) ? true : !prev_s;  // run reaches back into the previous 64 bytes
          st = s0 ^ ((lane & 1) != 0);
        }
      st = st && valid;
      const uint64_t sm = __ballot(st);
      prev_h = (hm >> 63) & 1;
      prev_s = (sm >> 63) & 1;

      int contrib = 0, delta = 0;
      if (st)
        { if (g == 0)
            { delta = h ? (((x & 0x7f) << 8) | nx) : x; contrib = 1; }
          else if ((x & 0xc0) == 0)
            contrib = x;                                           // 00rrrrrr: run of the current count
          else
            { contrib = 1;
              if (h)                                               // 1sxxxxxx yyyyyyyy: 15-bit two's complement delta
                { const int v = ((x & 0x3f) << 8) | nx;
                  delta = (x & 0x40) ? v-16384 : v;
                }
              else if (x & 0x20) delta = (x & 0x1f)-32;            // 011xxxxx
              else               delta = x & 0x1f;                 // 010xxxxx
            }
          if (h && g+1 >= clen) bad = 1;
        }
      int o = contrib, v = delta;
      for (int k = 1; k < WAVE; k <<= 1)
        { const int a = __shfl_up(o,k), b = __shfl_up(v,k);
          if (lane >= k) { o += a; v += b; }
        }
      const int T = __shfl(o,WAVE-1), D = __shfl(v,WAVE-1);
      if (st && g > 0 && !h && (x & 0x40))                         // a 6-bit delta must not leave [0,32767]
        { const int t = ((dcur+v-delta) & 0x7fff)+delta;
          if (t < 0 || t > 32767) bad = 1;
        }
      v = (dcur+v) & 0x7fff;
      wave_sync();
      s_o[lane] = npos+o-contrib;
      s_v[lane] = v;
      wave_sync();
      for (int j = lane; j < T; j += WAVE)
        { const int target = npos+j;
          int lo = 0;
          #pragma unroll
          for (int step = WAVE/2; step > 0; step >>= 1)
            if (s_o[lo+step] <= target) lo += step;
          if (target < cap) out[target] = (uint16_t)s_v[lo];
        }
      npos += T;
      dcur = (dcur+D) & 0x7fff;
    }
  if (__ballot(bad) != 0 || npos != cap)
    { if (lane == 0) atomicOr(err,4); }
}

// ---------------------------------------------------------------------------------------------
//  k_unpack_bases: Dazzler 2-bit bases (4 per byte, first base in the top bits: Compress_Read /
//  Uncompress_Read, DB.c:319-363) -> the upper-case characters Load_Read(...,2) hands ClassPro
//  (a database ships 0.25 B/base over PCIe instead of 1).  One block per read,
//  one packed byte (four characters) per thread and step.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_unpack_bases(const uint8_t *__restrict__ packed, const int64_t *__restrict__ pack_off,
               const int64_t *__restrict__ seq_off, int nreads, char *__restrict__ seq)
{ // a thread turns four packed bytes (one 4-byte load) into sixteen letters (one 16-byte store; byte by byte -- one byte in,
  // four single-byte stores out -- this kernel ran at 0.3 TB/s, a millisecond per 250-Mbase batch of the PCIe pipeline)
  __shared__ uint32_t s_tab[256];                        // the four letters of a packed byte, first base in the low byte
  { const unsigned b = threadIdx.x;
    const char *L = "ACGT";
    s_tab[b] = (unsigned)L[(b >> 6) & 3] | ((unsigned)L[(b >> 4) & 3] << 8) | ((unsigned)L[(b >> 2) & 3] << 16) | ((unsigned)L[b & 3] << 24);
  }
  __syncthreads();
  const int r = blockIdx.x;
  if (r >= nreads) return;
  const uint8_t *src = packed+pack_off[r];
  char *dst = seq+seq_off[r];
  const int rlen = (int)(seq_off[r+1]-seq_off[r]);
  const int clen = (rlen+3) >> 2;
  for (int k = 4*threadIdx.x; k < clen; k += 4*blockDim.x)
    { if (k+4 <= clen && 4*k+16 <= rlen)
        { const unsigned w = reinterpret_cast<const cp_u8x4 *>(src+k)->v;
          cp_u8x16 o;
          o.v[0] = s_tab[w & 0xff]; o.v[1] = s_tab[(w >> 8) & 0xff]; o.v[2] = s_tab[(w >> 16) & 0xff]; o.v[3] = s_tab[w >> 24];
          *reinterpret_cast<cp_u8x16 *>(dst+4*k) = o;
        }
      else
        for (int kk = k; kk < k+4 && kk < clen; kk++)
          { const unsigned b = src[kk];
            const int p = 4*kk;
            for (int q = 0; q < 4; q++)
              if (p+q < rlen)
                dst[p+q] = "ACGT"[(b >> (6-2*q)) & 3];
          }
    }
}

// ---------------------------------------------------------------------------------------------
//  k_label_runs: the label string of a read as RUNS -- (end, class) per maximal stretch of one class, the K-1 'N' in
//  front implicit.  A read has ~100 intervals and fewer runs, so labels cross PCIe at ~0.05 B/base instead of 1 (characters)
//  or 0.25 (2-bit codes), and the painted string (1 B/base of HBM writes) is not needed at all: the caller stops the
//  pipeline at the classification stage.  One wave per read, a lane per interval, ordered compaction by ballot; read r's
//  runs go to index ioff[r] .. of `ends` / `cls` (the capacity offsets of the interval arrays: runs <= intervals).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(WAVE)
k_label_runs(const cp_dev_params *__restrict__ P, int nreads, const cp_intvl *__restrict__ intvl_all, const int64_t *__restrict__ ioff,
             const int32_t *__restrict__ nintvl, int32_t *__restrict__ ends, uint8_t *__restrict__ cls, int32_t *__restrict__ nruns,
             const uint32_t *__restrict__ pcls_all)
{ const int r = blockIdx.x;
  if (r >= nreads) return;
  const int lane = lane_id();
  const int Km1 = P->K-1;
  const int64_t o = ioff[r];
  const cp_intvl *intvl = intvl_all+o;
  const int N = nintvl[r];
  int nr = 0;
  for (int base = 0; base < N; base += WAVE)
    { const int k = base+lane;
      int e = 0; unsigned c = 0; bool last = false;
      if (k < N && pcls_all)                               // after a whole-path call the classes are in the (end, class) words
        { const uint32_t x = pcls_all[o+k];
          e = CP_PCLS_END(x)+Km1; c = cp_label_char(CP_PCLS_CLS(x));
          last = (k == N-1) || cp_label_char(CP_PCLS_CLS(pcls_all[o+k+1])) != c;
        }
      else if (k < N)
        { e = intvl[k].e+Km1; c = cp_label_char(intvl[k].asgn);
          last = (k == N-1) || cp_label_char(intvl[k+1].asgn) != c;
        }
      const uint64_t m = __ballot(last);
      if (last)
        { const int64_t at = o+nr+__popcll(m & ((1ull << lane)-1));
          ends[at] = e; cls[at] = (uint8_t)c;
        }
      nr += __popcll(m);
    }
  if (lane == 0) nruns[r] = nr;
}

// ---------------------------------------------------------------------------------------------
//  k_pack_labels: the label string of a read as 2-bit codes, four per byte, first label in the top bits --
//  ctos (const.c:21-36: N, E -> 0, R -> 1, H -> 2, D -> 3) followed by Compress_Read (gene_core.c:235-254), i.e. the
//  read's payload of the .class.data track (ClassPro.c:291-300): labels then cross PCIe at 0.25 B/base instead of 1.
//  One block per read, one output byte (four labels, one unaligned 4-byte load) per thread and step.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned cp_label_code(unsigned c) { return c == 'R' ? 1u : c == 'H' ? 2u : c == 'D' ? 3u : 0u; }
__global__ void __launch_bounds__(256)
k_pack_labels(const char *__restrict__ labels, const int64_t *__restrict__ seq_off, const int64_t *__restrict__ pack_off,
              int nreads, uint8_t *__restrict__ packed)
{ const int r = blockIdx.x;
  if (r >= nreads) return;
  const char *src = labels+seq_off[r];
  uint8_t *dst = packed+pack_off[r];
  const int rlen = (int)(seq_off[r+1]-seq_off[r]);
  const int clen = (rlen+3) >> 2;
  auto four = [](unsigned w) -> unsigned
    { return (cp_label_code(w & 0xff) << 6) | (cp_label_code((w >> 8) & 0xff) << 4) | (cp_label_code((w >> 16) & 0xff) << 2) | cp_label_code(w >> 24); };
  for (int k = 4*threadIdx.x; k < clen; k += 4*blockDim.x)       // sixteen labels in (one 16-byte load), four bytes out (one store)
    { if (k+4 <= clen && 4*k+16 <= rlen)
        { const cp_u8x16 x = *reinterpret_cast<const cp_u8x16 *>(src+4*k);
          cp_u8x4 o;
          o.v = four(x.v[0]) | (four(x.v[1]) << 8) | (four(x.v[2]) << 16) | (four(x.v[3]) << 24);
          *reinterpret_cast<cp_u8x4 *>(dst+k) = o;
        }
      else
        for (int kk = k; kk < k+4 && kk < clen; kk++)
          { const int p = 4*kk;
            unsigned b = 0;
            for (int q = 0; q < 4; q++)
              if (p+q < rlen) b |= cp_label_code((unsigned char)src[p+q]) << (6-2*q);
            dst[kk] = (uint8_t)b;
          }
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


// ---------------------------------------------------------------------------------------------
//  k_skellam_table: the values of logp_trans (util.c:35-44 -> prob.c:41-44 -> bessel.c:478-521) for every pair
//  (cov*|e-b| <= cdmax, |ce-cb| <= kmax), by the function the kernels would otherwise run on the spot.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_skellam_table(const cp_dev_params *__restrict__ P, double *__restrict__ tab, int kmax, long long cdmax)
{ const long long n = (cdmax+1)*(kmax+1);
  for (long long i = (long long)blockIdx.x*blockDim.x+threadIdx.x; i < n; i += (long long)gridDim.x*blockDim.x)
    { const long long cd = i/(kmax+1);
      const int k = (int)(i-cd*(kmax+1));
      tab[i] = cp_logp_trans_calc(P,k,(double)cd);
    }
}

// the table of the walk's P(error in) values (cp_types.h), by the function the kernels would otherwise run
__global__ void __launch_bounds__(256)
k_eskel_table(const double *__restrict__ skel, double *__restrict__ tab, long long n)      // exp of the logp_trans table, by the kernels' own exp
{ for (long long i = (long long)blockIdx.x*blockDim.x+threadIdx.x; i < n; i += (long long)gridDim.x*blockDim.x)
    tab[i] = cp_exp(skel[i]);
}

__global__ void __launch_bounds__(256)
k_pe_table(const cp_dev_params *__restrict__ P, double *__restrict__ tab, int cmax)
{ const long long per = (long long)(cmax+1)*(cmax+1), n = 2*63*per;
  for (long long i = (long long)blockIdx.x*blockDim.x+threadIdx.x; i < n; i += (long long)gridDim.x*blockDim.x)
    { const int et = (int)(i/per);
      const long long q = i-(long long)et*per;
      const int cout = (int)(q/(cmax+1)), cin = (int)(q-(long long)cout*(cmax+1));
      const int e = et/63, tl = et%63;
      tab[i] = (cin <= cout) ? cp_p_errorin_calc(P,e,tl/21,tl%21,cout,cin) : 0.;
    }
}

// the table of classify_unrel's binomial-test logs (cp_types.h), by the function the kernels would otherwise run
__global__ void __launch_bounds__(256)
k_uerr_table(const cp_dev_params *__restrict__ P, double *__restrict__ tab, int emax)
{ const long long n = (long long)(emax+1)*(emax+1);
  for (long long i = (long long)blockIdx.x*blockDim.x+threadIdx.x; i < n; i += (long long)gridDim.x*blockDim.x)
    { const int est = (int)(i/(emax+1)), c = (int)(i-(long long)est*(emax+1));
      tab[i] = (c <= est) ? cp_logp_uerr_calc(P,est,c) : 0.;
    }
}

// ---------------------------------------------------------------------------------------------
//  -s seed path (src/seed.c:966-1032): cp_seed_wave.h, one wave per read.
//  k_seed_caps: per read, the number of count runs and of label runs of its k-mers: the scratch a read's
//  seed selection needs.  Segments of a selection: every count run starts at most one, every change of the valid
//  flag inside a count run one more; the flag changes with the label and, for the repeat selection, at k-mers that
//  already carry an H/D seed (a handful per 1000 k-mers: a taken segment masks 1000 k-mers either side of it), so
//  count runs + label runs + plen/64 + 6 holds them with a wide margin; a read that still needs more is reported
//  (error bit 4), never written past.  .rep intervals <= label runs / 2 + 2.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(WAVE)
k_seed_caps(const uint16_t *__restrict__ prof, const int64_t *__restrict__ prof_off, const char *__restrict__ labels,
            const int64_t *__restrict__ seq_off, int K, int nreads, int64_t *__restrict__ scap, int64_t *__restrict__ rcap,
            int32_t *__restrict__ plen_key)
{ const int r = blockIdx.x;
  if (r >= nreads) return;
  const int lane = lane_id();
  const int64_t po = prof_off[r];
  const int plen = (int)(prof_off[r+1]-po);
  const uint16_t *p = prof+po;
  const char *c = labels+seq_off[r]+(K-1);
  int nc = 0, nl = 0;
  // eight positions per lane and step from one 16-byte and one 8-byte load (the addresses are only 2- / 1-byte aligned;
  // gfx9 global loads take unaligned addresses) plus the element before them; position by position this pass over the
  // counts and labels of the whole batch ran at 3.6 TB/s of narrow loads
  int i0 = 1;
  for (; i0+8*WAVE <= plen; i0 += 8*WAVE)
    { const int i = i0+8*lane;
      const cp_u16x8 x = cp_load_u16x8(p,i);
      struct __attribute__((packed, aligned(1))) u8x8 { uint32_t x, y; };
      const u8x8 y = *reinterpret_cast<const u8x8 *>(c+i);
      unsigned pc = p[i-1], pl = (unsigned char)c[i-1];
#pragma unroll
      for (int q = 0; q < 8; q++)
        { const unsigned cc = x.v[q], cl = ((q < 4 ? y.x : y.y) >> (8*(q & 3))) & 0xff;
          nc += (cc != pc) ? 1 : 0; nl += (cl != pl) ? 1 : 0;
          pc = cc; pl = cl;
        }
    }
  for (int i = i0+lane; i < plen; i += WAVE)
    { nc += (p[i] != p[i-1]) ? 1 : 0;
      nl += (c[i] != c[i-1]) ? 1 : 0;
    }
  for (int o = 32; o > 0; o >>= 1) { nc += __shfl_xor(nc,o); nl += __shfl_xor(nl,o); }
  if (lane == 0)
    { scap[r] = (int64_t)nc+nl+6+plen/64;
      rcap[r] = (int64_t)(nl+1)/2+2;
      plen_key[r] = plen;
    }
}

// 'N' over the first K-1 bases of every read (the k-mer part is prefilled with 'E' by a memset)
__global__ void __launch_bounds__(WAVE)
k_seed_prefix(const int64_t *__restrict__ seq_off, int K, int nreads, char *__restrict__ seeds)
{ const int r = blockIdx.x;
  if (r >= nreads) return;
  char *out = seeds+seq_off[r];
  for (int i = lane_id(); i < K-1; i += WAVE) out[i] = 'N';
}

#ifdef CP_SEED_PROF
__device__ unsigned long long g_seed_prof[8];
#endif
#if defined(CP_SEED_DEBUG) || defined(CP_SEED_DEBUG_TAKES)
__device__ int4 g_seed_dbg[8192];
__device__ int g_seed_dbg_n[4];
__device__ int g_seed_dbg_read;
#endif
#include "cp_seed_wave.h"
// One wave per read (cp_seed_wave.h).
#ifndef SW_WAVES
#define SW_WAVES 6
#endif
__global__ void __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(SW_WAVES)))
k_find_seeds(const char *__restrict__ seq, const int64_t *__restrict__ seq_off, const uint16_t *__restrict__ prof,
             const int64_t *__restrict__ prof_off, const char *__restrict__ labels, int K, int nreads,
             const int64_t *__restrict__ soff, const int64_t *__restrict__ roff, const int32_t *__restrict__ perm,
             int32_t *__restrict__ seg, int32_t *__restrict__ aux, int32_t *__restrict__ mi,
             int32_t *__restrict__ rep_pairs, int32_t *__restrict__ rep_cnt, char *__restrict__ seeds, int32_t *__restrict__ err,
             int64_t totalS)
{ if ((int)blockIdx.x >= nreads) return;
  const int r = perm[blockIdx.x], lane = (int)threadIdx.x;
  const int64_t so = seq_off[r], po = prof_off[r], o = soff[r];
  cp_seedw_read R;
  R.seq = seq+so; R.cls = labels+so+(K-1); R.prof = prof+po; R.state = seeds+so+(K-1);
  R.plen = (int)(prof_off[r+1]-po); R.K = K; R.cap = (int)(soff[r+1]-o);
  R.rec = (int4 *)seg+o; R.orec = (int4 *)seg+totalS+o; R.tmp = aux+o;
  R.gmi_b = mi+2*(o+3*(int64_t)r); R.gmi_e = R.gmi_b+R.cap+3;
  R.rep_pairs = rep_pairs+2*roff[r]; R.rep_cap = (int)(roff[r+1]-roff[r]); R.err = err;
  R.dbg_read = 0;
#if defined(CP_SEED_DEBUG) || defined(CP_SEED_DEBUG_TAKES)
  R.dbg_read = (r == g_seed_dbg_read) ? 1 : 0;
#endif
#ifdef CP_SEED_PROF
  unsigned long long sw_t[8] = {0,0,0,0,0,0,0,0}, sw_last = wall_clock64();
  const int n = sw_find_seeds(R,lane,sw_t,sw_last);
  if (lane == 0) for (int k = 0; k < 8; k++) atomicAdd(&g_seed_prof[k],sw_t[k]);
#else
  const int n = sw_find_seeds(R,lane);
#endif
  if (lane == 0) rep_cnt[r] = n < R.rep_cap ? n : R.rep_cap;
}
